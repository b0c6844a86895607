"""Same-process A/B of the two merged-run variants at several S (CPF_MSTORE is read per launch)."""
import os, sys, numpy as np, torch
sys.path.insert(0, '.')
from cpflow_amd.engine import TemplateEngine
CCZ = np.diag([1,1,1,1,1,1,1,-1]).astype(complex)
TOFF3 = np.eye(8)[[0,1,2,3,4,5,7,6]].astype(complex)
for name, layer, K, tgt in (('cfg2', [[0,1],[1,2]], 12, CCZ), ('cfg3', [[0,1],[0,2],[1,2]], 16, TOFF3)):
    eng = TemplateEngine(layer, K, tgt, 'xyz')
    for S in (1024, 4096, 16384):
        init = eng.initial_angles(0, S)
        res = {}
        for rep in range(3):
            for m in ('1', '0'):
                os.environ['CPF_MSTORE'] = m
                eng.optimize(init); ms = eng.last_kernel_ms()
                res.setdefault(m, []).append(ms)
        print(name, 'S', S, 'store %.2f ms' % min(res['1']), 'rebuild %.2f ms' % min(res['0']), flush=True)
