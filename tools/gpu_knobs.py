"""Same-process A/B over environment knobs the library reads at launch.
usage: python tools/gpu_knobs.py cfg S "CPF_KSPEC=0,CPF_PSTORE=0" "CPF_KSPEC=1,CPF_PSTORE=0" ...   (min of 5 launches each, interleaved)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, '.')
from cpflow_amd.engine import TemplateEngine

CCZ = np.diag([1,1,1,1,1,1,1,-1]).astype(complex)
TOFF3 = np.eye(8)[[0,1,2,3,4,5,7,6]].astype(complex)
CFG = {'cfg2': ([[0,1],[1,2]], 12, CCZ), 'cfg3': ([[0,1],[0,2],[1,2]], 16, TOFF3)}
layer, K, tgt = CFG[sys.argv[1]]
S = int(sys.argv[2])
combos = sys.argv[3:]
T = int(os.environ.get('T', '2000'))
eng = TemplateEngine(layer, K, tgt, 'xyz', num_gd_iterations=T)
init = eng.initial_angles(0, S)
res = {c: [] for c in combos}; var = {}; stats = {}
for rep in range(5):
    for c in combos:
        for kv in c.split(','):
            k, v = kv.split('='); os.environ[k] = v
        out = eng.optimize(init); torch.cuda.synchronize()
        res[c].append(eng.last_kernel_ms()); var[c] = eng.last_variant()
        stats[c] = (int(out['passed'].sum()), int(out['cz'][out['passed']].min()))
for c in combos:
    print(f'{sys.argv[1]} S={S} {c:40s} min {min(res[c]):.3f} ms  {S*T/min(res[c])*1e3:.3e}/s  [{var[c]}] passed/minCZ {stats[c]}', flush=True)
