"""Stored-state (CPF_PSTORE=1) vs un-applying (CPF_PSTORE=0) gradient halves of the two-ended 3-qubit kernel:
agreement on short runs, timing over S.  usage: python tools/gpu_pstore.py [cfg2|cfg3] (CPF_SPLIT=1 is forced)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, '.')
from cpflow_amd.engine import TemplateEngine

CCZ = np.diag([1,1,1,1,1,1,1,-1]).astype(complex)
TOFF3 = np.eye(8)[[0,1,2,3,4,5,7,6]].astype(complex)
CFG = {'cfg2': ([[0,1],[1,2]], 12, CCZ, [512, 1024, 2048, 4096, 16384]),
       'cfg3': ([[0,1],[0,2],[1,2]], 16, TOFF3, [512, 1024, 4096])}
name = (sys.argv[1:] or ['cfg2'])[0]
layer, K, tgt, Ss = CFG[name]
os.environ['CPF_SPLIT'] = '1'

def run(ps, S, T, rot='xyz'):
    os.environ['CPF_PSTORE'] = str(ps)
    eng = TemplateEngine(layer, K, tgt, rot, num_gd_iterations=T)
    init = eng.initial_angles(0, S)
    out = eng.optimize(init); torch.cuda.synchronize()
    ms = []
    for _ in range(3 if T >= 1000 else 1):
        out = eng.optimize(init); ms.append(eng.last_kernel_ms())
    return {k: v.cpu().numpy() for k, v in out.items() if torch.is_tensor(v)}, min(ms), eng.last_variant()

for rot in ('xyz', 'xz'):
    for T in (0, 1, 2, 10, 50):
        a, _, va = run(1, 64, T, rot)
        b, _, vb = run(0, 64, T, rot)
        d = {k: float(np.max(np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)))) for k in ('regloss', 'loss', 'reg', 'params')}
        print(f'{rot} T={T:3d} [{va}] vs [{vb}] max|d|', {k: f'{v:.2e}' for k, v in d.items()}, 'cz equal', bool((a['cz'] == b['cz']).all()), flush=True)
T = int(os.environ.get('T', '2000'))
for S in Ss:
    a, ms1, va = run(1, S, T)
    b, ms0, vb = run(0, S, T)
    print(f'{name} S={S}: states {ms1:.2f} ms  unapply {ms0:.2f} ms  ratio {ms0/ms1:.3f}  passed {int(a["passed"].sum())}/{int(b["passed"].sum())} '
          f'minCZ {int(a["cz"][a["passed"]].min())}/{int(b["cz"][b["passed"]].min())} [{va}] vs [{vb}]', flush=True)
