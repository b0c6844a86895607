"""Ad-hoc timing of the optimise kernel (not a test).  usage: python tools/gpu_perf.py [cfg ...]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from cpflow_amd.engine import TemplateEngine

def rand_u(n, seed):
    rng = np.random.default_rng(seed)
    a = rng.normal(size=(2**n, 2**n)) + 1j*rng.normal(size=(2**n, 2**n))
    q, r = np.linalg.qr(a); return q*(np.diag(r)/np.abs(np.diag(r)))

CCZ = np.diag([1,1,1,1,1,1,1,-1]).astype(complex)
TOFF3 = np.eye(8)[[0,1,2,3,4,5,7,6]].astype(complex)
w = np.exp(2j*np.pi/16); QFT4 = np.array([[w**(j*k) for k in range(16)] for j in range(16)])/4
CFG = {
 'cfg2': ([[0,1],[1,2]], 12, CCZ, [1024, 2048, 4096, 16384]),
 'cfg3': ([[0,1],[0,2],[1,2]], 16, TOFF3, [512, 4096]),
 'cfg4': ([[0,1],[1,2],[2,3]], 24, QFT4, [2048]),
 'cfg5': ([[i,(i+1)%6] for i in range(6)], 60, rand_u(6, 0), [1024]),
 'n5':   ([[0,1],[1,2],[2,3],[3,4]], 36, rand_u(5, 0), [1000]),
}
T = int(os.environ.get('T', '2000'))
for name in (sys.argv[1:] or ['cfg2']):
    layer, K, tgt, Ss = CFG[name]
    eng = TemplateEngine(layer, K, tgt, 'xyz', num_gd_iterations=T)
    for S in (Ss[:1] if os.environ.get('FIRST') else Ss):
        init = eng.initial_angles(0, S)
        out = eng.optimize(init); torch.cuda.synchronize()
        ms = []
        for _ in range(3):
            out = eng.optimize(init); ms.append(eng.last_kernel_ms())
        ms = min(ms)
        print(f'{name} S={S} T={T} kernel {ms:.2f} ms  {S*T/ms*1e3:.3e} sample-steps/s  {eng.flops_per_sample_step*S*T/ms*1e3/1e12:.2f} TFLOP/s  '
              f'[{eng.last_variant()}] passed {int(out["passed"].sum())} minCZ {int(out["cz"][out["passed"]].min()) if out["passed"].any() else -1}', flush=True)
