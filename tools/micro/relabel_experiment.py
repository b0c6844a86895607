import sys, os, itertools, numpy as np
sys.path.insert(0, os.getcwd())
from cpflow_amd import workloads as w
from cpflow_amd.engine import TemplateEngine
import torch
wl = w.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else 'cfg4']
n = max(max(p) for p in wl.layer) + 1
def permuted(perm):
    layer = [[perm[a], perm[b]] for a, b in wl.layer]
    N = 1 << n
    idx = np.zeros(N, dtype=int)
    for r in range(N):
        bits = [(r >> (n - 1 - q)) & 1 for q in range(n)]
        nr = 0
        for q in range(n):
            nr |= bits[q] << (n - 1 - perm[q])
        idx[r] = nr
    V = np.zeros((N, N), dtype=complex)
    V[np.ix_(idx, idx)] = wl.target
    return layer, V
for perm in ([0,1,2,3], [0,2,3,1], [1,2,3,0], [2,0,1,3], [2,3,0,1], [3,1,0,2]) if n == 4 else ([0,1,2,3,4],):
    layer, V = permuted(perm)
    eng = TemplateEngine(layer, wl.num_cp_gates, V, 'xyz', num_gd_iterations=2000)
    init = eng.initial_angles(0, wl.samples_per_gpu)
    out = eng.optimize(init); torch.cuda.synchronize()
    ms = []
    for _ in range(3):
        out = eng.optimize(init); ms.append(eng.last_kernel_ms())
    print(perm, layer, 'min %.2f ms' % min(ms), eng.last_variant(), 'passed', int(out['passed'].sum()), flush=True)
