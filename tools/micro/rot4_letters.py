import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from cpflow_amd.engine import TemplateEngine
from cpflow_amd import _ffi
sys.path.insert(0, '/root/repo/tools')
from gpu import rand_u
for n, layer, K, S in ((4, [[0, 1], [1, 2], [2, 3]], 12, 2048), (5, [[0, 1], [1, 2], [2, 3], [3, 4]], 16, 1000), (4, [[0, 1], [1, 2], [2, 3]], 12, 100)):
    for dense in ('0', '1'):
        _ffi.set_tuning('CPF_DENSE', dense)
        eng = TemplateEngine(layer, K, rand_u(n, 1), 'zyxz', num_gd_iterations=2000)
        init = eng.initial_angles(0, S)
        eng.optimize(init); torch.cuda.synchronize()
        ms = []
        for _ in range(3):
            out = eng.optimize(init); ms.append(eng.last_kernel_ms())
        print(n, K, S, 'CPF_DENSE=' + dense, f'{min(ms):.2f} ms', eng.last_variant(), int(out['passed'].sum()), flush=True)
