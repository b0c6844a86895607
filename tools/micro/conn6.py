"""6 qubits, all-to-all layer (15 pairs, 30 blocks): the ZYZ-form kernel built at run time against the dense-block kernel and the
float64 oracle (robustness of the long straight-line layer code; not a benchmark shape)."""
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
from cpflow_amd.engine import TemplateEngine
from cpflow_amd import _ffi
from gpu import rand_u
from oracle import Oracle
layer = [[i, j] for i in range(6) for j in range(i + 1, 6)]
K, S, T = 30, 256, 200
target = rand_u(6, 3)
orc = Oracle(layer, K, target, 'xyz', num_gd_iterations=8)
for zyz in ('1', '0'):
    _ffi.set_tuning('CPF_ZYZ', zyz)
    eng = TemplateEngine(layer, K, target, 'xyz', num_gd_iterations=T)
    init = eng.initial_angles(0, S)
    t0 = time.time(); eng.optimize(init); torch.cuda.synchronize(); t1 = time.time() - t0
    out = eng.optimize(init); ms = eng.last_kernel_ms()
    eng8 = TemplateEngine(layer, K, target, 'xyz', num_gd_iterations=8)
    o8 = eng8.optimize(init[:3], keep_history=True)
    ref = orc.run(init[:3].cpu().numpy(), np.float64, keep_history=True)
    err = np.abs(o8['history_regloss'].cpu().numpy() - ref['history_loss']).max()
    print(f'CPF_ZYZ={zyz}: first launch {t1:.1f} s (build included), {ms:.2f} ms per {T} iterations [{eng.last_variant()}], |loss history - oracle| {err:.1e}', flush=True)
