#!/usr/bin/env python3
"""Where a short Adam run's first iterates leave the float64 oracle: the parameter, its float64 gradient at the start, the deviation.
(Adam's first step is lr * g / (|g| + eps): an angle whose gradient is of the size of float32's noise moves by anything up to lr.)
    python tools/micro/first_step_dev.py          the 3-qubit case of ZYZL_CASES, relabel 1"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import Oracle  # noqa: E402
from test_gpu_kernels import _rand_unitary, _engine  # noqa: E402
from cpflow_amd import _ffi  # noqa: E402

layer, K, rot, ent = [[0, 1], [1, 2]], 12, 'xyz', 'cp'
n, T, S = 3, 12, 6
target = _rand_unitary(n, 5)
for k, v in (('CPF_LW', '0'), ('CPF_ZYZL', '1'), ('CPF_RELABEL', '1')):
    _ffi.set_tuning(k, v)
kw = dict(entangler=ent, num_gd_iterations=T, r=0.003)
orc = Oracle(layer, K, target, rot, **kw)
eng = _engine(layer, K, target, rot, **kw)
init = orc.initial_angles(3, S)
out = {k: v.cpu().numpy() for k, v in eng.optimize(init, keep_history=True).items()}
print(eng.last_variant())
ref = orc.run(init, np.float64, keep_history=True)
r32 = orc.run(init, np.float32, keep_history=True)
g64 = orc.loss_grad(init, np.float64)['grad']
for name, o in (('kernel', out), ('float32 oracle', r32)):
    dev = np.abs(o['history_params'] - ref['history_params'])[:4, :2]
    idx = np.argsort(dev.ravel())[::-1][:6]
    print(name, 'max', dev.max())
    for i in idx:
        s, t, j = np.unravel_index(i, dev.shape)
        print(f'   sample {s} iterate {t} angle {j}: deviation {dev[s, t, j]:.3e}   float64 gradient at the start {g64[s, j]:+.3e}')
