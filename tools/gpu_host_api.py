"""End-to-end timing of the host-pointer entry point cpf_run_static (hipMalloc + H2D + kernel + D2H), config 2."""
import ctypes as C, sys, time, numpy as np
sys.path.insert(0, '.')
from cpflow_amd import _ffi
from cpflow_amd.engine import placements_for
lib = _ffi.lib()
layer, K, S, T = [[0, 1], [1, 2]], 12, 1024, 2000
pl = placements_for(layer, K); tgt = np.ascontiguousarray(np.diag([1,1,1,1,1,1,1,-1]).astype(np.complex128)); rot = b'xyz'
pb = _ffi.Problem(3, K, pl.ctypes.data, 3, rot, 0, 0, tgt.ctypes.data, _ffi.Penalty(0, np.pi/2, 2, .05, .05, .05), 1, 0.00055,
                  0.1, 0.9, 0.999, 1e-8, T, 0.2, 1e-3)
P = 93
out = dict(params=np.empty((S, 2, P), np.float32), regloss=np.empty((S, 2), np.float32), loss=np.empty((S, 2), np.float32),
           reg=np.empty((S, 2), np.float32), cz=np.empty(S, np.int32), passed=np.empty(S, np.uint8))
res = _ffi.Result(*[out[k].ctypes.data for k in ('params', 'regloss', 'loss', 'reg', 'cz', 'passed')])
for rep in range(4):
    t0 = time.perf_counter()
    rc = lib.cpf_run_static(C.byref(pb), 0, 0, S, 0, S, 0, None, None, C.byref(res))
    dt = time.perf_counter() - t0
    assert rc == 0, lib.cpf_last_error()
    print(f'cpf_run_static rep {rep}: {dt*1e3:.2f} ms end to end -> {S*T/dt:.3e} sample-steps/s; passed {int(out["passed"].sum())}')
