"""Timing prototype of the E = 4 layout for the 3-qubit two-ended kernel: two row qubits in registers, four
samples per wave.  A wave of that layout holds 4 x 64 = 256 amplitudes, four per lane - which is exactly the one-wave 4-qubit kernels'
layout (cpf_zyzl.h / cpf_layered.h: 16 x 16 amplitudes, the two most-used row qubits in registers after relabelling).  So the 4-qubit
kernels run on config 2's layer [[0,1],[1,2]] with qubit 3 idle execute the block sweeps an E = 4 wave would execute (two used row
qubits in registers, one on lanes, the fourth row bit standing in for a sample bit), with the per-step phases of ONE 96- or 54-angle
sample where the real thing would serve four samples' 372 angles: a LOWER bound.
    K = 12, S = 256  : one-ended E = 4 at config 2's 1024 samples (256 lone waves)
    K =  6, S = 512  : each wave of a two-ended E = 4 pair (512 lone waves, half the blocks each, no exchange, no second half's owner work)
against the shipped kernel on config 2 in the same process."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
from cpflow_amd.engine import TemplateEngine
from cpflow_amd import _ffi, workloads as w
from gpu import rand_u

def run(tag, eng, S):
    init = eng.initial_angles(0, S)
    eng.optimize(init); torch.cuda.synchronize()
    ms = []
    for _ in range(5):
        eng.optimize(init); ms.append(eng.last_kernel_ms())
    print(f'{tag:46s} S={S:5d} P={eng.num_angles:3d} kernel {min(ms):.3f} ms (median {sorted(ms)[2]:.3f}) [{eng.last_variant()}]', flush=True)

cfg = w.CONFIGS['cfg2']
run('config 2, shipped', TemplateEngine(cfg.layer, cfg.num_cp_gates, cfg.target, 'xyz', num_gd_iterations=2000), 1024)
_ffi.set_tuning('CPF_LW', '0')                      # one wave per 4-qubit state: the E = 4 wave
for zyzl in ('1', '0'):
    _ffi.set_tuning('CPF_ZYZL', zyzl)
    for K, S in ((12, 256), (6, 512)):
        eng = TemplateEngine([[0, 1], [1, 2]], K, rand_u(4, 3), 'xyz', num_gd_iterations=2000, num_qubits=4)
        run(f'4-qubit wave, qubit 3 idle, K={K} CPF_ZYZL={zyzl}', eng, S)
