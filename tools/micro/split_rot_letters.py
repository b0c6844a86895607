"""Timing prototype for the 'one real Ry between merged diagonal runs' lever of the 3-qubit two-ended kernel.
The shipped kernel already merges runs of diagonal gates (Rz . CP . Rz) into one phase per amplitude, so running config 2's shape with the
rotation string 'zyz' executes exactly the sweeps a two-ended ZYZ-form kernel would execute - one real rotation per qubit and block between
merged diagonals - with the same 7 angles per block in the owner phase, and WITHOUT the ZYZ form's per-step build / chain work.
'xyz' is the shipped configuration; 'yz' and 'z' bracket the share of the inner rotations."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
from cpflow_amd.engine import TemplateEngine
from cpflow_amd import _ffi, workloads as w

# every string on the same footing: the run-time build specialised on the block count (what the shipped 'xyz' instance is)
_ffi.set_tuning('CPF_JIT_K', '1')
_ffi.set_tuning('CPF_NO_AOT3', '1')

cfg = w.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else 'cfg2']
for S in (int(x) for x in (sys.argv[2].split(',') if len(sys.argv) > 2 else ['1024', '512', '256'])):
    for rot in ('xyz', 'zyz', 'yz', 'z'):
        eng = TemplateEngine(cfg.layer, cfg.num_cp_gates, cfg.target, rot, num_gd_iterations=2000)
        init = eng.initial_angles(0, S)
        eng.optimize(init); torch.cuda.synchronize()
        ms = []
        for _ in range(5):
            eng.optimize(init); ms.append(eng.last_kernel_ms())
        print(f'S={S} rot={rot:4s} P={eng.num_angles:3d} kernel {min(ms):.3f} ms (median {sorted(ms)[2]:.3f}) [{eng.last_variant()}]', flush=True)
