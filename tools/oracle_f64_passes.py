#!/usr/bin/env python3
"""How many runs of the stored 5-qubit benchmark problem (paper/results/benchmarks/res_4gt13_91 trial 99: chain, 'xyz', 86 blocks)
end below entry_loss on the float64 oracle, the float32 oracle and the device - the pooled replicate behind DESIGN.md section 2
(which of the ~8 % of the runs converge is decided by rounding; the question is whether the kernel's count sits with float64).
    python tools/oracle_f64_passes.py [samples per seed = 500] [seeds = 2]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from cpflow_amd.engine import TemplateEngine  # noqa: E402

S, nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 500, int(sys.argv[2]) if len(sys.argv) > 2 else 2
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'raw_4gt13_xyz_t99.npz'))
layer, K, target, rot, r = g['layer'].tolist(), int(g['num_cp_gates']), g['target'], str(g['rotation_gates']), float(g['r'])
tot = {'f64': 0, 'f32': 0, 'device': 0}
for seed in range(int(g['seed']), int(g['seed']) + nseeds):
    eng = TemplateEngine(layer, K, target, rot, r=r, num_gd_iterations=2000)
    init = eng.initial_angles(seed, S)
    dev = eng.optimize(init)['passed'].cpu().numpy()
    orc = oracle.Oracle(layer, K, target, rot, num_gd_iterations=2000, r=r)
    nt = int(orc.lib.cpfo_max_threads())
    p64 = orc.run(init.cpu().numpy(), np.float64, nthreads=nt)['passed'].astype(bool)
    p32 = orc.run(init.cpu().numpy(), np.float32, nthreads=nt)['passed'].astype(bool)
    print(f'seed {seed}: device [{eng.last_variant()}] {dev.sum()}  float32 oracle {p32.sum()}  float64 oracle {p64.sum()}  of {S}; '
          f'same flag device/f64 {np.mean(dev == p64):.3f} f32/f64 {np.mean(p32 == p64):.3f} device/f32 {np.mean(dev == p32):.3f}', flush=True)
    tot['f64'] += int(p64.sum()); tot['f32'] += int(p32.sum()); tot['device'] += int(dev.sum())
print('pooled:', tot, 'of', S * nseeds)
