#!/usr/bin/env python3
"""Sustained launch time of run-time built variants of the two-ended 3-qubit kernel: N back-to-back launches per variant (the clock of a
launch depends on what the chip has been doing: tools/gpu.py knobs alternates variants and so mixes their power states).
    CPF_LIB=cpflow_amd/libcpflow_amd.diag.so python tools/micro/steady.py cfg2 1024 150 "-DCPF_Z3_FUSED=1" "-DCPF_Z3_FUSED=1 -DCPF_Z3_STAGGER=4" ...
Each variant is visited twice (A B ... A B ...), statistics per visit."""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import gpu  # noqa: E402
import torch  # noqa: E402

cfg, S, N, variants = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4:]
eng, _ = gpu.engine(cfg)
init = eng.initial_angles(0, S)
for visit in range(2):
    for v in variants:
        os.environ.pop('CPF_ZYZ3_DEFS', None)
        os.environ.pop('CPF_PRIO', None)
        for part in v.split(';'):
            if part.startswith('CPF_PRIO='):       # the launch's priority / stagger mask (cpflow_amd.hip try_zyz3)
                os.environ['CPF_PRIO'] = part.split('=')[1]
            elif part != 'release':
                os.environ['CPF_ZYZ3_DEFS'] = part
        gpu.sync_tuning()
        for _ in range(3):
            eng.optimize(init)
        torch.cuda.synchronize()
        t = []
        for _ in range(N):
            eng.optimize(init)
            torch.cuda.synchronize()
            t.append(eng.last_kernel_ms())
        t = np.array(t)
        print(f'{cfg} S={S} visit {visit} {v:48s} mean {t.mean():.3f} median {np.median(t):.3f} min {t.min():.3f} p90 {np.quantile(t, 0.9):.3f} max {t.max():.3f}  [{eng.last_variant()}]', flush=True)
