#!/usr/bin/env python3
"""Which waves share a SIMD in a launch of the two-ended 3-qubit kernel (cpf_zyz3.h)?  Needs a diagnostics build:

    python -c "from cpflow_amd import build; build.build_variant('hwid', ['CPF_DIAGNOSTICS', 'CPF_HWID'])"
    CPF_LIB=cpflow_amd/libcpflow_amd.hwid.so python tools/micro/hwid3.py [cfg2|cfg3] [launches]

Every wave writes HW_ID / XCC_ID into its sample's loss history (T = 4).  Printed per launch: how many SIMDs host which combination of
wave roles (A = surface gates + the first blocks, B = the block angles + the other blocks), and the kernel time of a full-length launch
of the same process right after - so that a placement and a time can be seen side by side."""
import collections
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    from cpflow_amd import workloads
    from cpflow_amd.engine import TemplateEngine
    cfg = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    wl = workloads.CONFIGS[cfg]
    S = wl.samples_per_gpu
    short = TemplateEngine(wl.layer, wl.num_cp_gates, wl.target, wl.rotation_gates, r=wl.r, num_gd_iterations=6)
    full = TemplateEngine(wl.layer, wl.num_cp_gates, wl.target, wl.rotation_gates, r=wl.r, num_gd_iterations=2000)
    init = short.initial_angles(0, S * wl.gpus, 0, S)
    full.optimize(init)
    for _ in range(n):
        h = short.optimize(init, keep_history=True)['history_regloss'].cpu().numpy()
        hf = full.optimize(init, keep_history=True)['history_regloss'].cpu().numpy()
        torch.cuda.synchronize()
        ms = full.last_kernel_ms()
        ghz = np.median(hf[:, 4] / hf[:, 5] * 0.1)           # shader clocks per 100 MHz tick
        h = hf                                               # the placement of the FULL launch itself
        slots, where = collections.defaultdict(list), {}
        for s in range(S):
            for role, (hw, xcc) in (('A', h[s, 0:2]), ('B', h[s, 2:4])):
                hw, xcc = int(hw), int(xcc)
                key = (xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, (hw >> 4) & 3)
                slots[key].append(role)
                where[(s, role)] = key
        combos = collections.Counter(''.join(sorted(v)) for v in slots.values())
        # clocks per step of a sample's wave B by what its SIMD hosts
        by = collections.defaultdict(list)
        for s in range(S):
            by[''.join(sorted(slots[where[(s, 'B')]]))].append(hf[s, 4] / 2000)
        print('   clocks per step of wave B by the roles on its SIMD:', {k: (len(v), round(float(np.median(v))), round(float(np.max(v)))) for k, v in sorted(by.items())},
              '| all samples: median %.0f max %.0f' % (np.median(hf[:, 4]) / 2000, hf[:, 4].max() / 2000))
        per_cu = collections.Counter()
        for k, v in slots.items():
            per_cu[k[:4]] += len(v)
        print(f'{cfg} [{short.last_variant()}] SIMDs used {len(slots)}; roles per SIMD {dict(combos)}; waves per CU {dict(collections.Counter(per_cu.values()))}; '
              f'full launch right after: {ms:.3f} ms at {ghz:.3f} GHz ({np.median(hf[:, 4]) / 2000:.0f} clocks per step)', flush=True)


if __name__ == '__main__':
    main()
