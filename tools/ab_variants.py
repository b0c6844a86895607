#!/usr/bin/env python3
"""Same-box A/B of library builds (cpflow_amd.build.build_variant): one child process per build and round, alternating,
kernel time of one BASELINE configuration.   python tools/ab_variants.py cfg5 base nosurf ...   [ROUNDS=2]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = '''
import sys; sys.path.insert(0, %r)
from cpflow_amd import workloads, _ffi
import ctypes
_h = ctypes.CDLL(_ffi.LIB_PATH)
for _k in list(_ffi.SYMBOLS):
    if not hasattr(_h, _k): _ffi.SYMBOLS.pop(_k)          # an older build (A/B reference) may lack newer entry points
from cpflow_amd.engine import TemplateEngine
import torch
wl = workloads.CONFIGS[sys.argv[1]]
eng = TemplateEngine(wl.layer, wl.num_cp_gates, wl.target, wl.rotation_gates, r=wl.r, num_gd_iterations=wl.num_gd_iterations)
init = eng.initial_angles(wl.seed, wl.samples_per_gpu * wl.gpus, 0, wl.samples_per_gpu)
eng.optimize(init); torch.cuda.synchronize()
ms = []
for _ in range(2):
    eng.optimize(init); ms.append(eng.last_kernel_ms())
print('MS', min(ms), eng.last_variant())
''' % ROOT


def main():
    cfg, names = sys.argv[1], sys.argv[2:]
    res = {n: [] for n in names}
    for _ in range(int(os.environ.get('ROUNDS', '2'))):
        for n in names:
            lib = os.path.join(ROOT, 'cpflow_amd', 'libcpflow_amd.so' if n == 'shipped' else f'libcpflow_amd.{n}.so')
            out = subprocess.run([sys.executable, '-c', CHILD, cfg], env=dict(os.environ, CPF_LIB=lib), capture_output=True, text=True)
            line = [x for x in out.stdout.splitlines() if x.startswith('MS')]
            if not line:
                print(n, 'FAILED', out.stderr[-500:], flush=True)
                continue
            res[n].append(float(line[0].split()[1]))
            var = ' '.join(line[0].split()[2:])
            print(f'{cfg} {n:12s} {res[n][-1]:9.3f} ms  [{var}]', flush=True)
    for n in names:
        if res[n]:
            print(f'{cfg} {n:12s} min {min(res[n]):9.3f} ms', flush=True)


if __name__ == '__main__':
    main()
