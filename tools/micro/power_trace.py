#!/usr/bin/env python3
"""Board power and shader clock while a variant of the two-ended 3-qubit kernel runs back to back (the evidence behind "the launch is
power-limited"): a child process launches the kernel N times (tools/micro/steady.py), this one polls rocm-smi.
    CPF_LIB=cpflow_amd/libcpflow_amd.diag.so python tools/micro/power_trace.py cfg2 1024 600 release "-DCPF_Z3_FUSED=1" ...
One line per variant: launches' mean / median / p90 (ms), power mean / max (W), sclk min / mean / max (MHz), samples."""
import os
import re
import subprocess
import sys
import time
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
cfg, S, N, variants = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4:]


def poll():
    out = subprocess.run(['rocm-smi', '--showpower', '--showclocks'], capture_output=True, text=True).stdout
    w = re.search(r'Power \(W\):\s*([0-9.]+)', out)
    c = re.search(r'sclk clock level:\s*\d+:?\s*\((\d+)Mhz\)', out)
    return (float(w.group(1)) if w else np.nan, float(c.group(1)) if c else np.nan)


for v in variants:
    child = subprocess.Popen([sys.executable, os.path.join(HERE, 'steady.py'), cfg, S, N, v], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    time.sleep(6.0)                                   # import, plan, run-time build, warm-up
    samples = []
    while child.poll() is None:
        samples.append(poll())
    lines = [x for x in child.stdout.read().splitlines() if x.startswith(cfg)]
    p = np.array(samples)
    p = p[p[:, 0] > 600.0] if (p[:, 0] > 600.0).any() else p          # the launches' stretch only (import, plan and run-time build idle at ~250 W)
    stat = lines[-1].split('  [')[0] if lines else '(no output)'
    print(f'{stat}   | power mean {np.nanmean(p[:, 0]):.0f} max {np.nanmax(p[:, 0]):.0f} W   sclk min {np.nanmin(p[:, 1]):.0f} mean {np.nanmean(p[:, 1]):.0f} '
          f'max {np.nanmax(p[:, 1]):.0f} MHz, below 2300 MHz in {100 * np.nanmean(p[:, 1] < 2300):.0f} % of {len(p)} samples', flush=True)
