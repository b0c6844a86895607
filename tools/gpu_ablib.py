"""A/B of two builds of the library on the same box: usage gpu_ablib.py libA.so libB.so [cfg] — alternates processes."""
import os, subprocess, sys
libs = sys.argv[1:3]
cfg = sys.argv[3] if len(sys.argv) > 3 else 'cfg2'
res = {l: [] for l in libs}
for rep in range(3):
    for l in libs:
        env = dict(os.environ, CPF_LIB=os.path.abspath(l), FIRST='1')
        out = subprocess.run([sys.executable, 'tools/gpu_perf.py', cfg], env=env, capture_output=True, text=True).stdout
        line = [x for x in out.splitlines() if x.startswith(cfg)][0]
        res[l].append(float(line.split('kernel')[1].split('ms')[0]))
for l in libs:
    print(os.path.basename(l), cfg, 'S=first', 'min %.2f ms' % min(res[l]), res[l])
