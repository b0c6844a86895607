"""One optimise launch of a named config (for rocprofv3 --pmc passes). usage: gpu_one_cfg.py cfg4|cfg5|n5 [T]"""
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import os
os.environ.setdefault('T', sys.argv[2] if len(sys.argv) > 2 else '200')
import importlib.util
spec = importlib.util.spec_from_file_location('gp', 'tools/gpu_perf.py')
src = open('tools/gpu_perf.py').read().split("T = int(os.environ")[0]
ns = {}
exec(src, ns)
from cpflow_amd.engine import TemplateEngine
layer, K, tgt, Ss = ns['CFG'][sys.argv[1]]
T = int(os.environ['T'])
eng = TemplateEngine(layer, K, tgt, 'xyz', num_gd_iterations=T)
init = eng.initial_angles(0, Ss[0])
eng.optimize(init); torch.cuda.synchronize()
print('kernel ms', eng.last_kernel_ms(), 'S', Ss[0], 'T', T, 'P', eng.num_angles, 'G', eng.num_gates)
