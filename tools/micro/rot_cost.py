import sys; sys.path.insert(0,'/root/repo')
import torch
from cpflow_amd import workloads
from cpflow_amd.engine import TemplateEngine
wl=workloads.CONFIGS['cfg2']
for rot in ('xyz','xz','z'):
    eng=TemplateEngine(wl.layer, wl.num_cp_gates, wl.target, rot, r=wl.r, num_gd_iterations=2000)
    init=eng.initial_angles(0,1024)
    eng.optimize(init); torch.cuda.synchronize()
    ms=[]
    for _ in range(5):
        eng.optimize(init); ms.append(eng.last_kernel_ms())
    print(rot, 'P', eng.num_angles, min(ms), eng.last_variant())
