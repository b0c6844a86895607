"""One config-2 optimise launch (for rocprofv3 --pmc passes)."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from cpflow_amd.engine import TemplateEngine
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
eng = TemplateEngine([[0,1],[1,2]], 12, np.diag([1,1,1,1,1,1,1,-1]).astype(complex), 'xyz')
init = eng.initial_angles(0, S)
for _ in range(2):
    out = eng.optimize(init)
torch.cuda.synchronize()
print('kernel ms', eng.last_kernel_ms())
