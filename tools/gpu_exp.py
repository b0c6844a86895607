import os, sys, numpy as np, torch
sys.path.insert(0, '.')
from cpflow_amd.engine import TemplateEngine
CCZ = np.diag([1,1,1,1,1,1,1,-1]).astype(complex)
for use_reg in (True, False):
    eng = TemplateEngine([[0,1],[1,2]], 12, CCZ, 'xyz', use_reg=use_reg)
    init = eng.initial_angles(0, 1024)
    eng.optimize(init); torch.cuda.synchronize()
    ms = min((eng.optimize(init), eng.last_kernel_ms())[1] for _ in range(3))
    print(os.environ.get('CPF_LIB', 'default'), 'use_reg', use_reg, 'kernel ms %.2f' % ms)
