"""Diagnostic: per-phase cycle shares of the layered kernel (library built with -DCPF_STAMPS, see cpf_layered.h)."""
import os, sys, numpy as np, torch
sys.path.insert(0, '.')
from cpflow_amd.engine import TemplateEngine
CCZ = np.diag([1,1,1,1,1,1,1,-1]).astype(complex)
T = 500
eng = TemplateEngine([[0,1],[1,2]], 12, CCZ, 'xyz', num_gd_iterations=T)
init = eng.initial_angles(0, int(os.environ.get('S', '1024')))
out = eng.optimize(init, keep_history=True)
if eng.last_variant().startswith('split'):
    h = out['history_regloss'][0, :16].cpu().numpy().astype(np.float64)
    names = ['forward half', 'wait for other wave', 'loss/track', 'backward half', 'wait for other wave', 'owner', 'wait (owner)', 'loop top']
    for w in (0, 1):
        print('wave', 'AB'[w]); st = h[8 * w:8 * w + 8]; tot = st.sum()
        for n, v in zip(names, st):
            print(f'  {n:30s} {v / T:9.1f} cycles/step  {100 * v / tot:5.1f} %')
        print('  total', tot / T, 'cycles/step')
    sys.exit(0)
st = out['history_regloss'][0, :8].cpu().numpy().astype(np.float64)
names = ['surface fwd', 'blocks fwd', 'loss/track', 'blocks bwd', 'surface bwd', 'owner (Adam, sincos, penalty)', '-', 'loop top']
tot = st.sum()
for n, v in zip(names, st):
    print(f'{n:32s} {v / T:9.1f} cycles/step  {100 * v / tot:5.1f} %')
print('total', tot / T, 'cycles/step (history writes included)')
