import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import Oracle
from cpflow_amd.engine import TemplateEngine
g = np.load('tests/golden/raw_toff3_chain_t49.npz')
orc = Oracle(g['layer'], int(g['num_cp_gates']), g['target'], str(g['rotation_gates']), r=float(g['r']))
eng = TemplateEngine(g['layer'], int(g['num_cp_gates']), g['target'], str(g['rotation_gates']), r=float(g['r']))
for k, name in ((0, 'init'), (1, 'best')):
    th = g['params'][:, k, :]
    r64 = orc.loss_grad(th, np.float64); r32 = orc.loss_grad(th, np.float32)
    gg = eng.loss_and_grad(th)
    G = gg['grad'].cpu().numpy(); L = gg['loss'].cpu().numpy()
    print(name, 'grad err: gpu-vs-f64 max %.2e rms %.2e | c32-vs-f64 max %.2e rms %.2e | |grad| max %.2e' % (
        np.abs(G-r64['grad']).max(), np.sqrt(((G-r64['grad'])**2).mean()), np.abs(r32['grad']-r64['grad']).max(),
        np.sqrt(((r32['grad']-r64['grad'])**2).mean()), np.abs(r64['grad']).max()))
    print(name, 'loss err: gpu %.2e c32 %.2e' % (np.abs(L-r64['loss']).max(), np.abs(r32['loss']-r64['loss']).max()))
