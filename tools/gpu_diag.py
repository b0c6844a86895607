"""Ad-hoc GPU diagnostics (not a test): trajectory error growth, G1 end-of-run stats, timing."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from oracle import Oracle
from cpflow_amd.engine import TemplateEngine

def rand_u(n, seed):
    rng = np.random.default_rng(seed)
    a = rng.normal(size=(2**n, 2**n)) + 1j*rng.normal(size=(2**n, 2**n))
    q, r = np.linalg.qr(a); return q*(np.diag(r)/np.abs(np.diag(r)))

for layer, K, rot in [([[0,1],[1,2]],12,'xyz'), ([[0,1],[1,2],[2,3]],24,'xyz'), ([[i,(i+1)%6] for i in range(6)],13,'xyz')]:
    n = max(max(p) for p in layer)+1
    tgt = rand_u(n, 3); T = 40
    orc = Oracle(layer, K, tgt, rot, num_gd_iterations=T); eng = TemplateEngine(layer, K, tgt, rot, num_gd_iterations=T)
    init = orc.initial_angles(11, 6)
    r32 = orc.run(init, np.float32, keep_history=True); r64 = orc.run(init, np.float64, keep_history=True)
    out = eng.optimize(init, keep_history=True)
    hl = out['history_regloss'].cpu().numpy()
    print('n', n, 'gpu-vs-f64', np.abs(hl-r64['history_loss']).max(axis=0)[[0,5,10,20,30,39]])
    print('      c32-vs-f64', np.abs(r32['history_loss']-r64['history_loss']).max(axis=0)[[0,5,10,20,30,39]])
    print('      gpu-vs-c32', np.abs(hl-r32['history_loss']).max(axis=0)[[0,5,10,20,30,39]])

g = np.load('tests/golden/raw_toff3_chain_t49.npz')
eng = TemplateEngine(g['layer'], int(g['num_cp_gates']), g['target'], str(g['rotation_gates']), r=float(g['r']))
init = eng.initial_angles(int(g['seed']), 100)
torch.cuda.synchronize(); t = time.time()
out = eng.optimize(init); torch.cuda.synchronize(); el = time.time()-t
print('G1 run: wall', el, 'kernel ms', eng.last_kernel_ms())
idx = g['sample_index']
cz = out['cz'].cpu().numpy(); ps = out['passed'].cpu().numpy()
print(' passed', ps.sum(), 'failing', np.where(~ps)[0], 'cz equal', (cz[idx]==g['cz']).sum(), '/', len(idx))
d = np.abs(out['regloss'].cpu().numpy()[idx,1]-g['regloss'][:,1])
print(' regloss diff median/p90/max', np.median(d), np.percentile(d,90), d.max(), (d<=1e-5).sum())
print(' loss at best vs stored', np.abs(out['loss'].cpu().numpy()[idx,1]-g['loss'][:,1]).max())

# config 2 timing
V = np.diag([1,1,1,1,1,1,1,-1]).astype(complex)
eng = TemplateEngine([[0,1],[1,2]], 12, V, 'xyz')
for S in (1024, 2048, 4096, 16384):
    init = eng.initial_angles(0, S)
    out = eng.optimize(init); torch.cuda.synchronize()
    ms = []
    for _ in range(3):
        out = eng.optimize(init); ms.append(eng.last_kernel_ms())
    ms = min(ms)
    print('cfg2-like S', S, 'kernel ms', ms, 'sample-steps/s %.3e' % (S*2000/ms*1e3), 'TFLOP/s %.2f' % (eng.flops_per_sample_step*S*2000/ms*1e3/1e12),
          'passed', int(out['passed'].sum()))
if S:
    cz = out['cz'].cpu().numpy()[:10]; print(cz, out['passed'].cpu().numpy()[:10])
