"""Randomised parity sweep (not part of the test suite): random 3..5-qubit layers, block counts and rotation strings
through the kernel variants the knobs select, against the float64 oracle over a short run with history.
usage: python tools/gpu_fuzz.py [n_cases] [seed]"""
import itertools, os, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from cpflow_amd.engine import TemplateEngine
from oracle import Oracle

def rand_u(n, rng):
    a = rng.normal(size=(2**n, 2**n)) + 1j * rng.normal(size=(2**n, 2**n))
    q, r = np.linalg.qr(a); return q * (np.diag(r) / np.abs(np.diag(r)))

N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ROTS = ['xyz', 'xz', 'yz', 'zxz', 'xyz', 'z', 'yxz', 'xy', 'zyx']
VARIANTS = [dict(CPF_PSTORE='1', CPF_KSPEC='1', CPF_JIT_K='1'), dict(CPF_PSTORE='0', CPF_KSPEC='0', CPF_JIT_K='0'),
            dict(CPF_PSTORE='1', CPF_KSPEC='0', CPF_JIT_K='0'), dict(CPF_SPLIT='0')]
bad = 0
t0 = time.time()
for case in range(N):
    n = int(rng.choice([3, 3, 3, 4, 5]))
    pairs = [list(p) for p in itertools.permutations(range(n), 2)]
    LL = int(rng.integers(1, 4))
    layer = [pairs[i] for i in rng.choice(len(pairs), LL, replace=False)]
    if max(max(p) for p in layer) != n - 1: layer[0] = [int(rng.integers(n - 1)), n - 1]     # the layer defines n (topology.py:36-38)
    K = int(rng.integers(2, 19)) if n == 3 else int(rng.integers(2, 9))
    rot = ROTS[int(rng.integers(len(ROTS)))]
    target = rand_u(n, rng)
    T, S = 6, 3
    r = float(rng.choice([0.0, 0.004]))
    orc = Oracle(layer, K, target, rot, num_gd_iterations=T, r=r)
    init = orc.initial_angles(int(rng.integers(1 << 20)), S)
    ref = orc.run(init, np.float64, keep_history=True)
    for v in (VARIANTS if n == 3 else VARIANTS[:1]):
        for k in ('CPF_PSTORE', 'CPF_KSPEC', 'CPF_JIT_K', 'CPF_SPLIT'): os.environ.pop(k, None)
        os.environ.update(v)
        eng = TemplateEngine(layer, K, target, rot, num_gd_iterations=T, r=r)
        out = {k: x.cpu().numpy() for k, x in eng.optimize(init, keep_history=True).items() if x is not None}
        e0 = np.abs(out['history_regloss'][:, 0] - ref['history_loss'][:, 0]).max()
        eh = np.abs(out['history_regloss'] - ref['history_loss']).max()
        ep = np.abs(out['history_params'][:, :3] - ref['history_params'][:, :3]).max()
        ok = e0 < 1e-5 and eh < 5e-5 and ep < 2e-4 and (out['cz'] == ref['cz']).all()
        bad += not ok
        print(f"{'ok ' if ok else 'BAD'} n={n} layer={layer} K={K} rot={rot} r={r} {v} [{eng.last_variant()}] first {e0:.1e} hist {eh:.1e} params {ep:.1e}", flush=True)
print('cases', N, 'failures', bad, 'time %.0fs' % (time.time() - t0))
