"""End-to-end Synthesize.static() on a 4-qubit problem of the tutorial's size (toff4_star trial 14: star layer, 29 CP gates, 500 samples):
where the wall time goes (raw stage vs verification vs Python)."""
import contextlib, io, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cpflow_amd import StaticOptions, Synthesize
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'raw_toff4_star_t14.npz'))
layer, K, target = g['layer'].tolist(), int(g['num_cp_gates']), g['target']
syn = Synthesize(layer, target_unitary=target, label='toff4_star')
opts = StaticOptions(num_cp_gates=K, accepted_num_cz_gates=40, num_samples=int(g['num_samples']), r=float(g['r']), random_seed=int(g['seed']))
for rep in range(3):
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        raw = syn._generate_raw(opts)
        t1 = time.perf_counter()
        res = syn.static(opts, save_results=False)
    t2 = time.perf_counter()
    print(f'rep {rep}: raw stage {1e3 * (t1 - t0):.1f} ms, whole static() {1e3 * (t2 - t1):.1f} ms, {len(res.decompositions)} decompositions'
          + (f', best {min(d.cz_count for d in res.decompositions)} CZ' if res.decompositions else ''), flush=True)
