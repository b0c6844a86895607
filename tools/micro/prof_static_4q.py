import contextlib, cProfile, io, os, pstats, sys
import numpy as np
ROOT = '/root/repo'
sys.path.insert(0, ROOT)
from cpflow_amd import StaticOptions, Synthesize
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'raw_toff4_star_t14.npz'))
layer, K, target = g['layer'].tolist(), int(g['num_cp_gates']), g['target']
syn = Synthesize(layer, target_unitary=target, label='toff4_star')
opts = StaticOptions(num_cp_gates=K, accepted_num_cz_gates=40, num_samples=int(g['num_samples']), r=float(g['r']), random_seed=int(g['seed']))
with contextlib.redirect_stdout(io.StringIO()):
    syn.static(opts, save_results=False)
pr = cProfile.Profile()
with contextlib.redirect_stdout(io.StringIO()):
    pr.enable(); syn.static(opts, save_results=False); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
