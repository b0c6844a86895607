"""Wall time of the whole Synthesize.static() call (raw stage + batched verification + decomposition objects)."""
import sys, time, io, contextlib, numpy as np
sys.path.insert(0, '.')
from cpflow_amd import Synthesize, StaticOptions
CCZ = np.diag([1, 1, 1, 1, 1, 1, 1, -1]).astype(complex)
for S in (10, 1024):
    syn = Synthesize([[0, 1], [1, 2]], target_unitary=CCZ, label='ccz')
    opts = StaticOptions(num_cp_gates=12, accepted_num_cz_gates=10, num_samples=S)
    for rep in range(2):
        buf = io.StringIO()
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(buf):
            t1 = time.perf_counter(); raw = syn._generate_raw(opts); t2 = time.perf_counter()
            res = syn.static(opts, save_results=False)
        dt = time.perf_counter() - t0
        print(f'S={S} rep {rep}: raw stage {1e3*(t2-t1):.1f} ms, whole static() {1e3*(dt-(t2-t1)):.1f} ms, '
              f'{len(res.decompositions)} decompositions, best {min(d.cz_count for d in res.decompositions)} CZ')
