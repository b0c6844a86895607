"""GPU parity at the level the reference's callers see: golden end-of-run statistics, the
Synthesize front end, rank-independence of sharding, determinism, BASELINE-size properties."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle  # noqa: E402
from oracle import Oracle  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), 'golden')
CCZ = np.diag([1, 1, 1, 1, 1, 1, 1, -1]).astype(np.complex128)
TOFF3 = np.eye(8)[[0, 1, 2, 3, 4, 5, 7, 6]].astype(np.complex128)


def _engine(*a, **kw):
    from cpflow_amd.engine import TemplateEngine
    return TemplateEngine(*a, **kw)


def _np(d):
    return {k: v.cpu().numpy() for k, v in d.items() if v is not None}


@pytest.mark.parametrize('name,min_same', [('raw_toff3_chain_t49', 0.95), ('raw_toff3_chain_t19', 0.95),
                                           ('raw_toff4_star_t14', 0.65)])
def test_T3_golden_end_of_run(name, min_same):
    """Full 2000-iteration runs from the reference's seeds reproduce its stored results:
    same pass/fail set, same CZ counts except boundary samples, best regloss statistically equal.
    On the 4-qubit fixture 2000 Adam steps at lr=0.1 are chaotic: the float64 oracle reproduces 55/63 of the
    reference's per-sample CZ counts, the float32 oracle 53/63, the float32 oracle started one ulp away 55/63, and
    these three agree with EACH OTHER on only 84-86 % of the samples; the kernels here landed between 73 % and
    90 % depending on rounding order (un-applying sweep, packed arithmetic, sin/cos routine).  Per-sample equality
    is therefore a noisy statistic there (sigma = 4.5 % at n = 63): the bar is 65 %, plus distribution-level
    agreement (mean CZ count of the common samples within 0.5, adaptive score within 0.35)."""
    g = np.load(os.path.join(GOLD, name + '.npz'))
    S = int(g['num_samples'])
    eng = _engine(g['layer'], int(g['num_cp_gates']), g['target'], str(g['rotation_gates']), r=float(g['r']))
    init = eng.initial_angles(int(g['seed']), S)
    out = _np(eng.optimize(init))
    if 'toff3' in name:      # 14 / 15 blocks, 100 samples: the shipped K-specialised two-ended kernel, one sample per CU
        assert eng.last_variant().startswith('zyz3/jit(hipcc) waves=2 K cut='), eng.last_variant()      # shipped (cpflow_amd/jit_cache)
    idx = g['sample_index']
    assert (out['params'][idx, 0].view(np.uint32) == g['params'][:, 0].view(np.uint32)).all()   # T0
    passed = set(np.flatnonzero(out['passed']))
    # a sample whose final loss sits within 2e-4 of entry_loss may land on either side
    border = set(np.flatnonzero(np.abs(np.take_along_axis(out['loss'], out['regloss'].argmin(1)[:, None], 1)[:, 0] - 1e-3) < 2e-4))
    # 3-qubit fixtures: strict.  4-qubit fixture: the oracle reproduces the stored pass set exactly in float32 and
    # float64, but a kernel that rounds differently may send a run to another basin (measured: sample 428 of 500
    # reaches loss 6e-4 on the layered kernel and stalls at 3.8e-2 in the reference, the oracle and the generic
    # kernel); allow <= 1 % of the samples to do so.
    flips = (passed ^ set(idx)) - border
    assert len(flips) <= (0 if min_same >= 0.95 else S // 100), sorted(flips)
    both = [k for k, j in enumerate(idx) if j in passed]
    same = out['cz'][idx[both]] == g['cz'][both]
    assert same.mean() >= min_same, f'{(~same).sum()} CZ counts differ'
    assert abs(out['cz'][idx[both]].mean() - g['cz'][both].mean()) < 0.5
    score = -np.log2((2.0 ** -out['cz'][out['passed']].astype(np.float32)).sum() / S)
    assert abs(score - float(g['score'])) < 0.35
    # 3q: a mismatch is either a threshold-boundary sample (some CP angle within 0.05 of a 0.2-window edge)
    # or, rarely, a run that the rounding noise sent to a different local minimum; allow <= 3 % of the latter
    # (measured: trial 19 one-wave kernel 1/97, two-ended kernel 2/97 — samples 88 and 82, whose nearest CP angle
    # sits 0.06-0.11 from a window edge; trial 49 0/99 on both kernels)
    other_basin = 0
    for k in (np.flatnonzero(~same) if min_same >= 0.95 else []):
        j = idx[both][k]
        th = out['params'][j, 1][eng.cp_mask == 1] % np.float32(2 * np.pi)
        dist = np.minimum.reduce([np.abs(th - 0.2), np.abs(th - (np.pi - 0.2)), np.abs(th - (np.pi + 0.2)),
                                  np.abs(th - (2 * np.pi - 0.2))])
        other_basin += int(dist.min() >= 0.05)
    # the oracle itself, over five roundings of fixture t19 (float32, float64, initial angles moved by +1 / -1 / +2 ulp):
    # 0-2 such samples and up to 4 differing CZ counts of 97; the kernels: 0-3.  Allow 4 (the 95 % bar above still binds).
    assert other_basin <= 4
    d = np.abs(out['regloss'][idx[both], 1] - g['regloss'][both, 1])
    # a run that ended in another basin differs by a few penalty units (r * 2 per CZ-equivalent = 1.1e-3 here)
    assert np.median(d) < (2e-5 if min_same >= 0.95 else 1e-4) and d[same].max() < 2e-3 and d.max() < 5e-3
    # T1 at the reference's own best angles
    at = _np(eng.loss_and_grad(g['params'][:, 1, :]))
    assert np.abs(at['loss'] - g['loss'][:, 1]).max() < 1.5e-5
    assert np.abs(at['reg'] - g['reg'][:, 1]).max() < 1e-7


def test_G6_decompositions_rebuild_target():
    for name in ('dec_refine_test', 'dec_toff3_chain', 'dec_toff4_star'):
        g = np.load(os.path.join(GOLD, name + '.npz'))
        for i in range(len(g['angles'])):
            K, rot = int(g['num_cp_gates'][i]), str(g['rotation_gates'][i])
            eng = _engine(g['layer'], K, g['target'], rot, use_reg=False)
            res = eng.loss_and_grad(g['angles'][i, :eng.num_angles], with_grad=False)
            assert float(res["loss"][0]) < 6e-6      # stored loss <= 1e-6; float32 evaluation noise of 1 - |T|^2/N^2 is a few 1e-6
            eng.close()


def test_sharded_runs_equal_single_run_bitwise():
    """Rank-independence: running the sample blocks of 1, 2 or 3 'ranks' separately gives
    bit-identical rows; two identical launches are bit-identical too."""
    eng = _engine([[0, 1], [0, 2], [1, 2]], 16, TOFF3, 'xyz', num_gd_iterations=300)
    S = 50
    full = _np(eng.optimize(eng.initial_angles(7, S)))
    again = _np(eng.optimize(eng.initial_angles(7, S)))
    for k in full:
        assert (full[k] == again[k]).all()
    from cpflow_amd.distributed import shard_range
    for R in (2, 3):
        parts = []
        for r in range(R):
            first, count = shard_range(S, r, R)
            parts.append(_np(eng.optimize(eng.initial_angles(7, S, first, count))))
        for k in full:
            assert (np.concatenate([p[k] for p in parts]) == full[k]).all()


def test_static_front_end_config1(capsys):
    """BASELINE config 1 through the reference-shaped API: CCZ, chain, 12 CP gates, 10 samples, seed 0."""
    from cpflow_amd import StaticOptions, Synthesize
    layer = [[0, 1], [1, 2]]
    opts = StaticOptions(num_cp_gates=12, accepted_num_cz_gates=10, num_samples=10)
    syn = Synthesize(layer, target_unitary=CCZ, label='ccz_chain')
    raw = syn._generate_raw(opts)
    assert len(raw) == 10 and set(raw[0]) == {'params', 'loss', 'reg', 'regloss'}
    assert raw[0]['params'].shape == (2, 93) and raw[0]['regloss'].shape == (2,)
    ref = Oracle(layer, 12, CCZ, 'xyz').run(oracle.prng.initial_angles(0, 10, 93), np.float32)
    assert (raw.passed == ref['passed']).all()
    assert (raw.cz[raw.passed] == ref['cz'][ref['passed']]).mean() >= 0.8
    sel = syn._evaluate_raw(raw, opts)
    assert [c for c, _ in sel] == sorted(c for c, _ in sel) and len(sel) == int(raw.passed.sum())
    results = syn.static(opts, save_results=False)
    decs = results.decompositions
    assert len(decs) >= 5                       # restatement: 7 prospective
    for d in decs:
        assert d.loss < 1e-5 and d.cz_count <= 10 and d.type == 'Approximate'
        assert d.circuit.count_ops().get('cp', 0) == 0 and d.cz_depth <= d.cz_count
        assert repr(d).startswith('< ccz_chain| Approximate | loss: ')
    assert min(d.cz_count for d in decs) == 8   # README.md:37-41: an 8-CZ decomposition exists
    text = capsys.readouterr().out
    assert 'Computing raw results...' in text and 'successful. cz counts are:' in text


def test_tutorial_known_answer_K1():
    """Tutorial cell 27: Toffoli, chain, K=14, r=0.00055, 4 samples, seed 0 -> 4 successful,
    cz counts [8, 9, 14, 15]; one sample sits 0.02 inside the 0.2 window, so its 15 may read 14."""
    from cpflow_amd import StaticOptions, Synthesize
    opts = StaticOptions(num_cp_gates=14, accepted_num_cz_gates=20, num_samples=4)
    res = Synthesize([[0, 1], [1, 2]], target_unitary=TOFF3, label='toff3').static(opts, save_results=False)
    counts = sorted(d.cz_count for d in res.decompositions)
    assert len(counts) == 4 and counts[:2] == [8, 9] and counts[2] == 14 and counts[3] in (14, 15)


@pytest.mark.parametrize('cfg', ['cfg2', 'cfg4'])
def test_full_size_properties(cfg):
    """BASELINE-size runs (config 2: 3q CCZ S=1024; config 4: 4q QFT S=2048) checked through
    size-independent properties of the outputs."""
    if cfg == 'cfg2':
        layer, K, target, S = [[0, 1], [1, 2]], 12, CCZ, 1024
    else:
        w = np.exp(2j * np.pi / 16)
        target = np.array([[w ** (j * k) for k in range(16)] for j in range(16)]) / 4
        layer, K, S = [[0, 1], [1, 2], [2, 3]], 24, 2048
    eng = _engine(layer, K, target, 'xyz')
    init = eng.initial_angles(0, S)
    out = _np(eng.optimize(init))
    assert (out['params'][:, 0] == init.cpu().numpy()).all()
    assert (out['regloss'][:, 1] <= out['regloss'][:, 0]).all()
    assert (out['loss'] == out['regloss'] - out['reg']).all()
    assert (out['reg'] >= 0).all() and np.isfinite(out['params']).all()
    bi = out['regloss'].argmin(1)
    lbest = np.take_along_axis(out['loss'], bi[:, None], 1)[:, 0]
    assert (out['passed'] == (lbest <= np.float32(1e-3))).all()
    # re-evaluating the stored best angles reproduces the stored best regloss (another kernel: the generic one applies every
    # rotation as cos / sin products; config 2's and config 4's optimisers run the ZYZ form of the blocks - another
    # factorisation of the same gates in float32: 3.2e-6 / 5.2e-6 at most over the 1024 / 2048 samples, signs mixed.
    # With the hardware's un-centred 1 / sqrt alone the ZYZ form sat 3-6e-6 ABOVE on every sample: cpf_zyz_math.h zrsqrt)
    re = _np(eng.loss_and_grad(out['params'][:, 1], with_grad=False))
    d = re['regloss'] - out['regloss'][:, 1]
    assert np.abs(d).max() < 8e-6 and abs(np.median(d)) < 1e-6, (np.abs(d).max(), np.median(d))
    # CZ count recomputed on the host from the best angles
    cp = out['params'][np.arange(S), bi][:, eng.cp_mask == 1] % np.float32(2 * np.pi)
    v = np.full(cp.shape, 2)
    v[cp < 0.2] = 0
    v[np.abs(cp - np.float32(2 * np.pi)) < 0.2] = 0
    v[np.abs(cp - np.float32(np.pi)) < 0.2] = 1
    assert (v.sum(1) == out['cz']).all()
    if cfg == 'cfg2':
        assert out['passed'].mean() > 0.5
        assert out['cz'][out['passed']].min() <= 8


def test_adaptive_front_end_small(tmp_path):
    """Adaptive routine end to end (own TPE instead of hyperopt): seed chain as the reference, trials saved
    after every evaluation, resume continues the chain, found decompositions verify to target_loss."""
    from cpflow_amd import AdaptiveOptions, Results, Synthesize
    from cpflow_amd import prng
    opts = AdaptiveOptions(min_num_cp_gates=8, max_num_cp_gates=14, max_evals=3, num_samples=24,
                           target_num_cz_gates=8, stop_if_target_reached=False)
    path = str(tmp_path / 'res' / 'toff3')
    syn = Synthesize([[0, 1], [1, 2]], target_unitary=TOFF3, label='toff3')
    res = syn.adaptive(opts, save_to=path)
    assert len(res.trials.results) == 3
    seed = 0
    for r in res.trials.results:
        seed = prng.next_adaptive_seed(seed)
        assert r['random_seed'] == seed and 8 <= r['num_cp_gates'] <= 14
        assert 'prospective_decompositions' not in r
    for d in res.decompositions:
        assert d.loss < 1e-5 and d._adaptive_options is opts
    back = Results.load(path)
    assert len(back.trials.results) == 3
    opts.max_evals = 4
    res2 = Synthesize([[0, 1], [1, 2]], target_unitary=TOFF3, label='toff3').adaptive(opts, save_to=path)
    assert len(res2.trials.results) == 4
    assert res2.trials.results[3]['random_seed'] == prng.next_adaptive_seed(seed)


def test_ansatz_unitary_circuit_and_learn():
    """Ansatz.unitary == the circuit IR's own unitary; a 'cz' template learns a reachable target."""
    from cpflow_amd import Ansatz, fill_layers
    rng = np.random.default_rng(5)
    for ent, rot in (('cp', 'xyz'), ('cz', 'xz'), ('cx', 'xyz'), ('cx', 'y')):
        anz = Ansatz(3, ent, fill_layers([[0, 1], [2, 1], [0, 2]], 5), rot)
        th = rng.uniform(0, 2 * np.pi, anz.num_angles).astype(np.float32)
        u = anz.unitary(th)
        ref = anz.circuit(th).unitary()
        assert np.abs(u - ref).max() < 5e-6
    anz = Ansatz(3, 'cz', fill_layers([[0, 1], [1, 2]], 4))
    target = anz.circuit(rng.uniform(0, 2 * np.pi, anz.num_angles)).unitary()
    out = anz.learn(target, num_iterations=1500, num_repeats=8, keep_history=False)
    assert len(out) == 8 and out[0]['params'].shape == (2, anz.num_angles)
    assert min(o['loss'][1] for o in out) < 1e-3
    hist = anz.learn(target, num_iterations=50, initial_angles=out[0]['params'][0])
    assert hist['params'].shape == (50, anz.num_angles) and hist['loss'].shape == (50,)
    assert (hist['params'][0] == out[0]['params'][0]).all()


def test_state_preparation_and_relative_phase_front_end():
    """Tutorial section 'custom losses': GHZ state preparation on a line (best known: 3 CZ) and the
    relative-phase Toffoli (3 CZ), through the reference-shaped API with loss objects instead of lambdas."""
    from cpflow_amd import RelativePhaseLoss, StaticOptions, Synthesize
    ghz = np.zeros(16)
    ghz[0] = ghz[15] = 1 / np.sqrt(2)
    syn = Synthesize([[0, 1], [1, 2], [2, 3]], target_state=ghz, label='GHZ_4q_linear')
    res = syn.static(StaticOptions(num_cp_gates=4, accepted_num_cz_gates=6, num_samples=64, r=0.002), save_results=False)
    assert res.decompositions and min(d.cz_count for d in res.decompositions) == 3          # the tutorial's result (cell 83): 3 CZ
    for d in res.decompositions:
        assert d.loss < 1e-5 and abs(abs(np.vdot(ghz, d.unitary[:, 0])) - 1) < 1e-5
    syn = Synthesize([[0, 1], [1, 2], [0, 2]], unitary_loss_func=RelativePhaseLoss(TOFF3), label='relative_toff3')
    res = syn.static(StaticOptions(num_cp_gates=6, accepted_num_cz_gates=6, num_samples=64, r=0.002), save_results=False)
    assert res.decompositions and min(d.cz_count for d in res.decompositions) == 3          # the tutorial's result (cell 91): 3 CZ
    d = min(res.decompositions, key=lambda x: x.cz_count)
    m = TOFF3.conj().T @ d.unitary
    assert np.abs(m - np.diag(np.diag(m))).max() < 1e-2        # V^dagger U is diagonal


def test_paper_known_answer_K6_cz_ansatz_success_ratio():
    """peppar.ipynb cells 10-14: anz = Ansatz(3, 'cz', fill_layers(connected_layer(3), 4)); the target is the
    ansatz itself at random_angles(P) (PRNGKey(0)); learn(adam, lr 0.1, 5000 iterations, 100 repeats) reached
    loss < 1e-4 in 31 % of the repeats.  Same PRNG chain here; the ratio is a binomial statistic (sigma 0.046)."""
    from cpflow_amd import Ansatz, connected_layer, fill_layers, prng
    anz = Ansatz(3, 'cz', fill_layers(connected_layer(3), 4))
    assert anz.num_angles == 33
    target_angles = prng.uniform_angles(prng.prng_key(0), anz.num_angles)
    target = anz.unitary(target_angles)
    results = anz.learn(target, method='adam', learning_rate=0.1, num_iterations=5000, num_repeats=100,
                        keep_history=False)
    ratio = np.mean([r['loss'].min() < 1e-4 for r in results])
    assert abs(ratio - 0.31) <= 0.1, ratio        # measured here: 0.31


def test_rccl_collective_path_single_rank():
    """The all-gather runs through RCCL (backend "nccl") on device tensors: one rank here, the same code path as N ranks."""
    import torch
    import torch.distributed as dist
    from cpflow_amd import distributed
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        eng = _engine([[0, 1], [1, 2]], 4, CCZ, 'xyz', num_gd_iterations=20)
        out = eng.optimize(eng.initial_angles(0, 9))
        fields = {k: out[k] for k in ('params', 'regloss', 'loss', 'reg', 'cz', 'passed')}
        got = distributed.gather_samples(fields, 9, force_collective=True)
        for k in fields:
            assert got[k].dtype == fields[k].dtype and torch.equal(got[k], fields[k])
        info = distributed.describe_job(eng.device)           # the object all-gather of the bench line, over RCCL
        assert info['backend'] == 'nccl' and info['world_size'] == 1 and info['distinct_devices'] == 1
        assert info['ranks'][0]['device_index'] == eng.device.index and info['ranks'][0]['uuid']
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('name', ['raw_toff3_chain_t49', 'raw_toff3_chain_t19', 'raw_toff4_star_t14', 'raw_toff4_star_t49',
                                  'raw_toff3_conn_t99', 'raw_4gt10_xz_t30', 'raw_4gt13_xyz_t99'])
def test_T0_T1_all_golden_fixtures(name):
    """Every stored trial (3q / 4q / 5q, 'xyz' / 'xz'): the device draws the reference's initial angles bit for bit
    and reproduces its loss / reg / regloss at the stored {initial, best} angles (float32, 1.5e-5)."""
    g = np.load(os.path.join(GOLD, name + '.npz'))
    eng = _engine(g['layer'], int(g['num_cp_gates']), g['target'], str(g['rotation_gates']), r=float(g['r']))
    S, idx = int(g['num_samples']), g['sample_index']
    lo, hi = int(idx.min()), int(idx.max()) + 1
    rows = eng.initial_angles(int(g['seed']), S, first=lo, count=hi - lo).cpu().numpy()
    assert (rows[idx - lo].view(np.uint32) == g['params'][:, 0].view(np.uint32)).all()
    # float32 noise grows with depth: 1.5e-5 up to 40 blocks; the 5-qubit K = 57 / 86 trials get 3e-5
    # (the survey's own restatement saw 1.2e-5 on the K = 86 file)
    tol = 1.5e-5 if int(g['num_cp_gates']) <= 40 else 3e-5
    for k in (0, 1):
        at = _np(eng.loss_and_grad(g['params'][:, k, :], with_grad=False))
        assert np.abs(at['loss'] - g['loss'][:, k]).max() < tol
        if bool(g['pin_reg']):
            assert np.abs(at['reg'] - g['reg'][:, k]).max() < 1e-7
            assert np.abs(at['regloss'] - g['regloss'][:, k]).max() < tol
    # the stored CZ counts follow from the stored best angles (device evaluation with zero iterations)
    ev = _engine(g['layer'], int(g['num_cp_gates']), g['target'], str(g['rotation_gates']), r=float(g['r']),
                 num_gd_iterations=0)
    best = g['params'][np.arange(len(idx)), g['regloss'].argmin(1)]
    assert (_np(ev.optimize(best))['cz'] == g['cz']).all()


def test_integration_stub_from_the_document_runs():
    """The ctypes stub printed in INTEGRATION.md (what a cpflow maintainer would paste) is executed as written
    against the built library and must agree with the engine bit for bit."""
    import re
    from cpflow_amd import StaticOptions, _ffi
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, 'INTEGRATION.md')).read()
    code = re.search(r'```python\n(import ctypes as C.*?)```', text, re.S).group(1)
    code = code.replace("C.CDLL('libcpflow_amd.so')", f"C.CDLL({_ffi.LIB_PATH!r})")
    ns = {}
    exec(code, ns)
    opts = StaticOptions(num_cp_gates=12, accepted_num_cz_gates=10, num_samples=10, num_gd_iterations=300)
    layer = [[0, 1], [1, 2]]
    raw, cz, passed = ns['generate_raw'](layer, 3, CCZ, opts)
    eng = _engine(layer, 12, CCZ, 'xyz', num_gd_iterations=300)
    ref = _np(eng.optimize(eng.initial_angles(0, 10)))
    assert len(raw) == 10 and set(raw[0]) == {'params', 'loss', 'reg', 'regloss'}
    for s in range(10):
        assert (raw[s]['params'] == ref['params'][s]).all() and (raw[s]['regloss'] == ref['regloss'][s]).all()
    assert (cz == ref['cz']).all() and (passed == ref['passed']).all()


def test_two_ranks_on_one_gpu_equal_single_process(tmp_path):
    """The N > 1 path end to end on one GPU: two worker processes (gloo rendezvous, both on cuda:0) run the real front
    end - raw stage sharded by sample blocks, ONE gather, verification batch sharded too - and what rank 0 ends up with is
    bit for bit what a single process computes: 37 samples (uneven shards 19 + 18), the same kernel variant either way."""
    import socket
    import subprocess
    import sys
    worker = os.path.join(os.path.dirname(__file__), 'dist_worker.py')
    S = 37
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    subprocess.check_call([sys.executable, worker, str(tmp_path / 'one.npz'), str(S)], env=env)
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    procs = [subprocess.Popen([sys.executable, worker, str(tmp_path / 'two.npz'), str(S)],
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port)))
             for r in range(2)]
    assert [p.wait(timeout=600) for p in procs] == [0, 0]
    one, two = np.load(tmp_path / 'one.npz'), np.load(tmp_path / 'two.npz')
    for k in one.files:
        assert one[k].shape == two[k].shape and (one[k] == two[k]).all(), k
    assert len(one['dec_cz']) >= 5 and one['passed'].sum() >= 20 and (one['dec_loss'] < 1e-5).all()


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` without torchrun starts two ranks itself (here both on the one GPU: CPF_ONE_DEVICE=1 ->
    gloo) and reports n_gpus = 2 with the whole-job sample count; with --strong-ref the line also carries the one-GPU time of the whole job."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT')}
    env['CPF_ONE_DEVICE'] = '1'
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--samples-per-gpu', '256',
                          '--strong-ref'], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line['n_gpus'] == 2 and line['scaling'] == 'weak' and '512 total' in line['config']['workload']
    assert line['value'] > 0 and line['roofline']['bound'] == 'valu' and 'cpu_baseline' not in line
    # the line describes the job torch.distributed ran: here gloo, two ranks, both on the box's one GPU
    di = line['distributed']
    assert di['backend'] == 'gloo' and di['world_size'] == 2 and [e['rank'] for e in di['ranks']] == [0, 1] and di['distinct_devices'] == 1
    # ... and separates, per rank, the kernel's own time, the time inside the ONE collective of a step (where a rank that arrives early
    # waits for the slowest one) and the rank's wall time per step: launch skew, RCCL latency and kernel time in one line
    pr = di['per_rank']
    assert [e['rank'] for e in pr] == [0, 1] and di['collective']['bytes_per_rank'] == 256 * 6 * 4
    for e in pr:
        assert e['kernel_ms'] > 0 and 0 < e['collective_ms_min'] <= e['collective_ms'] <= e['collective_ms_max']
        assert e['elapsed_ms_per_step'] >= e['kernel_ms'] and e['elapsed_ms_per_step'] <= line['ms_per_step'] * (1 + 1e-9)
    # --strong-ref: rank 0 also ran the whole job's 512 samples on its one GPU (the strong-scaling reference of the same line)
    sr = line['strong_ref']
    assert sr['samples'] == 512 and sr['ms_per_step_one_gpu'] > 0 and abs(sr['strong_scaling_efficiency'] - sr['ms_per_step_one_gpu'] / (2 * line['ms_per_step'])) < 1e-9


def _score(cz_passed, S):
    """main.py:831-835: -log2 of the mean of 2^-cz over all samples (failed samples contribute 0)."""
    return float(-np.log2((2.0 ** -np.asarray(cz_passed, dtype=np.float32)).sum() / S)) if len(cz_passed) else float('inf')


# (fixture, kernel variant the launch must use, |pass count - stored| allowed, stored passes that must pass here too,
#  CZ-histogram samples that may move, score tolerance)
T3_MORE = [
    # G3, 3 qubits all-to-all, 200 samples: 198 stored; float32 / float64 oracle: 199 passes, all 198 stored ones among them.
    # Its stored reg values come from older penalty constants, which moves a few CZ counts, not the pass set
    ('raw_toff3_conn_t99', 'zyz3/jit(hipcc) waves=2 K cut=5', 3, 195, 14, 0.15),             # 10 blocks: a shipped K-specialised kernel
    # G2 trial 49, 4-qubit star, 500 samples, 7 stored passes (the oracle finds the same 7)
    ('raw_toff4_star_t49', 'zyzl/jit(hipcc)', 4, 5, 4, 1.6),                 # 500 samples: one wave per sample, the ZYZ form
    # G5, 5-qubit chain 'xyz', 86 blocks, 1000 samples, 39 stored passes.  The stored run used OLDER penalty constants (its
    # stored reg values are 1.4-1.7 x what penalty.py:44-71 gives at the same angles; SURVEY App. D: "don't pin reg"), and a
    # heavier penalty leaves fewer runs below entry_loss: the float32 oracle with today's penalty passes 8 of the first 160
    # samples (5.0 %), the kernel 88 of 1000 (8.8 %), the stored run 39 (3.9 %; 6 of the first 160).  So against the stored
    # numbers only the order of magnitude is asserted here; the kernel-vs-oracle statistics of this very problem are in
    # tests/test_gpu_baseline_configs.py::test_stored_5q_trial_unbiased_against_float32_oracle
    ('raw_4gt13_xyz_t99', 'zyzl/jit(hipcc)', 70, 0, None, 4.0),               # one wave per sample: the ZYZ form of the gate-by-gate layout
    # G4, 5 qubits all-to-all 'xz', 57 blocks, 2000 samples, ONE stored pass
    ('raw_4gt10_xz_t30', 'zyzl/jit(hipcc)', 12, 0, None, None),
]


@pytest.mark.parametrize('name,variant,dpass,min_common,moved_max,score_tol', T3_MORE, ids=[t[0][4:] for t in T3_MORE])
def test_T3_more_stored_trials_end_of_run(name, variant, dpass, min_common, moved_max, score_tol, monkeypatch):
    """Full 2000-iteration runs of the remaining stored trials (3 qubits all-to-all, 4-qubit star, the two 5-qubit
    benchmark files - the only reference data that exercise the 5-qubit kernels end to end), from the shipped kernel cache
    (CPF_JIT_FORBID_COMPILE=1).  Bars: the number of samples below entry_loss against the stored number (3 sigma of the
    difference of two binomial counts where the stored run used today's penalty; Poisson for the single stored pass of
    G4; order of magnitude for G5, see T3_MORE), the stored passes found again where runs are reproducible (3 qubits), the CZ histogram
    and the adaptive score main.py:831-835 computes from it."""
    monkeypatch.setenv('CPF_JIT_FORBID_COMPILE', '1')
    g = np.load(os.path.join(GOLD, name + '.npz'))
    S, idx = int(g['num_samples']), g['sample_index']
    eng = _engine(g['layer'], int(g['num_cp_gates']), g['target'], str(g['rotation_gates']), r=float(g['r']))
    init = eng.initial_angles(int(g['seed']), S)
    out = _np(eng.optimize(init))
    assert eng.last_variant().startswith(variant), eng.last_variant()
    assert (out['params'][idx, 0].view(np.uint32) == g['params'][:, 0].view(np.uint32)).all()   # T0
    passed = np.flatnonzero(out['passed'])
    common = len(set(passed.tolist()) & set(idx.tolist()))
    cz = out['cz'][out['passed']]
    score = _score(cz, S)
    print(f'{name} [{eng.last_variant()}]: {len(passed)} passed (stored {len(idx)}, {common} in common); CZ {np.bincount(cz).tolist() if len(cz) else []} '
          f'vs stored {np.bincount(g["cz"]).tolist()}; score {score:.3f} (stored {float(g["score"]):.3f})')
    assert abs(len(passed) - len(idx)) <= dpass
    assert common >= min_common
    if moved_max is not None:
        top = max(int(cz.max()), int(g['cz'].max())) + 1
        moved = np.abs(np.bincount(cz, minlength=top) - np.bincount(g['cz'], minlength=top)).sum() / 2
        assert moved <= moved_max, moved
    if score_tol is not None:
        assert abs(score - float(g['score'])) < score_tol
    if name == 'raw_4gt13_xyz_t99':          # distribution level: the converged runs use as many CZ gates as the stored ones
        assert abs(np.median(cz) - np.median(g['cz'])) <= 6 and abs(int(cz.min()) - int(g['cz'].min())) <= 6
    _check_sizeless(eng, init, out, S)


def _check_sizeless(eng, init, out, S):
    assert (out['regloss'][:, 1] <= out['regloss'][:, 0]).all() and (out['loss'] == out['regloss'] - out['reg']).all()
    re = _np(eng.loss_and_grad(out['params'][:, 1], with_grad=False))
    assert np.abs(re['regloss'] - out['regloss'][:, 1]).max() < 1e-5


@pytest.mark.parametrize('name', ['K3_ghz_state_preparation', 'K4_relative_phase_toffoli'])
def test_tutorial_known_answers_K3_K4_custom_losses(name):
    """Tutorial cells 81 / 89 through the reference-shaped front end with the loss OBJECTS that stand for the tutorial's
    lambdas: Synthesize(layer, target_state=GHZ) / Synthesize(layer, unitary_loss_func=RelativePhaseLoss(toffoli)),
    _generate_raw + _evaluate_raw at the (num_cp_gates, r, seed) of the notebook's first productive adaptive trial.  The
    notebook printed 83 prospective results (52 x 3, 30 x 4, 1 x 5 CZ; score 3.5724) and 100 prospective (score 5.2313);
    same bars as the CPU oracle test (tests/test_oracle_golden.py)."""
    from cpflow_amd import RelativePhaseLoss, StaticOptions, Synthesize
    from test_oracle_golden import check_known_answer, known_answer_case
    layer, K, r, seed, S, score, cz_ref, target, kind = known_answer_case(GOLD, name)
    if kind == 1:
        syn = Synthesize(layer, target_state=target[:, 0], label='GHZ_4q_linear')
    else:
        syn = Synthesize(layer, unitary_loss_func=RelativePhaseLoss(target), label='relative_toff3_connected')
    opts = StaticOptions(num_cp_gates=K, r=r, accepted_num_cz_gates=int(cz_ref.max()), num_samples=S, random_seed=seed)
    raw = syn._generate_raw(opts)
    sel = syn._evaluate_raw(raw, opts)
    check_known_answer(np.array([c for c, _ in sel]), S, score, cz_ref, name + ' (device)')
    # against the oracle on the same seed: same pass flags, same CZ counts except boundary samples
    orc = Oracle(layer, K, target, 'xyz', r=r, loss_kind=kind)
    ref = orc.run(orc.initial_angles(seed, S), np.float32)
    assert np.mean(raw.passed == ref['passed']) >= 0.97
    both = raw.passed & ref['passed']
    assert np.mean(raw.cz[both] == ref['cz'][both]) >= 0.9
    # the verification stage on the best candidates reaches the tutorial's result: 3 CZ gates, loss ~ 0
    best = [res for res in sel if res[0] <= 3][:4]
    ok = [v for v in syn._verify_batch(best, K, opts) if v[0]]
    assert ok and min(v[1] for v in ok) == 3


def test_adaptive_run_against_the_stored_hyperopt_trials(tmp_path):
    """SURVEY 8 f-4 on the device: Synthesize.adaptive on the tutorial's toff3_chain problem (range [10, 20], 100 samples,
    seed 0: the range the stored draws imply, tests/test_host_logic.py) for 50 evaluations against the stored run (tutorial/results/toff3_chain, 50 hyperopt
    trials).  The first 20 evaluations use the reference's own (seed, num_cp_gates, r) (start-up phase, reproduced exactly
    by cpflow_amd.hyper.suggest), so their scores must be the stored ones up to the run-to-run noise of a score (one
    boundary sample moving between CZ counts changes it by ~0.01-0.05): >= 17 of 20 within 0.1, all within 0.5.  The
    remaining 30 come from the own estimator (hyperopt's TPE sequence is not reproduced): the best score reached must
    lie within the range the stored run explored there."""
    from cpflow_amd import AdaptiveOptions, Synthesize
    a = np.load(os.path.join(GOLD, 'adaptive_toff3_chain.npz'))
    opts = AdaptiveOptions(min_num_cp_gates=10, max_num_cp_gates=20, max_evals=50, num_samples=100, target_num_cz_gates=6)
    res = Synthesize([[0, 1], [1, 2]], target_unitary=TOFF3, label='toff3_chain').adaptive(opts, save_to=str(tmp_path / 'r' / 'toff3'))
    tr = res.trials.results
    assert len(tr) == 50
    d = []
    for i in range(20):
        assert tr[i]['random_seed'] == int(a['seeds'][i]) and tr[i]['num_cp_gates'] == int(a['num_cp_gates'][i])
        assert abs(tr[i]['r'] / float(a['r'][i]) - 1) < 4e-16
        d.append(abs(tr[i]['loss'] - float(a['score'][i])))
    print('start-up scores: |d| sorted', np.round(sorted(d), 3).tolist())
    assert sum(x < 0.1 for x in d) >= 17 and max(d) < 0.5
    best, stored = min(t['loss'] for t in tr), a['score']
    print(f'best score {best:.3f}; stored run: best {stored.min():.3f}, median {np.median(stored):.3f}')
    assert stored.min() - 0.3 <= best <= np.median(stored)
    assert min(d.cz_count for d in res.decompositions) <= 8      # the tutorial's run ends with 8 CZ gates on the chain
