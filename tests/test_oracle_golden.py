"""CPU: pin the oracle (oracle/) against vectors the reference itself produced (tests/golden/*.npz,
mined by tools/extract_golden.py from /root/reference/{tutorial,paper}/results)."""
import glob
import os

import numpy as np
import pytest

import oracle
from oracle import Oracle, cost_hst, numpy_unitary, prng

RAW = sorted(glob.glob(os.path.join(os.path.dirname(__file__), 'golden', 'raw_*.npz')))
DEC = sorted(glob.glob(os.path.join(os.path.dirname(__file__), 'golden', 'dec_*.npz')))


def _oracle_for(g, **kw):
    return Oracle(g['layer'], int(g['num_cp_gates']), g['target'], str(g['rotation_gates']), r=float(g['r']), **kw)


@pytest.mark.parametrize('path', RAW, ids=[os.path.basename(p)[4:-4] for p in RAW])
def test_T0_initial_angles_bit_exact(path):
    """T0: the threefry restatement reproduces the stored initial angle rows bit for bit
    (NumPy restatement and the C one)."""
    g = np.load(path)
    S, P = int(g['num_samples']), g['params'].shape[2]
    idx = g['sample_index']
    ref = g['params'][:, 0, :]
    if S * P <= 300_000:
        a = prng.initial_angles(int(g['seed']), S, P)
        assert (a[idx].view(np.uint32) == ref.view(np.uint32)).all()
    orc = _oracle_for(g)
    for j, row in zip(idx, ref):
        c = orc.initial_angles(int(g['seed']), S, first=int(j), count=1)[0]
        assert (c.view(np.uint32) == row.view(np.uint32)).all()


@pytest.mark.parametrize('path', RAW, ids=[os.path.basename(p)[4:-4] for p in RAW])
def test_T1_loss_and_reg_at_stored_angles(path):
    """T1: loss / reg / regloss at the stored {initial, best} angles within 1e-5 (float32)."""
    g = np.load(path)
    orc = _oracle_for(g)
    for k in (0, 1):
        th = g['params'][:, k, :]
        for dtype in (np.float32, np.float64):
            res = orc.loss_grad(th, dtype, with_grad=False)
            # stored loss is regloss - reg in float32, so it carries ~1e-6 of noise itself
            assert np.abs(res['loss'] - g['loss'][:, k]).max() < 1.5e-5
            if bool(g['pin_reg']):
                assert np.abs(res['reg'] - g['reg'][:, k]).max() < 1e-8
                assert np.abs(res['regloss'] - g['regloss'][:, k]).max() < 1.5e-5


@pytest.mark.parametrize('path', RAW[:1] + RAW[3:5], ids=lambda p: os.path.basename(p)[4:-4])
def test_stored_cz_counts(path):
    """count_cz at argmin regloss equals the CZ count the reference stored next to each result."""
    g = np.load(path)
    orc = _oracle_for(g)
    for i in range(len(g['cz'])):
        best = int(np.argmin(g['regloss'][i]))
        assert orc.count_cz(g['params'][i, best]) == int(g['cz'][i])
    score = -np.log2((2.0 ** -g['cz_counts'].astype(np.float32)).sum() / int(g['num_samples']))
    assert abs(score - float(g['score'])) < 1e-5


@pytest.mark.parametrize('path', DEC, ids=[os.path.basename(p)[4:-4] for p in DEC])
def test_G6_verified_angles_rebuild_target(path):
    """Post-verification angle vectors of stored decompositions hit their target (loss <= 1e-6 + fp32 noise)."""
    g = np.load(path)
    for i in range(len(g['angles'])):
        K, rot = int(g['num_cp_gates'][i]), str(g['rotation_gates'][i])
        orc = Oracle(g['layer'], K, g['target'], rot, use_reg=False)
        th = g['angles'][i, :orc.P]
        res = orc.loss_grad(th, np.float64, with_grad=False)
        assert res['loss'][0] < 2e-6
        idx = g['frozen_idx'][i]
        idx = idx[idx >= 0]
        assert orc.cp_mask[idx].all()          # only CP angles were frozen
        assert orc.count_cz(th) <= int(g['cz_count'][i])


def test_conventions_against_literal_tensordot():
    """C oracle == literal tensordot/transpose restatement on random circuits incl. twisted placements."""
    rng = np.random.default_rng(0)
    for layer, K, rot in [([[0, 1], [1, 2]], 5, 'xyz'), ([[2, 0], [1, 3], [3, 2]], 7, 'xz'),
                          ([[4, 1], [0, 3], [2, 4]], 4, 'zy')]:
        n = max(max(p) for p in layer) + 1
        orc = Oracle(layer, K, np.eye(2 ** n), rot)
        th = rng.uniform(-4, 9, orc.P)
        assert np.abs(orc.unitary(th, np.float64) - numpy_unitary(layer, K, th, rot)).max() < 1e-12
        u = orc.unitary(th, np.float32)
        assert np.abs(u.conj().T @ u - np.eye(2 ** n)).max() < 1e-5


def test_gradient_against_finite_differences():
    rng = np.random.default_rng(1)
    layer, K = [[0, 1], [1, 2], [2, 0]], 6
    tgt = np.linalg.qr(rng.normal(size=(8, 8)) + 1j * rng.normal(size=(8, 8)))[0]
    orc = Oracle(layer, K, tgt, 'xyz', r=0.01)
    th = rng.uniform(0, 2 * np.pi, orc.P)
    g = orc.loss_grad(th, np.float64)['grad'][0]
    eps = 1e-6
    for i in range(orc.P):
        tp, tm = th.copy(), th.copy()
        tp[i] += eps
        tm[i] -= eps
        fd = (orc.loss_grad(tp, np.float64, False)['regloss'][0] - orc.loss_grad(tm, np.float64, False)['regloss'][0]) / (2 * eps)
        assert abs(fd - g[i]) < 1e-7


def test_penalty_shape():
    """penalty.py:44-71: 0 at 0 / 2pi, 2 at pi/2 and 3pi/2, 1 at pi, plateaus of half-width 0.05, floored mod."""
    orc = Oracle([[0, 1]], 1, np.eye(4))
    val = lambda a: orc.penalty(a, np.float64)[0]  # noqa: E731
    assert val(0.0) == 0 and val(0.04) == 0 and abs(val(np.pi / 2) - 2) < 1e-12 and abs(val(np.pi) - 1) < 1e-12
    assert abs(val(np.pi + 0.03) - 1) < 1e-12 and abs(val(1.5 * np.pi - 0.02) - 2) < 1e-12
    assert abs(val(-0.3) - val(2 * np.pi - 0.3)) < 1e-9 and abs(val(9.0) - val(9.0 - 2 * np.pi)) < 1e-9
    assert abs(orc.penalty(0.8, np.float64)[1] - 2 / (np.pi / 2 - 0.1)) < 1e-9


def test_adaptive_seed_chain(golden_dir):
    for name in ('adaptive_toff3_chain.npz', 'adaptive_toff4_star.npz'):
        g = np.load(os.path.join(golden_dir, name))
        seed = int(g['initial_seed'])
        for stored in g['seeds']:
            seed = prng.next_adaptive_seed(seed)
            assert seed == int(stored)


def test_T3_end_of_run_statistics_G1(golden_dir):
    """T3: a full 2000-step float32 run reproduces the reference's pass/fail set and
    per-sample CZ counts on G1 (toff3_chain trial 49); best regloss agrees statistically."""
    g = np.load(os.path.join(golden_dir, 'raw_toff3_chain_t49.npz'))
    orc = _oracle_for(g)
    init = orc.initial_angles(int(g['seed']), int(g['num_samples']))
    out = orc.run(init, np.float32)
    idx = g['sample_index']
    assert sorted(np.flatnonzero(out['passed'])) == sorted(idx)          # same 99 of 100 pass (sample 72 fails)
    assert (out['cz'][idx] == g['cz']).mean() >= 0.97                    # boundary samples may flip
    d = np.abs(out['regloss'][idx, 1] - g['regloss'][:, 1])
    assert np.median(d) < 1.5e-5 and d.max() < 5e-4
    sel = oracle.filter_results(out)
    assert [c for c, _ in sel] == sorted(c for c, _ in sel)
