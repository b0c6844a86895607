"""GPU: every BASELINE.json configuration at its real size and target, the kernel-vs-float32-oracle
unbiasedness statistics on config 2, and the paper's local-minimum success ratios."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import Oracle  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def _np(d):
    return {k: v.cpu().numpy() for k, v in d.items() if v is not None}


def _engine(wl, **kw):
    from cpflow_amd.engine import TemplateEngine
    kw.setdefault('num_gd_iterations', wl.num_gd_iterations)
    return TemplateEngine(wl.layer, wl.num_cp_gates, wl.target, wl.rotation_gates, r=wl.r, **kw)


def _check_size_independent_properties(eng, init, out, S):
    """What must hold for any launch, whatever its size (cpflow/optimization.py:28-94, 365-370; cp_utils.py:45-67, 144-164)."""
    assert (out['params'][:, 0] == init.cpu().numpy()).all()                      # row 0 = the initial angles, bit for bit
    assert (out['regloss'][:, 1] <= out['regloss'][:, 0]).all()                   # best <= first
    assert (out['loss'] == out['regloss'] - out['reg']).all()                     # loss = regloss - reg in float32
    assert (out['reg'] >= 0).all() and np.isfinite(out['params']).all() and np.isfinite(out['regloss']).all()
    bi = out['regloss'].argmin(1)
    lbest = np.take_along_axis(out['loss'], bi[:, None], 1)[:, 0]
    assert (out['passed'] == (lbest <= np.float32(1e-3))).all()                   # entry_loss flag
    re = _np(eng.loss_and_grad(out['params'][:, 1], with_grad=False))             # the stored best angles give the stored best regloss
    # (another kernel and another factorisation of the same gates in float32: the generic kernel applies every rotation as cos / sin
    # products, the optimisers run the ZYZ form of the blocks - 2.7e-6 at most over config 3's 512 samples of 16 blocks, signs mixed)
    # The bound is on the 99th percentile as well as on the worst sample: the maximum over 4096 samples is a tail statistic that moves
    # with the trajectories (same box, config 3 at 4096 samples: round 5's kernel max 3.6e-6, 99.9 % 3.0e-6, 99 % 2.4e-6, median 6.6e-7;
    # round 6's max 4.4e-6, 99.9 % 3.4e-6, 99 % 2.5e-6, median 6.6e-7, the same signed mean 4.3e-7).  (cos, sin) from v_sin_f32 /
    # v_cos_f32 was measured in the 3-qubit kernel and not adopted: cpf_zyz3.h CPF_Z3_HWSINCOS.)
    d = np.abs(re['regloss'] - out['regloss'][:, 1])
    assert np.quantile(d, 0.99) < (3e-6 if eng.num_qubits <= 4 else 8e-6) and d.max() < (6e-6 if eng.num_qubits <= 4 else 1e-5)
    cp = out['params'][np.arange(S), bi][:, eng.cp_mask == 1] % np.float32(2 * np.pi)
    v = np.full(cp.shape, 2)
    v[cp < 0.2] = 0
    v[np.abs(cp - np.float32(2 * np.pi)) < 0.2] = 0
    v[np.abs(cp - np.float32(np.pi)) < 0.2] = 1
    assert (v.sum(1) == out['cz']).all()                                          # CZ count recomputed on the host


def _short_run_parity(wl, S, T, tol):
    """The real template and target of the configuration over a short run with history against the float64 oracle."""
    orc = Oracle(wl.layer, wl.num_cp_gates, wl.target, wl.rotation_gates, num_gd_iterations=T, r=wl.r)
    eng = _engine(wl, num_gd_iterations=T)
    init = orc.initial_angles(wl.seed, S)
    assert (eng.initial_angles(wl.seed, S).cpu().numpy().view(np.uint32) == init.view(np.uint32)).all()
    out = _np(eng.optimize(init, keep_history=True))
    variant = eng.last_variant()
    ref = orc.run(init, np.float64, keep_history=True)
    assert np.abs(out['history_regloss'][:, 0] - ref['history_loss'][:, 0]).max() < tol / 2      # loss at the initial angles
    assert np.abs(out['history_regloss'] - ref['history_loss']).max() < 5 * tol
    assert np.abs(out['history_params'] - ref['history_params'])[:, :3].max() < 10 * tol          # = the first gradients
    assert np.abs(out['reg'] - ref['reg']).max() < 1e-6
    assert (out['cz'] == ref['cz']).all()
    lg = _np(eng.loss_and_grad(init))
    rg = orc.loss_grad(init, np.float64)
    assert np.abs(lg['loss'] - rg['loss']).max() < tol and np.abs(lg['grad'] - rg['grad']).max() < tol
    return variant


def test_config3_toffoli_all_to_all_at_size():
    """BASELINE config 3: Toffoli, all-to-all, 16 CP gates; its per-GPU share (512 of 4096 samples on 8 GPUs) and the
    whole 4096 on one GPU, 2000 iterations; plus short-run oracle parity on the same target."""
    from cpflow_amd import workloads
    wl = workloads.CONFIGS['cfg3']
    assert _short_run_parity(wl, S=5, T=14, tol=1e-5) == 'zyz3/aot waves=2 K cut=8'
    eng = _engine(wl)
    stats = []
    for S, first in ((512, 1024), (4096, 0)):          # rank 2's block of the 8-GPU run; all samples on one GPU
        init = eng.initial_angles(wl.seed, 4096, first, S)
        out = _np(eng.optimize(init))
        # 512 samples = two per CU (one wave on every SIMD): 4 of the 16 blocks in wave A; 4096 = four rounds of four per CU: 7
        assert eng.last_variant() == f'zyz3/aot waves=2 K cut={4 if S == 512 else 7}', eng.last_variant()
        _check_size_independent_properties(eng, init, out, S)
        ok = out['passed']
        assert ok.mean() > 0.5 and out['cz'][ok].min() <= 8          # Toffoli needs 6 CZ all-to-all; most starts converge
        stats.append((out['passed'][1024:1536] if S == 4096 else out['passed'], out['cz'][1024:1536] if S == 4096 else out['cz'],
                      out['params'][1024:1536, 0] if S == 4096 else out['params'][:, 0]))
    # rows 1024..1535 of the 4096-sample run are the 512-sample share: the same initial angles bit for bit; the two launches
    # use different kernel variants (stored states or not), so their 2000-step trajectories agree statistically, not bitwise
    (p0, c0, i0), (p1, c1, i1) = stats
    assert (i0 == i1).all() and np.mean(p0 == p1) > 0.97 and np.mean(c0[p0 & p1] == c1[p0 & p1]) > 0.9


def test_config5_su64_ring_at_size():
    """BASELINE config 5: 6 qubits, ring, 60 CP gates, 438 angles, the committed SU(64) target; one GPU's share of the
    8-GPU run (1024 of 8192 samples), 2000 iterations; plus short-run oracle parity at the real K = 60."""
    from cpflow_amd import workloads
    wl = workloads.CONFIGS['cfg5']
    u = wl.target
    assert u.shape == (64, 64) and np.abs(u.conj().T @ u - np.eye(64)).max() < 1e-12 and abs(np.linalg.det(u) - 1) < 1e-9
    variant = _short_run_parity(wl, S=3, T=10, tol=2e-5)
    assert variant == 'zyz/aot waves=4', variant            # three samples: four waves per sample
    eng = _engine(wl)
    assert eng.num_angles == 438
    S = 1024
    init = eng.initial_angles(wl.seed, 8192, 3 * S, S)      # rank 3's block
    out = _np(eng.optimize(init))
    assert eng.last_variant() == 'zyz/aot waves=2', eng.last_variant()      # one GPU's share: two
    _check_size_independent_properties(eng, init, out, S)
    # a generic SU(64) needs ~1000 CZ gates: 60 blocks cannot reach it, the loss only falls
    assert not out['passed'].any() and (out['loss'][:, 1] < out['loss'][:, 0]).all() and out['loss'][:, 1].max() < 1.0


def test_config4_qft_short_run_parity():
    """BASELINE config 4 (4-qubit QFT, chain, 24 CP gates) at its real target over a short run (the full-size run is
    test_full_size_properties[cfg4])."""
    from cpflow_amd import workloads
    variant = _short_run_parity(workloads.CONFIGS['cfg4'], S=5, T=12, tol=1e-5)
    assert variant.split(' ')[0] in ('layered/aot', 'dense/aot', 'zyz/aot', 'layered/jit(hipcc)', 'dense/jit(hipcc)', 'zyz/jit(hipcc)'), variant     # 5 samples: 4 waves per sample


def _chi2_two_sample(a, b, bins):
    """Two-sample chi-square statistic of integer samples a, b over `bins` (last bin open); returns (stat, dof)."""
    ha = np.array([(a == k).sum() for k in bins[:-1]] + [(a >= bins[-1]).sum()], dtype=np.float64)
    hb = np.array([(b == k).sum() for k in bins[:-1]] + [(b >= bins[-1]).sum()], dtype=np.float64)
    keep = (ha + hb) >= 10                                 # pool sparse bins into one
    ha = np.append(ha[keep], ha[~keep].sum())
    hb = np.append(hb[keep], hb[~keep].sum())
    m = (ha + hb) > 0
    ha, hb = ha[m], hb[m]
    ka, kb = np.sqrt(hb.sum() / ha.sum()), np.sqrt(ha.sum() / hb.sum())
    return float((((ka * ha - kb * hb) ** 2) / (ha + hb)).sum()), int(m.sum() - 1)


def _ks_two_sample_p(a, b):
    from scipy.stats import ks_2samp
    return float(ks_2samp(a, b).pvalue)


_ORACLE_RUNS = {}          # (workload, samples, seed) -> the float32 oracle's run: several kernel variants are compared with the same one


def _oracle_run(wl, init, T):
    key = (wl.name, init.shape[0], int(wl.seed), T)
    if key not in _ORACLE_RUNS:
        orc = Oracle(wl.layer, wl.num_cp_gates, wl.target, wl.rotation_gates, num_gd_iterations=T, r=wl.r)
        _ORACLE_RUNS[key] = orc.run(init, np.float32, nthreads=int(orc.lib.cpfo_max_threads()))
    return _ORACLE_RUNS[key]


def _unbiased_against_float32_oracle(wl, S, cz_bins, same_cz_min, q_tol, pair_tol, expect_variant=None, S_total=None, same_flag_min=0.95, pooled=None):
    """One BASELINE configuration at full length (S samples x 2000 iterations) on the kernel and on the float32 oracle
    (same initial angles, all host cores).  Individual runs are chaotic (lr 0.1 for 2000 steps), so the comparison is
    statistical: it would catch a systematic drift from fast_sincos, the rcp / sqrt Adam step, -fno-signed-zeros, a wrong
    gradient scale or a biased gradient of one kernel family, none of which a short-run test sees.  Bounds: pass count
    within 3 sigma of a binomial with the pooled rate; >= same_flag_min (95 %) of the samples get the same pass flag; CZ-count histograms
    of the converged samples compatible (two-sample chi-square below the 99.9 % quantile) and >= same_cz_min of the
    commonly converged samples with the same CZ count; quantiles of the best regularised loss over ALL samples within
    q_tol; median paired difference below pair_tol."""
    T = wl.num_gd_iterations
    eng = _engine(wl)
    init = eng.initial_angles(wl.seed, S_total or S, 0, S)
    out = _np(eng.optimize(init))
    variant = eng.last_variant()
    if expect_variant:
        assert variant.startswith(expect_variant), variant
    ref = _oracle_run(wl, init.cpu().numpy(), T)
    pg, po = out['passed'], ref['passed'].astype(bool)
    if pooled is not None:                                    # several seeds of one problem: the caller asserts on the pooled counts
        pooled.append((int(pg.sum()), int(po.sum()), S, out['regloss'][:, 1].copy(), ref['regloss'][:, 1].copy()))
    p = (pg.sum() + po.sum()) / (2 * S)
    sigma = np.sqrt(2 * S * p * (1 - p))                      # sd of the difference of two binomial counts
    print(f'[{wl.name} {variant}] passed: kernel {pg.sum()} oracle {po.sum()} (3 sigma = {3 * sigma:.1f}); same flag {np.mean(pg == po):.3f}')
    assert abs(int(pg.sum()) - int(po.sum())) <= max((3 if pooled is None else 4) * sigma, 1)      # pooled runs: the 3-sigma bar is on the pool
    assert np.mean(pg == po) >= same_flag_min
    both = pg & po
    if both.sum() >= 50:
        same_cz = np.mean(out['cz'][both] == ref['cz'][both])
        stat, dof = _chi2_two_sample(out['cz'][pg], ref['cz'][po], bins=cz_bins)
        from scipy.stats import chi2
        print(f'cz: same on {same_cz:.3f} of the commonly converged; chi2 {stat:.1f} (dof {dof}, 99.9 % = {chi2.ppf(0.999, dof):.1f}); '
              f'mean {out["cz"][pg].mean():.3f} vs {ref["cz"][po].mean():.3f}; min {out["cz"][pg].min()} vs {ref["cz"][po].min()}')
        assert same_cz >= same_cz_min
        assert stat < chi2.ppf(0.999, dof)
        assert abs(out['cz'][pg].mean() - ref['cz'][po].mean()) < 0.1 * (1 if same_cz_min >= 0.85 else 3)
        assert abs(int(out['cz'][pg].min()) - int(ref['cz'][po].min())) <= (0 if same_cz_min >= 0.85 else 1)
    # distribution of the best regularised loss over ALL samples: two-sample Kolmogorov-Smirnov, and quantile by quantile -
    # either the values agree to q_tol, or (where the distribution has a gap between two basins, e.g. config 4's q0.9 between
    # the 0.137 and 0.150 clusters) the kernel's quantile sits at a statistically compatible position of the oracle's CDF
    ks_p = _ks_two_sample_p(out['regloss'][:, 1], ref['regloss'][:, 1])
    print(f'best regloss: two-sample KS p = {ks_p:.3f}')
    assert ks_p > 1e-3
    for q in (0.1, 0.25, 0.5, 0.75, 0.9):
        a, b = np.quantile(out['regloss'][:, 1], q), np.quantile(ref['regloss'][:, 1], q)
        pos = np.mean(ref['regloss'][:, 1] <= a)
        band = 3 * np.sqrt(2 * q * (1 - q) / S) + 1.0 / S
        print(f'best regloss q{q}: kernel {a:.6f} oracle {b:.6f} (oracle CDF at the kernel value {pos:.3f}, band {band:.3f})')
        assert abs(a - b) < q_tol or abs(pos - q) <= band
    diff = out['regloss'][:, 1] - ref['regloss'][:, 1]
    d = np.abs(diff)
    print(f'|d best regloss| over all samples: median {np.median(d):.2e} p90 {np.quantile(d, 0.9):.2e}; '
          f'median signed {np.median(diff):+.2e}')
    assert np.median(d) < pair_tol and abs(np.median(diff)) < pair_tol / 2
    return variant


@pytest.mark.parametrize('kernel', ['zyz3', 'split'])
def test_config2_unbiased_against_float32_oracle(kernel, tune):
    """Config 2 at full size (1024 samples x 2000 iterations) on the two-ended ZYZ-form kernel (cpf_zyz3.h, the default since
    round 5) and on the gate-by-gate two-ended kernel it replaced, against ONE float32 oracle run: quantiles within 2e-4 (a tenth
    of one CZ-equivalent of penalty, r * 2 = 1.1e-3), >= 85 % of the commonly converged samples with the same CZ count."""
    from cpflow_amd import workloads
    wl = workloads.CONFIGS['cfg2']
    tune.set('CPF_ZYZ3', '1' if kernel == 'zyz3' else '0')
    _unbiased_against_float32_oracle(wl, wl.samples_per_gpu, list(range(6, 14)), 0.85, 2e-4, 5e-5, kernel + '/aot')


@pytest.mark.parametrize('kernel', ['zyzl', 'layered'])
def test_config4_unbiased_against_float32_oracle(kernel, tune):
    """Config 4 at full size (4-qubit QFT, 2048 samples x 2000 iterations) on the relabelled one-wave kernels - the ZYZ form
    (cpf_zyzl.h, the default) and the gate-by-gate kernel it replaced - against ONE float32 oracle run.  4-qubit runs are more
    chaotic than 3-qubit ones (tests/test_gpu_synthesis.py, fixture toff4_star_t14: independent roundings of the oracle agree
    with each other on 84-86 % of the per-sample CZ counts)."""
    from cpflow_amd import workloads
    wl = workloads.CONFIGS['cfg4']
    tune.set('CPF_ZYZL', '1' if kernel == 'zyzl' else '0')
    _unbiased_against_float32_oracle(wl, wl.samples_per_gpu, list(range(14, 30)), 0.5, 1e-3, 1e-3, kernel + '/aot waves=1')


def test_stored_5q_trial_unbiased_against_float32_oracle(monkeypatch):
    """The 5-qubit benchmark problem of paper/results/benchmarks/res_4gt13_91 trial 99 (chain, 'xyz', 86 blocks, 617
    angles, its stored target, seed and r): 2 x 500 samples over 2000 iterations on the
    four-wave ZYZ-form kernel from the shipped cache against the float32 oracle (the one-wave form runs the same problem at its
    stored 1000 samples in tests/test_gpu_synthesis.py::test_T3_more_stored_trials_end_of_run).  The stored run itself used older penalty
    constants, so it pins loss and initial angles only (tests/test_gpu_synthesis.py); this is the full-length pin of the
    5-qubit kernel."""
    monkeypatch.setenv('CPF_JIT_FORBID_COMPILE', '1')
    g = np.load(os.path.join(GOLD, 'raw_4gt13_xyz_t99.npz'))

    class Trial:
        name, layer, num_cp_gates, target, rotation_gates = '4gt13 trial 99', g['layer'].tolist(), int(g['num_cp_gates']), g['target'], str(g['rotation_gates'])
        r, seed, num_gd_iterations = float(g['r']), int(g['seed']), 2000
    # 86 blocks x 2000 steps on 5 qubits: which of the ~8 % of the runs end below entry_loss is decided by rounding (kernel and
    # oracle agree on 89 % of the flags; independent draws at this rate would agree on 86 %), so the per-sample bars are low
    # and the distribution-level ones carry the test.  Two seeds (the stored one and a second: round 3 saw 88 against 67 passes
    # on the stored seed alone, 1.8 sigma) of 500 samples each; the pass counts are asserted on the pool.
    pooled = []
    for seed in (Trial.seed, Trial.seed + 1):
        Trial.seed = seed
        _unbiased_against_float32_oracle(Trial, 500, list(range(40, 80, 4)), 0.0, 3e-3, 3e-3, 'zyzl/jit(hipcc) waves=4', same_flag_min=0.8, pooled=pooled)      # 500 samples: four waves per sample
    kp, op, n = sum(p[0] for p in pooled), sum(p[1] for p in pooled), sum(p[2] for p in pooled)
    rate = (kp + op) / (2 * n)
    sigma = np.sqrt(2 * n * rate * (1 - rate))
    ks_p = _ks_two_sample_p(np.concatenate([p[3] for p in pooled]), np.concatenate([p[4] for p in pooled]))
    print(f'pooled over {len(pooled)} seeds: kernel {kp} oracle {op} passes of {n} (3 sigma = {3 * sigma:.1f}); KS p = {ks_p:.3f}')
    assert abs(kp - op) <= 3 * sigma and ks_p > 1e-3


@pytest.mark.parametrize('name,variant,same_flag_min', [('raw_toff4_star_t14', 'layered/jit(hipcc) waves=4', 0.9), ('raw_4gt10_xz_t30', 'zyzl/jit(hipcc) waves=1', 0.98)])
def test_stored_tutorial_and_paper_shapes_unbiased_against_float32_oracle(name, variant, same_flag_min, monkeypatch, tune):
    """The two remaining kernel variants the reference's stored trials exercise, at full length against the float32 oracle:
    the four-waves-per-sample layout (4-qubit star of tutorial/results/toff4_star trial 14: 29 blocks, 500 samples - fewer
    samples than SIMDs) and the one-wave kernel on an 'xz' template (5 qubits all-to-all, res_4gt10-v1_81 trial 30: 57 blocks,
    2000 samples, where almost no run converges).  Kernels from the shipped cache."""
    monkeypatch.setenv('CPF_JIT_FORBID_COMPILE', '1')
    if variant.endswith('waves=4'):
        tune.set('CPF_LW', '2')                  # 500 samples would take the one-wave ZYZ-form kernel since round 4: this pins the four-wave layout
    g = np.load(os.path.join(GOLD, name + '.npz'))

    class Trial:
        layer, num_cp_gates, target, rotation_gates = g['layer'].tolist(), int(g['num_cp_gates']), g['target'], str(g['rotation_gates'])
        r, seed, num_gd_iterations = float(g['r']), int(g['seed']), 2000
    Trial.name = name
    _unbiased_against_float32_oracle(Trial, int(g['num_samples']), list(range(18, 40, 2)), 0.0, 2e-3, 2e-3, variant, same_flag_min=same_flag_min)


@pytest.mark.parametrize('kernel', ['default', 'zyz', 'dense', 'gate-by-gate'])
def test_config5_unbiased_against_float32_oracle(kernel, tune):
    """Config 5 (6 qubits, ring, 60 blocks, SU(64) target): 128 samples x 2000 iterations on the kernel the plan picks,
    on the dense-block kernel and on the gate-by-gate kernel, each against the float32 oracle.  No run converges
    (60 blocks cannot reach a generic SU(64)), so the statistics are those of the final losses: a biased gradient of the
    per-block 4x4 environments would show as a shifted distribution of the best loss after 2000 steps."""
    from cpflow_amd import workloads
    wl = workloads.CONFIGS['cfg5']
    if kernel == 'dense':
        tune.set('CPF_DENSE', '1')
        tune.set('CPF_ZYZ', '0')
    elif kernel == 'zyz':
        tune.set('CPF_DENSE', '1')
        tune.set('CPF_ZYZ', '1')
    elif kernel == 'gate-by-gate':
        tune.set('CPF_DENSE', '0')
    variant = _unbiased_against_float32_oracle(wl, 128, list(range(60, 121)), 0.5, 4e-3, 4e-3)
    assert variant.split('/')[0] == {'default': variant.split('/')[0], 'zyz': 'zyz', 'dense': 'dense', 'gate-by-gate': 'layered'}[kernel]


def test_local_minimums_success_ratios():
    """paper/results/local_minimums (peppar.ipynb cells 14-17): for a target unitary and a number k of CZ gates, the
    fraction of 1000 random starts of Ansatz(3, 'cz', fill_layers(connected_layer(3), k)).learn(target) (adam, lr 0.1,
    5000 iterations, the PRNGKey(0) repeat chain) that end within 1e-4 of the best loss any start reached (for a 'random
    instance' of the template that best loss is 0, i.e. the criterion of cell 14; for a random unitary it is the global
    minimum of the template - the stored ratios are reproduced by exactly this rule from the stored losses).  The stored
    targets of two 'random instance' and two 'random unitary' files are re-run: each measured ratio must sit within 4
    binomial sigma (+ 0.01) of the stored one - the trajectories are chaotic, the ratio is not - and the best loss found
    must be the stored one wherever at least 1 % of the starts reach it."""
    from cpflow_amd import Ansatz, connected_layer, fill_layers
    g = np.load(os.path.join(GOLD, 'local_minimums_targets.npz'))
    worst = 0.0
    for i in range(len(g['success_ratio'])):
        k, want, R, vmin = int(g['num_cz_gates'][i]), float(g['success_ratio'][i]), int(g['num_repeats'][i]), float(g['min_loss'][i])
        anz = Ansatz(3, 'cz', fill_layers(connected_layer(3), k))
        res = anz.learn(g['u_target'][i].astype(np.complex128), num_iterations=5000, num_repeats=R, keep_history=False)
        best = np.array([r['loss'].min() for r in res])
        got = float(np.mean(best < best.min() + 1e-4))
        sigma = np.sqrt(max(want * (1 - want), 1e-3) / R * 2)
        print(f"{g['source_file'][i]} k={k}: stored {want:.3f} (min loss {vmin:.6f}) measured {got:.3f} (min loss {best.min():.6f}) "
              f"{(got - want) / sigma:+.1f} sigma")
        worst = max(worst, abs(got - want) / sigma)
        if want >= 0.01:
            assert abs(best.min() - vmin) < 1e-4, (str(g['source_file'][i]), k, vmin, float(best.min()))
            assert abs(got - want) <= 4 * sigma + 0.01, (str(g['source_file'][i]), k, want, got)
    print('worst deviation', worst, 'sigma')
