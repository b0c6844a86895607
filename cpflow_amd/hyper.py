"""Adaptive synthesis: search over (num_cp_gates, r) (counterpart of Synthesize.adaptive,
cpflow/main.py:695-864).

The reference drives hyperopt's `fmin(..., algo=tpe.suggest)` one evaluation at a time.  hyperopt is a
third-party dependency that is not available here, so the suggestion step is restated over the same search space
(num_cp_gates ~ quniform(min, max, 1), r ~ lognormal(log r_mean, r_variance)) with hyperopt's defaults (20 random
start-up evaluations, gamma = 0.25, 24 candidates):

* the 20 START-UP evaluations draw from the prior exactly as hyperopt 0.2.7 does under
  `fmin(..., rstate=np.random.default_rng(seed_i))` (main.py:801-810): fmin hands `rstate.integers(2**31 - 1)` to
  `tpe.suggest`, which seeds a fresh `default_rng` with it for `rand.suggest`; the space is evaluated r first
  (`exp(rng.normal(float32 log r_mean, r_variance))`), then num_cp_gates (`round(rng.uniform(min, max))`).  Pinned:
  the first 20 (num_cp_gates, r) of both stored adaptive runs (tests/golden/adaptive_*.npz) and the trials the tutorial
  printed (cells 81, 89) are reproduced exactly (tests/test_host_logic.py);
* after that an own tree-structured Parzen estimator takes over; its suggestions differ from hyperopt's
  (parity unpinned beyond the start-up phase).

Everything around it follows the reference: the objective and its score, the seed chain (verified against stored
seeds), saving after every evaluation, resuming from stored trials, the scoreboard and the verify-on-improvement logic.
"""
import math

import numpy as np

from . import distributed, prng

STATUS_OK = 'ok'
N_STARTUP, GAMMA, N_CANDIDATES = 20, 0.25, 24


class Trials:
    """Minimal stand-in for hyperopt.Trials: `.results` (one dict per evaluation) and `.trials`."""

    def __init__(self):
        self.trials = []

    @property
    def results(self):
        return [t['result'] for t in self.trials]

    def append(self, vals, result):
        self.trials.append({'tid': len(self.trials), 'misc': {'vals': {k: [v] for k, v in vals.items()}},
                            'result': result, 'state': 2})

    def __len__(self):
        return len(self.trials)


def _parzen_logpdf(x, centers, sigma_prior, mu_prior, lo=None, hi=None):
    """Equal-weight mixture of Gaussians at the observations plus the prior component."""
    mus = np.concatenate([[mu_prior], centers])
    order = np.argsort(mus)
    mus = mus[order]
    if len(mus) > 1:
        gaps = np.diff(mus)
        sig = np.maximum(np.concatenate([[gaps[0]], gaps]), np.concatenate([gaps, [gaps[-1]]]))
    else:
        sig = np.array([sigma_prior])
    sig = np.clip(sig, sigma_prior / max(10.0, 1.0 + len(mus)), sigma_prior)
    sig[np.flatnonzero(order == 0)] = sigma_prior
    z = (np.asarray(x, dtype=float)[:, None] - mus[None, :]) / sig[None, :]
    comp = -0.5 * z * z - np.log(sig[None, :]) - 0.5 * math.log(2 * math.pi)
    return np.logaddexp.reduce(comp, axis=1) - math.log(len(mus)), mus, sig


def _sample_parzen(rng, mus, sig, n, lo=None, hi=None):
    out = []
    while len(out) < n:
        i = rng.integers(len(mus))
        v = rng.normal(mus[i], sig[i])
        if (lo is None or v >= lo) and (hi is None or v <= hi):
            out.append(v)
    return np.array(out)


def suggest(random_seed, history, k_min, k_max, r_mean, r_variance):
    """One (num_cp_gates, r) suggestion for the trial whose link of the seed chain is random_seed.
    history: list of (num_cp_gates, r, score)."""
    # fmin(rstate=default_rng(random_seed)) -> algo(..., seed=rstate.integers(2**31 - 1)) -> default_rng(seed)
    rng = np.random.default_rng(int(np.random.default_rng(int(random_seed)).integers(2 ** 31 - 1)))

    def prior():
        # hp.lognormal('r', jnp.log(r_mean), r_variance): the mean arrives as a float32; r is drawn before num_cp_gates
        r = float(np.exp(rng.normal(float(np.log(np.float32(r_mean))), r_variance, size=(1,))[0]))
        k = int(np.round(rng.uniform(k_min, k_max, size=(1,))[0]))           # scope.int(hp.quniform(..., 1))
        return k, r

    finite = [(k, r, s) for k, r, s in history]
    if len(finite) < N_STARTUP:
        return prior()
    scores = np.array([s for _, _, s in finite], dtype=float)
    n_good = max(1, int(math.ceil(GAMMA * math.sqrt(len(finite)))))
    order = np.argsort(scores, kind='stable')
    good, bad = order[:n_good], order[n_good:]
    ks = np.array([k for k, _, _ in finite], dtype=float)
    lrs = np.log(np.array([r for _, r, _ in finite], dtype=float))
    k_mu, k_sig = 0.5 * (k_min + k_max), max(1.0, float(k_max - k_min))
    _, gm, gs = _parzen_logpdf([k_mu], ks[good], k_sig, k_mu)
    kc = np.clip(np.round(_sample_parzen(rng, gm, gs, N_CANDIDATES, k_min - 0.5, k_max + 0.5)), k_min, k_max)
    lk = _parzen_logpdf(kc, ks[good], k_sig, k_mu)[0] - _parzen_logpdf(kc, ks[bad], k_sig, k_mu)[0]
    r_mu, r_sig = math.log(r_mean), float(r_variance)
    _, gm, gs = _parzen_logpdf([r_mu], lrs[good], r_sig, r_mu)
    rc = _sample_parzen(rng, gm, gs, N_CANDIDATES)
    lr_ = _parzen_logpdf(rc, lrs[good], r_sig, r_mu)[0] - _parzen_logpdf(rc, lrs[bad], r_sig, r_mu)[0]
    return int(kc[int(np.argmax(lk))]), float(math.exp(rc[int(np.argmax(lr_))]))


def score_from_cz_counts(cz_counts, num_samples):
    """-log2( sum_i 2^-cz_i / num_samples ) in float32, inf when nothing passed (main.py:735-737)."""
    s = (np.float32(2.0) ** (-np.asarray(cz_counts, dtype=np.float32))).sum(dtype=np.float32) / np.float32(num_samples)
    with np.errstate(divide='ignore'):
        return float(-np.log2(np.float32(s)))


def run_adaptive(syn, options, save_results=True, save_to=''):
    from .synthesis import theoretical_lower_bound

    def evaluate(random_seed, num_cp_gates, r):
        print(f'\nnum_cp_gates: {num_cp_gates}, r: {r}')
        static_options = options.get_static(num_cp_gates, r)
        static_options.random_seed = random_seed
        raw = syn._generate_raw(static_options)
        evaluated = syn._evaluate_raw(raw, static_options, disable_tqdm=True)
        cz_counts = [res[0] for res in evaluated]
        score = score_from_cz_counts(cz_counts, options.num_samples)
        print(f'score: {score}, cz counts of prospective results: {cz_counts}')
        return {'loss': score, 'status': STATUS_OK, 'random_seed': random_seed, 'cz_counts': cz_counts,
                'num_cp_gates': num_cp_gates, 'r': r, 'layer': syn.layer,
                'prospective_decompositions': evaluated}

    print('\nStarting decomposition routine with the following options:')
    print('\n', options)
    results = syn._initialize_results(save_results, save_to)
    is_writer = distributed.world()[0] == 0

    if results.trials is not None:
        print('\nFound existing trials, resuming from here.')
        trials = results.trials
        random_seed = trials.results[-1]['random_seed']
        num_existing = len(trials.results)
    else:
        trials, random_seed, num_existing = Trials(), options.random_seed, 0

    if results.decompositions:
        scoreboard = sorted(set(d.cz_count for d in results.decompositions))
    else:
        scoreboard = [theoretical_lower_bound(syn.num_qubits)]
    if num_existing >= options.max_evals:
        print('Maximum number of evaluations reached.')

    for i in range(num_existing, options.max_evals):
        print('\n' + '-' * 42)
        print(f'iteration {i}/{options.max_evals}')
        random_seed = prng.next_adaptive_seed(random_seed)
        history = [(t['num_cp_gates'], t['r'], t['loss']) for t in trials.results]
        k, r = suggest(random_seed, history, options.min_num_cp_gates, options.max_num_cp_gates, options.r_mean,
                       options.r_variance)
        last = evaluate(random_seed, k, r)
        trials.append({'num_cp_gates': k, 'r': r}, last)
        results.trials = trials
        successful = last['prospective_decompositions']
        if not options.keep_logs:
            last.pop('prospective_decompositions')
        if save_results and is_writer:
            results.save()

        current_best = scoreboard[0]
        to_verify = [[cz, res] for cz, res in successful if cz < current_best]
        if to_verify:
            print(f'\nFound {len(to_verify)} decompositions potentially improving the current best count '
                  f'{current_best}, verifying...')
        else:
            print(f'\nFound no decompositions potentially improving the current best count {current_best}.')
        verified = syn._verify_batch(to_verify, k, options.get_static(k, r))
        for success, num_cz, circ, u, best_angs in verified:
            if success:
                print(f'\nFound a new decomposition with {num_cz} gates.')
                scoreboard.insert(0, num_cz)
                d = syn._make_decomposition(u, circ, best_angs, adaptive_options=options,
                                            static_options=options.get_static(k, r))
                results.decompositions = list(results.decompositions) + [d]
                if save_results and is_writer:
                    results.save()
                break
        else:
            if to_verify:
                print('\nNone of prospective decompositions passed.')
        if options.stop_if_target_reached and scoreboard[0] <= options.target_num_cz_gates:
            print('\nTarget number of gates reached.')
            break
    return results
