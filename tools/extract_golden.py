#!/usr/bin/env python3
"""Mine golden vectors from the result files the reference ships.

Run once in the build container (where /root/reference exists):

    python tools/extract_golden.py

Writes small .npz fixtures into tests/golden/.  Only *data* is extracted
(float arrays, seeds, hyper-parameters, targets); no reference source text is
copied and nothing from the pickles is executed (tools/inert_pickle.py).

Fixture families:
  raw_*   per-sample {initial, best} angles with loss / reg / regloss as the
          reference computed them, together with seed, K, r, S, layer, target.
  dec_*   post-verification angle vectors of stored Decomposition objects
          (fixed CP angles + free angles) whose unitary must hit the target.
  adaptive_*  the adaptive seed chain and (num_cp_gates, r, score, cz_counts).
  known_answers_custom_losses  printed outputs of the executed tutorial for the state-preparation and
          relative-phase losses (K3 / K4): seed-chain position, (K, r), score, CZ counts.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

from inert_pickle import Stub, load, to_array  # noqa: E402
from oracle import prng  # noqa: E402

REF = '/root/reference'
OUT = os.path.join(ROOT, 'tests', 'golden')


def _options(results_state):
    """First available option dict (adaptive preferred) from stored decompositions."""
    adaptive = static = None
    for d in results_state['decompositions']:
        st = d._state
        if adaptive is None and st.get('_adaptive_options') is not None:
            adaptive = dict(st['_adaptive_options']._state)
        if static is None and st.get('_static_options') is not None:
            static = dict(st['_static_options']._state)
    return adaptive, static


def _target(results_state):
    d = results_state['decompositions'][0]._state
    return np.asarray(to_array(d['_decomposer']._state['target_unitary']), dtype=np.complex128)


def _num_qubits(layer):
    return max(max(p) for p in layer) + 1


def extract_raw(rel_path, trial_ids, name, pin_reg=True, num_samples=None):
    R = load(os.path.join(REF, rel_path))._state
    layer = [list(map(int, p)) for p in R['layer']]
    n = _num_qubits(layer)
    adaptive, static = _options(R)
    opts = adaptive or static
    rot = opts['rotation_gates']
    # toff3_chain keeps no adaptive options object; its adaptive run used 100 samples
    S = int(num_samples if num_samples is not None else opts['num_samples'])
    target = _target(R)
    trials = R['trials']._state['_trials']
    for tid in trial_ids:
        res = trials[tid]['result']
        K = int(res['num_cp_gates'])
        seed = int(res['random_seed'])
        r = float(res['r'])
        raw = res['prospective_decompositions']
        nb = 2 * len(rot) + 1
        P = 3 * n + nb * K
        init = prng.initial_angles(seed, S, P)
        cz, params, loss, reg, regloss, index = [], [], [], [], [], []
        for c, d in raw:
            p = to_array(d['params']).astype(np.float32)
            assert p.shape == (2, P), (p.shape, P)
            hit = np.where((init.view(np.uint32) == p[0].view(np.uint32)).all(axis=1))[0]
            assert len(hit) == 1, 'initial angles do not match the PRNG restatement'
            index.append(int(hit[0]))
            cz.append(int(c))
            params.append(p)
            loss.append(to_array(d['loss']).astype(np.float32))
            reg.append(to_array(d['reg']).astype(np.float32))
            regloss.append(to_array(d['regloss']).astype(np.float32))
        out = os.path.join(OUT, f'raw_{name}_t{tid}.npz')
        np.savez_compressed(
            out,
            source=np.array(f'{rel_path} trials[{tid}]'),
            layer=np.array(layer, dtype=np.int32), num_qubits=n, rotation_gates=np.array(rot),
            seed=np.uint32(seed), num_cp_gates=K, r=np.float64(r), num_samples=S,
            entry_loss=float(opts['entry_loss']), threshold_cp=float(opts['threshold_cp']),
            learning_rate=float(opts['learning_rate']), num_gd_iterations=int(opts['num_gd_iterations']),
            target=target, score=np.float64(res['loss']),
            cz_counts=np.array([int(c) for c in res['cz_counts']], dtype=np.int32),
            sample_index=np.array(index, dtype=np.int32), cz=np.array(cz, dtype=np.int32),
            params=np.array(params, dtype=np.float32).reshape(len(raw), 2, P),
            loss=np.array(loss, dtype=np.float32).reshape(len(raw), 2),
            reg=np.array(reg, dtype=np.float32).reshape(len(raw), 2),
            regloss=np.array(regloss, dtype=np.float32).reshape(len(raw), 2),
            pin_reg=bool(pin_reg))
        print(f'{out}: n={n} K={K} P={P} S={S} raw={len(raw)} seed={seed}')


def extract_decompositions(rel_path, name, limit=None):
    R = load(os.path.join(REF, rel_path))._state
    layer = [list(map(int, p)) for p in R['layer']]
    n = _num_qubits(layer)
    target = _target(R)
    recs = []
    for d in R['decompositions'][:limit]:
        st = d._state
        so = st['_static_options']._state
        u_func, _circ_func, angles = st['_cp_data']
        cells = u_func._args[4]
        fixed = to_array(cells[1]._args[0]).astype(np.float32).reshape(-1)
        indices = np.array([int(i) for i in cells[2]._args[0]], dtype=np.int32)
        free = to_array(angles).astype(np.float32)
        rot = so['rotation_gates']
        K = int(so['num_cp_gates'])
        P = 3 * n + (2 * len(rot) + 1) * K
        assert len(free) + len(indices) == P
        full = np.zeros(P, dtype=np.float32)
        mask = np.ones(P, dtype=bool)
        mask[indices] = False
        full[mask] = free
        full[indices] = fixed
        recs.append(dict(K=K, rot=rot, full=full, indices=indices, fixed=fixed,
                         cz_count=int(st['cz_count']), cz_depth=int(st['cz_depth']),
                         loss=float(to_array(st['loss'])), type=str(st['type']),
                         r=float(so['r']), seed=int(so['random_seed'])))
    maxP = max(len(r['full']) for r in recs)
    maxI = max(len(r['indices']) for r in recs)
    full = np.zeros((len(recs), maxP), dtype=np.float32)
    frozen_idx = -np.ones((len(recs), maxI), dtype=np.int32)
    frozen_val = np.zeros((len(recs), maxI), dtype=np.float32)
    for i, r in enumerate(recs):
        full[i, :len(r['full'])] = r['full']
        frozen_idx[i, :len(r['indices'])] = r['indices']
        frozen_val[i, :len(r['indices'])] = r['fixed']
    out = os.path.join(OUT, f'dec_{name}.npz')
    np.savez_compressed(
        out, source=np.array(rel_path), layer=np.array(layer, dtype=np.int32), num_qubits=n,
        target=target,
        rotation_gates=np.array([r['rot'] for r in recs]),
        num_cp_gates=np.array([r['K'] for r in recs], dtype=np.int32),
        angles=full, frozen_idx=frozen_idx, frozen_val=frozen_val,
        cz_count=np.array([r['cz_count'] for r in recs], dtype=np.int32),
        cz_depth=np.array([r['cz_depth'] for r in recs], dtype=np.int32),
        stored_loss=np.array([r['loss'] for r in recs], dtype=np.float64),
        type=np.array([r['type'] for r in recs]),
        r=np.array([r['r'] for r in recs], dtype=np.float64),
        seed=np.array([r['seed'] for r in recs], dtype=np.uint32))
    print(f'{out}: {len(recs)} decompositions, n={n}')


def extract_adaptive(rel_path, name):
    R = load(os.path.join(REF, rel_path))._state
    trials = R['trials']._state['_trials']
    adaptive, _ = _options(R)
    res = [t['result'] for t in trials]
    maxc = max(len(r['cz_counts']) for r in res)
    czc = -np.ones((len(res), maxc), dtype=np.int32)
    for i, r in enumerate(res):
        czc[i, :len(r['cz_counts'])] = [int(c) for c in r['cz_counts']]
    out = os.path.join(OUT, f'adaptive_{name}.npz')
    np.savez_compressed(
        out, source=np.array(rel_path),
        initial_seed=np.uint32(adaptive['random_seed'] if adaptive else 0),
        num_samples=int(adaptive['num_samples']) if adaptive else -1,
        seeds=np.array([int(r['random_seed']) for r in res], dtype=np.uint32),
        num_cp_gates=np.array([int(r['num_cp_gates']) for r in res], dtype=np.int32),
        r=np.array([float(r['r']) for r in res], dtype=np.float64),
        score=np.array([float(r['loss']) for r in res], dtype=np.float64),
        cz_counts=czc)
    print(f'{out}: {len(res)} trials')


def extract_local_minimums():
    """paper/results/local_minimums/*: plain pickles of numpy-only dict lists."""
    import pickle

    class NumpyOnly(pickle.Unpickler):
        def find_class(self, module, name):
            if module.startswith('numpy'):
                module = module.replace('numpy.core', 'numpy._core')
                return getattr(__import__(module, fromlist=[name]), name)
            if module == 'builtins' and name in ('complex', 'slice', 'range', 'set', 'frozenset'):
                return getattr(__import__('builtins'), name)
            raise pickle.UnpicklingError(f'blocked {module}.{name}')

    d = os.path.join(REF, 'paper/results/local_minimums')
    rows = []
    for fn in sorted(os.listdir(d)):
        try:
            with open(os.path.join(d, fn), 'rb') as f:
                lst = NumpyOnly(f).load()
        except Exception as e:  # a non-numpy pickle: skip, report
            print('skip', fn, e)
            continue
        for rec in lst:
            rows.append((fn, str(rec.get('target_type')), int(rec.get('random_seed', -1)),
                         int(rec['num_cz_gates']), int(rec['num_qubits']), float(rec['success_ratio'])))
    # the targets themselves for a subset (3 qubits, instances rs0 / rs1, 3..8 CZ gates), so that the success ratios can
    # be re-measured on the same matrices (tests/test_gpu_baseline_configs.py::test_local_minimums_success_ratios)
    sub = []
    for fn in sorted(os.listdir(d)):
        if not (fn.startswith('3q_') and fn.endswith(('rs0', 'rs1'))):
            continue
        with open(os.path.join(d, fn), 'rb') as f:
            lst = NumpyOnly(f).load()
        for rec in lst:
            if 3 <= int(rec['num_cz_gates']) <= 8:
                sub.append((fn, str(rec['target_type']), int(rec['num_cz_gates']), np.asarray(rec['u_target'], dtype=np.complex64),
                            float(rec['success_ratio']), len(rec['losses']), float(np.min(rec['losses']))))
    np.savez_compressed(
        os.path.join(OUT, 'local_minimums_targets.npz'), source_file=np.array([r[0] for r in sub]),
        target_type=np.array([r[1] for r in sub]), num_cz_gates=np.array([r[2] for r in sub], dtype=np.int32),
        u_target=np.stack([r[3] for r in sub]), success_ratio=np.array([r[4] for r in sub]),
        num_repeats=np.array([r[5] for r in sub], dtype=np.int32), min_loss=np.array([r[6] for r in sub], dtype=np.float32))
    print(f'local_minimums_targets.npz: {len(sub)} records')
    out = os.path.join(OUT, 'local_minimums_summary.npz')
    np.savez_compressed(
        out, source_file=np.array([r[0] for r in rows]), target_type=np.array([r[1] for r in rows]),
        random_seed=np.array([r[2] for r in rows], dtype=np.int64),
        num_cz_gates=np.array([r[3] for r in rows], dtype=np.int32),
        num_qubits=np.array([r[4] for r in rows], dtype=np.int32),
        success_ratio=np.array([r[5] for r in rows], dtype=np.float64))
    print(f'{out}: {len(rows)} records')


def extract_known_answers():
    """Printed outputs of the executed tutorial notebook for the two custom losses: the first
    adaptive trial that produced prospective results - its position in the seed chain, (num_cp_gates, r), the printed score
    and the printed CZ counts of the prospective results.  Cell 81: GHZ state preparation, 4 qubits on a line; cell 89:
    relative-phase Toffoli, 3 qubits all-to-all.  Only numbers are kept."""
    import json
    import re
    nb = json.load(open(os.path.join(REF, 'tutorial', 'CPFlow_tutorial.ipynb')))
    rows = {}
    for cell, name in ((81, 'K3_ghz_state_preparation'), (89, 'K4_relative_phase_toffoli')):
        c = nb['cells'][cell]
        src = ''.join(c['source'])
        layer = json.loads(re.search(r'layer = (\[\[.*?\]\])', src).group(1))
        text = ''.join(''.join(o.get('text', '')) for o in c.get('outputs', []))
        opts = re.search(r'AdaptiveOptions\((.*?)\)\n', text).group(1)
        num_samples = int(re.search(r'num_samples=(\d+)', opts).group(1))
        seed0 = int(re.search(r'random_seed=(\d+)', opts).group(1))
        its = re.findall(r'iteration (\d+)/\d+\s+num_cp_gates: (\d+), r: ([0-9.e-]+)\s+score: (\S+), cz counts of prospective results: \[(.*?)\]', text)
        it, K, r, score, counts = next(x for x in its if x[4].strip())
        seed = seed0
        for _ in range(int(it) + 1):                     # main.py:798-799: the seed of trial i is the (i+1)-th link of the chain
            seed = int(prng.next_adaptive_seed(seed)) if hasattr(prng, 'next_adaptive_seed') else None
        rows[name] = dict(layer=np.array(layer, dtype=np.int32), trial=int(it), seed=np.uint32(seed), num_cp_gates=int(K), r=float(r),
                          num_samples=num_samples, score=float(score), cz_counts=np.array([int(x) for x in counts.split(',')], dtype=np.int32))
    out = os.path.join(OUT, 'known_answers_custom_losses.npz')
    np.savez_compressed(out, **{f'{k}.{f}': v for k, d in rows.items() for f, v in d.items()})
    for k, d in rows.items():
        print(f"{out}: {k} trial {d['trial']} seed {d['seed']} K={d['num_cp_gates']} r={d['r']} -> {len(d['cz_counts'])} prospective, score {d['score']}")


def main():
    os.makedirs(OUT, exist_ok=True)
    extract_raw('tutorial/results/toff3_chain', [19, 49], 'toff3_chain', num_samples=100)  # G1, G1b
    extract_raw('tutorial/results/toff4_star', [14, 49], 'toff4_star')              # G2
    extract_raw('paper/results/toff3_conn_xyz', [99], 'toff3_conn', pin_reg=False)  # G3 (older penalty constants)
    extract_raw('paper/results/benchmarks/res_4gt10-v1_81', [30], '4gt10_xz', pin_reg=False)   # G4
    extract_raw('paper/results/benchmarks/res_4gt13_91', [99], '4gt13_xyz', pin_reg=False)     # G5
    extract_decompositions('paper/results/refine_test', 'refine_test')              # G6
    extract_decompositions('tutorial/results/toff3_chain', 'toff3_chain')
    extract_decompositions('tutorial/results/toff4_star', 'toff4_star')
    extract_adaptive('tutorial/results/toff3_chain', 'toff3_chain')
    extract_adaptive('tutorial/results/toff4_star', 'toff4_star')
    extract_known_answers()
    try:
        extract_local_minimums()
    except Exception as e:  # optional family
        print('local_minimums skipped:', e)


if __name__ == '__main__':
    main()
