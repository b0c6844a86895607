"""Synthesis front end: `Synthesize`, `Results`, `Decomposition`.

Same public surface as the reference (cpflow/main.py:242-325, 429-502, 505-864) with the
compute moved to the GPU engine:

    Synthesize._generate_raw   -> TemplateEngine.initial_angles + TemplateEngine.optimize
                                  (one persistent HIP kernel per call; samples sharded over ranks)
    Synthesize._evaluate_raw   -> the kernel's epilogue already produced CZ counts / entry-loss flags
    verify_cp_result           -> ONE batched frozen-angle Adam run for all prospective samples
                                  (the reference runs them one at a time, cp_utils.py:205-247)

Circuits are `cpflow_amd.circuit.Circuit` objects (no qiskit needed).
"""
import math
import os
from dataclasses import dataclass

import numpy as np

from . import distributed
from .ansatz import Ansatz
from .circuit import Circuit
from .options import AdaptiveOptions, BasicOptions, RegularizationOptions, StaticOptions  # noqa: F401
from .topology import fill_layers, num_qubits_from_layer


# ---------------------------------------------------------------------------- losses
class HSTLoss:
    """u -> 1 - |sum(u * conj(target))|^2 / N^2   (cost_HST, cpflow/matrix_utils.py:35-42).
    A picklable object instead of the reference's lambda (cpflow/main.py:533)."""
    kind = 'hst'

    def __init__(self, target_unitary):
        self.target_unitary = np.asarray(target_unitary)

    def __call__(self, u):
        u = np.asarray(u)
        n = self.target_unitary.shape[0]
        return 1 - np.abs((u * np.conj(self.target_unitary)).sum()) ** 2 / n ** 2


class StatePrepLoss:
    """u -> 1 - |<psi| u |0>|^2 for a target state psi (tutorial cell 79; the `target_state` the
    reference's docstring promises, cpflow/main.py:513)."""
    kind = 'state'

    def __init__(self, target_state):
        self.target_state = np.asarray(target_state)

    @property
    def target_unitary(self):          # what the device receives: the state as column 0
        return self.target_state

    def __call__(self, u):
        return 1 - np.abs(np.conj(self.target_state) @ np.asarray(u)[:, 0]) ** 2


class RelativePhaseLoss:
    """u -> 1 - sum |conj(V) * u|^2 / N: zero when V^dagger u is diagonal (tutorial cell 87)."""
    kind = 'relative_phase'

    def __init__(self, target_unitary):
        self.target_unitary = np.asarray(target_unitary)

    def __call__(self, u):
        n = self.target_unitary.shape[0]
        return 1 - (np.abs(np.conj(self.target_unitary) * np.asarray(u)) ** 2).sum() / n


def theoretical_lower_bound(n):
    """Minimum number of CNOTs for a generic n-qubit unitary (cpflow/matrix_utils.py:11-14)."""
    return int((4 ** n - 3 * n - 1) / 4 + 1)


# ---------------------------------------------------------------------------- frozen-angle templates
class ConstrainedTemplate:
    """Template with some angles fixed: free angles -> unitary / circuit.

    Plays the role of `constrained_function(anz.unitary | anz.circuit, fixed, indices)`
    (cpflow/cp_utils.py:80-108) as a picklable object."""

    def __init__(self, ansatz, fixed_params, indices, kind):
        self.ansatz = ansatz
        self.fixed_params = np.asarray(fixed_params, dtype=np.float32)
        self.indices = [int(i) for i in indices]
        self.kind = kind

    def full_angles(self, free_params):
        total = len(free_params) + len(self.indices)
        full = np.zeros(total, dtype=np.float32)
        keep = np.ones(total, dtype=bool)
        keep[self.indices] = False
        full[keep] = np.asarray(free_params, dtype=np.float32)
        full[self.indices] = self.fixed_params
        return full

    def __call__(self, free_params):
        full = self.full_angles(free_params)
        return self.ansatz.unitary(full) if self.kind == 'unitary' else self.ansatz.circuit(full)


def project_cp_angles(angles, cp_mask, threshold):
    """Indices and values of the CP angles that round to identity (0) or CZ (pi)
    (project_cp_angle / convert_cp_to_cz, cpflow/cp_utils.py:70-77, 111-141)."""
    two_pi = np.float32(2 * np.pi)
    idx, val = [], []
    for i in np.flatnonzero(np.asarray(cp_mask)):
        a = np.float32(angles[i]) % two_pi
        if abs(a - np.float32(np.pi)) < threshold:
            idx.append(int(i)); val.append(np.float32(np.pi))
        elif abs(a) < threshold or abs(a - two_pi) < threshold:
            idx.append(int(i)); val.append(np.float32(0))
    return idx, np.array(val, dtype=np.float32)


# ---------------------------------------------------------------------------- Decomposition
class Decomposition:
    """A found decomposition.

    Attributes: unitary_loss_func, circuit, unitary, label, loss, type ('Approximate' | 'Rational' |
    'Clifford+T'), cz_count, cz_depth, t_count, t_depth, and the provenance fields _cp_data,
    _static_options, _adaptive_options, _decomposer.
    """

    def __init__(self, unitary_loss_func, circuit, label='', type='Approximate'):
        self.unitary_loss_func = unitary_loss_func
        self.circuit = circuit
        self.unitary = circuit.unitary()
        self.label = label
        self.loss = self.unitary_loss_func(self.unitary)
        self.type = type
        self.cz_count = circuit.count_ops().get('cz', 0)
        self.cz_depth = circuit.depth(['cz'])
        self.t_count = None
        self.t_depth = None
        self._cp_data = None
        self._static_options = None
        self._adaptive_options = None
        self._decomposer = None

    @classmethod
    def _from_cp_circuit(cls, unitary_loss_func, u_func, circ_func, angles, label):
        qc = circ_func(angles)
        qc = qc.cp_to_cz(cp_threshold=1e-6)
        qc = qc.to_zxz()
        d = cls(unitary_loss_func, qc, label=label)
        d._cp_data = [u_func, circ_func, angles]
        return d

    def refine(self, max_denominator=32, angle_threshold=0.01, cp_threshold=0.01, reduce_threshold=1e-5,
               recursion_degree=0, recursion_depth=5):
        """Signature of cpflow's Decomposition.refine (main.py:283-317).  Refinement (angle rationalisation, Solovay-Kitaev)
        is qiskit-bound CPU post-processing that stays in cpflow (exact_decompositions.py; SURVEY 2a: out of scope here)."""
        raise NotImplementedError('refine is CPU post-processing that stays in cpflow (cpflow/exact_decompositions.py); '
                                  'hand decomposition.circuit.data (gate list) to it')

    def __repr__(self):
        text = (f"< {self.label}| {self.type} | loss: {self.loss}  | CZ count: {self.cz_count} "
                f"| CZ depth: {self.cz_depth}  >")
        if self.type == 'Clifford+T':
            text = text[:-1] + f'| T count: {self.t_count} | T depth: {self.t_depth} >'
        return text


# ---------------------------------------------------------------------------- Results
@dataclass
class Results:
    """Results of static / adaptive runs: loss_function, layer, label, trials, decompositions, save_to.
    save() / load(path) use dill; best_hyperparameters() and plot_trials() summarise adaptive trials."""

    loss_function: callable
    layer: list
    label: str = ''
    trials: object = None
    decompositions: tuple = ()
    save_to: str = ''

    def __post_init__(self):
        if self.save_to == '':
            self.save_to = f'results/{self.label}'

    def save(self):
        import dill
        folder = os.path.dirname(self.save_to)
        if folder:
            os.makedirs(folder, exist_ok=True)
        with open(self.save_to, 'wb') as f:
            dill.dump(self, f)

    @staticmethod
    def load(path):
        import dill
        with open(path, 'rb') as f:
            return dill.load(f)

    def best_hyperparameters(self):
        """[num_cp_gates, r] pairs ordered by increasing score."""
        ranked = sorted(self.trials.results, key=lambda res: res['loss'])
        return [[res['num_cp_gates'], res['r']] for res in ranked]

    def plot_trials(self):
        import matplotlib.pyplot as plt
        res = self.trials.results
        num = np.array([r['num_cp_gates'] for r in res])
        rr = np.array([r['r'] for r in res])
        score = np.array([r['loss'] for r in res], dtype=float)
        ok = score < np.inf
        plt.scatter(num[ok], rr[ok], c=score[ok], cmap='jet', edgecolors='black')
        plt.colorbar()
        plt.scatter(num[~ok], rr[~ok], marker='x', color='red')
        n_best, r_best = self.best_hyperparameters()[0]
        plt.scatter([n_best], [r_best], marker='*', facecolors='gold', edgecolors='black', s=[250])
        plt.xlabel('Number of CP gates')
        plt.ylabel('r: regularization weight')
        plt.title('Score')


class RawResults(list):
    """List of per-sample dicts {'params', 'loss', 'reg', 'regloss'} as `mynimize_repeated` returns
    (cpflow/optimization.py:365-371), carrying the device-side evaluation alongside."""
    cz = None          # int32[S]  CZ count at argmin regloss
    passed = None      # bool[S]   loss[argmin] <= entry_loss
    kernel_ms = None


# ---------------------------------------------------------------------------- Synthesize
class Synthesize:
    """Automated synthesis of unitaries into CZ + 1q gates.

    Synthesize(layer, unitary_loss_func=None, target_unitary=None, label=None, cp_regularization_func=None)
      layer: qubit pairs that may interact, e.g. [[0, 1], [1, 2]].
      target_unitary: the loss is the Hilbert-Schmidt distance to it.
      unitary_loss_func: only loss objects the device implements (HSTLoss) are accepted.
      cp_regularization_func: None (default linear penalty) or a RegularizationOptions instance.
    Methods: static(options), adaptive(options).
    """

    def __init__(self, layer, unitary_loss_func=None, target_unitary=None, label=None, cp_regularization_func=None,
                 target_state=None):
        self.layer = layer
        self.num_qubits = num_qubits_from_layer(self.layer)
        self.target_unitary = target_unitary
        if unitary_loss_func is None and target_state is not None:
            unitary_loss_func = StatePrepLoss(target_state)
        if unitary_loss_func is not None:
            if not isinstance(unitary_loss_func, (HSTLoss, StatePrepLoss, RelativePhaseLoss)):
                raise NotImplementedError(
                    'cpflow_amd runs the loss inside a HIP kernel; arbitrary Python loss callables are outside '
                    'its scope. Pass target_unitary=... (Hilbert-Schmidt loss), target_state=..., or an HSTLoss / '
                    'StatePrepLoss / RelativePhaseLoss instance.')
            self.unitary_loss_func = unitary_loss_func
            self.target_unitary = unitary_loss_func.target_unitary
        else:
            assert self.target_unitary is not None, 'Neither unitary loss function nor target unitary is provided.'
            assert np.shape(self.target_unitary) == (2 ** self.num_qubits, 2 ** self.num_qubits), \
                'Number of qubits in target unitary and layer do not match.'
            self.unitary_loss_func = HSTLoss(self.target_unitary)
        self.label = label
        if cp_regularization_func is None:
            self.cp_regularization_func = RegularizationOptions()
        elif isinstance(cp_regularization_func, RegularizationOptions):
            self.cp_regularization_func = cp_regularization_func
        else:
            raise NotImplementedError('cp_regularization_func must be None or a RegularizationOptions instance '
                                      '(the penalty is evaluated inside the HIP kernel).')
        self._engines = {}

    def __getstate__(self):
        state = dict(self.__dict__)
        state['_engines'] = {}
        return state

    # ------------------------------------------------------------------ engines
    def _ansatz(self, options_or_k, rotation_gates=None):
        k = options_or_k if isinstance(options_or_k, int) else options_or_k.num_cp_gates
        rot = rotation_gates if rotation_gates is not None else options_or_k.rotation_gates
        return Ansatz(self.num_qubits, 'cp', fill_layers(self.layer, k), rot)

    def _engine(self, **kw):
        from .engine import TemplateEngine
        key = tuple(sorted(kw.items()))
        if key not in self._engines:
            if len(self._engines) >= 16:
                self._engines.clear()
            reg = self.cp_regularization_func
            self._engines[key] = TemplateEngine(
                self.layer, kw.pop('num_cp_gates'), self.target_unitary, kw.pop('rotation_gates'),
                penalty=reg.function, xmax=reg.xmax, ymax=reg.ymax, plato_0=reg.plato_0, plato_1=reg.plato_1,
                plato_2=reg.plato_2, num_qubits=self.num_qubits, loss=self.unitary_loss_func.kind, **kw)
        return self._engines[key]

    def _raw_engine(self, options):
        if options.method != 'adam':
            raise NotImplementedError(f"Method {options.method} not supported on the device path (only 'adam').")
        return self._engine(num_cp_gates=int(options.num_cp_gates), rotation_gates=options.rotation_gates,
                            r=float(options.r), use_reg=True, learning_rate=float(options.learning_rate),
                            num_gd_iterations=int(options.num_gd_iterations),
                            threshold_cp=float(options.threshold_cp), entry_loss=float(options.entry_loss))

    # ------------------------------------------------------------------ raw stage (the hot path)
    def _generate_raw(self, options, initial_angles_array=None, keep_history=False):
        """S independent Adam runs on the CP template; returns a RawResults list of S dicts."""
        import torch
        eng = self._raw_engine(options)
        S = int(options.num_samples)
        rank, size = distributed.world()
        first, count = distributed.shard_range(S, rank, size)
        if initial_angles_array is None:
            init = eng.initial_angles(options.random_seed, S, first, count, cp_dist=options.cp_distribution)
        else:
            full = np.asarray(initial_angles_array, dtype=np.float32)
            if full.ndim == 1:
                full = full[None]
            S = full.shape[0]
            first, count = distributed.shard_range(S, rank, size)
            init = full[first:first + count]
        out = eng.optimize(init, keep_history=keep_history) if count else None
        if out is None:
            P = eng.num_angles
            z = lambda *shape, dt=torch.float32: torch.zeros(shape, dtype=dt, device=eng.device)  # noqa: E731
            out = dict(params=z(0, 2, P), regloss=z(0, 2), loss=z(0, 2), reg=z(0, 2), cz=z(0, dt=torch.int32),
                       passed=z(0, dt=torch.bool))
        kernel_ms = eng.last_kernel_ms() if count else 0.0
        fields = {k: out[k] for k in ('params', 'regloss', 'loss', 'reg', 'cz', 'passed')}
        if keep_history:
            fields['history_params'], fields['history_regloss'] = out['history_params'], out['history_regloss']
            fields['history_reg'] = out['history_reg']
        fields = distributed.gather_samples(fields, S)
        host = {k: v.cpu().numpy() for k, v in fields.items()}
        raw = RawResults()
        for s in range(S):
            if keep_history:     # [T, P] / [T] histories, reg recomputed row by row (optimization.py:365-370)
                raw.append({'params': host['history_params'][s],
                            'loss': host['history_regloss'][s] - host['history_reg'][s],
                            'reg': host['history_reg'][s], 'regloss': host['history_regloss'][s]})
            else:
                raw.append({'params': host['params'][s], 'loss': host['loss'][s], 'reg': host['reg'][s],
                            'regloss': host['regloss'][s]})
        raw.cz = host['cz'].astype(np.int32)
        raw.passed = host['passed'].astype(bool)
        raw.kernel_ms = kernel_ms
        return raw

    def _evaluate_raw(self, raw_results, options, disable_tqdm=False):
        """[[cz, res], ...] of the samples whose loss is below entry_loss, stably sorted by CZ count
        (filter_cp_results, cpflow/cp_utils.py:167-202)."""
        if not isinstance(raw_results, RawResults) or raw_results.cz is None:
            raise TypeError('_evaluate_raw expects the RawResults returned by _generate_raw')
        selected = [[int(raw_results.cz[i]), res] for i, res in enumerate(raw_results) if raw_results.passed[i]]
        selected.sort(key=lambda x: x[0])
        return selected

    # ------------------------------------------------------------------ verification (batched)
    def _verify_batch(self, prospective, num_cp_gates, options):
        """verify_cp_result for many results at once: project CP angles, freeze them, and run ONE
        unregularised Adam launch (lr / iterations of the verification stage) over all of them.
        Returns a list of (success, num_cz_gates, circ_func, u_func, best_free_angles)."""
        if not prospective:
            return []
        anz = self._ansatz(int(num_cp_gates), options.rotation_gates)
        P = anz.num_angles
        init = np.zeros((len(prospective), P), dtype=np.float32)
        frozen = np.zeros((len(prospective), P), dtype=np.uint8)
        proj = []
        for k, (_, res) in enumerate(prospective):
            best_i = int(np.argmin(res['regloss']))
            angles = np.asarray(res['params'][best_i], dtype=np.float32)
            idx, val = project_cp_angles(angles, anz.cp_mask, options.threshold_cp)
            init[k] = angles
            init[k, idx] = val
            frozen[k, idx] = 1
            proj.append((idx, val))
        eng = self._engine(num_cp_gates=int(num_cp_gates), rotation_gates=options.rotation_gates, r=0.0, use_reg=False,
                           learning_rate=float(options.learning_rate_at_verification),
                           num_gd_iterations=int(options.num_gd_iterations_at_verification),
                           threshold_cp=float(options.threshold_cp), entry_loss=float(options.target_loss))
        # the batch is sharded over the ranks like the raw stage (contiguous blocks), one gather afterwards
        import torch
        rank, size = distributed.world()
        first, count = distributed.shard_range(len(prospective), rank, size)
        if count:
            out = eng.optimize(init[first:first + count], frozen=frozen[first:first + count])
            fields = {'regloss': out['regloss'], 'params': out['params']}
        else:
            fields = {'regloss': torch.zeros(0, 2, device=eng.device), 'params': torch.zeros(0, 2, P, device=eng.device)}
        fields = distributed.gather_samples(fields, len(prospective))
        losses = fields['regloss'].cpu().numpy()
        params = fields['params'].cpu().numpy()
        verified = []
        for k, (num_cz, _res) in enumerate(prospective):
            best_i = int(np.argmin(losses[k]))
            idx, val = proj[k]
            keep = np.ones(P, dtype=bool)
            keep[idx] = False
            free = params[k, best_i][keep]
            u_func = ConstrainedTemplate(anz, val, idx, 'unitary')
            circ_func = ConstrainedTemplate(anz, val, idx, 'circuit')
            verified.append((bool(losses[k, best_i] <= options.target_loss), num_cz, circ_func, u_func, free))
        return verified

    # ------------------------------------------------------------------ bookkeeping
    def _initialize_results(self, save_results, save_to):
        results = Results(self.unitary_loss_func, self.layer, label=self.label)
        if save_results:
            assert self.label or save_to, \
                'To save results on a disk either `label` or `save_to` must be provided. ' \
                'If you insist on not saving the results call the decomposition routine with `save_results=False` flag.'
            if save_to:
                results.save_to = save_to
            try:
                results = Results.load(results.save_to)
            except FileNotFoundError:
                pass
        return results

    def _make_decomposition(self, u_func, circ_func, best_angs, static_options=None, adaptive_options=None,
                            circuit=None):
        d = circuit
        if d is None:
            d = Decomposition._from_cp_circuit(self.unitary_loss_func, u_func, circ_func, best_angs, self.label)
        d._static_options = static_options
        d._adaptive_options = adaptive_options
        d._decomposer = self
        return d

    # ------------------------------------------------------------------ static
    def static(self, options, save_results=True, save_to=''):
        """Synthesis with a fixed CP template and regularisation weight -> Results."""
        results = self._initialize_results(save_results, save_to)
        print('\nStarting decomposition routine with the following options:')
        print('\n', options)

        print('\nComputing raw results...')
        raw_results = self._generate_raw(options)

        print('\nSelecting prospective results...')
        prospective = [res for res in self._evaluate_raw(raw_results, options)
                       if res[0] <= options.accepted_num_cz_gates]
        successful = []
        if prospective:
            print(f'\nFound {len(prospective)}. Verifying...')
            for success, _num_cz, circ, u, best_angs in self._verify_batch(prospective, options.num_cp_gates, options):
                if success:
                    successful.append(self._make_decomposition(u, circ, best_angs, static_options=options))
            if successful:
                print(f'\n{len(successful)} successful. cz counts are:')
                print(sorted(d.cz_count for d in successful))
                results.decompositions = list(results.decompositions) + successful
                if save_results and distributed.world()[0] == 0:
                    results.save()
            else:
                print('\nAll prospective results failed.')
        else:
            print('\nNo results passed.')
        return results

    # ------------------------------------------------------------------ adaptive
    def adaptive(self, options, save_results=True, save_to=''):
        """Synthesis with template length and regularisation weight chosen by a hyper-parameter
        search (cpflow_amd.hyper; the reference drives hyperopt's TPE, cpflow/main.py:695-864)."""
        from .hyper import run_adaptive
        return run_adaptive(self, options, save_results=save_results, save_to=save_to)
