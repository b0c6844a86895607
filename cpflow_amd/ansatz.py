"""Template circuits (counterpart of `Ansatz`, cpflow/main.py:149-239).

`Ansatz.unitary` and `Ansatz.learn` run on the GPU through `TemplateEngine`;
`Ansatz.circuit` emits the light-weight `Circuit` IR instead of a qiskit circuit.
"""
import numpy as np

from .circuit import Circuit


def block_num_angles(entangling_gate_name, rotation_gates):
    """2 angles per rotation letter (+1 for the CP angle) — EntanglingBlock.get_num_angles, main.py:32-34."""
    return 2 * len(rotation_gates) + (1 if entangling_gate_name == 'cp' else 0)


class Ansatz:
    """Parametrised template: surface Rz Rx Rz on every qubit, then entangling blocks.

    Attributes: num_qubits, entangling_gate_name ('cp' | 'cz' | 'cx'), rotation_gates,
    placements ({'layers': [layer, num_layers], 'free': [...]}), all_placements, num_blocks,
    num_angles, cp_mask (for 'cp').
    Methods: unitary(angles), circuit(angles), learn(u_target, ...).
    """

    def __init__(self, num_qubits, entangling_gate_name, placements, rotation_gates='xyz'):
        if entangling_gate_name not in ('cp', 'cz', 'cx'):
            raise TypeError(f"Gate '{entangling_gate_name}' not implemented.")
        self.num_qubits = num_qubits
        self.entangling_gate_name = entangling_gate_name
        self.rotation_gates = rotation_gates
        placements.setdefault('layers', [[], 0])
        placements.setdefault('free', [])
        self.placements = placements
        self.layer, self.num_layers = placements['layers']
        self.free_placements = placements['free']
        self.all_placements = [list(p) for p in self.layer] * self.num_layers + [list(p) for p in self.free_placements]
        self.num_blocks = len(self.all_placements)
        self.num_block_angles = block_num_angles(entangling_gate_name, rotation_gates)
        self.num_angles = 3 * num_qubits + self.num_block_angles * self.num_blocks
        if entangling_gate_name == 'cp':
            mask = np.zeros(self.num_angles, dtype=np.int32)
            mask[3 * num_qubits + self.num_block_angles - 1::self.num_block_angles] = 1
            self.cp_mask = mask
        self._engines = {}

    def __getstate__(self):
        state = dict(self.__dict__)
        state['_engines'] = {}
        return state

    # ------------------------------------------------------------------ device engine
    def engine(self, target=None, **kw):
        """TemplateEngine for this template (cached per option set)."""
        from .engine import TemplateEngine
        N = 2 ** self.num_qubits
        tgt = np.eye(N) if target is None else np.asarray(target)
        key = (tgt.astype(np.complex128).tobytes(), tuple(sorted(kw.items())))
        if key not in self._engines:
            if len(self._engines) > 8:
                self._engines.clear()
            self._engines[key] = TemplateEngine(
                self.layer or [[0, 1]], self.num_blocks, tgt, self.rotation_gates,
                entangler=self.entangling_gate_name, num_qubits=self.num_qubits,
                placements=self.all_placements, **kw)
        return self._engines[key]

    def unitary(self, angles):
        """2^n x 2^n complex64 unitary of the template at `angles` (numpy; computed on the GPU)."""
        angles = np.asarray(angles, dtype=np.float32)
        u = self.engine(use_reg=False).unitary(angles)
        return u.cpu().numpy()

    # ------------------------------------------------------------------ circuit
    def circuit(self, angles=None):
        if angles is None:
            angles = [f'a_{i}' for i in range(self.num_angles)]
        else:
            angles = [float(a) for a in np.asarray(angles).reshape(-1)]
            assert len(angles) == self.num_angles
        n, nb = self.num_qubits, self.num_block_angles
        qc = Circuit(n)
        for q in range(n):
            qc.rz(angles[3 * q], q).rx(angles[3 * q + 1], q).rz(angles[3 * q + 2], q)
        for b, (up, dn) in enumerate(self.all_placements):
            a = angles[3 * n + nb * b:3 * n + nb * (b + 1)]
            if self.entangling_gate_name == 'cp':
                qc.cp(a[-1], up, dn)
            else:
                qc.append(self.entangling_gate_name, (up, dn))
            for i, letter in enumerate(self.rotation_gates):
                qc.append('r' + letter, (up,), (a[2 * i],))
                qc.append('r' + letter, (dn,), (a[2 * i + 1],))
        return qc

    # ------------------------------------------------------------------ learning
    def learn(self, u_target, method='adam', learning_rate=0.1, target_loss=1e-7, keep_history=True,
              num_iterations=5000, initial_angles=None, num_repeats=1, **kwargs):
        """Minimise the Hilbert-Schmidt distance to `u_target` (unitary_learn, optimization.py:385-419).

        Returns one result dict (or a list for a batch) with 'params', 'loss', 'reg', 'regloss':
        full histories [T, P] / [T] when keep_history, else {initial, best} pairs.
        Without `initial_angles`, repeat i starts from the i-th angle vector of the reference's
        PRNGKey(0) split chain (optimization.py:306-311).
        """
        if method != 'adam':
            raise NotImplementedError(f"Method {method} not supported on the device path")
        if kwargs:
            raise TypeError(f'unexpected arguments {sorted(kwargs)}')
        eng = self.engine(u_target, use_reg=False, learning_rate=learning_rate, num_gd_iterations=num_iterations)
        if initial_angles is None:
            from .prng import repeats_angles
            init = repeats_angles(num_repeats, self.num_angles)
            is_batch = num_repeats != 1
        else:
            if num_repeats != 1:
                print('Warning, initial conditions provided and number of repeats will be ignored.')
            init = np.asarray(initial_angles, dtype=np.float32)
            is_batch = init.ndim == 2
            init = np.atleast_2d(init)
        out = eng.optimize(init, keep_history=keep_history)
        results = []
        for s in range(init.shape[0]):
            if keep_history:
                params = out['history_params'][s].cpu().numpy()
                regloss = out['history_regloss'][s].cpu().numpy()
            else:
                params = out['params'][s].cpu().numpy()
                regloss = out['regloss'][s].cpu().numpy()
            reg = np.zeros_like(regloss)
            results.append({'params': params, 'loss': regloss - reg, 'reg': reg, 'regloss': regloss})
        return results if is_batch else results[0]
