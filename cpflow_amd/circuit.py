"""A small gate-list circuit IR, so that decompositions are real objects without qiskit.

The reference hands every circuit to qiskit (`QuantumCircuit`, `Operator`, `transpile`);
qiskit is not on the hot path and is absent on the GPU box, so the pieces of it the
synthesis front end needs are provided here:
    gate list + counting + CZ depth   <- gates_count / gates_depth   cpflow/exact_decompositions.py:280-290
    CP -> {id, CZ, 2 CZ + 1q}         <- cp_to_cz_circuit / cp_to_cz_gate   cpflow/exact_decompositions.py:42-74
    merge 1q runs into Rz Rx Rz       <- convert_to_ZXZ              cpflow/exact_decompositions.py:178-190
    unitary (qubit 0 = MSB)           <- Operator(circuit.reverse_bits()).data   cpflow/main.py:265
`to_qiskit()` converts to a real QuantumCircuit when qiskit is importable.
"""
import cmath
import math

import numpy as np

_ONE_QUBIT = ('rx', 'ry', 'rz', 'h')
_TWO_QUBIT = ('cz', 'cp', 'cx')


def _rot(axis, a):
    c, s = math.cos(a / 2), math.sin(a / 2)
    if axis == 'x':
        return np.array([[c, -1j * s], [-1j * s, c]])
    if axis == 'y':
        return np.array([[c, -s], [s, c]], dtype=complex)
    return np.array([[cmath.exp(-0.5j * a), 0], [0, cmath.exp(0.5j * a)]])


_H = np.array([[1, 1], [1, -1]], dtype=complex) / math.sqrt(2)


def gate_matrix(name, params):
    if name in ('rx', 'ry', 'rz'):
        return _rot(name[1], float(params[0]))
    if name == 'h':
        return _H
    if name == 'cz':
        return np.diag([1, 1, 1, -1]).astype(complex)
    if name == 'cp':
        return np.diag([1, 1, 1, cmath.exp(1j * float(params[0]))])
    if name == 'cx':
        return np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, 1], [0, 0, 1, 0]], dtype=complex)
    raise TypeError(f"Gate '{name}' not implemented.")


_SQ = math.sqrt(0.5)


def _g1(name, params):
    """One-qubit gate as a tuple (u00, u01, u10, u11) of Python complex numbers: the run-merging
    loops below multiply hundreds of 2x2 matrices per circuit, where numpy's call overhead dominates."""
    if name == 'h':
        return (_SQ, _SQ, _SQ, -_SQ)
    a = float(params[0]) / 2
    c, s = math.cos(a), math.sin(a)
    if name == 'rx':
        return (c, -1j * s, -1j * s, c)
    if name == 'ry':
        return (c, -s, s, c)
    if name == 'rz':
        return (complex(c, -s), 0.0, 0.0, complex(c, s))
    raise TypeError(f"Gate '{name}' not implemented.")


def _mul2(g, h):
    return (g[0] * h[0] + g[1] * h[2], g[0] * h[1] + g[1] * h[3],
            g[2] * h[0] + g[3] * h[2], g[2] * h[1] + g[3] * h[3])


def zxz_angles(u):
    """(z1, x, z2) with u = phase * Rz(z2) Rx(x) Rz(z1); u is a 2x2 array or a (u00, u01, u10, u11) tuple."""
    if not isinstance(u, tuple):
        u = np.asarray(u, dtype=complex)
        u = (complex(u[0, 0]), complex(u[0, 1]), complex(u[1, 0]), complex(u[1, 1]))
    root = cmath.sqrt(u[0] * u[3] - u[1] * u[2])
    alpha, beta = u[0] / root, u[2] / root
    x = 2 * math.atan2(abs(beta), abs(alpha))
    if abs(beta) < 1e-12:
        return -2 * cmath.phase(alpha), 0.0, 0.0
    if abs(alpha) < 1e-12:
        return -2 * (cmath.phase(beta) + math.pi / 2), x, 0.0
    plus = -2 * cmath.phase(alpha)                       # z1 + z2
    minus = -2 * (cmath.phase(beta) + math.pi / 2)       # z1 - z2
    return (plus + minus) / 2, x, (plus - minus) / 2


def _bracket(a):
    """Angle shifted by a multiple of 2 pi into (-pi, pi]."""
    a = math.fmod(a + math.pi, 2 * math.pi)
    if a <= 0:
        a += 2 * math.pi
    return a - math.pi


class Circuit:
    """Ordered list of (name, qubits, params); qubit 0 is the most significant bit of matrix indices."""

    def __init__(self, num_qubits):
        self.num_qubits = int(num_qubits)
        self.data = []

    # ---- construction
    def append(self, name, qubits, params=()):
        qubits = tuple(int(q) for q in qubits)
        if name not in _ONE_QUBIT + _TWO_QUBIT:
            raise TypeError(f"Gate '{name}' not implemented.")
        assert len(qubits) == (1 if name in _ONE_QUBIT else 2) and all(0 <= q < self.num_qubits for q in qubits)
        self.data.append((name, qubits, tuple(params)))
        return self

    def rx(self, a, q): return self.append('rx', (q,), (a,))
    def ry(self, a, q): return self.append('ry', (q,), (a,))
    def rz(self, a, q): return self.append('rz', (q,), (a,))
    def h(self, q): return self.append('h', (q,))
    def cz(self, q0, q1): return self.append('cz', (q0, q1))
    def cx(self, q0, q1): return self.append('cx', (q0, q1))
    def cp(self, a, q0, q1): return self.append('cp', (q0, q1), (a,))

    def copy(self):
        c = Circuit(self.num_qubits)
        c.data = list(self.data)
        return c

    def __len__(self):
        return len(self.data)

    # ---- analysis
    def count_ops(self):
        ops = {}
        for name, _, _ in self.data:
            ops[name] = ops.get(name, 0) + 1
        return ops

    def depth(self, gate_names=None):
        """Circuit depth counting only gates whose name is in gate_names (all gates if None);
        uncounted multi-qubit gates still synchronise their qubits, as in qiskit's filtered depth."""
        level = [0] * self.num_qubits
        for name, qubits, _ in self.data:
            counted = gate_names is None or name in gate_names
            top = max(level[q] for q in qubits) + (1 if counted else 0)
            if counted or len(qubits) > 1:
                for q in qubits:
                    level[q] = top
        return max(level) if level else 0

    def unitary(self):
        """Runs of one-qubit gates are multiplied as 2x2 matrices and applied once, when a
        two-qubit gate (or the end) needs the qubit; CZ / CP scale and CX swaps slices in place."""
        n = self.num_qubits
        dim = 2 ** n
        u = np.eye(dim, dtype=complex)
        pending = [None] * n

        def flush(q):
            if pending[q] is None:
                return u
            g = np.array(pending[q], dtype=complex).reshape(2, 2)
            pending[q] = None
            return np.matmul(g, u.reshape(2 ** q, 2, -1)).reshape(dim, dim)

        for name, qubits, params in self.data:
            if len(qubits) == 1:
                q, g = qubits[0], _g1(name, params)
                pending[q] = g if pending[q] is None else _mul2(g, pending[q])
                continue
            q0, q1 = qubits
            u = flush(q0)
            u = flush(q1)
            lo, hi = min(q0, q1), max(q0, q1)
            v = u.reshape(2 ** lo, 2, 2 ** (hi - lo - 1), 2, -1)      # axes 1, 3 = qubits lo, hi
            if name == 'cx':
                w = v[:, 1] if q0 == lo else v[:, :, :, 1]            # control = 1
                w[...] = np.flip(w, axis=2 if q0 == lo else 1).copy()  # exchange the target's two slices
            else:
                v[:, 1, :, 1, :] *= -1.0 if name == 'cz' else cmath.exp(1j * float(params[0]))
        for q in range(n):
            u = flush(q)
        return u

    # ---- rewriting
    def cp_to_cz(self, cp_threshold=0.2):
        """CP(a) -> nothing if |a| <= threshold, CZ if |a - pi| <= threshold, otherwise
        Rz(a/2) Rz(a/2) H CZ Rx(-a/2) CZ H  (two CZ gates), exact up to a global phase."""
        out = Circuit(self.num_qubits)
        for name, qubits, params in self.data:
            if name != 'cp':
                out.data.append((name, qubits, params))
                continue
            a = float(params[0])
            c, t = qubits
            if abs(a) <= cp_threshold:
                continue
            if abs(a - math.pi) <= cp_threshold:
                out.cz(c, t)
                continue
            out.h(t).cz(c, t).rx(-a / 2, t).cz(c, t).h(t).rz(a / 2, t).rz(a / 2, c)
        return out

    def to_zxz(self, drop_identity=1e-12):
        """Merge every run of 1q gates on a wire into one Rz Rx Rz triple."""
        out = Circuit(self.num_qubits)
        pending = [None] * self.num_qubits

        def flush(q):
            u = pending[q]
            pending[q] = None
            if u is None:
                return
            z1, x, z2 = zxz_angles(u)
            if abs(_bracket(x)) < drop_identity and abs(_bracket(z1 + z2)) < drop_identity:
                return
            out.rz(_bracket(z1), q).rx(_bracket(x), q).rz(_bracket(z2), q)

        for name, qubits, params in self.data:
            if len(qubits) == 1:
                q, g = qubits[0], _g1(name, params)
                pending[q] = g if pending[q] is None else _mul2(g, pending[q])
            else:
                for q in qubits:
                    flush(q)
                out.data.append((name, qubits, params))
        for q in range(self.num_qubits):
            flush(q)
        return out

    # ---- export
    def qasm(self):
        lines = ['OPENQASM 2.0;', 'include "qelib1.inc";', f'qreg q[{self.num_qubits}];']
        for name, qubits, params in self.data:
            arg = f"({','.join(repr(float(p)) for p in params)})" if params else ''
            gate = 'cu1' if name == 'cp' else name
            lines.append(f"{gate}{arg} {','.join(f'q[{q}]' for q in qubits)};")
        return '\n'.join(lines) + '\n'

    def to_qiskit(self):
        from qiskit import QuantumCircuit   # optional dependency
        qc = QuantumCircuit(self.num_qubits)
        for name, qubits, params in self.data:
            getattr(qc, name)(*[float(p) for p in params], *qubits)
        return qc

    def draw(self, max_width=120):
        """Plain-text drawing, one row per qubit."""
        rows = [[] for _ in range(self.num_qubits)]

        def pad():
            w = max(sum(len(x) for x in r) for r in rows)
            for r in rows:
                r.append('-' * (w - sum(len(x) for x in r)))

        for name, qubits, params in self.data:
            if len(qubits) == 1:
                label = name.upper() + ((f'({params[0]:.3g})' if not isinstance(params[0], str) else f'({params[0]})') if params else '')
                rows[qubits[0]].append(f'-[{label}]-')
            else:
                pad()
                lo, hi = min(qubits), max(qubits)
                tag = ('@', '@') if name == 'cz' else (('@', 'X') if name == 'cx' else ('@', f'P({float(params[0]):.3g})'))
                w = max(len(tag[0]), len(tag[1])) + 2
                for q in range(self.num_qubits):
                    if q == qubits[0]:
                        rows[q].append(('-' + tag[0]).ljust(w, '-'))
                    elif q == qubits[1]:
                        rows[q].append(('-' + tag[1]).ljust(w, '-'))
                    elif lo < q < hi:
                        rows[q].append('-|'.ljust(w, '-'))
                    else:
                        rows[q].append('-' * w)
        pad()
        text = [f'q{q}: ' + ''.join(r) for q, r in enumerate(rows)]
        if max_width and len(text[0]) > max_width:
            chunks = []
            for start in range(0, len(text[0]), max_width):
                chunks.append('\n'.join(t[start:start + max_width] for t in text))
            return '\n\n'.join(chunks)
        return '\n'.join(text)

    def __repr__(self):
        ops = ', '.join(f'{k}: {v}' for k, v in sorted(self.count_ops().items()))
        return f'<Circuit {self.num_qubits} qubits | {ops}>'
