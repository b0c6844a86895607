"""ORACLE — test infrastructure only (see oracle/cpf_oracle.c header).

Python face of the CPU restatement: a ctypes wrapper over
oracle/_build/libcpf_oracle.so (`Oracle`), the NumPy threefry restatement
(`oracle.prng`) and a tiny literal tensordot/transpose NumPy restatement
(`numpy_unitary`) used to cross-check qubit conventions on small cases.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  cpflow_amd never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import prng  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, '_build', 'libcpf_oracle.so')


class _Penalty(C.Structure):
    _fields_ = [('kind', C.c_int32), ('xmax', C.c_double), ('ymax', C.c_double),
                ('plato_0', C.c_double), ('plato_1', C.c_double), ('plato_2', C.c_double)]


class _Problem(C.Structure):
    _fields_ = [('n_qubits', C.c_int32), ('num_blocks', C.c_int32), ('placements', C.c_void_p),
                ('n_rot', C.c_int32), ('rot_axes', C.c_char_p), ('entangler', C.c_int32),
                ('loss_kind', C.c_int32), ('target', C.c_void_p), ('penalty', _Penalty),
                ('use_reg', C.c_int32), ('r', C.c_double),
                ('lr', C.c_double), ('b1', C.c_double), ('b2', C.c_double), ('eps', C.c_double),
                ('num_iterations', C.c_int32), ('threshold_cp', C.c_double), ('entry_loss', C.c_double),
                ('frozen_mask', C.c_void_p)]


def build(native=False):
    """Compile the C restatement (gcc).  Building the checker is not using it."""
    subprocess.check_call(['make', '-s', '-C', _HERE] + (['native'] if native else []))


def _load(path=None):
    path = path or _LIB_PATH
    if not os.path.exists(path):
        build()
    return C.CDLL(path)


def placements_for(layer, num_blocks):
    """topology.py:15-20 fill_layers + main.py:175 all_placements: block b sits on layer[b % len(layer)]."""
    layer = [list(map(int, p)) for p in layer]
    return np.array([layer[b % len(layer)] for b in range(num_blocks)], dtype=np.int32).reshape(-1, 2)


_ENT = {'cp': 0, 'cz': 1, 'cx': 2}


class Oracle:
    """One CP-template problem (layer, K, rotation letters, target, options)."""

    def __init__(self, layer, num_cp_gates, target, rotation_gates='xyz', r=0.00055, entangler='cp',
                 learning_rate=0.1, num_gd_iterations=2000, threshold_cp=0.2, entry_loss=1e-3,
                 use_reg=True, penalty='linear', xmax=np.pi / 2, ymax=2.0, plato_0=0.05, plato_1=0.05,
                 plato_2=0.05, loss_kind=0, frozen_mask=None, lib_path=None):
        self.lib = _load(lib_path)
        self.layer = [list(map(int, p)) for p in layer]
        self.n = max(max(p) for p in self.layer) + 1
        self.N = 1 << self.n
        self.K = int(num_cp_gates)
        self.rot = str(rotation_gates)
        self.entangler = entangler
        self.nb = 2 * len(self.rot) + (1 if entangler == 'cp' else 0)
        self.P = 3 * self.n + self.nb * self.K
        self._placements = placements_for(self.layer, self.K)
        self._target = np.ascontiguousarray(np.asarray(target, dtype=np.complex128))
        assert self._target.shape == (self.N, self.N)
        self._rot_b = self.rot.encode()
        self._frozen = None if frozen_mask is None else np.ascontiguousarray(frozen_mask, dtype=np.uint8)
        pb = _Problem()
        pb.n_qubits, pb.num_blocks = self.n, self.K
        pb.placements = self._placements.ctypes.data
        pb.n_rot, pb.rot_axes = len(self.rot), self._rot_b
        pb.entangler, pb.loss_kind = _ENT[entangler], loss_kind
        pb.target = self._target.ctypes.data
        pb.penalty = _Penalty(0 if penalty == 'linear' else 1, xmax, ymax, plato_0, plato_1, plato_2)
        pb.use_reg, pb.r = int(bool(use_reg) and entangler == 'cp'), float(r)
        pb.lr, pb.b1, pb.b2, pb.eps = float(learning_rate), 0.9, 0.999, 1e-8
        pb.num_iterations = int(num_gd_iterations)
        pb.threshold_cp, pb.entry_loss = float(threshold_cp), float(entry_loss)
        pb.frozen_mask = None if self._frozen is None else self._frozen.ctypes.data
        self.pb = pb

    @property
    def cp_mask(self):
        m = np.zeros(self.P, dtype=np.uint8)
        if self.entangler == 'cp':
            m[3 * self.n + self.nb - 1::self.nb] = 1
        return m

    @staticmethod
    def _sfx(dtype):
        dtype = np.dtype(dtype)
        return ('f32', C.c_float) if dtype == np.float32 else ('f64', C.c_double)

    def unitary(self, theta, dtype=np.float32):
        sfx, _ = self._sfx(dtype)
        th = np.ascontiguousarray(theta, dtype=dtype)
        assert th.shape == (self.P,)
        U = np.empty((self.N, self.N, 2), dtype=dtype)
        getattr(self.lib, f'cpfo_unitary_{sfx}')(C.byref(self.pb), th.ctypes.data_as(C.c_void_p), U.ctypes.data_as(C.c_void_p))
        return U[..., 0] + 1j * U[..., 1]

    def loss_grad(self, theta, dtype=np.float32, with_grad=True):
        """-> dict(loss, reg, regloss, grad) for a batch of angle vectors [S, P]."""
        sfx, _ = self._sfx(dtype)
        th = np.ascontiguousarray(np.atleast_2d(theta), dtype=dtype)
        S = th.shape[0]
        assert th.shape[1] == self.P
        loss, reg, regloss = (np.empty(S, dtype=dtype) for _ in range(3))
        grad = np.empty((S, self.P), dtype=dtype) if with_grad else None
        getattr(self.lib, f'cpfo_loss_grad_{sfx}')(
            C.byref(self.pb), C.c_int32(S), th.ctypes.data_as(C.c_void_p), loss.ctypes.data_as(C.c_void_p),
            reg.ctypes.data_as(C.c_void_p), regloss.ctypes.data_as(C.c_void_p),
            grad.ctypes.data_as(C.c_void_p) if with_grad else None)
        return dict(loss=loss, reg=reg, regloss=regloss, grad=grad)

    def run(self, init, dtype=np.float32, keep_history=False, nthreads=None):
        """mynimize_repeated + evaluate fields for a batch of initial angles [S, P]."""
        sfx, _ = self._sfx(dtype)
        init = np.ascontiguousarray(np.atleast_2d(init), dtype=dtype)
        S, P, T = init.shape[0], self.P, self.pb.num_iterations
        assert init.shape[1] == P
        out = dict(params=np.empty((S, 2, P), dtype=dtype), regloss=np.empty((S, 2), dtype=dtype),
                   loss=np.empty((S, 2), dtype=dtype), reg=np.empty((S, 2), dtype=dtype),
                   cz=np.empty(S, dtype=np.int32), passed=np.empty(S, dtype=np.uint8))
        hp = np.empty((S, T, P), dtype=dtype) if keep_history else None
        hl = np.empty((S, T), dtype=dtype) if keep_history else None
        if nthreads is None:
            nthreads = self.lib.cpfo_max_threads()
        p = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None  # noqa: E731
        err = getattr(self.lib, f'cpfo_run_{sfx}')(
            C.byref(self.pb), C.c_int32(S), p(init), p(out['params']), p(out['regloss']), p(out['loss']),
            p(out['reg']), p(out['cz']), p(out['passed']), p(hp), p(hl), C.c_int32(int(nthreads)))
        if err:
            raise MemoryError('oracle workspace allocation failed')
        out['passed'] = out['passed'].astype(bool)
        if keep_history:
            out['history_params'], out['history_loss'] = hp, hl
        return out

    def run_simd(self, init, nthreads=None):
        """The float32 run over 16 samples at a time (cpf_oracle_simd.h): bench.py's "port-simd" CPU baseline.  Same outputs as
        run(init, np.float32) without history; raises NotImplementedError for what the SIMD build does not cover."""
        init = np.ascontiguousarray(np.atleast_2d(init), dtype=np.float32)
        S, P = init.shape[0], self.P
        assert init.shape[1] == P
        out = dict(params=np.empty((S, 2, P), dtype=np.float32), regloss=np.empty((S, 2), dtype=np.float32),
                   loss=np.empty((S, 2), dtype=np.float32), reg=np.empty((S, 2), dtype=np.float32),
                   cz=np.empty(S, dtype=np.int32), passed=np.empty(S, dtype=np.uint8))
        if nthreads is None:
            nthreads = self.lib.cpfo_max_threads()
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        err = self.lib.cpfo_run_f32_simd(C.byref(self.pb), C.c_int32(S), p(init), p(out['params']), p(out['regloss']), p(out['loss']),
                                         p(out['reg']), p(out['cz']), p(out['passed']), C.c_int32(int(nthreads)))
        if err == -2:
            raise NotImplementedError('the SIMD build covers cost_HST with CP / CZ blocks and no frozen mask only')
        if err:
            raise MemoryError('oracle workspace allocation failed')
        out['passed'] = out['passed'].astype(bool)
        return out

    def count_cz(self, theta, dtype=np.float32):
        sfx, _ = self._sfx(dtype)
        th = np.ascontiguousarray(theta, dtype=dtype)
        return int(getattr(self.lib, f'cpfo_count_cz_{sfx}')(C.byref(self.pb), th.ctypes.data_as(C.c_void_p)))

    def penalty(self, a, dtype=np.float32):
        sfx, ct = self._sfx(dtype)
        f = getattr(self.lib, f'cpfo_penalty_{sfx}')
        f.restype = ct
        slope = ct(0)
        v = f(C.byref(self.pb), ct(float(a)), C.byref(slope))
        return float(v), float(slope.value)

    def initial_angles(self, seed, num_samples, first=0, count=None, cp_dist='uniform'):
        count = num_samples - first if count is None else count
        out = np.empty((count, self.P), dtype=np.float32)
        mask = self.cp_mask
        self.lib.cpfo_initial_angles(C.c_uint32(int(seed) & 0xFFFFFFFF), C.c_int32(num_samples), C.c_int32(first),
                                     C.c_int32(count), C.c_int32(self.P), mask.ctypes.data_as(C.c_void_p),
                                     C.c_int32(1 if cp_dist == '0' else 0), out.ctypes.data_as(C.c_void_p))
        return out


def filter_results(out, threshold_cz=float('inf')):
    """cp_utils.py:167-202 filter_cp_results on an Oracle.run() dict: [(cz, sample_index)], stable sort by cz."""
    sel = [(int(c), i) for i, (c, ok) in enumerate(zip(out['cz'], out['passed'])) if ok and c <= threshold_cz]
    sel.sort(key=lambda x: x[0])
    return sel


# ---- literal NumPy restatement of the reference's tensor contraction (convention cross-check) ----
def _rot(axis, a):
    c, s = np.cos(a / 2), np.sin(a / 2)
    sig = {'x': np.array([[0, 1], [1, 0]], dtype=complex), 'y': np.array([[0, -1j], [1j, 0]]),
           'z': np.array([[1, 0], [0, -1]], dtype=complex)}[axis]
    return c * np.eye(2) - 1j * s * sig                                  # gates.py:22-23


def _apply(gate, tensor, placement):
    """circuit_assembly.py:16-45: tensordot over the gate's input legs, then move the
    gate's output legs (which tensordot puts first) back to the placement positions."""
    k, n = gate.ndim // 2, tensor.ndim // 2
    out = np.tensordot(gate, tensor, axes=[list(range(k, 2 * k)), list(placement)])
    rest = [ax for ax in range(2 * n) if ax not in placement]
    perm = [0] * (2 * n)
    for j, pos in enumerate(placement):
        perm[pos] = j
    for j, pos in enumerate(rest):
        perm[pos] = k + j
    return np.transpose(out, perm)


def numpy_unitary(layer, num_blocks, theta, rotation_gates='xyz', entangler='cp'):
    """main.py:106-146 build_unitary in complex128, gate by gate."""
    theta = np.asarray(theta, dtype=np.float64)
    n = max(max(p) for p in layer) + 1
    nb = 2 * len(rotation_gates) + (entangler == 'cp')
    u = np.eye(2 ** n, dtype=complex).reshape([2] * (2 * n))
    for i in range(n):
        a = theta[3 * i:3 * i + 3]
        u = _apply(_rot('z', a[2]) @ _rot('x', a[1]) @ _rot('z', a[0]), u, [i])
    pl = placements_for(layer, num_blocks)
    for b in range(num_blocks):
        a = theta[3 * n + nb * b:3 * n + nb * (b + 1)]
        if entangler == 'cp':
            g = np.diag([1, 1, 1, np.exp(1j * a[-1])])
        elif entangler == 'cz':
            g = np.diag([1, 1, 1, -1]).astype(complex)
        else:
            g = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, 1], [0, 0, 1, 0]], dtype=complex)
        for j, ax in enumerate(rotation_gates):
            g = np.kron(_rot(ax, a[2 * j]), _rot(ax, a[2 * j + 1])) @ g
        u = _apply(g.reshape(2, 2, 2, 2), u, list(pl[b]))
    return u.reshape(2 ** n, 2 ** n)


def cost_hst(u, v):
    """matrix_utils.py:35-42."""
    n = v.shape[0]
    return 1 - np.abs((u * np.conj(v)).sum()) ** 2 / n ** 2
