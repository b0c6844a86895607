"""CPU: the per-block algebra of the ZYZ-form kernel (cpflow_amd/csrc/cpf_zyz_math.h, compiled here with g++).  The test plays the
kernel: build phase and chain phase through the header's functions (float32), the state sweeps in NumPy exactly as cpf_zyz.h
runs them (merged phase per amplitude of the pair in the kernel's index order, unnormalised real stages, generator sums, the
|11> sum before conj(d)), against the gate-by-gate unitary of the template (cpflow/main.py:69-82,106-146, gates.py:22-58) and
central differences - including identity rotations (t = 0), exact flips (theta = pi) and all-diagonal blocks."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KIND = {'x': 0, 'y': 1, 'z': 2}
PAULI = (np.array([[0, 1], [1, 0]], complex), np.array([[0, -1j], [1j, 0]]), np.diag([1.0 + 0j, -1.0]))
P = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731


@pytest.fixture(scope='module')
def shim(tmp_path_factory):
    out = str(tmp_path_factory.mktemp('zm') / 'libzm.so')
    subprocess.check_call(['g++', '-O1', '-shared', '-fPIC', '-Wall', '-Wno-unknown-pragmas', '-I', os.path.join(ROOT, 'cpflow_amd', 'csrc'),
                           '-o', out, os.path.join(ROOT, 'tests', 'zyz_math_shim.cpp')])
    lib = C.CDLL(out)
    lib.zm_chain_cp.restype = C.c_float
    lib.zm_chain_cp.argtypes = [C.c_float, C.c_float, C.c_float, C.c_int, C.c_float]
    lib.zm_chain_side.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    lib.zm_chain_surface.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_int, C.c_void_p]
    lib.zm_block_table.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_float]
    return lib


def rot(k, a):
    return np.cos(a / 2) * np.eye(2) - 1j * np.sin(a / 2) * PAULI[k]


def embed(n, q, m):
    out = np.eye(1)
    for i in range(n):
        out = np.kron(out, m if i == q else np.eye(2))
    return out


def bit(n, r, q):
    return (r >> (n - 1 - q)) & 1


def reference_unitary(n, placements, theta, rots, ent):
    nb = 2 * len(rots) + (1 if ent == 'cp' else 0)
    U = np.eye(1 << n, dtype=complex)
    for q in range(n):
        U = embed(n, q, rot(2, theta[3 * q + 2]) @ rot(0, theta[3 * q + 1]) @ rot(2, theta[3 * q])) @ U
    for b, (up, dn) in enumerate(placements):
        t = theta[3 * n + nb * b:3 * n + nb * (b + 1)]
        lam = np.exp(1j * t[-1]) if ent == 'cp' else -1.0
        U = np.array([lam if bit(n, r, up) and bit(n, r, dn) else 1.0 for r in range(1 << n)])[:, None] * U
        for i, g in enumerate(rots):
            U = embed(n, up, rot(KIND[g], t[2 * i])) @ embed(n, dn, rot(KIND[g], t[2 * i + 1])) @ U
    return U


def loss_of(U, V):
    return 1 - abs(np.sum(U * np.conj(V))) ** 2 / U.shape[0] ** 2


def kernel_play(shim, n, placements, theta, V, rots, ent, xyz=False, up_low_of=lambda b: b % 2 == 0, layer_len=None):
    """loss and gradient the way cpf_zyz.h computes them; the build / chain algebra through the C++ header.
    layer_len: the real scale once per layer (cpf_zyz.h "build B") - the phases of a block have unit modulus unless it is the last of
    its layer or of the circuit, which carries the product of d_0 over its layer so far; a block's sums are divided by phi^2."""
    N, K, nr = 1 << n, len(placements), len(rots)
    nb = 2 * nr + (1 if ent == 'cp' else 0)
    th32 = theta.astype(np.float32)
    recs = []
    for b in range(K):
        t = th32[3 * n + nb * b:3 * n + nb * (b + 1)]
        rec = np.zeros((1 + 2 * nr, 4), np.float32)
        rec[0] = [np.cos(t[-1]), np.sin(t[-1]), 3, 0] if ent == 'cp' else [-1, 0, 4, 0]
        for i, g in enumerate(rots):
            for s in (0, 1):
                rec[1 + 2 * i + s] = [np.cos(t[2 * i + s] / 2), np.sin(t[2 * i + s] / 2), KIND[g], 0]
        recs.append(rec)
    sides = np.zeros((K, 2, 8), np.float32)
    for b in range(K):
        for s in (0, 1):
            shim.zm_side_build(P(recs[b]), nr, s, int(xyz), P(sides[b, s]))
            assert sides[b, s, 0] <= 1 + 1e-6 and sides[b, s, 1] >= 0.7071
    prev, last, pend = [[None, None] for _ in range(K)], {}, [0] * n
    for b, pl in enumerate(placements):
        for s, q in enumerate(pl):
            prev[b][s] = last.get(q)
            last[q] = (b, s)
    for b in range(K - 1, -1, -1):
        for s, q in enumerate(placements[b]):
            sides[b, s, 7] = pend[q]
            pend[q] ^= int(sides[b, s, 6])
    m_surf = pend
    tabs, cacb = np.zeros((K, 10), np.float32), [0] * K
    d0 = (sides[:, 0, 1] * sides[:, 1, 1]).astype(np.float32)
    invphi2 = np.ones(K)
    for b in range(K):
        pu = None if prev[b][0] is None else P(sides[prev[b][0]])
        pd = None if prev[b][1] is None else P(sides[prev[b][1]])
        scale = -1.0
        if layer_len:
            pre = np.float32(np.prod(d0[b - b % layer_len:b + 1]))
            carries = b % layer_len == layer_len - 1 or b == K - 1
            scale = float(pre) if carries else 1.0
            invphi2[b] = 1.0 if carries else 1.0 / float(pre) ** 2
        cacb[b] = shim.zm_block_table(P(tabs[b]), P(sides[b, 0]), P(sides[b, 1]), pu, pd, recs[b][0, 0], recs[b][0, 1], int(up_low_of(b)), scale)
        assert tabs[b, 1] == 0
        mods = np.hypot(tabs[b, 2:8:2], tabs[b, 3:8:2])
        assert np.abs(mods - tabs[b, 0]).max() < 1e-6 and abs(tabs[b, 0] - (scale if layer_len else d0[b])) < 1e-7

    def kidx(b, r):          # kernel index of row r within block b's pair: 2 hi + lo
        up, dn = placements[b]
        lo, hi = (up, dn) if up_low_of(b) else (dn, up)
        return 2 * bit(n, r, hi) + bit(n, r, lo)

    def dvec(b):
        d = np.array([tabs[b, 0], tabs[b, 2] + 1j * tabs[b, 3], tabs[b, 4] + 1j * tabs[b, 5], tabs[b, 6] + 1j * tabs[b, 7]])
        return np.array([d[kidx(b, r)] for r in range(N)])

    def ystage(b, sign):
        up, dn = placements[b]
        lo, hi = (up, dn) if up_low_of(b) else (dn, up)
        tl, thh = sign * float(tabs[b, 8]), sign * float(tabs[b, 9])
        return embed(n, lo, np.array([[1, -tl], [tl, 1]])) @ embed(n, hi, np.array([[1, -thh], [thh, 1]]))

    U = np.ones((N, N), complex)
    surf_cs = np.zeros((n, 6), np.float32)
    for q in range(n):
        a = th32[3 * q:3 * q + 3] / 2
        surf_cs[q] = [np.cos(a[0]), np.sin(a[0]), np.cos(a[1]), np.sin(a[1]), np.cos(a[2]), np.sin(a[2])]
        sm = np.zeros(8, np.float32)
        shim.zm_surface_matrix(P(sm), P(surf_cs[q]), m_surf[q])
        S = (sm[0::2] + 1j * sm[1::2]).reshape(2, 2)
        for r in range(N):
            for c in range(N):
                U[r, c] *= S[bit(n, r, q), bit(n, c, q)]
    for b in range(K):
        U = ystage(b, 1) @ (dvec(b)[:, None] * U)
    lfin = [(1.0, 1.0) if q not in last else (sides[last[q]][2] + 1j * sides[last[q]][3], sides[last[q]][4] + 1j * sides[last[q]][5]) for q in range(n)]
    dfin = np.array([np.prod([lfin[q][bit(n, r, q)] for q in range(n)]) for r in range(N)])
    Vd = np.conj(dfin)[:, None] * V
    tr = np.sum(U * np.conj(Vd))
    loss = 1 - abs(tr) ** 2 / N ** 2
    W = -(tr / N ** 2) * Vd
    gen = lambda q, k, W, U: np.sum(np.conj(W) * (embed(n, q, PAULI[k]) @ U)).imag  # noqa: E731
    Pxyz, S11 = np.zeros((K, 2, 3)), np.zeros(K)
    for b in range(K - 1, -1, -1):
        for s, q in enumerate(placements[b]):
            Pxyz[b, s] = [invphi2[b] * gen(q, k, W, U) for k in range(3)]
        U, W = ystage(b, -1) @ U, ystage(b, -1) @ W
        mask = np.array([kidx(b, r) == 3 for r in range(N)])
        S11[b] = invphi2[b] * np.sum(np.conj(W[mask]) * U[mask]).imag
        U, W = np.conj(dvec(b))[:, None] * U, np.conj(dvec(b))[:, None] * W
    Psurf = np.array([[gen(q, k, W, U) for k in range(3)] for q in range(n)])
    grad = np.zeros_like(theta)
    for b in range(K):
        for s in (0, 1):
            g = np.zeros(nr, np.float32)
            shim.zm_chain_side(P(g), *[float(v) for v in Pxyz[b, s]], P(sides[b, s]), P(recs[b]), nr, s, int(xyz))
            grad[3 * n + nb * b + s:3 * n + nb * b + 2 * nr:2] = g
        if ent == 'cp':
            pz = [Psurf[placements[b][s], 2] if prev[b][s] is None else Pxyz[prev[b][s][0], prev[b][s][1], 2] for s in (0, 1)]
            grad[3 * n + nb * b + nb - 1] = shim.zm_chain_cp(float(S11[b]), float(pz[0]), float(pz[1]), cacb[b], float(d0[b]))
    for q in range(n):
        g = np.zeros(3, np.float32)
        shim.zm_chain_surface(P(g), *[float(v) for v in Psurf[q]], m_surf[q], P(surf_cs[q]))
        grad[3 * q:3 * q + 3] = g
    return loss, grad, int(sides[:, :, 6].sum())


def _check(shim, n, placements, theta, rots, ent, rng, xyz=False, layer_len=None):
    a = rng.normal(size=(1 << n, 1 << n)) + 1j * rng.normal(size=(1 << n, 1 << n))
    V = np.linalg.qr(a)[0]
    loss, grad, flips = kernel_play(shim, n, placements, theta, V, rots, ent, xyz, layer_len=layer_len)
    assert abs(loss - loss_of(reference_unitary(n, placements, theta, rots, ent), V)) < 3e-6
    fd = np.zeros_like(theta)
    for j in range(len(theta)):
        e = np.zeros_like(theta)
        e[j] = 1e-6
        fd[j] = (loss_of(reference_unitary(n, placements, theta + e, rots, ent), V) - loss_of(reference_unitary(n, placements, theta - e, rots, ent), V)) / 2e-6
    assert np.abs(grad - fd).max() < 3e-6, (np.abs(grad - fd).max(), np.argmax(np.abs(grad - fd)))
    return flips


@pytest.mark.parametrize('rots,ent', [('xyz', 'cp'), ('xz', 'cp'), ('zyx', 'cz'), ('y', 'cp'), ('zxzy', 'cp'), ('xyz', 'cz')])
def test_random_circuits_loss_and_gradient(shim, rots, ent):
    rng = np.random.default_rng(len(rots) + (7 if ent == 'cz' else 0))
    nb = 2 * len(rots) + (1 if ent == 'cp' else 0)
    flips = 0
    for _ in range(6):
        n = int(rng.integers(2, 5))
        pairs = [(i, j) for i in range(n) for j in range(n) if i != j]
        placements = [pairs[int(rng.integers(len(pairs)))] for _ in range(int(rng.integers(1, 7)))]
        theta = rng.uniform(-2 * np.pi, 2 * np.pi, 3 * n + nb * len(placements))
        flips += _check(shim, n, placements, theta, rots, ent, rng)
    assert flips > 0                                   # the flipped branch was exercised


@pytest.mark.parametrize('layer_len', [1, 2, 3, 5])
def test_real_scale_once_per_layer(shim, layer_len):
    """cpf_zyz.h applies a block's real scale d_0 once per layer: unit-modulus phases, the last block of a layer (or of the
    circuit: the block counts below leave truncated last layers) carries the layer's product, sums divided by phi^2."""
    rng = np.random.default_rng(40 + layer_len)
    flips = 0
    for rots, ent, K in (('xyz', 'cp', 7), ('xz', 'cz', 6), ('xyz', 'cp', 11)):
        n = 4
        layer = [tuple(int(q) for q in rng.choice(n, 2, replace=False)) for _ in range(layer_len)]
        placements = [layer[b % layer_len] for b in range(K)]
        nb = 2 * len(rots) + (1 if ent == 'cp' else 0)
        theta = rng.uniform(-2 * np.pi, 2 * np.pi, 3 * n + nb * K)
        flips += _check(shim, n, placements, theta, rots, ent, rng, layer_len=layer_len)
    assert flips > 0


@pytest.mark.parametrize('case', ['identity', 'flip', 'diagonal', 'mixed'])
def test_degenerate_angles(shim, case):
    """t = 0 (identity rotations: the split of the diagonal is arbitrary), exact flips theta = pi (|p| = 0: the unflipped form
    would have an infinite tangent), all-diagonal blocks (only 'z' letters carry an angle), and random picks of all of them."""
    rng = np.random.default_rng({'identity': 1, 'flip': 2, 'diagonal': 3, 'mixed': 4}[case])
    n, rots, ent = 3, 'xyz', 'cp'
    placements = [(0, 1), (1, 2), (2, 0), (0, 1), (2, 1)]
    for rep in range(4):
        theta = rng.uniform(-2 * np.pi, 2 * np.pi, 3 * n + 7 * len(placements))
        blk = theta[3 * n:].reshape(-1, 7)
        if case == 'identity':
            blk[:, :6] = 0.0
            blk[rep % len(placements), :6] = rng.uniform(-3, 3, 6)
        elif case == 'flip':
            blk[:, 2:4] = np.pi * rng.choice([1.0, -1.0, 3.0], (len(placements), 2))
            blk[:, 0:2] = 0.0
        elif case == 'diagonal':
            blk[:, 0:4] = 0.0
        else:
            blk[:, :6] = rng.choice([0.0, np.pi, -np.pi, 2 * np.pi, 0.5 * np.pi], (len(placements), 6))
        _check(shim, n, placements, theta, rots, ent, rng)


def test_compile_time_rotation_string_equals_run_time_path(shim):
    """side_build / chain_side specialised on 'xyz' (template argument ROT = 8451, the ahead-of-time kernels) give bit for bit
    what the run-time-kind path gives on the same records."""
    rng = np.random.default_rng(5)
    for _ in range(20):
        rec = np.zeros((7, 4), np.float32)
        ang = rng.uniform(-7, 7, 7)
        rec[0] = [np.cos(ang[6]), np.sin(ang[6]), 3, 0]
        for i in range(3):
            for s in (0, 1):
                rec[1 + 2 * i + s] = [np.cos(ang[2 * i + s] / 2), np.sin(ang[2 * i + s] / 2), i, 0]
        for s in (0, 1):
            a, b = np.zeros(8, np.float32), np.zeros(8, np.float32)
            shim.zm_side_build(P(rec), 3, s, 0, P(a))
            shim.zm_side_build(P(rec), 3, s, 1, P(b))
            assert (a == b).all()
            ga, gb = np.zeros(3, np.float32), np.zeros(3, np.float32)
            px, py, pz = (float(v) for v in rng.normal(size=3))
            shim.zm_chain_side(P(ga), px, py, pz, P(a), P(rec), 3, s, 0)
            shim.zm_chain_side(P(gb), px, py, pz, P(a), P(rec), 3, s, 1)
            assert (ga == gb).all()


def test_numpy_specification_agrees():
    """tools/zyz_model.py (float64 NumPy statement of the whole scheme) passes its own check."""
    r = subprocess.run(['python3', os.path.join(ROOT, 'tools', 'zyz_model.py')], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
