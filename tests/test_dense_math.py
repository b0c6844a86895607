"""CPU: the per-block 4x4 algebra of the dense-block kernel (cpflow_amd/csrc/cpf_dense_math.h, compiled here with g++)
against a NumPy construction of the same block (cpflow/main.py:69-82, gates.py:22-58) and finite differences."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K_RX, K_RY, K_RZ, K_CP, K_CZ, K_CX = range(6)
PAULI = {K_RX: np.array([[0, 1], [1, 0]], complex), K_RY: np.array([[0, -1j], [1j, 0]]), K_RZ: np.diag([1, -1]).astype(complex)}


@pytest.fixture(scope='module')
def shim(tmp_path_factory):
    out = str(tmp_path_factory.mktemp('dm') / 'libdm.so')
    subprocess.check_call(['g++', '-O1', '-shared', '-fPIC', '-Wno-unknown-pragmas', '-I', os.path.join(ROOT, 'cpflow_amd', 'csrc'),
                           '-o', out, os.path.join(ROOT, 'tests', 'dense_math_shim.cpp')])
    return C.CDLL(out)


def rot(kind, a):
    return np.cos(a / 2) * np.eye(2) - 1j * np.sin(a / 2) * PAULI[kind]


def block(ent, kinds, angles):
    """G = prod_i (R_i(dn) R_i(up)) E in the basis 2 up + dn; angles = [entangler, up_0, dn_0, up_1, dn_1, ...]."""
    if ent == K_CP:
        G = np.diag([1, 1, 1, np.exp(1j * angles[0])])
    elif ent == K_CZ:
        G = np.diag([1, 1, 1, -1]).astype(complex)
    else:
        G = np.eye(4)[[0, 1, 3, 2]].astype(complex)
    for i, k in enumerate(kinds):
        G = np.kron(rot(k, angles[1 + 2 * i]), np.eye(2)) @ G
        G = np.kron(np.eye(2), rot(k, angles[2 + 2 * i])) @ G
    return G


def records(ent, kinds, angles):
    rec = np.zeros((1 + 2 * len(kinds), 4), np.float32)
    rec[0] = [np.cos(angles[0]), np.sin(angles[0]), ent, 0] if ent == K_CP else [1, 0, ent, 0]
    for i, k in enumerate(kinds):
        for h in (0, 1):
            a = angles[1 + 2 * i + h]
            rec[1 + 2 * i + h] = [np.cos(a / 2), np.sin(a / 2), k, 0]
    return rec


@pytest.mark.parametrize('ent', [K_CP, K_CZ, K_CX])
@pytest.mark.parametrize('kinds', [(K_RX, K_RY, K_RZ), (K_RX, K_RZ), (K_RZ, K_RY, K_RX), (K_RY,), (), (K_RZ, K_RX, K_RZ, K_RY)])
def test_block_matrix_and_gradient(shim, ent, kinds):
    rng = np.random.default_rng(len(kinds) * 7 + ent)
    n_rot = len(kinds)
    for _ in range(5):
        angles = rng.uniform(-4, 7, 1 + 2 * n_rot)
        rec = records(ent, kinds, angles)
        G = np.zeros(32, np.float32)
        shim.dm_block_matrix(rec.ctypes.data_as(C.c_void_p), n_rot, G.ctypes.data_as(C.c_void_p))
        Gc = (G[0::2] + 1j * G[1::2]).reshape(4, 4)
        ref = block(ent, kinds, angles)
        assert np.abs(Gc - ref).max() < 2e-6
        # gradient of the linear functional L = 2 Re sum_ij G_ij M_ij (what the environment M of the sweep defines)
        M = rng.normal(size=(4, 4)) + 1j * rng.normal(size=(4, 4))
        Mf = np.stack([M.real, M.imag], -1).astype(np.float32).reshape(-1)
        grad = np.zeros(1 + 2 * n_rot, np.float32)
        shim.dm_block_gradient(rec.ctypes.data_as(C.c_void_p), n_rot, G.ctypes.data_as(C.c_void_p), Mf.ctypes.data_as(C.c_void_p),
                               grad.ctypes.data_as(C.c_void_p))
        fd = np.zeros_like(angles)
        for k in range(len(angles)):
            h = 1e-5
            ap, am = angles.copy(), angles.copy()
            ap[k] += h
            am[k] -= h
            fd[k] = (2 * np.real((block(ent, kinds, ap) * M).sum()) - 2 * np.real((block(ent, kinds, am) * M).sum())) / (2 * h)
        if ent != K_CP:
            fd[0] = 0
        assert np.abs(grad - fd).max() < 2e-4 * max(1.0, np.abs(fd).max()), (grad, fd)


@pytest.mark.parametrize('ent', [K_CP, K_CZ, K_CX])
@pytest.mark.parametrize('fold', [(1, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize('kinds', [(K_RX, K_RY, K_RZ), (K_RX, K_RZ), ()])
def test_block_with_folded_surface_gates(shim, ent, fold, kinds):
    """The first block that touches a qubit carries that qubit's surface gate Rz(t2) Rx(t1) Rz(t0) (main.py:122-124) on
    its input side: G' = G (S_up (x) S_dn).  Matrix against NumPy, all gradients (block angles and the folded surface
    angles) against finite differences of L = 2 Re sum_ij G'_ij M_ij."""
    rng = np.random.default_rng(11 * len(kinds) + 3 * ent + fold[0] + 2 * fold[1])
    n_rot = len(kinds)

    def full(angles, sa):
        s_up = rot(K_RZ, sa[0, 2]) @ rot(K_RX, sa[0, 1]) @ rot(K_RZ, sa[0, 0]) if fold[0] else np.eye(2)
        s_dn = rot(K_RZ, sa[1, 2]) @ rot(K_RX, sa[1, 1]) @ rot(K_RZ, sa[1, 0]) if fold[1] else np.eye(2)
        return block(ent, kinds, angles) @ np.kron(s_up, s_dn)
    for _ in range(4):
        angles = rng.uniform(-4, 7, 1 + 2 * n_rot)
        sa = rng.uniform(-4, 7, (2, 3))
        rec = records(ent, kinds, angles)
        surf = np.stack([np.cos(sa / 2), np.sin(sa / 2)], -1).astype(np.float32).reshape(-1)
        G = np.zeros(32, np.float32)
        shim.dm_block_matrix_folded(rec.ctypes.data_as(C.c_void_p), n_rot, fold[0], fold[1], surf.ctypes.data_as(C.c_void_p), G.ctypes.data_as(C.c_void_p))
        Gc = (G[0::2] + 1j * G[1::2]).reshape(4, 4)
        assert np.abs(Gc - full(angles, sa)).max() < 3e-6
        M = rng.normal(size=(4, 4)) + 1j * rng.normal(size=(4, 4))
        Mf = np.stack([M.real, M.imag], -1).astype(np.float32).reshape(-1)
        grad, gu, gd = np.zeros(1 + 2 * n_rot, np.float32), np.full(3, 7.0, np.float32), np.full(3, 7.0, np.float32)
        shim.dm_block_gradient_folded(rec.ctypes.data_as(C.c_void_p), n_rot, fold[0], fold[1], surf.ctypes.data_as(C.c_void_p),
                                      G.ctypes.data_as(C.c_void_p), Mf.ctypes.data_as(C.c_void_p), grad.ctypes.data_as(C.c_void_p),
                                      gu.ctypes.data_as(C.c_void_p), gd.ctypes.data_as(C.c_void_p))
        L = lambda a, q: 2 * np.real((full(a, q) * M).sum())  # noqa: E731
        h = 1e-5
        fd = np.zeros_like(angles)
        for k in range(len(angles)):
            ap, am = angles.copy(), angles.copy()
            ap[k] += h
            am[k] -= h
            fd[k] = (L(ap, sa) - L(am, sa)) / (2 * h)
        if ent != K_CP:
            fd[0] = 0
        scale = max(1.0, np.abs(fd).max())
        assert np.abs(grad - fd).max() < 3e-4 * scale, (grad, fd)
        for side, got in ((0, gu), (1, gd)):
            if not fold[side]:
                assert (got == 7.0).all()                  # an unfolded side is not written
                continue
            for k in range(3):
                qp, qm = sa.copy(), sa.copy()
                qp[side, k] += h
                qm[side, k] -= h
                want = (L(angles, qp) - L(angles, qm)) / (2 * h)
                assert abs(got[k] - want) < 3e-4 * scale, (side, k, got, want)


@pytest.mark.parametrize('fold', [(0, 0), (1, 1), (0, 1)])
def test_compile_time_rotation_string_equals_run_time_path(shim, fold):
    """The walks specialised on 'xyz' + CP (template argument ROT = 8451, what the ahead-of-time dense kernels use) give
    bit for bit what the run-time-kind path gives on the same records."""
    rng = np.random.default_rng(3 + fold[0] + 2 * fold[1])
    kinds = (K_RX, K_RY, K_RZ)
    for _ in range(4):
        angles = rng.uniform(-4, 7, 7)
        sa = rng.uniform(-4, 7, (2, 3))
        rec = records(K_CP, kinds, angles)
        surf = np.stack([np.cos(sa / 2), np.sin(sa / 2)], -1).astype(np.float32).reshape(-1)
        G0, G1 = np.zeros(32, np.float32), np.zeros(32, np.float32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        shim.dm_block_matrix_folded(p(rec), 3, fold[0], fold[1], p(surf), p(G0))
        shim.dm_block_matrix_xyz(p(rec), fold[0], fold[1], p(surf), p(G1))
        assert (G0 == G1).all()
        M = (rng.normal(size=(4, 4, 2))).astype(np.float32).reshape(-1)
        out = []
        for fn, extra in ((shim.dm_block_gradient_folded, (3,)), (shim.dm_block_gradient_xyz, ())):
            grad, gu, gd = np.zeros(7, np.float32), np.zeros(3, np.float32), np.zeros(3, np.float32)
            fn(p(rec), *extra, fold[0], fold[1], p(surf), p(G0), p(M), p(grad), p(gu), p(gd))
            out.append(np.concatenate([grad, gu, gd]))
        assert (out[0] == out[1]).all()
