#!/usr/bin/env python3
"""Executable specification (NumPy, float64) of the ZYZ form of the entangling blocks priced in DESIGN.md section 8, item 3 -
NOT product code: it checks the algebra a kernel would run, on small random circuits, against the oracle's gate-by-gate build
and against finite differences.

A block is G = (A_up (x) A_dn) CP(phi) with A = Rz(c) Ry(b) Rx(a) (main.py:43-46, 77-80; gates.py:22-58).  Every A is a
general SU(2), so A = Rz(f1) Ry(th) Rz(f2) with cos(th/2) = |A00|, sin(th/2) = |A10| (both >= 0: tan(th/4) <= 1, three real
shears as well conditioned as the rotation) and diagonal unit factors x = exp(-i f1/2), y = exp(-i f2/2) with x y = A00/|A00|, conj(x) y = A10/|A10|.
Forward: per block  psi <- L Y R psi  with R = diag(y_u^-+ (x) y_d^-+) CP, Y = Ry(th_u) (x) Ry(th_d), L = diag(x_u^-+ (x) x_d^-+);
consecutive diagonal factors (L of one block, R of the next) merge into one phase per amplitude.
Backward: with psi and the cotangent w right AFTER the Y stage, per qubit P_k = sum Im(conj(w) sigma_k psi), k = x, y, z
(and the CP sum on the |11> amplitudes before Y); the gradient of (a, b, c) is that 3-vector rotated into the frames of the
three generators:  dL/dc ~ P_z' , dL/db ~ (R_z(c) e_y) . P' , dL/da ~ (R_z(c) R_y(b) e_x) . P'  with P' = R_z(f1) P the vector
after the left diagonal - no Euler-angle Jacobian (singular at th = 0) appears.

    python tools/zyz_model.py        # prints the worst deviations; exits non-zero above 1e-9 / 1e-6
"""
import sys

import numpy as np

X = np.array([[0, 1], [1, 0]], complex)
Y = np.array([[0, -1j], [1j, 0]], complex)
Z = np.diag([1.0 + 0j, -1.0])
PAULI = (X, Y, Z)


def rot(sig, a):
    return np.cos(a / 2) * np.eye(2) - 1j * np.sin(a / 2) * sig


def zyz(a, b, c):
    """(cos(th/2), sin(th/2), x, y) with Rz(c) Ry(b) Rx(a) = diag(x, conj x) Ry(th) diag(y, conj y)."""
    A = rot(Z, c) @ rot(Y, b) @ rot(X, a)
    p, q = A[0, 0], A[1, 0]
    ct, st = abs(p), abs(q)
    ph = p / ct if ct > 1e-30 else 1.0
    qh = q / st if st > 1e-30 else 1.0             # th -> 0: the split of the diagonal is arbitrary, any unit number serves
    y = np.sqrt(ph * qh + 0j)                      # x y = ph, conj(x) y = qh  ->  y^2 = ph qh
    x = ph * np.conj(y)
    return ct, st, x, y, A


def so3(sig_axis, ang):
    """rotation of the generator 3-vector under conjugation by exp(-i ang/2 sigma_axis): R sigma_k R^+ = sum_j M[j, k] sigma_j"""
    R = rot(PAULI[sig_axis], ang)
    M = np.zeros((3, 3))
    for k in range(3):
        C = R @ PAULI[k] @ R.conj().T
        for j in range(3):
            M[j, k] = 0.5 * np.trace(PAULI[j] @ C).real
    return M


def embed(n, q, m):
    ops = [np.eye(2)] * n
    ops[q] = m
    out = ops[0]
    for o in ops[1:]:
        out = np.kron(out, o)
    return out


def cp_diag(n, up, dn, phi):
    d = np.ones(1 << n, complex)
    for r in range(1 << n):
        if (r >> (n - 1 - up)) & 1 and (r >> (n - 1 - dn)) & 1:
            d[r] = np.exp(1j * phi)
    return d


def reference_unitary(n, placements, theta):
    U = np.eye(1 << n, dtype=complex)
    for q in range(n):
        U = embed(n, q, rot(Z, theta[3 * q + 2]) @ rot(X, theta[3 * q + 1]) @ rot(Z, theta[3 * q])) @ U
    for b, (up, dn) in enumerate(placements):
        t = theta[3 * n + 7 * b:3 * n + 7 * b + 7]
        U = np.diag(cp_diag(n, up, dn, t[6])) @ U
        U = embed(n, up, rot(X, t[0])) @ embed(n, dn, rot(X, t[1])) @ U
        U = embed(n, up, rot(Y, t[2])) @ embed(n, dn, rot(Y, t[3])) @ U
        U = embed(n, up, rot(Z, t[4])) @ embed(n, dn, rot(Z, t[5])) @ U
    return U


def loss_of(U, V):
    return 1 - abs(np.sum(U * np.conj(V))) ** 2 / U.shape[0] ** 2


def zyz_forward_backward(n, placements, theta, V):
    """The ZYZ sweeps: returns (loss, gradient of the block angles) computed the way the kernel would."""
    N = 1 << n
    U = np.eye(N, dtype=complex)
    for q in range(n):            # surface gates: kept gate by gate here (Rz Rx Rz is ZYZ with Rx = Rz(-pi/2) Ry Rz(pi/2))
        U = embed(n, q, rot(Z, theta[3 * q + 2]) @ rot(X, theta[3 * q + 1]) @ rot(Z, theta[3 * q])) @ U
    stages, pending = [], np.ones(N, complex)          # pending: merged diagonal carried into the next block's R
    for b, (up, dn) in enumerate(placements):
        t = theta[3 * n + 7 * b:3 * n + 7 * b + 7]
        zu, zd = zyz(t[0], t[2], t[4]), zyz(t[1], t[3], t[5])
        Rd = np.diag(embed(n, up, np.diag([zu[3], np.conj(zu[3])])) @ embed(n, dn, np.diag([zd[3], np.conj(zd[3])]))) * cp_diag(n, up, dn, t[6])
        Ld = np.diag(embed(n, up, np.diag([zu[2], np.conj(zu[2])])) @ embed(n, dn, np.diag([zd[2], np.conj(zd[2])])))
        Ymat = embed(n, up, np.array([[zu[0], -zu[1]], [zu[1], zu[0]]])) @ embed(n, dn, np.array([[zd[0], -zd[1]], [zd[1], zd[0]]]))
        stages.append((pending * Rd, Ymat, up, dn, t, zu, zd))
        pending = Ld
    states = []
    for D, Ymat, *_ in stages:
        U = D[:, None] * U                      # ONE phase per amplitude: previous block's left factor, CP, this block's right factor
        before_y = U
        U = Ymat @ U
        states.append((before_y, U))
    U = pending[:, None] * U
    tr = np.sum(U * np.conj(V))
    loss = 1 - abs(tr) ** 2 / N ** 2
    W = -(tr / N ** 2) * V                      # dL = 2 Re sum conj(W) dU
    W = np.conj(pending)[:, None] * W           # cotangent right after the last Y stage
    grad = np.zeros(7 * len(placements))
    for b in range(len(stages) - 1, -1, -1):
        D, Ymat, up, dn, t, zu, zd = stages[b]
        before_y, after_y = states[b]
        for side, (q, zz) in enumerate(((up, zu), (dn, zd))):
            P = np.array([2 * np.sum(np.conj(W) * (embed(n, q, -0.5j * s) @ after_y)).real for s in PAULI])     # generator sums after Y
            f1 = -2 * np.angle(zz[2])
            Pp = so3(2, f1) @ P                  # ... seen after the left diagonal Rz(f1): the frame right after A
            a, bb, c = t[0 + side], t[2 + side], t[4 + side]
            grad[7 * b + 4 + side] = Pp[2]
            grad[7 * b + 2 + side] = (so3(2, c) @ np.array([0, 1.0, 0])) @ Pp
            grad[7 * b + 0 + side] = (so3(2, c) @ so3(1, bb) @ np.array([1.0, 0, 0])) @ Pp
        W = Ymat.conj().T @ W                   # un-apply the Y stage on the cotangent; psi before Y was kept here (a kernel un-applies it)
        mask = cp_diag(n, up, dn, np.pi).real < 0
        grad[7 * b + 6] = 2 * np.sum(np.conj(W[mask]) * (1j * before_y[mask])).real
        W = np.conj(D)[:, None] * W
    return loss, grad


def main():
    rng = np.random.default_rng(0)
    worst_u = worst_g = 0.0
    for case in range(12):
        n = int(rng.integers(2, 5))
        pairs = [(i, j) for i in range(n) for j in range(n) if i != j]
        placements = [pairs[int(rng.integers(len(pairs)))] for _ in range(int(rng.integers(1, 6)))]
        theta = rng.uniform(-2 * np.pi, 2 * np.pi, 3 * n + 7 * len(placements))
        if case % 3 == 0:
            theta[3 * n:3 * n + 4] = 0.0                   # th = 0 on the first block: the Euler angles' gimbal lock
        if case % 4 == 1:
            theta[3 * n + 2] = np.pi
        a = rng.normal(size=(1 << n, 1 << n)) + 1j * rng.normal(size=(1 << n, 1 << n))
        V = np.linalg.qr(a)[0]
        for b, (up, dn) in enumerate(placements):          # ZYZ factors reproduce every single-qubit part
            t = theta[3 * n + 7 * b:3 * n + 7 * b + 7]
            for side in range(2):
                ct, st, x, y, A = zyz(t[side], t[2 + side], t[4 + side])
                assert ct >= 0 and st >= 0 and st / (1 + ct) <= 1 + 1e-12
                B = np.diag([x, np.conj(x)]) @ np.array([[ct, -st], [st, ct]]) @ np.diag([y, np.conj(y)])
                worst_u = max(worst_u, np.abs(A - B).max())
        loss, grad = zyz_forward_backward(n, placements, theta, V)
        ref = loss_of(reference_unitary(n, placements, theta), V)
        worst_u = max(worst_u, abs(loss - ref))
        for j in range(len(grad)):
            e = np.zeros_like(theta)
            e[3 * n + j] = 1e-6
            fd = (loss_of(reference_unitary(n, placements, theta + e), V) - loss_of(reference_unitary(n, placements, theta - e), V)) / 2e-6
            worst_g = max(worst_g, abs(fd - grad[j]))
    print(f'ZYZ factors / loss: worst deviation {worst_u:.2e};  gradient against central differences: worst {worst_g:.2e}')
    return 0 if worst_u < 1e-9 and worst_g < 1e-6 else 1


if __name__ == '__main__':
    sys.exit(main())
