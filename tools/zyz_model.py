#!/usr/bin/env python3
"""Executable specification (NumPy, float64) of the ZYZ form of the entangling blocks that cpflow_amd/csrc/cpf_zyz.h runs -
NOT product code: it restates, phase by phase, what the kernel's build / sweep / chain phases compute (cpf_zyz_math.h is the
same algebra in C++) and checks it on small random circuits against the gate-by-gate unitary and central differences.

A block is G = (A_up (x) A_dn) E with A = R_m .. R_1 a product of rotations (main.py:43-46, 77-80; gates.py:22-58) and
E = CP(phi) or CZ.  Every A is an SU(2) element [[p, -conj q], [q, conj p]], and

    A = diag(l0, l1) . sc [[1, -t], [t, 1]] . diag(1, r1) . F^f          F = Ry(pi) = [[0, -1], [1, 0]]

with f = [|q| > |p|] (then A F^-1 is factored instead, so that t = tan <= 1 always), sc = cos, l0 = p / |p|, l1 = q / |q|,
r1 = conj(l0 l1): no square roots, no Euler angles.  Flips F commute with every Ry stage and turn a diagonal factor into the
same factor with its two entries exchanged, so they are pushed back through all earlier blocks onto the initial state (a signed
permutation of the product state of the surface gates): the "held" state of the sweeps is  (pending flips) x (true state).

Forward, per block and group of four amplitudes of the pair:  x_ab <- d_ab x_ab  (ONE merged phase per amplitude: the left
factors of the previous blocks on the two qubits, this block's right factors and the entangler's phase, the real scale sc_u sc_d
folded in and the global phase divided out so that d_00 is real), then the two unnormalised real stages [[1, -t], [t, 1]].
Backward: the same in reverse with conj(d); per qubit three generator sums P_k = sum Im(conj(w) sigma_k psi) right after the Y
stage and one sum over the |11> amplitudes before it.  The gradient of a rotation angle is P (rotated into the frame right
after A by the left diagonal) dotted with the generator carried back through the later rotations; the entangler's angle uses
the |11> sum and - when flips moved its phase to another amplitude of the pair - the z sums of the previous blocks on the two
qubits (sum over everything of Im(conj(w) psi) is zero for every loss kind, so the other three amplitudes follow).

The dense-layout kernel additionally applies the real scale once per LAYER instead of once per block (unit-modulus phases, the
layer's last block carries the product, every block's sums divided by the square of the factor psi and the cotangent are off by
there): that bookkeeping is modelled, through the C++ header, by tests/test_zyz_math.py::test_real_scale_once_per_layer.

    python tools/zyz_model.py        # prints the worst deviations; exits non-zero above 1e-9 / 1e-6
"""
import sys

import numpy as np

X = np.array([[0, 1], [1, 0]], complex)
Y = np.array([[0, -1j], [1j, 0]], complex)
Z = np.diag([1.0 + 0j, -1.0])
PAULI = (X, Y, Z)
FLIP = np.array([[0, -1.0], [1.0, 0]], complex)          # Ry(pi)
AX = {'x': 0, 'y': 1, 'z': 2}


def rot(sig, a):
    return np.cos(a / 2) * np.eye(2) - 1j * np.sin(a / 2) * sig


def embed(n, q, m):
    ops = [np.eye(2)] * n
    ops[q] = m
    out = ops[0]
    for o in ops[1:]:
        out = np.kron(out, o)
    return out


def bit(n, r, q):
    return (r >> (n - 1 - q)) & 1


def reference_unitary(n, placements, theta, rots='xyz', ent='cp'):
    nb = 2 * len(rots) + (1 if ent == 'cp' else 0)
    U = np.eye(1 << n, dtype=complex)
    for q in range(n):
        U = embed(n, q, rot(Z, theta[3 * q + 2]) @ rot(X, theta[3 * q + 1]) @ rot(Z, theta[3 * q])) @ U
    for b, (up, dn) in enumerate(placements):
        t = theta[3 * n + nb * b:3 * n + nb * (b + 1)]
        lam = np.exp(1j * t[-1]) if ent == 'cp' else -1.0
        d = np.array([lam if bit(n, r, up) and bit(n, r, dn) else 1.0 for r in range(1 << n)])
        U = d[:, None] * U
        for i, g in enumerate(rots):
            U = embed(n, up, rot(PAULI[AX[g]], t[2 * i])) @ embed(n, dn, rot(PAULI[AX[g]], t[2 * i + 1])) @ U
    return U


def loss_of(U, V):
    return 1 - abs(np.sum(U * np.conj(V))) ** 2 / U.shape[0] ** 2


# ---- build phase, per (block, side) ---------------------------------------------------------------------------------
def side_build(angles, rots, use_flips=True):
    """angles: this side's rotation angles in program order.  Returns dict(t, sc, l0, l1, r1, f)."""
    p, q = 1.0 + 0j, 0.0 + 0j
    for a, g in zip(angles, rots):
        c, s = np.cos(a / 2), np.sin(a / 2)
        if g == 'x':
            p, q = c * p - 1j * s * q, -1j * s * p + c * q
        elif g == 'y':
            p, q = c * p - s * q, s * p + c * q
        else:
            p, q = (c - 1j * s) * p, (c + 1j * s) * q
    f = int(use_flips and abs(q) > abs(p))
    if f:
        p, q = np.conj(q), -np.conj(p)                # A F^-1
    sc = abs(p)
    l0 = p / sc
    ok = abs(q) ** 2 > 1e-30
    l1 = q / abs(q) if ok else 1.0 + 0j
    t = abs(q) / sc if ok else 0.0
    return dict(t=t, sc=sc, l0=l0, l1=l1, r1=np.conj(l0 * l1), f=f)


def so3_inv(axis, c, s, P):
    """so3(R)^T P for R = exp(-i a sigma_axis / 2) with (c, s) = (cos a/2, sin a/2): the two other components rotate by -a."""
    ca, sa = c * c - s * s, 2 * c * s
    i, j = (axis + 1) % 3, (axis + 2) % 3              # (y, z) for x; (z, x) for y; (x, y) for z
    out = P.copy()
    out[i] = ca * P[i] + sa * P[j]
    out[j] = -sa * P[i] + ca * P[j]
    return out


def chain_side(P_held, m, side, angles, rots):
    """gradients of this side's rotation angles from the held generator sums right after the Y stage"""
    P = np.array(P_held, float)
    if m:
        P[0], P[2] = -P[0], -P[2]                      # pending flips of later blocks: sigma_x, sigma_z change sign
    u = side['l0'] * np.conj(side['l1'])               # frame right after A: rotation about z by alpha, exp(-i alpha) = u
    ca, sa = u.real, -u.imag
    P = np.array([ca * P[0] - sa * P[1], sa * P[0] + ca * P[1], P[2]])
    g = np.zeros(len(rots))
    for i in range(len(rots) - 1, -1, -1):
        ax = AX[rots[i]]
        g[i] = P[ax]
        P = so3_inv(ax, np.cos(angles[i] / 2), np.sin(angles[i] / 2), P)
    return g


def zyz_forward_backward(n, placements, theta, V, rots='xyz', ent='cp', use_flips=True):
    """Loss and the full gradient, computed the way the kernel does."""
    N, K, nr = 1 << n, len(placements), len(rots)
    nb = 2 * nr + (1 if ent == 'cp' else 0)
    blk = lambda b: theta[3 * n + nb * b:3 * n + nb * (b + 1)]
    side_angles = lambda b, s: blk(b)[s:2 * nr:2]
    # ---- build: sides, flip parities, merged diagonals
    sides = [[side_build(side_angles(b, s), rots, use_flips) for s in (0, 1)] for b in range(K)]
    m = [[0, 0] for _ in range(K)]                     # pending flips of LATER blocks on the side's qubit
    prev = [[None, None] for _ in range(K)]            # previous (block, side) on the same qubit
    last, pend = {}, [0] * n
    for b, pl in enumerate(placements):
        for s, q in enumerate(pl):
            prev[b][s] = last.get(q)
            last[q] = (b, s)
    for b in range(K - 1, -1, -1):
        for s, q in enumerate(placements[b]):
            m[b][s] = pend[q]
            pend[q] ^= sides[b][s]['f']
    m_surf = pend                                      # all flips on the qubit reach the surface state
    swap = lambda pair, mm: (pair[1], pair[0]) if mm else pair
    tables = []
    for b, (up, dn) in enumerate(placements):
        e = []
        for s in (0, 1):
            sd = sides[b][s]
            pl = (1.0, 1.0) if prev[b][s] is None else swap((sides[prev[b][s][0]][prev[b][s][1]]['l0'], sides[prev[b][s][0]][prev[b][s][1]]['l1']), m[prev[b][s][0]][prev[b][s][1]])
            r = swap((1.0, sd['r1']), m[b][s])
            e.append((pl[0] * r[0], pl[1] * r[1]))
        ca, cb = 1 ^ m[b][0] ^ sides[b][0]['f'], 1 ^ m[b][1] ^ sides[b][1]['f']       # where the entangler's phase sits in the held frame
        lam = np.exp(1j * blk(b)[-1]) if ent == 'cp' else -1.0
        d = np.array([[e[0][a] * e[1][c] * (lam if (a, c) == (ca, cb) else 1.0) for c in (0, 1)] for a in (0, 1)])
        d = d * np.conj(d[0, 0]) * sides[b][0]['sc'] * sides[b][1]['sc']               # global phase out, real scale in
        assert abs(d[0, 0].imag) < 1e-12
        tables.append((d, sides[b][0]['t'], sides[b][1]['t'], ca, cb))
    lfin = [(1.0, 1.0) if q not in last else (sides[last[q][0]][last[q][1]]['l0'], sides[last[q][0]][last[q][1]]['l1']) for q in range(n)]
    # ---- forward (held frame)
    U = np.ones((N, N), complex)
    for q in range(n):                                 # product state of the surface gates, flips applied
        S = rot(Z, theta[3 * q + 2]) @ rot(X, theta[3 * q + 1]) @ rot(Z, theta[3 * q])
        if m_surf[q]:
            S = FLIP @ S
        for r in range(N):
            for c in range(N):
                U[r, c] *= S[bit(n, r, q), bit(n, c, q)]
    psi_S = U.copy()

    def dvec(b):
        d, up, dn = tables[b][0], *placements[b]
        return np.array([d[bit(n, r, up), bit(n, r, dn)] for r in range(N)])

    def ystage(b, sign):
        up, dn = placements[b]
        tu, td = tables[b][1], tables[b][2]
        return embed(n, up, np.array([[1, -sign * tu], [sign * tu, 1]])) @ embed(n, dn, np.array([[1, -sign * td], [sign * td, 1]]))

    for b in range(K):
        U = ystage(b, 1) @ (dvec(b)[:, None] * U)
    dfin = np.array([np.prod([lfin[q][bit(n, r, q)] for q in range(n)]) for r in range(N)])
    Vd = np.conj(dfin)[:, None] * V                    # the final diagonal meets the target, not the state
    tr = np.sum(U * np.conj(Vd))
    loss = 1 - abs(tr) ** 2 / N ** 2
    W = -(tr / N ** 2) * Vd
    # ---- backward
    Pxyz = np.zeros((K, 2, 3))
    S11 = np.zeros(K)
    gen = lambda q, k, W, U: np.sum(np.conj(W) * (embed(n, q, PAULI[k]) @ U)).imag
    for b in range(K - 1, -1, -1):
        up, dn = placements[b]
        for s, q in enumerate((up, dn)):
            Pxyz[b, s] = [gen(q, k, W, U) for k in range(3)]
        U, W = ystage(b, -1) @ U, ystage(b, -1) @ W
        mask = np.array([bit(n, r, up) & bit(n, r, dn) for r in range(N)], bool)
        # both matrices carry 1 / (sc_u sc_d) here (the unnormalised stage is undone before the scaled conj(d) is applied):
        # the chain phase multiplies the sum by d_00^2
        S11[b] = np.sum(np.conj(W[mask]) * U[mask]).imag * tables[b][0][0, 0].real ** 2
        U, W = np.conj(dvec(b))[:, None] * U, np.conj(dvec(b))[:, None] * W
    Psurf = np.array([[gen(q, k, W, U) for k in range(3)] for q in range(n)])
    unapply_err = np.abs(U - psi_S).max()
    # ---- chain
    grad = np.zeros_like(theta)
    for b in range(K):
        for s in (0, 1):
            grad[3 * n + nb * b + s:3 * n + nb * b + 2 * nr:2] = chain_side(Pxyz[b, s], m[b][s], sides[b][s], side_angles(b, s), rots)
        if ent == 'cp':
            pz = [Psurf[placements[b][s], 2] if prev[b][s] is None else Pxyz[prev[b][s][0], prev[b][s][1], 2] for s in (0, 1)]
            ca, cb = tables[b][3], tables[b][4]
            T = {(1, 1): S11[b], (1, 0): -pz[0] / 2 - S11[b], (0, 1): -pz[1] / 2 - S11[b], (0, 0): pz[0] / 2 + pz[1] / 2 + S11[b]}
            grad[3 * n + nb * b + nb - 1] = -2 * T[(ca, cb)]
    for q in range(n):
        P = Psurf[q].copy()
        if m_surf[q]:
            P[0], P[2] = -P[0], -P[2]
        t0, t1, t2 = theta[3 * q:3 * q + 3]
        grad[3 * q + 2] = P[2]
        P = so3_inv(2, np.cos(t2 / 2), np.sin(t2 / 2), P)
        grad[3 * q + 1] = P[0]
        P = so3_inv(0, np.cos(t1 / 2), np.sin(t1 / 2), P)
        grad[3 * q] = P[2]
    return loss, grad, unapply_err, sum(sd['f'] for row in sides for sd in row)


def main():
    rng = np.random.default_rng(0)
    worst_l = worst_g = worst_u = 0.0
    flips = 0
    for case in range(40):
        n = int(rng.integers(2, 5))
        rots = ['xyz', 'xz', 'zyx', 'y', 'xyz', 'zxzy'][case % 6]
        ent = 'cz' if case % 5 == 3 else 'cp'
        nb = 2 * len(rots) + (1 if ent == 'cp' else 0)
        pairs = [(i, j) for i in range(n) for j in range(n) if i != j]
        placements = [pairs[int(rng.integers(len(pairs)))] for _ in range(int(rng.integers(1, 7)))]
        theta = rng.uniform(-2 * np.pi, 2 * np.pi, 3 * n + nb * len(placements))
        if case % 3 == 0:
            theta[3 * n:3 * n + nb] = 0.0                  # all-diagonal first block (identity rotations)
        if case % 4 == 1 and 'y' in rots:
            theta[3 * n + 2 * rots.index('y')] = np.pi     # an exact flip
        if case % 7 == 2:
            theta[3 * n:] = rng.choice([0.0, np.pi, -np.pi, 2 * np.pi], len(theta) - 3 * n)     # degenerate everywhere
        a = rng.normal(size=(1 << n, 1 << n)) + 1j * rng.normal(size=(1 << n, 1 << n))
        V = np.linalg.qr(a)[0]
        for use_flips in (True, False):
            if not use_flips and case % 7 == 2:
                continue                                    # without flips t is unbounded at an exact flip
            loss, grad, uerr, nf = zyz_forward_backward(n, placements, theta, V, rots, ent, use_flips)
            flips += nf
            ref = loss_of(reference_unitary(n, placements, theta, rots, ent), V)
            worst_l = max(worst_l, abs(loss - ref))
            worst_u = max(worst_u, uerr)
            for j in range(len(theta)):
                e = np.zeros_like(theta)
                e[j] = 1e-6
                fd = (loss_of(reference_unitary(n, placements, theta + e, rots, ent), V) - loss_of(reference_unitary(n, placements, theta - e, rots, ent), V)) / 2e-6
                worst_g = max(worst_g, abs(fd - grad[j]))
    print(f'loss: worst deviation {worst_l:.2e};  un-applied state vs the surface product state {worst_u:.2e};  '
          f'gradient against central differences: worst {worst_g:.2e}  ({flips} flipped sides)')
    return 0 if worst_l < 1e-9 and worst_g < 1e-6 and worst_u < 1e-9 else 1


if __name__ == '__main__':
    sys.exit(main())
