#!/usr/bin/env python3
"""Executable specification (NumPy, float64) of the TWO-ENDED ZYZ-form sweep that cpflow_amd/csrc/cpf_zyz3.h runs on 3 qubits -
NOT product code.  The per-block algebra (sides, flips, merged phases, chain rule) is tools/zyz_model.py's; what this file
restates is the part that is new in the two-ended kernel:

  * U = H2 H1 with H1 = blocks 0 .. M-1 on the (flipped) product state of the surface gates, H2 = blocks M .. K-1.
    Wave A builds psi(s_b) (the held state right after the real stages of block b; s_-1 = the surface state) for b < M and
    keeps every one of them; wave B starts from Vd = conj(dfin) V = W(s_{K-1}) and un-applies blocks K-1 .. M, keeping W(s_b).
    T = sum conj(W(s_{M-1})) psi(s_{M-1}).
  * Gradient data are read SLICE BY SLICE from one live matrix and one stored matrix: at slice s_b
        u_l   = Im(conj(w_l) psi_l)                      -> Pz of block b's two qubits (signed by the qubit's bit) and the |11>
                                                            sum of block b + 1's pair (no scale factor: both matrices carry
                                                            conj(d) of block b + 1 there, whose modulus is the block's real scale)
        conj(w_l) psi_{l ^ q}                            -> Px (imaginary part) and Py (real part, signed) of qubit q
    wave A: w = f W live (un-applied block by block), psi read back;  wave B: psi' = conj(f) psi live (applied block by block),
    W read back at the lane and at its partner lanes (roles swapped: the sign of Py changes).

    python tools/zyz3_model.py        # worst deviations against tools/zyz_model.py and central differences
"""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit('/', 1)[0])
import zyz_model as zm  # noqa: E402


def two_ended(n, placements, theta, V, M, rots='xyz', ent='cp'):
    N, K, nr = 1 << n, len(placements), len(rots)
    nb = 2 * nr + (1 if ent == 'cp' else 0)
    blk = lambda b: theta[3 * n + nb * b:3 * n + nb * (b + 1)]
    side_angles = lambda b, s: blk(b)[s:2 * nr:2]
    sides = [[zm.side_build(side_angles(b, s), rots) for s in (0, 1)] for b in range(K)]
    m = [[0, 0] for _ in range(K)]
    prev = [[None, None] for _ in range(K)]
    last, pend = {}, [0] * n
    for b, pl in enumerate(placements):
        for s, q in enumerate(pl):
            prev[b][s] = last.get(q)
            last[q] = (b, s)
    for b in range(K - 1, -1, -1):
        for s, q in enumerate(placements[b]):
            m[b][s] = pend[q]
            pend[q] ^= sides[b][s]['f']
    m_surf = pend
    swap = lambda pair, mm: (pair[1], pair[0]) if mm else pair
    tables = []
    for b, (up, dn) in enumerate(placements):
        e = []
        for s in (0, 1):
            sd = sides[b][s]
            if prev[b][s] is None:
                pl = (1.0, 1.0)
            else:
                pb, ps = prev[b][s]
                # the previous side's pending parity = this side's parity xor its own flip (later(prev) = later(this) + this)
                assert m[pb][ps] == m[b][s] ^ sd['f']
                pl = swap((sides[pb][ps]['l0'], sides[pb][ps]['l1']), m[b][s] ^ sd['f'])
            r = swap((1.0, sd['r1']), m[b][s])
            e.append((pl[0] * r[0], pl[1] * r[1]))
        ca, cb = 1 ^ m[b][0] ^ sides[b][0]['f'], 1 ^ m[b][1] ^ sides[b][1]['f']
        lam = np.exp(1j * blk(b)[-1]) if ent == 'cp' else -1.0
        d = np.array([[e[0][a] * e[1][c] * (lam if (a, c) == (ca, cb) else 1.0) for c in (0, 1)] for a in (0, 1)])
        d = d * np.conj(d[0, 0]) * sides[b][0]['sc'] * sides[b][1]['sc']
        tables.append((d, sides[b][0]['t'], sides[b][1]['t'], ca, cb))
    lfin = [(1.0, 1.0) if q not in last else (sides[last[q][0]][last[q][1]]['l0'], sides[last[q][0]][last[q][1]]['l1']) for q in range(n)]
    bit = zm.bit

    def dvec(b):
        d, up, dn = tables[b][0], *placements[b]
        return np.array([d[bit(n, r, up), bit(n, r, dn)] for r in range(N)])

    def stage(x, q, t):                                  # (a, b) <- (a - t b, b + t a) on qubit q: x += (+-t) x[partner]
        out = x.copy()
        for r in range(N):
            sg = 1.0 if bit(n, r, q) else -1.0
            out[r] = x[r] + sg * t * x[r ^ (1 << (n - 1 - q))]
        return out

    # ---- wave A forward
    psi = np.ones((N, N), complex)
    for q in range(n):
        S = zm.rot(zm.Z, theta[3 * q + 2]) @ zm.rot(zm.X, theta[3 * q + 1]) @ zm.rot(zm.Z, theta[3 * q])
        if m_surf[q]:
            S = zm.FLIP @ S
        for r in range(N):
            for c in range(N):
                psi[r, c] *= S[bit(n, r, q), bit(n, c, q)]
    st_psi = {-1: psi.copy()}
    for b in range(M):
        up, dn = placements[b]
        psi = dvec(b)[:, None] * psi
        psi = stage(stage(psi, up, tables[b][1]), dn, tables[b][2])
        st_psi[b] = psi.copy()
    # ---- wave B forward (un-applying from the target)
    dfin = np.array([np.prod([lfin[q][bit(n, r, q)] for q in range(n)]) for r in range(N)])
    W = np.conj(dfin)[:, None] * V
    st_W = {K - 1: W.copy()}
    for b in range(K - 1, M - 1, -1):
        up, dn = placements[b]
        W = stage(stage(W, dn, -tables[b][2]), up, -tables[b][1])
        W = np.conj(dvec(b))[:, None] * W
        st_W[b - 1] = W.copy()
    tr = np.sum(st_psi[M - 1] * np.conj(st_W[M - 1]))
    loss = 1 - abs(tr) ** 2 / N ** 2
    f = -tr / N ** 2

    # ---- slice sums: w = the TRUE cotangent's role (conj(w) multiplies), p = psi's role
    def partner(x, q):
        return x[[r ^ (1 << (n - 1 - q)) for r in range(N)], :]

    def sgn(q):
        return np.array([1.0 if bit(n, r, q) else -1.0 for r in range(N)])[:, None]

    def slice_sums_A(w, p, qubits, nxt):                 # wave A: w live, p stored (own + partner reads)
        u = (np.conj(w) * p).imag
        out = {}
        for q in qubits:
            z = np.conj(w) * partner(p, q)
            out[q] = np.array([z.imag.sum(), (sgn(q) * z.real).sum(), (-sgn(q) * u).sum()])
        s11 = None
        if nxt is not None:
            mask = np.array([bit(n, r, nxt[0]) & bit(n, r, nxt[1]) for r in range(N)], float)[:, None]
            s11 = (mask * u).sum()
        return out, s11

    def slice_sums_B(Wst, pp, qubits, nxt):              # wave B: pp = conj(f) psi live, W stored (own + partner reads)
        u = (np.conj(Wst) * pp).imag
        out = {}
        for q in qubits:
            z = np.conj(partner(Wst, q)) * pp
            out[q] = np.array([z.imag.sum(), (-sgn(q) * z.real).sum(), (-sgn(q) * u).sum()])
        s11 = None
        if nxt is not None:
            mask = np.array([bit(n, r, nxt[0]) & bit(n, r, nxt[1]) for r in range(N)], float)[:, None]
            s11 = (mask * u).sum()
        return out, s11

    Pxyz = np.zeros((K, 2, 3))
    S11 = np.zeros(K)
    # wave A backward
    w = f * st_W[M - 1]
    for b in range(M - 1, -2, -1):
        qubits = placements[b] if b >= 0 else tuple(range(n))
        nxt = placements[b + 1] if b + 1 < K else None
        out, s11 = slice_sums_A(w, st_psi[b], qubits, nxt)
        if b >= 0:
            Pxyz[b, 0], Pxyz[b, 1] = out[qubits[0]], out[qubits[1]]
            up, dn = placements[b]
            w = stage(stage(w, dn, -tables[b][2]), up, -tables[b][1])
            w = np.conj(dvec(b))[:, None] * w
        else:
            Psurf = np.array([out[q] for q in range(n)])
        if nxt is not None:
            S11[b + 1] = s11
    # wave B backward (forwards through its blocks)
    pp = np.conj(f) * st_psi[M - 1]
    for b in range(M, K):
        up, dn = placements[b]
        pp = dvec(b)[:, None] * pp
        pp = stage(stage(pp, up, tables[b][1]), dn, tables[b][2])
        nxt = placements[b + 1] if b + 1 < K else None
        out, s11 = slice_sums_B(st_W[b], pp, (up, dn), nxt)
        Pxyz[b, 0], Pxyz[b, 1] = out[up], out[dn]
        if nxt is not None:
            S11[b + 1] = s11
    # ---- chain (as tools/zyz_model.py, the |11> sums already carry the block's scale)
    grad = np.zeros_like(theta)
    for b in range(K):
        for s in (0, 1):
            grad[3 * n + nb * b + s:3 * n + nb * b + 2 * nr:2] = zm.chain_side(Pxyz[b, s], m[b][s], sides[b][s], side_angles(b, s), rots)
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
        P = zm.so3_inv(2, np.cos(t2 / 2), np.sin(t2 / 2), P)
        grad[3 * q + 1] = P[0]
        P = zm.so3_inv(0, np.cos(t1 / 2), np.sin(t1 / 2), P)
        grad[3 * q] = P[2]
    return loss, grad


def main():
    rng = np.random.default_rng(1)
    worst_l = worst_g = worst_fd = 0.0
    for case in range(30):
        n = 3
        rots = ['xyz', 'xz', 'zyx', 'y', 'xyz', 'yz'][case % 6]
        ent = 'cz' if case % 5 == 3 else 'cp'
        nb = 2 * len(rots) + (1 if ent == 'cp' else 0)
        layer = [[(0, 1), (1, 2)], [(0, 1), (0, 2), (1, 2)], [(2, 0), (1, 0)]][case % 3]
        K = int(rng.integers(2, 9))
        placements = [layer[b % len(layer)] for b in range(K)]
        theta = rng.uniform(-2 * np.pi, 2 * np.pi, 3 * n + nb * K)
        if case % 4 == 1:
            theta[3 * n:3 * n + nb] = 0.0
        if case % 7 == 2:
            theta[3 * n:] = rng.choice([0.0, np.pi, -np.pi, 2 * np.pi], len(theta) - 3 * n)
        a = rng.normal(size=(8, 8)) + 1j * rng.normal(size=(8, 8))
        V = np.linalg.qr(a)[0]
        ref_l, ref_g, _, _ = zm.zyz_forward_backward(n, placements, theta, V, rots, ent, True)
        for M in range(1, K):
            loss, grad = two_ended(n, placements, theta, V, M, rots, ent)
            worst_l = max(worst_l, abs(loss - ref_l))
            worst_g = max(worst_g, np.abs(grad - ref_g).max())
        for j in range(len(theta)):
            e = np.zeros_like(theta)
            e[j] = 1e-6
            fd = (zm.loss_of(zm.reference_unitary(n, placements, theta + e, rots, ent), V) - zm.loss_of(zm.reference_unitary(n, placements, theta - e, rots, ent), V)) / 2e-6
            worst_fd = max(worst_fd, abs(fd - grad[j]))
    print(f'two-ended against the one-ended model: loss {worst_l:.2e}, gradient {worst_g:.2e};  gradient against central differences {worst_fd:.2e}')
    return 0 if worst_l < 1e-12 and worst_g < 1e-11 and worst_fd < 1e-6 else 1


if __name__ == '__main__':
    sys.exit(main())
