// cpflow_amd: per-block 4x4 algebra of the dense-block kernel (cpf_dense.h), plain C++ so that the same code is
// unit-tested on the CPU (tests/test_dense_math.py builds it with g++).
//
// One entangling block of the template (cpflow/main.py:69-82) acts on its qubit pair as the 4x4 matrix
//     G = A_m ... A_2 A_1 E,     E = CP(a) | CZ | CX,   A_1 = R_g0(up), A_2 = R_g0(dn), A_3 = R_g1(up), ...
// in the basis index 2 * up + dn (up = first qubit of the placement, gates.py / circuit_assembly.py:16-28).
//   block_matrix      G from the block's (cos, sin) records (optionally with the surface gates of its qubits folded in:
//                     G' = G (S_up (x) S_dn), see "Surface gates folded into a block")         [before the forward sweep]
//   block_gradient    the block's angle gradients from the 4x4 environment
//                         M_ij = sum_{other qubits, columns} conj(C_i) psi_j        [after the backward sweep]
//                     where psi is the state before the block and C the loss cotangent after it:
//                         dL = 2 Re sum_ij dG_ij M_ij   =>   dL/dtheta_k = Im tr(sigma_k R_k M^T L_k),
//                     R_k = A_k .. A_1 E, L_k = A_m .. A_{k+1}: a miniature adjoint sweep over 4x4 matrices
//                     (the same formula the gate-by-gate kernels evaluate on the full state, cpf_device.h rot_grad).
#pragma once

#if defined(__HIPCC__)
#define CPF_HD __host__ __device__ __forceinline__
#else
#define CPF_HD inline
#endif

namespace cpf {
namespace dense {

enum : int { K_RX = 0, K_RY = 1, K_RZ = 2, K_CP = 3, K_CZ = 4, K_CX = 5 };     // = OP_* of cpf_device.h

// A complex number.  On the device it is a two-float vector, so that every helper below is one or two packed instructions
// (v_pk_mul_f32 / v_pk_fma_f32): the per-block walks run on one thread per block while the other lanes wait, so their
// instruction count is serial time (config 5: the gradient walk cost 5 % of the step with scalar arithmetic).  On the CPU
// (tests/dense_math_shim.cpp) the same helpers are plain float code.
#if defined(__HIP_DEVICE_COMPILE__)
typedef float cxv __attribute__((ext_vector_type(2)));
struct cx { cxv v; };
CPF_HD cx mk(float re, float im) { cx r; r.v = cxv{re, im}; return r; }
CPF_HD float re(cx a) { return a.v.x; }
CPF_HD float im(cx a) { return a.v.y; }
CPF_HD cx cmulc(cx a, cx b) { cx r; r.v = __builtin_elementwise_fma(cxv{-a.v.y, a.v.y}, cxv{b.v.y, b.v.x}, cxv{a.v.x, a.v.x} * b.v); return r; }
// (c a.re + p b.im,  c a.im + q b.re)
CPF_HD cx lin_swap(float c, cx a, float p, float q, cx b) { cx r; r.v = __builtin_elementwise_fma(cxv{p, q}, cxv{b.v.y, b.v.x}, cxv{c, c} * a.v); return r; }
// c a + s b
CPF_HD cx lin(float c, cx a, float s, cx b) { cx r; r.v = __builtin_elementwise_fma(cxv{s, s}, b.v, cxv{c, c} * a.v); return r; }
// accumulators of sum x z: acc_swap collects (x.re z.im, x.im z.re), acc_same (x.re z.re, x.im z.im); NEG subtracts the term
template <bool NEG> CPF_HD cx acc_swap(cx acc, cx x, cx z) { cx r; r.v = __builtin_elementwise_fma(NEG ? -x.v : x.v, cxv{z.v.y, z.v.x}, acc.v); return r; }
template <bool NEG> CPF_HD cx acc_same(cx acc, cx x, cx z) { cx r; r.v = __builtin_elementwise_fma(NEG ? -x.v : x.v, z.v, acc.v); return r; }
CPF_HD cx sel(bool t, cx a, cx b) { cx r; r.v = cxv{t ? a.v.x : b.v.x, t ? a.v.y : b.v.y}; return r; }
#else
struct cx { float re, im; };
CPF_HD cx mk(float re, float im) { cx r; r.re = re; r.im = im; return r; }
CPF_HD float re(cx a) { return a.re; }
CPF_HD float im(cx a) { return a.im; }
CPF_HD cx cmulc(cx a, cx b) { return mk(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
CPF_HD cx lin_swap(float c, cx a, float p, float q, cx b) { return mk(c * a.re + p * b.im, c * a.im + q * b.re); }
CPF_HD cx lin(float c, cx a, float s, cx b) { return mk(c * a.re + s * b.re, c * a.im + s * b.im); }
template <bool NEG> CPF_HD cx acc_swap(cx acc, cx x, cx z) { return NEG ? mk(acc.re - x.re * z.im, acc.im - x.im * z.re) : mk(acc.re + x.re * z.im, acc.im + x.im * z.re); }
template <bool NEG> CPF_HD cx acc_same(cx acc, cx x, cx z) { return NEG ? mk(acc.re - x.re * z.re, acc.im - x.im * z.im) : mk(acc.re + x.re * z.re, acc.im + x.im * z.im); }
CPF_HD cx sel(bool t, cx a, cx b) { return t ? a : b; }
#endif

// X <- R X (rows) for R = exp(-i a sigma / 2) on the up (bit 1) or dn (bit 0) index; (c, s) = (cos a/2, sin a/2);
// s -> -s gives the inverse.  X[r][j], r = 2 up + dn.
template <bool ON_UP>
CPF_HD void rot_rows(cx (&X)[4][4], int kind, float c, float s) {
    constexpr int b = ON_UP ? 2 : 1;
#pragma unroll
    for (int r0 = 0; r0 < 4; r0++) {
        if (r0 & b) continue;
        const int r1 = r0 | b;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const cx x0 = X[r0][j], x1 = X[r1][j];
            if (kind == K_RX) {            // [[c, -is], [-is, c]]
                X[r0][j] = lin_swap(c, x0, s, -s, x1);
                X[r1][j] = lin_swap(c, x1, s, -s, x0);
            } else if (kind == K_RY) {     // [[c, -s], [s, c]]
                X[r0][j] = lin(c, x0, -s, x1);
                X[r1][j] = lin(c, x1, s, x0);
            } else {                       // diag(c - is, c + is)
                X[r0][j] = lin_swap(c, x0, s, -s, x0);
                X[r1][j] = lin_swap(c, x1, -s, s, x1);
            }
        }
    }
}
// Z <- Z R (columns)
template <bool ON_UP>
CPF_HD void rot_cols(cx (&Z)[4][4], int kind, float c, float s) {
    constexpr int b = ON_UP ? 2 : 1;
#pragma unroll
    for (int c0 = 0; c0 < 4; c0++) {
        if (c0 & b) continue;
        const int c1 = c0 | b;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const cx z0 = Z[i][c0], z1 = Z[i][c1];
            if (kind == K_RX) {            // (Z R)[i][c0] = c z0 - i s z1, [i][c1] = -i s z0 + c z1
                Z[i][c0] = lin_swap(c, z0, s, -s, z1);
                Z[i][c1] = lin_swap(c, z1, s, -s, z0);
            } else if (kind == K_RY) {     // [i][c0] = c z0 + s z1, [i][c1] = -s z0 + c z1
                Z[i][c0] = lin(c, z0, s, z1);
                Z[i][c1] = lin(c, z1, -s, z0);
            } else {
                Z[i][c0] = lin_swap(c, z0, s, -s, z0);
                Z[i][c1] = lin_swap(c, z1, -s, s, z1);
            }
        }
    }
}
// Im tr(sigma X Z), sigma the Pauli matrix of `kind` on the up / dn index
template <bool ON_UP>
CPF_HD float im_trace_sigma(const cx (&X)[4][4], const cx (&Z)[4][4], int kind) {
    constexpr int b = ON_UP ? 2 : 1;
    // four accumulators (one per row r) advance in turn: a packed multiply-add that reads the result of the instruction
    // right before it is padded with an s_nop on this chip
    cx a4[4] = {mk(0.f, 0.f), mk(0.f, 0.f), mk(0.f, 0.f), mk(0.f, 0.f)};
#pragma unroll
    for (int j = 0; j < 4; j++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const bool hi = (r & b) != 0;
            const cx z = Z[j][r];
            if (kind == K_RX) {            // (sigma X)[r] = X[r ^ b]:  Im(x z) = x.re z.im + x.im z.re
                a4[r] = acc_swap<false>(a4[r], X[r ^ b][j], z);
            } else if (kind == K_RY) {     // (sigma X)[r] = -i X[r ^ b] (bit clear), +i X[r ^ b] (bit set):  Im(+-i x z) = +-Re(x z)
                a4[r] = hi ? acc_same<false>(a4[r], X[r ^ b][j], z) : acc_same<true>(a4[r], X[r ^ b][j], z);
            } else {                       // (sigma X)[r] = +X[r] (bit clear), -X[r] (bit set)
                a4[r] = hi ? acc_swap<true>(a4[r], X[r][j], z) : acc_swap<false>(a4[r], X[r][j], z);
            }
        }
    }
    const cx acc = mk((re(a4[0]) + re(a4[1])) + (re(a4[2]) + re(a4[3])), (im(a4[0]) + im(a4[1])) + (im(a4[2]) + im(a4[3])));
    return kind == K_RY ? re(acc) - im(acc) : re(acc) + im(acc);      // Re(x z) = x.re z.re - x.im z.im
}

// Entangler as the starting matrix.  (cc, sc) = (cos a, sin a) of the CP angle (unused for CZ / CX).
CPF_HD void entangler_matrix(cx (&X)[4][4], int ent, float cc, float sc) {
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) X[i][j] = mk(0.f, 0.f);
    X[0][0] = mk(1.f, 0.f); X[1][1] = mk(1.f, 0.f);
    // selects, not conditional stores (a store under a branch makes the compiler index the array dynamically -> scratch)
    const bool is_cx = ent == K_CX, is_cz = ent == K_CZ;       // CX: control = up, target = dn (gates.py:40-43)
    X[2][2] = mk(is_cx ? 0.f : 1.f, 0.f); X[2][3] = mk(is_cx ? 1.f : 0.f, 0.f);
    X[3][2] = mk(is_cx ? 1.f : 0.f, 0.f);
    X[3][3] = mk(is_cx ? 0.f : (is_cz ? -1.f : cc), (is_cx || is_cz) ? 0.f : sc);
}

// X <- E X (ADJ: E^+ X) on the rows, for the entangler `ent` with (cc, sc) = (cos a, sin a) of the CP angle
template <bool ADJ>
CPF_HD void entangler_rows(cx (&X)[4][4], int ent, float cc, float sc) {
    const bool is_cx = ent == K_CX, is_cz = ent == K_CZ;
    const cx ph = mk(is_cx ? 1.f : (is_cz ? -1.f : cc), (is_cx || is_cz) ? 0.f : (ADJ ? -sc : sc));
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const cx r2 = X[2][j], r3 = X[3][j];
        X[2][j] = sel(is_cx, r3, r2);
        X[3][j] = cmulc(ph, sel(is_cx, r2, r3));
    }
}
// Z <- Z E on the columns
CPF_HD void entangler_cols(cx (&Z)[4][4], int ent, float cc, float sc) {
    const bool is_cx = ent == K_CX, is_cz = ent == K_CZ;
    const cx ph = mk(is_cx ? 1.f : (is_cz ? -1.f : cc), (is_cx || is_cz) ? 0.f : sc);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const cx c2 = Z[i][2], c3 = Z[i][3];
        Z[i][2] = sel(is_cx, c3, c2);
        Z[i][3] = cmulc(ph, sel(is_cx, c2, c3));
    }
}

// Surface gates folded into a block.  The template starts with Rz(t2) Rx(t1) Rz(t0) on every qubit (main.py:122-124);
// the first block that touches a qubit can carry that qubit's three rotations on its input side,
//     G' = A_m ... A_1 E (S_up (x) S_dn),      S = Rz(t2) Rx(t1) Rz(t0)  or the identity,
// so that the state sweeps never apply them: the forward sweep starts from the identity and the environment of the block
// also yields the gradients of the folded angles (the walk below simply continues through E into the S factors).
// `surf(side, k, &c, &s)`: (cos, sin) of the half angle of surface rotation k = 0, 1, 2 of the up (side 0) / dn (side 1) qubit.
struct NoSurf { CPF_HD void operator()(int, int, float& c, float& s) const { c = 1.f; s = 0.f; } };
CPF_HD int surf_kind(int k) { return k == 1 ? K_RX : K_RZ; }

// rec: the block's gate records in program order, 4 floats each {cos, sin, code, -} with code = kind | q0 << 4 | q1 << 8
// (the generic kernel's LDS records): rec[0] the entangler, then R_g(up), R_g(dn) per rotation letter.
// `read(g, &c, &s, &kind)` abstracts the record fetch so that device code reads LDS and the CPU test reads an array.
// ROT != 0: the template's rotation string and entangler are compile-time constants (packed as cpf_blocks.h rot_pack /
// rot_with_ent: letters | kind_i << (4 + 4 i) | entangler << 16): the walks below unroll completely and every three-way
// choice folds away.  ROT = 0: kinds come with the records at run time (any rotation string up to 8 letters).
template <int ROT> CPF_HD constexpr int rotc_n(int n_rot) { return ROT ? (ROT & 15) : n_rot; }
template <int ROT> CPF_HD constexpr int rotc_kind(int i, int runtime_kind) { return ROT ? ((ROT >> (4 + 4 * i)) & 15) : runtime_kind; }
template <int ROT> CPF_HD constexpr int rotc_ent(int runtime_kind) { return ROT ? K_CP + ((ROT >> 16) & 3) : runtime_kind; }

template <int ROT = 0, class Rec, class Surf = NoSurf>
CPF_HD void block_matrix(cx (&G)[4][4], const Rec& rec, int n_rot, bool fold_up = false, bool fold_dn = false, const Surf& surf = Surf()) {
    float c, s; int kind;
    rec(0, c, s, kind);
    kind = rotc_ent<ROT>(kind);
    entangler_matrix(G, kind, c, s);
    if (fold_up || fold_dn) {                    // G = E S: start from the identity, S on the rows, then E
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) G[i][j] = mk(i == j ? 1.f : 0.f, 0.f);
        float cs, ss;
        if (fold_up) for (int k = 0; k < 3; k++) { surf(0, k, cs, ss); rot_rows<true>(G, surf_kind(k), cs, ss); }
        if (fold_dn) for (int k = 0; k < 3; k++) { surf(1, k, cs, ss); rot_rows<false>(G, surf_kind(k), cs, ss); }
        entangler_rows<false>(G, kind, c, s);
    }
#pragma unroll
    for (int i = 0; i < rotc_n<ROT>(n_rot); i++) {
        rec(1 + 2 * i, c, s, kind);
        rot_rows<true>(G, rotc_kind<ROT>(i, kind), c, s);
        rec(2 + 2 * i, c, s, kind);
        rot_rows<false>(G, rotc_kind<ROT>(i, kind), c, s);
    }
}
// grad[0] = entangler angle (CP only, else 0), grad[1 + k] = rotation k in program order.  G = the block matrix.
// gs_up[k] / gs_dn[k] (k = 0, 1, 2): gradients of the folded surface angles (written only for a folded side).
template <int ROT = 0, class Rec, class Surf = NoSurf>
CPF_HD void block_gradient(float* grad, const cx (&G)[4][4], const cx (&M)[4][4], const Rec& rec, int n_rot,
                           bool fold_up = false, bool fold_dn = false, const Surf& surf = Surf(), float* gs_up = nullptr, float* gs_dn = nullptr) {
    cx X[4][4], Z[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) { X[i][j] = G[i][j]; Z[i][j] = M[j][i]; }
    float c, s; int kind;
#pragma unroll
    for (int i = rotc_n<ROT>(n_rot) - 1; i >= 0; i--) {
        rec(2 + 2 * i, c, s, kind);                       // R_i(dn), the later of the pair
        const int kd = rotc_kind<ROT>(i, kind);
        grad[2 + 2 * i] = im_trace_sigma<false>(X, Z, kd);
        rot_rows<false>(X, kd, c, -s);
        rot_cols<false>(Z, kd, c, s);
        rec(1 + 2 * i, c, s, kind);                       // R_i(up)
        const int ku = rotc_kind<ROT>(i, kind);
        grad[1 + 2 * i] = im_trace_sigma<true>(X, Z, ku);
        rot_rows<true>(X, ku, c, -s);
        rot_cols<true>(Z, ku, c, s);
    }
    rec(0, c, s, kind);
    kind = rotc_ent<ROT>(kind);
    // here X = E S and Z = M^T A_m .. A_1.  dE/da = i |11><11| E  ->  dL/da = 2 Re(i (X Z)[3][3]) = -2 Im (X Z)[3][3]
    // (S = 1: X[3][j] = e^{ia} delta_j3, i.e. -2 Im(e^{ia} Z[3][3]))
    cx a33 = mk(0.f, 0.f);
#pragma unroll
    for (int j = 0; j < 4; j++) a33 = acc_swap<false>(a33, X[3][j], Z[j][3]);
    grad[0] = kind == K_CP ? -2.f * (re(a33) + im(a33)) : 0.f;
    if (fold_up || fold_dn) {
        entangler_rows<true>(X, kind, c, s);              // X = S
        entangler_cols(Z, kind, c, s);                    // Z = M^T A_m .. A_1 E
        float cs, ss;
        if (fold_dn) for (int k = 2; k >= 0; k--) {
            surf(1, k, cs, ss);
            gs_dn[k] = im_trace_sigma<false>(X, Z, surf_kind(k));
            rot_rows<false>(X, surf_kind(k), cs, -ss);
            rot_cols<false>(Z, surf_kind(k), cs, ss);
        }
        if (fold_up) for (int k = 2; k >= 0; k--) {
            surf(0, k, cs, ss);
            gs_up[k] = im_trace_sigma<true>(X, Z, surf_kind(k));
            rot_rows<true>(X, surf_kind(k), cs, -ss);
            rot_cols<true>(Z, surf_kind(k), cs, ss);
        }
    }
}

// When the placement's first qubit sits on the LOWER of the two register bits the kernel applies the block with the
// roles of the index bits exchanged: G'[(i_d, i_u)][(j_d, j_u)] = G[(i_u, i_d)][(j_u, j_d)].
CPF_HD int swap_bits2(int i) { return ((i & 1) << 1) | (i >> 1); }

}  // namespace dense
}  // namespace cpf
