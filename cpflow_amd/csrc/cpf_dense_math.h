// cpflow_amd: per-block 4x4 algebra of the dense-block kernel (cpf_dense.h), plain C++ so that the same code is
// unit-tested on the CPU (tests/test_dense_math.py builds it with g++).
//
// One entangling block of the template (cpflow/main.py:69-82) acts on its qubit pair as the 4x4 matrix
//     G = A_m ... A_2 A_1 E,     E = CP(a) | CZ | CX,   A_1 = R_g0(up), A_2 = R_g0(dn), A_3 = R_g1(up), ...
// in the basis index 2 * up + dn (up = first qubit of the placement, gates.py / circuit_assembly.py:16-28).
//   block_matrix      G from the block's (cos, sin) records (column by column: block_matrix_column; optionally with the surface gates of its qubits folded in:
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

// ---- slices -------------------------------------------------------------------------------------------------------
// Row operations X <- R X act on every COLUMN of X separately, column operations Z <- Z R on every ROW of Z separately, and
// Im tr(sigma X Z) = sum_j [a term of column j of X and row j of Z].  So a block's walks split four ways: slice j holds
// x[r] = X[r][j] and z[c] = Z[j][c], applies every operation to its two 4-vectors and contributes one term per trace.
// On the device four neighbouring lanes take the four slices of a block (the traces meet in a two-step quad sum): the walks
// run on one wave while the sample's other lanes wait, so their length is serial time, and a slice is a quarter of it.
// On the CPU the whole-matrix functions below loop over the slices.

// v <- R v for a column of X (COLS = false) or v <- v R for a row of Z (COLS = true); R = exp(-i a sigma / 2) on the up
// (index bit 1) or dn (bit 0) qubit, (c, s) = (cos a/2, sin a/2); s -> -s gives the inverse.
template <bool ON_UP, bool COLS>
CPF_HD void rot_slice(cx (&v)[4], int kind, float c, float s) {
    constexpr int b = ON_UP ? 2 : 1;
#pragma unroll
    for (int i0 = 0; i0 < 4; i0++) {
        if (i0 & b) continue;
        const int i1 = i0 | b;
        const cx v0 = v[i0], v1 = v[i1];
        if (kind == K_RX) {                // rows: [[c, -is], [-is, c]] v;  columns: the same (R is symmetric)
            v[i0] = lin_swap(c, v0, s, -s, v1);
            v[i1] = lin_swap(c, v1, s, -s, v0);
        } else if (kind == K_RY) {         // rows: [[c, -s], [s, c]] v;  columns: v [[c, -s], [s, c]] = (c v0 + s v1, -s v0 + c v1)
            v[i0] = lin(c, v0, COLS ? s : -s, v1);
            v[i1] = lin(c, v1, COLS ? -s : s, v0);
        } else {                           // diag(c - is, c + is)
            v[i0] = lin_swap(c, v0, s, -s, v0);
            v[i1] = lin_swap(c, v1, -s, s, v1);
        }
    }
}
// slice j's term of Im tr(sigma X Z): sum_r (sigma X)[r][j] Z[j][r], sigma the Pauli matrix of `kind` on the up / dn index
template <bool ON_UP>
CPF_HD float im_trace_slice(const cx (&x)[4], const cx (&z)[4], int kind) {
    constexpr int b = ON_UP ? 2 : 1;
    // two accumulators advance in turn: a packed multiply-add that reads the result of the instruction right before it is
    // padded with an s_nop on this chip
    cx a2[2] = {mk(0.f, 0.f), mk(0.f, 0.f)};
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const bool hi = (r & b) != 0;
        if (kind == K_RX) {                // (sigma X)[r] = X[r ^ b]:  Im(x z) = x.re z.im + x.im z.re
            a2[r & 1] = acc_swap<false>(a2[r & 1], x[r ^ b], z[r]);
        } else if (kind == K_RY) {         // (sigma X)[r] = -i X[r ^ b] (bit clear), +i X[r ^ b] (bit set):  Im(+-i x z) = +-Re(x z)
            a2[r & 1] = hi ? acc_same<false>(a2[r & 1], x[r ^ b], z[r]) : acc_same<true>(a2[r & 1], x[r ^ b], z[r]);
        } else {                           // (sigma X)[r] = +X[r] (bit clear), -X[r] (bit set)
            a2[r & 1] = hi ? acc_swap<true>(a2[r & 1], x[r], z[r]) : acc_swap<false>(a2[r & 1], x[r], z[r]);
        }
    }
    const float pr = re(a2[0]) + re(a2[1]), pi = im(a2[0]) + im(a2[1]);
    return kind == K_RY ? pr - pi : pr + pi;                   // Re(x z) = x.re z.re - x.im z.im
}
// v <- E v (column of X; ADJ: E^+ v) or v <- v E (row of Z, ADJ = false): both touch entries 2 and 3 only
template <bool ADJ>
CPF_HD void entangler_slice(cx (&v)[4], int ent, float cc, float sc) {
    const bool is_cx = ent == K_CX, is_cz = ent == K_CZ;
    const cx ph = mk(is_cx ? 1.f : (is_cz ? -1.f : cc), (is_cx || is_cz) ? 0.f : (ADJ ? -sc : sc));
    const cx v2 = v[2], v3 = v[3];
    v[2] = sel(is_cx, v3, v2);
    v[3] = cmulc(ph, sel(is_cx, v2, v3));
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

// column j of the block matrix G' = A_m .. A_1 E (S_up (x) S_dn): the unit vector e_j through every row operation
template <int ROT = 0, class Rec, class Surf = NoSurf>
CPF_HD void block_matrix_column(cx (&x)[4], int j, const Rec& rec, int n_rot, bool fold_up = false, bool fold_dn = false, const Surf& surf = Surf()) {
#pragma unroll
    for (int i = 0; i < 4; i++) x[i] = mk(i == j ? 1.f : 0.f, 0.f);
    float c, s, cs, ss; int kind;
    if (fold_up) for (int k = 0; k < 3; k++) { surf(0, k, cs, ss); rot_slice<true, false>(x, surf_kind(k), cs, ss); }
    if (fold_dn) for (int k = 0; k < 3; k++) { surf(1, k, cs, ss); rot_slice<false, false>(x, surf_kind(k), cs, ss); }
    rec(0, c, s, kind);
    entangler_slice<false>(x, rotc_ent<ROT>(kind), c, s);
#pragma unroll
    for (int i = 0; i < rotc_n<ROT>(n_rot); i++) {
        rec(1 + 2 * i, c, s, kind);
        rot_slice<true, false>(x, rotc_kind<ROT>(i, kind), c, s);
        rec(2 + 2 * i, c, s, kind);
        rot_slice<false, false>(x, rotc_kind<ROT>(i, kind), c, s);
    }
}
// Slice j's share of the block's gradients: x = column j of G', z = row j of M^T (= column j of the environment M).
// grad[0] = entangler angle (CP only, else 0), grad[1 + k] = rotation k in program order; gs_up[k] / gs_dn[k] (k = 0, 1, 2):
// the folded surface angles (written only for a folded side).  The four slices' shares add up to the gradients.
template <int ROT = 0, class Rec, class Surf = NoSurf>
CPF_HD void block_gradient_slice(float* grad, cx (&x)[4], cx (&z)[4], const Rec& rec, int n_rot,
                                 bool fold_up = false, bool fold_dn = false, const Surf& surf = Surf(), float* gs_up = nullptr, float* gs_dn = nullptr) {
    float c, s; int kind;
#pragma unroll
    for (int i = rotc_n<ROT>(n_rot) - 1; i >= 0; i--) {
        rec(2 + 2 * i, c, s, kind);                       // R_i(dn), the later of the pair
        const int kd = rotc_kind<ROT>(i, kind);
        grad[2 + 2 * i] = im_trace_slice<false>(x, z, kd);
        rot_slice<false, false>(x, kd, c, -s);
        rot_slice<false, true>(z, kd, c, s);
        rec(1 + 2 * i, c, s, kind);                       // R_i(up)
        const int ku = rotc_kind<ROT>(i, kind);
        grad[1 + 2 * i] = im_trace_slice<true>(x, z, ku);
        rot_slice<true, false>(x, ku, c, -s);
        rot_slice<true, true>(z, ku, c, s);
    }
    rec(0, c, s, kind);
    kind = rotc_ent<ROT>(kind);
    // here X = E S and Z = M^T A_m .. A_1.  dE/da = i |11><11| E  ->  dL/da = 2 Re(i (X Z)[3][3]) = -2 Im (X Z)[3][3]
    // (S = 1: X[3][j] = e^{ia} delta_j3, i.e. -2 Im(e^{ia} Z[3][3])); slice j's term is X[3][j] Z[j][3]
    const cx a33 = acc_swap<false>(mk(0.f, 0.f), x[3], z[3]);
    grad[0] = kind == K_CP ? -2.f * (re(a33) + im(a33)) : 0.f;
    if (fold_up || fold_dn) {
        entangler_slice<true>(x, kind, c, s);             // X = S
        entangler_slice<false>(z, kind, c, s);            // Z = M^T A_m .. A_1 E
        float cs, ss;
        if (fold_dn) for (int k = 2; k >= 0; k--) {
            surf(1, k, cs, ss);
            gs_dn[k] = im_trace_slice<false>(x, z, surf_kind(k));
            rot_slice<false, false>(x, surf_kind(k), cs, -ss);
            rot_slice<false, true>(z, surf_kind(k), cs, ss);
        }
        if (fold_up) for (int k = 2; k >= 0; k--) {
            surf(0, k, cs, ss);
            gs_up[k] = im_trace_slice<true>(x, z, surf_kind(k));
            rot_slice<true, false>(x, surf_kind(k), cs, -ss);
            rot_slice<true, true>(z, surf_kind(k), cs, ss);
        }
    }
}

// ---- whole matrices (the CPU tests; the device takes the slices directly) ---------------------------------------------
template <int ROT = 0, class Rec, class Surf = NoSurf>
CPF_HD void block_matrix(cx (&G)[4][4], const Rec& rec, int n_rot, bool fold_up = false, bool fold_dn = false, const Surf& surf = Surf()) {
    for (int j = 0; j < 4; j++) {
        cx x[4];
        block_matrix_column<ROT>(x, j, rec, n_rot, fold_up, fold_dn, surf);
        for (int i = 0; i < 4; i++) G[i][j] = x[i];
    }
}
constexpr int MAX_BLOCK_ANGLES = 17;                      // 1 + 2 x 8 rotation letters
template <int ROT = 0, class Rec, class Surf = NoSurf>
CPF_HD void block_gradient(float* grad, const cx (&G)[4][4], const cx (&M)[4][4], const Rec& rec, int n_rot,
                           bool fold_up = false, bool fold_dn = false, const Surf& surf = Surf(), float* gs_up = nullptr, float* gs_dn = nullptr) {
    const int na = 1 + 2 * rotc_n<ROT>(n_rot);
    for (int k = 0; k < na; k++) grad[k] = 0.f;
    float su[3] = {0.f, 0.f, 0.f}, sd[3] = {0.f, 0.f, 0.f};
    for (int j = 0; j < 4; j++) {
        cx x[4], z[4];
        for (int i = 0; i < 4; i++) { x[i] = G[i][j]; z[i] = M[i][j]; }        // Z[j][i] = M[i][j]
        float g[MAX_BLOCK_ANGLES], pu[3] = {0.f, 0.f, 0.f}, pd[3] = {0.f, 0.f, 0.f};
        block_gradient_slice<ROT>(g, x, z, rec, n_rot, fold_up, fold_dn, surf, pu, pd);
        for (int k = 0; k < na; k++) grad[k] += g[k];
        for (int k = 0; k < 3; k++) { su[k] += pu[k]; sd[k] += pd[k]; }
    }
    if (fold_up) for (int k = 0; k < 3; k++) gs_up[k] = su[k];
    if (fold_dn) for (int k = 0; k < 3; k++) gs_dn[k] = sd[k];
}

// When the placement's first qubit sits on the LOWER of the two register bits the kernel applies the block with the
// roles of the index bits exchanged: G'[(i_d, i_u)][(j_d, j_u)] = G[(i_u, i_d)][(j_u, j_d)].
CPF_HD int swap_bits2(int i) { return ((i & 1) << 1) | (i >> 1); }

}  // namespace dense
}  // namespace cpf
