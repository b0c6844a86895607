// cpflow_amd: per-block algebra of the ZYZ-form kernel (cpf_zyz.h), plain C++ so that the same code is unit-tested on the CPU
// (tests/test_zyz_math.py builds it with g++; tools/zyz_model.py is the NumPy specification of the whole scheme).
//
// One entangling block of the template (cpflow/main.py:69-82) is G = (A_up (x) A_dn) E with A = R_m .. R_1 a product of
// rotations and E = CP(a) | CZ (gates.py:22-58).  Every A is an SU(2) element [[p, -conj q], [q, conj p]] and factors as
//
//     A = diag(l0, l1) . sc [[1, -t], [t, 1]] . diag(1, r1) . F^f        F = Ry(pi) = [[0, -1], [1, 0]],
//
// f = [|q| > |p|] (A F^-1 is factored then, so that t = tan <= 1 always), sc = cos, l0 = p / |p|, l1 = q / |q|, r1 = conj(l0 l1):
// no square roots, no Euler angles, nothing singular at t = 0.  Flips F commute with every Y stage and turn a diagonal factor
// into the same factor with its two entries exchanged; they are pushed back through all earlier blocks onto the initial state,
// so the state the sweeps hold is (pending flips of later blocks) x (true state).  Between two Y stages everything is diagonal
// and merges into ONE phase per amplitude of the block's pair:
//     d_ab = [L~_prev(up)]_a [R~(up)]_a [L~_prev(dn)]_b [R~(dn)]_b [entangler]_ab,   global phase divided out, sc_u sc_d folded in.
// Sweeps (cpf_zyz.h): forward x_ab <- d_ab x_ab, then the two unnormalised real stages [[1, -t], [t, 1]]; backward the reverse
// with conj(d), three generator sums P_k = sum Im(conj(w) sigma_k psi) per qubit right after the Y stage, one sum over the |11>
// amplitudes before it.  This header holds what runs once per block and step around the sweeps:
//   side_build     the factors of one side of a block from the (cos, sin) records of its rotations      [before the sweeps]
//   block_table    the merged phases + the two tangents of a block, in the kernel's index order          [before the sweeps]
//   chain_side     the gradients of a side's rotation angles from its three generator sums              [after the sweeps]
//   chain_cp       the gradient of the entangler's angle                                                [after the sweeps]
//   surface_matrix / chain_surface     the surface gate Rz Rx Rz of a qubit (main.py:122-124) and its three gradients
#pragma once

#if defined(__HIPCC__)
#define CPF_ZHD __host__ __device__ __forceinline__
#else
#define CPF_ZHD inline
#include <cmath>
#endif

namespace cpf {
namespace zyz {

enum : int { Z_RX = 0, Z_RY = 1, Z_RZ = 2, Z_CP = 3, Z_CZ = 4, Z_CX = 5 };     // = OP_* of cpf_device.h

struct C2 { float re, im; };
CPF_ZHD C2 c2(float re, float im) { C2 r; r.re = re; r.im = im; return r; }
CPF_ZHD C2 cmul(C2 a, C2 b) { return c2(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
CPF_ZHD C2 cconj(C2 a) { return c2(a.re, -a.im); }
CPF_ZHD C2 cscale(C2 a, float s) { return c2(a.re * s, a.im * s); }
// 1 / sqrt(x).  The hardware approximation (1 ulp, and not centred: 60 blocks of it cost config 5's unitary 1e-5 of its norm)
// gets one Newton step, after which only the roundings of the last two operations remain.
CPF_ZHD float zrsqrt(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float y = __builtin_amdgcn_rsqf(x);
    const float e = __builtin_fmaf(-(x * y), 0.5f * y, 0.5f);
    return __builtin_fmaf(y, e, y);
#else
    return 1.0f / std::sqrt(x);
#endif
}

// rotation string / entangler as a compile-time code (cpf_blocks.h rot_pack / rot_with_ent) or, ROT = 0, from the records
template <int ROT> CPF_ZHD constexpr int rotc_n(int n_rot) { return ROT ? (ROT & 15) : n_rot; }
template <int ROT> CPF_ZHD constexpr int rotc_kind(int i, int runtime_kind) { return ROT ? ((ROT >> (4 + 4 * i)) & 15) : runtime_kind; }

// One side (up or dn) of a block.  f: the side carries a flip on its input side; m: flips of LATER blocks on the same qubit
// are pending at its output side (filled in by the caller from the f bits of the whole circuit).
struct Side { float t, sc; C2 l0, l1; int f, m; };

// `rec(i, &c, &s, &kind)`: (cos, sin) of the half angle and the kind of gate i of the block in program order: 0 the entangler,
// then R_g(up), R_g(dn) per rotation letter.  side: 0 = up, 1 = dn.
template <int ROT = 0, class Rec>
CPF_ZHD Side side_build(const Rec& rec, int n_rot, int side) {
    C2 p = c2(1.f, 0.f), q = c2(0.f, 0.f);
#pragma unroll
    for (int i = 0; i < rotc_n<ROT>(n_rot); i++) {
        float c, s; int kind;
        rec(1 + 2 * i + side, c, s, kind);
        kind = rotc_kind<ROT>(i, kind);
        C2 pn, qn;
        if (i == 0) {                        // the first rotation acts on the identity: its first column, without the products by 0 and 1
            pn = c2(c, kind == Z_RZ ? -s : 0.f);
            qn = kind == Z_RX ? c2(0.f, -s) : (kind == Z_RY ? c2(s, 0.f) : c2(0.f, 0.f));
        } else if (kind == Z_RX) {           // [[c, -is], [-is, c]]
            pn = c2(c * p.re + s * q.im, c * p.im - s * q.re);
            qn = c2(c * q.re + s * p.im, c * q.im - s * p.re);
        } else if (kind == Z_RY) {           // [[c, -s], [s, c]]
            pn = c2(c * p.re - s * q.re, c * p.im - s * q.im);
            qn = c2(c * q.re + s * p.re, c * q.im + s * p.im);
        } else {                             // diag(c - is, c + is)
            pn = c2(c * p.re + s * p.im, c * p.im - s * p.re);
            qn = c2(c * q.re - s * q.im, c * q.im + s * q.re);
        }
        p = pn; q = qn;
    }
    const float n2 = p.re * p.re + p.im * p.im, m2 = q.re * q.re + q.im * q.im;
    Side sd;
    // Selects, not branches (the owner phase of the two-ended 3-qubit kernel is on every step's critical path, and the compiler turned
    // the two `if` blocks this used to be into exec-mask regions of moves): the same operations on the same numbers.
    const bool fl = m2 > n2;                 // A F^-1 = [[conj q, p], [-conj p, q]]
    sd.f = fl ? 1 : 0;
    sd.m = 0;
    const C2 pp = c2(fl ? q.re : p.re, fl ? -q.im : p.im), qq = c2(fl ? -p.re : q.re, fl ? p.im : q.im);
    const float nn = fl ? m2 : n2, mm = fl ? n2 : m2;
    const float ip = zrsqrt(nn);
    sd.l0 = cscale(pp, ip);
    const bool ok = mm > 1e-30f;
    const float iq = zrsqrt(ok ? mm : 1.f);
    sd.l1 = c2(ok ? qq.re * iq : 1.f, ok ? qq.im * iq : 0.f);
    sd.t = ok ? (mm * iq) * ip : 0.f;
    // the scale from the tangent that is actually applied: sc [[1, -t], [t, 1]] is then unitary to rounding whatever the
    // roundings of (p, q) were (sc = |p| would inherit their |p|^2 + |q|^2 - 1)
    sd.sc = zrsqrt(1.f + sd.t * sd.t);
    return sd;
}

// What a block's sweep needs, in the kernel's index order k = 2 * hi + lo (lo / hi: the pair's lower / higher register bit;
// up_low: the placement's first qubit sits on the lower one):
//   out[0] = d_0 (real: the scale sc_u sc_d, or `scale`), out[1] = 0, out[2 k], out[2 k + 1] = d_k (k = 1, 2, 3), out[8] = t_lo, out[9] = t_hi.
// up, dn: the block's sides with f and m set.  has_pu / has_pd: a previous block touches the qubit; pu / pd: that block's side
// (its m set too).  (cc, sc): cos / sin of the CP angle, (-1, 0) for CZ.  Returns (ca | cb << 1): the amplitude of the pair
// (up bit ca, dn bit cb) that carries the entangler's phase in the held frame.
CPF_ZHD int block_table(float* out, const Side& up, const Side& dn, bool has_pu, const Side& pu, bool has_pd, const Side& pd,
                        float cc, float sc, bool up_low, float scale = -1.f) {
    C2 eu[2], ed[2];
    {
        const C2 r1 = cconj(cmul(up.l0, up.l1));
        C2 p0 = c2(1.f, 0.f), p1 = c2(1.f, 0.f);
        if (has_pu) { p0 = pu.m ? pu.l1 : pu.l0; p1 = pu.m ? pu.l0 : pu.l1; }
        eu[0] = up.m ? cmul(p0, r1) : p0;
        eu[1] = up.m ? p1 : cmul(p1, r1);
    }
    {
        const C2 r1 = cconj(cmul(dn.l0, dn.l1));
        C2 p0 = c2(1.f, 0.f), p1 = c2(1.f, 0.f);
        if (has_pd) { p0 = pd.m ? pd.l1 : pd.l0; p1 = pd.m ? pd.l0 : pd.l1; }
        ed[0] = dn.m ? cmul(p0, r1) : p0;
        ed[1] = dn.m ? p1 : cmul(p1, r1);
    }
    const int ca = 1 ^ up.m ^ up.f, cb = 1 ^ dn.m ^ dn.f;
    const C2 lam = c2(cc, sc);
    C2 d[4];                                                   // index 2 a + b, a the up bit
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) {
            C2 v = cmul(eu[a], ed[b]);
            if (a == ca && b == cb) v = cmul(v, lam);
            d[2 * a + b] = v;
        }
    // scale >= 0: the modulus the caller wants on the phases instead of this block's own sc_u sc_d (cpf_zyz.h: the last block of a
    // layer carries the scale of the whole layer, the others 1)
    const float s = scale >= 0.f ? scale : up.sc * dn.sc;
    const C2 g = cscale(cconj(d[0]), s);                       // every factor has unit modulus: conj(d_00) is its inverse
    out[0] = s; out[1] = 0.f;
#pragma unroll
    for (int k = 1; k < 4; k++) {
        const int lo = k & 1, hi = k >> 1;
        const int a = up_low ? lo : hi, b = up_low ? hi : lo;
        const C2 v = cmul(d[2 * a + b], g);
        out[2 * k] = v.re; out[2 * k + 1] = v.im;
    }
    out[8] = up_low ? up.t : dn.t;
    out[9] = up_low ? dn.t : up.t;
    return ca | (cb << 1);
}

// ---- The same table in two lane-parallel steps (the two-ended 3-qubit kernel, cpf_zyz3.h: one lane per side, then one lane per
// table entry).  side_epair: what one SIDE contributes to its block's merged phases - (left factors the previous block on the qubit
// left behind) x (this side's right factors), the entries exchanged by the pending parities - and the bit of the pair's amplitude
// that carries the entangler's phase in the held frame.  The previous side's parity is this side's xor its own flip (the sides
// later than the previous one are this one and those later than it).
struct EPair { C2 e0, e1; int c; };
CPF_ZHD EPair side_epair(const Side& sd, bool has_prev, C2 pl0, C2 pl1) {
    const C2 r1 = cconj(cmul(sd.l0, sd.l1));
    C2 p0 = c2(1.f, 0.f), p1 = c2(1.f, 0.f);
    if (has_prev) { const int mp = sd.m ^ sd.f; p0 = mp ? pl1 : pl0; p1 = mp ? pl0 : pl1; }
    EPair e;
    e.e0 = sd.m ? cmul(p0, r1) : p0;
    e.e1 = sd.m ? p1 : cmul(p1, r1);
    e.c = 1 ^ sd.m ^ sd.f;
    return e;
}
// entry (a, b) of the block's table (a: the up qubit's bit, b: the dn qubit's): out[0], out[1] = d_ab with the global phase divided
// out and the real scale s = sc_u sc_d folded in (entry (0, 0): (s, 0) exactly), out[2], out[3] = the tangents of the two stages
// signed for a lane whose bits are (a, b): x <- x + out[2] x[partner on up], then the same with out[3] on dn.
CPF_ZHD void table_entry(float* out, int a, int b, const EPair& eu, const EPair& ed, float tu, float scu, float td, float scd, float cc, float sc) {
    const C2 lam = c2(cc, sc);
    C2 d00 = cmul(eu.e0, ed.e0);
    if (eu.c == 0 && ed.c == 0) d00 = cmul(d00, lam);
    C2 d = cmul(a ? eu.e1 : eu.e0, b ? ed.e1 : ed.e0);
    if (a == eu.c && b == ed.c) d = cmul(d, lam);
    const float s = scu * scd;
    const C2 v = cmul(d, cscale(cconj(d00), s));
    const bool first = a == 0 && b == 0;
    out[0] = first ? s : v.re; out[1] = first ? 0.f : v.im;
    out[2] = a ? tu : -tu; out[3] = b ? td : -td;
}

// so3(R)^T P for R = exp(-i a sigma_axis / 2), (c, s) = (cos a/2, sin a/2): the two other components rotate by -a
CPF_ZHD void so3_inv(int axis, float c, float s, float (&P)[3]) {
    const float ca = c * c - s * s, sa = 2.f * c * s;
    const int i = axis == 2 ? 0 : axis + 1, j = i == 2 ? 0 : i + 1;     // (y, z) for x; (z, x) for y; (x, y) for z
    const float pi = P[i], pj = P[j];
    P[i] = ca * pi + sa * pj;
    P[j] = -sa * pi + ca * pj;
}

// Gradients of a side's rotation angles: grad[i] for rotation letter i.  px, py, pz: the held generator sums
// sum Im(conj(w) sigma_k psi) of the side's qubit right after the block's Y stage.
template <int ROT = 0, class Rec>
CPF_ZHD void chain_side(float* grad, float px, float py, float pz, const Side& sd, const Rec& rec, int n_rot, int side) {
    float P[3] = {sd.m ? -px : px, py, sd.m ? -pz : pz};       // pending flips: sigma_x and sigma_z change sign
    const C2 u = cmul(sd.l0, cconj(sd.l1));                    // the frame right after A: rotation about z by alpha, exp(-i alpha) = u
    const float x = u.re * P[0] + u.im * P[1], y = -u.im * P[0] + u.re * P[1];
    P[0] = x; P[1] = y;
#pragma unroll
    for (int i = rotc_n<ROT>(n_rot) - 1; i >= 0; i--) {
        float c, s; int kind;
        rec(1 + 2 * i + side, c, s, kind);
        kind = rotc_kind<ROT>(i, kind);
        grad[i] = kind == Z_RX ? P[0] : (kind == Z_RY ? P[1] : P[2]);
        so3_inv(kind, c, s, P);
    }
}

// Gradient of the CP angle.  s11: the held sum of Im(conj(w) psi) over the pair's |11> amplitudes taken after the block's Y stage
// was undone and before conj(d) was applied (both matrices carry 1 / d_0 there); pzu, pzd: the held z sums of the two qubits
// before the block's Y stage, i.e. those of the previous block on each qubit (or of the surface state); cacb: block_table's
// return value; d0: the block's real scale.  sum over all amplitudes of Im(conj(w) psi) = 0 for every loss kind.
CPF_ZHD float chain_cp(float s11, float pzu, float pzd, int cacb, float d0) {
    const float t11 = s11 * d0 * d0;
    const int ca = cacb & 1, cb = cacb >> 1;
    float t;
    if (ca && cb) t = t11;
    else if (ca) t = -0.5f * pzu - t11;
    else if (cb) t = -0.5f * pzd - t11;
    else t = 0.5f * (pzu + pzd) + t11;
    return -2.f * t;
}

// S~ = F^m Rz(t2) Rx(t1) Rz(t0) (main.py:122-124), row-major (re, im) x 4.  (c_k, s_k): cos / sin of t_k / 2.
CPF_ZHD void surface_matrix(float* out, float c0, float s0, float c1, float s1, float c2_, float s2, int m) {
    // Rz(t2) Rx(t1) Rz(t0) = [[c1 e^{-i(a+b)}, -i s1 e^{-i(b-a)}], [-i s1 e^{i(b-a)}, c1 e^{i(a+b)}]], a = t0 / 2, b = t2 / 2
    const C2 ep = cmul(c2(c0, s0), c2(c2_, s2));               // e^{i(a+b)}
    const C2 em = cmul(c2(c0, -s0), c2(c2_, s2));              // e^{i(b-a)}
    const C2 s00 = cscale(cconj(ep), c1), s11 = cscale(ep, c1);
    const C2 s01 = cscale(cmul(c2(0.f, -1.f), cconj(em)), s1), s10 = cscale(cmul(c2(0.f, -1.f), em), s1);
    C2 r[4] = {s00, s01, s10, s11};
    if (m) { r[0] = c2(-s10.re, -s10.im); r[1] = c2(-s11.re, -s11.im); r[2] = s00; r[3] = s01; }       // F S: row 0 = -row 1, row 1 = row 0
#pragma unroll
    for (int k = 0; k < 4; k++) { out[2 * k] = r[k].re; out[2 * k + 1] = r[k].im; }
}
// grad[k]: gradient of t_k from the held generator sums of the qubit at the surface state; m: all flips on the qubit
CPF_ZHD void chain_surface(float* grad, float px, float py, float pz, int m, float c1, float s1, float c2_, float s2) {
    float P[3] = {m ? -px : px, py, m ? -pz : pz};
    grad[2] = P[2];
    so3_inv(Z_RZ, c2_, s2, P);
    grad[1] = P[0];
    so3_inv(Z_RX, c1, s1, P);
    grad[0] = P[2];
}

}  // namespace zyz
}  // namespace cpf
