// cpflow_amd: the ZYZ form of the entangling blocks (cpf_zyz_math.h, tools/zyz_model.py) on the gate-by-gate layout - one wave
// per sample (4 and 5 qubits; 3 on request), or four waves per sample for launches of at most two waves per SIMD (5 qubits).
//
// Layout Lay<NQ, LW> (cpf_device.h): the first NLR row qubits sit on DPP-reachable lane bits, the others on register bits
// (E = 2^NRR amplitudes per lane); the plan relabels the problem's qubits so that the most-used ones are register qubits.
// A block is  x <- d_ab x  (one merged phase per amplitude of the pair, read from a per-block LDS table; a lane qubit's bit
// selects the lane's entries) followed by the two unnormalised real stages (a, b) <- (a - t b, b + t a):
//     register qubit   2 packed instructions per amplitude pair
//     lane qubit       x += (+-t) x[partner]: two v_fmac_f32_dpp per amplitude (the sign is the lane's bit)
// where the gate-by-gate kernel (cpf_layered.h) applies a CP, two x, two y and two z rotations per block.  Backward: the same
// in reverse on psi and on the cotangent, three generator sums per qubit right after the stages (a lane qubit's partner
// amplitudes are fetched once and serve its sums and psi's un-application), one sum over the |11> amplitudes before conj(d).
// Build and chain phases follow cpf_zyz.h, but every item (a side of a block, a qubit's surface gate) has an OWNER lane that keeps
// its angles, Adam state, (cos, sin) and factors in registers from step to step (zyzl_body).  Up to four samples per workgroup, one
// per wave (KArgs::group, cpf_layered.h).
#pragma once
#include "cpf_zyz.h"

namespace cpf {

// per block: k = 2 a_up + a_dn -> (re, im, -im, im) of d_k (the second pair lets d * x be two compiler-scheduled packed instructions:
// (re, re) is a free half-select of the first pair, (-im, im) would cost an instruction of its own); then t_dn, t_up; two spare
constexpr int ZYZL_TAB_FLOATS = 20;

struct ZyzlArgs { KArgs k; const ZyzSideInfo* sides; int K, n_rot; };

struct ZyzlLds {                                     // float offsets, per sample; [0, 28) = penalty table
    int side, stat, btab, surfm, lfin, fw, red, sums, gsurf, keptl, total;
    __host__ __device__ static constexpr int up4(int x) { return (x + 3) & ~3; }
    // keptl_states: block input states kept in LDS from the forward sweep (zyzl_body SAVEL; 64 lanes x 8 floats each)
    __host__ __device__ ZyzlLds(int K, int nq, int wps = 1, int keptl_states = 0) {
        side = PEN_FLOATS;                     // 2 K x {t, sc, l0, l1, f, -}: the side records
        stat = side + 16 * K;                  // (2 K + nq) x {prev, qubit, -, -, later masks}: static structure (cpf_zyz.h ZyzSideInfo)
        btab = stat + 12 * (2 * K + nq);       // K per-block tables
        surfm = btab + ZYZL_TAB_FLOATS * K;    // nq x 2x2 complex
        lfin = surfm + 8 * nq;                 // nq x (l0, l1)
        fw = up4(lfin + 4 * nq);               // 4 x 64 flip bits
        red = fw + 8;                          // [wps][3]: loss / penalty partial sums of the waves
        sums = red + 16;                       // [wps][K][32]: generator sums, reduce8_rows slabs
        gsurf = sums + wps * K * 32;           // [wps][NBATCH][32]
        keptl = up4(gsurf + wps * ((3 * nq + 7) / 8) * 32);
        total = keptl + keptl_states * 512;
    }
};

// d * x / conj(d) * x with d a (re, im) register pair: exactly two packed instructions (written in C++ the compiler builds the
// (-im, im) pair with an instruction of its own, cpf_device.h cmul)
__device__ __forceinline__ f2 zl_cmul(f2 d, f2 x) {
    f2 o;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=&v"(o) : "v"(d), "v"(x));
    return o;
}
__device__ __forceinline__ f2 zl_cmul_conj(f2 d, f2 x) {
    f2 o;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[1,0,0]" : "=&v"(o) : "v"(d), "v"(x));
    return o;
}

// per-lane constants of the lane row qubits
template <class L>
struct ZlLane { bool lb[L::NLR > 0 ? L::NLR : 1]; float sg[L::NLR > 0 ? L::NLR : 1]; };

// what a lane needs of one block's table: its phases (a register qubit's bit indexes them) and the two tangents, a lane
// qubit's signed by the lane's bit (-t where the bit is clear, +t where it is set)
template <class L, int UP, int DN>
struct ZlTab {
    static constexpr bool UL = L::row_on_lane(UP), DL = L::row_on_lane(DN);
    static constexpr int NU = UL ? 1 : 2, ND = DL ? 1 : 2, NPH = NU * ND;
    f2 ph[NPH], pn[NPH];                  // d_k and (-im, im) of it
    float tu, td;
    __host__ __device__ static constexpr int idx(int e) {           // phase / z-accumulator index of amplitude e
        return (UL ? 0 : ((e >> L::row_bit(UP)) & 1)) * ND + (DL ? 0 : ((e >> L::row_bit(DN)) & 1));
    }
    __host__ __device__ static constexpr bool both_set(int e) {      // register part of the |11> predicate
        return (UL || ((e >> L::row_bit(UP)) & 1)) && (DL || ((e >> L::row_bit(DN)) & 1));
    }
};
// The LDS reads of a block's table are issued one block ahead (a block of 4 amplitudes per lane is ~100 instructions: an exposed
// LDS round trip per block would be a fifth of it): ZlRaw holds what was read, zl_use turns it into the block's ZlTab.
struct ZlRaw { f2 ph[4], pn[4]; float2 tt; };
// d * x / conj(d) * x from the table's two pairs: two packed instructions the compiler is free to interleave with its neighbours'
// (dependent packed instructions issue at half rate: tools/micro/issue_rate.hip)
__device__ __forceinline__ f2 zl_mul(f2 d, f2 n, f2 x) { return pk_fma(n, swp(x), splat(d.x) * x); }
__device__ __forceinline__ f2 zl_mul_conj(f2 d, f2 n, f2 x) { return pk_fma(-n, swp(x), splat(d.x) * x); }
template <class L, int UP, int DN>
__device__ __forceinline__ void zl_load_raw(ZlRaw& r, const float* bt, const ZlLane<L>& ln) {
    using T = ZlTab<L, UP, DN>;
    r.tt = *reinterpret_cast<const float2*>(bt + 16);                  // (t_dn, t_up)
#pragma unroll
    for (int iu = 0; iu < T::NU; iu++)
#pragma unroll
        for (int id = 0; id < T::ND; id++) {
            int au = iu, ad = id;
            if constexpr (T::UL) au = ln.lb[UP] ? 1 : 0;
            if constexpr (T::DL) ad = ln.lb[DN] ? 1 : 0;
            const float4 v = *reinterpret_cast<const float4*>(bt + 4 * (2 * au + ad));
            r.ph[iu * T::ND + id] = f2{v.x, v.y};
            r.pn[iu * T::ND + id] = f2{v.z, v.w};
        }
}
template <class L, int UP, int DN>
__device__ __forceinline__ void zl_use(ZlTab<L, UP, DN>& t, const ZlRaw& r, const ZlLane<L>& ln) {
    using T = ZlTab<L, UP, DN>;
    if constexpr (T::UL) t.tu = ln.sg[UP] * r.tt.y; else t.tu = r.tt.y;
    if constexpr (T::DL) t.td = ln.sg[DN] * r.tt.x; else t.td = r.tt.x;
#pragma unroll
    for (int k = 0; k < T::NPH; k++) { t.ph[k] = r.ph[k]; t.pn[k] = r.pn[k]; }
}
// A lane qubit's stage on two amplitudes: x += (+-t) x[partner], four v_fmac_f32_dpp whose DPP operand is the destination itself.
// (Written in C++ the compiler does not fold a DPP read into an accumulate onto the register it reads: it emitted v_mov_b32_dpp +
// v_fma_f32, four issue slots per amplitude instead of two - 16 of the 45 vector instructions of a forward block with a lane qubit.)
// quad_perm / row_ror move within a row of 16 lanes, which one pass of the instruction reads before it writes.  s_nop 1: the
// amplitudes may come straight from a vector write (two wait states before a DPP read).
#ifndef CPF_ZL_INPLACE
#define CPF_ZL_INPLACE 1         // A/B: 0 = the compiler's v_mov_b32_dpp + v_fma_f32
#endif
template <int BIT, bool NEG>
__device__ __forceinline__ void zl_stage_lane2(f2& x0, f2& x1, float t) {
    static_assert(BIT == 0 || BIT == 1 || BIT == 3, "lane bits one DPP control reaches");
#define CPF_ZL_ST(CTRL, SGN)                                                                                                 \
    asm("s_nop 1\n\t"                                                                                                        \
        "v_fmac_f32_dpp %0, %0, " SGN "%4 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                              \
        "v_fmac_f32_dpp %1, %1, " SGN "%4 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                              \
        "v_fmac_f32_dpp %2, %2, " SGN "%4 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                              \
        "v_fmac_f32_dpp %3, %3, " SGN "%4 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1"                                   \
        : "+v"(x0.x), "+v"(x0.y), "+v"(x1.x), "+v"(x1.y) : "v"(t));
#define CPF_ZL_STS(CTRL) if constexpr (NEG) { CPF_ZL_ST(CTRL, "-") } else { CPF_ZL_ST(CTRL, "") }
    if constexpr (BIT == 0) { CPF_ZL_STS("quad_perm:[1,0,3,2]") }
    else if constexpr (BIT == 1) { CPF_ZL_STS("quad_perm:[2,3,0,1]") }
    else { CPF_ZL_STS("row_ror:8") }
#undef CPF_ZL_STS
#undef CPF_ZL_ST
}
// one unnormalised stage on qubit Q: (a, b) <- (a - t b, b + t a) (NEG: the inverse, t -> -t); t arrives signed for a lane qubit
template <class L, int Q, bool NEG = false>
__device__ __forceinline__ void zl_stage(f2 (&x)[L::E], float t) {
    if constexpr (L::row_on_lane(Q)) {
        constexpr int BIT = L::row_bit(Q);
        if constexpr (CPF_ZL_INPLACE != 0 && (BIT == 0 || BIT == 1 || BIT == 3) && L::E >= 2) {
#pragma unroll
            for (int e = 0; e < L::E; e += 2) zl_stage_lane2<BIT, NEG>(x[e], x[e + 1], t);
        } else {
            const float ts = NEG ? -t : t;
#pragma unroll
            for (int e = 0; e < L::E; e++) {
                const float px = xor_lane<BIT>(x[e].x), py = xor_lane<BIT>(x[e].y);
                x[e].x = fmaf(ts, px, x[e].x);
                x[e].y = fmaf(ts, py, x[e].y);
            }
        }
    } else {
        constexpr int BIT = L::row_bit(Q);
        const float ts = NEG ? -t : t;
#pragma unroll
        for (int e0 = 0; e0 < L::E; e0++) {
            if (e0 & (1 << BIT)) continue;
            const int e1 = e0 | (1 << BIT);
            const f2 a = x[e0], b = x[e1];
            x[e0] = pk_fma(splat(-ts), b, a);
            x[e1] = pk_fma(splat(ts), a, b);
        }
    }
}
template <class L, int UP, int DN>
__device__ __forceinline__ void zl_fwd(f2 (&x)[L::E], const ZlTab<L, UP, DN>& t) {
    using T = ZlTab<L, UP, DN>;
#pragma unroll
    for (int e = 0; e < L::E; e++) x[e] = zl_mul(t.ph[T::idx(e)], t.pn[T::idx(e)], x[e]);
    zl_stage<L, UP>(x, t.tu);
    zl_stage<L, DN>(x, t.td);
}
// this lane's shares of the three generator sums of qubit Q (psi = p, cotangent = c): sum Im(conj(c) sigma_k p).  A lane
// qubit's partner amplitudes are fetched once into pp (the caller un-applies psi's stage with them); its z sum is the
// caller's (it shares the products Im(conj(c) p) with everything else that is diagonal).
template <class L, int Q>
__device__ __forceinline__ void zl_xy_sums(const f2 (&p)[L::E], const f2 (&c)[L::E], const ZlLane<L>& ln, float& px, float& py, f2 (&pp)[L::E]) {
    constexpr int BIT = L::row_bit(Q);
    f2 ax = {0.f, 0.f}, ay = {0.f, 0.f};
    if constexpr (L::row_on_lane(Q)) {
#pragma unroll
        for (int e = 0; e < L::E; e++) {
            pp[e] = f2{xor_lane<BIT>(p[e].x), xor_lane<BIT>(p[e].y)};
            ax = pk_fma(c[e], swp(pp[e]), ax);
            ay = pk_fma(c[e], pp[e], ay);
        }
        px = ax.x - ax.y; py = ln.sg[Q] * (ay.x + ay.y);
    } else {
#pragma unroll
        for (int e0 = 0; e0 < L::E; e0++) {
            if (e0 & (1 << BIT)) continue;
            const int e1 = e0 | (1 << BIT);
            ax = pk_fma(c[e0], swp(p[e1]), ax); ay = pk_fma(-c[e0], p[e1], ay);
            ax = pk_fma(c[e1], swp(p[e0]), ax); ay = pk_fma(c[e1], p[e0], ay);
        }
        px = ax.x - ax.y; py = ay.x + ay.y;
    }
}
// backward pass of one block; d8 = this lane's shares {Px, Py, Pz of the up qubit, the same of the dn qubit, S11, -}
template <class L, int UP, int DN>
__device__ __forceinline__ void zl_bwd(f2 (&p)[L::E], f2 (&c)[L::E], const ZlTab<L, UP, DN>& t, const ZlLane<L>& ln, float (&d8)[8]) {
    using T = ZlTab<L, UP, DN>;
    constexpr int E = L::E;
    // sums right after the stages.  z products by the register bits of the pair
    f2 z[T::NPH];
#pragma unroll
    for (int k = 0; k < T::NPH; k++) z[k] = f2{0.f, 0.f};
#pragma unroll
    for (int e = 0; e < E; e++) z[T::idx(e)] = pk_fma(c[e], swp(p[e]), z[T::idx(e)]);
    float tz[T::NPH];
#pragma unroll
    for (int k = 0; k < T::NPH; k++) tz[k] = z[k].x - z[k].y;
    {   // Pz = (bit clear) - (bit set)
        float u0 = 0.f, u1 = 0.f, v0 = 0.f, v1 = 0.f;
#pragma unroll
        for (int iu = 0; iu < T::NU; iu++)
#pragma unroll
            for (int id = 0; id < T::ND; id++) {
                const float v = tz[iu * T::ND + id];
                if (iu == 0) u0 += v; else u1 += v;
                if (id == 0) v0 += v; else v1 += v;
            }
        if constexpr (T::UL) d8[2] = -ln.sg[UP] * u0; else d8[2] = u0 - u1;
        if constexpr (T::DL) d8[5] = -ln.sg[DN] * v0; else d8[5] = v0 - v1;
    }
    f2 ppu[E], ppd[E];
    zl_xy_sums<L, UP>(p, c, ln, d8[0], d8[1], ppu);
    zl_xy_sums<L, DN>(p, c, ln, d8[3], d8[4], ppd);
    // un-apply the stages (they commute).  psi: a lane qubit's stage first, with the partners fetched above (at most one stage can
    // use them: the second lane qubit of a pair fetches again through its v_fmac_f32_dpp)
    if constexpr (T::DL) {
#pragma unroll
        for (int e = 0; e < E; e++) p[e] = pk_fma(splat(-t.td), ppd[e], p[e]);
        zl_stage<L, UP, true>(p, t.tu);
    } else if constexpr (T::UL) {
#pragma unroll
        for (int e = 0; e < E; e++) p[e] = pk_fma(splat(-t.tu), ppu[e], p[e]);
        zl_stage<L, DN, true>(p, t.td);
    } else {
        zl_stage<L, DN, true>(p, t.td);
        zl_stage<L, UP, true>(p, t.tu);
    }
    zl_stage<L, DN, true>(c, t.td);
    zl_stage<L, UP, true>(c, t.tu);
    // the |11> sum before conj(d) (both matrices carry 1 / d_0 here: cpf_zyz_math.h chain_cp)
    {
        f2 cc = {0.f, 0.f};
#pragma unroll
        for (int e = 0; e < E; e++) if (T::both_set(e)) cc = pk_fma(c[e], swp(p[e]), cc);
        float s11 = cc.x - cc.y;
        if constexpr (T::UL) s11 = ln.lb[UP] ? s11 : 0.f;
        if constexpr (T::DL) s11 = ln.lb[DN] ? s11 : 0.f;
        d8[6] = s11; d8[7] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < E; e++) { p[e] = zl_mul_conj(t.ph[T::idx(e)], t.pn[T::idx(e)], p[e]); c[e] = zl_mul_conj(t.ph[T::idx(e)], t.pn[T::idx(e)], c[e]); }
}

// The same for a block whose input state the forward sweep kept (zyzl_body, SAVE): p = the state right after the block (kept as the
// next block's input, or the sweep's final state), q = the state in front of it.  Nothing is un-applied from psi - its two stages and
// its conj(d), 16 of the block's ~110 instructions - and the |11> sum is taken AFTER conj(d), from the cotangent in front of the
// block and q: both carry the block's real scale d_0 there instead of 1 / d_0, so the sum is d_0^2 times the one zl_bwd forms and
// chain_cp gets d_0 = 1 for such a block.  (The recomputed state equals q up to rounding: the unnormalised stages' (1 + t^2)
// factors and |d|^2 cancel.)
template <class L, int UP, int DN>
__device__ __forceinline__ void zl_bwd_saved(const f2 (&p)[L::E], const f2 (&q)[L::E], f2 (&c)[L::E], const ZlTab<L, UP, DN>& t, const ZlLane<L>& ln, float (&d8)[8]) {
    using T = ZlTab<L, UP, DN>;
    constexpr int E = L::E;
    f2 z[T::NPH];
#pragma unroll
    for (int k = 0; k < T::NPH; k++) z[k] = f2{0.f, 0.f};
#pragma unroll
    for (int e = 0; e < E; e++) z[T::idx(e)] = pk_fma(c[e], swp(p[e]), z[T::idx(e)]);
    float tz[T::NPH];
#pragma unroll
    for (int k = 0; k < T::NPH; k++) tz[k] = z[k].x - z[k].y;
    {
        float u0 = 0.f, u1 = 0.f, v0 = 0.f, v1 = 0.f;
#pragma unroll
        for (int iu = 0; iu < T::NU; iu++)
#pragma unroll
            for (int id = 0; id < T::ND; id++) {
                const float v = tz[iu * T::ND + id];
                if (iu == 0) u0 += v; else u1 += v;
                if (id == 0) v0 += v; else v1 += v;
            }
        if constexpr (T::UL) d8[2] = -ln.sg[UP] * u0; else d8[2] = u0 - u1;
        if constexpr (T::DL) d8[5] = -ln.sg[DN] * v0; else d8[5] = v0 - v1;
    }
    f2 ppu[E], ppd[E];
    zl_xy_sums<L, UP>(p, c, ln, d8[0], d8[1], ppu);
    zl_xy_sums<L, DN>(p, c, ln, d8[3], d8[4], ppd);
    zl_stage<L, DN, true>(c, t.td);
    zl_stage<L, UP, true>(c, t.tu);
#pragma unroll
    for (int e = 0; e < E; e++) c[e] = zl_mul_conj(t.ph[T::idx(e)], t.pn[T::idx(e)], c[e]);
    {
        f2 cc = {0.f, 0.f};
#pragma unroll
        for (int e = 0; e < E; e++) if (T::both_set(e)) cc = pk_fma(c[e], swp(q[e]), cc);
        float s11 = cc.x - cc.y;
        if constexpr (T::UL) s11 = ln.lb[UP] ? s11 : 0.f;
        if constexpr (T::DL) s11 = ln.lb[DN] ? s11 : 0.f;
        d8[6] = s11; d8[7] = 0.f;
    }
}

// x[e] = base * prod over the register qubits of f_q[bit of e] (register bit rb holds qubit NLR + rb)
template <class L, class F>
__device__ __forceinline__ void zl_product(f2 (&x)[L::E], f2 base, const F& fac) {
    x[0] = base;
    static_for([&](auto bb) __attribute__((always_inline)) {
        constexpr int RB = decltype(bb)::value;
        const f2 f0 = fac(L::NLR + RB, 0), f1 = fac(L::NLR + RB, 1);
#pragma unroll
        for (int e = 0; e < (1 << RB); e++) { x[e | (1 << RB)] = zl_cmul(f1, x[e]); x[e] = zl_cmul(f0, x[e]); }
    }, typename make_iseq<L::NRR>::type{});
}

// The step's three wave sums (Re T, Im T, penalty) returned wave-uniform.  ONE v_permlane32_swap folds the wave's halves for T -
// lanes 0..31 then hold re(l) + re(l + 32), lanes 32..63 im(l - 32) + im(l) - so its four row stages and one row_bcast run on one
// register (cpf_zyz3.h z3_allsum); the penalty's chain alternates with it, each filling one of the other's two DPP wait states:
// 11 DPP instructions and 6 s_nop instead of the 18 + 15 of three wave_sum calls.  Fixed order: run-to-run deterministic.
__device__ __forceinline__ void zl_wave_sum3(float& re, float& im, float& pen, bool with_pen) {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(re), __float_as_uint(im), false, false);
    float s = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
#define CPF_ZL_T " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
    if (with_pen) {
        float p = pen;
        asm("s_nop 1\n\t"
            "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2]" CPF_ZL_T "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2]" CPF_ZL_T "s_nop 0\n\t"
            "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1]" CPF_ZL_T "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1]" CPF_ZL_T "s_nop 0\n\t"
            "v_add_f32_dpp %0, %0, %0 row_ror:4" CPF_ZL_T "v_add_f32_dpp %1, %1, %1 row_ror:4" CPF_ZL_T "s_nop 0\n\t"
            "v_add_f32_dpp %0, %0, %0 row_ror:8" CPF_ZL_T "v_add_f32_dpp %1, %1, %1 row_ror:8" CPF_ZL_T "s_nop 0\n\t"
            "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
            "v_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
            "s_nop 1\n\t"
            "v_add_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf"
            : "+v"(s), "+v"(p));
        pen = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p), 63));
    } else {
        asm("s_nop 1\n\t"
            "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2]" CPF_ZL_T "s_nop 1\n\t"
            "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1]" CPF_ZL_T "s_nop 1\n\t"
            "v_add_f32_dpp %0, %0, %0 row_ror:4" CPF_ZL_T "s_nop 1\n\t"
            "v_add_f32_dpp %0, %0, %0 row_ror:8" CPF_ZL_T "s_nop 1\n\t"
            "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf"
            : "+v"(s));
        pen = 0.f;
    }
#undef CPF_ZL_T
    re = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), 31));
    im = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), 63));
}

// Items: one per side (i = 2 b + side < 2 K), then one per qubit (i = 2 K + q).  Item i belongs to thread i mod THREADS of the
// sample's 64 or 256, slot i / THREADS (NSLOT slots per lane, a template argument), and that lane OWNS the item's angles - a side's rotation angles (and, for side 0, the
// block's CP angle), a qubit's three surface angles: Adam state, (cos, sin), the side's factors and the gradients the chain phase
// produces all stay in the owner's registers; only what other lanes need crosses LDS (the side records, the per-block tables, the
// surface matrices, the reduced sums).
// LW: log2 of the waves per sample - 0 (one wave; up to four samples per workgroup) or 2 (four waves: the columns are split over
// the waves, for launches with at most one sample per CU; the items spread over all 256 lanes).
// NSLOT_SAVE = NSLOT + 16 SAVE.  SAVE > 0 (one wave per sample, 4 amplitudes per lane; the host guarantees K = a whole number of
// layers >= SAVE): the input states of the blocks of the last SAVE layers stay in registers from the forward sweep (8 registers per
// block: a launch of this kernel has at most two waves per SIMD = 256 registers per lane), the backward sweep's first SAVE layers take
// psi from there instead of un-applying the blocks from it (zl_bwd_saved) - config 4: 15 of 24 blocks, 16 of ~110 instructions each.
template <int NQ, int LW, int ROT, int NSLOT_SAVE, int... UD>
__device__ __forceinline__ void zyzl_body(const ZyzlArgs& za) {
    constexpr int NSLOT = NSLOT_SAVE & 15, SAVE = (NSLOT_SAVE >> 4) & 15, SAVEL = (NSLOT_SAVE >> 8) & 15;
    using L = Lay<NQ, LW>;
    constexpr int WPS = L::WPS, THREADS = L::THREADS;
    using PL = PairList<UD...>;
    constexpr int LL = sizeof...(UD) / 2;
    static_assert(LL > 0 && ROT != 0, "the ZYZ-form kernel is specialised on the layer and the rotation string");
    constexpr int E = L::E, NS = 3 * NQ, NBATCH = (NS + 7) / 8;
    constexpr int NSAVE = SAVE * LL;                       // blocks K - NSAVE .. K - 1: input states in registers
    constexpr int NSL = SAVEL * LL;                        // blocks K - NSAVE - NSL .. K - NSAVE - 1: input states in LDS (when the registers are full:
                                                           // two ds_write_b128 forward and two ds_read_b128 backward instead of 16 instructions)
    static_assert(SAVEL == 0 || (SAVE > 0 && E == 4), "LDS-kept states come in front of register-kept ones");
    static_assert(SAVE == 0 || (WPS == 1 && E <= 4 && NSAVE * E * 2 <= 160), "kept states: one wave per sample, at most 160 registers of them");
    constexpr int NR = ROT & 15, ENT = (ROT >> 16) & 3, NBP = 2 * NR + (ENT == 0 ? 1 : 0);
    static_assert(NR >= 1 && NR <= 3 && ENT <= 1, "1..3 rotation letters, CP or CZ blocks");
    const KArgs& a = za.k;
    extern __shared__ __attribute__((aligned(16))) float smem_wg[];
    const int P = a.P, K = za.K;
    const ZyzlLds lds(K, NQ, WPS, NSL);
    const int sub = WPS == 1 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) : 0;
    const int tid = WPS == 1 ? ((int)threadIdx.x & 63) : (int)threadIdx.x, lane = tid & 63;
    const int wave = WPS == 1 ? 0 : __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t s_idx = (int64_t)blockIdx.x * (a.group > 1 && WPS == 1 ? a.group : 1) + sub;
    if (WPS == 1 && s_idx >= a.nsamples) return;
    float* smem = smem_wg + sub * lds.total;
    float* red = smem + lds.red;
    float* sums_w = smem + lds.sums + wave * K * 32;
    float* gsurf_w = smem + lds.gsurf + wave * NBATCH * 32;
    float4* sidev = reinterpret_cast<float4*>(smem + lds.side);
    unsigned long long* fw = reinterpret_cast<unsigned long long*>(smem + lds.fw);
    const float* init = a.init + s_idx * P;
    const uint8_t* frozen = a.frozen ? a.frozen + s_idx * P : nullptr;
    float* out_best = a.out_params + (s_idx * 2 + 1) * P;
    const int col = L::col_index(lane, wave);
    const int dst_lane = reduce8_dst_lane(lane);
    const int nw = (2 * K + 63) >> 6, NI = 2 * K + NQ;
    ZlLane<L> ln;
#pragma unroll
    for (int i = 0; i < L::NLR; i++) { ln.lb[i] = (lane >> L::row_lane_bit(i)) & 1; ln.sg[i] = ln.lb[i] ? 1.f : -1.f; }
    uint32_t toff_i = (uint32_t)(L::row_index(lane, 0) * L::D + col) * 8u;
    int lane_i = lane, col_i = col;
    const char* tbase = reinterpret_cast<const char*>(a.target);
    // 4 qubits (4 amplitudes per lane): the lane's target elements stay in registers for the whole run (8 registers of a kernel that may
    // use 256: two waves per SIMD at most) instead of four global loads in the loss and four more before the backward sweep of every step
    constexpr bool TREG = E <= 4 && SAVE == 0;      // (the instance that keeps states has no registers to spare: it re-reads)
    float2 treg[TREG ? E : 1];
    if constexpr (TREG) {
#pragma unroll
        for (int e = 0; e < E; e++) treg[e] = *reinterpret_cast<const float2*>(tbase + (toff_i + (uint32_t)((L::row_index(0, e)) * L::D * 8)));
    }
    auto target_at = [&](int e) __attribute__((always_inline)) {
        if constexpr (TREG) return treg[e];
        else return *reinterpret_cast<const float2*>(tbase + (toff_i + (uint32_t)((L::row_index(0, e)) * L::D * 8)));
    };
    penalty_table_init(smem, a, tid);

    // ---- the items of this lane
    struct Item {
        float th[4], bt[4], mm[4], vv[4], g[4], nlr[4], cs[4][2];     // angle k; nlr: minus the learning rate, 0 for an absent or frozen angle: rotation letter k (side) / surface angle k (qubit); k = 3: the CP angle (side 0)
        int pj[4];                                     // parameter index, -1 = none
        bool frz[4];
        int kind;                                      // 0 = side, 1 = qubit, 2 = none
        int prev;                                      // side: previous side on the qubit; qubit: last side on it
        zyz::Side sd;
        int cacb; float d0, psl, pval;                 // side 0: where the CP phase sits, the block's scale, r * penalty slope, penalty
        int m;                                         // qubit: all flips on it
    } itm[NSLOT];
    // One wave per sample and one item per lane (2 K + n <= 64: config 4): the flip bits of all sides are ONE ballot, which stays in
    // scalar registers, and the masks of the later sides on the qubit (this item's and, for side 0, its block's other side's) in this
    // lane's - the parities of build B are an AND and a bit count each, without the LDS round trip and the word loops of the general form.
    constexpr bool ONE = WPS == 1 && NSLOT == 1;
    unsigned long long flips_all = 0, mask_own = 0, mask_nb = 0;
#pragma unroll
    for (int s = 0; s < NSLOT; s++) {
        Item& t = itm[s];
        const int i = tid + THREADS * s;
        t.kind = i < 2 * K ? 0 : (i < NI ? 1 : 2);
        t.prev = -1; t.cacb = 0; t.d0 = 1.f; t.psl = 0.f; t.pval = 0.f; t.m = 0;
        t.sd = zyz::Side{};
#pragma unroll
        for (int k = 0; k < 4; k++) { t.pj[k] = -1; t.th[k] = 0.f; t.bt[k] = 0.f; t.nlr[k] = 0.f; t.mm[k] = 0.f; t.vv[k] = 0.f; t.g[k] = 0.f; t.cs[k][0] = 1.f; t.cs[k][1] = 0.f; t.frz[k] = false; }
        if (t.kind == 0) {
            const int b = i >> 1, side = i & 1;
#pragma unroll
            for (int k = 0; k < NR; k++) t.pj[k] = NS + NBP * b + 2 * k + side;
            if (ENT == 0 && side == 0) t.pj[3] = NS + NBP * b + 2 * NR;
            t.prev = za.sides[i].prev;
        } else if (t.kind == 1) {
            const int q = i - 2 * K;
            for (int p = 0; p < NQ; p++)               // the surface gates of kernel qubit q carry the angles of the problem qubit relabelled to q
                if (a.pgate[3 * p] == 3 * q) { t.pj[0] = 3 * p; t.pj[1] = 3 * p + 1; t.pj[2] = 3 * p + 2; }
            t.prev = za.sides[i].prev;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (t.pj[k] >= 0) {
                t.th[k] = init[t.pj[k]];
                t.bt[k] = t.th[k];                     // the best iterate's angles stay with their owner until the run ends
                t.frz[k] = frozen && frozen[t.pj[k]];
                t.nlr[k] = t.frz[k] ? 0.f : -a.lr;
                a.out_params[(s_idx * 2) * P + t.pj[k]] = t.th[k];
            }
        }
        // static structure other lanes need: {prev, -, later masks} per item
        if (t.kind != 2) {
            const ZyzSideInfo si = za.sides[i];
            int* st = reinterpret_cast<int*>(smem + lds.stat + 12 * i);
            if constexpr (ONE) {
                mask_own = si.later[0];
                if (t.kind == 0 && (i & 1) == 0) mask_nb = za.sides[i + 1].later[0];
            }
            st[0] = si.prev; st[1] = si.qubit;
            for (int k = 0; k < 4; k++) { st[4 + 2 * k] = (int)(si.later[k] & 0xffffffffull); st[5 + 2 * k] = (int)(si.later[k] >> 32); }
        }
    }
    lds_sync<WPS>();
    auto load_side = [&](int w) {
        const float4 s0 = sidev[2 * w], s1 = sidev[2 * w + 1];
        zyz::Side sd;
        sd.t = s0.x; sd.sc = s0.y; sd.l0 = zyz::c2(s0.z, s0.w); sd.l1 = zyz::c2(s1.x, s1.y); sd.f = __float_as_int(s1.z); sd.m = 0;
        return sd;
    };
    auto parity_of = [&](int i, bool neighbour = false) {            // flips of the sides item i's mask names (i: this lane's item or, neighbour, the next one)
        if constexpr (ONE) return (int)(__popcll(flips_all & (neighbour ? mask_nb : mask_own)) & 1);
        const unsigned long long* mask = reinterpret_cast<const unsigned long long*>(smem + lds.stat + 12 * i + 4);
        int n = 0;
        for (int k = 0; k < nw; k++) n += __popcll(fw[k] & mask[k]);
        return n & 1;
    };
    auto slab_sum = [&](int base, int stride_w, int idx, int v) {          // over the waves' slabs and their four row partials
        float sum = 0.f;
#pragma unroll
        for (int wv = 0; wv < WPS; wv++) {
            const float4 q = *reinterpret_cast<const float4*>(smem + base + wv * stride_w + 32 * idx + 4 * v);
            sum += (q.x + q.y) + (q.z + q.w);
        }
        return sum;
    };

    f2 psi[E], cot[E];
    f2 kept[NSAVE > 0 ? NSAVE : 1][E];
    float4* keptl = reinterpret_cast<float4*>(smem + lds.keptl) + lane;          // [state][half][lane]: conflict-free ds_*_b128
    float best = 0.f, first = 0.f;
    const int iters = a.T > 0 ? a.T : 1;
    for (int it = 0; it < iters; it++) {
        CPF_MARK(top);
        // Adam closes the PREVIOUS step here; the reciprocals of its bias corrections come from the host (cpflow_amd.hip: ibias[t] = 1 / bias[t])
        const float2 ibc = const_load_f2(a.ibias, a.T > 0 ? (it > 0 ? it - 1 : 0) : 0);
        asm volatile("" : "+v"(toff_i), "+v"(lane_i), "+v"(col_i));
        // ================= owner + build A: Adam on the gradients of the previous step, (cos, sin), the side's factors
#pragma unroll
        for (int s = 0; s < NSLOT; s++) {
            Item& t = itm[s];
            bool flip = false;
            if (t.kind != 2) {
                // Two angles per packed instruction and no branch on the slot: an absent or frozen angle has the step size 0 (nlr) and,
                // at the first iterate, every gradient is 0 - m and v of such a slot move, its angle does not.  The same operations in
                // the same order as the scalar form (optax 0.1.1 scale_by_adam, optimization.py:61-75).
                if (it > 0) {
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        f2 g = {t.g[2 * h], t.g[2 * h + 1]};
                        if (h == 1 && ENT == 0 && a.use_reg) g.y += t.psl;
                        const f2 m = pk_fma(splat(a.b1), f2{t.mm[2 * h], t.mm[2 * h + 1]}, splat(a.omb1) * g);
                        const f2 v = pk_fma(splat(a.b2), f2{t.vv[2 * h], t.vv[2 * h + 1]}, splat(a.omb2) * (g * g));
                        t.mm[2 * h] = m.x; t.mm[2 * h + 1] = m.y; t.vv[2 * h] = v.x; t.vv[2 * h + 1] = v.y;
                        const f2 hh = v * splat(ibc.y);
                        const f2 d = f2{__builtin_amdgcn_sqrtf(hh.x), __builtin_amdgcn_sqrtf(hh.y)} + splat(a.eps);
                        const f2 u = (m * splat(ibc.x)) * f2{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
                        const f2 th = pk_fma(f2{t.nlr[2 * h], t.nlr[2 * h + 1]}, u, f2{t.th[2 * h], t.th[2 * h + 1]});
                        t.th[2 * h] = th.x; t.th[2 * h + 1] = th.y;
                    }
                }
                {   // (cos, sin) of the four angles on packed arithmetic, two per call (bit-identical to fast_sincos per half); slot 3
                    // is a CP angle (full angle), the others are half angles; an absent angle is 0 -> (1, 0)
                    f2 s01, c01, s23, c23;
                    fast_sincos2(f2{0.5f * t.th[0], 0.5f * t.th[1]}, s01, c01);
                    fast_sincos2(f2{0.5f * t.th[2], t.th[3]}, s23, c23);
                    t.cs[0][0] = c01.x; t.cs[0][1] = s01.x; t.cs[1][0] = c01.y; t.cs[1][1] = s01.y;
                    t.cs[2][0] = c23.x; t.cs[2][1] = s23.x; t.cs[3][0] = c23.y; t.cs[3][1] = s23.y;
                }
                if (t.kind == 0) {
                    if (ENT == 0 && t.pj[3] >= 0 && a.use_reg) {
                        float sl;
                        t.pval = penalty_lds(smem, a.pen_kind, t.th[3], sl);
                        t.psl = a.r * sl;
                    }
                    const int side = (tid + THREADS * s) & 1;
                    auto rrec = [&](int idx, float& c, float& sn, int& kind) __attribute__((always_inline)) {
                        const int k = (idx - 1) >> 1;
                        c = t.cs[k][0]; sn = t.cs[k][1]; kind = 0;
                    };
                    const int m_keep = t.sd.m;
                    t.sd = zyz::side_build<ROT>(rrec, NR, side);
                    t.sd.m = m_keep;
                    const int w = tid + THREADS * s;
                    sidev[2 * w] = make_float4(t.sd.t, t.sd.sc, t.sd.l0.re, t.sd.l0.im);
                    sidev[2 * w + 1] = make_float4(t.sd.l1.re, t.sd.l1.im, __int_as_float(t.sd.f), 0.f);
                    flip = t.sd.f != 0;
                }
            }
            const unsigned long long bits = __builtin_amdgcn_ballot_w64(flip);
            if constexpr (ONE) flips_all = bits;
            else if (lane == 0) fw[s * WPS + wave] = bits;
        }
        lds_sync<WPS>();
        CPF_MARK(buildB);
        // ================= build B: pending-flip parities; side 0: the block's table; qubit: surface matrix, final left factor
#pragma unroll
        for (int s = 0; s < NSLOT; s++) {
            Item& t = itm[s];
            const int i = tid + THREADS * s;
            if (t.kind == 0) {
                t.sd.m = parity_of(i);
                if ((i & 1) == 0) {
                    const int b = i >> 1;
                    zyz::Side sd = load_side(i + 1), pu = t.sd, pd = sd;
                    sd.m = parity_of(i + 1, true);
                    const int wu = t.prev, wd = reinterpret_cast<const int*>(smem + lds.stat + 12 * (i + 1))[0];
                    if (wu >= 0) { pu = load_side(wu); pu.m = t.sd.m ^ t.sd.f; }
                    if (wd >= 0) { pd = load_side(wd); pd.m = sd.m ^ sd.f; }
                    float out[10];
                    t.cacb = zyz::block_table(out, t.sd, sd, wu >= 0, pu, wd >= 0, pd, ENT == 0 ? t.cs[3][0] : -1.f, ENT == 0 ? t.cs[3][1] : 0.f, false);
                    t.d0 = out[0];
                    float* bt = smem + lds.btab + ZYZL_TAB_FLOATS * b;
#pragma unroll
                    for (int k = 0; k < 4; k++) *reinterpret_cast<float4*>(bt + 4 * k) = make_float4(out[2 * k], out[2 * k + 1], -out[2 * k + 1], out[2 * k + 1]);
                    *reinterpret_cast<float2*>(bt + 16) = make_float2(out[8], out[9]);
                }
            } else if (t.kind == 1) {
                const int q = i - 2 * K;
                t.m = parity_of(i);
                float sm[8];
                zyz::surface_matrix(sm, t.cs[0][0], t.cs[0][1], t.cs[1][0], t.cs[1][1], t.cs[2][0], t.cs[2][1], t.m);
                *reinterpret_cast<float4*>(smem + lds.surfm + 8 * q) = make_float4(sm[0], sm[1], sm[2], sm[3]);
                *reinterpret_cast<float4*>(smem + lds.surfm + 8 * q + 4) = make_float4(sm[4], sm[5], sm[6], sm[7]);
                float4 lf = make_float4(1.f, 0.f, 1.f, 0.f);
                if (t.prev >= 0) { const zyz::Side sl = load_side(t.prev); lf = make_float4(sl.l0.re, sl.l0.im, sl.l1.re, sl.l1.im); }
                *reinterpret_cast<float4*>(smem + lds.lfin + 4 * q) = lf;
            }
        }
        lds_sync<WPS>();

        CPF_MARK(surface);
        // ================= forward: the (flipped) product state of the surface gates, then the blocks
        {
            auto sfac = [&](int q, int r) __attribute__((always_inline)) {
                const int cb = (col_i >> (NQ - 1 - q)) & 1;
                const float2 v = *reinterpret_cast<const float2*>(smem + lds.surfm + 8 * q + 4 * r + 2 * cb);
                return f2{v.x, v.y};
            };
            f2 base = {1.f, 0.f};
            static_for([&](auto qq) __attribute__((always_inline)) {
                constexpr int Q = decltype(qq)::value;
                const f2 v = sfac(Q, (lane_i >> L::row_lane_bit(Q)) & 1);
                base = Q == 0 ? v : zl_cmul(v, base);
            }, typename make_iseq<L::NLR>::type{});
            zl_product<L>(psi, base, sfac);
        }
        CPF_MARK(fwd);
        {
            int b = 0;
            ZlRaw nxt;
            // The table the next read takes: a running address (one add per layer, the blocks of a layer at immediate offsets) instead of
            // a clamped block index per read (s_cmp, s_cselect, s_mulk, s_add, v_mov: 5 of a forward block's 45 instructions).  The read
            // behind the last block takes the 20 floats that follow the tables (the surface matrices: inside the sample's LDS) and is
            // never used.
            const float* btn = smem + lds.btab;
            zl_load_raw<L, PL::v[0], PL::v[1]>(nxt, btn, ln);
            btn += ZYZL_TAB_FLOATS;
            auto fwd_step = [&](auto jj) __attribute__((always_inline)) {
                constexpr int J = decltype(jj)::value, JN = (J + 1) % LL;
                const ZlRaw cur = nxt;
                zl_load_raw<L, PL::v[2 * JN], PL::v[2 * JN + 1]>(nxt, btn, ln);
                btn += ZYZL_TAB_FLOATS;
                __builtin_amdgcn_sched_barrier(0);               // the reads stay here (the scheduler would sink them to their first use)
                ZlTab<L, PL::v[2 * J], PL::v[2 * J + 1]> t;
                zl_use(t, cur, ln);
                zl_fwd<L, PL::v[2 * J], PL::v[2 * J + 1]>(psi, t);
                gate_fence<L>();
                b++;
            };
            for (int full = K / LL - SAVE - SAVEL; full > 0; full--) static_for(fwd_step, typename make_iseq<LL>::type{});
            if constexpr (SAVEL > 0) {
                static_for([&](auto ii) __attribute__((always_inline)) {
                    constexpr int I = decltype(ii)::value;
                    keptl[(2 * I) * 64] = make_float4(psi[0].x, psi[0].y, psi[1].x, psi[1].y);          // the state in front of block K - NSAVE - NSL + I
                    keptl[(2 * I + 1) * 64] = make_float4(psi[2].x, psi[2].y, psi[3].x, psi[3].y);
                    fwd_step(ic<I % LL>{});
                }, typename make_iseq<NSL>::type{});
            }
            if constexpr (SAVE > 0) {
                static_for([&](auto ii) __attribute__((always_inline)) {
                    constexpr int I = decltype(ii)::value;
#pragma unroll
                    for (int e = 0; e < E; e++) kept[I][e] = psi[e];          // the state in front of block K - NSAVE + I
                    fwd_step(ic<I % LL>{});
                }, typename make_iseq<NSAVE>::type{});
            } else static_for([&](auto jj) __attribute__((always_inline)) { if (b < K) fwd_step(jj); }, typename make_iseq<LL>::type{});
        }

        CPF_MARK(loss);
        // ================= loss and penalty; the final diagonal meets the target: Vd = conj(dfin) V in the cotangent's registers
        float acc[3] = {0.f, 0.f, 0.f};
        if (a.loss_kind == 2) {
#pragma unroll
            for (int e = 0; e < E; e++) {
                const float2 t = target_at(e);
                acc[0] = fmaf(fmaf(t.x, t.x, t.y * t.y), fmaf(psi[e].x, psi[e].x, psi[e].y * psi[e].y), acc[0]);
            }
        } else {
            auto lfac = [&](int q, int r) __attribute__((always_inline)) {
                const float2 v = *reinterpret_cast<const float2*>(smem + lds.lfin + 4 * q + 2 * r);
                return f2{v.x, v.y};
            };
            f2 base = {1.f, 0.f};
            static_for([&](auto qq) __attribute__((always_inline)) {
                constexpr int Q = decltype(qq)::value;
                const f2 v = lfac(Q, (lane_i >> L::row_lane_bit(Q)) & 1);
                base = Q == 0 ? v : zl_cmul(v, base);
            }, typename make_iseq<L::NLR>::type{});
            zl_product<L>(cot, base, lfac);
            f2 tt = {0.f, 0.f};
#pragma unroll
            for (int e = 0; e < E; e++) {
                const float2 t = target_at(e);
                cot[e] = zl_cmul_conj(cot[e], f2{t.x, t.y});
                tt = pk_fma(splat(cot[e].x), psi[e], tt);
                tt = pk_fma(f2{cot[e].y, -cot[e].y}, swp(psi[e]), tt);
            }
            acc[0] = tt.x; acc[1] = tt.y;
        }
        if (a.use_reg) {
#pragma unroll
            for (int s = 0; s < NSLOT; s++) acc[2] += itm[s].pval;
        }
        zl_wave_sum3(acc[0], acc[1], acc[2], a.use_reg != 0);
        if constexpr (WPS > 1) {                 // fixed order: deterministic
            lds_sync<WPS>();
            if (lane == 0) { red[wave * 3] = acc[0]; red[wave * 3 + 1] = acc[1]; red[wave * 3 + 2] = acc[2]; }
            lds_sync<WPS>();
#pragma unroll
            for (int k = 0; k < 3; k++) { float sum = 0.f; for (int w = 0; w < WPS; w++) sum += red[w * 3 + k]; acc[k] = sum; }
        }
        CPF_MARK(track);
        const float tr = acc[0], ti = acc[1];
        const float ab = sqrtf(fmaf(tr, tr, ti * ti));
        const float loss = (a.loss_kind == 2) ? 1.f - tr * a.inv_dim2 : 1.f - ab * ab * a.inv_dim2;
        const float reg = a.use_reg ? a.r * acc[2] : 0.f;
        const float regloss = loss + reg;
        bool improved = false;
        if (it == 0) { first = regloss; best = regloss; }
        else if (regloss < best) { best = regloss; improved = true; }
#pragma unroll
        for (int s = 0; s < NSLOT; s++)
#pragma unroll
            for (int k = 0; k < 4; k++) itm[s].bt[k] = improved ? itm[s].th[k] : itm[s].bt[k];       // a select, not a store per improvement
        if (a.hist_params) {
#pragma unroll
            for (int s = 0; s < NSLOT; s++)
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int j = itm[s].pj[k];
                    if (j >= 0) a.hist_params[(s_idx * a.T + it) * P + j] = itm[s].th[k];
                }
        }
        if (a.hist_loss && tid == 0) a.hist_loss[s_idx * a.T + it] = regloss;
        if (a.T == 0) break;
        if (it + 1 == iters) break;                    // the last step's gradient would move the angles past the last evaluated iterate

        CPF_MARK(bwdinit);
        // ================= backward
        if (a.loss_kind == 2) {
#pragma unroll
            for (int e = 0; e < E; e++) {
                const float2 t = target_at(e);
                cot[e] = splat(-a.inv_dim2 * fmaf(t.x, t.x, t.y * t.y)) * psi[e];
            }
        } else {
            const f2 f = {-a.inv_dim2 * tr, -a.inv_dim2 * ti};
#pragma unroll
            for (int e = 0; e < E; e++) cot[e] = zl_cmul(f, cot[e]);
        }
        CPF_MARK(bwd);
        {
            int b = K - 1;
            ZlRaw nxt;
            // running addresses here too; the read in front of block 0 takes the 20 floats before the tables (static structure: inside
            // the sample's LDS) and is never used
            const float* btn = smem + lds.btab + ZYZL_TAB_FLOATS * b;
            float* sums_b = sums_w + 32 * b;
            {
                const int jt = (K - 1) % LL;                     // the last block's layer position
                static_for([&](auto jj) __attribute__((always_inline)) {
                    constexpr int J = decltype(jj)::value;
                    if (J == jt) zl_load_raw<L, PL::v[2 * J], PL::v[2 * J + 1]>(nxt, btn, ln);
                }, typename make_iseq<LL>::type{});
                btn -= ZYZL_TAB_FLOATS;
            }
            auto bwd_step = [&](auto jj) __attribute__((always_inline)) {
                constexpr int J = LL - 1 - decltype(jj)::value, JN = (J + LL - 1) % LL;
                const ZlRaw cur = nxt;
                zl_load_raw<L, PL::v[2 * JN], PL::v[2 * JN + 1]>(nxt, btn, ln);
                btn -= ZYZL_TAB_FLOATS;
                __builtin_amdgcn_sched_barrier(0);
                ZlTab<L, PL::v[2 * J], PL::v[2 * J + 1]> t;
                zl_use(t, cur, ln);
                float d8[8];
                zl_bwd<L, PL::v[2 * J], PL::v[2 * J + 1]>(psi, cot, t, ln, d8);
                reduce8_rows<true>(d8, sums_b, lane, dst_lane);
                sums_b -= 32;
                gate_fence<L>();
                b--;
            };
            if constexpr (SAVE > 0) {
                static_for([&](auto ii) __attribute__((always_inline)) {
                    constexpr int I = NSAVE - 1 - decltype(ii)::value, J = I % LL, JN = (J + LL - 1) % LL;
                    const ZlRaw cur = nxt;
                    zl_load_raw<L, PL::v[2 * JN], PL::v[2 * JN + 1]>(nxt, btn, ln);
                    btn -= ZYZL_TAB_FLOATS;
                    __builtin_amdgcn_sched_barrier(0);
                    ZlTab<L, PL::v[2 * J], PL::v[2 * J + 1]> t;
                    zl_use(t, cur, ln);
                    float d8[8];
                    if constexpr (I == NSAVE - 1) zl_bwd_saved<L, PL::v[2 * J], PL::v[2 * J + 1]>(psi, kept[I], cot, t, ln, d8);
                    else zl_bwd_saved<L, PL::v[2 * J], PL::v[2 * J + 1]>(kept[I + 1 < NSAVE ? I + 1 : I], kept[I], cot, t, ln, d8);
                    reduce8_rows<true>(d8, sums_b, lane, dst_lane);
                    sums_b -= 32;
                    gate_fence<L>();
                    b--;
                }, typename make_iseq<NSAVE>::type{});
#pragma unroll
                for (int e = 0; e < E; e++) psi[e] = kept[0][e];             // the state in front of block K - NSAVE: the other layers un-apply from here
                if constexpr (SAVEL > 0) {
                    static_for([&](auto ii) __attribute__((always_inline)) {
                        constexpr int I = NSL - 1 - decltype(ii)::value, J = I % LL, JN = (J + LL - 1) % LL;
                        const float4 q0 = keptl[(2 * I) * 64], q1 = keptl[(2 * I + 1) * 64];
                        const ZlRaw cur = nxt;
                        zl_load_raw<L, PL::v[2 * JN], PL::v[2 * JN + 1]>(nxt, btn, ln);
                        btn -= ZYZL_TAB_FLOATS;
                        __builtin_amdgcn_sched_barrier(0);
                        ZlTab<L, PL::v[2 * J], PL::v[2 * J + 1]> t;
                        zl_use(t, cur, ln);
                        const f2 q[E] = {f2{q0.x, q0.y}, f2{q0.z, q0.w}, f2{q1.x, q1.y}, f2{q1.z, q1.w}};
                        float d8[8];
                        zl_bwd_saved<L, PL::v[2 * J], PL::v[2 * J + 1]>(psi, q, cot, t, ln, d8);       // psi: the state behind the block
                        reduce8_rows<true>(d8, sums_b, lane, dst_lane);
                        sums_b -= 32;
#pragma unroll
                        for (int e = 0; e < E; e++) psi[e] = q[e];
                        gate_fence<L>();
                        b--;
                    }, typename make_iseq<NSL>::type{});
                }
            } else {
                const int tail = K % LL;                   // blocks beyond the last whole layer come first
                static_for([&](auto jj) __attribute__((always_inline)) {
                    constexpr int J = LL - 1 - decltype(jj)::value;
                    if (J < tail) bwd_step(jj);
                }, typename make_iseq<LL>::type{});
            }
            for (int full = K / LL - SAVE - SAVEL; full > 0; full--) static_for(bwd_step, typename make_iseq<LL>::type{});
        }
        CPF_MARK(gsurf);
        {   // generator sums of every qubit at the surface state
            float ds[8 * NBATCH];
#pragma unroll
            for (int k = 0; k < 8 * NBATCH; k++) ds[k] = 0.f;
            f2 zt = {0.f, 0.f};                         // lane qubits: Pz = -sg * sum over the lane's amplitudes of Im(conj(c) p)
            if constexpr (L::NLR > 0) {
#pragma unroll
                for (int e = 0; e < E; e++) zt = pk_fma(cot[e], swp(psi[e]), zt);
            }
            static_for([&](auto qq) __attribute__((always_inline)) {
                constexpr int Q = decltype(qq)::value;
                f2 pp[E];
                zl_xy_sums<L, Q>(psi, cot, ln, ds[3 * Q], ds[3 * Q + 1], pp);
                if constexpr (L::row_on_lane(Q)) ds[3 * Q + 2] = -ln.sg[Q] * (zt.x - zt.y);
                else {
                    constexpr int BIT = L::row_bit(Q);
                    f2 az = {0.f, 0.f};
#pragma unroll
                    for (int e0 = 0; e0 < E; e0++) {
                        if (e0 & (1 << BIT)) continue;
                        az = pk_fma(cot[e0], swp(psi[e0]), az);
                        az = pk_fma(-cot[e0 | (1 << BIT)], swp(psi[e0 | (1 << BIT)]), az);
                    }
                    ds[3 * Q + 2] = az.x - az.y;
                }
            }, typename make_iseq<NQ>::type{});
#pragma unroll
            for (int bt = 0; bt < NBATCH; bt++) {
                float d8[8];
#pragma unroll
                for (int k = 0; k < 8; k++) d8[k] = ds[8 * bt + k];
                reduce8_rows(d8, gsurf_w + 32 * bt, lane, dst_lane);
            }
        }
        lds_sync<WPS>();

        CPF_MARK(chain);
        // ================= chain: the gradients of the item's own angles, into the owner's registers
#pragma unroll
        for (int s = 0; s < NSLOT; s++) {
            Item& t = itm[s];
            const int i = tid + THREADS * s;
            if (t.kind == 0) {
                const int b = i >> 1, side = i & 1, kx = 3 * side;
                const float px = slab_sum(lds.sums, K * 32, b, kx), py = slab_sum(lds.sums, K * 32, b, kx + 1), pz = slab_sum(lds.sums, K * 32, b, kx + 2);
                auto rrec = [&](int idx, float& c, float& sn, int& kind) __attribute__((always_inline)) {
                    const int k = (idx - 1) >> 1;
                    c = t.cs[k][0]; sn = t.cs[k][1]; kind = 0;
                };
                float gpart[4];
                zyz::chain_side<ROT>(gpart, px, py, pz, t.sd, rrec, NR, side);
#pragma unroll
                for (int k = 0; k < NR; k++) t.g[k] = gpart[k];
                if (ENT == 0 && side == 0) {
                    float pzq[2];
#pragma unroll
                    for (int sd2 = 0; sd2 < 2; sd2++) {
                        const int* st = reinterpret_cast<const int*>(smem + lds.stat + 12 * (i + sd2));
                        const int wp = st[0], q = st[1];
                        pzq[sd2] = wp < 0 ? slab_sum(lds.gsurf, NBATCH * 32, (3 * q + 2) >> 3, (3 * q + 2) & 7) : slab_sum(lds.sums, K * 32, wp >> 1, 3 * (wp & 1) + 2);
                    }
                    t.g[3] = zyz::chain_cp(slab_sum(lds.sums, K * 32, b, 6), pzq[0], pzq[1], t.cacb, (NSAVE > 0 && b >= K - NSAVE - NSL) ? 1.f : t.d0);
                }
            } else if (t.kind == 1) {
                const int q = i - 2 * K;
                float pv[3], gs[3];
#pragma unroll
                for (int k = 0; k < 3; k++) pv[k] = slab_sum(lds.gsurf, NBATCH * 32, (3 * q + k) >> 3, (3 * q + k) & 7);
                zyz::chain_surface(gs, pv[0], pv[1], pv[2], t.m, t.cs[1][0], t.cs[1][1], t.cs[2][0], t.cs[2][1]);
#pragma unroll
                for (int k = 0; k < 3; k++) t.g[k] = gs[k];
            }
        }
        CPF_MARK(end);
        // one wave: no barrier - the next step's owner phase reads only this lane's registers.  Several waves: another wave may still be
        // reading the side records / static structure this wave's owner phase is about to overwrite
        if constexpr (WPS > 1) lds_sync<WPS>();
    }

    // ---- epilogue: reg recompute, loss = regloss - reg (optimization.py:365-370); argmin, CZ count, entry-loss flag.  Every CP
    // angle is read back by the lane that wrote it.
    const int bi = (best < first) ? 1 : 0;
    float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NSLOT; s++) {
#pragma unroll
        for (int k = 0; k < 4; k++) if (itm[s].pj[k] >= 0) out_best[itm[s].pj[k]] = itm[s].bt[k];
        const int j = itm[s].pj[3];
        if (j >= 0) {
            const float t0 = a.out_params[(s_idx * 2 + 0) * P + j], t1 = itm[s].bt[3];
            float sl;
            if (a.use_reg) { acc[0] += penalty_lds(smem, a.pen_kind, t0, sl); acc[1] += penalty_lds(smem, a.pen_kind, t1, sl); }
            acc[2] += cz_value(bi ? t1 : t0, a.thr_cp);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) acc[k] = wave_sum(acc[k]);
    if constexpr (WPS > 1) {
        lds_sync<WPS>();
        if (lane == 0) { red[wave * 3] = acc[0]; red[wave * 3 + 1] = acc[1]; red[wave * 3 + 2] = acc[2]; }
        lds_sync<WPS>();
#pragma unroll
        for (int k = 0; k < 3; k++) { float sum = 0.f; for (int w = 0; w < WPS; w++) sum += red[w * 3 + k]; acc[k] = sum; }
    }
    if (tid == 0) {
        const float reg0 = a.use_reg ? a.r * acc[0] : 0.f, reg1 = a.use_reg ? a.r * acc[1] : 0.f;
        const float l0 = first - reg0, l1 = best - reg1;
        a.out_regloss[s_idx * 2 + 0] = first; a.out_regloss[s_idx * 2 + 1] = best;
        a.out_reg[s_idx * 2 + 0] = reg0;      a.out_reg[s_idx * 2 + 1] = reg1;
        a.out_loss[s_idx * 2 + 0] = l0;       a.out_loss[s_idx * 2 + 1] = l1;
        a.out_cz[s_idx] = (int32_t)acc[2];
        a.out_pass[s_idx] = ((bi ? l1 : l0) <= a.entry_loss) ? 1 : 0;
    }
}

// (the instance that keeps states runs two waves per SIMD like the others: its registers, kept states included, must fit 256 per lane -
// without the second bound the compiler parks the kept states in accumulation registers, 394 in all, and halves the residency)
template <int NQ, int LW, int ROT, int NSLOT_SAVE, int... UD>
__global__ __launch_bounds__(256, ((NSLOT_SAVE >> 4) & 15) != 0 ? 2 : 1) void k_cp_zyzl(const ZyzlArgs za) {
    zyzl_body<NQ, LW, ROT, NSLOT_SAVE, UD...>(za);
}

}  // namespace cpf
