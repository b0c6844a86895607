// cpflow_amd: the ZYZ form of the entangling blocks (cpf_zyz_math.h, tools/zyz_model.py) on the gate-by-gate layout - one wave
// per sample, 3..5 qubits.
//
// Layout Lay<NQ, 0> (cpf_device.h): the first NLR row qubits sit on DPP-reachable lane bits, the others on register bits
// (E = 2^NRR amplitudes per lane); the plan relabels the problem's qubits so that the most-used ones are register qubits.
// A block is  x <- d_ab x  (one merged phase per amplitude of the pair, read from a per-block LDS table; a lane qubit's bit
// selects the lane's entries) followed by the two unnormalised real stages (a, b) <- (a - t b, b + t a):
//     register qubit   2 packed instructions per amplitude pair
//     lane qubit       x += (+-t) x[partner]: two v_fmac_f32_dpp per amplitude (the sign is the lane's bit)
// where the gate-by-gate kernel (cpf_layered.h) applies a CP, two x, two y and two z rotations per block.  Backward: the same
// in reverse on psi and on the cotangent, three generator sums per qubit right after the stages (a lane qubit's partner
// amplitudes are fetched once and serve its sums and psi's un-application), one sum over the |11> amplitudes before conj(d).
// Build and chain phases are those of cpf_zyz.h (one thread per side / block / qubit around the sweeps), except that the
// per-block table stays in LDS.  Up to four samples per workgroup, one per wave (KArgs::group, cpf_layered.h).
#pragma once
#include "cpf_zyz.h"

namespace cpf {

constexpr int ZYZL_TAB_FLOATS = 12;                  // per block: d_k (re, im), k = 2 a_up + a_dn; t_dn, t_up; two spare

struct ZyzlArgs { KArgs k; const ZyzSideInfo* sides; int K, n_rot; };

struct ZyzlLds {                                     // float offsets, per sample
    int rec, side, binfo, btab, surfm, lfin, msf, fw, th, mm, vv, psl, gw, sums, gsurf, red, total;
    __host__ __device__ static constexpr int up4(int x) { return (x + 3) & ~3; }
    __host__ __device__ ZyzlLds(int P, int G, int K, int nq) {
        rec = PEN_FLOATS;
        side = rec + 4 * G;
        binfo = side + 16 * K;
        btab = up4(binfo + 2 * K);
        surfm = btab + ZYZL_TAB_FLOATS * K;
        lfin = surfm + 8 * nq;
        msf = lfin + 4 * nq;
        fw = up4(msf + nq);
        th = fw + 8;
        mm = th + P; vv = mm + P; psl = vv + P;
        gw = psl + P;
        sums = up4(gw + G);                    // [K][32]: generator sums, reduce8_rows slabs
        gsurf = sums + K * 32;
        red = gsurf + ((3 * nq + 7) / 8) * 32;
        total = up4(red + 8);
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
    f2 ph[NPH];
    float tu, td;
    __host__ __device__ static constexpr int idx(int e) {           // phase / z-accumulator index of amplitude e
        return (UL ? 0 : ((e >> L::row_bit(UP)) & 1)) * ND + (DL ? 0 : ((e >> L::row_bit(DN)) & 1));
    }
    __host__ __device__ static constexpr bool both_set(int e) {      // register part of the |11> predicate
        return (UL || ((e >> L::row_bit(UP)) & 1)) && (DL || ((e >> L::row_bit(DN)) & 1));
    }
};
template <class L, int UP, int DN>
__device__ __forceinline__ void zl_load(ZlTab<L, UP, DN>& t, const float* bt, const ZlLane<L>& ln) {
    using T = ZlTab<L, UP, DN>;
    const float2 tt = *reinterpret_cast<const float2*>(bt + 8);        // (t_dn, t_up)
    if constexpr (T::UL) t.tu = ln.sg[UP] * tt.y; else t.tu = tt.y;
    if constexpr (T::DL) t.td = ln.sg[DN] * tt.x; else t.td = tt.x;
#pragma unroll
    for (int iu = 0; iu < T::NU; iu++)
#pragma unroll
        for (int id = 0; id < T::ND; id++) {
            int au = iu, ad = id;
            if constexpr (T::UL) au = ln.lb[UP] ? 1 : 0;
            if constexpr (T::DL) ad = ln.lb[DN] ? 1 : 0;
            const float2 v = *reinterpret_cast<const float2*>(bt + 2 * (2 * au + ad));
            t.ph[iu * T::ND + id] = f2{v.x, v.y};
        }
}
// one unnormalised stage on qubit Q: (a, b) <- (a - t b, b + t a); t arrives signed for a lane qubit
template <class L, int Q>
__device__ __forceinline__ void zl_stage(f2 (&x)[L::E], float t) {
    if constexpr (L::row_on_lane(Q)) {
        constexpr int BIT = L::row_bit(Q);
#pragma unroll
        for (int e = 0; e < L::E; e++) {
            const float px = xor_lane<BIT>(x[e].x), py = xor_lane<BIT>(x[e].y);
            x[e].x = fmaf(t, px, x[e].x);
            x[e].y = fmaf(t, py, x[e].y);
        }
    } else {
        constexpr int BIT = L::row_bit(Q);
#pragma unroll
        for (int e0 = 0; e0 < L::E; e0++) {
            if (e0 & (1 << BIT)) continue;
            const int e1 = e0 | (1 << BIT);
            const f2 a = x[e0], b = x[e1];
            x[e0] = pk_fma(splat(-t), b, a);
            x[e1] = pk_fma(splat(t), a, b);
        }
    }
}
template <class L, int UP, int DN>
__device__ __forceinline__ void zl_fwd(f2 (&x)[L::E], const ZlTab<L, UP, DN>& t) {
    using T = ZlTab<L, UP, DN>;
#pragma unroll
    for (int e = 0; e < L::E; e++) x[e] = zl_cmul(t.ph[T::idx(e)], x[e]);
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
        zl_stage<L, UP>(p, -t.tu);
    } else if constexpr (T::UL) {
#pragma unroll
        for (int e = 0; e < E; e++) p[e] = pk_fma(splat(-t.tu), ppu[e], p[e]);
        zl_stage<L, DN>(p, -t.td);
    } else {
        zl_stage<L, DN>(p, -t.td);
        zl_stage<L, UP>(p, -t.tu);
    }
    zl_stage<L, DN>(c, -t.td);
    zl_stage<L, UP>(c, -t.tu);
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
    for (int e = 0; e < E; e++) { p[e] = zl_cmul_conj(t.ph[T::idx(e)], p[e]); c[e] = zl_cmul_conj(t.ph[T::idx(e)], c[e]); }
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

template <int NQ, int ROT, int... UD>
__device__ __forceinline__ void zyzl_body(const ZyzlArgs& za) {
    using L = Lay<NQ, 0>;
    using PL = PairList<UD...>;
    constexpr int LL = sizeof...(UD) / 2;
    static_assert(LL > 0, "the ZYZ-form kernel is layer-specialised");
    constexpr int E = L::E, THREADS = 64, NS = 3 * NQ, NBATCH = (NS + 7) / 8;
    const KArgs& a = za.k;
    extern __shared__ __attribute__((aligned(16))) float smem_wg[];
    const int P = a.P, G = a.G, K = za.K, NR = za.n_rot;
    const ZyzlLds lds(P, G, K, NQ);
    // one wave per sample; a workgroup may hold several samples, one per wave, each with its own copy of the LDS map
    const int sub = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int tid = (int)threadIdx.x & 63, lane = tid;
    const int64_t s_idx = (int64_t)blockIdx.x * (a.group > 1 ? a.group : 1) + sub;
    if (s_idx >= a.nsamples) return;
    float* smem = smem_wg + sub * lds.total;
    float4* rec = reinterpret_cast<float4*>(smem + lds.rec);
    float* th = smem + lds.th;
    float* mm = smem + lds.mm;
    float* vv = smem + lds.vv;
    float* psl = smem + lds.psl;
    float* gw = smem + lds.gw;
    float4* sidev = reinterpret_cast<float4*>(smem + lds.side);
    unsigned long long* fw = reinterpret_cast<unsigned long long*>(smem + lds.fw);
    const float* init = a.init + s_idx * P;
    const uint8_t* frozen = a.frozen ? a.frozen + s_idx * P : nullptr;
    const int col = L::col_index(lane, 0);
    const int dst_lane = reduce8_dst_lane(lane);
    const int per_block = 1 + 2 * NR;
    const int nw = (2 * K + 63) >> 6;
    ZlLane<L> ln;
#pragma unroll
    for (int i = 0; i < L::NLR; i++) { ln.lb[i] = (lane >> L::row_lane_bit(i)) & 1; ln.sg[i] = ln.lb[i] ? 1.f : -1.f; }
    uint32_t toff_i = (uint32_t)(L::row_index(lane, 0) * L::D + col) * 8u;
    int lane_i = lane, col_i = col;
    const char* tbase = reinterpret_cast<const char*>(a.target);
    auto target_at = [&](int e) __attribute__((always_inline)) {
        return *reinterpret_cast<const float2*>(tbase + (toff_i + (uint32_t)((L::row_index(0, e)) * L::D * 8)));
    };

    penalty_table_init(smem, a, tid);
    for (int g = tid; g < G; g += THREADS) {
        const int4 op = a.prog[g];
        rec[g] = make_float4(op.x == OP_CZ ? -1.f : 1.f, 0.f, __int_as_float(op.x | (op.y << 4) | (op.z << 8)), 0.f);
    }
    lds_sync<1>();
    auto publish = [&](int j, float t) {
        float sn, cn;
        const bool cp = a.pkind[j];
        fast_sincos(cp ? t : 0.5f * t, sn, cn);
        *reinterpret_cast<float2*>(&rec[a.pgate[j]]) = make_float2(cn, sn);
    };
    for (int j = tid; j < P; j += THREADS) {
        const float t0 = init[j];
        th[j] = t0; mm[j] = 0.f; vv[j] = 0.f;
        a.out_params[(s_idx * 2 + 0) * P + j] = t0;
        a.out_params[(s_idx * 2 + 1) * P + j] = t0;
        publish(j, t0);
    }
    lds_sync<1>();

    auto block_rec = [&](int g0) {
        return [rec, g0](int i, float& c, float& s, int& kind) {
            const float4 r = rec[g0 + i];
            c = r.x; s = r.y; kind = __builtin_amdgcn_readfirstlane(__float_as_int(r.z) & 15);
        };
    };
    auto load_side = [&](int w) {
        const float4 s0 = sidev[2 * w], s1 = sidev[2 * w + 1];
        zyz::Side sd;
        sd.t = s0.x; sd.sc = s0.y; sd.l0 = zyz::c2(s0.z, s0.w); sd.l1 = zyz::c2(s1.x, s1.y); sd.f = __float_as_int(s1.z); sd.m = __float_as_int(s1.w);
        return sd;
    };
    auto parity = [&](const unsigned long long* mask) {
        int n = 0;
        for (int k = 0; k < nw; k++) n += __popcll(fw[k] & mask[k]);
        return n & 1;
    };
    auto slab_sum = [&](int base, int idx, int v) {
        const float4 q = *reinterpret_cast<const float4*>(smem + base + 32 * idx + 4 * v);
        return (q.x + q.y) + (q.z + q.w);
    };

    f2 psi[E], cot[E];
    float best = 0.f, first = 0.f;
    const int iters = a.T > 0 ? a.T : 1;
    for (int it = 0; it < iters; it++) {
        const float2 bc = const_load_f2(a.bias, a.T > 0 ? it : 0);
        asm volatile("" : "+v"(toff_i), "+v"(lane_i), "+v"(col_i));
        // ================= build A: one thread per side (cpf_zyz.h)
        for (int base = 0; base < 2 * K; base += THREADS) {
            const int w = base + tid;
            bool flip = false;
            if (w < 2 * K) {
                const zyz::Side sd = zyz::side_build<ROT>(block_rec(NS + per_block * (w >> 1)), NR, w & 1);
                sidev[2 * w] = make_float4(sd.t, sd.sc, sd.l0.re, sd.l0.im);
                sidev[2 * w + 1] = make_float4(sd.l1.re, sd.l1.im, __int_as_float(sd.f), 0.f);
                flip = sd.f != 0;
            }
            const unsigned long long bits = __builtin_amdgcn_ballot_w64(flip);
            if (lane == 0) fw[base >> 6] = bits;
        }
        lds_sync<1>();
        // ================= build B: one thread per block (its table), one per qubit (surface matrix, final left factor)
        for (int w = tid; w < K + NQ; w += THREADS) {
            if (w < K) {
                const int b = w, g0 = NS + per_block * b;
                const ZyzSideInfo* iu = za.sides + 2 * b;
                zyz::Side su = load_side(2 * b), sd = load_side(2 * b + 1), pu = su, pd = sd;
                su.m = parity(iu[0].later); sd.m = parity(iu[1].later);
                const int wu = iu[0].prev, wd = iu[1].prev;
                if (wu >= 0) { pu = load_side(wu); pu.m = su.m ^ su.f; }
                if (wd >= 0) { pd = load_side(wd); pd.m = sd.m ^ sd.f; }
                const float4 r0 = rec[g0];
                float out[10];
                const int cacb = zyz::block_table(out, su, sd, wu >= 0, pu, wd >= 0, pd, r0.x, r0.y, false);      // index 2 a_up + a_dn
                float* bt = smem + lds.btab + ZYZL_TAB_FLOATS * b;
                *reinterpret_cast<float4*>(bt) = make_float4(out[0], out[1], out[2], out[3]);
                *reinterpret_cast<float4*>(bt + 4) = make_float4(out[4], out[5], out[6], out[7]);
                *reinterpret_cast<float2*>(bt + 8) = make_float2(out[8], out[9]);
                smem[lds.side + 16 * b + 7] = __int_as_float(su.m);
                smem[lds.side + 16 * b + 15] = __int_as_float(sd.m);
                *reinterpret_cast<float2*>(smem + lds.binfo + 2 * b) = make_float2(__int_as_float(cacb), out[0]);
            } else {
                const int q = w - K;
                const ZyzSideInfo* iq = za.sides + 2 * K + q;
                const int m = parity(iq->later), last = iq->prev;
                const float4 r0 = rec[3 * q], r1 = rec[3 * q + 1], r2 = rec[3 * q + 2];
                float sm[8];
                zyz::surface_matrix(sm, r0.x, r0.y, r1.x, r1.y, r2.x, r2.y, m);
                *reinterpret_cast<float4*>(smem + lds.surfm + 8 * q) = make_float4(sm[0], sm[1], sm[2], sm[3]);
                *reinterpret_cast<float4*>(smem + lds.surfm + 8 * q + 4) = make_float4(sm[4], sm[5], sm[6], sm[7]);
                float4 lf = make_float4(1.f, 0.f, 1.f, 0.f);
                if (last >= 0) { const zyz::Side sl = load_side(last); lf = make_float4(sl.l0.re, sl.l0.im, sl.l1.re, sl.l1.im); }
                *reinterpret_cast<float4*>(smem + lds.lfin + 4 * q) = lf;
                smem[lds.msf + q] = __int_as_float(m);
            }
        }
        lds_sync<1>();

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
        {
            int b = 0;
            auto fwd_step = [&](auto jj) __attribute__((always_inline)) {
                constexpr int J = decltype(jj)::value;
                ZlTab<L, PL::v[2 * J], PL::v[2 * J + 1]> t;
                zl_load(t, smem + lds.btab + ZYZL_TAB_FLOATS * b, ln);
                zl_fwd<L, PL::v[2 * J], PL::v[2 * J + 1]>(psi, t);
                gate_fence<L>();
                b++;
            };
            for (int full = K / LL; full > 0; full--) static_for(fwd_step, typename make_iseq<LL>::type{});
            static_for([&](auto jj) __attribute__((always_inline)) { if (b < K) fwd_step(jj); }, typename make_iseq<LL>::type{});
        }

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
            for (int j = tid; j < P; j += THREADS) {
                if (a.pkind[j]) {
                    float sl;
                    acc[2] += penalty_lds(smem, a.pen_kind, th[j], sl);
                    psl[j] = a.r * sl;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 3; k++) acc[k] = wave_sum(acc[k]);
        const float tr = acc[0], ti = acc[1];
        const float ab = sqrtf(fmaf(tr, tr, ti * ti));
        const float loss = (a.loss_kind == 2) ? 1.f - tr * a.inv_dim2 : 1.f - ab * ab * a.inv_dim2;
        const float reg = a.use_reg ? a.r * acc[2] : 0.f;
        const float regloss = loss + reg;
        bool improved = false;
        if (it == 0) { first = regloss; best = regloss; }
        else if (regloss < best) { best = regloss; improved = true; }
        if (improved || a.hist_params) {
            for (int j = tid; j < P; j += THREADS) {
                const float t = th[j];
                if (improved) a.out_params[(s_idx * 2 + 1) * P + j] = t;
                if (a.hist_params) a.hist_params[(s_idx * a.T + it) * P + j] = t;
            }
        }
        if (a.hist_loss && tid == 0) a.hist_loss[s_idx * a.T + it] = regloss;
        if (a.T == 0) break;

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
        {
            int b = K - 1;
            auto bwd_step = [&](auto jj) __attribute__((always_inline)) {
                constexpr int J = LL - 1 - decltype(jj)::value;
                ZlTab<L, PL::v[2 * J], PL::v[2 * J + 1]> t;
                zl_load(t, smem + lds.btab + ZYZL_TAB_FLOATS * b, ln);
                float d8[8];
                zl_bwd<L, PL::v[2 * J], PL::v[2 * J + 1]>(psi, cot, t, ln, d8);
                reduce8_rows<true>(d8, smem + lds.sums + 32 * b, lane, dst_lane);
                gate_fence<L>();
                b--;
            };
            const int tail = K % LL;                   // blocks beyond the last whole layer come first
            static_for([&](auto jj) __attribute__((always_inline)) {
                constexpr int J = LL - 1 - decltype(jj)::value;
                if (J < tail) bwd_step(jj);
            }, typename make_iseq<LL>::type{});
            for (int full = K / LL; full > 0; full--) static_for(bwd_step, typename make_iseq<LL>::type{});
        }
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
                reduce8_rows(d8, smem + lds.gsurf + 32 * bt, lane, dst_lane);
            }
        }
        lds_sync<1>();

        // ================= chain: one thread per side (its rotation angles; side 0 also the CP angle), one per qubit
        for (int w = tid; w < 2 * K + NQ; w += THREADS) {
            if (w < 2 * K) {
                const int b = w >> 1, side = w & 1, g0 = NS + per_block * b, kx = 3 * side;
                const float px = slab_sum(lds.sums, b, kx), py = slab_sum(lds.sums, b, kx + 1), pz = slab_sum(lds.sums, b, kx + 2);
                const zyz::Side sd = load_side(w);
                float gpart[8];
                zyz::chain_side<ROT>(gpart, px, py, pz, sd, block_rec(g0), NR, side);
#pragma unroll
                for (int i = 0; i < (ROT ? (ROT & 15) : 8); i++) {
                    if (i < zyz::rotc_n<ROT>(NR)) gw[g0 + 1 + 2 * i + side] = gpart[i];
                }
                const bool is_cp = ROT ? ((ROT >> 16) & 3) == 0 : (__float_as_int(rec[g0].z) & 15) == OP_CP;
                if (side == 0 && is_cp) {
                    float pzq[2];
#pragma unroll
                    for (int s = 0; s < 2; s++) {
                        const int wp = za.sides[w + s].prev, q = za.sides[w + s].qubit;
                        pzq[s] = wp < 0 ? slab_sum(lds.gsurf, (3 * q + 2) >> 3, (3 * q + 2) & 7) : slab_sum(lds.sums, wp >> 1, 3 * (wp & 1) + 2);
                    }
                    const float2 bi = *reinterpret_cast<const float2*>(smem + lds.binfo + 2 * b);
                    gw[g0] = zyz::chain_cp(slab_sum(lds.sums, b, 6), pzq[0], pzq[1], __float_as_int(bi.x), bi.y);
                }
            } else {
                const int q = w - 2 * K;
                float pv[3], gs[3];
#pragma unroll
                for (int k = 0; k < 3; k++) pv[k] = slab_sum(lds.gsurf, (3 * q + k) >> 3, (3 * q + k) & 7);
                const float4 r1 = rec[3 * q + 1], r2 = rec[3 * q + 2];
                zyz::chain_surface(gs, pv[0], pv[1], pv[2], __float_as_int(smem[lds.msf + q]), r1.x, r1.y, r2.x, r2.y);
#pragma unroll
                for (int k = 0; k < 3; k++) gw[3 * q + k] = gs[k];
            }
        }
        lds_sync<1>();

        // ================= Adam (optax 0.1.1 scale_by_adam + scale(-lr)), owners only
        const float ibc1 = __builtin_amdgcn_rcpf(bc.x), ibc2 = __builtin_amdgcn_rcpf(bc.y);
        for (int j = tid; j < P; j += THREADS) {
            float g = gw[a.pgate[j]];
            if (a.use_reg && a.pkind[j]) g += psl[j];
            if (frozen && frozen[j]) continue;
            const float m = fmaf(a.b1, mm[j], a.omb1 * g);
            const float v = fmaf(a.b2, vv[j], a.omb2 * (g * g));
            mm[j] = m; vv[j] = v;
            const float t = fmaf(-a.lr, (m * ibc1) * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v * ibc2) + a.eps), th[j]);
            th[j] = t;
            publish(j, t);
        }
        lds_sync<1>();
    }

    // ---- epilogue: reg recompute, loss = regloss - reg (optimization.py:365-370); argmin, CZ count, entry-loss flag
    const int bi = (best < first) ? 1 : 0;
    float acc[3] = {0.f, 0.f, 0.f};
    for (int j = tid; j < P; j += THREADS) {
        if (a.pkind[j]) {
            const float t0 = a.out_params[(s_idx * 2 + 0) * P + j];
            const float t1 = a.out_params[(s_idx * 2 + 1) * P + j];
            float sl;
            if (a.use_reg) { acc[0] += penalty_lds(smem, a.pen_kind, t0, sl); acc[1] += penalty_lds(smem, a.pen_kind, t1, sl); }
            acc[2] += cz_value(bi ? t1 : t0, a.thr_cp);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) acc[k] = wave_sum(acc[k]);
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

template <int NQ, int ROT, int... UD>
__global__ __launch_bounds__(256, 1) void k_cp_zyzl(const ZyzlArgs za) {
    zyzl_body<NQ, ROT, UD...>(za);
}

}  // namespace cpf
