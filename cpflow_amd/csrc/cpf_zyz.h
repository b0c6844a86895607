// cpflow_amd: the ZYZ-form kernel for 4..6 qubits - every entangling block of the template (cpflow/main.py:69-82) as ONE merged
// phase per amplitude of its qubit pair followed by two unnormalised real 2x2 stages.
//
// Why.  The dense-block kernel (cpf_dense.h) pays 32 packed instructions per group of four amplitudes and block application
// (a general 4x4) and 32 more for the 4x4 environment: 128 per group, block and step.  But a block's single-qubit part
// A = Rz Ry Rx is an SU(2) element, A = diag(l0, l1) . cos [[1, -t], [t, 1]] . diag(1, r1) (. a flip when |tan| would exceed 1),
// every diagonal factor commutes with its neighbours up to the next real stage, and the flips can be pushed onto the initial
// state (cpf_zyz_math.h, tools/zyz_model.py).  What is left per group of four amplitudes (cpf_zyz_asm.h):
//     forward     x_k <- d_k x_k (d_0 real), two stages (a, b) <- (a - t b, b + t a)                         15 instructions
//     backward    the same in reverse on psi and on the cotangent, 20 for the generator sums + 1              51
// i.e. 66 instead of 128, and 7 sums per block cross the lanes instead of a 32-value environment.
// The real scale d_0 is applied by the last block of each layer only, for the whole layer ("build B" below): the other blocks run
// the 14- / 49-instruction forms of the two routines.
//
// Layout, exchanges of lane qubits against register qubits, the layer as a template argument and the scalar-register table
// (a global scratch written by the build phase, read back with scalar loads) are those of cpf_dense.h: DLay<NQ, LW>,
// dense_layer_entry, dense_same_swap.  Only layer-specialised instances exist (ahead of time for the BASELINE shapes, at run
// time otherwise); CX blocks (not diagonal) and launches without a compiler or cached object stay on the dense kernel.
//
// Phases of one optimiser step (T of them in one launch, one workgroup per sample):
//   owner    Adam on the angles this thread owns, their (cos, sin) to the gate records            [LDS: th, m, v, rec]
//   build A  thread (b, side): the factors of one side of block b; the flip bits of all sides by ballot      [LDS: side, fw]
//   build B  thread b: pending-flip parities, the merged phases and tangents of block b                  [global: tab; LDS: side.m, binfo]
//            thread q: the (flipped) surface matrix of qubit q, the left factor its last block leaves behind   [LDS: surfm, lfin]
//   forward  psi = product state of the surface gates; blocks b = 0 .. K-1
//   loss     the final diagonal meets the target: Vd = conj(dfin) V;  T = sum conj(Vd) psi over the workgroup
//   backward C = f Vd; blocks K-1 .. 0 (sums reduced over the wave -> LDS); generator sums of every qubit at the surface state
//   chain    thread (b, side): the side's rotation gradients (and the CP angle's); thread q: the surface gate's       [LDS: gw]
#pragma once
#include "cpf_dense.h"
#include "cpf_zyz_math.h"
#include "cpf_zyz_asm.h"

namespace cpf {

// Static structure of the circuit for the build and chain phases, one entry per side w = 2 b + side, then one per qubit:
//   prev       side: the previous side on the same qubit (or -1);   qubit entry: the last side on the qubit (or -1)
//   later[k]   side: bit mask (bits 64 k ..) of the LATER sides on the same qubit;   qubit entry: all sides on the qubit
struct ZyzSideInfo { int prev, qubit, pad0, pad1; unsigned long long later[4]; };
constexpr int ZYZ_MAX_BLOCKS = 128;                  // 2 K side bits in four 64-bit words
constexpr int ZYZ_TAB_FLOATS = 16;                   // floats per block in the global scratch (10 used; one 64-byte line)

struct ZyzArgs { KArgs k; const DenseBlock* blocks; const ZyzSideInfo* sides; int K, n_rot; float* tab; };

// ---- LDS map (float offsets) -------------------------------------------------------------------------------------------
struct ZyzLds {
    int rec, side, binfo, surfm, lfin, msf, fw, th, mm, vv, psl, gw, sums, gsurf, red, total;
    __host__ __device__ static constexpr int up4(int x) { return (x + 3) & ~3; }
    __host__ __device__ ZyzLds(int P, int G, int K, int nq, int wps) {
        rec = PEN_FLOATS;                      // G x {cos, sin, code, -}
        side = rec + 4 * G;                    // 2 K x {t, sc, l0, l1, f, m}
        binfo = side + 16 * K;                 // K x {cacb, d0, 1 / phi^2, -}
        surfm = binfo + 4 * K;                 // nq x 2x2 complex
        lfin = surfm + 8 * nq;                 // nq x (l0, l1)
        msf = lfin + 4 * nq;                   // nq flip parities
        fw = up4(msf + nq);                    // 4 x 64 flip bits
        th = fw + 8;                           // owner state and r * penalty slope, P each
        mm = th + P; vv = mm + P; psl = vv + P;
        gw = psl + P;                          // G: gradient of each gate's angle
        sums = up4(gw + G);                    // [wps][K][32]: generator sums, reduce8_rows slabs
        gsurf = sums + wps * K * 32;           // [wps][NBATCH][32]: surface generator sums
        red = gsurf + wps * ((3 * nq + 7) / 8) * 32;
        total = up4(red + 8 * wps + 4);
    }
};

// the scalar-register table of block b: ten floats of its 64-byte line
typedef float f8v __attribute__((ext_vector_type(8)));
typedef const volatile f8v __attribute__((address_space(4)))* zs8_ptr;            // volatile: rebuilt every step, never hoist the read
typedef const volatile f2 __attribute__((address_space(4)))* zs2_ptr;
__device__ __forceinline__ void zyz_load_tab(const float* gp, ZyzTab& g) {
    zs8_ptr p8 = reinterpret_cast<zs8_ptr>((const __attribute__((address_space(4))) void*)gp);
    zs2_ptr p2 = reinterpret_cast<zs2_ptr>((const __attribute__((address_space(4))) void*)(gp + 8));
    const f8v v = p8[0];
    const f2 t = p2[0];
    g.sd = f2{v[0], v[1]}; g.d1 = f2{v[2], v[3]}; g.d2 = f2{v[4], v[5]}; g.d3 = f2{v[6], v[7]}; g.tt = t;
}
// scalar loads return out of order, every wait on them is lgkmcnt(0): this empty statement makes the compiler wait for the
// current table here, the caller issues the next block's loads right after it (cpf_dense.h dense_touch_s)
__device__ __forceinline__ void zyz_touch(const ZyzTab& g) { asm volatile("" :: "s"(g.sd), "s"(g.d1), "s"(g.d2), "s"(g.d3), "s"(g.tt)); }

#ifndef CPF_ZYZ_LAYER_SCALE
#define CPF_ZYZ_LAYER_SCALE 1    // A/B: 0 = every block multiplies by its own real scale d_0 (three packed multiplies per group, block and step more)
#endif
#ifndef CPF_ZYZ_FWD2
#define CPF_ZYZ_FWD2 1           // A/B: 0 = one group of four amplitudes per asm statement in the forward sweep
#endif
// SC: the block multiplies by its real scale d_0 (the last block of a layer, which carries the layer's; the others do not: a
// layer's blocks leave psi and the cotangent off by known real factors in between, see "build B")
template <int E, int RA, int RB, bool SC>
__device__ __forceinline__ void zyz_apply(f2 (&x)[E], const ZyzTab& g) {
    // the groups of the block in pairs: two groups per statement, their instructions alternating
    int prev = -1;
#pragma unroll
    for (int e = 0; e < E; e++) {
        if (e & ((1 << RA) | (1 << RB))) continue;
        if (CPF_ZYZ_FWD2 != 0 && E >= 8) {
            if (prev < 0) { prev = e; continue; }
            if constexpr (SC) zyz_group_fwd2(x[prev], x[prev | (1 << RA)], x[prev | (1 << RB)], x[prev | (1 << RA) | (1 << RB)],
                                             x[e], x[e | (1 << RA)], x[e | (1 << RB)], x[e | (1 << RA) | (1 << RB)], g);
            else zyz_group_fwd2_ns(x[prev], x[prev | (1 << RA)], x[prev | (1 << RB)], x[prev | (1 << RA) | (1 << RB)],
                                   x[e], x[e | (1 << RA)], x[e | (1 << RB)], x[e | (1 << RA) | (1 << RB)], g);
            prev = -1;
        } else {
            if constexpr (SC) zyz_group_fwd(x[e], x[e | (1 << RA)], x[e | (1 << RB)], x[e | (1 << RA) | (1 << RB)], g);
            else zyz_group_fwd_ns(x[e], x[e | (1 << RA)], x[e | (1 << RB)], x[e | (1 << RA) | (1 << RB)], g);
        }
    }
}
// backward pass of one block; d8 receives this thread's shares {Px, Py, Pz of the lo qubit, the same of the hi qubit, S11, -}
template <int E, int RA, int RB, bool SC>
__device__ __forceinline__ void zyz_backward(f2 (&p)[E], f2 (&c)[E], const ZyzTab& g, float (&d8)[8]) {
    ZyzAcc a;
    bool first = true;
#pragma unroll
    for (int e = 0; e < E; e++) {
        if (e & ((1 << RA) | (1 << RB))) continue;
        const int e1 = e | (1 << RA), e2 = e | (1 << RB), e3 = e1 | e2;
        if constexpr (SC) {
            if (first) zyz_group_bwd_first(p[e], p[e1], p[e2], p[e3], c[e], c[e1], c[e2], c[e3], a, g);
            else zyz_group_bwd(p[e], p[e1], p[e2], p[e3], c[e], c[e1], c[e2], c[e3], a, g);
        } else {
            if (first) zyz_group_bwd_first_ns(p[e], p[e1], p[e2], p[e3], c[e], c[e1], c[e2], c[e3], a, g);
            else zyz_group_bwd_ns(p[e], p[e1], p[e2], p[e3], c[e], c[e1], c[e2], c[e3], a, g);
        }
        first = false;
    }
    const f2 zl = a.ta + a.tb, zh = a.ta - a.tb;
    d8[0] = a.axlo.x - a.axlo.y; d8[1] = a.aylo.x + a.aylo.y; d8[2] = zl.x - zl.y;
    d8[3] = a.axhi.x - a.axhi.y; d8[4] = a.ayhi.x + a.ayhi.y; d8[5] = zh.x - zh.y;
    d8[6] = a.cc.x - a.cc.y; d8[7] = 0.f;
}
// this thread's share of the three generator sums of the qubit on register bit BIT: sum Im(conj(c) sigma_k psi)
template <int E, int BIT>
__device__ __forceinline__ void zyz_gen_sums(const f2 (&p)[E], const f2 (&c)[E], float& px, float& py, float& pz) {
    f2 ax = {0.f, 0.f}, ay = {0.f, 0.f}, az = {0.f, 0.f};
#pragma unroll
    for (int e0 = 0; e0 < E; e0++) {
        if (e0 & (1 << BIT)) continue;
        const int e1 = e0 | (1 << BIT);
        ax = pk_fma(c[e0], swp(p[e1]), ax); ay = pk_fma(-c[e0], p[e1], ay); az = pk_fma(c[e0], swp(p[e0]), az);
        ax = pk_fma(c[e1], swp(p[e0]), ax); ay = pk_fma(c[e1], p[e0], ay);  az = pk_fma(-c[e1], swp(p[e1]), az);
    }
    px = ax.x - ax.y; py = ay.x + ay.y; pz = az.x - az.y;
}

// x[e] = base * prod over the register qubits of f_q[bit of e]: doubling over the register bits.  fac(q, which) returns the
// factor of qubit q for bit value `which`.
template <class L, class F>
__device__ __forceinline__ void zyz_product(f2 (&x)[L::E], f2 base, const F& fac) {
    x[0] = base;
    static_for([&](auto bb) __attribute__((always_inline)) {
        constexpr int RB = decltype(bb)::value;                  // register bit RB holds qubit NRR - 1 - RB
        const f2 f0 = fac(L::NRR - 1 - RB, 0), f1 = fac(L::NRR - 1 - RB, 1);
#pragma unroll
        for (int e = 0; e < (1 << RB); e++) { x[e | (1 << RB)] = cmul(x[e], f1); x[e] = cmul(x[e], f0); }
    }, typename make_iseq<L::NRR>::type{});
}

// UD... = the layer as (up, dn) pairs (block b sits on pair b mod length); ROT: the rotation string and entangler as a
// compile-time code (cpf_blocks.h rot_pack / rot_with_ent), or 0 = kinds from the records (any string up to 8 letters).
template <int NQ, int LW, int ROT, int... UD>
__device__ __forceinline__ void zyz_body(const ZyzArgs& za) {
    using L = DLay<NQ, LW>;
    using PL = PairList<UD...>;
    constexpr int LL = sizeof...(UD) / 2;
    static_assert(LL > 0, "the ZYZ-form kernel is layer-specialised");
    constexpr int E = L::E, WPS = L::WPS, THREADS = L::THREADS, NS = 3 * NQ, NBATCH = (NS + 7) / 8;
    const KArgs& a = za.k;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int P = a.P, G = a.G, K = za.K, NR = za.n_rot;
    const ZyzLds lds(P, G, K, NQ, WPS);
    float4* rec = reinterpret_cast<float4*>(smem + lds.rec);
    float* th = smem + lds.th;
    float* mm = smem + lds.mm;
    float* vv = smem + lds.vv;
    float* psl = smem + lds.psl;
    float* gw = smem + lds.gw;
    float* red = smem + lds.red;
    float4* sidev = reinterpret_cast<float4*>(smem + lds.side);
    unsigned long long* fw = reinterpret_cast<unsigned long long*>(smem + lds.fw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t s_idx = blockIdx.x;
    const float* init = a.init + s_idx * P;
    const uint8_t* frozen = a.frozen ? a.frozen + s_idx * P : nullptr;
    const int col = L::col_index(lane, wave);
    const int surf_dst = reduce8_dst_lane(lane);
    const char* tbase = reinterpret_cast<const char*>(a.target);
    const uint32_t toff = (uint32_t)(L::row_index(lane, 0) * L::D + col) * 8u;
    uint32_t toff_i = toff;
    int lane_i = lane, col_i = col;
    auto target_at = [&](int e) __attribute__((always_inline)) {
        return *reinterpret_cast<const float2*>(tbase + (toff_i + (uint32_t)(L::row_of_reg(e) * L::D * 8)));
    };
    float* sums_w = smem + lds.sums + wave * K * 32;
    float* gsurf_w = smem + lds.gsurf + wave * NBATCH * 32;
    const_i_ptr blk = (const_i_ptr)(const int*)za.blocks;          // scalar loads: the table is wave-uniform
    const int nw = (2 * K + 63) >> 6;                               // 64-bit words of flip bits

    penalty_table_init(smem, a, tid);
    for (int g = tid; g < G; g += THREADS) {
        const int4 op = a.prog[g];
        rec[g] = make_float4(op.x == OP_CZ ? -1.f : 1.f, 0.f, __int_as_float(op.x | (op.y << 4) | (op.z << 8)), 0.f);
    }
    lds_sync<WPS>();
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
    lds_sync<WPS>();

    auto block_rec = [&](int g0) {
        return [rec, g0](int i, float& c, float& s, int& kind) {
            const float4 r = rec[g0 + i];
            c = r.x; s = r.y; kind = __builtin_amdgcn_readfirstlane(__float_as_int(r.z) & 15);      // the same for every block: a scalar
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
    // sum of value v (0 .. 7) of the reduce8_rows slab of (wave, slab index) over the waves and the four row partials
    auto slab_sum = [&](int base, int stride_w, int idx, int v) {
        float s = 0.f;
        for (int wv = 0; wv < WPS; wv++) {
            const float4 q = *reinterpret_cast<const float4*>(smem + base + wv * stride_w + 32 * idx + 4 * v);
            s += (q.x + q.y) + (q.z + q.w);
        }
        return s;
    };

    f2 psi[E], cot[E];
    float best = 0.f, first = 0.f;
    const int iters = a.T > 0 ? a.T : 1;
    for (int it = 0; it < iters; it++) {
        const float2 bc = const_load_f2(a.bias, a.T > 0 ? it : 0);
        // per-lane constants pass through an opaque copy once per step: otherwise everything derived from them is hoisted out of
        // the step loop and occupies registers for the whole run (cpf_dense.h)
        asm volatile("" : "+v"(toff_i), "+v"(lane_i), "+v"(col_i));
        // ================= build A: one thread per side
        for (int base = 0; base < 2 * K; base += THREADS) {
            const int w = base + tid;
            bool flip = false;
            if (w < 2 * K) {
                const int g0 = blk[4 * (w >> 1) + 3];
                const zyz::Side sd = zyz::side_build<ROT>(block_rec(g0), NR, w & 1);
                sidev[2 * w] = make_float4(sd.t, sd.sc, sd.l0.re, sd.l0.im);
                sidev[2 * w + 1] = make_float4(sd.l1.re, sd.l1.im, __int_as_float(sd.f), 0.f);
                flip = sd.f != 0;
            }
            const unsigned long long bits = __builtin_amdgcn_ballot_w64(flip);
            if (lane == 0) fw[(base >> 6) + wave] = bits;
        }
        lds_sync<WPS>();
        // ================= build B: one thread per block (tables), one per qubit (surface matrix, final left factor)
        for (int w = tid; w < K + NQ; w += THREADS) {
            if (w < K) {
                const int b = w, z = blk[4 * b + 2], g0 = blk[4 * b + 3];
                const ZyzSideInfo* iu = za.sides + 2 * b;
                zyz::Side su = load_side(2 * b), sd = load_side(2 * b + 1), pu = su, pd = sd;
                su.m = parity(iu[0].later); sd.m = parity(iu[1].later);
                const int wu = iu[0].prev, wd = iu[1].prev;
                if (wu >= 0) { pu = load_side(wu); pu.m = su.m ^ su.f; }
                if (wd >= 0) { pd = load_side(wd); pd.m = sd.m ^ sd.f; }
                const float4 r0 = rec[g0];
                // The real scale d_0 = sc_u sc_d of a block is applied once per LAYER: the layer's last block (and the circuit's
                // last block) multiplies by the product over its layer so far, the others by nothing.  In between psi runs ahead
                // by 1 / (d_0 of the blocks passed) in the forward sweep - at most 2 per block - and psi and the cotangent fall
                // behind by phi = the product of d_0 up to and including the block whose sums are being taken in the backward
                // sweep: the chain phase divides the block's sums by phi^2.
                float pre = 1.f;
                for (int i = CPF_ZYZ_LAYER_SCALE ? b - b % LL : b; i <= b; i++) pre *= smem[lds.side + 16 * i + 1] * smem[lds.side + 16 * i + 9];
                const bool carries = !CPF_ZYZ_LAYER_SCALE || (b % LL == LL - 1) || b == K - 1;
                float out[10];
                const int cacb = zyz::block_table(out, su, sd, wu >= 0, pu, wd >= 0, pd, r0.x, r0.y, (z & 256) != 0, carries ? pre : 1.f);
                float* gt = za.tab + ((size_t)s_idx * K + b) * ZYZ_TAB_FLOATS;
                *reinterpret_cast<float4*>(gt) = make_float4(out[0], out[1], out[2], out[3]);
                *reinterpret_cast<float4*>(gt + 4) = make_float4(out[4], out[5], out[6], out[7]);
                *reinterpret_cast<float2*>(gt + 8) = make_float2(out[8], out[9]);
                smem[lds.side + 16 * b + 7] = __int_as_float(su.m);
                smem[lds.side + 16 * b + 15] = __int_as_float(sd.m);
                *reinterpret_cast<float4*>(smem + lds.binfo + 4 * b) = make_float4(__int_as_float(cacb), su.sc * sd.sc, carries ? 1.f : __builtin_amdgcn_rcpf(pre * pre), 0.f);
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
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the tables have reached L2 ...
        lds_sync<WPS>();
        __builtin_amdgcn_s_dcache_inv();                             // ... and no stale line of theirs stays in the scalar cache
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const float* gsample = za.tab + (size_t)s_idx * K * ZYZ_TAB_FLOATS;

        // ================= forward: the (flipped) product state of the surface gates, then the blocks
        {
            // qubit q's factor for row bit r: S~_q[r][column bit of this lane]
            auto sfac = [&](int q, int r) __attribute__((always_inline)) {
                const int cb = (col_i >> (NQ - 1 - q)) & 1;
                const float2 v = *reinterpret_cast<const float2*>(smem + lds.surfm + 8 * q + 4 * r + 2 * cb);
                return f2{v.x, v.y};
            };
            f2 base = {1.f, 0.f};
            bool have = false;
            static_for([&](auto qq) __attribute__((always_inline)) {
                constexpr int Q = decltype(qq)::value;
                if constexpr (L::on_lane(Q)) {
                    const int r = (lane_i >> L::lane_bit(Q)) & 1;
                    const int cb = (col_i >> (NQ - 1 - Q)) & 1;
                    const float2 v = *reinterpret_cast<const float2*>(smem + lds.surfm + 8 * Q + 4 * r + 2 * cb);
                    base = have ? cmul(base, f2{v.x, v.y}) : f2{v.x, v.y};
                    have = true;
                }
            }, typename make_iseq<NQ>::type{});
            zyz_product<L>(psi, base, sfac);
        }
        {
            int b = 0;
            auto fwd_layer = [&](auto tail) __attribute__((always_inline)) {
                constexpr bool TAIL = decltype(tail)::value != 0;        // the truncated last layer: blocks while b < K
                ZyzTab sa, sb;                                       // scalar registers: this block's table and the next one's
                if (!TAIL || b < K) zyz_load_tab(gsample + ZYZ_TAB_FLOATS * b, sa);
                static_for([&](auto jj) __attribute__((always_inline)) {
                    constexpr int J = decltype(jj)::value;
                    constexpr DenseBlock ent = dense_layer_entry<L>(PL::v, LL, J, 0);
                    constexpr int RA = ent.z & 15, RB = (ent.z >> 4) & 15;
                    constexpr bool KEPT = dense_same_swap<L>(PL::v, LL, J - 1, J), KEEP = dense_same_swap<L>(PL::v, LL, J, J + 1);
                    if (TAIL && b >= K) return;
                    auto body = [&](ZyzTab& g, ZyzTab& gnext) __attribute__((always_inline)) {
                        zyz_touch(g);
                        if constexpr (J + 1 < LL) { if (!TAIL || b + 1 < K) zyz_load_tab(gsample + ZYZ_TAB_FLOATS * (b + 1), gnext); }
                        if constexpr (ent.x >= 0 && !KEPT) swap_lane_reg<E, (ent.x >> 8), (ent.x & 255)>(psi);
                        if constexpr (ent.y >= 0) swap_lane_reg<E, (ent.y >> 8), (ent.y & 255)>(psi);
                        zyz_apply<E, RA, RB, !CPF_ZYZ_LAYER_SCALE || TAIL || J == LL - 1>(psi, g);
                        if constexpr (ent.y >= 0) swap_lane_reg<E, (ent.y >> 8), (ent.y & 255)>(psi);
                        if constexpr (ent.x >= 0) {
                            if (!KEEP || b + 1 >= K) swap_lane_reg<E, (ent.x >> 8), (ent.x & 255)>(psi);     // the last block of the circuit always exchanges back
                        }
                        b++;
                    };
                    if constexpr (J & 1) body(sb, sa); else body(sa, sb);
                }, typename make_iseq<LL>::type{});
            };
            for (int full = K / LL; full > 0; full--) fwd_layer(ic<0>{});
            if (b < K) fwd_layer(ic<1>{});
        }

        // ================= loss (matrix_utils.py:35-42) and penalty (main.py:563-564).  The left factors the last blocks leave
        // behind (one diagonal per qubit) meet the target instead of the state: Vd = conj(dfin) V, kept in the cotangent's registers.
        float acc[3] = {0.f, 0.f, 0.f};
        if (a.loss_kind == 2) {                                      // relative phase: moduli only, the diagonal drops out
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
            bool have = false;
            static_for([&](auto qq) __attribute__((always_inline)) {
                constexpr int Q = decltype(qq)::value;
                if constexpr (L::on_lane(Q)) {
                    const int r = (lane_i >> L::lane_bit(Q)) & 1;
                    const float2 v = *reinterpret_cast<const float2*>(smem + lds.lfin + 4 * Q + 2 * r);
                    base = have ? cmul(base, f2{v.x, v.y}) : f2{v.x, v.y};
                    have = true;
                }
            }, typename make_iseq<NQ>::type{});
            zyz_product<L>(cot, base, lfac);
            f2 tt = {0.f, 0.f};
#pragma unroll
            for (int e = 0; e < E; e++) {
                const float2 t = target_at(e);
                cot[e] = cmul_conj(f2{t.x, t.y}, cot[e]);
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
        {   // sum over the workgroup (fixed order: deterministic)
#pragma unroll
            for (int k = 0; k < 3; k++) acc[k] = wave_sum(acc[k]);
            if constexpr (WPS > 1) {
                lds_sync<WPS>();
                if (lane == 0) { red[wave * 3] = acc[0]; red[wave * 3 + 1] = acc[1]; red[wave * 3 + 2] = acc[2]; }
                lds_sync<WPS>();
#pragma unroll
                for (int k = 0; k < 3; k++) { float s = 0.f; for (int w = 0; w < WPS; w++) s += red[w * 3 + k]; acc[k] = s; }
            }
        }
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

        // ================= backward: C = (1/2) dL/dU* = f Vd (or -|V|^2 psi / D for the relative-phase loss)
        if (a.loss_kind == 2) {
#pragma unroll
            for (int e = 0; e < E; e++) {
                const float2 t = target_at(e);
                cot[e] = splat(-a.inv_dim2 * fmaf(t.x, t.x, t.y * t.y)) * psi[e];
            }
        } else {
            const f2 f = {-a.inv_dim2 * tr, -a.inv_dim2 * ti};
#pragma unroll
            for (int e = 0; e < E; e++) cot[e] = cmul(cot[e], f);
        }
        {
            int b = K - 1;
            // one layer, last block first; TAIL: the truncated last layer of the circuit (swept first), blocks J < K % LL only
            auto bwd_layer = [&](auto tail) __attribute__((always_inline)) {
                constexpr bool TAIL = decltype(tail)::value != 0;
                const int tail_len = K % LL;
                ZyzTab g, g2;
                bool first_blk = true;
                static_for([&](auto jj) __attribute__((always_inline)) {
                    constexpr int J = LL - 1 - decltype(jj)::value;
                    if (TAIL && J >= tail_len) return;
                    ZyzTab& gc = (J & 1) ? g2 : g;
                    ZyzTab& gn = (J & 1) ? g : g2;
                    if (first_blk) { zyz_load_tab(gsample + ZYZ_TAB_FLOATS * b, gc); first_blk = false; }
                    zyz_touch(gc);
                    if constexpr (J > 0) zyz_load_tab(gsample + ZYZ_TAB_FLOATS * (b - 1), gn);
                    constexpr DenseBlock ent = dense_layer_entry<L>(PL::v, LL, J, 0);
                    constexpr int RA = ent.z & 15, RB = (ent.z >> 4) & 15;
                    // mirror image of the forward sweep (cpf_dense.h)
                    constexpr bool KEPT = dense_same_swap<L>(PL::v, LL, J, J + 1), KEEP = dense_same_swap<L>(PL::v, LL, J - 1, J);
                    if constexpr (ent.x >= 0) {
                        if (!KEPT || b + 1 >= K) { swap_lane_reg<E, (ent.x >> 8), (ent.x & 255)>(psi); swap_lane_reg<E, (ent.x >> 8), (ent.x & 255)>(cot); }
                    }
                    if constexpr (ent.y >= 0) { swap_lane_reg<E, (ent.y >> 8), (ent.y & 255)>(psi); swap_lane_reg<E, (ent.y >> 8), (ent.y & 255)>(cot); }
                    float d8[8];
                    zyz_backward<E, RA, RB, !CPF_ZYZ_LAYER_SCALE || TAIL || J == LL - 1>(psi, cot, gc, d8);
                    reduce8_rows<true>(d8, sums_w + 32 * b, lane, surf_dst);
                    if constexpr (ent.y >= 0) { swap_lane_reg<E, (ent.y >> 8), (ent.y & 255)>(psi); swap_lane_reg<E, (ent.y >> 8), (ent.y & 255)>(cot); }
                    if constexpr (ent.x >= 0 && !KEEP) { swap_lane_reg<E, (ent.x >> 8), (ent.x & 255)>(psi); swap_lane_reg<E, (ent.x >> 8), (ent.x & 255)>(cot); }
                    b--;
                }, typename make_iseq<LL>::type{});
            };
            if (K % LL) bwd_layer(ic<1>{});                  // blocks beyond the last whole layer come first
            for (int full = K / LL; full > 0; full--) bwd_layer(ic<0>{});
        }
        {   // generator sums of every qubit at the surface state (psi is the product state again, C its cotangent); a lane qubit
            // is exchanged onto register bit 0 for its sums (register qubits first: nothing is exchanged back)
            float ds[8 * NBATCH];
#pragma unroll
            for (int k = 0; k < 8 * NBATCH; k++) ds[k] = 0.f;
            static_for([&](auto qq) __attribute__((always_inline)) {
                constexpr int Q = decltype(qq)::value;
                if constexpr (!L::on_lane(Q)) zyz_gen_sums<E, L::reg_bit(Q)>(psi, cot, ds[3 * Q], ds[3 * Q + 1], ds[3 * Q + 2]);
            }, typename make_iseq<NQ>::type{});
            static_for([&](auto qq) __attribute__((always_inline)) {
                constexpr int Q = decltype(qq)::value;
                if constexpr (L::on_lane(Q)) {
                    swap_lane_reg<E, L::lane_bit(Q), 0>(psi); swap_lane_reg<E, L::lane_bit(Q), 0>(cot);
                    zyz_gen_sums<E, 0>(psi, cot, ds[3 * Q], ds[3 * Q + 1], ds[3 * Q + 2]);
                }
            }, typename make_iseq<NQ>::type{});
#pragma unroll
            for (int bt = 0; bt < NBATCH; bt++) {
                float d8[8];
#pragma unroll
                for (int k = 0; k < 8; k++) d8[k] = ds[8 * bt + k];
                reduce8_rows(d8, gsurf_w + 32 * bt, lane, surf_dst);
            }
        }
        lds_sync<WPS>();

        // ================= chain: one thread per side (its rotation angles; side 0 also the CP angle), one per qubit (surface gate)
        for (int w = tid; w < 2 * K + NQ; w += THREADS) {
            if (w < 2 * K) {
                const int b = w >> 1, side = w & 1, z = blk[4 * b + 2], g0 = blk[4 * b + 3];
                const int kx = ((side == 0) == ((z & 256) != 0)) ? 0 : 3;           // this side's qubit sits on the pair's lo / hi register bit
                const float4 bi = *reinterpret_cast<const float4*>(smem + lds.binfo + 4 * b);          // bi.z: 1 / phi^2 of the block's sums (build B)
                const float px = bi.z * slab_sum(lds.sums, K * 32, b, kx), py = bi.z * slab_sum(lds.sums, K * 32, b, kx + 1), pz = bi.z * slab_sum(lds.sums, K * 32, b, kx + 2);
                const zyz::Side sd = load_side(w);
                float gpart[8];
                zyz::chain_side<ROT>(gpart, px, py, pz, sd, block_rec(g0), NR, side);
#pragma unroll
                for (int i = 0; i < (ROT ? (ROT & 15) : 8); i++) {
                    if (i < zyz::rotc_n<ROT>(NR)) gw[g0 + 1 + 2 * i + side] = gpart[i];
                }
                const bool is_cp = ROT ? ((ROT >> 16) & 3) == 0 : (__float_as_int(rec[g0].z) & 15) == OP_CP;
                if (side == 0 && is_cp) {
                    // the z sums of the two qubits before this block's Y stage: those of the previous block on each (after its
                    // Y stage nothing but diagonals touches the qubit), or of the surface state
                    float pzq[2];
#pragma unroll
                    for (int s = 0; s < 2; s++) {
                        const int wp = za.sides[w + s].prev, q = za.sides[w + s].qubit;
                        if (wp < 0) pzq[s] = slab_sum(lds.gsurf, NBATCH * 32, (3 * q + 2) >> 3, (3 * q + 2) & 7);
                        else {
                            const int zp = blk[4 * (wp >> 1) + 2];
                            pzq[s] = smem[lds.binfo + 4 * (wp >> 1) + 2] * slab_sum(lds.sums, K * 32, wp >> 1, ((((wp & 1) == 0) == ((zp & 256) != 0)) ? 0 : 3) + 2);
                        }
                    }
                    gw[g0] = zyz::chain_cp(bi.z * slab_sum(lds.sums, K * 32, b, 6), pzq[0], pzq[1], __float_as_int(bi.x), bi.y);
                }
            } else {
                const int q = w - 2 * K;
                float pv[3], gs[3];
#pragma unroll
                for (int k = 0; k < 3; k++) pv[k] = slab_sum(lds.gsurf, NBATCH * 32, (3 * q + k) >> 3, (3 * q + k) & 7);
                const float4 r1 = rec[3 * q + 1], r2 = rec[3 * q + 2];
                zyz::chain_surface(gs, pv[0], pv[1], pv[2], __float_as_int(smem[lds.msf + q]), r1.x, r1.y, r2.x, r2.y);
#pragma unroll
                for (int k = 0; k < 3; k++) gw[3 * q + k] = gs[k];
            }
        }
        lds_sync<WPS>();

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
        lds_sync<WPS>();
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
    if constexpr (WPS > 1) {
        lds_sync<WPS>();
        if (lane == 0) { red[wave * 3] = acc[0]; red[wave * 3 + 1] = acc[1]; red[wave * 3 + 2] = acc[2]; }
        lds_sync<WPS>();
#pragma unroll
        for (int k = 0; k < 3; k++) { float s = 0.f; for (int w = 0; w < WPS; w++) s += red[w * 3 + k]; acc[k] = s; }
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

template <int NQ, int LW, int ROT, int... UD>
__global__ __launch_bounds__(64 << LW, 2) void k_cp_zyz(const ZyzArgs za) {
    zyz_body<NQ, LW, ROT, UD...>(za);
}

}  // namespace cpf
