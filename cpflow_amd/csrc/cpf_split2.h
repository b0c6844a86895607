// cpflow_amd: the two-ended 3-qubit kernel with TWO SAMPLES PER WAVE (one row qubit in registers).
//
// cpf_split.h spreads one sample's 8x8 unitary over the 64 lanes of a wave, one amplitude per lane: every rotation costs
// 2 x v_mul_f32_dpp + v_pk_fma_f32, every 8-value gradient reduction 15 instructions, and the loss / owner phases run
// once per sample and wave.  With S = 1024 samples the chip holds one sample per SIMD and the kernel is bound by that
// instruction count.  Here a wave holds two samples, 32 lanes each (3 column bits + 2 row bits on lanes), and the third
// row qubit RQ in a register bit (two amplitudes per lane):
//   - a rotation on RQ is 2 x (v_pk_mul_f32 + v_pk_fma_f32) for BOTH samples (2 instructions per sample instead of 3),
//   - the DPP reduce-scatter of 8 gradient partials serves both samples at once (its 16-lane rows never mix: rows 0, 1
//     belong to sample 0, rows 2, 3 to sample 1),
//   - the trace / loss / best-tracking / owner phases are per wave, i.e. half per sample,
//   - every gate offers two independent chains (the two elements), so that a lone wave per SIMD (1024 samples = 512
//     workgroups of 2 waves) keeps issuing through the 8..10-cycle dependent latencies.
// Same two-ended scheme, stored states and merged diagonal runs as cpf_split.h (U = H2 H1, wave A builds psi = H1, wave B
// builds W = H2^+ V, exchange at the cut, A differentiates backwards, B forwards; wave B owns every angle).  Only the
// K-specialised stored-state form exists (straight-line sweeps, K and the cut m compile-time constants): BASELINE configs
// 2 and 3; every other launch uses cpf_split.h.
//
// Qubits are relabelled internally so that the register qubit is internal qubit 2 and the lane qubits keep their order:
// internal (0, 1, 2) = original (1, 2, 0) | (0, 2, 1) | (0, 1, 2) for RQ = 0 | 1 | 2.
#pragma once
#include "cpf_split.h"

namespace cpf {

struct Lay32 {
    static constexpr int NQ = 3, LW = 0, WPS = 1, THREADS = 64, D = 8;
    static constexpr int NLC = 3, NLR = 2, NRR = 1, E = 2;
    __host__ __device__ static constexpr int row_lane_bit(int i) { return i; }
    __host__ __device__ static constexpr bool row_on_lane(int q) { return q < 2; }
    __host__ __device__ static constexpr int row_bit(int q) { return q < 2 ? q : 0; }
};
// blocks given to wave A.  Static instruction counts of the two waves' steps (chain, K = 12): A = 365 + 131 m (surface
// gates both ways + its blocks), B = 300 + 145 (K - m) (loss, best tracking, owner phase + its blocks); the cut need
// not sit on a layer boundary (every block index is a compile-time constant).
#ifndef CPF_CUT2_M
#define CPF_CUT2_M 0             // A/B builds: this many blocks to wave A
#endif
__host__ __device__ constexpr int split2_point(int K) {
    int m = CPF_CUT2_M > 0 ? CPF_CUT2_M : (145 * K - 65 + 138) / 276;
    if (m > K - 1) m = K - 1;
    return m < 1 ? 1 : m;
}
__host__ __device__ constexpr int s2_orig_of(int rq, int qi) { return rq == 0 ? (qi == 0 ? 1 : (qi == 1 ? 2 : 0)) : (rq == 1 ? (qi == 0 ? 0 : (qi == 1 ? 2 : 1)) : qi); }
__host__ __device__ constexpr int s2_int_of(int rq, int qo) { return s2_orig_of(rq, 0) == qo ? 0 : (s2_orig_of(rq, 1) == qo ? 1 : 2); }

// LDS map (float offsets).  [0, 32): penalty table (shared).  Then one region per sample, then the gradient slabs of the
// workgroup's two waves (one slab each: a slab's 16-lane rows 0, 1 hold sample 0's partial sums, rows 2, 3 sample 1's).
struct Split2Lds {
    static constexpr int G0 = 32, SURF_F = 20, NBATCH = 2;
    __host__ __device__ static constexpr int surf() { return 0; }
    __host__ __device__ static constexpr int blk() { return SURF_F; }
    __host__ __device__ static constexpr int ident(int K) { return blk() + 16 * K; }
    __host__ __device__ static constexpr int xch(int K) { return ident(K) + 16; }                       // 2 x 64 x (re, im)
    __host__ __device__ static constexpr int phase(int K) { return xch(K) + 256; }                      // (K + 2) units x 128
    __host__ __device__ static constexpr int states(int K) { return phase(K) + 128 * (K + 2); }         // (NI K + 2) slots x 128
    __host__ __device__ static constexpr int sample_floats(int K, int ni) { return states(K) + 128 * (ni * K + 2); }
    __host__ __device__ static constexpr int slab(int K, int ni) { return G0 + 2 * sample_floats(K, ni); }           // (NBATCH + K) x 32
    __host__ __device__ static constexpr int total(int K, int ni) { return slab(K, ni) + 32 * (NBATCH + K); }
};

template <int RQ>
struct S2 {
    using L = Lay32;
    using St = State<L>;
    // +1 / -1 by the bit of internal qubit Q of (lane, element e)
    template <int Q, int EI> __device__ __forceinline__ static float sgn(const St& st) { if constexpr (Q < 2) return st.sg[Q]; else return EI ? 1.f : -1.f; }
    template <int Q, int EI> __device__ __forceinline__ static bool bit(const St& st) { if constexpr (Q < 2) return st.lb[Q]; else return EI != 0; }
    // merged diagonal run for element EI: Rz(prev up) Rz(prev dn) CP(up, dn)   (cpf_layered.h diag_phase)
    template <int PUP, int PDN, int UP, int DN, int EI>
    __device__ __forceinline__ static f2 diag(const St& st, float cu, float su, float cd, float sd, float cc, float sc) {
        const f2 zu = f2{cu, su * sgn<PUP, EI>(st)}, zd = f2{cd, sd * sgn<PDN, EI>(st)};
        const bool p = bit<UP, EI>(st) && bit<DN, EI>(st);
        const f2 ze = {p ? cc : 1.f, p ? sc : 0.f};
        return cmul(cmul(zu, zd), ze);
    }
    // -2 on the |11> rows of the pair, else 0 (CP gradient factor)
    template <int UP, int DN, int EI> __device__ __forceinline__ static float cpm2(const St& st) { return (bit<UP, EI>(st) && bit<DN, EI>(st)) ? -2.f : 0.f; }
    // w = Im-type product shared by the gradients of a diagonal run: c.x p.y - c.y p.x
    __device__ __forceinline__ static float wprod(f2 c, f2 p) { return fmaf(c.x, p.y, -c.y * p.x); }
    // this lane's share of a rotation's gradient from C (own) and psi read at the partner (pp[e] = partner of element e)
    template <int KIND, int Q>
    __device__ __forceinline__ static float grad_c_own(const St& st, const f2 (&c)[2], const f2 (&pp)[2]) {
        if constexpr (KIND == OP_RX) return wprod(c[0], pp[0]) + wprod(c[1], pp[1]);
        else if constexpr (KIND == OP_RY) return sgn<Q, 0>(st) * fmaf(c[0].x, pp[0].x, c[0].y * pp[0].y) + sgn<Q, 1>(st) * fmaf(c[1].x, pp[1].x, c[1].y * pp[1].y);
        else return -(sgn<Q, 0>(st) * wprod(c[0], pp[0]) + sgn<Q, 1>(st) * wprod(c[1], pp[1]));
    }
    // roles swapped: psi own, C (= W) read at the partner, whose sign on that qubit is the opposite one
    template <int KIND, int Q>
    __device__ __forceinline__ static float grad_p_own(const St& st, const f2 (&cp)[2], const f2 (&p)[2]) {
        if constexpr (KIND == OP_RX) return wprod(cp[0], p[0]) + wprod(cp[1], p[1]);
        else if constexpr (KIND == OP_RY) return -(sgn<Q, 0>(st) * fmaf(cp[0].x, p[0].x, cp[0].y * p[0].y) + sgn<Q, 1>(st) * fmaf(cp[1].x, p[1].x, cp[1].y * p[1].y));
        else return -(sgn<Q, 0>(st) * wprod(cp[0], p[0]) + sgn<Q, 1>(st) * wprod(cp[1], p[1]));
    }
};

// a stored state: both elements of a lane as one 16-byte word
__device__ __forceinline__ void st_store(float* slot_lane, const f2 (&x)[2]) { *reinterpret_cast<float4*>(slot_lane) = make_float4(x[0].x, x[0].y, x[1].x, x[1].y); }
__device__ __forceinline__ void st_load(const float* slot_lane, f2 (&x)[2]) { const float4 v = *reinterpret_cast<const float4*>(slot_lane); x[0] = f2{v.x, v.y}; x[1] = f2{v.z, v.w}; }

// everything one block's gradient step reads from LDS, fetched one block ahead (the sweeps' own stores fence later loads,
// and a lone wave per SIMD has nothing else to cover an LDS round trip): record, the state before the block's diagonal
// unit at the own lane, its phase, the state after each inner gate at the partner on that gate's qubit
template <int NI> struct Blk2 { float r[16]; f2 own[2]; f2 rho[2]; f2 part[NI][2]; };

// sums over the 32 lanes of each sample (lanes 0..31 / 32..63), returned per lane (its own sample's total)
__device__ __forceinline__ void half_sum2(float& x, float& y, bool hi) {
    x += dppz<0xB1>(x);  y += dppz<0xB1>(y);
    x += dppz<0x4E>(x);  y += dppz<0x4E>(y);
    x += dppz<0x124>(x); y += dppz<0x124>(y);
    x += dppz<0x128>(x); y += dppz<0x128>(y);
    x += dpp<0x142, 0xA, 0xF>(0.f, x); y += dpp<0x142, 0xA, 0xF>(0.f, y);       // row_bcast:15 into rows 1, 3
    const float x0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 31)), x1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
    const float y0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y), 31)), y1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y), 63));
    x = hi ? x1 : x0; y = hi ? y1 : y0;
}
__device__ __forceinline__ float half_sum(float x, bool hi) { float y = 0.f; half_sum2(x, y, hi); return x; }

// ROT: rotation string (cpf_blocks.h rot_pack), KC: block count, RQ: the original qubit kept in registers,
// UD...: the layer in ORIGINAL qubit labels
template <int ROT, int KC, int RQ, int... UD>
__device__ __forceinline__ void split2_body(const KArgs& a, int64_t S) {
    using L = Lay32;
    using H = S2<RQ>;
    using Lds = Split2Lds;
    using PL = PairList<UD...>;
    constexpr int LL = sizeof...(UD) / 2, K = KC;
    constexpr int NR = rot_nr(ROT), NB = 2 * NR + 1, NS = 9, NI = 2 * (NR - 1), P = NS + NB * K;
    constexpr int M = split2_point(K), KB = K - M;
    constexpr int PPT = (P + 63) / 64;                     // register pairs of angles per owner lane (32 lanes x 2 per pair per sample)
    static_assert(layered_merge(3, 0, ROT) && NI >= 2, "split2: rotation string must end in 'z' and have an inner letter");
    static_assert(K >= 2 && K <= 32, "split2: the CP angles must fit half 0 of pair 0");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31;
    const bool hi = lane >= 32;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (a.P != P) return;                                  // host error: plan and specialisation disagree
    const int64_t s_raw = (int64_t)blockIdx.x * 2 + (hi ? 1 : 0);
    const bool vs = s_raw < S;                             // an odd batch: the last workgroup's second sample is a dummy copy of the first
    const int64_t s_idx = vs ? s_raw : S - 1;
    const float* init = a.init + s_idx * P;
    const uint8_t* frozen = a.frozen ? a.frozen + s_idx * P : nullptr;
    float* out_best = a.out_params + (s_idx * 2 + 1) * P;
    float* sb = smem + Lds::G0 + (hi ? Lds::sample_floats(K, NI) : 0);          // this lane's sample region
    float* gs = smem + Lds::slab(K, NI);
    float* xch_l = sb + Lds::xch(K) + 4 * l32;             // [0]: psi of wave A, [128]: W of wave B
    float* ph_l = sb + Lds::phase(K) + 4 * l32 + (wave ? 128 * (M + 1) : 0);
    float* sp_l = sb + Lds::states(K) + 4 * l32 + (wave ? 128 * (NI * M + 1) : 0);
    const float* rec = sb + Lds::blk();

    typename H::St st;
#pragma unroll
    for (int i = 0; i < 2; i++) { st.lb[i] = (lane >> i) & 1; st.sg[i] = st.lb[i] ? 1.f : -1.f; st.sgp[i] = f2{1.f, st.sg[i]}; }
    // original row index of (lane, element), column index
    auto row_of = [&](int e) {
        int r = 0;
#pragma unroll
        for (int qo = 0; qo < 3; qo++) { const int qi = s2_int_of(RQ, qo); r |= (qi < 2 ? ((lane >> qi) & 1) : e) << (2 - qo); }
        return r;
    };
    const int col = (lane >> 2) & 7;
    f2 vt[2];
#pragma unroll
    for (int e = 0; e < 2; e++) { const float2 t = a.target[row_of(e) * 8 + col]; vt[e] = f2{t.x, t.y}; }
    const int dst_lane = reduce8_dst_lane(lane);
    // partner-lane displacement (in floats) of the stored states for lane qubit q
    const int pdisp0 = 4 * ((l32 ^ 1) - l32), pdisp1 = 4 * ((l32 ^ 2) - l32);

    // ---- owner state (wave B): slot u = l32 + 32 (2 s + h) of this lane's sample; the K <= 32 CP angles come first
    f2 th[PPT], mm[PPT], vv[PPT], hscale[PPT], nlr[PPT];
    float psl = 0.f, pval = 0.f;
    int cs_off[PPT][2], g_off[PPT][2], pj[PPT][2];
    bool valid[PPT][2];
    bool is_cp0 = false;
    penalty_table_init(smem, a, tid);
#pragma unroll
    for (int s = 0; s < PPT; s++) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int u = l32 + 32 * (2 * s + h);
            valid[s][h] = wave != 0 && u < P;
            int j = 0;
            if (valid[s][h]) {
                if (u < K) j = NS + NB * u + NB - 1;
                else if (u - K < NS) j = u - K;
                else { const int q = u - K - NS; j = NS + NB * (q / (NB - 1)) + q % (NB - 1); }
            }
            pj[s][h] = j;
            const bool cp = valid[s][h] && a.pkind[j];
            if (s == 0 && h == 0) is_cp0 = cp;
            const bool frz = valid[s][h] && frozen && frozen[j];
            // gradient slots: owner reads its sample's two row partials at g_off + 2 * (sample)
            if (j < NS) { cs_off[s][h] = Lds::surf() + 2 * j; g_off[s][h] = (j < 8 ? 4 * j : 32 * Lds::NBATCH + 4 * 7) + (hi ? 2 : 0); }
            else {
                const int bb = (j - NS) / NB, k = (j - NS) % NB;
                cs_off[s][h] = Lds::blk() + 16 * bb + 2 * k; g_off[s][h] = 32 * Lds::NBATCH + 32 * bb + 4 * k + (hi ? 2 : 0);
            }
            const float t0 = valid[s][h] ? init[j] : 0.f;
            const float hs = cp ? 1.f : 0.5f;
            const float nl = (valid[s][h] && !frz) ? -a.lr : 0.f;
            if (h == 0) { th[s].x = t0; hscale[s].x = hs; nlr[s].x = nl; } else { th[s].y = t0; hscale[s].y = hs; nlr[s].y = nl; }
            if (valid[s][h] && vs) { a.out_params[(s_idx * 2) * P + j] = t0; out_best[j] = t0; }
        }
        mm[s] = f2{0.f, 0.f}; vv[s] = f2{0.f, 0.f};
    }
    if (tid < 64) {                                       // per sample: zero pad of the block records, identity record
        for (int b = l32; b < K; b += 32) { sb[Lds::blk() + 16 * b + 14] = 0.f; sb[Lds::blk() + 16 * b + 15] = 0.f; }
        if (l32 < 16) sb[Lds::ident(K) + l32] = (l32 & 1) ? 0.f : 1.f;
    }
    auto publish = [&](int s) __attribute__((always_inline)) {
        const bool pen = s == 0 && a.use_reg && is_cp0;
        const bool lin = pen && a.pen_kind == 0;
        float x = 0.f;
        float2 si = make_float2(0.f, 0.f);
        if (lin) { x = floor_mod_2pi(th[0].x); si = penalty_lookup(smem, x); }
        f2 sn, cn;
        fast_sincos2(th[s] * hscale[s], sn, cn);
        if (valid[s][0]) *reinterpret_cast<float2*>(sb + cs_off[s][0]) = make_float2(cn.x, sn.x);
        if (valid[s][1]) *reinterpret_cast<float2*>(sb + cs_off[s][1]) = make_float2(cn.y, sn.y);
        if (lin) { pval = __fadd_rn(__fmul_rn(si.x, x), si.y); psl = a.r * si.x; }
        else if (pen) { float sl; pval = penalty_lds(smem, a.pen_kind, th[0].x, sl); psl = a.r * sl; }
    };
    split_barrier();
    if (wave != 0) {
#pragma unroll
        for (int s = 0; s < PPT; s++) publish(s);
    }
    split_barrier();

    // internal pair of block b (compile-time)
#define S2_UP(b) s2_int_of(RQ, PL::v[2 * ((b) % LL)])
#define S2_DN(b) s2_int_of(RQ, PL::v[2 * ((b) % LL) + 1])
    // inner gate G of a block: letter G / 2, on up (even G) or dn (odd G)
    auto inner_apply = [&](auto bb, auto gg, f2 (&x)[2], const float (&r)[16], bool inverse) __attribute__((always_inline)) {
        constexpr int B = decltype(bb)::value, G = decltype(gg)::value;
        constexpr int KIND = rot_kind(ROT, G / 2), Q = (G & 1) ? S2_DN(B) : S2_UP(B), CS = 4 * (G / 2) + 2 * (G & 1);
        rot_apply<L, KIND, Q>(x, r[CS], inverse ? -r[CS + 1] : r[CS + 1], st);
    };
    // partner read of a stored state for inner gate G of block B: psi / W of the lane that differs in the gate's qubit
    auto load_part = [&](auto bb, auto gg, const float* slot_lane, f2 (&pp)[2]) __attribute__((always_inline)) {
        constexpr int B = decltype(bb)::value, G = decltype(gg)::value;
        constexpr int KIND = rot_kind(ROT, G / 2), Q = (G & 1) ? S2_DN(B) : S2_UP(B);
        if constexpr (KIND == OP_RZ) st_load(slot_lane, pp);
        else if constexpr (Q == 0) st_load(slot_lane + pdisp0, pp);
        else if constexpr (Q == 1) st_load(slot_lane + pdisp1, pp);
        else { f2 t[2]; st_load(slot_lane, t); pp[0] = t[1]; pp[1] = t[0]; }          // register qubit: the other element of this lane
    };

    float best = 0.f, first = 0.f;
    const int iters = a.T > 0 ? a.T : 1;
    for (int it = 0; it < iters; it++) {
        const float2 bc = const_load_f2(a.bias, a.T > 0 ? it : 0);
        float pen_sum = 0.f;
        // ================= forward: A builds H1 from the identity, B builds W = H2^+ V
        if (wave == 0) {
#pragma unroll
            for (int e = 0; e < 2; e++) st.p[e] = f2{(row_of(e) == col) ? 1.f : 0.f, 0.f};
            float sr[Lds::SURF_F], ra[16], rb[16];
#pragma unroll
            for (int i = 0; i < Lds::SURF_F / 4; i++) {
                const float4 v = reinterpret_cast<const float4*>(sb + Lds::surf())[i];
                sr[4 * i] = v.x; sr[4 * i + 1] = v.y; sr[4 * i + 2] = v.z; sr[4 * i + 3] = v.w;
            }
            load_rec(rec, ra);
            static_for([&](auto qq) __attribute__((always_inline)) {       // surface gate Rz(t2) Rx(t1) Rz(t0) of every qubit, main.py:122-124
                constexpr int QI = decltype(qq)::value, QO = s2_orig_of(RQ, QI);
                rot_apply<L, OP_RZ, QI>(st.p, sr[6 * QO], sr[6 * QO + 1], st);
                rot_apply<L, OP_RX, QI>(st.p, sr[6 * QO + 2], sr[6 * QO + 3], st);
                rot_apply<L, OP_RZ, QI>(st.p, sr[6 * QO + 4], sr[6 * QO + 5], st);
            }, typename make_iseq<3>::type{});
            st_store(sp_l, st.p);
            float pz[4] = {1.f, 0.f, 1.f, 0.f};
            static_for([&](auto bb) __attribute__((always_inline)) {
                constexpr int B = decltype(bb)::value, BP = B > 0 ? B - 1 : 0;
                float (&r)[16] = (B & 1) ? rb : ra;
                float (&rn)[16] = (B & 1) ? ra : rb;
                if constexpr (B + 1 < M) load_rec(rec + 16 * (B + 1), rn);       // one block ahead: the stores below fence later loads
                f2 rho[2];
                rho[0] = H::template diag<S2_UP(BP), S2_DN(BP), S2_UP(B), S2_DN(B), 0>(st, pz[0], pz[1], pz[2], pz[3], r[2 * (NB - 1)], r[2 * (NB - 1) + 1]);
                rho[1] = H::template diag<S2_UP(BP), S2_DN(BP), S2_UP(B), S2_DN(B), 1>(st, pz[0], pz[1], pz[2], pz[3], r[2 * (NB - 1)], r[2 * (NB - 1) + 1]);
                st_store(ph_l + 128 * B, rho);
                st.p[0] = cmul(st.p[0], rho[0]); st.p[1] = cmul(st.p[1], rho[1]);
                static_for([&](auto gg) __attribute__((always_inline)) {
                    inner_apply(bb, gg, st.p, r, false);
                    st_store(sp_l + 128 * (NI * B + 1 + decltype(gg)::value), st.p);
                }, typename make_iseq<NI>::type{});
#pragma unroll
                for (int i = 0; i < 4; i++) pz[i] = r[4 * (NR - 1) + i];
            }, typename make_iseq<M>::type{});
            {   // final unit: the Rz pair of block M - 1, not followed by a CP of this half
                f2 rho[2];
                rho[0] = H::template diag<S2_UP(M - 1), S2_DN(M - 1), S2_UP(M - 1), S2_DN(M - 1), 0>(st, pz[0], pz[1], pz[2], pz[3], 1.f, 0.f);
                rho[1] = H::template diag<S2_UP(M - 1), S2_DN(M - 1), S2_UP(M - 1), S2_DN(M - 1), 1>(st, pz[0], pz[1], pz[2], pz[3], 1.f, 0.f);
                st_store(ph_l + 128 * M, rho);
                st.p[0] = cmul(st.p[0], rho[0]); st.p[1] = cmul(st.p[1], rho[1]);
            }
            st_store(xch_l, st.p);
        } else {
            if (a.use_reg) pen_sum = half_sum(pval, hi);
            st.p[0] = vt[0]; st.p[1] = vt[1];
            float ra[16], rb[16];
            load_rec(rec + 16 * (K - 1), ra);
            {   // last unit of the circuit: the Rz pair of block K - 1
                f2 rho[2];
                rho[0] = H::template diag<S2_UP(K - 1), S2_DN(K - 1), S2_UP(K - 1), S2_DN(K - 1), 0>(st, ra[4 * (NR - 1)], ra[4 * (NR - 1) + 1], ra[4 * (NR - 1) + 2], ra[4 * (NR - 1) + 3], 1.f, 0.f);
                rho[1] = H::template diag<S2_UP(K - 1), S2_DN(K - 1), S2_UP(K - 1), S2_DN(K - 1), 1>(st, ra[4 * (NR - 1)], ra[4 * (NR - 1) + 1], ra[4 * (NR - 1) + 2], ra[4 * (NR - 1) + 3], 1.f, 0.f);
                st_store(ph_l + 128 * KB, rho);
                st.p[0] = cmul_conj(st.p[0], rho[0]); st.p[1] = cmul_conj(st.p[1], rho[1]);
            }
            static_for([&](auto jj) __attribute__((always_inline)) {       // local blocks KB - 1 .. 0 = global M + b
                constexpr int J = decltype(jj)::value, BL = KB - 1 - J, B = M + BL;
                float (&r)[16] = (J & 1) ? rb : ra;
                float (&rp)[16] = (J & 1) ? ra : rb;                        // the previous block's record: next round's own record
                if constexpr (BL > 0) load_rec(rec + 16 * (B - 1), rp); else load_rec(sb + Lds::ident(K), rp);
                static_for([&](auto gg) __attribute__((always_inline)) {
                    constexpr int G = NI - 1 - decltype(gg)::value;
                    st_store(sp_l + 128 * (NI * BL + 1 + G), st.p);           // the state after gate G (forward-circuit order)
                    inner_apply(ic<B>{}, ic<G>{}, st.p, r, true);
                }, typename make_iseq<NI>::type{});
                constexpr int BP = B - 1;                                   // for BL == 0 the identity record stands in (block M - 1's Rz pair sits in A's last unit)
                f2 rho[2];
                rho[0] = H::template diag<S2_UP(BP), S2_DN(BP), S2_UP(B), S2_DN(B), 0>(st, rp[4 * (NR - 1)], rp[4 * (NR - 1) + 1], rp[4 * (NR - 1) + 2], rp[4 * (NR - 1) + 3], r[2 * (NB - 1)], r[2 * (NB - 1) + 1]);
                rho[1] = H::template diag<S2_UP(BP), S2_DN(BP), S2_UP(B), S2_DN(B), 1>(st, rp[4 * (NR - 1)], rp[4 * (NR - 1) + 1], rp[4 * (NR - 1) + 2], rp[4 * (NR - 1) + 3], r[2 * (NB - 1)], r[2 * (NB - 1) + 1]);
                st_store(ph_l + 128 * BL, rho);
                st.p[0] = cmul_conj(st.p[0], rho[0]); st.p[1] = cmul_conj(st.p[1], rho[1]);
            }, typename make_iseq<KB>::type{});
            st_store(xch_l + 128, st.p);
        }
        split_barrier();
        // ================= loss (both waves, identical arithmetic), best tracking (wave B)
        f2 w[2];                                          // W = H2^+ V at the cut
        if (wave == 0) st_load(xch_l + 128, w);
        else { w[0] = st.p[0]; w[1] = st.p[1]; st_load(xch_l, st.p); }
        const f2 tt = cmul_conj(st.p[0], w[0]) + cmul_conj(st.p[1], w[1]);       // T = sum psi conj(W), this lane's share
        float tr = tt.x, ti = tt.y;
        half_sum2(tr, ti, hi);
        bool improved = false;
        if (wave != 0) {
            const float reg = a.use_reg ? a.r * pen_sum : 0.f;
            const float ab = sqrtf(fmaf(tr, tr, ti * ti));
            const float loss = 1.f - ab * ab * a.inv_dim2;
            const float regloss = loss + reg;
            if (it == 0) { first = regloss; best = regloss; }
            else if (regloss < best) { best = regloss; improved = true; }
            if (improved && vs) {
#pragma unroll
                for (int s = 0; s < PPT; s++) {
                    if (valid[s][0]) out_best[pj[s][0]] = th[s].x;
                    if (valid[s][1]) out_best[pj[s][1]] = th[s].y;
                }
            }
            if (a.hist_loss && l32 == 0 && vs) a.hist_loss[s_idx * a.T + it] = regloss;
            if (a.hist_params && vs) {
#pragma unroll
                for (int s = 0; s < PPT; s++) {
                    if (valid[s][0]) a.hist_params[(s_idx * a.T + it) * P + pj[s][0]] = th[s].x;
                    if (valid[s][1]) a.hist_params[(s_idx * a.T + it) * P + pj[s][1]] = th[s].y;
                }
            }
        }
        if (a.T == 0) break;

        // ================= backward: C = f W at the cut (wave A); psi' = conj(f) H1 (wave B: f moves onto psi)
        const f2 fco = f2{-a.inv_dim2 * tr, -a.inv_dim2 * ti};
        if (wave == 0) {
            st.c[0] = cmul(w[0], fco); st.c[1] = cmul(w[1], fco);
            float d[8], d0[8];
            float dzu, dzd;
            Blk2<NI> qa, qb;
            float sr[Lds::SURF_F];
            auto load_a = [&](auto bb, Blk2<NI>& q) __attribute__((always_inline)) {
                constexpr int B = decltype(bb)::value;
                load_rec(rec + 16 * B, q.r);
                st_load(sp_l + 128 * (NI * B), q.own);                      // psi before unit B
                st_load(ph_l + 128 * B, q.rho);
                static_for([&](auto gg) __attribute__((always_inline)) {
                    load_part(bb, gg, sp_l + 128 * (NI * B + 1 + decltype(gg)::value), q.part[decltype(gg)::value]);
                }, typename make_iseq<NI>::type{});
            };
            load_a(ic<M - 1>{}, ((M - 1) & 1) ? qb : qa);
            {   // unit M (registers): gradients of block M - 1's final Rz pair, then C across the unit
                f2 rho[2];
                st_load(ph_l + 128 * M, rho);
                const float w0 = H::wprod(st.c[0], st.p[0]), w1 = H::wprod(st.c[1], st.p[1]);
                dzu = -(H::template sgn<S2_UP(M - 1), 0>(st) * w0 + H::template sgn<S2_UP(M - 1), 1>(st) * w1);
                dzd = -(H::template sgn<S2_DN(M - 1), 0>(st) * w0 + H::template sgn<S2_DN(M - 1), 1>(st) * w1);
                st.c[0] = cmul_conj(st.c[0], rho[0]); st.c[1] = cmul_conj(st.c[1], rho[1]);
            }
            static_for([&](auto jj) __attribute__((always_inline)) {
                constexpr int B = M - 1 - decltype(jj)::value, BP = B > 0 ? B - 1 : 0;
                Blk2<NI>& cur = (B & 1) ? qb : qa;
                Blk2<NI>& nxt = (B & 1) ? qa : qb;
                if constexpr (B > 0) load_a(ic<BP>{}, nxt);
                else {
#pragma unroll
                    for (int i = 0; i < Lds::SURF_F / 4; i++) {
                        const float4 v = reinterpret_cast<const float4*>(sb + Lds::surf())[i];
                        sr[4 * i] = v.x; sr[4 * i + 1] = v.y; sr[4 * i + 2] = v.z; sr[4 * i + 3] = v.w;
                    }
                }
#pragma unroll
                for (int k = 0; k < 8; k++) d[k] = 0.f;
                static_for([&](auto gg) __attribute__((always_inline)) {
                    constexpr int G = NI - 1 - decltype(gg)::value;
                    constexpr int KIND = rot_kind(ROT, G / 2), Q = (G & 1) ? S2_DN(B) : S2_UP(B);
                    d[G] = H::template grad_c_own<KIND, Q>(st, st.c, cur.part[G]);
                    inner_apply(ic<B>{}, ic<G>{}, st.c, cur.r, true);
                }, typename make_iseq<NI>::type{});
                st.c[0] = cmul_conj(st.c[0], cur.rho[0]); st.c[1] = cmul_conj(st.c[1], cur.rho[1]);     // C now sits before unit B
                const float w0 = H::wprod(st.c[0], cur.own[0]), w1 = H::wprod(st.c[1], cur.own[1]);
                d[2 * (NR - 1)] = dzu; d[2 * (NR - 1) + 1] = dzd;
                d[NB - 1] = H::template cpm2<S2_UP(B), S2_DN(B), 0>(st) * w0 + H::template cpm2<S2_UP(B), S2_DN(B), 1>(st) * w1;
                dzu = -(H::template sgn<S2_UP(BP), 0>(st) * w0 + H::template sgn<S2_UP(BP), 1>(st) * w1);
                dzd = -(H::template sgn<S2_DN(BP), 0>(st) * w0 + H::template sgn<S2_DN(BP), 1>(st) * w1);
                if constexpr (B > 0) reduce8_rows<true>(d, gs + 32 * Lds::NBATCH + 32 * B, lane, dst_lane);
                else {
#pragma unroll
                    for (int k = 0; k < 8; k++) d0[k] = d[k];
                    st.p[0] = cur.own[0]; st.p[1] = cur.own[1];
                }
            }, typename make_iseq<M>::type{});
            // surface gates backwards (psi and C after the surface layer)
            float ds[9];
            static_for([&](auto qq) __attribute__((always_inline)) {       // per qubit from the state after the WHOLE layer (gates on different qubits commute)
                constexpr int QI = decltype(qq)::value, QO = s2_orig_of(RQ, QI);
                typename H::St s = st;
                ds[3 * QO + 2] = rot_grad<L, OP_RZ, QI>(s);
                rot_apply<L, OP_RZ, QI>(s.p, sr[6 * QO + 4], -sr[6 * QO + 5], s);
                rot_apply<L, OP_RZ, QI>(s.c, sr[6 * QO + 4], -sr[6 * QO + 5], s);
                ds[3 * QO + 1] = rot_grad<L, OP_RX, QI>(s);
                rot_apply<L, OP_RX, QI>(s.p, sr[6 * QO + 2], -sr[6 * QO + 3], s);
                rot_apply<L, OP_RX, QI>(s.c, sr[6 * QO + 2], -sr[6 * QO + 3], s);
                ds[3 * QO + 0] = rot_grad<L, OP_RZ, QI>(s);
            }, typename make_iseq<3>::type{});
            {
                float d8[8];
#pragma unroll
                for (int k = 0; k < 8; k++) d8[k] = ds[k];
                reduce8_rows(d8, gs, lane, dst_lane);
            }
            d0[7] = ds[8];                                 // the ninth surface gradient rides in block 0's spare slot
            reduce8_rows(d0, gs + 32 * Lds::NBATCH, lane, dst_lane);
        } else {
            float da[8], db[8];
#pragma unroll
            for (int k = 0; k < 8; k++) { da[k] = 0.f; db[k] = 0.f; }
            Blk2<NI> qa, qb;
            f2 held_last[2];
            auto load_b = [&](auto ll, Blk2<NI>& q) __attribute__((always_inline)) {
                constexpr int BL = decltype(ll)::value, B = M + BL;
                load_rec(rec + 16 * B, q.r);
                if constexpr (BL > 0) st_load(sp_l + 128 * (NI * BL), q.own); else { q.own[0] = w[0]; q.own[1] = w[1]; }   // W before unit BL
                st_load(ph_l + 128 * BL, q.rho);
                static_for([&](auto gg) __attribute__((always_inline)) {
                    load_part(ic<B>{}, gg, sp_l + 128 * (NI * BL + 1 + decltype(gg)::value), q.part[decltype(gg)::value]);
                }, typename make_iseq<NI>::type{});
            };
            load_b(ic<0>{}, qa);
            st.p[0] = cmul_conj(st.p[0], fco); st.p[1] = cmul_conj(st.p[1], fco);
            static_for([&](auto jj) __attribute__((always_inline)) {       // local blocks 0 .. KB - 1 APPLIED to psi'
                constexpr int BL = decltype(jj)::value, B = M + BL, BP = B - 1;
                float (&dcur)[8] = (BL & 1) ? db : da;
                float (&dprv)[8] = (BL & 1) ? da : db;
                Blk2<NI>& cur = (BL & 1) ? qb : qa;
                Blk2<NI>& nxt = (BL & 1) ? qa : qb;
                if constexpr (BL + 1 < KB) load_b(ic<BL + 1>{}, nxt);
                else st_load(sp_l + 128 * (NI * KB), held_last);            // W after the last inner gate of block K - 1
                const float w0 = H::wprod(cur.own[0], st.p[0]), w1 = H::wprod(cur.own[1], st.p[1]);
                st.p[0] = cmul(st.p[0], cur.rho[0]); st.p[1] = cmul(st.p[1], cur.rho[1]);
                dprv[2 * (NR - 1)] = -(H::template sgn<S2_UP(BP), 0>(st) * w0 + H::template sgn<S2_UP(BP), 1>(st) * w1);
                dprv[2 * (NR - 1) + 1] = -(H::template sgn<S2_DN(BP), 0>(st) * w0 + H::template sgn<S2_DN(BP), 1>(st) * w1);
                if constexpr (BL > 0) reduce8_rows<true>(dprv, gs + 32 * Lds::NBATCH + 32 * (B - 1), lane, dst_lane);
#pragma unroll
                for (int k = 0; k < 8; k++) dcur[k] = 0.f;
                dcur[NB - 1] = H::template cpm2<S2_UP(B), S2_DN(B), 0>(st) * w0 + H::template cpm2<S2_UP(B), S2_DN(B), 1>(st) * w1;
                static_for([&](auto gg) __attribute__((always_inline)) {
                    constexpr int G = decltype(gg)::value;
                    constexpr int KIND = rot_kind(ROT, G / 2), Q = (G & 1) ? S2_DN(B) : S2_UP(B);
                    inner_apply(ic<B>{}, gg, st.p, cur.r, false);
                    dcur[G] = H::template grad_p_own<KIND, Q>(st, cur.part[G], st.p);
                }, typename make_iseq<NI>::type{});
            }, typename make_iseq<KB>::type{});
            {   // last unit: the Rz pair of block K - 1
                const float w0 = H::wprod(held_last[0], st.p[0]), w1 = H::wprod(held_last[1], st.p[1]);
                float (&dl)[8] = ((KB - 1) & 1) ? db : da;
                dl[2 * (NR - 1)] = -(H::template sgn<S2_UP(K - 1), 0>(st) * w0 + H::template sgn<S2_UP(K - 1), 1>(st) * w1);
                dl[2 * (NR - 1) + 1] = -(H::template sgn<S2_DN(K - 1), 0>(st) * w0 + H::template sgn<S2_DN(K - 1), 1>(st) * w1);
                reduce8_rows<true>(dl, gs + 32 * Lds::NBATCH + 32 * (K - 1), lane, dst_lane);
            }
        }
        split_barrier();

        // ================= Adam (optax 0.1.1) + next step's (cos, sin) / penalty: wave B, two angles per lane and operation
        if (wave != 0) {
            const f2 ibc = f2{__builtin_amdgcn_rcpf(bc.x), __builtin_amdgcn_rcpf(bc.y)};
#pragma unroll
            for (int s = 0; s < PPT; s++) {
                const float2 g0 = *reinterpret_cast<const float2*>(gs + g_off[s][0]);
                const float2 g1 = *reinterpret_cast<const float2*>(gs + g_off[s][1]);
                f2 g = f2{g0.x + g0.y, g1.x + g1.y};
                if (s == 0 && a.use_reg && is_cp0) g.x += psl;
                const f2 mo = pk_fma(splat(a.b1), mm[s], splat(a.omb1) * g);
                const f2 v = pk_fma(splat(a.b2), vv[s], splat(a.omb2) * (g * g));
                mm[s] = mo; vv[s] = v;
                const f2 vh = v * splat(ibc.y);
                const f2 den = f2{__builtin_amdgcn_sqrtf(vh.x), __builtin_amdgcn_sqrtf(vh.y)} + splat(a.eps);
                const f2 upd = (mo * splat(ibc.x)) * f2{__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
                th[s] = pk_fma(nlr[s], upd, th[s]);
                publish(s);
            }
        }
        split_barrier();
    }
#undef S2_UP
#undef S2_DN

    // ================= epilogue (optimization.py:365-370, cp_utils.py:144-164): CP owners, first and best all sit in wave B
    if (wave != 0) {
        const bool bi = best < first;
        float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f;
        if (is_cp0) {
            const int j = pj[0][0];
            const float t0 = a.out_params[(s_idx * 2) * P + j], t1 = out_best[j];
            float sl;
            if (a.use_reg) { acc0 = penalty_lds(smem, a.pen_kind, t0, sl); acc1 = penalty_lds(smem, a.pen_kind, t1, sl); }
            acc2 = cz_value(bi ? t1 : t0, a.thr_cp);
        }
        half_sum2(acc0, acc1, hi);
        acc2 = half_sum(acc2, hi);
        if (l32 == 0 && vs) {
            const float reg0 = a.use_reg ? a.r * acc0 : 0.f, reg1 = a.use_reg ? a.r * acc1 : 0.f;
            const float l0 = first - reg0, l1 = best - reg1;
            a.out_regloss[s_idx * 2 + 0] = first; a.out_regloss[s_idx * 2 + 1] = best;
            a.out_reg[s_idx * 2 + 0] = reg0;      a.out_reg[s_idx * 2 + 1] = reg1;
            a.out_loss[s_idx * 2 + 0] = l0;       a.out_loss[s_idx * 2 + 1] = l1;
            a.out_cz[s_idx] = (int32_t)acc2;
            a.out_pass[s_idx] = ((bi ? l1 : l0) <= a.entry_loss) ? 1 : 0;
        }
    }
}

struct Split2Args { KArgs k; int64_t S; };

template <int ROT, int KC, int RQ, int... UD>
__global__ __launch_bounds__(128, 1) void k_cp_split2(const Split2Args a) {
    split2_body<ROT, KC, RQ, UD...>(a.k, a.S);
}

}  // namespace cpf
