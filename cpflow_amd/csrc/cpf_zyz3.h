// cpflow_amd: the two-ended ZYZ-form kernel for 3 qubits - BASELINE configs 1-3 (round 5).
//
// What it replaces, per optimiser step and sample (cpflow/main.py:69-82,106-146 build_unitary, matrix_utils.py:35-42 cost_HST,
// penalty.py:44-71, optimization.py:14-25,61-75,340): the work of cpf_split.h, in the algebra of cpf_zyz_math.h.
//
// One amplitude of the 8 x 8 unitary per lane (Lay<3, 0>: every row qubit on a DPP-reachable lane bit), two waves per sample:
//   wave A   psi(s_b), b = -1 .. M-1: the (flipped) product state of the surface gates, then blocks 0 .. M-1 applied
//   wave B   W(s_b), b = K-1 .. M-1: Vd = conj(dfin) V, then blocks K-1 .. M un-applied; it also OWNS every angle
// with s_b the slice right after the two real stages of block b (cpf_zyz_math.h: what the sweeps hold there is the true state
// up to pending flips and the left diagonals not yet merged).  A block is ONE complex multiply by the lane's merged phase and two
// stages x += (+-t) x[partner] (two v_fmac_f32_dpp each): 6 vector instructions where the gate-by-gate sweep of cpf_split.h spends
// 14 (four x / y rotations at 2 DPP + 1 packed, one merged diagonal).  Every state of the forward halves stays in LDS (K + 2 slots
// of 512 B), so the gradient halves move ONE matrix: wave A un-applies blocks M-1 .. 0 from the cotangent w = f W and reads psi(s_b)
// back at the lane and at its two partner lanes; wave B applies blocks M .. K-1 to psi' = conj(f) psi and reads W(s_b) back.
// At a slice the 7 generator sums of cpf_zyz_math.h come from three products per lane:
//     conj(w) psi            -> Pz of the block's two qubits (signed by the lane's bits), the |11> sum of the NEXT block's pair
//     conj(w) psi[partner]   -> Px (imaginary part), Py (real part, signed), once per qubit of the block
// (tools/zyz3_model.py is the executable specification, checked against tools/zyz_model.py and central differences).
//
// The sums do not cross the lanes in the sweeps at all: every lane writes its 7 shares to LDS (one 272-byte row per sum: 64 lanes
// + padding, conflict-free both ways), and after the backward halves the owner wave adds each row with one lane per sum - 16
// ds_read_b128 and 33 packed adds per 64 sums where the DPP reduce-scatter of cpf_blocks.h costs 28 issue slots per block in the
// wave that is on the critical path.  With one state per block instead of one per inner gate the rows fit next to the states
// (config 2: 25 KB + 7 KB per sample, four samples per CU).
//
// Owner (wave B, one lane per item: side w = 2 b + s of a block, or a qubit's surface gate): Adam, (cos, sin), the side's factors
// (cpf_zyz_math.h side_build), the flip bits by ONE ballot (all items sit in one wave, so pending parities are a popcount against
// a per-lane mask - no LDS), the side's e-pair, then one lane per table ENTRY (4 per block) builds what a sweep lane reads with
// one ds_read_b128: (d.re, d.im, signed t_up, signed t_dn).  After the sweeps: row sums, then the chain rule per item
// (chain_side, chain_cp, chain_surface) straight into the owner's registers.
//
// Requires: n = 3, 1..3 rotation letters (any), CP or CZ blocks, loss kinds with a linear cotangent (HST, state preparation),
// 2 K + 3 <= 64 items, K and the layer compile-time constants (instances: ahead of time for configs 2 / 3, cpf_jit.h otherwise).
#pragma once
#include "cpf_layered.h"
#include "cpf_zyz_math.h"

namespace cpf {

constexpr int ZYZ3_ROW = 68;                     // floats per row of shares: 64 lanes + 4 (16 lanes of one ds_read_b128 pass hit 16 different bank groups)

// bytes of LDS per sample of an instance with K blocks (Zyz3Lds<K>::TOTAL floats)
__host__ __device__ constexpr int zyz3_lds_floats(int K) {
    const int rows = 10 + 7 * K;
    const int tab = (PEN_FLOATS + 26 * K + 3) & ~3, tot = tab + 16 * K + 24, states = (tot + rows + 3) & ~3;
    return states + 128 * (K + 2) + rows * ZYZ3_ROW;
}
__host__ __device__ constexpr bool zyz3_supported(int K, int n_rot) { return K >= 2 && 2 * K + 3 <= 64 && n_rot >= 1 && n_rot <= 3; }

template <int K>
struct Zyz3Lds {                                  // float offsets, per sample; [0, 28) = penalty table
    static constexpr int up4(int x) { return (x + 3) & ~3; }
    static constexpr int ROWS = 10 + 7 * K;       // 9 surface sums, S11 of block 0, then 7 per block (Px Py Pz up, Px Py Pz dn, S11 of the next block)
    static constexpr int REC1 = PEN_FLOATS;       // 2 K x {l0, l1}: left factors of a side
    static constexpr int REC2 = REC1 + 8 * K;     // 2 K x {e0, e1}: the side's e-pair
    static constexpr int REC3 = REC2 + 8 * K;     // 2 K x {t, sc, c, -}
    static constexpr int REC4 = REC3 + 8 * K;     // K x {cos, sin} of the entangler's phase
    static constexpr int TAB = up4(REC4 + 2 * K); // K x 4 entries x {d.re, d.im, t_up signed, t_dn signed}
    static constexpr int SURFM = TAB + 16 * K;    // 3 x 2x2 complex: the flipped surface matrices
    static constexpr int TOT = SURFM + 24;        // ROWS row totals
    static constexpr int STATES = up4(TOT + ROWS);           // (K + 2) x 64 x (re, im)
    static constexpr int PART = STATES + 128 * (K + 2);      // ROWS x ZYZ3_ROW
    static constexpr int TOTAL = PART + ROWS * ZYZ3_ROW;
    static constexpr int row_blk(int b) { return 10 + 7 * b; }
    static_assert(TOTAL == zyz3_lds_floats(K), "the host's formula");
};

// d * x / conj(d) * x with d a (re, im) register pair: two packed instructions
__device__ __forceinline__ f2 z3_cmul(f2 d, f2 x) {
    f2 o;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=&v"(o) : "v"(d), "v"(x));
    return o;
}
__device__ __forceinline__ f2 z3_cmul_conj(f2 d, f2 x) {
    f2 o;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[1,0,0]" : "=&v"(o) : "v"(d), "v"(x));
    return o;
}
// one real stage on row qubit Q: x <- x + t x[partner], t signed for this lane
template <int Q>
__device__ __forceinline__ void z3_stage(f2& x, float t) {
    constexpr int BIT = Lay<3, 0>::row_lane_bit(Q);
    const float px = xor_lane<BIT>(x.x), py = xor_lane<BIT>(x.y);
    x.x = fmaf(t, px, x.x);
    x.y = fmaf(t, py, x.y);
}
// three sums at once, stages interleaved (each fills the others' DPP wait states)
__device__ __forceinline__ void wave_sum3(float& x, float& y, float& z) {
    x += dppz<0xB1>(x);  y += dppz<0xB1>(y);  z += dppz<0xB1>(z);
    x += dppz<0x4E>(x);  y += dppz<0x4E>(y);  z += dppz<0x4E>(z);
    x += dppz<0x124>(x); y += dppz<0x124>(y); z += dppz<0x124>(z);
    x += dppz<0x128>(x); y += dppz<0x128>(y); z += dppz<0x128>(z);
    x += dpp<0x142, 0xA, 0xF>(0.f, x); y += dpp<0x142, 0xA, 0xF>(0.f, y); z += dpp<0x142, 0xA, 0xF>(0.f, z);
    x += dpp<0x143, 0xC, 0xF>(0.f, x); y += dpp<0x143, 0xC, 0xF>(0.f, y); z += dpp<0x143, 0xC, 0xF>(0.f, z);
    x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
    y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y), 63));
    z = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(z), 63));
}

// ROT: rotation string and entangler (cpf_blocks.h rot_pack / rot_with_ent); KC: blocks; MC: blocks of wave A (1 .. KC-1);
// UD...: the layer (up0, dn0, up1, dn1, ...), block b sits on pair b mod len (topology.py:15-20)
template <int ROT, int KC, int MC, int... UD>
__device__ __forceinline__ void zyz3_body(const KArgs& a) {
    using L = Lay<3, 0>;
    using PL = PairList<UD...>;
    using Lds = Zyz3Lds<KC>;
    constexpr int LL = sizeof...(UD) / 2, K = KC, M = MC;
    constexpr int NR = rot_nr(ROT), ENT = rot_ent(ROT), NBP = 2 * NR + (ENT == 0 ? 1 : 0), NS = 9;
    constexpr int NITEM = 2 * K + 3, ROWS = Lds::ROWS, NRND = (ROWS + 63) / 64, NTAB = (4 * K + 63) / 64;
    static_assert(LL > 0 && NR >= 1 && NR <= 3 && ENT <= 1, "1..3 rotation letters, CP or CZ blocks");
    static_assert(M >= 1 && M < K && NITEM <= 64, "both waves sweep at least one block; one owner lane per item");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = (int)threadIdx.x & 127, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int P = NS + NBP * K;
    if (a.P != P) return;                                    // host error: plan and specialisation disagree
    const int64_t s_idx = blockIdx.x;
    const float* init = a.init + s_idx * P;
    const uint8_t* frozen = a.frozen ? a.frozen + s_idx * P : nullptr;
    float* out_best = a.out_params + (s_idx * 2 + 1) * P;

    // ---- sweep constants of this lane
    bool lb[3]; float sg[3];
#pragma unroll
    for (int q = 0; q < 3; q++) { lb[q] = (lane >> L::row_lane_bit(q)) & 1; sg[q] = lb[q] ? 1.f : -1.f; }
    const int col = L::col_index(lane, 0), row = L::row_index(lane, 0);
    const float2 vt_ = a.target[row * 8 + col];
    const f2 vt = {vt_.x, vt_.y};
    int kofs[LL]; float m11[LL];                             // per layer position: the lane's table entry, its |11> mask
    static_for([&](auto jj) __attribute__((always_inline)) {
        constexpr int J = decltype(jj)::value, UP = PL::v[2 * J], DN = PL::v[2 * J + 1];
        kofs[J] = 4 * (2 * (lb[UP] ? 1 : 0) + (lb[DN] ? 1 : 0));
        m11[J] = (lb[UP] && lb[DN]) ? 1.f : 0.f;
    }, typename make_iseq<LL>::type{});
    f2* sp = reinterpret_cast<f2*>(smem + Lds::STATES) + lane;             // slot s at sp[64 s]
    const f2* spx[3];
#pragma unroll
    for (int q = 0; q < 3; q++) spx[q] = sp + ((lane ^ (1 << L::row_lane_bit(q))) - lane);
    float* part = smem + Lds::PART + lane;                                 // row r at part[ZYZ3_ROW r]
    const float* tab = smem + Lds::TAB;
    penalty_table_init(smem, a, tid);

    // ---- owner constants and state (wave B): lane i < 2 K owns side i (block i >> 1, side i & 1), lanes 2 K .. 2 K + 2 the qubits
    const bool is_owner = wave == 1 && lane < NITEM;
    const bool is_side = is_owner && lane < 2 * K, is_qubit = is_owner && !is_side;
    const int ib = lane >> 1, iside = lane & 1, iq = lane - 2 * K;
    int prev = -1, prev_dn = -1;                             // previous side on the item's qubit (side-0 lanes: also of the block's dn side)
    unsigned long long later = 0;                            // sides: later sides on the qubit; qubits: every side on it
    int pj[4] = {-1, -1, -1, -1};
    if (is_owner) {
        auto qubit_of = [&](int w) {                         // qubit of side w
            int r = 0;
            static_for([&](auto jj) __attribute__((always_inline)) {
                constexpr int J = decltype(jj)::value;
                if ((w >> 1) % LL == J) r = (w & 1) ? PL::v[2 * J + 1] : PL::v[2 * J];
            }, typename make_iseq<LL>::type{});
            return r;
        };
        if (is_side) {
            const int q = qubit_of(lane);
            for (int w = lane - 1; w >= 0; w--) if (qubit_of(w) == q) { prev = w; break; }
            for (int w = lane + 1; w < 2 * K; w++) if (qubit_of(w) == q) later |= 1ull << w;
            if (iside == 0) { const int qd = qubit_of(lane + 1); for (int w = lane - 1; w >= 0; w--) if (qubit_of(w) == qd) { prev_dn = w; break; } }
#pragma unroll
            for (int k = 0; k < NR; k++) pj[k] = NS + NBP * ib + 2 * k + iside;
            if (ENT == 0 && iside == 0) pj[3] = NS + NBP * ib + 2 * NR;
        } else {
            for (int w = 0; w < 2 * K; w++) if (qubit_of(w) == iq) { later |= 1ull << w; prev = w; }
            pj[0] = 3 * iq; pj[1] = 3 * iq + 1; pj[2] = 3 * iq + 2;
        }
    }
    f2 th01 = {0.f, 0.f}, th23 = {0.f, 0.f}, mm01 = {0.f, 0.f}, mm23 = {0.f, 0.f}, vv01 = {0.f, 0.f}, vv23 = {0.f, 0.f};
    f2 g01 = {0.f, 0.f}, g23 = {0.f, 0.f}, nlr01 = {0.f, 0.f}, nlr23 = {0.f, 0.f};
    {
        float t0[4], nl[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            t0[k] = 0.f; nl[k] = 0.f;
            if (pj[k] >= 0) {
                t0[k] = init[pj[k]];
                nl[k] = (frozen && frozen[pj[k]]) ? 0.f : -a.lr;
                a.out_params[(s_idx * 2) * P + pj[k]] = t0[k];
                out_best[pj[k]] = t0[k];
            }
        }
        th01 = f2{t0[0], t0[1]}; th23 = f2{t0[2], t0[3]}; nlr01 = f2{nl[0], nl[1]}; nlr23 = f2{nl[2], nl[3]};
    }
    const bool has_cp = ENT == 0 && is_side && iside == 0;
    float cs[4][2] = {{1.f, 0.f}, {1.f, 0.f}, {1.f, 0.f}, {1.f, 0.f}};
    zyz::Side sd = zyz::Side{};
    int item_m = 0, cacb = 0;                                // qubit: all flips on it; side 0: where the entangler's phase sits
    float psl = 0.f, pval = 0.f;
    // LDS addresses of this item's totals (float offsets from smem + TOT)
    int rowP = 0, row_s11 = 0, row_pzu = 0, row_pzd = 0;
    if (is_side) {
        rowP = Lds::row_blk(ib) + 3 * iside;
        row_s11 = ib == 0 ? 9 : Lds::row_blk(ib - 1) + 6;
        auto pz_row = [&](int pw, int w) {                   // Pz of the previous side pw on the qubit of side w, or of its surface state
            if (pw >= 0) return Lds::row_blk(pw >> 1) + 3 * (pw & 1) + 2;
            int q = 0;
            static_for([&](auto jj) __attribute__((always_inline)) {
                constexpr int J = decltype(jj)::value;
                if ((w >> 1) % LL == J) q = (w & 1) ? PL::v[2 * J + 1] : PL::v[2 * J];
            }, typename make_iseq<LL>::type{});
            return 3 * q + 2;
        };
        row_pzu = pz_row(prev, lane);
        row_pzd = iside == 0 ? pz_row(prev_dn, lane + 1) : 0;
    } else if (is_qubit) rowP = 3 * iq;
    if constexpr (ENT != 0) { if (tid < K) { smem[Lds::REC4 + 2 * tid] = -1.f; smem[Lds::REC4 + 2 * tid + 1] = 0.f; } }      // CZ = CP(pi), exactly
    // final left factors: the last side on each row qubit (static), read by wave B at the start of every step
    int lfin_off[3];
#pragma unroll
    for (int q = 0; q < 3; q++) {
        int lastw = -1;
        static_for([&](auto jj) __attribute__((always_inline)) {
            constexpr int J = decltype(jj)::value;
            // the last block b < K with b mod LL == J
            constexpr int BL = ((K - 1 - J) / LL) * LL + J;
            if (BL >= 0 && J < K) {
                if (PL::v[2 * J] == q && 2 * BL > lastw) lastw = 2 * BL;
                if (PL::v[2 * J + 1] == q && 2 * BL + 1 > lastw) lastw = 2 * BL + 1;
            }
        }, typename make_iseq<LL>::type{});
        lfin_off[q] = lastw < 0 ? -1 : Lds::REC1 + 4 * lastw + 2 * (lb[q] ? 1 : 0);
    }
    // surface product state: element [r_q][c_q] of S~_q
    int surf_off[3];
#pragma unroll
    for (int q = 0; q < 3; q++) surf_off[q] = Lds::SURFM + 8 * q + 4 * (lb[q] ? 1 : 0) + 2 * ((col >> (2 - q)) & 1);

    auto lds_fence = [&]() __attribute__((always_inline)) { __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); };
    auto barrier = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    float best = 0.f, first = 0.f;
    const int iters = a.T > 0 ? a.T : 1;
    f2 x = {0.f, 0.f};                                       // the live matrix of this wave
    barrier();                                               // penalty table, constant records: written by wave A's lanes
#ifdef CPF_STAMPS
    unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev = 0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev) :: "memory");
#endif
    for (int it = 0; it < iters; it++) {
        // ================= owner (wave B): Adam closes the previous step; (cos, sin); sides, parities, e-pairs, tables
        if (wave == 1) {
            if (it > 0) {
                const float2 bc = const_load_f2(a.bias, it - 1);
                const f2 ibc = f2{__builtin_amdgcn_rcpf(bc.x), __builtin_amdgcn_rcpf(bc.y)};
                if (has_cp && a.use_reg) g23.y += psl;
                // optax scale_by_adam (optimization.py:14-25): the same operations in the same order as the scalar form, per half
                const f2 m0 = pk_fma(splat(a.b1), mm01, splat(a.omb1) * g01), m1 = pk_fma(splat(a.b1), mm23, splat(a.omb1) * g23);
                const f2 v0 = pk_fma(splat(a.b2), vv01, splat(a.omb2) * (g01 * g01)), v1 = pk_fma(splat(a.b2), vv23, splat(a.omb2) * (g23 * g23));
                mm01 = m0; mm23 = m1; vv01 = v0; vv23 = v1;
                const f2 h0 = v0 * splat(ibc.y), h1 = v1 * splat(ibc.y);
                const f2 d0 = f2{__builtin_amdgcn_sqrtf(h0.x), __builtin_amdgcn_sqrtf(h0.y)} + splat(a.eps);
                const f2 d1 = f2{__builtin_amdgcn_sqrtf(h1.x), __builtin_amdgcn_sqrtf(h1.y)} + splat(a.eps);
                const f2 u0 = (m0 * splat(ibc.x)) * f2{__builtin_amdgcn_rcpf(d0.x), __builtin_amdgcn_rcpf(d0.y)};
                const f2 u1 = (m1 * splat(ibc.x)) * f2{__builtin_amdgcn_rcpf(d1.x), __builtin_amdgcn_rcpf(d1.y)};
                th01 = pk_fma(nlr01, u0, th01);
                th23 = pk_fma(nlr23, u1, th23);
            }
            {
                f2 s01, c01, s23, c23;
                fast_sincos2(th01 * splat(0.5f), s01, c01);
                fast_sincos2(th23 * f2{0.5f, 1.f}, s23, c23);                 // slot 3 is a CP angle: the full angle
                cs[0][0] = c01.x; cs[0][1] = s01.x; cs[1][0] = c01.y; cs[1][1] = s01.y;
                cs[2][0] = c23.x; cs[2][1] = s23.x; cs[3][0] = c23.y; cs[3][1] = s23.y;
            }
            if (has_cp && a.use_reg) {
                float sl;
                pval = penalty_lds(smem, a.pen_kind, th23.y, sl);
                psl = a.r * sl;
            }
            auto rrec = [&](int idx, float& c, float& sn, int& kind) __attribute__((always_inline)) {
                const int k = (idx - 1) >> 1;
                c = cs[k][0]; sn = cs[k][1]; kind = 0;
            };
            bool flip = false;
            if (is_side) {
                sd = zyz::side_build<ROT & 0xFFFF>(rrec, NR, iside);
                flip = sd.f != 0;
                *reinterpret_cast<float4*>(smem + Lds::REC1 + 4 * lane) = make_float4(sd.l0.re, sd.l0.im, sd.l1.re, sd.l1.im);
                if (has_cp) *reinterpret_cast<float2*>(smem + Lds::REC4 + 2 * ib) = make_float2(cs[3][0], cs[3][1]);
            }
            const unsigned long long fbits = __builtin_amdgcn_ballot_w64(flip);
            const int par = __popcll(fbits & later) & 1;
            lds_fence();
            int ec = 0;
            if (is_side) {
                sd.m = par;
                float4 pl = make_float4(1.f, 0.f, 1.f, 0.f);
                if (prev >= 0) pl = *reinterpret_cast<const float4*>(smem + Lds::REC1 + 4 * prev);
                const zyz::EPair e = zyz::side_epair(sd, prev >= 0, zyz::c2(pl.x, pl.y), zyz::c2(pl.z, pl.w));
                *reinterpret_cast<float4*>(smem + Lds::REC2 + 4 * lane) = make_float4(e.e0.re, e.e0.im, e.e1.re, e.e1.im);
                *reinterpret_cast<float4*>(smem + Lds::REC3 + 4 * lane) = make_float4(sd.t, sd.sc, __int_as_float(e.c), 0.f);
                ec = e.c;
            } else if (is_qubit) {
                item_m = par;
                float sm[8];
                zyz::surface_matrix(sm, cs[0][0], cs[0][1], cs[1][0], cs[1][1], cs[2][0], cs[2][1], item_m);
                *reinterpret_cast<float4*>(smem + Lds::SURFM + 8 * iq) = make_float4(sm[0], sm[1], sm[2], sm[3]);
                *reinterpret_cast<float4*>(smem + Lds::SURFM + 8 * iq + 4) = make_float4(sm[4], sm[5], sm[6], sm[7]);
            }
            // side 0 keeps where the entangler's phase sits: its own bit and the dn side's (lane + 1); every lane takes part in the move
            cacb = ec | (__builtin_amdgcn_update_dpp(0, ec, 0xB1, 0xF, 0xF, true) << 1);       // quad_perm [1,0,3,2]
            lds_fence();
            // one lane per table entry j = 4 b + k, k = 2 a_up + a_dn
#pragma unroll
            for (int r = 0; r < NTAB; r++) {
                const int j = lane + 64 * r;
                if (j < 4 * K) {
                    const int b = j >> 2, k = j & 3;
                    const float4 u2 = *reinterpret_cast<const float4*>(smem + Lds::REC2 + 8 * b), d2 = *reinterpret_cast<const float4*>(smem + Lds::REC2 + 8 * b + 4);
                    const float4 u3 = *reinterpret_cast<const float4*>(smem + Lds::REC3 + 8 * b), d3 = *reinterpret_cast<const float4*>(smem + Lds::REC3 + 8 * b + 4);
                    const float2 lam = *reinterpret_cast<const float2*>(smem + Lds::REC4 + 2 * b);
                    zyz::EPair eu, ed;
                    eu.e0 = zyz::c2(u2.x, u2.y); eu.e1 = zyz::c2(u2.z, u2.w); eu.c = __float_as_int(u3.z);
                    ed.e0 = zyz::c2(d2.x, d2.y); ed.e1 = zyz::c2(d2.z, d2.w); ed.c = __float_as_int(d3.z);
                    float o[4];
                    zyz::table_entry(o, k >> 1, k & 1, eu, ed, u3.x, u3.y, d3.x, d3.y, lam.x, lam.y);
                    *reinterpret_cast<float4*>(smem + Lds::TAB + 4 * j) = make_float4(o[0], o[1], o[2], o[3]);
                }
            }
        }
        CPF_STAMP(0)                                  // owner / build
        barrier();
        CPF_STAMP(1)

        // ================= forward halves.  The table entries of the wave's own blocks are read once and serve both sweeps.
        constexpr int NMINE = M > K - M ? M : K - M;
        float4 ent[NMINE];
        float pen_sum = 0.f;
        if (wave == 0) {
            static_for([&](auto bb) __attribute__((always_inline)) {
                constexpr int B = decltype(bb)::value;
                ent[B] = *reinterpret_cast<const float4*>(tab + 16 * B + kofs[B % LL]);
            }, typename make_iseq<M>::type{});
            {
                const float2 f0 = *reinterpret_cast<const float2*>(smem + surf_off[0]), f1 = *reinterpret_cast<const float2*>(smem + surf_off[1]),
                             f2_ = *reinterpret_cast<const float2*>(smem + surf_off[2]);
                x = z3_cmul(f2{f2_.x, f2_.y}, z3_cmul(f2{f1.x, f1.y}, f2{f0.x, f0.y}));
            }
            sp[0] = x;
            static_for([&](auto bb) __attribute__((always_inline)) {
                constexpr int B = decltype(bb)::value, J = B % LL;
                x = z3_cmul(f2{ent[B].x, ent[B].y}, x);
                z3_stage<PL::v[2 * J]>(x, ent[B].z);
                z3_stage<PL::v[2 * J + 1]>(x, ent[B].w);
                sp[64 * (B + 1)] = x;
            }, typename make_iseq<M>::type{});
        } else {
            static_for([&](auto bb) __attribute__((always_inline)) {
                constexpr int B = M + decltype(bb)::value;
                ent[B - M] = *reinterpret_cast<const float4*>(tab + 16 * B + kofs[B % LL]);
            }, typename make_iseq<K - M>::type{});
            {   // Vd = conj(dfin) V: the left factors the last blocks leave behind meet the target
                f2 dfin = {1.f, 0.f};
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    if (lfin_off[q] >= 0) {
                        const float2 v = *reinterpret_cast<const float2*>(smem + lfin_off[q]);
                        dfin = z3_cmul(f2{v.x, v.y}, dfin);
                    }
                }
                x = z3_cmul_conj(dfin, vt);
            }
            sp[64 * (K + 1)] = x;
            static_for([&](auto bb) __attribute__((always_inline)) {
                constexpr int B = K - 1 - decltype(bb)::value, J = B % LL;
                z3_stage<PL::v[2 * J + 1]>(x, -ent[B - M].w);
                z3_stage<PL::v[2 * J]>(x, -ent[B - M].z);
                x = z3_cmul_conj(f2{ent[B - M].x, ent[B - M].y}, x);
                sp[64 * (B + 1)] = x;                                            // W(s_{B-1})
            }, typename make_iseq<K - M>::type{});
        }
        CPF_STAMP(2)                                  // forward half
        barrier();
        CPF_STAMP(3)

        // ================= loss (both waves, identical arithmetic): T = sum conj(W) psi at the cut
        const f2 other = wave == 0 ? sp[64 * (M + 1)] : sp[64 * M];
        const f2 psi_c = wave == 0 ? x : other, w_c = wave == 0 ? other : x;
        const f2 tt = z3_cmul_conj(w_c, psi_c);
        float tr = tt.x, ti = tt.y;
        if (wave == 1 && a.use_reg) { float pv = pval; wave_sum3(tr, ti, pv); pen_sum = pv; }
        else wave_sum2(tr, ti);
        bool improved = false;
        if (wave == 1) {
            const float reg = a.use_reg ? a.r * pen_sum : 0.f;
            const float ab = sqrtf(fmaf(tr, tr, ti * ti));
            const float loss = 1.f - ab * ab * a.inv_dim2;
            const float regloss = loss + reg;
            if (it == 0) { first = regloss; best = regloss; }
            else if (regloss < best) { best = regloss; improved = true; }
            if (improved || a.hist_params) {
                const float t0[4] = {th01.x, th01.y, th23.x, th23.y};
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (pj[k] < 0) continue;
                    if (improved) out_best[pj[k]] = t0[k];
                    if (a.hist_params) a.hist_params[(s_idx * a.T + it) * P + pj[k]] = t0[k];
                }
            }
            if (a.hist_loss && lane == 0) a.hist_loss[s_idx * a.T + it] = regloss;
        }
        if (a.T == 0) break;
        if (it + 1 == iters) break;                   // the last step's gradient would move the angles past the last evaluated iterate
        CPF_STAMP(4)                                  // loss

        // ================= gradient halves: one live matrix, one read back; every lane's shares go to its column of the rows
        const f2 fcot = {-a.inv_dim2 * tr, -a.inv_dim2 * ti};
        if (wave == 0) {
            f2 w = z3_cmul(fcot, w_c);                                     // the cotangent at the cut
            // slice s_b: psi at the lane (`own`) and at the partner lanes of the block's qubits
            static_for([&](auto bb) __attribute__((always_inline)) {
                constexpr int B = M - 1 - decltype(bb)::value, J = B % LL, JN = (B + 1) % LL;
                constexpr int UP = PL::v[2 * J], DN = PL::v[2 * J + 1], R0 = Lds::row_blk(B);
                const f2 own = B == M - 1 ? x : sp[64 * (B + 1)];
                const f2 pu = spx[UP][64 * (B + 1)], pd = spx[DN][64 * (B + 1)];
                const f2 zu = w * swp(pu), yu = w * pu, zd = w * swp(pd), yd = w * pd;
                const float u = fmaf(w.x, own.y, -w.y * own.x);
                part[ZYZ3_ROW * (R0 + 0)] = zu.x - zu.y;
                part[ZYZ3_ROW * (R0 + 1)] = sg[UP] * (yu.x + yu.y);
                part[ZYZ3_ROW * (R0 + 2)] = -sg[UP] * u;
                part[ZYZ3_ROW * (R0 + 3)] = zd.x - zd.y;
                part[ZYZ3_ROW * (R0 + 4)] = sg[DN] * (yd.x + yd.y);
                part[ZYZ3_ROW * (R0 + 5)] = -sg[DN] * u;
                if constexpr (B + 1 < K) part[ZYZ3_ROW * (R0 + 6)] = m11[JN] * u;
                z3_stage<DN>(w, -ent[B].w);
                z3_stage<UP>(w, -ent[B].z);
                w = z3_cmul_conj(f2{ent[B].x, ent[B].y}, w);
            }, typename make_iseq<M>::type{});
            {   // the surface slice s_-1: all three qubits, and the |11> sum of block 0
                const f2 own = sp[0];
                const float u = fmaf(w.x, own.y, -w.y * own.x);
                static_for([&](auto qq) __attribute__((always_inline)) {
                    constexpr int Q = decltype(qq)::value;
                    const f2 pq = spx[Q][0];
                    const f2 zq = w * swp(pq), yq = w * pq;
                    part[ZYZ3_ROW * (3 * Q + 0)] = zq.x - zq.y;
                    part[ZYZ3_ROW * (3 * Q + 1)] = sg[Q] * (yq.x + yq.y);
                    part[ZYZ3_ROW * (3 * Q + 2)] = -sg[Q] * u;
                }, typename make_iseq<3>::type{});
                part[ZYZ3_ROW * 9] = m11[0] * u;
            }
        } else {
            f2 pp = z3_cmul_conj(fcot, psi_c);                             // conj(f) psi: the scalar moves onto psi once
            static_for([&](auto bb) __attribute__((always_inline)) {
                constexpr int B = M + decltype(bb)::value, J = B % LL, JN = (B + 1) % LL;
                constexpr int UP = PL::v[2 * J], DN = PL::v[2 * J + 1], R0 = Lds::row_blk(B);
                pp = z3_cmul(f2{ent[B - M].x, ent[B - M].y}, pp);
                z3_stage<UP>(pp, ent[B - M].z);
                z3_stage<DN>(pp, ent[B - M].w);
                const f2 own = sp[64 * (B + 2)];                                  // W(s_B)
                const f2 wu = spx[UP][64 * (B + 2)], wd = spx[DN][64 * (B + 2)];
                const f2 zu = wu * swp(pp), yu = wu * pp, zd = wd * swp(pp), yd = wd * pp;
                const float u = fmaf(own.x, pp.y, -own.y * pp.x);
                part[ZYZ3_ROW * (R0 + 0)] = zu.x - zu.y;
                part[ZYZ3_ROW * (R0 + 1)] = -sg[UP] * (yu.x + yu.y);
                part[ZYZ3_ROW * (R0 + 2)] = -sg[UP] * u;
                part[ZYZ3_ROW * (R0 + 3)] = zd.x - zd.y;
                part[ZYZ3_ROW * (R0 + 4)] = -sg[DN] * (yd.x + yd.y);
                part[ZYZ3_ROW * (R0 + 5)] = -sg[DN] * u;
                if constexpr (B + 1 < K) part[ZYZ3_ROW * (R0 + 6)] = m11[JN] * u;
            }, typename make_iseq<K - M>::type{});
        }
        CPF_STAMP(5)                                  // gradient half
        barrier();
        CPF_STAMP(6)

        // ================= row sums and chain rule (wave B)
        if (wave == 1) {
#pragma unroll
            for (int r = 0; r < NRND; r++) {
                const int rowi = lane + 64 * r;
                const int rc = rowi < ROWS ? rowi : ROWS - 1;
                const float4* src = reinterpret_cast<const float4*>(smem + Lds::PART + ZYZ3_ROW * rc);
                f2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const float4 v = src[k];
                    a0 += f2{v.x, v.y}; a1 += f2{v.z, v.w};
                }
                if (rowi < ROWS) smem[Lds::TOT + rowi] = (a0.x + a0.y) + (a1.x + a1.y);
            }
            lds_fence();
            const float* tot = smem + Lds::TOT;
            if (is_side) {
                const float px = tot[rowP], py = tot[rowP + 1], pz = tot[rowP + 2];
                auto rrec = [&](int idx, float& c, float& sn, int& kind) __attribute__((always_inline)) {
                    const int k = (idx - 1) >> 1;
                    c = cs[k][0]; sn = cs[k][1]; kind = 0;
                };
                float gp[4] = {0.f, 0.f, 0.f, 0.f};
                zyz::chain_side<ROT & 0xFFFF>(gp, px, py, pz, sd, rrec, NR, iside);
                if (has_cp) gp[3] = zyz::chain_cp(tot[row_s11], tot[row_pzu], tot[row_pzd], cacb, 1.f);
                g01 = f2{gp[0], gp[1]}; g23 = f2{gp[2], gp[3]};
            } else if (is_qubit) {
                float gs[3];
                zyz::chain_surface(gs, tot[rowP], tot[rowP + 1], tot[rowP + 2], item_m, cs[1][0], cs[1][1], cs[2][0], cs[2][1]);
                g01 = f2{gs[0], gs[1]}; g23 = f2{gs[2], 0.f};
            }
        }
        CPF_STAMP(7)
    }
#ifdef CPF_STAMPS
    if (s_idx == 0 && lane == 0 && a.hist_loss) for (int k = 0; k < 8; k++) a.hist_loss[8 * wave + k] = (float)stamp_sum[k];
#endif

    // ---- epilogue (optimization.py:365-370, cp_utils.py:144-164): reg recompute, loss = regloss - reg, argmin, CZ count, entry-loss flag
    if (wave == 1) {
        const int bi = (best < first) ? 1 : 0;
        float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f;
        if (has_cp) {
            const int j = pj[3];
            const float t0 = a.out_params[(s_idx * 2) * P + j], t1 = out_best[j];
            float sl;
            if (a.use_reg) { acc0 = penalty_lds(smem, a.pen_kind, t0, sl); acc1 = penalty_lds(smem, a.pen_kind, t1, sl); }
            acc2 = cz_value(bi ? t1 : t0, a.thr_cp);
        }
        wave_sum3(acc0, acc1, acc2);
        if (lane == 0) {
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

template <int ROT, int KC, int MC, int... UD>
__global__ __launch_bounds__(128, 1) void k_cp_zyz3(const KArgs a) {
    zyz3_body<ROT, KC, MC, UD...>(a);
}

}  // namespace cpf
