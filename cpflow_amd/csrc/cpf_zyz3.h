// cpflow_amd: the two-ended ZYZ-form kernel for 3 qubits - BASELINE configs 1-3 (round 5).
//
// What it replaces, per optimiser step and sample (cpflow/main.py:69-82,106-146 build_unitary, matrix_utils.py:35-42 cost_HST,
// penalty.py:44-71, optimization.py:14-25,61-75,340): the work of cpf_split.h, in the algebra of cpf_zyz_math.h.
//
// One amplitude of the 8 x 8 unitary per lane (Lay<3, 0>: every row qubit on a DPP-reachable lane bit), two waves per sample:
//   wave A   psi(s_b), b = -1 .. M-1: the (flipped) product state of the surface gates, then blocks 0 .. M-1 applied;
//            lanes 0 .. 2 OWN the surface angles of the three qubits
//   wave B   W(s_b), b = K-1 .. M-1: Vd = conj(dfin) V, then blocks K-1 .. M un-applied;
//            lane w = 2 b + s OWNS the angles of side s of block b (side 0: the CP angle too)
// with s_b the slice right after the two real stages of block b (cpf_zyz_math.h: what the sweeps hold there is the true state
// up to pending flips and the left diagonals not yet merged).  A block is ONE complex multiply by the lane's merged phase and two
// stages x += (+-t) x[partner] (two v_fmac_f32_dpp each): 6 vector instructions where the gate-by-gate sweep of cpf_split.h spends
// 14 (four x / y rotations at 2 DPP + 1 packed, one merged diagonal).  Every state of the forward halves stays in LDS (K + 2 slots
// of 512 B) and in the registers of the wave that made it, so the gradient halves move ONE matrix: wave A un-applies blocks
// M-1 .. 0 from the cotangent w = f W with psi(s_b) at hand (the lane's own value kept from the forward half, the two partner
// lanes' read back in one batch before the sweep); wave B applies blocks M .. K-1 to psi' = conj(f) psi with W(s_b) at hand.
// At a slice the 7 generator sums of cpf_zyz_math.h come from three products per lane:
//     conj(w) psi            -> Pz of the block's two qubits (signed by the lane's bits), the |11> sum of the NEXT block's pair
//     conj(w) psi[partner]   -> Px (imaginary part), Py (real part, signed), once per qubit of the block
// (tools/zyz3_model.py is the executable specification, checked against tools/zyz_model.py and central differences).
//
// The sums do not cross the lanes in the sweeps at all: every lane writes its 7 shares to LDS (one 272-byte row per sum: 64 lanes
// + padding, conflict-free both ways), and each wave then adds ITS OWN rows with one lane per row - 16 ds_read_b128 and 33 packed
// adds per 64 sums where the DPP reduce-scatter of cpf_blocks.h costs 28 issue slots per block inside the sweep.  With one state
// per block instead of one per inner gate the rows fit next to the states (config 2: 25 KB + 7 KB per sample, four samples per CU).
//
// A step is a dependent chain - angles -> tables -> sweeps -> sums -> gradients -> angles - so what is kept short is that chain:
//   * the side lanes build their block's table themselves: two sides of a block are neighbouring lanes, so a quad_perm exchange of
//     the e-pairs replaces a trip through LDS, and each lane writes two of the four entries a sweep lane reads with one
//     ds_read_b128: (d.re, d.im, signed t_up, signed t_dn).  The common phase of a block's entries is left in (no loss sees it);
//   * the flip bits of all sides are ONE ballot (all sides sit in wave B), pending parities a popcount against a per-lane mask;
//   * T = sum conj(W) psi over the 64 lanes is two v_mfma_f32_16x16x4_f32 against a register of ones (the matrix core as an
//     adder tree: ~100 cycles where six DPP stages and two v_readlane take ~400), both waves, identical arithmetic;
//   * the penalty sum is K broadcast LDS reads in each wave (no cross-lane instruction);
//   * the surface gates belong to wave A, whose owner phase (chain rule, Adam, sin / cos, the two variants S and F S of each
//     surface matrix) runs while wave B does the same for the blocks; wave B only publishes the three flip parities.
// Both waves form the loss with the same instructions and track the best iterate for the angles they own.
//
// Requires: n = 3, 1..3 rotation letters (any), CP or CZ blocks, loss kinds with a linear cotangent (HST, state preparation),
// 2 K <= 64 sides, K and the layer compile-time constants (instances: ahead of time for configs 2 / 3, cpf_jit.h otherwise).
#pragma once
#include "cpf_layered.h"
#include "cpf_zyz_math.h"

namespace cpf {

constexpr int ZYZ3_ROW = 68;                     // floats per row of shares: 64 lanes + 4 (16 lanes of one ds_read_b128 pass hit 16 different bank groups)

// floats of LDS per sample of an instance with K blocks (Zyz3Lds<K>::TOTAL)
__host__ __device__ constexpr int zyz3_lds_floats(int K) {
    const int rows = 10 + 7 * K;
    const int pv = PEN_FLOATS + 8 * K + 4, tab = pv + ((K + 3) & ~3), surfm = tab + 16 * K, tot = surfm + 24, states = (tot + rows + 3) & ~3;
    return states + 128 * (K + 2) + rows * ZYZ3_ROW;
}
__host__ __device__ constexpr bool zyz3_supported(int K, int n_rot) { return K >= 2 && 2 * K + 3 <= 64 && n_rot >= 1 && n_rot <= 3; }

template <int K>
struct Zyz3Lds {                                  // float offsets, per sample; [0, 28) = penalty table
    static constexpr int up4(int x) { return (x + 3) & ~3; }
    static constexpr int ROWS = 10 + 7 * K;       // 9 surface sums, S11 of block 0, then 7 per block (Px Py Pz up, Px Py Pz dn, S11 of the next block)
    static constexpr int REC1 = PEN_FLOATS;       // (2 K + 1) x {l0, l1}: left factors of a side; the last entry is the identity ("no previous side")
    static constexpr int PV = REC1 + 8 * K + 4;   // K penalty values (padded to a multiple of 4)
    static constexpr int TAB = PV + up4(K);       // K x 4 entries x {d.re, d.im, t_up signed, t_dn signed}
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
// One real stage on row qubit Q: x <- x + t x[partner] (NEG: - t), t signed for this lane: two v_fmac_f32_dpp whose DPP operand is
// the destination itself (the compiler does not fold a DPP read into an accumulate onto the register it reads: it emitted
// v_mov_b32_dpp + v_fma_f32, six issue slots per stage instead of four).  quad_perm / row_ror move within a row of 16 lanes, which
// one pass of the instruction reads before it writes.  s_nop 1: x may come straight from a vector write (two wait states before a DPP read).
#define CPF_Z3_STAGE(CTRL)                                                                                                   \
    if constexpr (NEG) asm("s_nop 1\n\tv_fmac_f32_dpp %0, %0, -%2 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"      \
                           "v_fmac_f32_dpp %1, %1, -%2 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x.x), "+v"(x.y) : "v"(t)); \
    else asm("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %2 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                     \
             "v_fmac_f32_dpp %1, %1, %2 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x.x), "+v"(x.y) : "v"(t));
template <int Q, bool NEG = false>
__device__ __forceinline__ void z3_stage(f2& x, float t) {
    static_assert(Q >= 0 && Q < 3, "row qubits 0 .. 2 sit on lane bits 0, 1, 3");
    if constexpr (Q == 0) { CPF_Z3_STAGE("quad_perm:[1,0,3,2]") }
    else if constexpr (Q == 1) { CPF_Z3_STAGE("quad_perm:[2,3,0,1]") }
    else { CPF_Z3_STAGE("row_ror:8") }
}
#undef CPF_Z3_STAGE
// value of lane ^ 1 (quad_perm [1,0,3,2]); every lane takes part
__device__ __forceinline__ float z3_nb(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true)); }
__device__ __forceinline__ int z3_nb(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true); }

// Sum of a complex number over the 64 lanes, the same bits in every lane: the matrix core as an adder tree.
// v_mfma_f32_16x16x4_f32 with B = 1: D[i][j] = sum_k A[lane 16 k + i]; lane (g, j) receives rows i = 4 g .. 4 g + 3 in its four
// result registers.  Lanes are paired first (even lanes carry the real parts, odd lanes the imaginary parts), so one instruction
// serves both components; a second one adds the four lane groups.
#ifndef CPF_Z3_MFMA
#define CPF_Z3_MFMA 1            // A/B: 0 = six DPP stages + v_readlane (cpf_device.h wave_sum2)
#endif
#ifndef CPF_Z3_PRIO
#define CPF_Z3_PRIO 1            // issue priority of wave B in launches whose workgroups are all resident (zyz3_body; 0 = never raised)
#endif
#ifndef CPF_Z3_STAGGER
#define CPF_Z3_STAGGER 0         // units of 512 clocks by which every other sample starts late (zyz3_body)
#endif
typedef float z3_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void z3_allsum(float& re, float& im, bool odd) {
#if CPF_Z3_MFMA
    const float a = re + z3_nb(re), b = im + z3_nb(im);
    const z3_f4 zero = {0.f, 0.f, 0.f, 0.f};
    const z3_f4 d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(odd ? b : a, 1.0f, zero, 0, 0, 0);
    const float r1 = d1[0] + d1[2], i1 = d1[1] + d1[3];
    const z3_f4 d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(odd ? i1 : r1, 1.0f, zero, 0, 0, 0);
    re = d2[0]; im = d2[1];
#else
    wave_sum2(re, im);
#endif
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

// per-half sums of three values (two samples share the owner wave, lanes 0 .. 31 and 32 .. 63): lane l receives the sums over its half
__device__ __forceinline__ void half_sum3(float& x, float& y, float& z, bool upper) {
    x += dppz<0xB1>(x);  y += dppz<0xB1>(y);  z += dppz<0xB1>(z);
    x += dppz<0x4E>(x);  y += dppz<0x4E>(y);  z += dppz<0x4E>(z);
    x += dppz<0x124>(x); y += dppz<0x124>(y); z += dppz<0x124>(z);
    x += dppz<0x128>(x); y += dppz<0x128>(y); z += dppz<0x128>(z);
    x += dpp<0x142, 0xA, 0xF>(0.f, x); y += dpp<0x142, 0xA, 0xF>(0.f, y); z += dpp<0x142, 0xA, 0xF>(0.f, z);       // lanes 31 / 63: the halves' totals
    const float x0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 31)), x1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
    const float y0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y), 31)), y1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y), 63));
    const float z0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(z), 31)), z1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(z), 63));
    x = upper ? x1 : x0; y = upper ? y1 : y0; z = upper ? z1 : z0;
}

// ROT: rotation string and entangler (cpf_blocks.h rot_pack / rot_with_ent); KC: blocks; MC: blocks of wave A (1 .. KC-1);
// PAIRS: samples per workgroup (1 or 2); UD...: the layer (up0, dn0, up1, dn1, ...), block b sits on pair b mod len (topology.py:15-20).
// PAIRS = 2 (2 K + 3 <= 32 items): four waves, two samples, each with its own copy of the LDS map - and ONE owner: wave B of the
// first sample owns the angles of both (lanes 0 .. 31 the first sample's items, 32 .. 63 the second's).  The owner phases are
// instruction streams of ~400 vector instructions per step whatever the number of busy lanes, and a step is bound by the instructions
// issued: sharing them between two samples takes a fifth off the step.  The second sample's wave B only sweeps.
template <int ROT, int KC, int MC, int PAIRS, int... UD>
__device__ __forceinline__ void zyz3_body(const KArgs& a) {
    using L = Lay<3, 0>;
    using PL = PairList<UD...>;
    using Lds = Zyz3Lds<KC>;
    constexpr int LL = sizeof...(UD) / 2, K = KC, M = MC;
    constexpr int NR = rot_nr(ROT), ENT = rot_ent(ROT), NBP = 2 * NR + (ENT == 0 ? 1 : 0), NS = 9;
    constexpr int NITEM = 2 * K + 3;
    constexpr int ROWS = Lds::ROWS, RA = Lds::row_blk(M);     // rows [0, RA) are written and added by wave A, [RA, ROWS) by wave B
    constexpr int NRA = (RA + 63) / 64, NRB = (ROWS - RA + 63) / 64;
    static_assert(LL > 0 && NR >= 1 && NR <= 3 && ENT <= 1, "1..3 rotation letters, CP or CZ blocks");
    static_assert(M >= 1 && M < K && NITEM <= 64, "both waves sweep at least one block; one owner lane per item");
    static_assert(PAIRS == 1 || (PAIRS == 2 && NITEM <= 32), "two samples per workgroup: the items of each in half a wave");
    extern __shared__ __attribute__((aligned(16))) float smem_wg[];
    const int tid = (int)threadIdx.x & 127, lane = tid & 63;     // thread of its wave pair
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pair = PAIRS > 1 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 7) : 0;
    float* smem = smem_wg + pair * Lds::TOTAL;                   // every sample has its own copy of the map
    const int P = NS + NBP * K;
    if (a.P != P) return;                                    // host error: plan and specialisation disagree
    // an odd batch: the last workgroup's second pair runs a copy of the last sample (it must meet the same barriers) and writes nothing
    const int64_t s_raw = (int64_t)blockIdx.x * PAIRS + pair;
    const int64_t s_idx = (PAIRS == 1 || s_raw < a.nsamples) ? s_raw : (int64_t)a.nsamples - 1;

    // ---- sweep constants of this lane
    bool lb[3]; float sg[3];
#pragma unroll
    for (int q = 0; q < 3; q++) { lb[q] = (lane >> L::row_lane_bit(q)) & 1; sg[q] = lb[q] ? 1.f : -1.f; }
    const bool odd = lane & 1;
    const int col = L::col_index(lane, 0), row = L::row_index(lane, 0);
    int kofs[LL]; float m11[LL]; f2 sgp[LL];                 // per layer position: the lane's table entry, its |11> mask, (sg[up], sg[dn])
    static_for([&](auto jj) __attribute__((always_inline)) {
        constexpr int J = decltype(jj)::value, UP = PL::v[2 * J], DN = PL::v[2 * J + 1];
        kofs[J] = 4 * (2 * (lb[UP] ? 1 : 0) + (lb[DN] ? 1 : 0));
        m11[J] = (lb[UP] && lb[DN]) ? 1.f : 0.f;
        sgp[J] = f2{sg[UP], sg[DN]};
    }, typename make_iseq<LL>::type{});
    f2* sp = reinterpret_cast<f2*>(smem + Lds::STATES) + lane;             // slot s at sp[64 s]
    const float* spx[3];                                                   // the same slot seen from the partner lane on qubit q (re at [0], im at [1])
#pragma unroll
    for (int q = 0; q < 3; q++) spx[q] = reinterpret_cast<const float*>(sp + ((lane ^ (1 << L::row_lane_bit(q))) - lane));
    float* part = smem + Lds::PART + lane;                                 // row r at part[ZYZ3_ROW r]
    const float* tab = smem + Lds::TAB;
    penalty_table_init(smem, a, tid);
    if (tid == 0) *reinterpret_cast<float4*>(smem + Lds::REC1 + 8 * K) = make_float4(1.f, 0.f, 1.f, 0.f);
    if (tid < Lds::up4(K)) smem[Lds::PV + tid] = 0.f;

    auto lds_fence = [&]() __attribute__((always_inline)) { __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); };
    auto barrier = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    // one wave adds rows [r_lo, r_hi) of the shares (its own lanes wrote them: in order, no barrier), one lane per row
    auto row_sums = [&](int r_lo, int r_hi, int nrnd) __attribute__((always_inline)) {
        lds_fence();
        for (int r = 0; r < nrnd; r++) {
            const int rowi = r_lo + lane + 64 * r;
            const int rc = rowi < r_hi ? rowi : r_hi - 1;
            const float4* src = reinterpret_cast<const float4*>(smem + Lds::PART + ZYZ3_ROW * rc);
            f2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const float4 v = src[k];
                a0 += f2{v.x, v.y}; a1 += f2{v.z, v.w};
            }
            if (rowi < r_hi) smem[Lds::TOT + rowi] = (a0.x + a0.y) + (a1.x + a1.y);
        }
    };
    const int iters = a.T > 0 ? a.T : 1;
#ifdef CPF_STAMPS
    unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev = 0;
#endif

    if (wave == 0) {
        // =====================================================================================================================
        // wave A: blocks 0 .. M-1.  No angles: it needs T only for the cotangent.
        // =====================================================================================================================
        int surf_off[3];                                     // element [r_q][c_q] of the (flipped) surface matrix of qubit q
#pragma unroll
        for (int q = 0; q < 3; q++) surf_off[q] = Lds::SURFM + 8 * q + 4 * (lb[q] ? 1 : 0) + 2 * ((col >> (2 - q)) & 1);
        barrier();                                           // constant records of the set-up
#ifdef CPF_STAMPS
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev) :: "memory");
#endif
        for (int it = 0; it < iters; it++) {
            CPF_STAMP(0)
            barrier();                                       // tables and surface matrices of this step
            CPF_STAMP(1)
            float4 ent[M];
            f2 own[M + 1];
            float2 fs[3];
#pragma unroll
            for (int q = 0; q < 3; q++) fs[q] = *reinterpret_cast<const float2*>(smem + surf_off[q]);
            static_for([&](auto bb) __attribute__((always_inline)) {
                constexpr int B = decltype(bb)::value;
                ent[B] = *reinterpret_cast<const float4*>(tab + 16 * B + kofs[B % LL]);
            }, typename make_iseq<M>::type{});
            f2 x = z3_cmul(f2{fs[2].x, fs[2].y}, z3_cmul(f2{fs[1].x, fs[1].y}, f2{fs[0].x, fs[0].y}));
            sp[0] = x; own[0] = x;
            static_for([&](auto bb) __attribute__((always_inline)) {
                constexpr int B = decltype(bb)::value, J = B % LL;
                x = z3_cmul(f2{ent[B].x, ent[B].y}, x);
                z3_stage<PL::v[2 * J]>(x, ent[B].z);
                z3_stage<PL::v[2 * J + 1]>(x, ent[B].w);
                sp[64 * (B + 1)] = x; own[B + 1] = x;
            }, typename make_iseq<M>::type{});
            CPF_STAMP(2)                                  // forward half
            barrier();
            CPF_STAMP(3)
            // the partner lanes' states of every slice, one batch under the reduction: pre[b] = (re at the up partner, re at the dn
            // partner), pim[b] likewise, so that the two qubits of a block share packed instructions
            const f2 w_c = sp[64 * (M + 1)];
            f2 pre[M], pim[M], ps[3];
            static_for([&](auto bb) __attribute__((always_inline)) {
                constexpr int B = decltype(bb)::value, J = B % LL;
                const float* qu = spx[PL::v[2 * J]] + 128 * (B + 1);
                const float* qd = spx[PL::v[2 * J + 1]] + 128 * (B + 1);
                pre[B] = f2{qu[0], qd[0]}; pim[B] = f2{qu[1], qd[1]};
            }, typename make_iseq<M>::type{});
#pragma unroll
            for (int q = 0; q < 3; q++) ps[q] = f2{spx[q][0], spx[q][1]};
            const f2 tt = z3_cmul_conj(w_c, x);              // T = sum conj(W) psi at the cut
            float tr = tt.x, ti = tt.y;
            z3_allsum(tr, ti, odd);
            if (a.T == 0) break;
            if (it + 1 == iters) break;
            CPF_STAMP(4)                                  // loss
            f2 w = z3_cmul(f2{-a.inv_dim2 * tr, -a.inv_dim2 * ti}, w_c);       // the cotangent at the cut: f W, f = -T / N^2
            static_for([&](auto bb) __attribute__((always_inline)) {
                constexpr int B = M - 1 - decltype(bb)::value, J = B % LL, JN = (B + 1) % LL;
                constexpr int UP = PL::v[2 * J], DN = PL::v[2 * J + 1], R0 = Lds::row_blk(B);
                const f2 o = own[B + 1];
                // Im / Re of conj(w) psi[partner] for both qubits at once; Im(conj(w) psi) at the lane
                const f2 PX = pk_fma(splat(w.x), pim[B], -(splat(w.y) * pre[B]));
                const f2 PY = pk_fma(splat(w.x), pre[B], splat(w.y) * pim[B]) * sgp[J];
                const float u = fmaf(w.x, o.y, -w.y * o.x);
                const f2 PZ = sgp[J] * splat(-u);
                part[ZYZ3_ROW * (R0 + 0)] = PX.x;
                part[ZYZ3_ROW * (R0 + 1)] = PY.x;
                part[ZYZ3_ROW * (R0 + 2)] = PZ.x;
                part[ZYZ3_ROW * (R0 + 3)] = PX.y;
                part[ZYZ3_ROW * (R0 + 4)] = PY.y;
                part[ZYZ3_ROW * (R0 + 5)] = PZ.y;
                if constexpr (B + 1 < K) part[ZYZ3_ROW * (R0 + 6)] = m11[JN] * u;
                z3_stage<DN, true>(w, ent[B].w);
                z3_stage<UP, true>(w, ent[B].z);
                w = z3_cmul_conj(f2{ent[B].x, ent[B].y}, w);
            }, typename make_iseq<M>::type{});
            {   // the surface slice s_-1: all three qubits, and the |11> sum of block 0
                const f2 o = own[0];
                const float u = fmaf(w.x, o.y, -w.y * o.x);
                static_for([&](auto qq) __attribute__((always_inline)) {
                    constexpr int Q = decltype(qq)::value;
                    const f2 zq = w * swp(ps[Q]), yq = w * ps[Q];
                    part[ZYZ3_ROW * (3 * Q + 0)] = zq.x - zq.y;
                    part[ZYZ3_ROW * (3 * Q + 1)] = sg[Q] * (yq.x + yq.y);
                    part[ZYZ3_ROW * (3 * Q + 2)] = -sg[Q] * u;
                }, typename make_iseq<3>::type{});
                part[ZYZ3_ROW * 9] = m11[0] * u;
            }
            row_sums(0, RA, NRA);
            CPF_STAMP(5)                                  // gradient half + row sums
            barrier();
            CPF_STAMP(6)
        }
#ifdef CPF_STAMPS
        if (blockIdx.x == 0 && pair == 0 && lane == 0 && a.hist_loss) for (int k = 0; k < 8; k++) a.hist_loss[k] = (float)stamp_sum[k];
#endif
        return;
    }

    // =========================================================================================================================
    // wave B: blocks M .. K-1, and the owner of every angle: lane w < 2 K owns side w (block w >> 1, side w & 1), lanes 2 K .. 2 K + 2
    // the surface gates of the three qubits
    // =========================================================================================================================
#if CPF_Z3_PRIO
    // wave B's instruction stream is the critical path of every step: while every workgroup of the launch is resident (a.prio, set by
    // the host) it gets the higher issue priority on the SIMD it shares with a wave A (config 2: 5.38 -> 5.16 ms; in a launch of
    // several rounds of workgroups the priority starves the other waves: 2048 samples 11.6 -> 14.8 ms)
    if (a.prio) __builtin_amdgcn_s_setprio(CPF_Z3_PRIO);
#endif
    // the owner wave and, in it, the sample and the item of this lane
    const bool owner_wave = pair == 0;
    const bool upper = PAIRS > 1 && lane >= 32;
    const int il = PAIRS > 1 ? (lane & 31) : lane, lane0 = lane - il;
    float* osmem = smem_wg + (upper ? Lds::TOTAL : 0);
    const int64_t os_raw = (int64_t)blockIdx.x * PAIRS + (upper ? 1 : 0);
    const bool ovs = PAIRS == 1 || os_raw < a.nsamples;
    const int64_t os_idx = ovs ? os_raw : (int64_t)a.nsamples - 1;
    const float* init = a.init + os_idx * P;
    const uint8_t* frozen = a.frozen ? a.frozen + os_idx * P : nullptr;
    float* out_best = a.out_params + (os_idx * 2 + 1) * P;
    const float2 vt_ = a.target[row * 8 + col];
    const f2 vt = {vt_.x, vt_.y};
    // static structure: qubit of side w; every side on a qubit as a bit mask
    auto qubit_of = [&](int w) {
        int r = 0;
        static_for([&](auto jj) __attribute__((always_inline)) {
            constexpr int J = decltype(jj)::value;
            if ((w >> 1) % LL == J) r = (w & 1) ? PL::v[2 * J + 1] : PL::v[2 * J];
        }, typename make_iseq<LL>::type{});
        return r;
    };
    int lastside[3] = {-1, -1, -1};
    for (int w = 0; w < 2 * K; w++) lastside[qubit_of(w)] = w;
    const bool is_side = owner_wave && il < 2 * K, is_qubit = owner_wave && il >= 2 * K && il < NITEM;
    const int ib = il >> 1, iside = il & 1, iq = il - 2 * K;
    int prev = -1, prev_dn = -1;                             // previous side on the side's qubit (even lanes: also of the block's dn side)
    unsigned long long later = 0;                            // sides: later sides on the qubit; qubits: every side on it
    int pj[4] = {-1, -1, -1, -1};
    if (is_side) {
        const int q = qubit_of(il);
        for (int w = il - 1; w >= 0; w--) if (qubit_of(w) == q) { prev = w; break; }
        for (int w = il + 1; w < 2 * K; w++) if (qubit_of(w) == q) later |= 1ull << (lane0 + w);          // bit = the lane that owns the side
        if (iside == 0) { const int qd = qubit_of(il + 1); for (int w = il - 1; w >= 0; w--) if (qubit_of(w) == qd) { prev_dn = w; break; } }
#pragma unroll
        for (int k = 0; k < NR; k++) pj[k] = NS + NBP * ib + 2 * k + iside;
        if (ENT == 0 && iside == 0) pj[3] = NS + NBP * ib + 2 * NR;
    } else if (is_qubit) {
        for (int w = 0; w < 2 * K; w++) if (qubit_of(w) == iq) later |= 1ull << (lane0 + w);
        pj[0] = 3 * iq; pj[1] = 3 * iq + 1; pj[2] = 3 * iq + 2;
    }
    f2 th01, th23, mm01 = {0.f, 0.f}, mm23 = {0.f, 0.f}, vv01 = {0.f, 0.f}, vv23 = {0.f, 0.f};
    f2 g01 = {0.f, 0.f}, g23 = {0.f, 0.f}, nlr01, nlr23;
    {
        float t0[4], nl[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            t0[k] = 0.f; nl[k] = 0.f;
            if (pj[k] >= 0) {
                t0[k] = init[pj[k]];
                nl[k] = (frozen && frozen[pj[k]]) ? 0.f : -a.lr;
                if (ovs) { a.out_params[(os_idx * 2) * P + pj[k]] = t0[k]; out_best[pj[k]] = t0[k]; }
            }
        }
        th01 = f2{t0[0], t0[1]}; th23 = f2{t0[2], t0[3]}; nlr01 = f2{nl[0], nl[1]}; nlr23 = f2{nl[2], nl[3]};
    }
    const bool has_cp = ENT == 0 && is_side && iside == 0;
    float cs[4][2] = {{1.f, 0.f}, {1.f, 0.f}, {1.f, 0.f}, {1.f, 0.f}};
    zyz::Side sd = zyz::Side{};
    int cacb = 0;                                            // even side lanes: where the entangler's phase sits
    float psl = 0.f;
    // what the chain rule reads (float offsets into the row totals)
    int rowP = 0, row_s11 = 0, row_pzu = 0, row_pzd = 0;
    if (is_side) {
        rowP = Lds::row_blk(ib) + 3 * iside;
        row_s11 = ib == 0 ? 9 : Lds::row_blk(ib - 1) + 6;
        auto pz_row = [&](int pw, int w) { return pw >= 0 ? Lds::row_blk(pw >> 1) + 3 * (pw & 1) + 2 : 3 * qubit_of(w) + 2; };   // Pz of the previous side, or of the surface state
        row_pzu = pz_row(prev, il);
        row_pzd = iside == 0 ? pz_row(prev_dn, il + 1) : 0;
    } else if (is_qubit) rowP = 3 * iq;
    const int prev_off = Lds::REC1 + 4 * (prev >= 0 ? prev : 2 * K);
    // final left factors: the last side on each row qubit, read at the start of every step
    int lfin_off[3];
#pragma unroll
    for (int q = 0; q < 3; q++) lfin_off[q] = Lds::REC1 + 4 * (lastside[q] >= 0 ? lastside[q] : 2 * K) + 2 * (lb[q] ? 1 : 0);

    float best = 0.f, first = 0.f;
#if CPF_Z3_STAGGER > 0
    // Samples of a launch run in lock step (the same instruction streams), and a SIMD hosts wave A of one sample and wave B of
    // another: while every wave B is in its owner phase every wave A idles, and the sweeps of both then compete for the same issue
    // slots.  Half of the samples therefore start a fraction of a step late - by the SIMD wave B sits on, so that the two waves of
    // a SIMD belong to samples of opposite phase (the hardware places the two waves of consecutive workgroups on SIMDs (0, 1),
    // (1, 2), (2, 3), (3, 0)).  Wave A waits at the barrier below.
    {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        if ((hw >> 4) & 1) {
#pragma unroll
            for (int k = 0; k < CPF_Z3_STAGGER; k++) __builtin_amdgcn_s_sleep(8);          // 8 x 64 clocks each
        }
    }
#endif
    barrier();                                               // penalty table, constant records: written by wave A's lanes
#ifdef CPF_STAMPS
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev) :: "memory");
#endif
    for (int it = 0; it < iters; it++) {
        // ================= owner: Adam closes the previous step; (cos, sin); sides, parities, e-pairs, tables, surface matrices
        if (owner_wave) {
        if (it > 0) {
            // optax scale_by_adam (optimization.py:14-25): the same operations in the same order as the scalar form, per half
            const float2 bc = const_load_f2(a.bias, it - 1);
            const f2 ibc = f2{__builtin_amdgcn_rcpf(bc.x), __builtin_amdgcn_rcpf(bc.y)};
            if (has_cp && a.use_reg) g23.y += psl;
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
            osmem[Lds::PV + ib] = penalty_lds(osmem, a.pen_kind, th23.y, sl);
            psl = a.r * sl;
        }
        {
            auto rrec = [&](int idx, float& c, float& sn, int& kind) __attribute__((always_inline)) {
                const int k = (idx - 1) >> 1;
                c = cs[k][0]; sn = cs[k][1]; kind = 0;
            };
            sd = zyz::side_build<ROT & 0xFFFF>(rrec, NR, iside);
            if (is_side) *reinterpret_cast<float4*>(osmem + Lds::REC1 + 4 * il) = make_float4(sd.l0.re, sd.l0.im, sd.l1.re, sd.l1.im);
            const unsigned long long fbits = __builtin_amdgcn_ballot_w64(is_side && sd.f != 0);
            sd.m = __popcll(fbits & later) & 1;
            if (is_qubit) {                                  // the qubit's surface matrix with every flip on the qubit applied
                float sm[8];
                zyz::surface_matrix(sm, cs[0][0], cs[0][1], cs[1][0], cs[1][1], cs[2][0], cs[2][1], sd.m);
                *reinterpret_cast<float4*>(osmem + Lds::SURFM + 8 * iq) = make_float4(sm[0], sm[1], sm[2], sm[3]);
                *reinterpret_cast<float4*>(osmem + Lds::SURFM + 8 * iq + 4) = make_float4(sm[4], sm[5], sm[6], sm[7]);
            }
            lds_fence();
            const float4 pl = *reinterpret_cast<const float4*>(osmem + prev_off);
            const zyz::EPair e = zyz::side_epair(sd, true, zyz::c2(pl.x, pl.y), zyz::c2(pl.z, pl.w));      // "no previous side" reads the identity record
            // the block's other side is the neighbouring lane: exchange the e-pairs instead of a trip through LDS, then this lane
            // builds the two entries with a_up = iside (side 0: entries (0, 0), (0, 1); side 1: (1, 0), (1, 1))
            const f2 pe0 = {z3_nb(e.e0.re), z3_nb(e.e0.im)}, pe1 = {z3_nb(e.e1.re), z3_nb(e.e1.im)};
            const int pc = z3_nb(e.c);
            const float psc = z3_nb(sd.sc), pt = z3_nb(sd.t);
            const float plc = z3_nb(cs[3][0]), pls = z3_nb(cs[3][1]);          // the CP angle belongs to side 0
            const f2 lam = ENT != 0 ? f2{-1.f, 0.f} : (iside ? f2{plc, pls} : f2{cs[3][0], cs[3][1]});
            const f2 my0 = {e.e0.re, e.e0.im}, my1 = {e.e1.re, e.e1.im};
            const f2 X = iside ? pe1 : my0, Y0 = iside ? my0 : pe0, Y1 = iside ? my1 : pe1;
            const int cu = iside ? pc : e.c, cd = iside ? e.c : pc;
            f2 d0 = z3_cmul(X, Y0), d1 = z3_cmul(X, Y1);
            const f2 dl = z3_cmul(lam, cd ? d1 : d0);
            const bool hit = cu == iside;
            if (hit && cd == 0) d0 = dl;
            if (hit && cd != 0) d1 = dl;
            const float scale = sd.sc * psc;
            d0 *= splat(scale); d1 *= splat(scale);
            const float tu = iside ? pt : sd.t, td = iside ? sd.t : pt;
            const float tus = iside ? tu : -tu;
            cacb = e.c | (pc << 1);
            if (is_side) {
                float4* dst = reinterpret_cast<float4*>(osmem + Lds::TAB + 16 * ib + 8 * iside);
                dst[0] = make_float4(d0.x, d0.y, tus, -td);
                dst[1] = make_float4(d1.x, d1.y, tus, td);
            }
        }
        }
        CPF_STAMP(0)                                  // owner / build
        barrier();
        CPF_STAMP(1)

        // ================= forward half: Vd = conj(dfin) V = W(s_{K-1}), then blocks K-1 .. M un-applied.  The table entries are read
        // once and serve both sweeps; the states stay in registers for the gradient half.
        float4 ent[K - M];
        f2 own[K - M + 1];
        float4 pvq[(K + 3) / 4];
#pragma unroll
        for (int k = 0; k < (K + 3) / 4; k++) pvq[k] = reinterpret_cast<const float4*>(osmem + Lds::PV)[k];
        float2 lf[3];
#pragma unroll
        for (int q = 0; q < 3; q++) lf[q] = *reinterpret_cast<const float2*>(smem + lfin_off[q]);
        static_for([&](auto bb) __attribute__((always_inline)) {
            constexpr int B = M + decltype(bb)::value;
            ent[B - M] = *reinterpret_cast<const float4*>(tab + 16 * B + kofs[B % LL]);
        }, typename make_iseq<K - M>::type{});
        f2 x = z3_cmul_conj(z3_cmul(f2{lf[2].x, lf[2].y}, z3_cmul(f2{lf[1].x, lf[1].y}, f2{lf[0].x, lf[0].y})), vt);
        sp[64 * (K + 1)] = x; own[K - M] = x;
        static_for([&](auto bb) __attribute__((always_inline)) {
            constexpr int B = K - 1 - decltype(bb)::value, J = B % LL;
            z3_stage<PL::v[2 * J + 1], true>(x, ent[B - M].w);
            z3_stage<PL::v[2 * J], true>(x, ent[B - M].z);
            x = z3_cmul_conj(f2{ent[B - M].x, ent[B - M].y}, x);
            sp[64 * (B + 1)] = x; own[B - M] = x;                            // W(s_{B-1})
        }, typename make_iseq<K - M>::type{});
        CPF_STAMP(2)                                  // forward half
        barrier();
        CPF_STAMP(3)

        // ================= loss: T = sum conj(W) psi at the cut (wave A forms the same T for its cotangent)
        const f2 psi_c = sp[64 * M];
        f2 pre[K - M], pim[K - M];
        static_for([&](auto bb) __attribute__((always_inline)) {
            constexpr int B = M + decltype(bb)::value, J = B % LL;
            const float* qu = spx[PL::v[2 * J]] + 128 * (B + 2);
            const float* qd = spx[PL::v[2 * J + 1]] + 128 * (B + 2);
            pre[B - M] = f2{qu[0], qd[0]}; pim[B - M] = f2{qu[1], qd[1]};
        }, typename make_iseq<K - M>::type{});
        const f2 tt = z3_cmul_conj(x, psi_c);
        float tr = tt.x, ti = tt.y;
        z3_allsum(tr, ti, odd);
        if (owner_wave) {
            float otr = tr, oti = ti;                 // T of the sample this lane's item belongs to
            if constexpr (PAIRS > 1) {                // the second sample's T: its wave pair forms it too, but not where the angles live
                const f2* sp1 = reinterpret_cast<const f2*>(smem_wg + Lds::TOTAL + Lds::STATES) + lane;
                const f2 t1 = z3_cmul_conj(sp1[64 * (M + 1)], sp1[64 * M]);
                float tr1 = t1.x, ti1 = t1.y;
                z3_allsum(tr1, ti1, odd);
                otr = upper ? tr1 : tr; oti = upper ? ti1 : ti;
            }
            float pen_sum = 0.f;
            if (a.use_reg) {
#pragma unroll
                for (int k = 0; k < (K + 3) / 4; k++) pen_sum += (pvq[k].x + pvq[k].y) + (pvq[k].z + pvq[k].w);
            }
            const float reg = a.use_reg ? a.r * pen_sum : 0.f;
            const float loss = 1.f - fmaf(otr, otr, oti * oti) * a.inv_dim2;
            const float regloss = loss + reg;
            bool improved = false;
            if (it == 0) { first = regloss; best = regloss; }
            else if (regloss < best) { best = regloss; improved = true; }
            const float t0[4] = {th01.x, th01.y, th23.x, th23.y};
            if (improved && ovs) {
#pragma unroll
                for (int k = 0; k < 4; k++) if (pj[k] >= 0) out_best[pj[k]] = t0[k];
            }
#ifndef CPF_STAMPS
            if (a.hist_params && ovs) {
#pragma unroll
                for (int k = 0; k < 4; k++) if (pj[k] >= 0) a.hist_params[(os_idx * a.T + it) * P + pj[k]] = t0[k];
            }
            if (a.hist_loss && il == 0 && ovs) a.hist_loss[os_idx * a.T + it] = regloss;
#endif
        }
        if (a.T == 0) break;
        if (it + 1 == iters) break;                   // the last step's gradient would move the angles past the last evaluated iterate
        CPF_STAMP(4)                                  // loss

        // ================= gradient half: blocks M .. K-1 applied to psi' = conj(f) psi (the scalar moves onto psi once), W(s_b) at hand
        {
            f2 pp = z3_cmul_conj(f2{-a.inv_dim2 * tr, -a.inv_dim2 * ti}, psi_c);
            static_for([&](auto bb) __attribute__((always_inline)) {
                constexpr int B = M + decltype(bb)::value, J = B % LL, JN = (B + 1) % LL;
                constexpr int UP = PL::v[2 * J], DN = PL::v[2 * J + 1], R0 = Lds::row_blk(B);
                pp = z3_cmul(f2{ent[B - M].x, ent[B - M].y}, pp);
                z3_stage<UP>(pp, ent[B - M].z);
                z3_stage<DN>(pp, ent[B - M].w);
                const f2 o = own[B - M + 1];                                      // W(s_B)
                // the roles swapped: Im / Re of conj(W[partner]) psi' (the sign of the y sums changes), Im(conj(W) psi') at the lane
                const f2 PX = pk_fma(splat(pp.y), pre[B - M], -(splat(pp.x) * pim[B - M]));
                const f2 PY = pk_fma(splat(pp.x), pre[B - M], splat(pp.y) * pim[B - M]) * -sgp[J];
                const float u = fmaf(o.x, pp.y, -o.y * pp.x);
                const f2 PZ = sgp[J] * splat(-u);
                part[ZYZ3_ROW * (R0 + 0)] = PX.x;
                part[ZYZ3_ROW * (R0 + 1)] = PY.x;
                part[ZYZ3_ROW * (R0 + 2)] = PZ.x;
                part[ZYZ3_ROW * (R0 + 3)] = PX.y;
                part[ZYZ3_ROW * (R0 + 4)] = PY.y;
                part[ZYZ3_ROW * (R0 + 5)] = PZ.y;
                if constexpr (B + 1 < K) part[ZYZ3_ROW * (R0 + 6)] = m11[JN] * u;
            }, typename make_iseq<K - M>::type{});
        }
        row_sums(RA, ROWS, NRB);
        CPF_STAMP(5)                                  // gradient half + row sums
        barrier();
        CPF_STAMP(6)

        // ================= chain rule, into the owners' registers
        if (owner_wave) {
            const float* tot = osmem + Lds::TOT;
            const float px = tot[rowP], py = tot[rowP + 1], pz = tot[rowP + 2];
            const float s11 = tot[row_s11], pzu = tot[row_pzu], pzd = tot[row_pzd];
            auto rrec = [&](int idx, float& c, float& sn, int& kind) __attribute__((always_inline)) {
                const int k = (idx - 1) >> 1;
                c = cs[k][0]; sn = cs[k][1]; kind = 0;
            };
            float gp[4] = {0.f, 0.f, 0.f, 0.f};
            zyz::chain_side<ROT & 0xFFFF>(gp, px, py, pz, sd, rrec, NR, iside);
            if (ENT == 0) gp[3] = zyz::chain_cp(s11, pzu, pzd, cacb, 1.f);
            if (is_qubit) {
                float gs[3];
                zyz::chain_surface(gs, px, py, pz, sd.m, cs[1][0], cs[1][1], cs[2][0], cs[2][1]);
                gp[0] = gs[0]; gp[1] = gs[1]; gp[2] = gs[2]; gp[3] = 0.f;
            }
            g01 = f2{gp[0], gp[1]}; g23 = f2{gp[2], gp[3]};
        }
        CPF_STAMP(7)
    }
#ifdef CPF_STAMPS
    if (blockIdx.x == 0 && pair == 0 && lane == 0 && a.hist_loss) for (int k = 0; k < 8; k++) a.hist_loss[8 + k] = (float)stamp_sum[k];
#endif

    // ---- epilogue (optimization.py:365-370, cp_utils.py:144-164): reg recompute, loss = regloss - reg, argmin, CZ count, entry-loss flag
    if (owner_wave) {
        const int bi = (best < first) ? 1 : 0;
        float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f;
        if (has_cp) {
            const int j = pj[3];
            const float t0 = a.out_params[(os_idx * 2) * P + j], t1 = out_best[j];
            float sl;
            if (a.use_reg) { acc0 = penalty_lds(osmem, a.pen_kind, t0, sl); acc1 = penalty_lds(osmem, a.pen_kind, t1, sl); }
            acc2 = cz_value(bi ? t1 : t0, a.thr_cp);
        }
        if constexpr (PAIRS > 1) half_sum3(acc0, acc1, acc2, upper);
        else wave_sum3(acc0, acc1, acc2);
        if (il == 0 && ovs) {
            const float reg0 = a.use_reg ? a.r * acc0 : 0.f, reg1 = a.use_reg ? a.r * acc1 : 0.f;
            const float l0 = first - reg0, l1 = best - reg1;
            a.out_regloss[os_idx * 2 + 0] = first; a.out_regloss[os_idx * 2 + 1] = best;
            a.out_reg[os_idx * 2 + 0] = reg0;      a.out_reg[os_idx * 2 + 1] = reg1;
            a.out_loss[os_idx * 2 + 0] = l0;       a.out_loss[os_idx * 2 + 1] = l1;
            a.out_cz[os_idx] = (int32_t)acc2;
            a.out_pass[os_idx] = ((bi ? l1 : l0) <= a.entry_loss) ? 1 : 0;
        }
    }
}

template <int ROT, int KC, int MC, int PAIRS, int... UD>
__global__ __launch_bounds__(128 * PAIRS, 1) void k_cp_zyz3(const KArgs a) {
    zyz3_body<ROT, KC, MC, PAIRS, UD...>(a);
}

}  // namespace cpf
