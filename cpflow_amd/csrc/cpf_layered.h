// cpflow_amd: the layer-specialised persistent kernel, any n <= 6, 1 or 4 waves per sample.
//
// Layout Lay<NQ, LW> (cpf_device.h); the connectivity
// layer is a compile-time sequence of (up, dn) pairs, so the block loop is straight-line code per layer
// (no per-gate or per-block dispatch), the (cos, sin) of a block arrive as one prefetched 64-byte LDS
// record, the 7 gradient partials of a block are reduced together by the DPP reduce-scatter, and the
// optimiser state lives in the owner threads' registers.
//
// Instantiated ahead of time for the reference's two 3-qubit layers (chain, all-to-all) and at run time
// (cpf_jit.h) for every other (n, waves, rotation string, layer); the generic kernel is the fallback.
// Requires n == NQ (2-qubit problems, which run embedded, use the generic kernel).
#pragma once
#include "cpf_blocks.h"

namespace cpf {

// Diagnostic build only (-DCPF_STAMPS): per-phase cycle sums (s_memtime) of sample 0, written to out_grad[0..7].
#ifdef CPF_STAMPS
#define CPF_STAMP(slot)                                                                                        \
    {                                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        unsigned long long t_;                                                                                 \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");                           \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        stamp_sum[slot] += t_ - stamp_prev;                                                                    \
        stamp_prev = t_;                                                                                       \
    }
#else
#define CPF_STAMP(slot)
#endif

#ifndef CPF_UNR6
#define CPF_UNR6 1
#endif

template <int... I> struct iseq {};
template <int N, int... I> struct make_iseq : make_iseq<N - 1, N - 1, I...> {};
template <int... I> struct make_iseq<0, I...> { using type = iseq<I...>; };
template <int V> struct ic { static constexpr int value = V; };
template <class F, int... I>
__device__ __forceinline__ void static_for(F&& f, iseq<I...>) { (f(ic<I>{}), ...); }

template <int... UD> struct PairList { static constexpr int v[sizeof...(UD) > 0 ? sizeof...(UD) : 1] = {UD...}; };

template <int NQ>
struct LayeredLds {                                       // float offsets; [0, 28) = penalty table
    static constexpr int SURF_F = ((6 * NQ + 3) / 4) * 4;           // surface (cos, sin) record
    static constexpr int NBATCH = (3 * NQ + 7) / 8;                 // reduce8 batches of the surface gradients
    static constexpr int SURF_CS = PEN_FLOATS;
    static constexpr int BLK_CS = SURF_CS + SURF_F;                 // + 16 b
    __host__ __device__ static constexpr int g_base(int K) { return BLK_CS + 16 * K; }
    __host__ __device__ static constexpr int g_wave(int K) { return 32 * (NBATCH + K); }      // floats per wave
    __host__ __device__ static constexpr int g_surf(int K, int w) { return g_base(K) + w * g_wave(K); }
    __host__ __device__ static constexpr int g_blk(int K, int w, int b) { return g_surf(K, w) + 32 * NBATCH + 32 * b; }
    __host__ __device__ static constexpr int red(int K, int wps) { return g_base(K) + wps * g_wave(K); }
    // stored merged-diagonal phases (3-qubit kernel, MSTORE = 1): (K + 1) units x 64 lanes x (re, im)
    __host__ __device__ static constexpr int phase(int K, int wps) { return red(K, wps) + 8 * wps + 4; }
    __host__ __device__ static constexpr int total(int K, int wps, bool merge) {
        return phase(K, wps) + (merge ? 128 * (K + 1) : 0);
    }
};

#ifndef CPF_SHARE_PARTNER
#define CPF_SHARE_PARTNER 1   // A/B: 0 = gradient and un-application of a lane-qubit rotation fetch psi's partners separately
#endif
template <class L, int KIND, int Q, bool SH = false>
__device__ __forceinline__ float lay_rot_bwd(State<L>& st, float c, float s);
template <class L>
__device__ __forceinline__ void gate_fence();

// Merging runs of diagonal gates (one element per lane, rotation string ending in 'z').  In program order the
// circuit reads ... Rz(up) Rz(dn) | CP(next block) ...: three diagonal gates in a row.  Forward, their product is
// one per-lane phase rho (12 ops to build, 4 to apply: what the three gates cost anyway), which is also stored
// in LDS (8 bytes per lane).  Backward, conj(C) psi is invariant under a common phase, so the three angle
// gradients share one product w = Im(conj(C) psi), and un-applying the run is one complex multiply by
// conj(rho) for psi and one for C: 15 ops instead of 37 per block.
__host__ __device__ constexpr bool layered_merge(int nq, int lw, int rot) {
    return nq == 3 && lw == 0 && rot_nr(rot) >= 1 && rot_kind(rot, rot_nr(rot) - 1) == OP_RZ && rot_ent(rot) != 2;   // a CX block is not diagonal
}
template <class L, int PUP, int PDN, int UP, int DN>
__device__ __forceinline__ f2 diag_phase(const State<L>& st, float cu, float su, float cd, float sd, float cc, float sc) {
    constexpr int Q0 = UP < DN ? UP : DN, Q1 = UP < DN ? DN : UP;
    const f2 zu = f2{cu, su} * st.sgp[PUP], zd = f2{cd, sd} * st.sgp[PDN];   // Rz(a) multiplies by cos(a/2) + i sigma sin(a/2)
    const bool p = cp_lane_pred<L, Q0, Q1>(st);                              // CP multiplies the |11> rows by e^{ia}
    const f2 ze = {p ? cc : 1.f, p ? sc : 0.f};
    return cmul(cmul(zu, zd), ze);
}
template <class L>
__device__ __forceinline__ void phase_apply(f2& x, f2 rho, bool conj) { x = conj ? cmul_conj(x, rho) : cmul(x, rho); }
// the rotations of a block except the final letter's pair (used with merged diagonal runs)
template <class L, int UP, int DN, int ROT>
__device__ __forceinline__ void lay_block_fwd_inner(State<L>& st, const float (&r)[16]) {
    constexpr int NR = rot_nr(ROT);
#pragma unroll
    for (int i = 0; i < NR - 1; i++) {
        if (rot_kind(ROT, i) == OP_RX) {
            rot_apply<L, OP_RX, UP, shear_gate<L, OP_RX, UP>()>(st.p, r[4 * i], r[4 * i + 1], st); rot_apply<L, OP_RX, DN, shear_gate<L, OP_RX, DN>()>(st.p, r[4 * i + 2], r[4 * i + 3], st);
        } else if (rot_kind(ROT, i) == OP_RY) {
            rot_apply<L, OP_RY, UP, shear_gate<L, OP_RY, UP>()>(st.p, r[4 * i], r[4 * i + 1], st); rot_apply<L, OP_RY, DN, shear_gate<L, OP_RY, DN>()>(st.p, r[4 * i + 2], r[4 * i + 3], st);
        } else {
            rot_apply<L, OP_RZ, UP>(st.p, r[4 * i], r[4 * i + 1], st); rot_apply<L, OP_RZ, DN>(st.p, r[4 * i + 2], r[4 * i + 3], st);
        }
    }
}
template <class L, int UP, int DN, int ROT>
__device__ __forceinline__ void lay_block_bwd_inner(State<L>& st, const float (&r)[16], float (&d)[8]) {
    constexpr int NR = rot_nr(ROT);
#pragma unroll
    for (int i = NR - 2; i >= 0; i--) {
        if (rot_kind(ROT, i) == OP_RX) {
            d[2 * i + 1] = lay_rot_bwd<L, OP_RX, DN, shear_gate<L, OP_RX, DN>()>(st, r[4 * i + 2], r[4 * i + 3]); d[2 * i] = lay_rot_bwd<L, OP_RX, UP, shear_gate<L, OP_RX, UP>()>(st, r[4 * i], r[4 * i + 1]);
        } else if (rot_kind(ROT, i) == OP_RY) {
            d[2 * i + 1] = lay_rot_bwd<L, OP_RY, DN, shear_gate<L, OP_RY, DN>()>(st, r[4 * i + 2], r[4 * i + 3]); d[2 * i] = lay_rot_bwd<L, OP_RY, UP, shear_gate<L, OP_RY, UP>()>(st, r[4 * i], r[4 * i + 1]);
        } else {
            d[2 * i + 1] = lay_rot_bwd<L, OP_RZ, DN>(st, r[4 * i + 2], r[4 * i + 3]); d[2 * i] = lay_rot_bwd<L, OP_RZ, UP>(st, r[4 * i], r[4 * i + 1]);
        }
    }
}

// With 16+ elements per thread a layer body is a basic block of ~10^4 instructions; left alone the
// scheduler interleaves gates and drives the kernel into scratch (measured: 248 VGPRs + 464 B/lane,
// 4x slower).  A scheduling fence after every gate keeps the live set at psi + C + one record.
template <class L>
__device__ __forceinline__ void gate_fence() {
    if constexpr (L::E >= 16) __builtin_amdgcn_sched_barrier(0);
}

// one entangling block on the generic layout (CP first, then R_g(up) R_g(dn) per letter; main.py:69-82)
template <class L, int UP, int DN, int ROT>
__device__ __forceinline__ void lay_block_fwd(State<L>& st, const float (&r)[16]) {
    constexpr int NR = rot_nr(ROT), NB = 2 * NR + 1;
    constexpr int Q0 = UP < DN ? UP : DN, Q1 = UP < DN ? DN : UP;
    if constexpr (rot_ent(ROT) == 2) cx_apply<L, UP, DN>(st.p, st);            // gates.py:40-43: control = up, target = down
    else cp_apply<L, Q0, Q1>(st.p, r[2 * (NB - 1)], r[2 * (NB - 1) + 1], st);
    gate_fence<L>();
#pragma unroll
    for (int i = 0; i < NR; i++) {
        if (rot_kind(ROT, i) == OP_RX) {
            rot_apply<L, OP_RX, UP, shear_gate<L, OP_RX, UP>()>(st.p, r[4 * i], r[4 * i + 1], st); gate_fence<L>();
            rot_apply<L, OP_RX, DN, shear_gate<L, OP_RX, DN>()>(st.p, r[4 * i + 2], r[4 * i + 3], st); gate_fence<L>();
        } else if (rot_kind(ROT, i) == OP_RY) {
            rot_apply<L, OP_RY, UP, shear_gate<L, OP_RY, UP>()>(st.p, r[4 * i], r[4 * i + 1], st); gate_fence<L>();
            rot_apply<L, OP_RY, DN, shear_gate<L, OP_RY, DN>()>(st.p, r[4 * i + 2], r[4 * i + 3], st); gate_fence<L>();
        } else if constexpr (L::E > 1 && CPF_RZPAIR != 0) {
            const RzPair<L, UP, DN> z(r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3], st);
            z.apply(st.p); gate_fence<L>();
        } else {
            rot_apply<L, OP_RZ, UP>(st.p, r[4 * i], r[4 * i + 1], st); gate_fence<L>();
            rot_apply<L, OP_RZ, DN>(st.p, r[4 * i + 2], r[4 * i + 3], st); gate_fence<L>();
        }
    }
}
// The gradient of a rotation can be read on either side of it (sigma commutes with its own rotation, so
// Im(conj(C) sigma psi) is the same before and after the gate).  It is read BEFORE the gate, i.e. after un-applying it:
// read after, the gradient and the un-application both need psi's partner amplitudes, the compiler shares them, and a
// shared DPP read is no longer folded into its multiply (8 extra v_mov_b32_dpp per lane-qubit gate at 4 elements per lane).
// Only there: for a register qubit (and with one element per lane) reading the gradient first keeps it off the
// un-application's dependency chain, which matters when a SIMD holds a single wave (5-qubit chain at 1000 samples: 141 ms
// against 148 with the late read everywhere).
// SH: a block's x / y rotation on a register qubit, whose record holds (t, s) - the inverse is (-t, -s); the surface gates'
// records hold (cos, sin) whatever the qubit
template <class L, int KIND, int Q, bool SH>
__device__ __forceinline__ float lay_rot_bwd(State<L>& st, float c, float s) {
    constexpr bool LATE = L::row_on_lane(Q) && L::E > 1 && KIND != OP_RZ;
    static_assert(!SH || shear_gate<L, KIND, Q>(), "shear form: a block's x / y rotation on a register qubit");
    float d = 0.f;
#if CPF_SHARE_PARTNER
    if constexpr (LATE) {
        // Lane qubit, several amplitudes per lane: psi's partner amplitudes are fetched ONCE (two DPP moves per amplitude) and
        // serve both the gradient - read before the gate is un-applied, with the C that belongs to that psi - and psi's
        // un-application, which then is packed arithmetic: 8 DPP + 12 packed instructions per gate at 4 amplitudes per lane
        // where separate fetches folded into multiplies cost 16 DPP + 8 (a DPP instruction takes a double issue slot).
        constexpr int BIT = L::row_bit(Q);
        f2 acc = {0.f, 0.f};
        const float ss = KIND == OP_RY ? -st.sg[Q] * s : -s;          // the inverse rotation's sine (y: with the lane's sign)
#pragma unroll
        for (int e = 0; e < L::E; e++) {
            const f2 pp = {xor_lane<BIT>(st.p[e].x), xor_lane<BIT>(st.p[e].y)};
            if constexpr (KIND == OP_RX) {
                acc = pk_fma(st.c[e], swp(pp), acc);                    // (Cr X(psi_i), Ci X(psi_r))
                st.p[e] = pk_fma(f2{ss, -ss}, swp(pp), splat(c) * st.p[e]);
            } else {
                acc = pk_fma(st.c[e], pp, acc);                         // (Cr X(psi_r), Ci X(psi_i))
                st.p[e] = pk_fma(splat(ss), pp, splat(c) * st.p[e]);
            }
        }
        d = KIND == OP_RX ? acc.x - acc.y : st.sg[Q] * (acc.x + acc.y);
        gate_fence<L>();
        rot_apply<L, KIND, Q>(st.c, c, -s, st);
        gate_fence<L>();
        return d;
    }
#endif
    if constexpr (!LATE) { d = rot_grad<L, KIND, Q>(st); gate_fence<L>(); }
    rot_apply<L, KIND, Q, SH>(st.p, SH ? -c : c, -s, st);
    gate_fence<L>();
    rot_apply<L, KIND, Q, SH>(st.c, SH ? -c : c, -s, st);
    gate_fence<L>();
    if constexpr (LATE) { d = rot_grad<L, KIND, Q>(st); gate_fence<L>(); }
    return d;
}
template <class L, int UP, int DN, int ROT>
__device__ __forceinline__ void lay_block_bwd(State<L>& st, const float (&r)[16], float (&d)[8]) {
    constexpr int NR = rot_nr(ROT), NB = 2 * NR + 1;
    constexpr int Q0 = UP < DN ? UP : DN, Q1 = UP < DN ? DN : UP;
#pragma unroll
    for (int k = 0; k < 8; k++) d[k] = 0.f;
#pragma unroll
    for (int i = NR - 1; i >= 0; i--) {
        if (rot_kind(ROT, i) == OP_RX) {
            d[2 * i + 1] = lay_rot_bwd<L, OP_RX, DN, shear_gate<L, OP_RX, DN>()>(st, r[4 * i + 2], r[4 * i + 3]); d[2 * i] = lay_rot_bwd<L, OP_RX, UP, shear_gate<L, OP_RX, UP>()>(st, r[4 * i], r[4 * i + 1]);
        } else if (rot_kind(ROT, i) == OP_RY) {
            d[2 * i + 1] = lay_rot_bwd<L, OP_RY, DN, shear_gate<L, OP_RY, DN>()>(st, r[4 * i + 2], r[4 * i + 3]); d[2 * i] = lay_rot_bwd<L, OP_RY, UP, shear_gate<L, OP_RY, UP>()>(st, r[4 * i], r[4 * i + 1]);
        } else if constexpr (L::E > 1 && CPF_RZPAIR != 0) {
            // both gradients before the pair is un-applied (conj(C) sigma_z psi does not change under a diagonal gate on psi and C)
            d[2 * i + 1] = rot_grad<L, OP_RZ, DN>(st); d[2 * i] = rot_grad<L, OP_RZ, UP>(st); gate_fence<L>();
            const RzPair<L, UP, DN> z(r[4 * i], -r[4 * i + 1], r[4 * i + 2], -r[4 * i + 3], st);
            z.apply(st.p); gate_fence<L>();
            z.apply(st.c); gate_fence<L>();
        } else {
            d[2 * i + 1] = lay_rot_bwd<L, OP_RZ, DN>(st, r[4 * i + 2], r[4 * i + 3]); d[2 * i] = lay_rot_bwd<L, OP_RZ, UP>(st, r[4 * i], r[4 * i + 1]);
        }
    }
    if constexpr (rot_ent(ROT) == 2) {                   // CNOT: a row permutation, its own inverse, no parameter
        cx_apply<L, UP, DN>(st.p, st);
        cx_apply<L, UP, DN>(st.c, st);
    } else {
        const float cc = r[2 * (NB - 1)], sc = r[2 * (NB - 1) + 1];
        d[NB - 1] = cp_grad<L, Q0, Q1>(st);
        cp_apply<L, Q0, Q1>(st.p, cc, -sc, st);
        cp_apply<L, Q0, Q1>(st.c, cc, -sc, st);
    }
    gate_fence<L>();
}

// The surface gates, backwards.  Gates on different qubits commute, so each qubit's Rz Rx Rz is differentiated
// from the state after the WHOLE surface layer, un-applying only that qubit's own gates: independent chains
// (instruction-level parallelism for a lone wave), two un-applications per qubit instead of three, and the
// product conj(C) psi of the three final Rz gradients is shared.  Small tiles only: each chain works on a copy.
template <class L, int NQ, int SF, int ND>
__device__ __forceinline__ void surface_bwd_independent(const State<L>& st, const float (&sr)[SF], float (&ds)[ND]) {
    static_for([&](auto qq) __attribute__((always_inline)) {
        constexpr int Q = decltype(qq)::value;
        State<L> s = st;
        ds[3 * Q + 2] = rot_grad<L, OP_RZ, Q>(s);
        rot_apply<L, OP_RZ, Q>(s.p, sr[6 * Q + 4], -sr[6 * Q + 5], s);
        rot_apply<L, OP_RZ, Q>(s.c, sr[6 * Q + 4], -sr[6 * Q + 5], s);
        ds[3 * Q + 1] = rot_grad<L, OP_RX, Q>(s);
        rot_apply<L, OP_RX, Q>(s.p, sr[6 * Q + 2], -sr[6 * Q + 3], s);
        rot_apply<L, OP_RX, Q>(s.c, sr[6 * Q + 2], -sr[6 * Q + 3], s);
        ds[3 * Q + 0] = rot_grad<L, OP_RZ, Q>(s);
    }, typename make_iseq<NQ>::type{});
}

// UD... = up0, dn0, up1, dn1, ...: the layer (topology.py:15-20: block b sits on pair b mod len)
// MSTORE: see STORE below (only meaningful when layered_merge(NQ, LW, ROT))
template <int NQ, int LW, int ROT, int PPT, int MSTORE, int... UD>
__device__ __forceinline__ void layered_body(const KArgs& a) {
    using L = Lay<NQ, LW>;
    using Lds = LayeredLds<NQ>;
    using PL = PairList<UD...>;
    constexpr int E = L::E, WPS = L::WPS, THREADS = L::THREADS;
    constexpr int LL = sizeof...(UD) / 2;
    // blocks per unrolled loop body: a whole number of layers, even (record-buffer parity); for one element per
    // lane (blocks of ~35 / ~95 instructions) a multiple of 6, so that loop back-edges — where the compiler waits
    // for every outstanding LDS operation — come once per 6 blocks
    constexpr int UNR2 = (LL % 2 == 0) ? LL : 2 * LL;
    constexpr int UNR = (E == 1 && CPF_UNR6) ? ((UNR2 % 3 == 0) ? UNR2 : 3 * UNR2) : UNR2;
    constexpr int NR = rot_nr(ROT), NB = 2 * NR + 1, NS = 3 * NQ;
    constexpr int ENT = rot_ent(ROT), NBP = NB - (ENT != 0 ? 1 : 0);          // parameters per block (CZ: the record's last slot is a constant)
    static_assert(ENT >= 0 && ENT <= 2, "layered kernel: CP, CZ or CX blocks");
    constexpr bool MERGE = layered_merge(NQ, LW, ROT);
    // MSTORE = 1: the forward pass keeps each run's phase in LDS (8 B per lane per block) for the backward pass;
    // MSTORE = 0: the backward pass rebuilds it (12 more ops per block, no LDS: better once LDS limits occupancy)
    constexpr bool STORE = MERGE && (MSTORE != 0);
    static_assert(LL >= 1 && NR >= 1 && NR <= 3, "layered kernel: 1..3 rotation letters, non-empty layer");
    extern __shared__ __attribute__((aligned(16))) float smem_wg[];
    // One wave per sample: a workgroup may hold several samples (KArgs::group), one per wave, each with its own copy of the LDS
    // map - the waves never meet (lds_sync<1> is no hardware barrier).  The dispatcher puts the waves of ONE workgroup on the
    // CU's four SIMDs in turn, which single-wave workgroups are not guaranteed: 1000 of them on 1024 SIMDs left SIMDs with
    // two waves and others empty in some launches (5-qubit chain, 36 blocks, 1000 samples: 118 or 190 ms from run to run).
    const int sub = L::WPS == 1 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) : 0;
    const int tid = L::WPS == 1 ? ((int)threadIdx.x & 63) : (int)threadIdx.x, lane = tid & 63;
    const int wave = L::WPS == 1 ? 0 : __builtin_amdgcn_readfirstlane(tid >> 6);
    // NSP: surface PARAMETERS.  A 2-qubit problem runs as U (x) I on the 3-qubit kernel: the spectator's three surface
    // record slots hold the identity (cos, sin) = (1, 0) and belong to no parameter.
    const int NSP = 3 * a.n_actual;
    const int P = a.P, K = (P - NSP) / NBP;
    const int64_t s_idx = (int64_t)blockIdx.x * (a.group > 1 && L::WPS == 1 ? a.group : 1) + sub;
    if (L::WPS == 1 && s_idx >= a.nsamples) return;
    float* smem = smem_wg + sub * Lds::total(K, WPS, STORE);
    const float* init = a.init + s_idx * P;
    const uint8_t* frozen = a.frozen ? a.frozen + s_idx * P : nullptr;
    float* out_best = a.out_params + (s_idx * 2 + 1) * P;
    float* red = smem + Lds::red(K, WPS);
    f2* ph = reinterpret_cast<f2*>(smem + Lds::phase(K, WPS)) + lane;      // MERGE: unit u at ph[64 * u]

    State<L> st;
#pragma unroll
    for (int i = 0; i < L::NLR; i++) { st.lb[i] = (lane >> L::row_lane_bit(i)) & 1; st.sg[i] = st.lb[i] ? 1.f : -1.f; st.sgp[i] = f2{1.f, st.sg[i]}; }
    if constexpr (L::NQ == 3 && L::NLR == 3) {                // one element per lane: -2 on the |11> rows of each qubit pair, else 0
        st.cpm2[0] = (st.lb[0] && st.lb[1]) ? -2.f : 0.f; st.cpm2[1] = (st.lb[0] && st.lb[2]) ? -2.f : 0.f; st.cpm2[2] = (st.lb[1] && st.lb[2]) ? -2.f : 0.f;
    }
    const int col = L::col_index(lane, wave);
    // the target stays in registers for small tiles; with 16+ elements per thread it is re-read from
    // L2 once per step (2 x 8 E bytes per thread per ~1e5 cycles) to keep the kernel out of scratch
    constexpr bool KEEP_V = (E <= 4);
    f2 vt[KEEP_V ? E : 1];
#pragma unroll
    for (int e = 0; e < E; e++) {
        if constexpr (KEEP_V) { const float2 t = a.target[L::row_index(lane, e) * L::D + col]; vt[e] = f2{t.x, t.y}; }
    }
    const int dst_lane = reduce8_dst_lane(lane);

    // ---- owner state.  Owner slot o = tid + THREADS * s; the K CP angles come first.
    float th[PPT], mm[PPT], vv[PPT], psl[PPT], pval[PPT];
    int cs_off[PPT], g_off[PPT], pj[PPT];
    bool valid[PPT], is_cp[PPT], frz[PPT], shr[PPT];
    penalty_table_init(smem, a, tid);
#pragma unroll
    for (int s = 0; s < PPT; s++) {
        const int o = tid + THREADS * s;
        valid[s] = o < P;
        shr[s] = false;
        int j = 0;
        if (valid[s]) {
            if constexpr (ENT != 0) j = o;                       // no CP angles: slots in parameter order
            else if (o < K) j = NSP + NB * o + NB - 1;
            else if (o - K < NSP) j = o - K;
            else { const int q = o - K - NSP; j = NSP + NB * (q / (NB - 1)) + q % (NB - 1); }
        }
        pj[s] = j;
        is_cp[s] = valid[s] && a.pkind[j];
        frz[s] = valid[s] && frozen && frozen[j];
        if (j < NSP) {                                       // surface gate 3 k + g sits on kernel qubit k; pgate finds it (qubits may be relabelled)
            const int sj = valid[s] ? a.pgate[j] : 0;
            cs_off[s] = Lds::SURF_CS + 2 * sj; g_off[s] = 32 * (sj / 8) + 4 * (sj % 8);
        }
        else {
            const int b = (j - NSP) / NBP, k = (j - NSP) % NBP;
            cs_off[s] = Lds::BLK_CS + 16 * b + 2 * k; g_off[s] = 32 * Lds::NBATCH + 32 * b + 4 * k;
            // x / y rotation on a register qubit: its record holds (t, s) (cpf_device.h rot_apply, SH)
            if (valid[s] && k < 2 * NR && rot_kind(ROT, k >> 1) != OP_RZ) shr[s] = CPF_SHEAR != 0 && !L::row_on_lane(PL::v[2 * (b % LL) + (k & 1)]);
        }
        th[s] = valid[s] ? init[j] : 0.f;
        mm[s] = 0.f; vv[s] = 0.f; psl[s] = 0.f; pval[s] = 0.f;
        if (valid[s]) { a.out_params[(s_idx * 2) * P + j] = th[s]; out_best[j] = th[s]; }
    }
    for (int b = tid; b < K; b += THREADS) {
        smem[Lds::BLK_CS + 16 * b + 14] = 0.f; smem[Lds::BLK_CS + 16 * b + 15] = 0.f;
        if constexpr (ENT != 0) { smem[Lds::BLK_CS + 16 * b + 2 * (NB - 1)] = -1.f; smem[Lds::BLK_CS + 16 * b + 2 * (NB - 1) + 1] = 0.f; }   // CZ = CP(pi), exactly (unused by CX)
    }
    for (int i = NSP + tid; i < NS; i += THREADS) { smem[Lds::SURF_CS + 2 * i] = 1.f; smem[Lds::SURF_CS + 2 * i + 1] = 0.f; }
    auto publish = [&](int s) __attribute__((always_inline)) {
        // linear penalty: start the two LDS table lookups first, so they are in flight during sincos
        const bool pen = a.use_reg && (K > THREADS * s) && is_cp[s];
        const bool lin = pen && a.pen_kind == 0;
        float x = 0.f;
        float2 si = make_float2(0.f, 0.f);
        if (lin) { x = floor_mod_2pi(th[s]); si = penalty_lookup(smem, x); }
        float sn, cn;
        fast_sincos(is_cp[s] ? th[s] : 0.5f * th[s], sn, cn);       // 3.4 % of config 2's kernel time
        if constexpr (L::E > 1) { if (shr[s]) shear_record(cn, sn); }
        if (valid[s]) *reinterpret_cast<float2*>(smem + cs_off[s]) = make_float2(cn, sn);
        if (lin) { pval[s] = __fadd_rn(__fmul_rn(si.x, x), si.y); psl[s] = a.r * si.x; }
        else if (pen) { float sl; pval[s] = penalty_lds(smem, a.pen_kind, th[s], sl); psl[s] = a.r * sl; }
    };
    lds_sync<WPS>();
#pragma unroll
    for (int s = 0; s < PPT; s++) publish(s);
    lds_sync<WPS>();

    float best = 0.f, first = 0.f;
    const int iters = a.T > 0 ? a.T : 1;
#ifdef CPF_STAMPS
    unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev = 0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev) :: "memory");
#endif
    for (int it = 0; it < iters; it++) {
        const float2 bc = const_load_f2(a.bias, a.T > 0 ? it : 0);     // scalar load; measured free (no change without it)
        CPF_STAMP(7)
        // ================= forward
#pragma unroll
        for (int e = 0; e < E; e++) st.p[e] = f2{(L::row_index(lane, e) == col) ? 1.f : 0.f, 0.f};
        {
            // two record buffers used alternately with a compile-time parity (no register copies): a body
            // covers UNR blocks = one layer, or two layers when the layer length is odd.  The first block
            // record is issued together with the surface record so both arrive under one LDS latency.
            float sr[Lds::SURF_F], ra[16], rb[16];
#pragma unroll
            for (int i = 0; i < Lds::SURF_F / 4; i++) {
                const float4 v = reinterpret_cast<const float4*>(smem + Lds::SURF_CS)[i];
                sr[4 * i] = v.x; sr[4 * i + 1] = v.y; sr[4 * i + 2] = v.z; sr[4 * i + 3] = v.w;
            }
            if constexpr (E < 16) { if (K > 0) load_rec(smem + Lds::BLK_CS, ra); }
            static_for([&](auto qq) __attribute__((always_inline)) {       // surface gate on qubit q: Rz(t2) Rx(t1) Rz(t0), main.py:122-124
                constexpr int Q = decltype(qq)::value;
                rot_apply<L, OP_RZ, Q>(st.p, sr[6 * Q], sr[6 * Q + 1], st);
                rot_apply<L, OP_RX, Q>(st.p, sr[6 * Q + 2], sr[6 * Q + 3], st);
                rot_apply<L, OP_RZ, Q>(st.p, sr[6 * Q + 4], sr[6 * Q + 5], st);
            }, typename make_iseq<NQ>::type{});
            int b = 0;
            float pz[4] = {1.f, 0.f, 1.f, 0.f};          // MERGE: (cos, sin) of the previous block's final Rz pair
            auto fwd_step = [&](auto jj) __attribute__((always_inline)) {
                constexpr int J = decltype(jj)::value, I = J % LL;
                if constexpr (MERGE) {
                    constexpr int IP = ((J + UNR - 1) % UNR) % LL;                 // pair of the previous block
                    float (&cur)[16] = (J & 1) ? rb : ra;
                    float (&nxt)[16] = (J & 1) ? ra : rb;
                    // next record first: an LDS read cannot be moved above the phase store by the compiler, and
                    // issued mid-block it returns after the next block wants it (same-box A/B on config 2: 10.19 -> 10.07 ms)
                    load_rec(smem + Lds::BLK_CS + 16 * (b + 1 < K ? b + 1 : b), nxt);
                    const f2 rho = diag_phase<L, PL::v[2 * IP], PL::v[2 * IP + 1], PL::v[2 * I], PL::v[2 * I + 1]>(
                        st, pz[0], pz[1], pz[2], pz[3], cur[2 * (NB - 1)], cur[2 * (NB - 1) + 1]);
                    if constexpr (STORE) ph[64 * b] = rho;
                    phase_apply<L>(st.p[0], rho, false);
                    lay_block_fwd_inner<L, PL::v[2 * I], PL::v[2 * I + 1], ROT>(st, cur);
#pragma unroll
                    for (int i = 0; i < 4; i++) pz[i] = cur[4 * (NR - 1) + i];
                } else if constexpr (E >= 16) {            // long blocks: no prefetch, keep registers
                    load_rec(smem + Lds::BLK_CS + 16 * b, ra);
                    lay_block_fwd<L, PL::v[2 * I], PL::v[2 * I + 1], ROT>(st, ra);
                } else if constexpr ((J & 1) == 0) {
                    load_rec(smem + Lds::BLK_CS + 16 * (b + 1 < K ? b + 1 : b), rb);      // prefetch next record
                    lay_block_fwd<L, PL::v[2 * I], PL::v[2 * I + 1], ROT>(st, ra);
                } else {
                    load_rec(smem + Lds::BLK_CS + 16 * (b + 1 < K ? b + 1 : b), ra);
                    lay_block_fwd<L, PL::v[2 * I], PL::v[2 * I + 1], ROT>(st, rb);
                }
                b++;
            };
            for (int full = K / UNR; full > 0; full--)             // whole bodies: straight-line, no guards
                static_for(fwd_step, typename make_iseq<UNR>::type{});
            static_for([&](auto jj) __attribute__((always_inline)) { if (b < K) fwd_step(jj); },   // tail: K % UNR blocks
                       typename make_iseq<UNR>::type{});
            if constexpr (MERGE) {                   // the last block's Rz pair, not followed by a CP
                if (K > 0) {
                    const int jt = (K - 1) % UNR;
                    static_for([&](auto jj) __attribute__((always_inline)) {
                        constexpr int J = decltype(jj)::value, I = J % LL;
                        if (J == jt) {
                            const f2 rho = diag_phase<L, PL::v[2 * I], PL::v[2 * I + 1], PL::v[2 * I], PL::v[2 * I + 1]>(
                                st, pz[0], pz[1], pz[2], pz[3], 1.f, 0.f);
                            if constexpr (STORE) ph[64 * K] = rho;
                            phase_apply<L>(st.p[0], rho, false);
                        }
                    }, typename make_iseq<UNR>::type{});
                }
            }
        }
        CPF_STAMP(1)                                  // forward blocks done
        // ================= loss, penalty, best tracking
        float acc[3] = {0.f, 0.f, 0.f};
        {
            f2 tt = {0.f, 0.f};
#pragma unroll
            for (int e = 0; e < E; e++) {
                f2 t;
                if constexpr (KEEP_V) t = vt[e]; else { const float2 g = a.target[L::row_index(lane, e) * L::D + col]; t = f2{g.x, g.y}; }
                if (a.loss_kind == 2) {               // relative phase: sum |V|^2 |psi|^2
                    acc[0] = fmaf(fmaf(t.x, t.x, t.y * t.y), fmaf(st.p[e].x, st.p[e].x, st.p[e].y * st.p[e].y), acc[0]);
                } else {                              // T = sum psi conj(V)
                    tt = pk_fma(splat(t.x), st.p[e], tt);
                    tt = pk_fma(f2{t.y, -t.y}, swp(st.p[e]), tt);
                }
            }
            if (a.loss_kind != 2) { acc[0] = tt.x; acc[1] = tt.y; }
        }
        if (a.use_reg) {
#pragma unroll
            for (int s = 0; s < PPT; s++) acc[2] += pval[s];
        }
        block_sum<L, 3>(acc, red, wave, lane);
        const float tr = acc[0], ti = acc[1];
        const float reg = a.use_reg ? a.r * acc[2] : 0.f;
        const float ab = sqrtf(fmaf(tr, tr, ti * ti));
        const float loss = (a.loss_kind == 2) ? 1.f - tr * a.inv_dim2 : 1.f - ab * ab * a.inv_dim2;
        const float regloss = loss + reg;
        bool improved = false;
        if (it == 0) { first = regloss; best = regloss; }
        else if (regloss < best) { best = regloss; improved = true; }
        if (improved) {
#pragma unroll
            for (int s = 0; s < PPT; s++) if (valid[s]) out_best[pj[s]] = th[s];
        }
        if (a.hist_params) {
#pragma unroll
            for (int s = 0; s < PPT; s++) if (valid[s]) a.hist_params[(s_idx * a.T + it) * P + pj[s]] = th[s];
            if (tid == 0) a.hist_loss[s_idx * a.T + it] = regloss;
        }
        if (a.T == 0) break;

        CPF_STAMP(2)                                  // loss phase done
        // ================= backward sweep
        {
            const float fr = -a.inv_dim2 * tr, fi = -a.inv_dim2 * ti;
#pragma unroll
            for (int e = 0; e < E; e++) {
                f2 t;
                if constexpr (KEEP_V) t = vt[e]; else { const float2 g = a.target[L::row_index(lane, e) * L::D + col]; t = f2{g.x, g.y}; }
                if (a.loss_kind == 2) {
                    const float w = -a.inv_dim2 * fmaf(t.x, t.x, t.y * t.y);
                    st.c[e] = splat(w) * st.p[e];
                } else {
                    st.c[e] = cmul(t, f2{fr, fi});
                }
            }
        }
        float* gslab = smem + Lds::g_surf(K, wave);
        if (K > 0) {
            float ra[16], rb[16], d[8];
            int b = K - 1;
            float dzu = 0.f, dzd = 0.f;              // MERGE: gradients of block b's final Rz pair, found one unit earlier
            if constexpr (E < 16) load_rec(smem + Lds::BLK_CS + 16 * b, ra);      // record of block K-1
            if constexpr (MERGE) {
                const int jt = (K - 1) % UNR;
                f2 rho = {1.f, 0.f};
                if constexpr (STORE) rho = ph[64 * K];
                static_for([&](auto jj) __attribute__((always_inline)) {
                    constexpr int J = decltype(jj)::value, I = J % LL;
                    if (J == jt) {
                        if constexpr (!STORE)
                            rho = diag_phase<L, PL::v[2 * I], PL::v[2 * I + 1], PL::v[2 * I], PL::v[2 * I + 1]>(
                                st, ra[4 * (NR - 1)], ra[4 * (NR - 1) + 1], ra[4 * (NR - 1) + 2], ra[4 * (NR - 1) + 3], 1.f, 0.f);
                        const float w = fmaf(st.c[0].x, st.p[0].y, -st.c[0].y * st.p[0].x);
                        dzu = -st.sg[PL::v[2 * I]] * w;
                        dzd = -st.sg[PL::v[2 * I + 1]] * w;
                    }
                }, typename make_iseq<UNR>::type{});
                phase_apply<L>(st.p[0], rho, true);
                phase_apply<L>(st.c[0], rho, true);
            }
            // one block of the backward sweep; CUR = its record
            auto bwd_block = [&](auto ii, auto ip, float (&cur)[16], float (&prv)[16]) __attribute__((always_inline)) {
                constexpr int I = decltype(ii)::value, IP = decltype(ip)::value;
                if constexpr (MERGE) {
                    f2 rho;
                    if constexpr (STORE) rho = ph[64 * b];
#pragma unroll
                    for (int k = 0; k < 8; k++) d[k] = 0.f;
                    lay_block_bwd_inner<L, PL::v[2 * I], PL::v[2 * I + 1], ROT>(st, cur, d);
                    if constexpr (!STORE) {            // rebuild the run's phase: Rz pair of block b-1 (record prv) and CP(b)
                        const bool has_prev = b > 0;
                        rho = diag_phase<L, PL::v[2 * IP], PL::v[2 * IP + 1], PL::v[2 * I], PL::v[2 * I + 1]>(
                            st, has_prev ? prv[4 * (NR - 1)] : 1.f, has_prev ? prv[4 * (NR - 1) + 1] : 0.f,
                            has_prev ? prv[4 * (NR - 1) + 2] : 1.f, has_prev ? prv[4 * (NR - 1) + 3] : 0.f,
                            cur[2 * (NB - 1)], cur[2 * (NB - 1) + 1]);
                    }
                    d[2 * (NR - 1)] = dzu; d[2 * (NR - 1) + 1] = dzd;
                    const float w = fmaf(st.c[0].x, st.p[0].y, -st.c[0].y * st.p[0].x);     // shared by CP(b), Rz(up), Rz(dn) of block b-1
                    constexpr int Q0 = PL::v[2 * I] < PL::v[2 * I + 1] ? PL::v[2 * I] : PL::v[2 * I + 1];
                    constexpr int Q1 = PL::v[2 * I] < PL::v[2 * I + 1] ? PL::v[2 * I + 1] : PL::v[2 * I];
                    d[NB - 1] = st.cpm2[Q0 + Q1 - 1] * w;
                    dzu = -st.sg[PL::v[2 * IP]] * w;
                    dzd = -st.sg[PL::v[2 * IP + 1]] * w;
                    phase_apply<L>(st.p[0], rho, true);
                    phase_apply<L>(st.c[0], rho, true);
                } else {
                    lay_block_bwd<L, PL::v[2 * I], PL::v[2 * I + 1], ROT>(st, cur, d);
                }
                reduce8_rows<true>(d, gslab + 32 * Lds::NBATCH + 32 * b, lane, dst_lane);
                gate_fence<L>();
                b--;
            };
            // blocks K-1 .. 0.  First the K % UNR blocks beyond the last whole body (no prefetch), then whole
            // bodies of UNR blocks in reverse with compile-time pair index and buffer parity.
            const int tail = K % UNR;
            static_for([&](auto jj) __attribute__((always_inline)) {
                constexpr int J = UNR - 1 - decltype(jj)::value, I = J % LL, IP = ((J + UNR - 1) % UNR) % LL;
                if (J < tail) {
                    if constexpr (MERGE) {            // (tail blocks: plain loads, record b in ra, record b-1 in rb)
                        load_rec(smem + Lds::BLK_CS + 16 * b, ra);
                        load_rec(smem + Lds::BLK_CS + 16 * (b > 0 ? b - 1 : 0), rb);
                        bwd_block(ic<I>{}, ic<IP>{}, ra, rb);
                    } else {
                        load_rec(smem + Lds::BLK_CS + 16 * b, ra);
                        bwd_block(ic<I>{}, ic<IP>{}, ra, ra);
                    }
                }
            }, typename make_iseq<UNR>::type{});
            if constexpr (E < 16) { if (b >= 0 && tail > 0) load_rec(smem + Lds::BLK_CS + 16 * b, ra); }
            auto bwd_step = [&](auto jj) __attribute__((always_inline)) {
                constexpr int JJ = decltype(jj)::value, J = UNR - 1 - JJ, I = J % LL, IP = ((J + UNR - 1) % UNR) % LL;
                if constexpr (E >= 16) {
                    load_rec(smem + Lds::BLK_CS + 16 * b, ra);
                    bwd_block(ic<I>{}, ic<IP>{}, ra, ra);
                } else if constexpr ((JJ & 1) == 0) {
                    load_rec(smem + Lds::BLK_CS + 16 * (b > 0 ? b - 1 : 0), rb);
                    bwd_block(ic<I>{}, ic<IP>{}, ra, rb);
                } else {
                    load_rec(smem + Lds::BLK_CS + 16 * (b > 0 ? b - 1 : 0), ra);
                    bwd_block(ic<I>{}, ic<IP>{}, rb, ra);
                }
            };
            for (int full = K / UNR; full > 0; full--)
                static_for(bwd_step, typename make_iseq<UNR>::type{});
        }
        CPF_STAMP(3)                                  // backward blocks done
        {
            float sr[Lds::SURF_F], ds[8 * Lds::NBATCH];
#pragma unroll
            for (int i = 0; i < Lds::SURF_F / 4; i++) {
                const float4 v = reinterpret_cast<const float4*>(smem + Lds::SURF_CS)[i];
                sr[4 * i] = v.x; sr[4 * i + 1] = v.y; sr[4 * i + 2] = v.z; sr[4 * i + 3] = v.w;
            }
#pragma unroll
            for (int k = 0; k < 8 * Lds::NBATCH; k++) ds[k] = 0.f;
            if constexpr (E <= 4) {
                surface_bwd_independent<L, NQ>(st, sr, ds);
            } else {
                static_for([&](auto qq) __attribute__((always_inline)) {
                    constexpr int Q = NQ - 1 - decltype(qq)::value;
                    ds[3 * Q + 2] = lay_rot_bwd<L, OP_RZ, Q>(st, sr[6 * Q + 4], sr[6 * Q + 5]);
                    ds[3 * Q + 1] = lay_rot_bwd<L, OP_RX, Q>(st, sr[6 * Q + 2], sr[6 * Q + 3]);
                    ds[3 * Q + 0] = lay_rot_bwd<L, OP_RZ, Q>(st, sr[6 * Q], sr[6 * Q + 1]);
                }, typename make_iseq<NQ>::type{});
            }
#pragma unroll
            for (int bt = 0; bt < Lds::NBATCH; bt++) {
                float d8[8];
#pragma unroll
                for (int k = 0; k < 8; k++) d8[k] = ds[8 * bt + k];
                reduce8_rows(d8, gslab + 32 * bt, lane, dst_lane);
            }
        }
        lds_sync<WPS>();

        CPF_STAMP(4)                                  // backward surface gates done
        // ================= Adam (optax 0.1.1) + next step's (cos, sin) / penalty, owner threads
        const float ibc1 = __builtin_amdgcn_rcpf(bc.x), ibc2 = __builtin_amdgcn_rcpf(bc.y);
#pragma unroll
        for (int s = 0; s < PPT; s++) {
            float g = 0.f;
#pragma unroll
            for (int w = 0; w < WPS; w++) {
                const float4 gp = *reinterpret_cast<const float4*>(smem + Lds::g_surf(K, w) + g_off[s]);
                g += (gp.x + gp.y) + (gp.z + gp.w);
            }
            if (a.use_reg && is_cp[s]) g += psl[s];
            if (valid[s] && !frz[s]) {
                const float m = fmaf(a.b1, mm[s], a.omb1 * g);
                const float v = fmaf(a.b2, vv[s], a.omb2 * (g * g));
                mm[s] = m; vv[s] = v;
                th[s] = fmaf(-a.lr, (m * ibc1) * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v * ibc2) + a.eps), th[s]);
            }
            publish(s);
        }
        lds_sync<WPS>();
        CPF_STAMP(5)                                  // owner phase done
    }

#ifdef CPF_STAMPS
    if (s_idx == 0 && tid == 0 && a.hist_loss) for (int k = 0; k < 8; k++) a.hist_loss[k] = (float)stamp_sum[k];
#endif
    // ================= epilogue (optimization.py:365-370, cp_utils.py:144-164)
    const int bi = (best < first) ? 1 : 0;
    float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < PPT; s++) {
        if (is_cp[s]) {
            const int j = pj[s];
            const float t0 = a.out_params[(s_idx * 2) * P + j], t1 = out_best[j];
            float sl;
            if (a.use_reg) { acc[0] += penalty_lds(smem, a.pen_kind, t0, sl); acc[1] += penalty_lds(smem, a.pen_kind, t1, sl); }
            acc[2] += cz_value(bi ? t1 : t0, a.thr_cp);
        }
    }
    block_sum<L, 3>(acc, red, wave, lane);
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

template <int NQ, int LW, int ROT, int PPT, int MSTORE, int... UD>
__global__ __launch_bounds__(LW == 0 ? 256 : 64 << LW, 1) void k_cp_layered(const KArgs a) {
    layered_body<NQ, LW, ROT, PPT, MSTORE, UD...>(a);
}

}  // namespace cpf
