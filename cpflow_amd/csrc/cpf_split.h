// cpflow_amd: the two-ended 3-qubit kernel — two waves per sample, each owning half of the circuit.
//
// With S <= 4 x CUs samples every SIMD holds a single wave of the one-wave kernel (cpf_layered.h), and a step is
// one dependent chain of ~3000 issue slots with nothing to overlap its LDS / DPP / transcendental latencies.
// The chain is cut in two.  Write U = H2 H1 with H1 = B_{m-1} .. B_0 S (surface gates and the first m blocks) and
// H2 = B_{K-1} .. B_m.  Since  tr(V^+ H2 H1) = sum conj(W) o H1  with  W = H2^+ V :
//
//   forward    wave A builds psi = H1 from the identity      |  wave B builds W by un-applying blocks K-1 .. m from V
//   exchange   psi and W through LDS (512 B each way); both waves form T, the loss and f = -T / N^2
//   backward   wave A: C = f W, sweeps blocks m-1 .. 0 and    |  wave B: psi = H1, C = f W, sweeps blocks m .. K-1
//              the surface gates backwards (un-applying)      |  FORWARDS (applying each gate to psi and C, then
//                                                             |  reading its gradient: the same psi_j, C_j pair)
//   owner      128 threads own the angles (Adam state in registers), one round for P <= 128
//
// Same arithmetic per gate as the one-wave kernel (merged diagonal runs, DPP rotations, reduce8 gradient
// scatter); the differences in rounding are those of evaluating the same products in another order
// (|d loss| <= 1e-6, tests/test_gpu_kernels.py::test_split_*).  Requires: 3 qubits, rotation string ending in
// 'z', loss kinds with a linear cotangent (HST, state preparation), K <= 64, m a multiple of the layer length.
#pragma once
#include "cpf_layered.h"

namespace cpf {

// number of blocks given to wave A: balances (surface ~ 1.5 blocks) + m against K - m, multiple of the layer length
#ifndef CPF_CUT_EXTRA
#define CPF_CUT_EXTRA 0          // A/B builds: whole layers moved from wave B's half to wave A's
#endif
// Since wave B owns every angle (its owner phase adds ~90 instructions to its step, wave A's has none) the balance point
// sits later: same-box A/B on config 2 (K = 12, chain) 6.33 / 6.16 / 6.21 / 6.45 ms for 6 / 7 / 8 / 9 blocks in wave A; on
// config 3's shape (K = 16, all-to-all, layers of 3) 6.87 ms at 6 blocks against 7.04 at 7 and 6.98 at 9.  Hence one more
// layer for wave A where a layer is at most two blocks.
__host__ __device__ constexpr int split_point(int K, int LL) {
    int m = ((2 * K - 3 + 2 * LL) / (4 * LL)) * LL + CPF_CUT_EXTRA * LL;          // LL * round((K - 1.5) / 2 / LL)
    if (LL <= 2 && K >= 4 * LL) m += LL;
    if (m < LL) m = LL;
    const int top = ((K - 1) / LL) * LL;
    if (m > top) m = top;
    return m < 0 ? 0 : m;
}
__host__ __device__ constexpr bool split_supported(int K, int LL) { return K >= 2 * LL && K >= 2 && K <= 64; }
// the cut for launches that leave at most one wave on a SIMD (KArgs::lone): the step is then the longer wave's chain, so
// the two waves get the same number of blocks, rounded DOWN to a layer boundary (wave A also has the surface gates).  K from
// the plan, same box: config 2's shape 7.61 -> 6.67 ms at 100 and 256 samples (6 of 12 instead of 8); config 3's shape keeps
// 6 of 16 (9 measured 8.60 against 7.81 ms)
__host__ __device__ constexpr int lone_point(int K, int LL) {
    int m = (K / (2 * LL)) * LL;
    if (m < LL) m = LL;
    const int top = ((K - 1) / LL) * LL;
    return m > top ? top : m;
}
// With K a compile-time constant the cut need not fall on a layer boundary (wave B's pair indices carry the offset
// m mod LL).  Instruction counts of the packed stored-state sweeps: wave A = 160 (surface gates, both directions)
// + 110 per block, wave B = 109 per block + 90 (loss, penalty, best tracking)  ->  m = (109 K - 70) / 219 rounded:
// 6 of 12 (the layer cut), 8 of 16.
__host__ __device__ constexpr int split_point_any(int K) {
    int m = (2 * (109 * K - 70) + 219) / (2 * 219);
    if (m < 1) m = 1;
    if (m > K - 1) m = K - 1;
    return m;
}

struct SplitLds {                                          // float offsets; [0, 28) = penalty table
    using B = LayeredLds<3>;
    static constexpr int SURF_F = B::SURF_F, NBATCH = B::NBATCH, SURF_CS = B::SURF_CS, BLK_CS = B::BLK_CS;
    __host__ __device__ static constexpr int slab(int K) { return BLK_CS + 16 * K; }              // surface batches, then 32 per block
    __host__ __device__ static constexpr int red(int K) { return slab(K) + 32 * (NBATCH + K); }    // [0] penalty sum, [1] "best so far" flag
    __host__ __device__ static constexpr int ident(int K) { return red(K) + 4; }                   // record of identity gates ("block -1")
    __host__ __device__ static constexpr int xch(int K) { return ident(K) + 16; }                  // 2 x 64 x (re, im)
    __host__ __device__ static constexpr int phase(int K) { return xch(K) + 256; }                 // (K + 2) units x 64 lanes x (re, im)
    // stored states (MSTORE = 2): 64 lanes x (re, im) per slot.  Wave A: slot 0 = psi after the surface gates, slot
    // 1 + NI b + g = psi after inner gate g of block b (NI = inner gates per block); wave B likewise for W with its
    // slot 0 unused (W at the cut stays in registers): NI K + 2 slots in all.
    __host__ __device__ static constexpr int states(int K) { return phase(K) + 128 * (K + 2); }
    __host__ __device__ static constexpr int total(int K, int mstore, int ni = 0) {
        return mstore == 2 ? states(K) + 128 * (ni * K + 2) : phase(K) + (mstore ? 128 * (K + 2) : 0);
    }
};

template <int LL>
struct SplitUnroll {
    static constexpr int UNR2 = (LL % 2 == 0) ? LL : 2 * LL;
    static constexpr int UNR = (UNR2 % 3 == 0) ? UNR2 : 3 * UNR2;      // even, whole layers, multiple of 6 (see cpf_layered.h)
};

// inner gate g of a block (application order: letter 0 on up, letter 0 on dn, letter 1 on up, ...; the final
// letter's pair belongs to the merged diagonal run): kind, qubit and the (cos, sin) slot of the block record
template <int ROT, int G> struct InnerGate {
    static constexpr int LETTER = G / 2, KIND = rot_kind(ROT, LETTER), CS = 4 * LETTER + 2 * (G & 1);
    template <int UP, int DN> static constexpr int qubit() { return (G & 1) ? DN : UP; }
};
template <int ROT> __host__ __device__ constexpr int inner_gates() { return 2 * (rot_nr(ROT) - 1); }

// forward inner rotations of a block on psi, psi stored after each gate: slot[64 g]
template <class L, int UP, int DN, int ROT>
__device__ __forceinline__ void lay_block_fwd_inner_store(State<L>& st, const float (&r)[16], f2* slot) {
    static_for([&](auto gg) __attribute__((always_inline)) {
        constexpr int G = decltype(gg)::value;
        using IG = InnerGate<ROT, G>;
        rot_apply<L, IG::KIND, IG::template qubit<UP, DN>()>(st.p, r[IG::CS], r[IG::CS + 1], st);
        slot[64 * G] = st.p[0];
    }, typename make_iseq<inner_gates<ROT>()>::type{});
}
// inverse inner rotations of a block on psi, psi stored BEFORE each un-application (= the state after that gate)
template <class L, int UP, int DN, int ROT>
__device__ __forceinline__ void lay_block_inv_inner_store(State<L>& st, const float (&r)[16], f2* slot) {
    static_for([&](auto gg) __attribute__((always_inline)) {
        constexpr int G = inner_gates<ROT>() - 1 - decltype(gg)::value;
        using IG = InnerGate<ROT, G>;
        slot[64 * G] = st.p[0];
        rot_apply<L, IG::KIND, IG::template qubit<UP, DN>()>(st.p, r[IG::CS], -r[IG::CS + 1], st);
    }, typename make_iseq<inner_gates<ROT>()>::type{});
}

// ---- wave A forward: blocks [0, n) applied to psi; units (merged diagonal runs) 0 .. n stored at ph[64 u]
template <class L, class PL, int LL, int ROT, bool STORE, bool PS = false>
__device__ __forceinline__ void split_fwd(State<L>& st, const float* rec, int n, f2* ph, f2* sp = nullptr) {
    constexpr int UNR = SplitUnroll<LL>::UNR, NR = rot_nr(ROT), NB = 2 * NR + 1, NI = 2 * (NR - 1);
    float ra[16], rb[16];
    float pz[4] = {1.f, 0.f, 1.f, 0.f};
    int b = 0;
    if (n > 0) load_rec(rec, ra);
    auto step = [&](auto jj) __attribute__((always_inline)) {
        constexpr int J = decltype(jj)::value, I = J % LL, IP = ((J + UNR - 1) % UNR) % LL;
        float (&cur)[16] = (J & 1) ? rb : ra;
        float (&nxt)[16] = (J & 1) ? ra : rb;
        load_rec(rec + 16 * (b + 1 < n ? b + 1 : b), nxt);
        const f2 rho = diag_phase<L, PL::v[2 * IP], PL::v[2 * IP + 1], PL::v[2 * I], PL::v[2 * I + 1]>(
            st, pz[0], pz[1], pz[2], pz[3], cur[2 * (NB - 1)], cur[2 * (NB - 1) + 1]);
        if constexpr (STORE) ph[64 * b] = rho;
        phase_apply<L>(st.p[0], rho, false);
        if constexpr (PS) lay_block_fwd_inner_store<L, PL::v[2 * I], PL::v[2 * I + 1], ROT>(st, cur, sp + 64 * (NI * b + 1));
        else lay_block_fwd_inner<L, PL::v[2 * I], PL::v[2 * I + 1], ROT>(st, cur);
#pragma unroll
        for (int i = 0; i < 4; i++) pz[i] = cur[4 * (NR - 1) + i];
        b++;
    };
    for (int full = n / UNR; full > 0; full--) static_for(step, typename make_iseq<UNR>::type{});
    static_for([&](auto jj) __attribute__((always_inline)) { if (b < n) step(jj); }, typename make_iseq<UNR>::type{});
    if (n > 0) {
        const int jt = (n - 1) % UNR;
        static_for([&](auto jj) __attribute__((always_inline)) {
            constexpr int J = decltype(jj)::value, I = J % LL;
            if (J == jt) {
                const f2 rho = diag_phase<L, PL::v[2 * I], PL::v[2 * I + 1], PL::v[2 * I], PL::v[2 * I + 1]>(
                    st, pz[0], pz[1], pz[2], pz[3], 1.f, 0.f);
                if constexpr (STORE) ph[64 * n] = rho;
                phase_apply<L>(st.p[0], rho, false);
            }
        }, typename make_iseq<UNR>::type{});
    }
}

// the inner rotations of a block, inverted, on psi only
template <class L, int UP, int DN, int ROT>
__device__ __forceinline__ void lay_block_inv_inner(State<L>& st, const float (&r)[16]) {
    constexpr int NR = rot_nr(ROT);
#pragma unroll
    for (int i = NR - 2; i >= 0; i--) {
        if (rot_kind(ROT, i) == OP_RX) {
            rot_apply<L, OP_RX, DN>(st.p, r[4 * i + 2], -r[4 * i + 3], st); rot_apply<L, OP_RX, UP>(st.p, r[4 * i], -r[4 * i + 1], st);
        } else if (rot_kind(ROT, i) == OP_RY) {
            rot_apply<L, OP_RY, DN>(st.p, r[4 * i + 2], -r[4 * i + 3], st); rot_apply<L, OP_RY, UP>(st.p, r[4 * i], -r[4 * i + 1], st);
        } else {
            rot_apply<L, OP_RZ, DN>(st.p, r[4 * i + 2], -r[4 * i + 3], st); rot_apply<L, OP_RZ, UP>(st.p, r[4 * i], -r[4 * i + 1], st);
        }
    }
}

// ---- wave B forward: blocks n-1 .. 0 un-applied from psi (= V on entry, W = H2^+ V on exit); n >= 1.
// Two record buffers: block b (J = b mod UNR, UNR even) reads its own record from the buffer of J's parity and fetches
// the previous block's record - its own successor's - into the other one at the START of the block, so that every record
// is in flight for a whole block before it is used (also in the leading partial body: with K a compile-time constant
// and fewer blocks than one body, that is the whole sweep).
template <class L, class PL, int LL, int ROT, bool STORE, bool PS = false, int OFF = 0>
__device__ __forceinline__ void split_inv(State<L>& st, const float* rec, const float* ident, int n, f2* ph, f2* sp = nullptr) {
    constexpr int UNR = SplitUnroll<LL>::UNR, NR = rot_nr(ROT), NB = 2 * NR + 1, NI = 2 * (NR - 1);
    static_assert(UNR % 2 == 0, "record buffers alternate with the block index");
    float ra[16], rb[16];
    int b = n - 1;
    {
        const int jt = (n - 1) % UNR;
        static_for([&](auto jj) __attribute__((always_inline)) {
            constexpr int J = decltype(jj)::value, I = (J + OFF) % LL;
            if (J == jt) {
                float (&cur)[16] = (J & 1) ? ra : rb;
                load_rec(rec + 16 * b, cur);
                const f2 rho = diag_phase<L, PL::v[2 * I], PL::v[2 * I + 1], PL::v[2 * I], PL::v[2 * I + 1]>(
                    st, cur[4 * (NR - 1)], cur[4 * (NR - 1) + 1], cur[4 * (NR - 1) + 2], cur[4 * (NR - 1) + 3], 1.f, 0.f);
                if constexpr (STORE) ph[64 * n] = rho;
                phase_apply<L>(st.p[0], rho, true);
            }
        }, typename make_iseq<UNR>::type{});
    }
    auto step = [&](auto jj) __attribute__((always_inline)) {
        constexpr int J = UNR - 1 - decltype(jj)::value, I = (J + OFF) % LL, IP = (J + OFF + UNR - 1) % LL;
        float (&cur)[16] = (J & 1) ? ra : rb;
        float (&prv)[16] = (J & 1) ? rb : ra;
        load_rec(b > 0 ? rec + 16 * (b - 1) : ident, prv);
        if constexpr (PS) lay_block_inv_inner_store<L, PL::v[2 * I], PL::v[2 * I + 1], ROT>(st, cur, sp + 64 * (NI * b + 1));
        else lay_block_inv_inner<L, PL::v[2 * I], PL::v[2 * I + 1], ROT>(st, cur);
        const f2 rho = diag_phase<L, PL::v[2 * IP], PL::v[2 * IP + 1], PL::v[2 * I], PL::v[2 * I + 1]>(
            st, prv[4 * (NR - 1)], prv[4 * (NR - 1) + 1], prv[4 * (NR - 1) + 2], prv[4 * (NR - 1) + 3],
            cur[2 * (NB - 1)], cur[2 * (NB - 1) + 1]);
        if constexpr (STORE) ph[64 * b] = rho;
        phase_apply<L>(st.p[0], rho, true);
        b--;
    };
    const int tail = n % UNR;
    static_for([&](auto jj) __attribute__((always_inline)) {       // leading partial body: blocks n-1 .. n-tail
        if (UNR - 1 - decltype(jj)::value < tail) step(jj);
    }, typename make_iseq<UNR>::type{});
    for (int full = n / UNR; full > 0; full--) static_for(step, typename make_iseq<UNR>::type{});
}

// ---- wave A backward: blocks n-1 .. 0 un-applied from psi and C, gradients to gblk[32 b] (as cpf_layered.h);
// block 0's eight partials are returned in d0 (not reduced)
template <class L, class PL, int LL, int ROT, bool STORE>
__device__ __forceinline__ void split_bwd_grad(State<L>& st, const float* rec, int n, const f2* ph, float* gblk,
                                               int lane, int dst_lane, float (&d0)[8]) {
    constexpr int UNR = SplitUnroll<LL>::UNR, NR = rot_nr(ROT), NB = 2 * NR + 1;
    if (n <= 0) return;
    float ra[16], rb[16], d[8];
    int b = n - 1;
    float dzu = 0.f, dzd = 0.f;
    load_rec(rec + 16 * b, ra);
    {
        const int jt = (n - 1) % UNR;
        f2 rho = {1.f, 0.f};
        if constexpr (STORE) rho = ph[64 * n];
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
    auto blk = [&](auto ii, auto ip, float (&cur)[16], float (&prv)[16]) __attribute__((always_inline)) {
        constexpr int I = decltype(ii)::value, IP = decltype(ip)::value;
        f2 rho;
        if constexpr (STORE) rho = ph[64 * b];
#pragma unroll
        for (int k = 0; k < 8; k++) d[k] = 0.f;
        lay_block_bwd_inner<L, PL::v[2 * I], PL::v[2 * I + 1], ROT>(st, cur, d);
        if constexpr (!STORE) {
            const bool has_prev = b > 0;
            rho = diag_phase<L, PL::v[2 * IP], PL::v[2 * IP + 1], PL::v[2 * I], PL::v[2 * I + 1]>(
                st, has_prev ? prv[4 * (NR - 1)] : 1.f, has_prev ? prv[4 * (NR - 1) + 1] : 0.f,
                has_prev ? prv[4 * (NR - 1) + 2] : 1.f, has_prev ? prv[4 * (NR - 1) + 3] : 0.f,
                cur[2 * (NB - 1)], cur[2 * (NB - 1) + 1]);
        }
        d[2 * (NR - 1)] = dzu; d[2 * (NR - 1) + 1] = dzd;
        const float w = fmaf(st.c[0].x, st.p[0].y, -st.c[0].y * st.p[0].x);
        constexpr int Q0 = PL::v[2 * I] < PL::v[2 * I + 1] ? PL::v[2 * I] : PL::v[2 * I + 1];
        constexpr int Q1 = PL::v[2 * I] < PL::v[2 * I + 1] ? PL::v[2 * I + 1] : PL::v[2 * I];
        d[NB - 1] = st.cpm2[Q0 + Q1 - 1] * w;
        dzu = -st.sg[PL::v[2 * IP]] * w;
        dzd = -st.sg[PL::v[2 * IP + 1]] * w;
        phase_apply<L>(st.p[0], rho, true);
        phase_apply<L>(st.c[0], rho, true);
        if (b > 0) reduce8_rows<true>(d, gblk + 32 * b, lane, dst_lane);
        else {                                            // block 0: the caller reduces it with the ninth surface gradient in the spare slot
#pragma unroll
            for (int k = 0; k < 8; k++) d0[k] = d[k];
        }
        b--;
    };
    const int tail = n % UNR;
    static_for([&](auto jj) __attribute__((always_inline)) {
        constexpr int J = UNR - 1 - decltype(jj)::value, I = J % LL, IP = ((J + UNR - 1) % UNR) % LL;
        if (J < tail) {
            load_rec(rec + 16 * b, ra);
            load_rec(rec + 16 * (b > 0 ? b - 1 : 0), rb);
            blk(ic<I>{}, ic<IP>{}, ra, rb);
        }
    }, typename make_iseq<UNR>::type{});
    if (b >= 0 && tail > 0) load_rec(rec + 16 * b, ra);
    auto step = [&](auto jj) __attribute__((always_inline)) {
        constexpr int JJ = decltype(jj)::value, J = UNR - 1 - JJ, I = J % LL, IP = ((J + UNR - 1) % UNR) % LL;
        if constexpr ((JJ & 1) == 0) {
            load_rec(rec + 16 * (b > 0 ? b - 1 : 0), rb);
            blk(ic<I>{}, ic<IP>{}, ra, rb);
        } else {
            load_rec(rec + 16 * (b > 0 ? b - 1 : 0), ra);
            blk(ic<I>{}, ic<IP>{}, rb, ra);
        }
    };
    for (int full = n / UNR; full > 0; full--) static_for(step, typename make_iseq<UNR>::type{});
}

// a rotation applied to psi and C, then its gradient (psi_j, C_j right after the gate)
template <class L, int KIND, int Q>
__device__ __forceinline__ float lay_rot_fwd_grad(State<L>& st, float c, float s) {
    rot_apply<L, KIND, Q>(st.p, c, s, st);
    rot_apply<L, KIND, Q>(st.c, c, s, st);
    return rot_grad<L, KIND, Q>(st);
}
template <class L, int UP, int DN, int ROT>
__device__ __forceinline__ void lay_block_fwd_grad_inner(State<L>& st, const float (&r)[16], float (&d)[8]) {
    constexpr int NR = rot_nr(ROT);
#pragma unroll
    for (int i = 0; i < NR - 1; i++) {
        if (rot_kind(ROT, i) == OP_RX) {
            d[2 * i] = lay_rot_fwd_grad<L, OP_RX, UP>(st, r[4 * i], r[4 * i + 1]); d[2 * i + 1] = lay_rot_fwd_grad<L, OP_RX, DN>(st, r[4 * i + 2], r[4 * i + 3]);
        } else if (rot_kind(ROT, i) == OP_RY) {
            d[2 * i] = lay_rot_fwd_grad<L, OP_RY, UP>(st, r[4 * i], r[4 * i + 1]); d[2 * i + 1] = lay_rot_fwd_grad<L, OP_RY, DN>(st, r[4 * i + 2], r[4 * i + 3]);
        } else {
            d[2 * i] = lay_rot_fwd_grad<L, OP_RZ, UP>(st, r[4 * i], r[4 * i + 1]); d[2 * i + 1] = lay_rot_fwd_grad<L, OP_RZ, DN>(st, r[4 * i + 2], r[4 * i + 3]);
        }
    }
}

// ---- wave B backward: blocks 0 .. n-1 APPLIED to psi and C (psi = H1, C = f W on entry), gradients to gblk[32 b].
// Unit b carries the CP of block b and the final Rz pair of block b-1, so block b-1's eight partials are complete
// — and scattered — at unit b.
template <class L, class PL, int LL, int ROT, bool STORE, int OFF = 0>
__device__ __forceinline__ void split_fwd_grad(State<L>& st, const float* rec, int n, const f2* ph, float* gblk,
                                               int lane, int dst_lane) {
    constexpr int UNR = SplitUnroll<LL>::UNR, NR = rot_nr(ROT), NB = 2 * NR + 1;
    float ra[16], rb[16], da[8], db[8];
    float pz[4] = {1.f, 0.f, 1.f, 0.f};
    int b = 0;
    load_rec(rec, ra);
#pragma unroll
    for (int k = 0; k < 8; k++) { da[k] = 0.f; db[k] = 0.f; }
    auto step = [&](auto jj) __attribute__((always_inline)) {
        constexpr int J = decltype(jj)::value, I = (J + OFF) % LL, IP = (J + OFF + UNR - 1) % LL;
        float (&cur)[16] = (J & 1) ? rb : ra;
        float (&nxt)[16] = (J & 1) ? ra : rb;
        float (&dcur)[8] = (J & 1) ? db : da;
        float (&dprv)[8] = (J & 1) ? da : db;
        load_rec(rec + 16 * (b + 1 < n ? b + 1 : b), nxt);
        f2 rho;
        if constexpr (STORE) rho = ph[64 * b];
        else rho = diag_phase<L, PL::v[2 * IP], PL::v[2 * IP + 1], PL::v[2 * I], PL::v[2 * I + 1]>(
                 st, pz[0], pz[1], pz[2], pz[3], cur[2 * (NB - 1)], cur[2 * (NB - 1) + 1]);
        const float w = fmaf(st.c[0].x, st.p[0].y, -st.c[0].y * st.p[0].x);      // invariant under the run's common phase
        phase_apply<L>(st.p[0], rho, false);
        phase_apply<L>(st.c[0], rho, false);
        dprv[2 * (NR - 1)] = -st.sg[PL::v[2 * IP]] * w;
        dprv[2 * (NR - 1) + 1] = -st.sg[PL::v[2 * IP + 1]] * w;
        if (b > 0) reduce8_rows<true>(dprv, gblk + 32 * (b - 1), lane, dst_lane);
        constexpr int Q0 = PL::v[2 * I] < PL::v[2 * I + 1] ? PL::v[2 * I] : PL::v[2 * I + 1];
        constexpr int Q1 = PL::v[2 * I] < PL::v[2 * I + 1] ? PL::v[2 * I + 1] : PL::v[2 * I];
#pragma unroll
        for (int k = 0; k < 8; k++) dcur[k] = 0.f;
        dcur[NB - 1] = st.cpm2[Q0 + Q1 - 1] * w;
        lay_block_fwd_grad_inner<L, PL::v[2 * I], PL::v[2 * I + 1], ROT>(st, cur, dcur);
        if constexpr (!STORE) {
#pragma unroll
            for (int i = 0; i < 4; i++) pz[i] = cur[4 * (NR - 1) + i];
        }
        b++;
    };
    for (int full = n / UNR; full > 0; full--) static_for(step, typename make_iseq<UNR>::type{});
    static_for([&](auto jj) __attribute__((always_inline)) { if (b < n) step(jj); }, typename make_iseq<UNR>::type{});
    {
        const int jt = (n - 1) % UNR;
        const float w = fmaf(st.c[0].x, st.p[0].y, -st.c[0].y * st.p[0].x);      // last unit: only its gradient is needed
        static_for([&](auto jj) __attribute__((always_inline)) {
            constexpr int J = decltype(jj)::value, I = (J + OFF) % LL;
            if (J == jt) {
                float (&dl)[8] = (J & 1) ? db : da;
                dl[2 * (NR - 1)] = -st.sg[PL::v[2 * I]] * w;
                dl[2 * (NR - 1) + 1] = -st.sg[PL::v[2 * I + 1]] * w;
                reduce8_rows<true>(dl, gblk + 32 * (n - 1), lane, dst_lane);
            }
        }, typename make_iseq<UNR>::type{});
    }
}

// ================= stored-state sweeps (MSTORE = 2) =====================================================
// The forward halves keep psi (wave A) / W (wave B) after every inner gate in LDS, so the gradient halves move ONE
// matrix per gate instead of two: wave A un-applies gates from C only and reads psi_j back; wave B applies gates to
// psi' = conj(f) H1 only and reads W_j back (Im(conj(f W) s psi) = Im(conj(W) s conj(f) psi): the scalar f moves
// onto psi once).  Per block 4 x (4 + 3) + 13 VALU operations instead of 4 x (8 + 3) + 15, for NI + 1 more
// ds_read_b64 / NI more ds_write_b64, prefetched one block ahead; no float32 drift from un-applying psi.
// One block's loads: its record, the state before its diagonal unit read at the lane's own slot (for the unit's
// gradients) and the state after each inner gate read at the slot of the lane's PARTNER on that gate's qubit, so
// that the gate's gradient is two plain VALU operations (no DPP move, no DPP wait states).  spx[q] = the state
// array seen from the partner lane on qubit q (spx[3] = own lane).
template <int NI>
struct BlkBuf { float r[16]; f2 own; f2 part[NI > 0 ? NI : 1]; f2 rho; };
template <class L>
struct StatePtrs { const f2* p[L::NQ + 1]; };
template <class L>
__device__ __forceinline__ StatePtrs<L> state_ptrs(const f2* sp_lane, int lane) {
    StatePtrs<L> r;
#pragma unroll
    for (int q = 0; q < L::NQ; q++) r.p[q] = sp_lane + ((lane ^ (1 << L::row_lane_bit(q))) - lane);
    r.p[L::NQ] = sp_lane;
    return r;
}
// FIRST = slot of the state before the block's unit; the inner gates' states follow it
template <class L, int ROT, int UP, int DN, int NI>
__device__ __forceinline__ void load_blk(BlkBuf<NI>& q, const float* rec, const StatePtrs<L>& sx, int first, const f2* ph) {
    load_rec(rec, q.r);
    q.own = sx.p[L::NQ][64 * first];
    static_for([&](auto gg) __attribute__((always_inline)) {
        constexpr int G = decltype(gg)::value;
        using IG = InnerGate<ROT, G>;
        constexpr int QP = IG::KIND == OP_RZ ? L::NQ : IG::template qubit<UP, DN>();
        q.part[G] = sx.p[QP][64 * (first + 1 + G)];
    }, typename make_iseq<NI>::type{});
    q.rho = *ph;
}
// this lane's share of a rotation's gradient from C (own lane) and psi read at the partner lane (Rz: own lane)
template <class L, int KIND, int Q>
__device__ __forceinline__ float grad_c_own(const State<L>& st, f2 c, f2 pp) {
    if constexpr (KIND == OP_RX) return fmaf(c.x, pp.y, -c.y * pp.x);
    else if constexpr (KIND == OP_RY) return st.sg[Q] * fmaf(c.x, pp.x, c.y * pp.y);
    else return -st.sg[Q] * fmaf(c.x, pp.y, -c.y * pp.x);
}
// the same sum over the wave with the roles swapped: psi at the own lane, C read at the partner lane (whose sign
// on that qubit is the opposite one)
template <class L, int KIND, int Q>
__device__ __forceinline__ float grad_p_own(const State<L>& st, f2 cp, f2 p) {
    if constexpr (KIND == OP_RX) return fmaf(cp.x, p.y, -cp.y * p.x);
    else if constexpr (KIND == OP_RY) return -st.sg[Q] * fmaf(cp.x, p.x, cp.y * p.y);
    else return -st.sg[Q] * fmaf(cp.x, p.y, -cp.y * p.x);
}

// ---- wave A backward: blocks n-1 .. 0 un-applied from C only.  Slot NI b + 1 + g = psi after inner gate g of
// block b, slot 0 = psi after the surface gates; on exit st.p holds that slot-0 state.
template <class L, class PL, int LL, int ROT>
__device__ __forceinline__ void split_bwd_grad_ps(State<L>& st, const float* rec, int n, const f2* ph, const f2* sp,
                                                  float* gblk, int lane, int dst_lane, float (&d0)[8]) {
    constexpr int UNR = SplitUnroll<LL>::UNR, NR = rot_nr(ROT), NB = 2 * NR + 1, NI = 2 * (NR - 1);
    static_assert(NI >= 2, "stored-state sweep needs at least one inner rotation letter");
    const StatePtrs<L> sx = state_ptrs<L>(sp, lane);
    BlkBuf<NI> qa, qb;
    float d[8];
    int b = n - 1;
    const f2 rho_n = ph[64 * n];
    float dzu = 0.f, dzd = 0.f;
    {
        const int jt = (n - 1) % UNR;
        const float w = fmaf(st.c[0].x, st.p[0].y, -st.c[0].y * st.p[0].x);   // psi, C after unit n (registers)
        static_for([&](auto jj) __attribute__((always_inline)) {
            constexpr int J = decltype(jj)::value, I = J % LL;
            if (J == jt) {
                if constexpr (J & 1) load_blk<L, ROT, PL::v[2 * I], PL::v[2 * I + 1]>(qb, rec + 16 * b, sx, NI * b, ph + 64 * b);
                else load_blk<L, ROT, PL::v[2 * I], PL::v[2 * I + 1]>(qa, rec + 16 * b, sx, NI * b, ph + 64 * b);
                dzu = -st.sg[PL::v[2 * I]] * w; dzd = -st.sg[PL::v[2 * I + 1]] * w;
            }
        }, typename make_iseq<UNR>::type{});
        phase_apply<L>(st.c[0], rho_n, true);
    }
    auto blk = [&](auto jj, BlkBuf<NI>& cur, BlkBuf<NI>& nxt) __attribute__((always_inline)) {
        constexpr int J = decltype(jj)::value, I = J % LL, IP = ((J + UNR - 1) % UNR) % LL;
        constexpr int UP = PL::v[2 * I], DN = PL::v[2 * I + 1];
        const int bn = b > 0 ? b - 1 : 0;
        load_blk<L, ROT, PL::v[2 * IP], PL::v[2 * IP + 1]>(nxt, rec + 16 * bn, sx, NI * bn, ph + 64 * bn);
#pragma unroll
        for (int k = 0; k < 8; k++) d[k] = 0.f;
        static_for([&](auto gg) __attribute__((always_inline)) {
            constexpr int G = NI - 1 - decltype(gg)::value;
            using IG = InnerGate<ROT, G>;
            constexpr int Q = IG::template qubit<UP, DN>();
            d[G] = grad_c_own<L, IG::KIND, Q>(st, st.c[0], cur.part[G]);
            rot_apply<L, IG::KIND, Q>(st.c, cur.r[IG::CS], -cur.r[IG::CS + 1], st);
        }, typename make_iseq<NI>::type{});
        phase_apply<L>(st.c[0], cur.rho, true);                  // C now sits before unit b, where cur.own was stored
        const float w = fmaf(st.c[0].x, cur.own.y, -st.c[0].y * cur.own.x);
        d[2 * (NR - 1)] = dzu; d[2 * (NR - 1) + 1] = dzd;
        constexpr int Q0 = UP < DN ? UP : DN, Q1 = UP < DN ? DN : UP;
        d[NB - 1] = st.cpm2[Q0 + Q1 - 1] * w;
        dzu = -st.sg[PL::v[2 * IP]] * w;
        dzd = -st.sg[PL::v[2 * IP + 1]] * w;
        if (b > 0) reduce8_rows<true>(d, gblk + 32 * b, lane, dst_lane);
        else {                                            // block 0: the caller reduces it with the ninth surface gradient in the spare slot
#pragma unroll
            for (int k = 0; k < 8; k++) d0[k] = d[k];
        }
        st.p[0] = cur.own;
        b--;
    };
    const int tail = n % UNR;
    static_for([&](auto jj) __attribute__((always_inline)) {
        constexpr int J = UNR - 1 - decltype(jj)::value;
        if (J < tail) { if constexpr (J & 1) blk(ic<J>{}, qb, qa); else blk(ic<J>{}, qa, qb); }
    }, typename make_iseq<UNR>::type{});
    for (int full = n / UNR; full > 0; full--)
        static_for([&](auto jj) __attribute__((always_inline)) {
            constexpr int J = UNR - 1 - decltype(jj)::value;
            if constexpr (J & 1) blk(ic<J>{}, qb, qa); else blk(ic<J>{}, qa, qb);
        }, typename make_iseq<UNR>::type{});
}

// ---- wave B backward: blocks 0 .. n-1 applied to psi' = conj(f) H1 only.  wcut = W at the cut;
// slot NI b + 1 + g = W after inner gate g of (local) block b.
template <class L, class PL, int LL, int ROT, int OFF = 0>
__device__ __forceinline__ void split_fwd_grad_ps(State<L>& st, f2 wcut, const float* rec, int n, const f2* ph,
                                                  const f2* sp, float* gblk, int lane, int dst_lane) {
    constexpr int UNR = SplitUnroll<LL>::UNR, NR = rot_nr(ROT), NB = 2 * NR + 1, NI = 2 * (NR - 1);
    static_assert(NI >= 2, "stored-state sweep needs at least one inner rotation letter");
    const StatePtrs<L> sx = state_ptrs<L>(sp, lane);
    BlkBuf<NI> qa, qb;
    float da[8], db[8];
    int b = 0;
    load_blk<L, ROT, PL::v[2 * (OFF % LL)], PL::v[2 * (OFF % LL) + 1]>(qa, rec, sx, 0, ph);
#pragma unroll
    for (int k = 0; k < 8; k++) { da[k] = 0.f; db[k] = 0.f; }
    auto step = [&](auto jj) __attribute__((always_inline)) {
        constexpr int J = decltype(jj)::value, I = (J + OFF) % LL, IP = (J + OFF + UNR - 1) % LL, IN = (J + OFF + 1) % LL;
        constexpr int UP = PL::v[2 * I], DN = PL::v[2 * I + 1];
        BlkBuf<NI>& cur = (J & 1) ? qb : qa;
        BlkBuf<NI>& nxt = (J & 1) ? qa : qb;
        float (&dcur)[8] = (J & 1) ? db : da;
        float (&dprv)[8] = (J & 1) ? da : db;
        const int bn = b + 1 < n ? b + 1 : b;
        load_blk<L, ROT, PL::v[2 * IN], PL::v[2 * IN + 1]>(nxt, rec + 16 * bn, sx, NI * bn, ph + 64 * bn);
        const f2 held = b > 0 ? cur.own : wcut;                              // W before unit b (slot 0 is never written)
        const float w = fmaf(held.x, st.p[0].y, -held.y * st.p[0].x);
        phase_apply<L>(st.p[0], cur.rho, false);
        dprv[2 * (NR - 1)] = -st.sg[PL::v[2 * IP]] * w;
        dprv[2 * (NR - 1) + 1] = -st.sg[PL::v[2 * IP + 1]] * w;
        if (b > 0) reduce8_rows<true>(dprv, gblk + 32 * (b - 1), lane, dst_lane);
        constexpr int Q0 = UP < DN ? UP : DN, Q1 = UP < DN ? DN : UP;
#pragma unroll
        for (int k = 0; k < 8; k++) dcur[k] = 0.f;
        dcur[NB - 1] = st.cpm2[Q0 + Q1 - 1] * w;
        static_for([&](auto gg) __attribute__((always_inline)) {
            constexpr int G = decltype(gg)::value;
            using IG = InnerGate<ROT, G>;
            constexpr int Q = IG::template qubit<UP, DN>();
            rot_apply<L, IG::KIND, Q>(st.p, cur.r[IG::CS], cur.r[IG::CS + 1], st);
            dcur[G] = grad_p_own<L, IG::KIND, Q>(st, cur.part[G], st.p[0]);
        }, typename make_iseq<NI>::type{});
        b++;
    };
    for (int full = n / UNR; full > 0; full--) static_for(step, typename make_iseq<UNR>::type{});
    static_for([&](auto jj) __attribute__((always_inline)) { if (b < n) step(jj); }, typename make_iseq<UNR>::type{});
    {
        const int jt = (n - 1) % UNR;
        const f2 held = sp[64 * NI * n];                                     // W after the last inner gate of block n-1
        const float w = fmaf(held.x, st.p[0].y, -held.y * st.p[0].x);        // last unit: the Rz pair of block n-1
        static_for([&](auto jj) __attribute__((always_inline)) {
            constexpr int J = decltype(jj)::value, I = (J + OFF) % LL;
            if (J == jt) {
                float (&dl)[8] = (J & 1) ? db : da;
                dl[2 * (NR - 1)] = -st.sg[PL::v[2 * I]] * w;
                dl[2 * (NR - 1) + 1] = -st.sg[PL::v[2 * I + 1]] * w;
                reduce8_rows<true>(dl, gblk + 32 * (n - 1), lane, dst_lane);
            }
        }, typename make_iseq<UNR>::type{});
    }
}

__device__ __forceinline__ void split_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int ROT, int PPT, int MSTORE, int... UD>
__device__ __forceinline__ void split_body(const KArgs& a) {
    using L = Lay<3, 0>;
    using Lds = SplitLds;
    using PL = PairList<UD...>;
    constexpr int NQ = 3, TH = 128;
    constexpr int LL = sizeof...(UD) / 2;
    constexpr int NR = rot_nr(ROT), NB = 2 * NR + 1, NS = 3 * NQ;
    constexpr int ENT = rot_ent(ROT), NBP = NB - (ENT != 0 ? 1 : 0);          // parameters per block (CZ: see cpf_blocks.h rot_ent)
    static_assert(ENT == 0 || ENT == 1, "two-ended kernel: CP or CZ blocks");
    // MSTORE packs two things: bits 0-1 the storage mode (0 rebuild phases, 1 store phases, 2 store phases and
    // states), bits 2.. the block count K when the kernel is specialised on it (0 = K read from the plan): with K a
    // constant the sweeps are straight-line code - no prefetch-address clamps, tail guards or loop control
    // (~400 scalar / branch instructions per sample-step at config 2) and LDS addresses become immediates.
    // bit 16: TWO wave pairs (samples) per workgroup.  The hardware starts consecutive two-wave workgroups one SIMD apart,
    // (0, 1) then (1, 2), so with two or three workgroups per CU some SIMDs hold two waves and some none (read back through
    // HW_ID: at 512 samples 256 of the 1024 SIMDs idle and 256 doubly booked); four waves per workgroup land on the four SIMDs.
    constexpr int MS = MSTORE & 3, KC = (MSTORE >> 2) & 0xFFF, PAIRS = ((MSTORE >> 16) & 1) + 1;
    constexpr bool STORE = MS != 0;
    constexpr bool PS = MS == 2;                              // stored-state gradient halves
    constexpr int NI = 2 * (NR - 1);
    static_assert(layered_merge(3, 0, ROT), "two-ended kernel: rotation string must end in 'z'");
    static_assert(L::E == 1, "two-ended kernel: one element per lane");
    extern __shared__ __attribute__((aligned(16))) float smem_wg[];
    const int tid = threadIdx.x & 127, lane = tid & 63;     // thread of its wave pair
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pair = PAIRS > 1 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 7) : 0;
    const int NSP = 3 * a.n_actual;                          // surface parameters (2-qubit problems: see cpf_layered.h)
    const int P = KC > 0 ? NSP + NBP * KC : a.P, K = KC > 0 ? KC : (P - NSP) / NBP;
    if (KC > 0 && a.P != P) return;                          // host error: plan and specialisation disagree
    float* smem = smem_wg + pair * Lds::total(K, MS, NI);     // every wave pair has its own copy of the whole map
    // The cut stays on a layer boundary.  An instruction-count-balanced cut (5 of 12, 7 of 16; wave B's pair indices
    // then carry the offset OFF = m mod LL) measured 1 % slower on config 2 and equal on config 3's shape in a
    // same-box A/B (-DCPF_CUT_BALANCED builds it).
#if defined(CPF_CUT_M)
    constexpr int MC = KC > 0 ? CPF_CUT_M : 0, OFF = KC > 0 ? MC % LL : 0;          // A/B builds: this many blocks to wave A
#elif defined(CPF_CUT_BALANCED)
    constexpr int MC = KC > 0 ? split_point_any(KC > 0 ? KC : 2) : 0, OFF = KC > 0 ? MC % LL : 0;
#else
    // bit 17 of MSTORE: the cut for launches that leave at most one wave on a SIMD (K / 2: the step is then the longer
    // wave's chain and nothing else; measured best at 256 / 512 samples, config 3's shape 7 / 8 / 9 of 16: 6.26 / 6.03 /
    // 6.44 ms, config 2's shape 6 / 7 of 12: 5.09 / 5.52 ms); otherwise the cut measured best with two waves per SIMD
    constexpr bool LONE = KC > 0 && ((MSTORE >> 17) & 1) != 0;
    constexpr int MC = KC > 0 ? (LONE ? KC / 2 : split_point(KC > 0 ? KC : 2, LL)) : 0, OFF = KC > 0 ? MC % LL : 0;
#endif
    const int m = KC > 0 ? MC : (a.lone ? lone_point(K, LL) : split_point(K, LL)), kb = K - m;
    // The wave with the longer instruction stream sets the pace of every step and the other one has slack: while the
    // SIMDs are not saturated (a.prio, set by the host for S <= 8 x CUs) the longer one gets the higher issue
    // priority (instruction counts as in split_point_any, + 90 for wave B's loss / penalty / best tracking, + 90 for its owner phase).
    // Same box: config 2 S = 1024 6.70 -> 6.53 ms, S = 2048 12.05 -> 11.17 ms, but S = 4096 22.7 -> 23.9 and config 3's
    // shape at S = 4096 33.3 -> 35.3; with the priority on the SHORTER wave config 3's shape at S = 512 6.88 -> 8.18.
    if (a.prio && (wave == 0) == (160 + 110 * m > 109 * kb + 180)) __builtin_amdgcn_s_setprio(3);
    // an odd batch: the last workgroup's second pair runs a copy of the last sample (it must meet the same barriers) and
    // writes nothing
    const int64_t s_raw = (int64_t)blockIdx.x * PAIRS + pair;
    const bool vs = PAIRS == 1 || s_raw < a.nsamples;
    const int64_t s_idx = vs ? s_raw : (int64_t)a.nsamples - 1;
    const float* init = a.init + s_idx * P;
    const uint8_t* frozen = a.frozen ? a.frozen + s_idx * P : nullptr;
    float* out_best = a.out_params + (s_idx * 2 + 1) * P;
    float* gs = smem + Lds::slab(K);
    (void)Lds::red(K);
    f2* xch = reinterpret_cast<f2*>(smem + Lds::xch(K)) + lane;                 // [0]: psi of wave A, [64]: W of wave B
    f2* ph = reinterpret_cast<f2*>(smem + Lds::phase(K)) + lane + (wave ? 64 * (m + 1) : 0);
    f2* sp = reinterpret_cast<f2*>(smem + Lds::states(K)) + lane + (wave ? 64 * (NI * m + 1) : 0);

    State<L> st;
#pragma unroll
    for (int i = 0; i < L::NLR; i++) { st.lb[i] = (lane >> L::row_lane_bit(i)) & 1; st.sg[i] = st.lb[i] ? 1.f : -1.f; st.sgp[i] = f2{1.f, st.sg[i]}; }
    if constexpr (L::NQ == 3 && L::NLR == 3) {                // one element per lane: -2 on the |11> rows of each qubit pair, else 0
        st.cpm2[0] = (st.lb[0] && st.lb[1]) ? -2.f : 0.f; st.cpm2[1] = (st.lb[0] && st.lb[2]) ? -2.f : 0.f; st.cpm2[2] = (st.lb[1] && st.lb[2]) ? -2.f : 0.f;
    }
    const int col = L::col_index(lane, 0);
    const float2 vt_ = a.target[L::row_index(lane, 0) * L::D + col];
    const f2 vt = {vt_.x, vt_.y};
    const int dst_lane = reduce8_dst_lane(lane);

    // ---- owner state: ALL angles live in wave B, two per lane and register pair (slot o = lane + 64 (2 s + h), h = the
    // half of pair s; the K <= 64 CP angles come first, i.e. in half 0 of pair 0).  Adam, the half-angle scaling and the
    // sin / cos polynomials then run on v_pk_*_f32: one wave spends ~90 VALU operations per step on 2 x 64 angles where two
    // waves used to spend 113 each on 64 angles apiece (18 % of the kernel's VALU instructions).  Wave A (surface gates +
    // its half) is the critical path of every step and does none of this: the penalty of the CP angles, the loss,
    // best-iterate tracking and the epilogue sit in wave B as well.
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
            const int o = lane + 64 * (2 * s + h);
            valid[s][h] = wave != 0 && o < P;
            int j = 0;
            if (valid[s][h]) {
                if constexpr (ENT != 0) j = o;                   // no CP angles: slots in parameter order
                else if (o < K) j = NSP + NB * o + NB - 1;
                else if (o - K < NSP) j = o - K;
                else { const int q = o - K - NSP; j = NSP + NB * (q / (NB - 1)) + q % (NB - 1); }
            }
            pj[s][h] = j;
            const bool cp = valid[s][h] && a.pkind[j];
            if (s == 0 && h == 0) is_cp0 = cp;
            const bool frz = valid[s][h] && frozen && frozen[j];
            if (j < NSP) { cs_off[s][h] = Lds::SURF_CS + 2 * j; g_off[s][h] = j < 8 ? 4 * j : 32 * Lds::NBATCH + 4 * 7; }   // ninth: block 0's spare slot
            else {
                const int bb = (j - NSP) / NBP, k = (j - NSP) % NBP;
                cs_off[s][h] = Lds::BLK_CS + 16 * bb + 2 * k; g_off[s][h] = 32 * Lds::NBATCH + 32 * bb + 4 * k;
            }
            const float t0 = valid[s][h] ? init[j] : 0.f;
            const float hs = cp ? 1.f : 0.5f;                               // CP gates take the full angle, rotations half of it
            const float nl = (valid[s][h] && !frz) ? -a.lr : 0.f;           // frozen angles: the update is multiplied by -0
            if (h == 0) { th[s].x = t0; hscale[s].x = hs; nlr[s].x = nl; } else { th[s].y = t0; hscale[s].y = hs; nlr[s].y = nl; }
            if (valid[s][h] && vs) { a.out_params[(s_idx * 2) * P + j] = t0; out_best[j] = t0; }
        }
        mm[s] = f2{0.f, 0.f}; vv[s] = f2{0.f, 0.f};
    }
    for (int b = tid; b < K; b += TH) {
        smem[Lds::BLK_CS + 16 * b + 14] = 0.f; smem[Lds::BLK_CS + 16 * b + 15] = 0.f;
        if constexpr (ENT != 0) { smem[Lds::BLK_CS + 16 * b + 2 * (NB - 1)] = -1.f; smem[Lds::BLK_CS + 16 * b + 2 * (NB - 1) + 1] = 0.f; }   // CZ = CP(pi), exactly
    }
    if (tid < 16) smem[Lds::ident(K) + tid] = (tid & 1) ? 0.f : 1.f;
    for (int i = NSP + tid; i < NS; i += TH) { smem[Lds::SURF_CS + 2 * i] = 1.f; smem[Lds::SURF_CS + 2 * i + 1] = 0.f; }
    auto publish = [&](int s) __attribute__((always_inline)) {
        const bool pen = s == 0 && a.use_reg && is_cp0;
        const bool lin = pen && a.pen_kind == 0;
        float x = 0.f;
        float2 si = make_float2(0.f, 0.f);
        if (lin) { x = floor_mod_2pi(th[0].x); si = penalty_lookup(smem, x); }
        f2 sn, cn;
        fast_sincos2(th[s] * hscale[s], sn, cn);
        if (valid[s][0]) *reinterpret_cast<float2*>(smem + cs_off[s][0]) = make_float2(cn.x, sn.x);
        if (valid[s][1]) *reinterpret_cast<float2*>(smem + cs_off[s][1]) = make_float2(cn.y, sn.y);
        if (lin) { pval = __fadd_rn(__fmul_rn(si.x, x), si.y); psl = a.r * si.x; }
        else if (pen) { float sl; pval = penalty_lds(smem, a.pen_kind, th[0].x, sl); psl = a.r * sl; }
    };
    split_barrier();
    if (wave != 0) {
#pragma unroll
        for (int s = 0; s < PPT; s++) publish(s);
    }
    split_barrier();

    float best = 0.f, first = 0.f;
    const int iters = a.T > 0 ? a.T : 1;
#ifdef CPF_STAMPS
    unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev = 0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev) :: "memory");
#endif
    for (int it = 0; it < iters; it++) {
        const float2 bc = const_load_f2(a.bias, a.T > 0 ? it : 0);
        float pen_sum = 0.f;
        CPF_STAMP(7)
        // ================= forward: A builds H1 from the identity, B builds W = H2^+ V
        if (wave == 0) {
            st.p[0] = f2{(L::row_index(lane, 0) == col) ? 1.f : 0.f, 0.f};
            float sr[Lds::SURF_F];
#pragma unroll
            for (int i = 0; i < Lds::SURF_F / 4; i++) {
                const float4 v = reinterpret_cast<const float4*>(smem + Lds::SURF_CS)[i];
                sr[4 * i] = v.x; sr[4 * i + 1] = v.y; sr[4 * i + 2] = v.z; sr[4 * i + 3] = v.w;
            }
            static_for([&](auto qq) __attribute__((always_inline)) {       // surface gate on qubit q: Rz(t2) Rx(t1) Rz(t0), main.py:122-124
                constexpr int Q = decltype(qq)::value;
                rot_apply<L, OP_RZ, Q>(st.p, sr[6 * Q], sr[6 * Q + 1], st);
                rot_apply<L, OP_RX, Q>(st.p, sr[6 * Q + 2], sr[6 * Q + 3], st);
                rot_apply<L, OP_RZ, Q>(st.p, sr[6 * Q + 4], sr[6 * Q + 5], st);
            }, typename make_iseq<NQ>::type{});
            if constexpr (PS) sp[0] = st.p[0];
            split_fwd<L, PL, LL, ROT, STORE, PS>(st, smem + Lds::BLK_CS, m, ph, sp);
            xch[0] = st.p[0];
        } else {
            if (a.use_reg) pen_sum = wave_sum(pval);           // CP angles: all owned by wave B, half 0 of pair 0
            st.p[0] = vt;
            split_inv<L, PL, LL, ROT, STORE, PS, OFF>(st, smem + Lds::BLK_CS + 16 * m, smem + Lds::ident(K), kb, ph, sp);
            xch[64] = st.p[0];
        }
        CPF_STAMP(0)                                  // forward half done
        split_barrier();
        CPF_STAMP(1)                                  // waited for the other wave
        // ================= loss (both waves, identical arithmetic), best tracking
        f2 w;                                         // W = H2^+ V at the cut
        if (wave == 0) w = xch[64];
        else { w = st.p[0]; st.p[0] = xch[0]; }
        const f2 tt = cmul_conj(st.p[0], w);          // T = sum psi conj(W)
        float tr = tt.x, ti = tt.y;
        wave_sum2(tr, ti);
        // wave B alone turns T into the loss and tracks the best iterate (it owns every angle); wave A needs only T (for the
        // cotangent)
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
            if (a.hist_loss && lane == 0 && vs) a.hist_loss[s_idx * a.T + it] = regloss;
            if (a.hist_params && vs) {
#pragma unroll
                for (int s = 0; s < PPT; s++) {
                    if (valid[s][0]) a.hist_params[(s_idx * a.T + it) * P + pj[s][0]] = th[s].x;
                    if (valid[s][1]) a.hist_params[(s_idx * a.T + it) * P + pj[s][1]] = th[s].y;
                }
            }
        }
        if (a.T == 0) break;
        CPF_STAMP(2)                                  // loss phase done

        // ================= backward: C = f W at the cut
        {
            const float fr = -a.inv_dim2 * tr, fi = -a.inv_dim2 * ti;
            if (!PS || wave == 0) st.c[0] = cmul(w, f2{fr, fi});
            else st.p[0] = cmul_conj(st.p[0], f2{fr, fi});      // wave B, stored states: f moves onto psi
        }
        if (wave == 0) {
            float d0[8];
            if constexpr (PS) split_bwd_grad_ps<L, PL, LL, ROT>(st, smem + Lds::BLK_CS, m, ph, sp, gs + 32 * Lds::NBATCH, lane, dst_lane, d0);
            else split_bwd_grad<L, PL, LL, ROT, STORE>(st, smem + Lds::BLK_CS, m, ph, gs + 32 * Lds::NBATCH, lane, dst_lane, d0);
            float sr[Lds::SURF_F], ds[8 * Lds::NBATCH];
#pragma unroll
            for (int i = 0; i < Lds::SURF_F / 4; i++) {
                const float4 v = reinterpret_cast<const float4*>(smem + Lds::SURF_CS)[i];
                sr[4 * i] = v.x; sr[4 * i + 1] = v.y; sr[4 * i + 2] = v.z; sr[4 * i + 3] = v.w;
            }
#pragma unroll
            for (int k = 0; k < 8 * Lds::NBATCH; k++) ds[k] = 0.f;
            surface_bwd_independent<L, NQ>(st, sr, ds);
            // nine surface gradients: eight fill one reduce-scatter batch, the ninth rides in the spare slot of
            // block 0's batch (a block has 7 angles): one 17-instruction reduction less on the critical wave
            static_assert(NS == 9 && NB <= 7, "surface gradient packing assumes 3 qubits and a spare slot per block");
            {
                float d8[8];
#pragma unroll
                for (int k = 0; k < 8; k++) d8[k] = ds[k];
                reduce8_rows(d8, gs, lane, dst_lane);
            }
            d0[7] = ds[8];
            reduce8_rows(d0, gs + 32 * Lds::NBATCH, lane, dst_lane);
        } else {
            if constexpr (PS)
                split_fwd_grad_ps<L, PL, LL, ROT, OFF>(st, w, smem + Lds::BLK_CS + 16 * m, kb, ph, sp,
                                                  gs + 32 * Lds::NBATCH + 32 * m, lane, dst_lane);
            else
                split_fwd_grad<L, PL, LL, ROT, STORE, OFF>(st, smem + Lds::BLK_CS + 16 * m, kb, ph, gs + 32 * Lds::NBATCH + 32 * m,
                                                      lane, dst_lane);
        }
        CPF_STAMP(3)                                  // backward half done
        split_barrier();
        CPF_STAMP(4)                                  // waited for the other wave

        // ================= Adam (optax 0.1.1) + next step's (cos, sin) / penalty: wave B, two angles per lane and operation
        if (wave != 0) {
            const f2 ibc = f2{__builtin_amdgcn_rcpf(bc.x), __builtin_amdgcn_rcpf(bc.y)};
#pragma unroll
            for (int s = 0; s < PPT; s++) {
                const float4 g0 = *reinterpret_cast<const float4*>(gs + g_off[s][0]);
                const float4 g1 = *reinterpret_cast<const float4*>(gs + g_off[s][1]);
                f2 g = f2{(g0.x + g0.y) + (g0.z + g0.w), (g1.x + g1.y) + (g1.z + g1.w)};
                if (s == 0 && a.use_reg && is_cp0) g.x += psl;
                // optax scale_by_adam: the same operations in the same order as the scalar form, per half
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
        CPF_STAMP(5)                                  // owner phase done
        split_barrier();
        CPF_STAMP(6)
    }
#ifdef CPF_STAMPS
    if (s_idx == 0 && lane == 0 && a.hist_loss) for (int k = 0; k < 8; k++) a.hist_loss[8 * wave + k] = (float)stamp_sum[k];
#endif

#ifdef CPF_HWID
    // diagnostics build (tools/micro/hwid.py, keep_history, T >= 4): HW_ID / XCC_ID of the sample's two waves in place of
    // the first four entries of its loss history
    if (a.hist_loss && lane == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        a.hist_loss[s_idx * a.T + 2 * wave] = (float)(hw & 0xffff);
        a.hist_loss[s_idx * a.T + 2 * wave + 1] = (float)(xcc & 0xf);
    }
#endif
    // ================= epilogue (optimization.py:365-370, cp_utils.py:144-164): CP owners, first and best all sit in wave B
    if (wave != 0) {
        const int bi = (best < first) ? 1 : 0;
        float acc[3] = {0.f, 0.f, 0.f};
        if (is_cp0) {
            const int j = pj[0][0];
            const float t0 = a.out_params[(s_idx * 2) * P + j], t1 = out_best[j];
            float sl;
            if (a.use_reg) { acc[0] += penalty_lds(smem, a.pen_kind, t0, sl); acc[1] += penalty_lds(smem, a.pen_kind, t1, sl); }
            acc[2] += cz_value(bi ? t1 : t0, a.thr_cp);
        }
#pragma unroll
        for (int k = 0; k < 3; k++) acc[k] = wave_sum(acc[k]);
        if (lane == 0 && vs) {
            const float reg0 = a.use_reg ? a.r * acc[0] : 0.f, reg1 = a.use_reg ? a.r * acc[1] : 0.f;
            const float l0 = first - reg0, l1 = best - reg1;
            a.out_regloss[s_idx * 2 + 0] = first; a.out_regloss[s_idx * 2 + 1] = best;
            a.out_reg[s_idx * 2 + 0] = reg0;      a.out_reg[s_idx * 2 + 1] = reg1;
            a.out_loss[s_idx * 2 + 0] = l0;       a.out_loss[s_idx * 2 + 1] = l1;
            a.out_cz[s_idx] = (int32_t)acc[2];
            a.out_pass[s_idx] = ((bi ? l1 : l0) <= a.entry_loss) ? 1 : 0;
        }
    }
}

template <int ROT, int PPT, int MSTORE, int... UD>
__global__ __launch_bounds__(128 * (((MSTORE >> 16) & 1) + 1), 1) void k_cp_split(const KArgs a) {
    split_body<ROT, PPT, MSTORE, UD...>(a);
}

}  // namespace cpf
