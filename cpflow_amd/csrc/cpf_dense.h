// cpflow_amd: the dense-block kernel for 4..6 qubits — every entangling block is ONE 4x4 complex matrix applied to a
// register-resident qubit pair.
//
// Why.  The gate-by-gate kernels (cpf_layered.h) are bound by VALU issue at the instruction count of their formulation:
// a block is a CP and 2 x (1..3) rotations, each touching every amplitude (2 packed operations per amplitude for a
// rotation on a register bit, 3 with a DPP partner fetch on a lane bit), three times per optimiser step (psi forward,
// psi and C backward) plus one dot product per angle: ~47 packed operations per amplitude per block.  Here a block is
//     forward     psi <- G psi                                   4 complex multiply-adds = 8 operations per amplitude
//     backward    psi <- G^+ psi;  M += conj(C) (x) psi;  C <- G^+ C           3 x 8 operations per amplitude
// i.e. 32, whatever the rotation string, and the block's angles are differentiated afterwards from the 4x4 environment
// M by a miniature adjoint sweep (cpf_dense_math.h) - one thread per block, outside the sweeps.
//
// Layout.  Columns never interact: column bits -> wave bits (LW of them), then lane bits 0 .. NLC-1.  Row bits: the
// remaining NLR = 6 - NLC lane bits are the TOP lane bits (4 and / or 5), every other row bit is a register bit, so a
// thread holds E = 2^NRR complex amplitudes of psi (and of C) as (re, im) register pairs.  Row qubit q sits at canonical
// position q: positions 0 .. NRR-1 are register bits, the last NLR qubits sit on the lanes.  A block whose pair
// involves a lane qubit first exchanges that qubit with a register qubit outside the pair - v_permlane32_swap /
// v_permlane16_swap move lane bit 5 / 4 against a register bit in ONE instruction per register pair, E per matrix -
// and exchanges it back afterwards, so that between blocks the layout is always the canonical one.  No DPP and no LDS
// traffic inside a sweep.
//
// Instances.  Table-dispatched (<NQ, LW, 0>): one kernel per (qubits, waves per sample) serves every layer, rotation
// string and entangler without a compiler - which register bits a block acts on and which exchanges it needs come from a
// per-block table the host builds, dispatched with scalar branches; G is read from LDS into vector registers.
// Layer-specialised (<NQ, LW, ROT, UD...>, ahead of time for the BASELINE shapes, at run time otherwise): straight-line
// sweeps per layer, the per-block 4x4 walks unrolled on the rotation string, and (round 3) G in SCALAR registers: the build
// phase also writes every G to a global scratch, the sweeps read it back with s_load_dwordx16 and every group of four
// amplitudes is one asm statement of 32 packed instructions whose factors are scalar register pairs (cpf_dense_asm.h).
//
// Surface gates (round 3): the first block that touches a qubit carries that qubit's Rz Rx Rz on its input side
// (cpf_dense_math.h), so the sweeps start from the identity and apply no rotation; qubits no block touches keep the
// gate-by-gate path (DenseArgs::surf_mask).
//
// Phases of one optimiser step (T of them in one launch, one workgroup per sample):
//   owner    Adam on the angles this thread owns, their (cos, sin) to the gate records        [LDS: th, m, v, rec]
//   build    thread b builds G_b from the records of block b (and of the surface gates folded into it)  [LDS: Gm; global: gmat]
//   forward  psi = I; surface gates of untouched qubits; blocks b = 0 .. K-1
//   loss     T = sum conj(V) psi over the workgroup; best-iterate tracking
//   backward C = f V; blocks K-1 .. 0 (fused pass above, M reduced over the wave -> LDS env); surface gates of untouched qubits
//   chain    thread b: angle gradients of block b (and of its folded surface gates) from M_b and G_b   [LDS: gw]
#pragma once
#include "cpf_layered.h"
#include "cpf_dense_math.h"
#include "cpf_dense_asm.h"

namespace cpf {

// what the host tells the kernel about block b (int4 per block)
//   x, y : up to two lane<->register exchanges, each (lane_bit << 8) | reg_bit, or -1
//   z    : (ra | rb << 4) with ra < rb the register bits the 4x4 acts on, | 256 if the placement's first qubit sits at ra
//          (the lower bit; the block matrix is then built with its index bits exchanged); host table only: | 512 / 1024 if the
//          surface gate of the up / dn qubit is folded into this block, up qubit << 12, dn qubit << 16
//   w    : index of the block's first gate (the entangler) in the gate program
struct DenseBlock { int x, y, z, w; };

template <int NQ_, int LW_>
struct DLay {
    static constexpr int NQ = NQ_, LW = LW_;
    static constexpr int WPS = 1 << LW, THREADS = 64 * WPS, D = 1 << NQ;
    static constexpr int NLC = NQ - LW;                 // column bits on lanes: lane bits 0 .. NLC-1
    static constexpr int NLR = 6 - NLC;                 // row bits on lanes: lane bits NLC .. 5
    static constexpr int NRR = NQ - NLR;                // row bits in registers
    static constexpr int E = 1 << NRR;
    static_assert(NLC >= 4 && NLC <= 6 && NRR >= 2, "dense layout: row lane bits must be lane bits 4 / 5, a block needs two register bits");
    // canonical position of row qubit q: register bit NRR-1-q for q < NRR (qubit 0 = most significant), else lane bit NLC + (q - NRR)
    __host__ __device__ static constexpr bool on_lane(int q) { return q >= NRR; }
    __host__ __device__ static constexpr int reg_bit(int q) { return NRR - 1 - q; }
    __host__ __device__ static constexpr int lane_bit(int q) { return NLC + (q - NRR); }
    __device__ static int row_index(int lane, int e) {
        int r = 0;
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const int b = on_lane(q) ? ((lane >> lane_bit(q)) & 1) : ((e >> reg_bit(q)) & 1);
            r |= b << (NQ - 1 - q);
        }
        return r;
    }
    __device__ static int col_index(int lane, int wave) {      // column bits: wave bits first (most significant), then lane bits
        return (wave << NLC) | (lane & ((1 << NLC) - 1));
    }
    // row_index(lane, e) = row_of_lane(lane) | row_of_reg(e): the two parts use different bits
    __host__ __device__ static constexpr int row_of_reg(int e) {
        int r = 0;
        for (int q = 0; q < NRR; q++) r |= ((e >> reg_bit(q)) & 1) << (NQ - 1 - q);
        return r;
    }
};

// ---- host side: the per-block table ----------------------------------------------------------------------------
// up, dn: the placement (up = first qubit).  Lane qubits of the pair are exchanged with the lowest register qubits
// outside the pair.
// avoid: qubits not to exchange with (bit mask), when enough other register qubits exist.
template <class L>
__host__ __device__ constexpr DenseBlock dense_block_entry(int up, int dn, int first_gate, int avoid = 0) {
    DenseBlock d{-1, -1, 0, first_gate};
    int pos_up = L::on_lane(up) ? -1 : L::reg_bit(up), pos_dn = L::on_lane(dn) ? -1 : L::reg_bit(dn);
    int lanes = 0, freeq = 0;
    for (int q = 0; q < L::NQ; q++) {
        if (L::on_lane(q)) { if (q == up || q == dn) lanes++; }
        else if (q != up && q != dn && !((avoid >> q) & 1)) freeq++;
    }
    if (freeq < lanes) avoid = 0;                             // not enough register qubits left: plain rule
    int next_q = 0, nswap = 0;
    for (int which = 0; which < 2; which++) {
        const int q = which == 0 ? up : dn;
        if (!L::on_lane(q)) continue;
        while (next_q == up || next_q == dn || L::on_lane(next_q) || ((avoid >> next_q) & 1)) next_q++;       // a register qubit outside the pair
        const int rb = L::reg_bit(next_q++);
        if (nswap == 0) d.x = (L::lane_bit(q) << 8) | rb; else d.y = (L::lane_bit(q) << 8) | rb;
        nswap++;
        if (which == 0) pos_up = rb; else pos_dn = rb;
    }
    const int ra = pos_up < pos_dn ? pos_up : pos_dn, rb = pos_up < pos_dn ? pos_dn : pos_up;
    d.z = ra | (rb << 4) | (pos_up == ra ? 256 : 0);
    return d;
}

// Block at position J of a layer of LL pairs ud[]: when the previous / next block of the (cyclic) layer shares a lane qubit
// with it, neither exchanges that qubit with a register qubit of the other's pair - both then choose the SAME register
// qubit, and the sweeps of a layer-specialised kernel keep the qubit in registers across the two blocks instead of
// exchanging it back and forth (dense_same_swap).  The host's table uses the same rule (same pair positions).
template <class L>
__host__ __device__ constexpr DenseBlock dense_layer_entry(const int* ud, int LL, int J, int first_gate) {
    const int up = ud[2 * J], dn = ud[2 * J + 1];
    int avoid = 0;
    if (LL > 1) {
        const int nb[2] = {(J + LL - 1) % LL, (J + 1) % LL};
        for (int k = 0; k < 2; k++) {
            const int u2 = ud[2 * nb[k]], d2 = ud[2 * nb[k] + 1];
            const bool shares = (L::on_lane(up) && (up == u2 || up == d2)) || (L::on_lane(dn) && (dn == u2 || dn == d2));
            if (shares) avoid |= (1 << u2) | (1 << d2);
        }
    }
    return dense_block_entry<L>(up, dn, first_gate, avoid);
}
// blocks at layer positions J1, J2 make the same single exchange: it can stay in place from one to the other
template <class L>
__host__ __device__ constexpr bool dense_same_swap(const int* ud, int LL, int J1, int J2) {
    if (J1 < 0 || J2 < 0 || J1 >= LL || J2 >= LL) return false;
    const DenseBlock a = dense_layer_entry<L>(ud, LL, J1, 0), b = dense_layer_entry<L>(ud, LL, J2, 0);
    return a.x >= 0 && a.x == b.x && a.y < 0 && b.y < 0;
}

// ---- packed complex helpers --------------------------------------------------------------------------------------
// acc += a * b / conj(a) * b with a, b (re, im) register pairs: exactly two v_pk_fma_f32.  The sign sits on the broadcast
// FACTOR (-a.y, a.y), which the compiler folds into neg_lo / neg_hi next to the op_sel half-selects; written with the sign
// on the swapped operand (-b.y, b.x) it builds that pair with a v_xor and a v_mov per amplitude and pass (+25 % VALU).
__device__ __forceinline__ f2 cmac(f2 acc, f2 a, f2 b) { return pk_fma(f2{-a.y, a.y}, swp(b), pk_fma(splat(a.x), b, acc)); }
__device__ __forceinline__ f2 cmac_conj(f2 acc, f2 a, f2 b) { return pk_fma(f2{a.y, -a.y}, swp(b), pk_fma(splat(a.x), b, acc)); }
// the two halves of a multiply-add separately, so that callers can advance several accumulators in lock step
__device__ __forceinline__ f2 cmac_re(f2 acc, f2 a, f2 b) { return pk_fma(splat(a.x), b, acc); }
__device__ __forceinline__ f2 cmac_im(f2 acc, f2 a, f2 b, bool conj) { return conj ? pk_fma(f2{a.y, -a.y}, swp(b), acc) : pk_fma(f2{-a.y, a.y}, swp(b), acc); }
// acc + (a.y, -a.y) * (b.y, b.x): the imaginary half of acc += conj(a) b, as ONE instruction whatever the number of uses of a.
// (Written in C++ the compiler keeps a multiply-used (a.y, a.y) pair in a register pair of its own: one v_mov_b32 per factor
// and group, 32 per block in the environment accumulation, where every C_i meets four psi_j.)
__device__ __forceinline__ f2 cmac_im_conj_asm(f2 acc, f2 a, f2 b) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[1,0,0]" : "+v"(acc) : "v"(a), "v"(b));
    return acc;
}
__device__ __forceinline__ f2 cmul0(f2 a, f2 b) { return pk_fma(f2{-a.y, a.y}, swp(b), splat(a.x) * b); }
__device__ __forceinline__ f2 cmul0_conj(f2 a, f2 b) { return pk_fma(f2{a.y, -a.y}, swp(b), splat(a.x) * b); }

// exchange lane bit LB (4 or 5) with register bit RB of a matrix: one v_permlane{16,32}_swap per 32-bit register pair
template <int E, int LB, int RB>
__device__ __forceinline__ void swap_lane_reg(f2 (&x)[E]) {
#ifdef CPF_EXP_NOSWAP             // timing experiment only (wrong results)
    return;
#endif
#pragma unroll
    for (int e0 = 0; e0 < E; e0++) {
        if (e0 & (1 << RB)) continue;
        const int e1 = e0 | (1 << RB);
        // vdst = the register-bit-0 element, src = the register-bit-1 element: lanes with the lane bit set of vdst trade
        // places with lanes with the lane bit clear of src
        const f2 a = x[e0], b = x[e1];
        if constexpr (LB == 5) {
            const auto rx = __builtin_amdgcn_permlane32_swap(__float_as_uint(a.x), __float_as_uint(b.x), false, false);
            const auto ry = __builtin_amdgcn_permlane32_swap(__float_as_uint(a.y), __float_as_uint(b.y), false, false);
            x[e0] = f2{__uint_as_float(rx[0]), __uint_as_float(ry[0])};
            x[e1] = f2{__uint_as_float(rx[1]), __uint_as_float(ry[1])};
        } else {
            const auto rx = __builtin_amdgcn_permlane16_swap(__float_as_uint(a.x), __float_as_uint(b.x), false, false);
            const auto ry = __builtin_amdgcn_permlane16_swap(__float_as_uint(a.y), __float_as_uint(b.y), false, false);
            x[e0] = f2{__uint_as_float(rx[0]), __uint_as_float(ry[0])};
            x[e1] = f2{__uint_as_float(rx[1]), __uint_as_float(ry[1])};
        }
    }
}
template <class L>
__device__ __forceinline__ void do_swap(f2 (&x)[L::E], int code) {
    if (code < 0) return;
    const int lb = code >> 8, rb = code & 255;
#define CPF_SW(LBV, RBV) if constexpr ((LBV) >= L::NLC && (RBV) < L::NRR) { if (lb == (LBV) && rb == (RBV)) { swap_lane_reg<L::E, LBV, RBV>(x); return; } }
    CPF_SW(5, 0) CPF_SW(5, 1) CPF_SW(5, 2) CPF_SW(5, 3) CPF_SW(5, 4)
    CPF_SW(4, 0) CPF_SW(4, 1) CPF_SW(4, 2) CPF_SW(4, 3) CPF_SW(4, 4)
#undef CPF_SW
}

// ---- the 4x4 on register bits RA < RB (index = 2 * bit RB + bit RA ... see below) -------------------------------
// g[4 i + j] = G_ij as (re, im).  Basis index of a group: (bit RB of e) * 2 + (bit RA of e): the HIGHER register bit is the
// high index bit; the host / build phase arranges G accordingly (DenseBlock::z bit 8).
template <int E, int RA, int RB, bool ADJ>
__device__ __forceinline__ void dense_apply(f2 (&x)[E], const f2 (&g)[16]) {
#pragma unroll
    for (int e = 0; e < E; e++) {
        if (e & ((1 << RA) | (1 << RB))) continue;
        const int idx[4] = {e, e | (1 << RA), e | (1 << RB), e | (1 << RA) | (1 << RB)};
        const f2 in[4] = {x[idx[0]], x[idx[1]], x[idx[2]], x[idx[3]]};
        // the four outputs advance together (j outer, i inner): four independent accumulation chains in program order, so
        // that a lone wave does not wait out the 9-cycle dependent v_pk_fma_f32 latency (and its op_sel forwarding s_nop)
        f2 acc[4];
#pragma unroll
        for (int i = 0; i < 4; i++) acc[i] = splat((ADJ ? g[i] : g[4 * i]).x) * in[0];
#pragma unroll
        for (int i = 0; i < 4; i++) acc[i] = cmac_im(acc[i], ADJ ? g[i] : g[4 * i], in[0], ADJ);
#pragma unroll
        for (int j = 1; j < 4; j++) {
#pragma unroll
            for (int i = 0; i < 4; i++) acc[i] = cmac_re(acc[i], ADJ ? g[4 * j + i] : g[4 * i + j], in[j]);
#pragma unroll
            for (int i = 0; i < 4; i++) acc[i] = cmac_im(acc[i], ADJ ? g[4 * j + i] : g[4 * i + j], in[j], ADJ);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) x[idx[i]] = acc[i];
        // one group at a time: left alone the scheduler interleaves all E / 4 groups and the live set leaves the register file
        __builtin_amdgcn_sched_barrier(0);
    }
}
// backward pass of one block, fused per group of four amplitudes:  psi <- G^+ psi,  M_ij += conj(C_i) psi_j,  C <- G^+ C.
// PART = 0: everything in one pass (16 complex accumulators).  PART = 1 / 2: the same in two passes over the groups with 8
// accumulators each - rows 0, 1 of M while psi is un-applied, rows 2, 3 while C is - for layouts where psi, C and 16
// accumulators do not fit the register file together (32 amplitudes per thread).
template <int E, int RA, int RB, int PART>
__device__ __forceinline__ void dense_backward(f2 (&p)[E], f2 (&c)[E], const f2 (&g)[16], f2 (&m)[PART == 0 ? 16 : 8]) {
    constexpr int I0 = PART == 2 ? 2 : 0, NI = PART == 0 ? 4 : 2;
#pragma unroll
    for (int e = 0; e < E; e++) {
        if (e & ((1 << RA) | (1 << RB))) continue;
        const int idx[4] = {e, e | (1 << RA), e | (1 << RB), e | (1 << RA) | (1 << RB)};
        const f2 cin[4] = {c[idx[0]], c[idx[1]], c[idx[2]], c[idx[3]]};
        f2 pq[4];
        if constexpr (PART != 2) {
            const f2 pin[4] = {p[idx[0]], p[idx[1]], p[idx[2]], p[idx[3]]};
#pragma unroll
            for (int i = 0; i < 4; i++) pq[i] = splat(g[i].x) * pin[0];
#pragma unroll
            for (int i = 0; i < 4; i++) pq[i] = cmac_im(pq[i], g[i], pin[0], true);
#pragma unroll
            for (int j = 1; j < 4; j++) {
#pragma unroll
                for (int i = 0; i < 4; i++) pq[i] = cmac_re(pq[i], g[4 * j + i], pin[j]);
#pragma unroll
                for (int i = 0; i < 4; i++) pq[i] = cmac_im(pq[i], g[4 * j + i], pin[j], true);
            }
#pragma unroll
            for (int i = 0; i < 4; i++) p[idx[i]] = pq[i];
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) pq[i] = p[idx[i]];
        }
#pragma unroll
        for (int i = 0; i < NI; i++) {
#pragma unroll
            for (int j = 0; j < 4; j++) m[4 * i + j] = cmac_re(m[4 * i + j], cin[I0 + i], pq[j]);
#pragma unroll
            for (int j = 0; j < 4; j++) m[4 * i + j] = cmac_im_conj_asm(m[4 * i + j], cin[I0 + i], pq[j]);
        }
        if constexpr (PART != 1) {
            f2 cq[4];
#pragma unroll
            for (int i = 0; i < 4; i++) cq[i] = splat(g[i].x) * cin[0];
#pragma unroll
            for (int i = 0; i < 4; i++) cq[i] = cmac_im(cq[i], g[i], cin[0], true);
#pragma unroll
            for (int j = 1; j < 4; j++) {
#pragma unroll
                for (int i = 0; i < 4; i++) cq[i] = cmac_re(cq[i], g[4 * j + i], cin[j]);
#pragma unroll
                for (int i = 0; i < 4; i++) cq[i] = cmac_im(cq[i], g[4 * j + i], cin[j], true);
            }
#pragma unroll
            for (int i = 0; i < 4; i++) c[idx[i]] = cq[i];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// The block matrix: eight broadcast ds_read_b128 into 16 (re, im) register pairs.  As VECTOR register pairs the factors
// G_ij take part in v_pk_fma_f32 with their half-selects and signs as op_sel / neg bits (8 operations per amplitude and
// pass, nothing else); held in scalar registers instead the compiler builds every (-im, im) and (-y, x) pair with two
// extra VALU operations (measured: +37 % instructions in the forward sweep).
__device__ __forceinline__ void load_block_matrix(const float* gm, f2 (&g)[16]) {
#ifdef CPF_EXP_NOGLOAD            // timing experiment only (wrong results): every block reads block 0's matrix, hoistable
    extern __shared__ __attribute__((aligned(16))) float cpf_exp_smem[];
    gm = cpf_exp_smem + 64;
#endif
    const float4* gp = reinterpret_cast<const float4*>(gm);
#pragma unroll
    for (int i = 0; i < 8; i++) { const float4 v = gp[i]; g[2 * i] = f2{v.x, v.y}; g[2 * i + 1] = f2{v.z, v.w}; }
}

// ---- the block matrix in SCALAR registers -----------------------------------------------------------------------------
// Read into vector registers the matrix costs every lane 32 register writes per block and pass (8 broadcast ds_read_b128:
// 6.5 % of config 5's step, measured by leaving them out - not latency: fetching the next block's matrix early changed
// nothing).  The matrix is uniform over the wave, so it belongs in scalar registers: the build phase also writes G to a global
// scratch (DenseArgs::gmat, 128 B per block), the sweeps read it back with two s_load_dwordx16 - no vector-register traffic
// at all - and every complex multiply-add takes its factor straight from a scalar register pair (cpf_dense_asm.h: one asm
// statement of 32 instructions per group of four amplitudes; op_sel / neg on a scalar src0 measured in tools/micro/pk_sgpr.hip).
// The scalar data cache does not see vector stores: it is invalidated once per step after the build phase (dense_body).
typedef float f16v __attribute__((ext_vector_type(16)));
typedef const volatile f16v __attribute__((address_space(4)))* gs_ptr;          // volatile: G changes every step, never hoist the read
__device__ __forceinline__ void load_block_matrix_s(const float* gp, f2 (&g)[16]) {
    gs_ptr p = reinterpret_cast<gs_ptr>((const __attribute__((address_space(4))) void*)gp);
    const f16v lo = p[0], hi = p[1];
#pragma unroll
    for (int k = 0; k < 8; k++) { g[k] = f2{lo[2 * k], lo[2 * k + 1]}; g[8 + k] = f2{hi[2 * k], hi[2 * k + 1]}; }
}
// Scalar loads return out of order, so every wait on them is s_waitcnt lgkmcnt(0): to have the NEXT block's matrix in flight
// under the current block's arithmetic its loads must be issued after the wait for the current one.  dense_touch_s makes the
// compiler place that wait here (an empty statement that reads all sixteen pairs), the caller issues the next loads after it.
__device__ __forceinline__ void dense_touch_s(const f2 (&g)[16]) {
    asm volatile("" :: "s"(g[0]), "s"(g[1]), "s"(g[2]), "s"(g[3]), "s"(g[4]), "s"(g[5]), "s"(g[6]), "s"(g[7]),
                       "s"(g[8]), "s"(g[9]), "s"(g[10]), "s"(g[11]), "s"(g[12]), "s"(g[13]), "s"(g[14]), "s"(g[15]));
}
template <int E, int RA, int RB, bool ADJ>
__device__ __forceinline__ void dense_apply_s(f2 (&x)[E], const f2 (&g)[16]) {
#pragma unroll
    for (int e = 0; e < E; e++) {
        if (e & ((1 << RA) | (1 << RB))) continue;
        const int idx[4] = {e, e | (1 << RA), e | (1 << RB), e | (1 << RA) | (1 << RB)};
        f2 o0, o1, o2, o3;
        if constexpr (ADJ) dense_group_adj_s(o0, o1, o2, o3, x[idx[0]], x[idx[1]], x[idx[2]], x[idx[3]], g);
        else dense_group_fwd_s(o0, o1, o2, o3, x[idx[0]], x[idx[1]], x[idx[2]], x[idx[3]], g);
        x[idx[0]] = o0; x[idx[1]] = o1; x[idx[2]] = o2; x[idx[3]] = o3;
    }
}
// the backward pass of one block with the matrix in scalar registers (same parts as dense_backward)
template <int E, int RA, int RB, int PART>
__device__ __forceinline__ void dense_backward_s(f2 (&p)[E], f2 (&c)[E], const f2 (&g)[16], f2 (&m)[PART == 0 ? 16 : 8]) {
    constexpr int I0 = PART == 2 ? 2 : 0, NI = PART == 0 ? 4 : 2;
#pragma unroll
    for (int e = 0; e < E; e++) {
        if (e & ((1 << RA) | (1 << RB))) continue;
        const int idx[4] = {e, e | (1 << RA), e | (1 << RB), e | (1 << RA) | (1 << RB)};
        const f2 cin[4] = {c[idx[0]], c[idx[1]], c[idx[2]], c[idx[3]]};
        f2 pq[4];
        if constexpr (PART != 2) {
            dense_group_adj_s(pq[0], pq[1], pq[2], pq[3], p[idx[0]], p[idx[1]], p[idx[2]], p[idx[3]], g);
#pragma unroll
            for (int i = 0; i < 4; i++) p[idx[i]] = pq[i];
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) pq[i] = p[idx[i]];
        }
#pragma unroll
        for (int i = 0; i < NI; i++) {
#pragma unroll
            for (int j = 0; j < 4; j++) m[4 * i + j] = e == 0 ? splat(cin[I0 + i].x) * pq[j] : cmac_re(m[4 * i + j], cin[I0 + i], pq[j]);
#pragma unroll
            for (int j = 0; j < 4; j++) m[4 * i + j] = cmac_im_conj_asm(m[4 * i + j], cin[I0 + i], pq[j]);
        }
        if constexpr (PART != 1) {
            f2 cq[4];
            dense_group_adj_s(cq[0], cq[1], cq[2], cq[3], cin[0], cin[1], cin[2], cin[3], g);
#pragma unroll
            for (int i = 0; i < 4; i++) c[idx[i]] = cq[i];
        }
    }
}

#define CPF_PAIR_SWITCH(PAIR, BODY)                                                                   \
    switch (PAIR) {                                                                                   \
        case 0x10: { constexpr int RA = 0, RB = 1; BODY; } break;                                     \
        case 0x20: if constexpr (L::NRR > 2) { constexpr int RA = 0, RB = 2; BODY; } break;           \
        case 0x21: if constexpr (L::NRR > 2) { constexpr int RA = 1, RB = 2; BODY; } break;           \
        case 0x30: if constexpr (L::NRR > 3) { constexpr int RA = 0, RB = 3; BODY; } break;           \
        case 0x31: if constexpr (L::NRR > 3) { constexpr int RA = 1, RB = 3; BODY; } break;           \
        case 0x32: if constexpr (L::NRR > 3) { constexpr int RA = 2, RB = 3; BODY; } break;           \
        case 0x40: if constexpr (L::NRR > 4) { constexpr int RA = 0, RB = 4; BODY; } break;           \
        case 0x41: if constexpr (L::NRR > 4) { constexpr int RA = 1, RB = 4; BODY; } break;           \
        case 0x42: if constexpr (L::NRR > 4) { constexpr int RA = 2, RB = 4; BODY; } break;           \
        case 0x43: if constexpr (L::NRR > 4) { constexpr int RA = 3, RB = 4; BODY; } break;           \
        default: break;                                                                               \
    }

// ---- rotations on a register bit (surface gates; the register branch of cpf_device.h rot_apply / rot_grad) --------
template <int E, int KIND, int BIT>
__device__ __forceinline__ void rot_reg(f2 (&x)[E], float c, float s) {
#pragma unroll
    for (int e0 = 0; e0 < E; e0++) {
        if (e0 & (1 << BIT)) continue;
        const int e1 = e0 | (1 << BIT);
        const f2 a = x[e0], b = x[e1];
        if constexpr (KIND == OP_RX) {            // a' = c a - i s b, b' = c b - i s a
            x[e0] = pk_fma(f2{s, -s}, swp(b), splat(c) * a);
            x[e1] = pk_fma(f2{s, -s}, swp(a), splat(c) * b);
        } else {                                  // Rz: a' = (c - i s) a, b' = (c + i s) b
            x[e0] = pk_fma(f2{s, -s}, swp(a), splat(c) * a);
            x[e1] = pk_fma(f2{-s, s}, swp(b), splat(c) * b);
        }
    }
}
// this thread's share of sum Im(conj(C) sigma psi), psi and C right after the gate
template <int E, int KIND, int BIT>
__device__ __forceinline__ float rot_reg_grad(const f2 (&p)[E], const f2 (&c)[E]) {
    f2 acc = {0.f, 0.f};
#pragma unroll
    for (int e0 = 0; e0 < E; e0++) {
        if (e0 & (1 << BIT)) continue;
        const int e1 = e0 | (1 << BIT);
        if constexpr (KIND == OP_RX) { acc = pk_fma(c[e0], swp(p[e1]), acc); acc = pk_fma(c[e1], swp(p[e0]), acc); }
        else { acc = pk_fma(c[e0], swp(p[e0]), acc); acc = pk_fma(-c[e1], swp(p[e1]), acc); }
    }
    return acc.x - acc.y;
}
#define CPF_BIT_SWITCH(BITV, BODY)                                                        \
    switch (BITV) {                                                                       \
        case 0: { constexpr int BIT = 0; BODY; } break;                                   \
        case 1: { constexpr int BIT = 1; BODY; } break;                                   \
        case 2: if constexpr (L::NRR > 2) { constexpr int BIT = 2; BODY; } break;         \
        case 3: if constexpr (L::NRR > 3) { constexpr int BIT = 3; BODY; } break;         \
        case 4: if constexpr (L::NRR > 4) { constexpr int BIT = 4; BODY; } break;         \
        default: break;                                                                   \
    }

// ---- reduction of the environment partials over the wave -----------------------------------------------------------
// 16 values (8 complex accumulators, one half of the environment; value 2 k + c = component c of m[k]): value v summed over
// the wave, left in dst[v].  Lane bits 5 and 4 by permlane swaps of the SAME component of two neighbouring accumulators, so
// that one v_pk_add_f32 settles four values (2 swaps + 1 add per 4 values), then the 4-value DPP reduce-scatter over bits 2, 3
// and plain sums over bits 0, 1.  After the two permlane stages register pair r holds accumulator k = 4 r + 2 (bit 4) + (bit 5).
__device__ __forceinline__ int dense_env_dst16(int lane) {
    return 8 * ((lane >> 3) & 1) + 4 * ((lane >> 4) & 1) + 2 * ((lane >> 5) & 1) + ((lane >> 2) & 1);
}
__device__ __forceinline__ void dense_env_reduce16(const f2 (&m)[8], float* dst, int dst_lane) {
#ifdef CPF_EXP_NOREDUCE           // timing experiment only (wrong results): one add tree per lane, no cross-lane traffic
    dst[dst_lane] = ((m[0].x + m[1].y) + (m[2].x + m[3].y)) + ((m[4].x + m[5].y) + (m[6].x + m[7].y));
    return;
#endif
    f2 w[4], u2[2];
#pragma unroll
    for (int r = 0; r < 4; r++) {                        // bit 5: accumulators 2r (kept by the low half), 2r + 1 (high half)
        const auto tx = __builtin_amdgcn_permlane32_swap(__float_as_uint(m[2 * r].x), __float_as_uint(m[2 * r + 1].x), false, false);
        const auto ty = __builtin_amdgcn_permlane32_swap(__float_as_uint(m[2 * r].y), __float_as_uint(m[2 * r + 1].y), false, false);
        w[r] = f2{__uint_as_float(tx[0]), __uint_as_float(ty[0])} + f2{__uint_as_float(tx[1]), __uint_as_float(ty[1])};
    }
#pragma unroll
    for (int r = 0; r < 2; r++) {                        // bit 4
        const auto tx = __builtin_amdgcn_permlane16_swap(__float_as_uint(w[2 * r].x), __float_as_uint(w[2 * r + 1].x), false, false);
        const auto ty = __builtin_amdgcn_permlane16_swap(__float_as_uint(w[2 * r].y), __float_as_uint(w[2 * r + 1].y), false, false);
        u2[r] = f2{__uint_as_float(tx[0]), __uint_as_float(ty[0])} + f2{__uint_as_float(tx[1]), __uint_as_float(ty[1])};
    }
    const float u[4] = {u2[0].x, u2[0].y, u2[1].x, u2[1].y};
    float e0, e1, f0;                                    // bits 2, 3: reduce-scatter 4 -> 1 under bank masks (cpf_blocks.h)
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %3, %3 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %0, %4, %4 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %1, %5, %5 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %1, %6, %6 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %2, %0, %0 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %2, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xc"
        : "=&v"(e0), "=&v"(e1), "=&v"(f0)
        : "v"(u[0]), "v"(u[1]), "v"(u[2]), "v"(u[3]));
    float r = f0 + dppz<0xB1>(f0);                       // bits 0, 1: plain sums (four lanes end with the same value)
    r += dppz<0x4E>(r);
    dst[dst_lane] = r;
}

// all 16 accumulators (layouts that accumulate the environment in one pass): the two halves in turn
__device__ __forceinline__ void dense_env_reduce32(const f2 (&m)[16], float* dst, int dst_lane) {
    const f2 lo[8] = {m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7]}, hi[8] = {m[8], m[9], m[10], m[11], m[12], m[13], m[14], m[15]};
    dense_env_reduce16(lo, dst, dst_lane);
    dense_env_reduce16(hi, dst + 16, dst_lane);
}

// ---- LDS map (float offsets) -------------------------------------------------------------------------------------------
struct DenseLds {
    int rec, gm, th, mm, vv, psl, gw, env, gsurf, red, total;
    __host__ __device__ DenseLds(int P, int G, int K, int nq, int wps) {
        rec = PEN_FLOATS;                      // G x {cos, sin, code, -}
        gm = rec + 4 * G;                      // K x 16 x (re, im): the block matrices
        th = gm + 32 * K;                      // owner state and r * penalty slope, P each
        mm = th + P; vv = mm + P; psl = vv + P;
        gw = psl + P;                          // G: gradient of each gate's angle
        env = gw + G;                          // [wps][K][32]: environments, reduced over the wave
        gsurf = env + wps * K * 32;            // [wps][NBATCH][32]: surface-gate gradient partials (reduce8_rows slabs)
        red = gsurf + wps * ((3 * nq + 7) / 8) * 32;
        total = red + 8 * wps + 4;
        total = (total + 3) & ~3;
    }
};

// ---- the kernel ---------------------------------------------------------------------------------------------------
// surf_mask: qubits whose surface gate Rz Rx Rz the sweeps apply gate by gate; every other qubit's surface gate is folded
// into the first block that touches it (cpf_dense_math.h; DenseBlock::z bits 9 / 10), which removes 3 x 18 rotations of
// the full state per step (config 5: 6.4 % of the step, measured by leaving them out)
// gmat: global scratch [samples][K][32 floats] for the block matrices of the scalar-register path (layer-specialised kernels),
// each matrix COLUMN-major ((re, im) of G_ij at float 2 (4 j + i)): the build phase's four column walkers write 32 contiguous bytes each
struct DenseArgs { KArgs k; const DenseBlock* blocks; int K, n_rot, surf_mask; float* gmat; };
#ifndef CPF_DENSE_SGPR
#define CPF_DENSE_SGPR 1
#endif

// UD... = the layer as (up, dn) pairs (block b sits on pair b mod length): the block loops are then straight-line code per
// layer with compile-time register bits and exchanges.  Empty UD: any circuit, every block dispatched through the table.
// ROT: the rotation string and entangler as a compile-time code (cpf_blocks.h rot_pack / rot_with_ent) for the per-block
// 4x4 walks, or 0 = read the kinds from the records (any string up to 8 letters).
template <int NQ, int LW, int ROT, int... UD>
__device__ __forceinline__ void dense_body(const DenseArgs& da) {
    using L = DLay<NQ, LW>;
    using PL = PairList<UD...>;
    constexpr int LL = sizeof...(UD) / 2;
    constexpr bool HALVES = L::E >= 32;                  // environment accumulated in two halves (register budget)
    constexpr bool SG = LL > 0 && CPF_DENSE_SGPR != 0;   // block matrices in scalar registers (layer-specialised sweeps)
    constexpr int E = L::E, WPS = L::WPS, THREADS = L::THREADS, NS = 3 * NQ, NBATCH = (NS + 7) / 8;
    const KArgs& a = da.k;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int P = a.P, G = a.G, K = da.K, NR = da.n_rot;
    const DenseLds lds(P, G, K, NQ, WPS);
    float4* rec = reinterpret_cast<float4*>(smem + lds.rec);
    float* th = smem + lds.th;
    float* mm = smem + lds.mm;
    float* vv = smem + lds.vv;
    float* psl = smem + lds.psl;
    float* gw = smem + lds.gw;
    float* red = smem + lds.red;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t s_idx = blockIdx.x;
    const float* init = a.init + s_idx * P;
    const uint8_t* frozen = a.frozen ? a.frozen + s_idx * P : nullptr;
    const int col = L::col_index(lane, wave);
    const int env_dst = dense_env_dst16(lane), surf_dst = reduce8_dst_lane(lane);
    // target element of (lane, e): one 32-bit byte offset per lane + a compile-time offset per e against the scalar base
    // pointer (64-bit per-element addresses would be hoisted out of the step loop: 2 E registers for the whole run)
    const char* tbase = reinterpret_cast<const char*>(a.target);
    const uint32_t toff = (uint32_t)(L::row_index(lane, 0) * L::D + col) * 8u;
    uint32_t toff_i = toff;
    int lane_i = lane, col_i = col;
    auto target_at = [&](int e) __attribute__((always_inline)) {
        return *reinterpret_cast<const float2*>(tbase + (toff_i + (uint32_t)(L::row_of_reg(e) * L::D * 8)));
    };
    float* env_w = smem + lds.env + wave * K * 32;
    float* gsurf_w = smem + lds.gsurf + wave * NBATCH * 32;
    const_i_ptr blk = (const_i_ptr)(const int*)da.blocks;          // scalar loads: the table is wave-uniform

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

    // record fetch of block b for cpf_dense_math.h: gate g0 + i
    auto block_rec = [&](int g0) {
        return [rec, g0](int i, float& c, float& s, int& kind) {
            const float4 r = rec[g0 + i];
            // gate i of every block has the same kind (one rotation string, one entangler per template): a scalar, so that
            // the three-way choice in the 4x4 helpers is a scalar branch and not three masked passes
            c = r.x; s = r.y; kind = __builtin_amdgcn_readfirstlane(__float_as_int(r.z) & 15);
        };
    };

    // (cos, sin) of surface rotation k of the block's up (side 0) / dn (side 1) qubit: gate 3 q + k of the program
    auto surf_rec = [&](int qu, int qd) {
        return [rec, qu, qd](int side, int k, float& c, float& s) {
            const float4 r = rec[3 * (side ? qd : qu) + k];
            c = r.x; s = r.y;
        };
    };

    f2 psi[E], cot[E];
    float best = 0.f, first = 0.f;
    const int iters = a.T > 0 ? a.T : 1;
    for (int it = 0; it < iters; it++) {
        const float2 bc = const_load_f2(a.bias, a.T > 0 ? it : 0);
        // per-lane constants pass through an opaque copy once per step: otherwise everything derived from them (the E
        // identity amplitudes, the E target offsets) is hoisted out of the step loop and occupies registers for the whole run
        asm volatile("" : "+v"(toff_i), "+v"(lane_i), "+v"(col_i));
        // ================= build: column j of G_b from the gate records, four neighbouring threads per block (the walk of a
        // column is a quarter of the matrix's: cpf_dense_math.h "slices")
        for (int w = tid; w < 4 * K; w += THREADS) {
            const int b = w >> 2, j = w & 3;
            const int z = blk[4 * b + 2], g0 = blk[4 * b + 3];
            dense::cx x[4];
#ifdef CPF_EXP_NOBUILD            // timing experiment only (wrong results)
            for (int i = 0; i < 4; i++) x[i] = dense::mk(i == j ? 1.f : 0.f, th[b] * 1e-9f);
#else
            dense::block_matrix_column<ROT>(x, j, block_rec(g0), NR, (z & 512) != 0, (z & 1024) != 0, surf_rec((z >> 12) & 15, (z >> 16) & 15));
#endif
            const bool up_low = (z & 256) != 0;            // first qubit on the lower register bit: exchange the index bits
            float2* out = reinterpret_cast<float2*>(smem + lds.gm + 32 * b);
            float2* gout = SG ? reinterpret_cast<float2*>(da.gmat + ((size_t)s_idx * K + b) * 32) : nullptr;
            const int jj = up_low ? dense::swap_bits2(j) : j;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int ii = up_low ? dense::swap_bits2(i) : i;
                out[4 * ii + jj] = make_float2(dense::re(x[i]), dense::im(x[i]));
                if constexpr (SG) gout[4 * jj + ii] = make_float2(dense::re(x[i]), dense::im(x[i]));     // column-major: this lane's 32 contiguous bytes
            }
        }
        if constexpr (SG) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the matrices have reached L2 ...
        lds_sync<WPS>();
        if constexpr (SG) {                                                        // ... and no stale line of theirs stays in the scalar cache
            __builtin_amdgcn_s_dcache_inv();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        const float* gsample = SG ? da.gmat + (size_t)s_idx * K * 32 : nullptr;

        // ================= forward
#pragma unroll
        for (int e = 0; e < E; e++) psi[e] = f2{(L::row_index(lane_i, e) == col_i) ? 1.f : 0.f, 0.f};
        static_for([&](auto qq) __attribute__((always_inline)) {       // surface gate on qubit q: Rz(t2) Rx(t1) Rz(t0), main.py:122-124
            constexpr int Q = decltype(qq)::value;
            if (!((da.surf_mask >> Q) & 1)) return;                  // folded into the first block that touches the qubit
            const float4 r0 = rec[3 * Q], r1 = rec[3 * Q + 1], r2 = rec[3 * Q + 2];
            // a lane qubit borrows register bit 0 (qubit NRR-1) for its three gates
            constexpr int BIT = L::on_lane(Q) ? 0 : L::reg_bit(Q);
            if constexpr (L::on_lane(Q)) swap_lane_reg<E, L::lane_bit(Q), 0>(psi);
            rot_reg<E, OP_RZ, BIT>(psi, r0.x, r0.y);
            rot_reg<E, OP_RX, BIT>(psi, r1.x, r1.y);
            rot_reg<E, OP_RZ, BIT>(psi, r2.x, r2.y);
            if constexpr (L::on_lane(Q)) swap_lane_reg<E, L::lane_bit(Q), 0>(psi);
        }, typename make_iseq<NQ>::type{});
        if constexpr (LL == 0) {
            for (int b = 0; b < K; b++) {
                const int sx = blk[4 * b], sy = blk[4 * b + 1], pair = blk[4 * b + 2] & 255;
                f2 g[16];
                load_block_matrix(smem + lds.gm + 32 * b, g);
                do_swap<L>(psi, sx);
                do_swap<L>(psi, sy);
                CPF_PAIR_SWITCH(pair, (dense_apply<E, RA, RB, false>(psi, g)));
                do_swap<L>(psi, sy);
                do_swap<L>(psi, sx);
            }
        } else {
            int b = 0;
            // Forward sweep, one layer at a time.  The matrix of the NEXT block of the layer is fetched (eight broadcast LDS
            // reads) before the arithmetic of the current one - C does not exist yet, so its registers are free for a second
            // buffer - and only the first block of a layer waits for its matrix (measured: the exposed reads of all blocks
            // cost 6.5 % of config 5's step).
            auto fwd_layer = [&](auto tail) __attribute__((always_inline)) {
                constexpr bool TAIL = decltype(tail)::value != 0;        // the truncated last layer: blocks while b < K
                if constexpr (SG) {
                    f2 sa[16], sb[16];                               // scalar pairs: this block's matrix and the next one's
                    if (!TAIL || b < K) load_block_matrix_s(gsample + 32 * b, sa);
                    static_for([&](auto jj) __attribute__((always_inline)) {
                        constexpr int J = decltype(jj)::value;
                        constexpr DenseBlock ent = dense_layer_entry<L>(PL::v, LL, J, 0);
                        constexpr int RA = ent.z & 15, RB = (ent.z >> 4) & 15;
                        constexpr bool KEPT = dense_same_swap<L>(PL::v, LL, J - 1, J), KEEP = dense_same_swap<L>(PL::v, LL, J, J + 1);
                        if (TAIL && b >= K) return;
                        auto body = [&](f2 (&g)[16], f2 (&gnext)[16]) __attribute__((always_inline)) {
                            dense_touch_s(g);
                            if constexpr (J + 1 < LL) { if (!TAIL || b + 1 < K) load_block_matrix_s(gsample + 32 * (b + 1), gnext); }
                            if constexpr (ent.x >= 0 && !KEPT) swap_lane_reg<E, (ent.x >> 8), (ent.x & 255)>(psi);
                            if constexpr (ent.y >= 0) swap_lane_reg<E, (ent.y >> 8), (ent.y & 255)>(psi);
                            dense_apply_s<E, RA, RB, false>(psi, g);
                            if constexpr (ent.y >= 0) swap_lane_reg<E, (ent.y >> 8), (ent.y & 255)>(psi);
                            if constexpr (ent.x >= 0) {
                                if (!KEEP || b + 1 >= K) swap_lane_reg<E, (ent.x >> 8), (ent.x & 255)>(psi);
                            }
                            b++;
                        };
                        if constexpr (J & 1) body(sb, sa); else body(sa, sb);
                    }, typename make_iseq<LL>::type{});
                    return;
                }
                f2 ga[16], gb[16];
                if (!TAIL || b < K) load_block_matrix(smem + lds.gm + 32 * b, ga);
                static_for([&](auto jj) __attribute__((always_inline)) {
                    constexpr int J = decltype(jj)::value;
                    constexpr DenseBlock ent = dense_layer_entry<L>(PL::v, LL, J, 0);
                    constexpr int RA = ent.z & 15, RB = (ent.z >> 4) & 15;
                    // the exchange stays in place between two blocks of the layer that make the same one (not across the layer's end)
                    constexpr bool KEPT = dense_same_swap<L>(PL::v, LL, J - 1, J), KEEP = dense_same_swap<L>(PL::v, LL, J, J + 1);
                    auto body = [&](f2 (&g)[16], f2 (&gnext)[16]) __attribute__((always_inline)) {
                        if constexpr (J + 1 < LL) { if (!TAIL || b + 1 < K) load_block_matrix(smem + lds.gm + 32 * (b + 1), gnext); }
                        __builtin_amdgcn_sched_barrier(0);               // the reads stay here: the scheduler would sink them to their first use
                        if constexpr (ent.x >= 0 && !KEPT) swap_lane_reg<E, (ent.x >> 8), (ent.x & 255)>(psi);
                        if constexpr (ent.y >= 0) swap_lane_reg<E, (ent.y >> 8), (ent.y & 255)>(psi);
                        dense_apply<E, RA, RB, false>(psi, g);
                        if constexpr (ent.y >= 0) swap_lane_reg<E, (ent.y >> 8), (ent.y & 255)>(psi);
                        if constexpr (ent.x >= 0) {
                            if (!KEEP || b + 1 >= K) swap_lane_reg<E, (ent.x >> 8), (ent.x & 255)>(psi);       // the last block of the circuit always exchanges back
                        }
                        b++;
                    };
                    if (TAIL && b >= K) return;
                    if constexpr (J & 1) body(gb, ga); else body(ga, gb);
                }, typename make_iseq<LL>::type{});
            };
            for (int full = K / LL; full > 0; full--) fwd_layer(ic<0>{});
            if (b < K) fwd_layer(ic<1>{});
        }

        // ================= loss (matrix_utils.py:35-42) and penalty (main.py:563-564)
        float acc[3] = {0.f, 0.f, 0.f};
        {
            f2 tt = {0.f, 0.f};
#pragma unroll
            for (int e = 0; e < E; e++) {
                const float2 t = target_at(e);
                if (a.loss_kind == 2) acc[0] = fmaf(fmaf(t.x, t.x, t.y * t.y), fmaf(psi[e].x, psi[e].x, psi[e].y * psi[e].y), acc[0]);
                else { tt = pk_fma(splat(t.x), psi[e], tt); tt = pk_fma(f2{t.y, -t.y}, swp(psi[e]), tt); }
            }
            if (a.loss_kind != 2) { acc[0] = tt.x; acc[1] = tt.y; }
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

        // ================= backward: C = (1/2) dL/dU* = f V (or -|V|^2 psi / D for the relative-phase loss)
        {
            const float fr = -a.inv_dim2 * tr, fi = -a.inv_dim2 * ti;
#pragma unroll
            for (int e = 0; e < E; e++) {
                const float2 t = target_at(e);
                if (a.loss_kind == 2) cot[e] = splat(-a.inv_dim2 * fmaf(t.x, t.x, t.y * t.y)) * psi[e];
                else cot[e] = cmul(f2{t.x, t.y}, f2{fr, fi});
            }
        }
        // one block of the backward sweep on register bits RA < RB (psi and C already hold the pair in registers).
        // prefetch(g): called once the last use of the block's matrix has been issued - the layer-specialised sweep reads the
        // next block's matrix into the same registers there, under the environment reduction and the exchanges that follow.
        auto bwd_block = [&](auto ra, auto rb, f2 (&g)[16], int b, auto&& prefetch) __attribute__((always_inline)) {
            constexpr int RA = decltype(ra)::value, RB = decltype(rb)::value;
            if constexpr (SG && HALVES) {
                f2 m[8];
                dense_backward_s<E, RA, RB, 1>(psi, cot, g, m);
                dense_env_reduce16(m, env_w + 32 * b, env_dst);
                dense_backward_s<E, RA, RB, 2>(psi, cot, g, m);
                dense_env_reduce16(m, env_w + 32 * b + 16, env_dst);
            } else if constexpr (SG) {
                f2 m[16];
                dense_backward_s<E, RA, RB, 0>(psi, cot, g, m);
                dense_env_reduce32(m, env_w + 32 * b, env_dst);
            } else if constexpr (HALVES) {
                f2 m[8];
#pragma unroll
                for (int i = 0; i < 8; i++) m[i] = f2{0.f, 0.f};
                dense_backward<E, RA, RB, 1>(psi, cot, g, m);
                dense_env_reduce16(m, env_w + 32 * b, env_dst);
#pragma unroll
                for (int i = 0; i < 8; i++) m[i] = f2{0.f, 0.f};
                dense_backward<E, RA, RB, 2>(psi, cot, g, m);
                prefetch(g);
                dense_env_reduce16(m, env_w + 32 * b + 16, env_dst);
            } else {
                f2 m[16];
#pragma unroll
                for (int i = 0; i < 16; i++) m[i] = f2{0.f, 0.f};
                dense_backward<E, RA, RB, 0>(psi, cot, g, m);
                prefetch(g);
                dense_env_reduce32(m, env_w + 32 * b, env_dst);
            }
        };
        if constexpr (LL == 0) {
            for (int b = K - 1; b >= 0; b--) {
                const int sx = blk[4 * b], sy = blk[4 * b + 1], pair = blk[4 * b + 2] & 255;
                f2 g[16];
                load_block_matrix(smem + lds.gm + 32 * b, g);
                do_swap<L>(psi, sx); do_swap<L>(cot, sx);
                do_swap<L>(psi, sy); do_swap<L>(cot, sy);
                CPF_PAIR_SWITCH(pair, (bwd_block(ic<RA>{}, ic<RB>{}, g, b, [](f2 (&)[16]) {})));
                do_swap<L>(psi, sy); do_swap<L>(cot, sy);
                do_swap<L>(psi, sx); do_swap<L>(cot, sx);
            }
        } else {
            int b = K - 1;
            // one layer, last block first; TAIL: the truncated last layer of the circuit (swept first), blocks J < K % LL only
            auto bwd_layer = [&](auto tail) __attribute__((always_inline)) {
                constexpr bool TAIL = decltype(tail)::value != 0;
                const int tail_len = K % LL;
                f2 g[16], g2[16];                                    // g2: the scalar-register path's second buffer
                if constexpr (!SG) load_block_matrix(smem + lds.gm + 32 * b, g);        // the layer's first matrix: the only exposed read of the layer
                bool first = true;
                static_for([&](auto jj) __attribute__((always_inline)) {
                    constexpr int J = LL - 1 - decltype(jj)::value;
                    if (TAIL && J >= tail_len) return;
                    if constexpr (SG) {
                        // blocks J = LL - 1 ... 0 alternate between the buffers; the first one swept (LL - 1, or tail_len - 1) loads its own
                        f2 (&gc)[16] = (J & 1) ? g2 : g;
                        f2 (&gn)[16] = (J & 1) ? g : g2;
                        if (first) { load_block_matrix_s(gsample + 32 * b, gc); first = false; }
                        dense_touch_s(gc);
                        if constexpr (J > 0) load_block_matrix_s(gsample + 32 * (b - 1), gn);
                    }
                    constexpr DenseBlock ent = dense_layer_entry<L>(PL::v, LL, J, 0);
                    // mirror image of the forward sweep: block J + 1 (swept just before) leaves the exchange in place unless it
                    // was the last block of the circuit's own (b + 1 == K: nothing was swept before this block)
                    constexpr bool KEPT = dense_same_swap<L>(PL::v, LL, J, J + 1), KEEP = dense_same_swap<L>(PL::v, LL, J - 1, J);
                    if constexpr (ent.x >= 0) {
                        if (!KEPT || b + 1 >= K) { swap_lane_reg<E, (ent.x >> 8), (ent.x & 255)>(psi); swap_lane_reg<E, (ent.x >> 8), (ent.x & 255)>(cot); }
                    }
                    if constexpr (ent.y >= 0) { swap_lane_reg<E, (ent.y >> 8), (ent.y & 255)>(psi); swap_lane_reg<E, (ent.y >> 8), (ent.y & 255)>(cot); }
                    bwd_block(ic<(ent.z & 15)>{}, ic<((ent.z >> 4) & 15)>{}, (SG && (J & 1)) ? g2 : g, b, [&](f2 (&gg)[16]) __attribute__((always_inline)) {
                        if constexpr (J > 0 && !SG) {
                            load_block_matrix(smem + lds.gm + 32 * (b - 1), gg);
                            __builtin_amdgcn_sched_barrier(0);           // the reads stay here (the scheduler would sink them to their first use)
                        }
                    });
                    if constexpr (ent.y >= 0) { swap_lane_reg<E, (ent.y >> 8), (ent.y & 255)>(psi); swap_lane_reg<E, (ent.y >> 8), (ent.y & 255)>(cot); }
                    if constexpr (ent.x >= 0 && !KEEP) { swap_lane_reg<E, (ent.x >> 8), (ent.x & 255)>(psi); swap_lane_reg<E, (ent.x >> 8), (ent.x & 255)>(cot); }
                    b--;
                }, typename make_iseq<LL>::type{});
            };
            if (K % LL) bwd_layer(ic<1>{});                  // blocks beyond the last whole layer come first
            for (int full = K / LL; full > 0; full--) bwd_layer(ic<0>{});
        }
        {   // surface gates, backwards: gradient right after each gate, then un-apply it from psi and C
            float ds[8 * NBATCH];
#pragma unroll
            for (int k = 0; k < 8 * NBATCH; k++) ds[k] = 0.f;
            static_for([&](auto qq) __attribute__((always_inline)) {
                constexpr int Q = NQ - 1 - decltype(qq)::value;
                if (!((da.surf_mask >> Q) & 1)) return;
                const float4 r0 = rec[3 * Q], r1 = rec[3 * Q + 1], r2 = rec[3 * Q + 2];
                constexpr int BIT = L::on_lane(Q) ? 0 : L::reg_bit(Q);
                if constexpr (L::on_lane(Q)) { swap_lane_reg<E, L::lane_bit(Q), 0>(psi); swap_lane_reg<E, L::lane_bit(Q), 0>(cot); }
                ds[3 * Q + 2] = rot_reg_grad<E, OP_RZ, BIT>(psi, cot);
                rot_reg<E, OP_RZ, BIT>(psi, r2.x, -r2.y); rot_reg<E, OP_RZ, BIT>(cot, r2.x, -r2.y);
                ds[3 * Q + 1] = rot_reg_grad<E, OP_RX, BIT>(psi, cot);
                rot_reg<E, OP_RX, BIT>(psi, r1.x, -r1.y); rot_reg<E, OP_RX, BIT>(cot, r1.x, -r1.y);
                ds[3 * Q] = rot_reg_grad<E, OP_RZ, BIT>(psi, cot);
                rot_reg<E, OP_RZ, BIT>(psi, r0.x, -r0.y); rot_reg<E, OP_RZ, BIT>(cot, r0.x, -r0.y);
                if constexpr (L::on_lane(Q)) { swap_lane_reg<E, L::lane_bit(Q), 0>(psi); swap_lane_reg<E, L::lane_bit(Q), 0>(cot); }
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

        // ================= chain: the angle gradients of block b from its environment, four neighbouring threads per block:
        // thread j walks column j of G_b and row j of M_b^T and contributes one term to every trace (two-step quad sum)
        for (int w = tid; w < 4 * K; w += THREADS) {
            const int b = w >> 2, j = w & 3;
            const int z = blk[4 * b + 2], g0 = blk[4 * b + 3];
            const bool up_low = (z & 256) != 0, fu = (z & 512) != 0, fd = (z & 1024) != 0;
            const int jj = up_low ? dense::swap_bits2(j) : j;
            dense::cx x[4], zr[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int ii = up_low ? dense::swap_bits2(i) : i;
                float2 mv = make_float2(0.f, 0.f);
                for (int wv = 0; wv < WPS; wv++) {
                    const float2 t = *reinterpret_cast<const float2*>(smem + lds.env + (wv * K + b) * 32 + 2 * (4 * ii + jj));
                    mv.x += t.x; mv.y += t.y;
                }
                zr[i] = dense::mk(mv.x, mv.y);                                   // Z[j][i] = M[i][j]
                const float2 gv = *reinterpret_cast<const float2*>(smem + lds.gm + 32 * b + 2 * (4 * ii + jj));
                x[i] = dense::mk(gv.x, gv.y);
            }
            const int qu = (z >> 12) & 15, qd = (z >> 16) & 15;
            float gpart[dense::MAX_BLOCK_ANGLES], pu[3] = {0.f, 0.f, 0.f}, pd[3] = {0.f, 0.f, 0.f};
#ifdef CPF_EXP_NOCHAIN            // timing experiment only (wrong results)
            gpart[0] = dense::re(zr[0]) + dense::im(x[1]);
#else
            dense::block_gradient_slice<ROT>(gpart, x, zr, block_rec(g0), NR, fu, fd, surf_rec(qu, qd), pu, pd);
#endif
            auto quad_sum = [](float v) __attribute__((always_inline)) { v += xor_lane<0>(v); return v + xor_lane<1>(v); };
            // gate g0 + i carries angle i of the block; a folded surface gate's angles are gates 3 q .. 3 q + 2
            const int na = 1 + 2 * dense::rotc_n<ROT>(NR);
#pragma unroll
            for (int k = 0; k < (ROT ? 1 + 2 * (ROT & 15) : dense::MAX_BLOCK_ANGLES); k++) {
                if (k < na) { const float v = quad_sum(gpart[k]); if (j == 0) gw[g0 + k] = v; }
            }
            if (fu) {
#pragma unroll
                for (int k = 0; k < 3; k++) { const float v = quad_sum(pu[k]); if (j == 0) gw[3 * qu + k] = v; }
            }
            if (fd) {
#pragma unroll
                for (int k = 0; k < 3; k++) { const float v = quad_sum(pd[k]); if (j == 0) gw[3 * qd + k] = v; }
            }
        }
        for (int g = tid; g < NS; g += THREADS) {
            if (!((da.surf_mask >> (g / 3)) & 1)) continue;              // written by the block the gate is folded into
            float s = 0.f;
            for (int w = 0; w < WPS; w++) {
                const float4 gp = *reinterpret_cast<const float4*>(smem + lds.gsurf + w * NBATCH * 32 + 32 * (g / 8) + 4 * (g % 8));
                s += (gp.x + gp.y) + (gp.z + gp.w);
            }
            gw[g] = s;
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
__global__ __launch_bounds__(64 << LW, ((DLay<NQ, LW>::E >= 32 && sizeof...(UD) == 0) ? 1 : 2)) void k_cp_dense(const DenseArgs da) {
    dense_body<NQ, LW, ROT, UD...>(da);
}

}  // namespace cpf
