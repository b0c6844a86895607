// cpflow_amd device code: one workgroup optimises one sample (one CP-template start).
//
// Replaces, per optimiser step, what XLA runs for the reference's
//   build_unitary            cpflow/main.py:106-146   (gates.py:22-58, circuit_assembly.py:31-45)
//   cost_HST                 cpflow/matrix_utils.py:35-42
//   regularization_func      cpflow/main.py:563-564   (penalty.py:44-71)
//   value_and_grad           cpflow/optimization.py:340
//   optax_update_step + best tracking   cpflow/optimization.py:14-25, 61-75
// with a single persistent kernel: all T iterations of a sample run inside one launch.
//
// Data layout (gfx950, wave64).  The 2^n x 2^n unitary U (and its loss cotangent C) is
// spread over the 64*WPS threads of the workgroup, columns first:
//   column bits -> wave bits, then lane bits {5,4,2,3,1,0}   (columns never interact)
//   row bits    -> remaining lane bits {0,1,3,2,..} (DPP-reachable), then register bits
// A gate acts on ROW bits only: on a register bit it is pure in-register arithmetic, on a
// lane bit the partner amplitude comes over DPP (quad_perm / row_ror / row_shl+shr), never
// through LDS.  Gate coefficients (cos, sin) are wave-uniform: computed once per step by
// the parameter's owner thread and broadcast through LDS.
//
// Gradient: adjoint sweep.  After the forward pass psi = U(theta) and C = dL/dU* are walked
// backwards through the gate list; for a rotation exp(-i a sigma/2) the derivative is
// (1/2) sum Im(conj(C) sigma psi) taken right after the gate, then both psi and C are
// multiplied by the gate's inverse.  No intermediate unitaries are stored.
#pragma once
#if !defined(__HIPCC_RTC__)
#include <hip/hip_runtime.h>
#include <stdint.h>
#else      // hiprtc: the runtime header is built in, the fixed-width names are not
typedef unsigned char uint8_t;
typedef int int32_t;
typedef unsigned int uint32_t;
typedef long long int64_t;
typedef unsigned long long uint64_t;
#endif

// The release library and its run-time built kernels know none of the timing-experiment / instrumentation switches of the
// kernel headers ("wrong results by construction", cycle stamps that overwrite hist_loss, ...): without -DCPF_DIAGNOSTICS
// (tools/gpu.py, cpflow_amd.build.build_variant('diag', ['CPF_DIAGNOSTICS', ...])) they are undefined here, whatever the
// command line says.
#ifndef CPF_DIAGNOSTICS
#undef CPF_STAMPS
#undef CPF_HWID
#undef CPF_EXP_NORECLOAD
#undef CPF_EXP_NOSWAP
#undef CPF_EXP_NOGLOAD
#undef CPF_EXP_NOREDUCE
#undef CPF_EXP_NOBUILD
#undef CPF_EXP_NOCHAIN
#undef CPF_EXP_SWZ
#undef CPF_MARKS
#endif
// phase markers for the static listings (tools/phase_mix.py): an assembler comment, in -DCPF_DIAGNOSTICS -DCPF_MARKS builds only
#ifdef CPF_MARKS
#define CPF_MARK(name) asm volatile("; CPF_MARK " #name ::: "memory")
#else
#define CPF_MARK(name)
#endif

namespace cpf {

enum : int { OP_RX = 0, OP_RY = 1, OP_RZ = 2, OP_CP = 3, OP_CZ = 4, OP_CX = 5 };
enum : int { MODE_OPTIMIZE = 0, MODE_LOSS_GRAD = 1, MODE_UNITARY = 2 };

struct Piece { float lo, hi, slope, icpt; };

// read-only tables (gate program, bias corrections) are read through the constant address space so
// that wave-uniform reads become scalar loads (s_load) instead of per-lane global loads
typedef const float __attribute__((address_space(4)))* const_f_ptr;
typedef const int __attribute__((address_space(4)))* const_i_ptr;
__device__ __forceinline__ float2 const_load_f2(const float2* p, int i) {
    const_f_ptr q = (const_f_ptr)(const float*)p;
    return make_float2(q[2 * i], q[2 * i + 1]);
}
__device__ __forceinline__ int4 const_load_i4(const int4* p, int i) {
    const_i_ptr q = (const_i_ptr)(const int*)p;
    return make_int4(q[4 * i], q[4 * i + 1], q[4 * i + 2], q[4 * i + 3]);
}

struct KArgs {
    const int4* prog;          // [G] {kind, q0, q1, param}
    int G, P, T, mode;
    int n_actual;              // problem qubits (<= kernel qubits; spectators are the low bits)
    const float2* target;      // [D][D] embedded target, D = 2^NQ
    const uint8_t* pkind;      // [P] 1 = CP angle (full angle, penalised), 0 = rotation (half angle)
    const int32_t* pgate;      // [P] index of the gate that carries angle j
    const float2* bias;        // [T] {1 - b1^t, 1 - b2^t}, t = 1..T
    const float2* ibias;       // [T] their reciprocals in float32 (cpf_zyz3.h: two v_rcp_f32 fewer per wave and step)
    Piece pc[10];
    int pen_kind, use_reg;
    int loss_kind;             // 0 HST / state preparation (|T|^2 form, scale in inv_dim2), 2 relative phase
    float r, lr, b1, b2, eps, omb1, omb2, thr_cp, entry_loss, inv_dim2;
    const float* init;         // [S][P]
    const uint8_t* frozen;     // [S][P] or null
    float* out_params;         // [S][2][P]
    float* out_regloss;        // [S][2]   (MODE_LOSS_GRAD: [S])
    float* out_loss;           // [S][2]   (MODE_LOSS_GRAD: [S])
    float* out_reg;            // [S][2]   (MODE_LOSS_GRAD: [S])
    int32_t* out_cz;           // [S]
    uint8_t* out_pass;         // [S]
    float* hist_params;        // [S][T][P] or null
    float* hist_loss;          // [S][T] or null
    float* out_grad;           // MODE_LOSS_GRAD: [S][P] or null
    float2* out_U;             // MODE_UNITARY: [S][N][N]
    int prio;                  // two-ended kernels: bit 0 = raise the longer wave's issue priority (launches that leave SIMDs unsaturated); bit 1 (cpf_zyz3.h) = start the workgroups' step loops out of phase
    int nsamples;              // samples of this launch (kernels with several samples per workgroup mask the tail)
    int lone;                  // two-ended kernel, K from the plan: at most one sample per CU - balance the two waves' chains
    int group;                 // one-wave layered kernels: samples per workgroup (0 / 1: one; cpf_layered.h layered_body)
};

// ------------------------------------------------------------------ layout
template <int NQ_, int LW_>
struct Lay {
    static constexpr int NQ = NQ_, LW = LW_;
    static constexpr int WPS = 1 << LW;
    static constexpr int THREADS = 64 * WPS;
    static constexpr int D = 1 << NQ;
    static constexpr int NLC = NQ - LW;                          // column bits on lanes
    static constexpr int NLR = (6 - NLC) < NQ ? (6 - NLC) : NQ;  // row bits on lanes
    static constexpr int NRR = NQ - NLR;                         // row bits in registers
    static constexpr int E = 1 << NRR;                           // complex elements per thread
    static_assert(NLC >= 0 && NLC <= 6, "too many column bits for one wave");
    static_assert(NLC + NLR == 6, "layout must use all 64 lanes");
    __host__ __device__ static constexpr int row_lane_bit(int i) {
        constexpr int t[6] = {0, 1, 3, 2, 4, 5};
        return t[i];
    }
    __host__ __device__ static constexpr int col_lane_bit(int i) {
        constexpr int t[6] = {5, 4, 2, 3, 1, 0};
        return t[i];
    }
    __host__ __device__ static constexpr bool row_on_lane(int q) { return q < NLR; }
    __host__ __device__ static constexpr int row_bit(int q) { return q < NLR ? row_lane_bit(q) : q - NLR; }
    // row / column index (qubit 0 = most significant bit) of element e of (wave, lane)
    __device__ static int row_index(int lane, int e) {
        int r = 0;
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            int b = row_on_lane(q) ? ((lane >> row_bit(q)) & 1) : ((e >> row_bit(q)) & 1);
            r |= b << (NQ - 1 - q);
        }
        return r;
    }
    __device__ static int col_index(int lane, int wave) {
        int c = 0;
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            int b = q < LW ? ((wave >> q) & 1) : ((lane >> col_lane_bit(q - LW)) & 1);
            c |= b << (NQ - 1 - q);
        }
        return c;
    }
};

// ------------------------------------------------------------------ cross-lane helpers
template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF>
__device__ __forceinline__ float dpp(float old, float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(x), CTRL, ROW_MASK,
                                                      BANK_MASK, false));
}
// full-mask DPP read with bound_ctrl: the form GCNDPPCombine folds into the consuming VOP2
// (v_mul_f32_dpp / v_fmac_f32_dpp / v_add_f32_dpp), so the lane exchange costs no instruction
template <int CTRL>
__device__ __forceinline__ float dppz(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
// value held by lane (lane ^ (1 << BIT))
template <int BIT>
__device__ __forceinline__ float xor_lane(float x) {
#ifdef CPF_EXP_SWZ          // A/B build: the partner over the LDS crossbar (ds_swizzle_b32, bit-mask mode) instead of DPP
    if constexpr (BIT < 5) return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(x), ((1 << BIT) << 10) | 0x1f));
#endif
    if constexpr (BIT == 0) return dppz<0xB1>(x);            // quad_perm:[1,0,3,2]
    else if constexpr (BIT == 1) return dppz<0x4E>(x);       // quad_perm:[2,3,0,1]
    else if constexpr (BIT == 3) return dppz<0x128>(x);      // row_ror:8
    else if constexpr (BIT == 2) {
        float t = dpp<0x104, 0xF, 0x5>(x, x);                // row_shl:4 -> banks 0,2 read lane+4
        return dpp<0x114, 0xF, 0xA>(t, x);                   // row_shr:4 -> banks 1,3 read lane-4
    } else return __shfl_xor(x, 1 << BIT);
}
// sum over the 64 lanes, returned wave-uniform; fixed order => run-to-run deterministic
__device__ __forceinline__ float wave_sum(float x) {
    x += dppz<0xB1>(x);
    x += dppz<0x4E>(x);
    x += dppz<0x124>(x);                                     // row_ror:4
    x += dppz<0x128>(x);                                     // row_ror:8 -> every lane holds its row total
    x += dpp<0x142, 0xA, 0xF>(0.f, x);                       // row_bcast:15 into rows 1,3
    x += dpp<0x143, 0xC, 0xF>(0.f, x);                       // row_bcast:31 into rows 2,3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}
// two sums at once: the stages of the two reductions alternate, so each fills the other's DPP wait states
__device__ __forceinline__ void wave_sum2(float& x, float& y) {
    x += dppz<0xB1>(x);  y += dppz<0xB1>(y);
    x += dppz<0x4E>(x);  y += dppz<0x4E>(y);
    x += dppz<0x124>(x); y += dppz<0x124>(y);
    x += dppz<0x128>(x); y += dppz<0x128>(y);
    x += dpp<0x142, 0xA, 0xF>(0.f, x); y += dpp<0x142, 0xA, 0xF>(0.f, y);
    x += dpp<0x143, 0xC, 0xF>(0.f, x); y += dpp<0x143, 0xC, 0xF>(0.f, y);
    x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
    y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y), 63));
}

// ------------------------------------------------------------------ packed complex arithmetic
// A complex amplitude lives in an even-aligned register pair (re, im).  v_pk_mul_f32 / v_pk_fma_f32 take one
// issue slot like their scalar forms (measured: tools/micro/valu_rate.hip, 4 cycles per wave64 instruction per
// SIMD either way) and can swap, broadcast and negate the halves of an operand for free (op_sel / neg_lo /
// neg_hi), so a complex multiply is 2 instructions instead of 4 and a rotation whose partner amplitude arrives
// over DPP is 2 x v_mul_f32_dpp + 1 x v_pk_fma_f32 instead of 4 (tools/micro/pk_gate.hip: -21 % at saturation).
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 splat(float v) { return f2{v, v}; }
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 swp(f2 v) { return f2{v.y, v.x}; }
// x * (rho.x + i rho.y)  /  x * conj(rho).  hipcc folds a per-half sign into neg_lo / neg_hi only when the negated
// factor is a scalar of its own (as in the Rz forms below), not an element of a register pair; for a pair-valued
// rho the signed broadcast (-rho.y, rho.y) costs one v_pk_mul_f32 by a constant pair, shared when the same rho
// multiplies psi and C: 3 (or 2.5) instructions instead of 4 (tools/micro: /tmp experiments in DESIGN.md).
__device__ __forceinline__ f2 cmul(f2 x, f2 rho) { return pk_fma(splat(rho.y) * f2{-1.f, 1.f}, swp(x), splat(rho.x) * x); }
__device__ __forceinline__ f2 cmul_conj(f2 x, f2 rho) { return pk_fma(splat(rho.y) * f2{1.f, -1.f}, swp(x), splat(rho.x) * x); }
// (re, im) -> (i * s) * (re, im) = (-s im, s re);  (-i s) * (re, im) = (s im, -s re)
__device__ __forceinline__ f2 mul_pi(f2 x, float s) { return f2{-s, s} * swp(x); }
__device__ __forceinline__ f2 mul_mi(f2 x, float s) { return f2{s, -s} * swp(x); }

// ------------------------------------------------------------------ per-thread state
template <class L>
struct State {
    f2 p[L::E];                    // psi  (the unitary being built / un-built), (re, im)
    f2 c[L::E];                    // C    (cotangent of the loss, scaled by 1/2)
    float sg[L::NLR > 0 ? L::NLR : 1];   // +1 if this lane's bit of lane-row-qubit i is set, else -1
    f2 sgp[L::NLR > 0 ? L::NLR : 1];     // (1, sg): turns a record's (cos, sin) into cos + i sigma sin in one packed multiply
    bool lb[L::NLR > 0 ? L::NLR : 1];
    float cpm2[3];                       // 3-qubit kernels: the CP gradient's lane factor for the pairs (0,1), (0,2), (1,2): -2 or 0
};

// ---- rotations: forward / inverse application to one matrix
// s is sin(a/2) for the forward gate, -sin(a/2) for the inverse.
// SH (register qubits, x / y rotations; the layered kernels' blocks): the rotation as three shears (lifting steps)
//     a += u b,  b += v a,  a += u b      with u = -i t (x) or -t (y), v = -i s (x) or +s (y), t = tan(a/4) = s / (1 + c):
// [[1, u], [0, 1]] [[1, 0], [v, 1]] [[1, u], [0, 1]] = [[1 + uv, u (2 + uv)], [v, 1 + uv]] = [[c, -i s], [-i s, c]] (x: uv = -ts = c - 1,
// t (1 + c) = s) - three packed FMAs per amplitude pair instead of two multiplies and two FMAs.  The caller passes (t, s) for the
// gate and (-t, -s) for its inverse (which is then the exact inverse of what the forward sweep applied); the owner phase
// publishes t in the record's cosine slot, after taking the angle into |a/2| <= pi / 2 at the price of a global sign that no
// loss sees (shear_record below), so |t| <= tan(pi / 4) = 1 and the steps are as well conditioned as the rotation.
#ifndef CPF_SHEAR
#define CPF_SHEAR 1          // A/B: 0 = two multiplies + two FMAs per pair everywhere
#endif
template <class L, int KIND, int Q>
__host__ __device__ constexpr bool shear_gate() { return CPF_SHEAR != 0 && KIND != OP_RZ && !L::row_on_lane(Q); }
// (cos, sin) of a gate's half angle -> what a shear gate's record holds: (tan of the quarter angle, sin), signs as above
__device__ __forceinline__ void shear_record(float& c, float& s) {
    const float sg = c < 0.f ? -1.f : 1.f;
    s *= sg;
    c = s * __frcp_rn(1.f + sg * c);
}
template <class L, int KIND, int Q, bool SH = false>
__device__ __forceinline__ void rot_apply(f2 (&x)[L::E], float c, float s, const State<L>& st) {
    constexpr int BIT = L::row_bit(Q);
    static_assert(!SH || (KIND != OP_RZ && !L::row_on_lane(Q)), "shear form: x / y rotations on register qubits");
    if constexpr (L::row_on_lane(Q)) {
        const float ss = st.sg[Q] * s;
#pragma unroll
        for (int e = 0; e < L::E; e++) {
            // the partner amplitude enters through plain multiplies so that the DPP reads fold into them
            // (v_mul_f32_dpp); the c * x + t part is one v_pk_fma_f32
            if constexpr (KIND == OP_RX) {
                f2 t;
                t.x = s * xor_lane<BIT>(x[e].y);
                t.y = -s * xor_lane<BIT>(x[e].x);
                x[e] = pk_fma(splat(c), x[e], t);
            } else if constexpr (KIND == OP_RY) {
                f2 t;
                t.x = ss * xor_lane<BIT>(x[e].x);
                t.y = ss * xor_lane<BIT>(x[e].y);
                x[e] = pk_fma(splat(c), x[e], t);
            } else {
                x[e] = pk_fma(f2{-ss, ss}, swp(x[e]), splat(c) * x[e]);
            }
        }
    } else {
#pragma unroll
        for (int e0 = 0; e0 < L::E; e0++) {
            if (e0 & (1 << BIT)) continue;
            const int e1 = e0 | (1 << BIT);
            const f2 a = x[e0], b = x[e1];
            if constexpr (SH) {                       // c holds t
                if constexpr (KIND == OP_RX) {
                    const f2 a1 = pk_fma(f2{c, -c}, swp(b), a);
                    const f2 b1 = pk_fma(f2{s, -s}, swp(a1), b);
                    x[e0] = pk_fma(f2{c, -c}, swp(b1), a1);
                    x[e1] = b1;
                } else {
                    const f2 a1 = pk_fma(splat(-c), b, a);
                    const f2 b1 = pk_fma(splat(s), a1, b);
                    x[e0] = pk_fma(splat(-c), b1, a1);
                    x[e1] = b1;
                }
                continue;
            }
            if constexpr (KIND == OP_RX) {            // a' = c a - i s b, b' = c b - i s a
                x[e0] = pk_fma(f2{s, -s}, swp(b), splat(c) * a);
                x[e1] = pk_fma(f2{s, -s}, swp(a), splat(c) * b);
            } else if constexpr (KIND == OP_RY) {     // a' = c a - s b, b' = c b + s a
                x[e0] = pk_fma(splat(-s), b, splat(c) * a);
                x[e1] = pk_fma(splat(s), a, splat(c) * b);
            } else {                                  // a' = (c - i s) a, b' = (c + i s) b
                x[e0] = pk_fma(f2{s, -s}, swp(a), splat(c) * a);
                x[e1] = pk_fma(f2{-s, s}, swp(b), splat(c) * b);
            }
        }
    }
}

// ---- a block's Rz(up) (x) Rz(dn) as ONE complex multiply per amplitude (more than one amplitude per lane).
// Rz(a) multiplies an amplitude by cos(a/2) + i sigma sin(a/2), sigma = +1 / -1 with the qubit's bit set / clear, so the pair is
// (cu cd - su sd sigma_u sigma_d) + i (sigma_u su cd + sigma_d cu sd): four phases, two of them the conjugates of the other two -
// six scalar operations per block and sweep, then two packed instructions per amplitude instead of four.  A lane qubit's sigma
// is the lane's sign (folded into its sine), a register qubit's the bit of the amplitude index.  Inverse: both sines negated.
#ifndef CPF_RZPAIR
#define CPF_RZPAIR 1         // A/B: 0 = two separate Rz applications
#endif
template <class L, int UP, int DN>
struct RzPair {
    float ar, ai, br, bi;        // sigma_u = sigma_d = +1: (ar, ai);  sigma_u = -1, sigma_d = +1: (br, bi);  the others: conjugates
    __device__ __forceinline__ RzPair(float cu, float su, float cd, float sd, const State<L>& st) {
        float Su = su, Sd = sd;
        if constexpr (L::row_on_lane(UP)) Su *= st.sg[UP];
        if constexpr (L::row_on_lane(DN)) Sd *= st.sg[DN];
        const float p = cu * cd, t = cu * Sd;
        ar = fmaf(-Su, Sd, p); br = fmaf(Su, Sd, p);
        ai = fmaf(Su, cd, t);  bi = fmaf(-Su, cd, t);
    }
    __device__ __forceinline__ void apply(f2 (&x)[L::E]) const {
#pragma unroll
        for (int e = 0; e < L::E; e++) {
            const bool pu = L::row_on_lane(UP) || ((e >> L::row_bit(UP)) & 1), pd = L::row_on_lane(DN) || ((e >> L::row_bit(DN)) & 1);
            if (pu == pd) {
                if (pu) x[e] = pk_fma(f2{-ai, ai}, swp(x[e]), splat(ar) * x[e]);
                else x[e] = pk_fma(f2{ai, -ai}, swp(x[e]), splat(ar) * x[e]);
            } else {
                if (pd) x[e] = pk_fma(f2{-bi, bi}, swp(x[e]), splat(br) * x[e]);
                else x[e] = pk_fma(f2{bi, -bi}, swp(x[e]), splat(br) * x[e]);
            }
        }
    }
};

// ---- rotations: derivative contribution, evaluated with psi, C right after the gate.
// Returns this thread's share of  sum Im(conj(C) sigma psi)   (C carries the factor 1/2).
template <class L, int KIND, int Q>
__device__ __forceinline__ float rot_grad(const State<L>& st) {
    constexpr int BIT = L::row_bit(Q);
    if constexpr (L::row_on_lane(Q)) {
        if constexpr (L::E == 1) {
            if constexpr (KIND == OP_RX) {
                const float t1 = st.c[0].x * xor_lane<BIT>(st.p[0].y), t2 = st.c[0].y * xor_lane<BIT>(st.p[0].x);
                return t1 - t2;
            } else if constexpr (KIND == OP_RY) {
                const float t1 = st.c[0].x * xor_lane<BIT>(st.p[0].x), t2 = st.c[0].y * xor_lane<BIT>(st.p[0].y);
                return st.sg[Q] * (t1 + t2);
            } else {
                return -st.sg[Q] * fmaf(st.c[0].x, st.p[0].y, -st.c[0].y * st.p[0].x);
            }
        } else {
            f2 acc = {0.f, 0.f};
#pragma unroll
            for (int e = 0; e < L::E; e++) {
                if constexpr (KIND == OP_RX) {        // (Cr X(psi_i), Ci X(psi_r))
                    f2 t;
                    t.x = st.c[e].x * xor_lane<BIT>(st.p[e].y);
                    t.y = st.c[e].y * xor_lane<BIT>(st.p[e].x);
                    acc += t;
                } else if constexpr (KIND == OP_RY) { // (Cr X(psi_r), Ci X(psi_i))
                    f2 t;
                    t.x = st.c[e].x * xor_lane<BIT>(st.p[e].x);
                    t.y = st.c[e].y * xor_lane<BIT>(st.p[e].y);
                    acc += t;
                } else {                              // (Cr psi_i, Ci psi_r)
                    acc = pk_fma(st.c[e], swp(st.p[e]), acc);
                }
            }
            if constexpr (KIND == OP_RX) return acc.x - acc.y;
            else if constexpr (KIND == OP_RY) return st.sg[Q] * (acc.x + acc.y);
            else return -st.sg[Q] * (acc.x - acc.y);
        }
    } else {
        f2 acc = {0.f, 0.f};
#pragma unroll
        for (int e0 = 0; e0 < L::E; e0++) {
            if (e0 & (1 << BIT)) continue;
            const int e1 = e0 | (1 << BIT);
            if constexpr (KIND == OP_RX) {            // Cr0 pi1 - Ci0 pr1 + Cr1 pi0 - Ci1 pr0
                acc = pk_fma(st.c[e0], swp(st.p[e1]), acc);
                acc = pk_fma(st.c[e1], swp(st.p[e0]), acc);
            } else if constexpr (KIND == OP_RY) {     // -Cr0 pr1 - Ci0 pi1 + Cr1 pr0 + Ci1 pi0
                acc = pk_fma(st.c[e1], st.p[e0], acc);
                acc = pk_fma(-st.c[e0], st.p[e1], acc);
            } else {                                  // Cr0 pi0 - Ci0 pr0 - Cr1 pi1 + Ci1 pr1
                acc = pk_fma(st.c[e0], swp(st.p[e0]), acc);
                acc = pk_fma(-st.c[e1], swp(st.p[e1]), acc);
            }
        }
        if constexpr (KIND == OP_RY) return acc.x + acc.y;
        else return acc.x - acc.y;
    }
}

// ---- controlled phase diag(1,1,1,e^{ia}) on qubits Q0 < Q1 (symmetric in its qubits)
template <class L, int Q0, int Q1>
__device__ __forceinline__ bool cp_lane_pred(const State<L>& st) {
    bool p = true;
    if constexpr (L::row_on_lane(Q0)) p = p && st.lb[Q0];
    if constexpr (L::row_on_lane(Q1)) p = p && st.lb[Q1];
    return p;
}
template <class L, int Q0, int Q1>
__host__ __device__ constexpr bool cp_reg_sel(int e) {
    bool s = true;
    if (!L::row_on_lane(Q0)) s = s && ((e >> L::row_bit(Q0)) & 1);
    if (!L::row_on_lane(Q1)) s = s && ((e >> L::row_bit(Q1)) & 1);
    return s;
}
template <class L, int Q0, int Q1>
__device__ __forceinline__ void cp_apply(f2 (&x)[L::E], float cc, float sc, const State<L>& st) {
    const bool p = cp_lane_pred<L, Q0, Q1>(st);
    const f2 rho = {p ? cc : 1.f, p ? sc : 0.f};
#pragma unroll
    for (int e = 0; e < L::E; e++) {
        if (!cp_reg_sel<L, Q0, Q1>(e)) continue;
        x[e] = cmul(x[e], rho);
    }
}
// d/da: dU = i |11><11| U  ->  dL/da = -2 sum_{11} (Cr psi_i - Ci psi_r)   (C carries 1/2)
template <class L, int Q0, int Q1>
__device__ __forceinline__ float cp_grad(const State<L>& st) {
    f2 acc = {0.f, 0.f};
#pragma unroll
    for (int e = 0; e < L::E; e++) {
        if (!cp_reg_sel<L, Q0, Q1>(e)) continue;
        acc = pk_fma(st.c[e], swp(st.p[e]), acc);
    }
    return cp_lane_pred<L, Q0, Q1>(st) ? -2.f * (acc.x - acc.y) : 0.f;
}

// ---- CNOT (gates.py:40-43 cx_mat): control QC ("up"), target QT ("down"); a row permutation, its own inverse
template <class L, int QC, int QT>
__device__ __forceinline__ void cx_apply(f2 (&x)[L::E], const State<L>& st) {
    constexpr int BC = L::row_bit(QC), BT = L::row_bit(QT);
    if constexpr (L::row_on_lane(QT)) {
#pragma unroll
        for (int e = 0; e < L::E; e++) {
            bool ctl;
            if constexpr (L::row_on_lane(QC)) ctl = st.lb[QC]; else ctl = (e >> BC) & 1;
            const float qr = xor_lane<BT>(x[e].x), qi = xor_lane<BT>(x[e].y);
            x[e].x = ctl ? qr : x[e].x;
            x[e].y = ctl ? qi : x[e].y;
        }
    } else {
#pragma unroll
        for (int e0 = 0; e0 < L::E; e0++) {
            if (e0 & (1 << BT)) continue;
            const int e1 = e0 | (1 << BT);
            bool ctl;
            if constexpr (L::row_on_lane(QC)) ctl = st.lb[QC]; else ctl = (e0 >> BC) & 1;
            const f2 a = x[e0], b = x[e1];
            x[e0].x = ctl ? b.x : a.x; x[e0].y = ctl ? b.y : a.y;
            x[e1].x = ctl ? a.x : b.x; x[e1].y = ctl ? a.y : b.y;
        }
    }
}

// ------------------------------------------------------------------ dispatch on (kind, qubit)
#define CPF_Q_SWITCH(q, BODY)                                            \
    switch (q) {                                                         \
        case 0: { constexpr int Q = 0; BODY; } break;                    \
        case 1: { constexpr int Q = 1; BODY; } break;                    \
        case 2: { constexpr int Q = 2; BODY; } break;                    \
        case 3: if constexpr (L::NQ > 3) { constexpr int Q = 3; BODY; } break; \
        case 4: if constexpr (L::NQ > 4) { constexpr int Q = 4; BODY; } break; \
        case 5: if constexpr (L::NQ > 5) { constexpr int Q = 5; BODY; } break; \
        default: break;                                                  \
    }

#define CPF_PAIR_CASE(A, B)                                              \
    case A * 8 + B: if constexpr (L::NQ > B) { constexpr int Q0 = A, Q1 = B; BODY_PAIR; } break;
#define CPF_OPAIR_CASE(A, B)                                             \
    case A * 8 + B: if constexpr (L::NQ > A && L::NQ > B) { constexpr int Q0 = A, Q1 = B; BODY_PAIR; } break;
#define CPF_ALL_ORDERED_PAIRS                                                                                      \
    CPF_OPAIR_CASE(0, 1) CPF_OPAIR_CASE(0, 2) CPF_OPAIR_CASE(0, 3) CPF_OPAIR_CASE(0, 4) CPF_OPAIR_CASE(0, 5)       \
    CPF_OPAIR_CASE(1, 0) CPF_OPAIR_CASE(1, 2) CPF_OPAIR_CASE(1, 3) CPF_OPAIR_CASE(1, 4) CPF_OPAIR_CASE(1, 5)       \
    CPF_OPAIR_CASE(2, 0) CPF_OPAIR_CASE(2, 1) CPF_OPAIR_CASE(2, 3) CPF_OPAIR_CASE(2, 4) CPF_OPAIR_CASE(2, 5)       \
    CPF_OPAIR_CASE(3, 0) CPF_OPAIR_CASE(3, 1) CPF_OPAIR_CASE(3, 2) CPF_OPAIR_CASE(3, 4) CPF_OPAIR_CASE(3, 5)       \
    CPF_OPAIR_CASE(4, 0) CPF_OPAIR_CASE(4, 1) CPF_OPAIR_CASE(4, 2) CPF_OPAIR_CASE(4, 3) CPF_OPAIR_CASE(4, 5)       \
    CPF_OPAIR_CASE(5, 0) CPF_OPAIR_CASE(5, 1) CPF_OPAIR_CASE(5, 2) CPF_OPAIR_CASE(5, 3) CPF_OPAIR_CASE(5, 4)

// forward application of op to psi
template <class L>
__device__ __forceinline__ void op_forward(int kind, int q0, int q1, float c, float s, State<L>& st) {
    switch (kind) {
        case OP_RX: CPF_Q_SWITCH(q0, (rot_apply<L, OP_RX, Q>(st.p, c, s, st))); break;
        case OP_RY: CPF_Q_SWITCH(q0, (rot_apply<L, OP_RY, Q>(st.p, c, s, st))); break;
        case OP_RZ: CPF_Q_SWITCH(q0, (rot_apply<L, OP_RZ, Q>(st.p, c, s, st))); break;
        case OP_CX: {
#define BODY_PAIR cx_apply<L, Q0, Q1>(st.p, st)
            switch (q0 * 8 + q1) { CPF_ALL_ORDERED_PAIRS default: break; }
#undef BODY_PAIR
        } break;
        default: {
            if (kind == OP_CZ) { c = -1.f; s = 0.f; }
#define BODY_PAIR cp_apply<L, Q0, Q1>(st.p, c, s, st)
            switch (q0 * 8 + q1) {
                CPF_PAIR_CASE(0, 1) CPF_PAIR_CASE(0, 2) CPF_PAIR_CASE(1, 2)
                CPF_PAIR_CASE(0, 3) CPF_PAIR_CASE(1, 3) CPF_PAIR_CASE(2, 3)
                CPF_PAIR_CASE(0, 4) CPF_PAIR_CASE(1, 4) CPF_PAIR_CASE(2, 4) CPF_PAIR_CASE(3, 4)
                CPF_PAIR_CASE(0, 5) CPF_PAIR_CASE(1, 5) CPF_PAIR_CASE(2, 5) CPF_PAIR_CASE(3, 5) CPF_PAIR_CASE(4, 5)
                default: break;
            }
#undef BODY_PAIR
        } break;
    }
}

// backward step for op: returns this thread's gradient share, then un-applies the gate on psi and C
template <class L>
__device__ __forceinline__ float op_backward(int kind, int q0, int q1, float c, float s, State<L>& st) {
    float d = 0.f;
    switch (kind) {
#define CPF_ROT_BWD(KIND)                                                         \
        CPF_Q_SWITCH(q0, (d = rot_grad<L, KIND, Q>(st),                           \
                          rot_apply<L, KIND, Q>(st.p, c, -s, st),         \
                          rot_apply<L, KIND, Q>(st.c, c, -s, st)))
        case OP_RX: CPF_ROT_BWD(OP_RX); break;
        case OP_RY: CPF_ROT_BWD(OP_RY); break;
        case OP_RZ: CPF_ROT_BWD(OP_RZ); break;
#undef CPF_ROT_BWD
        case OP_CX: {
#define BODY_PAIR (cx_apply<L, Q0, Q1>(st.p, st), cx_apply<L, Q0, Q1>(st.c, st))
            switch (q0 * 8 + q1) { CPF_ALL_ORDERED_PAIRS default: break; }
#undef BODY_PAIR
        } break;
        default: {
            const bool is_cz = (kind == OP_CZ);
            if (is_cz) { c = -1.f; s = 0.f; }
#define BODY_PAIR (d = is_cz ? 0.f : cp_grad<L, Q0, Q1>(st),                      \
                   cp_apply<L, Q0, Q1>(st.p, c, -s, st),                  \
                   cp_apply<L, Q0, Q1>(st.c, c, -s, st))
            switch (q0 * 8 + q1) {
                CPF_PAIR_CASE(0, 1) CPF_PAIR_CASE(0, 2) CPF_PAIR_CASE(1, 2)
                CPF_PAIR_CASE(0, 3) CPF_PAIR_CASE(1, 3) CPF_PAIR_CASE(2, 3)
                CPF_PAIR_CASE(0, 4) CPF_PAIR_CASE(1, 4) CPF_PAIR_CASE(2, 4) CPF_PAIR_CASE(3, 4)
                CPF_PAIR_CASE(0, 5) CPF_PAIR_CASE(1, 5) CPF_PAIR_CASE(2, 5) CPF_PAIR_CASE(3, 5) CPF_PAIR_CASE(4, 5)
                default: break;
            }
#undef BODY_PAIR
        } break;
    }
    return d;
}

// ------------------------------------------------------------------ penalty (penalty.py:44-71)
// jnp.mod(a, float32(2 pi)) (floored).  a - floor(a/c) c in one fma is the exactly rounded value of
// the true remainder (what fmod + the reference's sign fix-up produce); the fix-up lines only
// act when the float32 quotient estimate is off by one.
__device__ __forceinline__ float floor_mod_2pi(float a) {
    const float twopi = 6.2831855f;
    const float k = floorf(a * 0.15915494f);
    float r = fmaf(-k, twopi, a);
    r = (r < 0.f) ? r + twopi : r;                        // selects, not branches: the owner phase is on every
    r = (r >= twopi) ? r - twopi : r;                     // step's critical path
    return r;
}
// sin and cos of one angle for the gate records: Cody-Waite reduction by pi/2 (three-term constant, exact
// products through fma: good for |x| < ~1e5, far beyond where Adam takes an angle) and the cephes single-precision
// polynomials on [-pi/4, pi/4]; <= 1 ulp of the result scale (max abs error 9e-8 over |x| < 1e4 against float64; every unitary / loss parity test runs through it).
// ~25 instructions and no branch, against ~100 with a large-argument path for the device library's sincosf.
__device__ __forceinline__ void fast_sincos(float x, float& sn, float& cn) {
    const float k = rintf(x * 0.636619772f);
    float r = fmaf(-k, 1.57079637f, x);
    r = fmaf(-k, -4.37113883e-8f, r);
    r = fmaf(-k, -1.71512489e-15f, r);
    const float z = r * r;
    float ps = fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    ps = fmaf(ps, z, -1.6666654611e-1f);
    const float s0 = fmaf(ps * z, r, r);
    float pc = fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    pc = fmaf(pc, z, 4.166664568298827e-2f);
    const float c0 = fmaf(pc, z * z, fmaf(-0.5f, z, 1.f));
    const int q = (int)k;
    const bool odd = q & 1;
    const float s1 = odd ? c0 : s0, c1 = odd ? s0 : c0;
    // quadrant signs: sin negative for q mod 4 in {2, 3}, cos negative for {1, 2}
    sn = __uint_as_float(__float_as_uint(s1) ^ (((unsigned)q & 2u) << 30));
    cn = __uint_as_float(__float_as_uint(c1) ^ ((((unsigned)q + 1u) & 2u) << 30));
}
// (m & a) | (~m & b) as one v_bfi_b32
__device__ __forceinline__ unsigned z_bfi(unsigned m, unsigned a, unsigned b) {
    unsigned o;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(o) : "v"(m), "v"(a), "v"(b));
    return o;
}
// The same for two angles in one register pair: the reduction and both polynomials on v_pk_*_f32 (one instruction for both
// angles), rounding / integer / select steps per half.  Per half the operations and their order are those of fast_sincos,
// so the two give bit-identical results.
__device__ __forceinline__ void fast_sincos2(f2 x, f2& sn, f2& cn) {
    const f2 t = x * splat(0.636619772f);
    const f2 k = f2{rintf(t.x), rintf(t.y)};
    f2 r = pk_fma(-k, splat(1.57079637f), x);
    r = pk_fma(-k, splat(-4.37113883e-8f), r);
    r = pk_fma(-k, splat(-1.71512489e-15f), r);
    const f2 z = r * r;
    f2 ps = pk_fma(splat(-1.9515295891e-4f), z, splat(8.3321608736e-3f));
    ps = pk_fma(ps, z, splat(-1.6666654611e-1f));
    const f2 s0 = pk_fma(ps * z, r, r);
    f2 pc = pk_fma(splat(2.443315711809948e-5f), z, splat(-1.388731625493765e-3f));
    pc = pk_fma(pc, z, splat(4.166664568298827e-2f));
    const f2 c0 = pk_fma(pc, z * z, pk_fma(splat(-0.5f), z, splat(1.f)));
    const int qx = (int)k.x, qy = (int)k.y;
    // odd quadrant: sin and cos change places.  A bit-field select under the sign-extended parity bit (v_bfe_i32, v_bfi_b32: three
    // instructions per half) instead of v_and / v_cmp / two v_cndmask and the two wait states between the compare and its readers
    // (round 5's listing); the same bits either way.
    const unsigned mx = (unsigned)__builtin_amdgcn_sbfe(qx, 0, 1), my = (unsigned)__builtin_amdgcn_sbfe(qy, 0, 1);
    const unsigned s0x = __float_as_uint(s0.x), c0x = __float_as_uint(c0.x), s0y = __float_as_uint(s0.y), c0y = __float_as_uint(c0.y);
    const unsigned sx = z_bfi(mx, c0x, s0x), cx = z_bfi(mx, s0x, c0x), sy = z_bfi(my, c0y, s0y), cy = z_bfi(my, s0y, c0y);
    sn = f2{__uint_as_float(sx ^ (((unsigned)qx & 2u) << 30)), __uint_as_float(sy ^ (((unsigned)qy & 2u) << 30))};
    cn = f2{__uint_as_float(cx ^ ((((unsigned)qx + 1u) & 2u) << 30)), __uint_as_float(cy ^ ((((unsigned)qy + 1u) & 2u) << 30))};
}
// LDS copy of the piece table: 8 lower break points (pieces 1..8), then 9 x (slope, intercept); 28 floats.
// Selecting among kernel-argument fields made hipcc select the field ADDRESS and issue a per-lane global
// load from the kernarg segment, so the table is staged in LDS once and read with one per-lane ds_read.
constexpr int PEN_LO = 0, PEN_SI = 8, PEN_FLOATS = 28;
__device__ __forceinline__ void penalty_table_init(float* tab, const KArgs& a, int tid) {
    if (tid < 8) tab[PEN_LO + tid] = a.pc[tid + 1].lo;
    if (tid < 9) { tab[PEN_SI + 2 * tid] = a.pc[tid].slope; tab[PEN_SI + 2 * tid + 1] = a.pc[tid].icpt; }
}
// the (lo, hi] pieces tile [0, 2 pi] and x lies in it, so jnp.piecewise's "last true condition wins"
// is the piece whose index is the number of break points below x
__device__ __forceinline__ float penalty_lds(const float* tab, int pen_kind, float ang, float& slope) {
    if (pen_kind == 1) {                                      // penalty.py:74-76
        slope = (ang > 0.f) ? 1.f : ((ang < 0.f) ? -1.f : 0.f);
        return fabsf(ang);
    }
    // The piece without compares: a packed (x - lo) * 2^40 clamped to [0, 1] is exactly 1 above a break point and 0 at or below it
    // (x and the eight interior break points are float32 numbers of [0, 2 pi]: where they differ they differ by more than 2^-30; the
    // product by a power of two is exact).  Four packed FMAs and four adds where eight v_cmp / v_cndmask pairs each waited two
    // states for their mask (~45 instructions with their s_nop in the listings of rounds 5 and 6); the same piece, the same two roundings.
    const float x = floor_mod_2pi(ang);
    const float4 l0 = *reinterpret_cast<const float4*>(tab + PEN_LO);
    const float4 l1 = *reinterpret_cast<const float4*>(tab + PEN_LO + 4);
    constexpr float BIG = 1099511627776.f;                   // 2^40
    const f2 xb = {x * BIG, x * BIG};
    f2 c0, c1, c2, c3;
    asm("v_pk_fma_f32 %0, %3, %4, %7 neg_lo:[1,0,0] neg_hi:[1,0,0] clamp\n\t"
        "v_pk_fma_f32 %1, %3, %5, %7 neg_lo:[1,0,0] neg_hi:[1,0,0] clamp\n\t"
        "v_pk_fma_f32 %2, %3, %6, %7 neg_lo:[1,0,0] neg_hi:[1,0,0] clamp"
        : "=&v"(c0), "=&v"(c1), "=&v"(c2) : "v"(f2{BIG, BIG}), "v"(f2{l0.x, l0.y}), "v"(f2{l0.z, l0.w}), "v"(f2{l1.x, l1.y}), "v"(xb));
    asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,0] neg_hi:[1,0,0] clamp" : "=v"(c3) : "v"(f2{BIG, BIG}), "v"(f2{l1.z, l1.w}), "v"(xb));
    const f2 cs01 = (c0 + c1) + (c2 + c3);
    const int idx = (int)(cs01.x + cs01.y);
    const float2 si = *reinterpret_cast<const float2*>(tab + PEN_SI + 2 * idx);
    slope = si.x;
    return __fadd_rn(__fmul_rn(si.x, x), si.y);
}
// the same in two halves, so that the table lookups are in flight while other work (sincos) executes
__device__ __forceinline__ float2 penalty_lookup(const float* tab, float x) {
    const float4 l0 = *reinterpret_cast<const float4*>(tab + PEN_LO);
    const float4 l1 = *reinterpret_cast<const float4*>(tab + PEN_LO + 4);
    const int idx = (x > l0.x) + (x > l0.y) + (x > l0.z) + (x > l0.w) + (x > l1.x) + (x > l1.y) + (x > l1.z) + (x > l1.w);
    return *reinterpret_cast<const float2*>(tab + PEN_SI + 2 * idx);
}
// cp_utils.py:45-50 cz_value
__device__ __forceinline__ float cz_value(float ang, float t) {
    const float pi = 3.1415927f, twopi = 6.2831855f;
    const float x = floor_mod_2pi(ang);
    float v = 2.f;
    if (x < t) v = 0.f;
    if (fabsf(x - twopi) < t) v = 0.f;
    if (fabsf(x - pi) < t) v = 1.f;
    return v;
}

// ------------------------------------------------------------------ workgroup synchronisation
// LDS hand-offs inside one workgroup.  One wave: a wave's LDS operations execute in order, so only the
// compiler has to keep the order.  Several waves: wait for this wave's LDS operations, then s_barrier —
// NOT __syncthreads(), whose fence also waits for outstanding global stores (best-angle / history rows),
// which are fire-and-forget here.
template <int WPS>
__device__ __forceinline__ void lds_sync() {
    if constexpr (WPS == 1) {
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
    } else {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
}

template <class L, int NV>
__device__ __forceinline__ void block_sum(float (&v)[NV], float* red, int wave, int lane) {
#pragma unroll
    for (int k = 0; k < NV; k++) v[k] = wave_sum(v[k]);
    if constexpr (L::WPS > 1) {
        lds_sync<L::WPS>();                               // previous readers of red are done
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < NV; k++) red[wave * NV + k] = v[k];
        }
        lds_sync<L::WPS>();
#pragma unroll
        for (int k = 0; k < NV; k++) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < L::WPS; w++) s += red[w * NV + k];
            v[k] = s;
        }
    }
}

// LDS map of the generic kernel (float offsets)
struct GenLds {
    int rec, th, mm, vv, psl, gw, red, total;
    __host__ __device__ GenLds(int P, int G, int wps) {
        rec = PEN_FLOATS;                 // G x {cos, sin, code, -}: one broadcast ds_read_b128 per gate
        th = rec + 4 * G;                 // theta, Adam m, v, r * penalty slope: owner state, P each
        mm = th + P; vv = mm + P; psl = vv + P;
        gw = psl + P;                     // [wps][G] per-wave gradient of the gate's angle
        red = gw + wps * G;
        total = red + 8 * wps + 4;
    }
};
__host__ __device__ inline size_t lds_floats(int P, int G, int wps) { return (size_t)GenLds(P, G, wps).total; }

// ------------------------------------------------------------------ the generic kernel
// One workgroup (1 or 4 waves) per sample, any n <= 6, any rotation string, CP or CZ entangler.
// Per gate: one 16-byte LDS record {cos, sin, opcode} prefetched one gate ahead, dispatch on
// (kind, qubit), and in the backward sweep one wave reduction of the gate's gradient share.
template <int NQ, int LW>
__global__ __launch_bounds__(64 << LW) void k_cp_template(const KArgs a) {
    using L = Lay<NQ, LW>;
    constexpr int E = L::E;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int P = a.P, G = a.G;
    const GenLds lds(P, G, L::WPS);
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

    State<L> st;
#pragma unroll
    for (int i = 0; i < L::NLR; i++) {
        st.lb[i] = (lane >> L::row_lane_bit(i)) & 1;
        st.sg[i] = st.lb[i] ? 1.f : -1.f;
        st.sgp[i] = f2{1.f, st.sg[i]};
    }
    const int col = L::col_index(lane, wave);
    f2 vt[E];
    bool diag[E];
#pragma unroll
    for (int e = 0; e < E; e++) {
        const int row = L::row_index(lane, e);
        const float2 t = a.target[row * L::D + col];
        vt[e] = f2{t.x, t.y};
        diag[e] = (row == col);
    }

    // static part of the gate records: opcode = kind | q0 << 4 | q1 << 8; CZ gates carry (cos, sin) = (-1, 0)
    penalty_table_init(smem, a, tid);
    for (int g = tid; g < G; g += L::THREADS) {
        const int4 op = a.prog[g];
        rec[g] = make_float4(op.x == OP_CZ ? -1.f : 1.f, 0.f, __int_as_float(op.x | (op.y << 4) | (op.z << 8)), 0.f);
    }
    lds_sync<L::WPS>();
    // owner threads: parameter j lives with thread j % THREADS; pgate[j] = the gate that carries it
    auto publish = [&](int j, float t) {
        float sn, cn;
        const bool cp = a.pkind[j];
        fast_sincos(cp ? t : 0.5f * t, sn, cn);
        float2* r = reinterpret_cast<float2*>(&rec[a.pgate[j]]);
        *r = make_float2(cn, sn);
    };
    for (int j = tid; j < P; j += L::THREADS) {
        const float t0 = init[j];
        th[j] = t0; mm[j] = 0.f; vv[j] = 0.f;
        if (a.mode == MODE_OPTIMIZE) {
            a.out_params[(s_idx * 2 + 0) * P + j] = t0;
            a.out_params[(s_idx * 2 + 1) * P + j] = t0;
        }
        publish(j, t0);
    }
    lds_sync<L::WPS>();

    float best = 0.f, first = 0.f;
    const int iters = (a.mode == MODE_OPTIMIZE && a.T > 0) ? a.T : 1;
    for (int it = 0; it < iters; it++) {
        const float2 bc = const_load_f2(a.bias, (a.mode == MODE_OPTIMIZE && a.T > 0) ? it : 0);
        // ---- forward: psi = G_{G-1} ... G_0
#pragma unroll
        for (int e = 0; e < E; e++) st.p[e] = f2{diag[e] ? 1.f : 0.f, 0.f};
        {
            float4 rc = rec[0];
            for (int g = 0; g < G; g++) {
                const float4 rn = rec[g + 1 < G ? g + 1 : g];                 // prefetch the next record
                const int code = __builtin_amdgcn_readfirstlane(__float_as_int(rc.z));
                op_forward<L>(code & 15, (code >> 4) & 15, (code >> 8) & 15, rc.x, rc.y, st);
                rc = rn;
            }
        }
        if (a.mode == MODE_UNITARY) {
            const int d = NQ - a.n_actual, msk = (1 << d) - 1, N = 1 << a.n_actual;
#pragma unroll
            for (int e = 0; e < E; e++) {
                const int row = L::row_index(lane, e);
                if (((row | col) & msk) == 0)
                    a.out_U[(s_idx * N + (row >> d)) * N + (col >> d)] = make_float2(st.p[e].x, st.p[e].y);
            }
            return;
        }
        // ---- loss (matrix_utils.py:35-42) and penalty (main.py:563-564)
        float acc[3] = {0.f, 0.f, 0.f};
        if (a.loss_kind == 2) {
#pragma unroll
            for (int e = 0; e < E; e++)                       // sum |V|^2 |psi|^2
                acc[0] = fmaf(fmaf(vt[e].x, vt[e].x, vt[e].y * vt[e].y), fmaf(st.p[e].x, st.p[e].x, st.p[e].y * st.p[e].y), acc[0]);
        } else {
            f2 tt = {0.f, 0.f};
#pragma unroll
            for (int e = 0; e < E; e++) {                     // T = sum psi conj(V)
                tt = pk_fma(splat(vt[e].x), st.p[e], tt);
                tt = pk_fma(f2{vt[e].y, -vt[e].y}, swp(st.p[e]), tt);
            }
            acc[0] = tt.x; acc[1] = tt.y;
        }
        if (a.use_reg) {
            for (int j = tid; j < P; j += L::THREADS) {
                if (a.pkind[j]) {
                    float sl;
                    acc[2] += penalty_lds(smem, a.pen_kind, th[j], sl);
                    psl[j] = a.r * sl;
                }
            }
        }
        block_sum<L, 3>(acc, red, wave, lane);
        const float tr = acc[0], ti = acc[1];
        const float ab = sqrtf(fmaf(tr, tr, ti * ti));
        const float loss = (a.loss_kind == 2) ? 1.f - tr * a.inv_dim2 : 1.f - ab * ab * a.inv_dim2;
        const float reg = a.use_reg ? a.r * acc[2] : 0.f;
        const float regloss = loss + reg;

        if (a.mode == MODE_OPTIMIZE) {
            // best-so-far tracking on the PRE-update angles, strict < (optimization.py:70-73)
            bool improved = false;
            if (it == 0) { first = regloss; best = regloss; }
            else if (regloss < best) { best = regloss; improved = true; }
            if (improved || a.hist_params) {
                for (int j = tid; j < P; j += L::THREADS) {
                    const float t = th[j];
                    if (improved) a.out_params[(s_idx * 2 + 1) * P + j] = t;
                    if (a.hist_params) a.hist_params[(s_idx * a.T + it) * P + j] = t;
                }
            }
            if (a.hist_loss && tid == 0) a.hist_loss[s_idx * a.T + it] = regloss;
            if (a.T == 0) break;
        } else {
            if (tid == 0) { a.out_loss[s_idx] = loss; a.out_reg[s_idx] = reg; a.out_regloss[s_idx] = regloss; }
            if (!a.out_grad) return;
        }

        // ---- backward sweep.  C = (1/2) dL/dU*: -(1/D^2) T V, or -(1/D) |V|^2 psi for the relative-phase loss
        if (a.loss_kind == 2) {
#pragma unroll
            for (int e = 0; e < E; e++) {
                const float w = -a.inv_dim2 * fmaf(vt[e].x, vt[e].x, vt[e].y * vt[e].y);
                st.c[e] = splat(w) * st.p[e];
            }
        } else {
            const float fr = -a.inv_dim2 * tr, fi = -a.inv_dim2 * ti;
#pragma unroll
            for (int e = 0; e < E; e++) {
                st.c[e] = cmul(vt[e], f2{fr, fi});
            }
        }
        {
            float4 rc = rec[G - 1];
            for (int g = G - 1; g >= 0; g--) {
                const float4 rn = rec[g > 0 ? g - 1 : 0];
                const int code = __builtin_amdgcn_readfirstlane(__float_as_int(rc.z));
                float d = op_backward<L>(code & 15, (code >> 4) & 15, (code >> 8) & 15, rc.x, rc.y, st);
                d = wave_sum(d);
                if (lane == 0) gw[wave * G + g] = d;
                rc = rn;
            }
        }
        lds_sync<L::WPS>();

        // ---- gradient read-out / Adam (optax 0.1.1 scale_by_adam + scale(-lr)), owners only
        if (a.mode == MODE_LOSS_GRAD) {
            for (int j = tid; j < P; j += L::THREADS) {
                const int pg = a.pgate[j];
                float g = 0.f;
#pragma unroll
                for (int w = 0; w < L::WPS; w++) g += gw[w * G + pg];
                if (a.use_reg && a.pkind[j]) g += psl[j];
                a.out_grad[s_idx * P + j] = g;
            }
            return;
        }
        const float ibc1 = __builtin_amdgcn_rcpf(bc.x), ibc2 = __builtin_amdgcn_rcpf(bc.y);
        for (int j = tid; j < P; j += L::THREADS) {
            const int pg = a.pgate[j];
            float g = 0.f;
#pragma unroll
            for (int w = 0; w < L::WPS; w++) g += gw[w * G + pg];
            if (a.use_reg && a.pkind[j]) g += psl[j];
            if (frozen && frozen[j]) continue;
            const float m = fmaf(a.b1, mm[j], a.omb1 * g);
            const float v = fmaf(a.b2, vv[j], a.omb2 * (g * g));
            mm[j] = m; vv[j] = v;
            // hardware sqrt / rcp (1 ulp): the step error is ~1e-8, below the angle's own ulp
            const float t = fmaf(-a.lr, (m * ibc1) * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v * ibc2) + a.eps), th[j]);
            th[j] = t;
            publish(j, t);
        }
        lds_sync<L::WPS>();
    }

    // ---- epilogue: reg recompute, loss = regloss - reg (optimization.py:365-370);
    //      argmin, CZ count, entry-loss flag (cp_utils.py:144-164, 193-198)
    const int bi = (best < first) ? 1 : 0;
    float acc[3] = {0.f, 0.f, 0.f};
    for (int j = tid; j < P; j += L::THREADS) {
        if (a.pkind[j]) {
            const float t0 = a.out_params[(s_idx * 2 + 0) * P + j];
            const float t1 = a.out_params[(s_idx * 2 + 1) * P + j];
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

// ------------------------------------------------------------------ reg(theta) for arbitrary angle rows
// r * sum_i R(theta_i * mask_i) (main.py:563-564), one thread per row: the recomputation
// jit(vmap(vmap(regularization_func))) of optimization.py:365-367 for history rows.
#ifndef CPF_NO_PLAIN_KERNELS          // non-template kernels: defined by one translation unit only (cpflow_amd.hip)
__global__ void k_reg_rows(const KArgs a, const float* theta, int64_t count, float* reg) {
    __shared__ __attribute__((aligned(16))) float tab[PEN_FLOATS + 4];
    penalty_table_init(tab, a, threadIdx.x);
    __syncthreads();
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= count) return;
    float acc = 0.f;
    if (a.use_reg)
        for (int j = 0; j < a.P; j++)
            if (a.pkind[j]) { float sl; acc += penalty_lds(tab, a.pen_kind, theta[row * a.P + j], sl); }
    reg[row] = a.use_reg ? a.r * acc : 0.f;
}

#endif

// ------------------------------------------------------------------ initial angles (jax threefry2x32)
// Restates what jax.random does for  main.py:541-548 / cp_utils.py:31-35 / trigonometric_utils.py:35-38:
//   keys = split(PRNGKey(seed), S+1)[1:];  angles[j] = uniform(split(keys[j])[1], (P,), 0, 2*pi)
__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
__device__ __forceinline__ void threefry2x32(uint32_t k0, uint32_t k1, uint32_t& x0, uint32_t& x1) {
    const uint32_t ks0 = k0, ks1 = k1, ks2 = k0 ^ k1 ^ 0x1BD11BDAu;
    x0 += ks0; x1 += ks1;
#define CPF_TF_R(r) { x0 += x1; x1 = rotl32(x1, r); x1 ^= x0; }
    CPF_TF_R(13) CPF_TF_R(15) CPF_TF_R(26) CPF_TF_R(6)   x0 += ks1; x1 += ks2 + 1u;
    CPF_TF_R(17) CPF_TF_R(29) CPF_TF_R(16) CPF_TF_R(24)  x0 += ks2; x1 += ks0 + 2u;
    CPF_TF_R(13) CPF_TF_R(15) CPF_TF_R(26) CPF_TF_R(6)   x0 += ks0; x1 += ks1 + 3u;
    CPF_TF_R(17) CPF_TF_R(29) CPF_TF_R(16) CPF_TF_R(24)  x0 += ks1; x1 += ks2 + 4u;
    CPF_TF_R(13) CPF_TF_R(15) CPF_TF_R(26) CPF_TF_R(6)   x0 += ks2; x1 += ks0 + 5u;
#undef CPF_TF_R
}
// element i of threefry_2x32(key, iota(n)) (counts padded to even, split in halves)
__device__ __forceinline__ uint32_t random_bits_at(uint32_t k0, uint32_t k1, uint32_t n, uint32_t i) {
    const uint32_t h = (n + 1u) >> 1, lo = (i < h) ? i : i - h, hi = lo + h;
    uint32_t x0 = lo, x1 = (hi < n) ? hi : 0u;
    threefry2x32(k0, k1, x0, x1);
    return (i < h) ? x0 : x1;
}
// jax.random.normal (jax 0.3.x _normal_real): u = uniform(key, minval = nextafter(-1, 0), maxval = 1), sqrt(2) * erf_inv(u).
// erf_inv restates XLA's single-precision ErfInv (Giles' polynomial in w = -log((1 - x)(1 + x))): third-party arithmetic, and
// no result file of the reference uses cp_distribution = 'normal', so this branch is "parity unpinned" (DESIGN.md).
__device__ __forceinline__ float erfinv_xla_f32(float x) {
    float w = -log1pf(-x * x);
    const bool lt = w < 5.0f;
    w = lt ? w - 2.5f : sqrtf(w) - 3.0f;
    float p = lt ? 2.81022636e-08f : -0.000200214257f;
    p = fmaf(p, w, lt ? 3.43273939e-07f : 0.000100950558f);
    p = fmaf(p, w, lt ? -3.5233877e-06f : 0.00134934322f);
    p = fmaf(p, w, lt ? -4.39150654e-06f : -0.00367342844f);
    p = fmaf(p, w, lt ? 0.00021858087f : 0.00573950773f);
    p = fmaf(p, w, lt ? -0.00125372503f : -0.0076224613f);
    p = fmaf(p, w, lt ? -0.00417768164f : 0.00943887047f);
    p = fmaf(p, w, lt ? 0.246640727f : 1.00167406f);
    p = fmaf(p, w, lt ? 1.50140941f : 2.83297682f);
    return p * x;
}
// cp_mode: 0 'uniform', 1 '0' (CP angles zero), 2 'normal' (CP angles 1.5 * normal(split(key')[1]), cp_utils.py:38-40)
#ifndef CPF_NO_PLAIN_KERNELS
__global__ void k_initial_angles(uint32_t seed, uint32_t s_total, uint32_t first, uint32_t count, uint32_t P,
                                 const uint8_t* pkind, int cp_mode, float* out) {
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (uint64_t)count * P) return;
    const uint32_t j = (uint32_t)(idx / P), i = (uint32_t)(idx % P);
    const uint32_t nkeys = 2u * (s_total + 1u), row = first + j + 1u;
    const uint32_t ka = random_bits_at(0u, seed, nkeys, 2u * row), kb = random_bits_at(0u, seed, nkeys, 2u * row + 1u);
    const uint32_t s0 = random_bits_at(ka, kb, 4u, 2u), s1 = random_bits_at(ka, kb, 4u, 3u);
    const uint32_t bits = random_bits_at(s0, s1, P, i);
    float f = __uint_as_float((bits >> 9) | 0x3F800000u) - 1.0f;
    f = __fmul_rn(f, 6.2831855f);
    f = fmaxf(0.f, f);
    if (cp_mode == 1 && pkind[i]) f = 0.f;
    if (cp_mode == 2 && pkind[i]) {
        // key, subkey = split(key) a second time: key = split(k)[0], subkey = split(key)[1]
        const uint32_t n0 = random_bits_at(ka, kb, 4u, 0u), n1 = random_bits_at(ka, kb, 4u, 1u);
        const uint32_t t0 = random_bits_at(n0, n1, 4u, 2u), t1 = random_bits_at(n0, n1, 4u, 3u);
        const uint32_t nb = random_bits_at(t0, t1, P, i);
        const float lo = -0.99999994f;                                  // nextafter(-1, 0); maxval - minval rounds to 2
        float u = __fadd_rn(__fmul_rn(__uint_as_float((nb >> 9) | 0x3F800000u) - 1.0f, 2.0f), lo);
        u = fmaxf(lo, u);
        f = __fmul_rn(1.5f, __fmul_rn(1.41421354f, erfinv_xla_f32(u)));
    }
    out[idx] = f;
}

#endif

}  // namespace cpf
