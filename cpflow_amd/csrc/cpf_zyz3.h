// cpflow_amd: the two-ended ZYZ-form kernel for 3 qubits - BASELINE configs 1-3 (round 5).
//
// What it replaces, per optimiser step and sample (cpflow/main.py:69-82,106-146 build_unitary, matrix_utils.py:35-42 cost_HST,
// penalty.py:44-71, optimization.py:14-25,61-75,340): the work of cpf_split.h, in the algebra of cpf_zyz_math.h.
//
// One amplitude of the 8 x 8 unitary per lane (Lay<3, 0>: every row qubit on a DPP-reachable lane bit), two waves per sample:
//   wave A   psi(s_b), b = -1 .. M-1: the (flipped) product state of the surface gates, then blocks 0 .. M-1 applied;
//            lanes 2 q, 2 q + 1 OWN the surface angles of qubit q, lanes 0 .. K-1 evaluate the penalty of the CP angles
//   wave B   W(s_b), b = K-1 .. M-1: Vd = conj(dfin) V, then blocks K-1 .. M un-applied;
//            side w = 2 b + s of block b (its three rotation angles and, side 0, the CP angle) is OWNED by lanes 2 w, 2 w + 1
//            (up to 16 blocks; by lane w alone beyond that and with two samples per workgroup)
// with s_b the slice right after the two real stages of block b (cpf_zyz_math.h: what the sweeps hold there is the true state
// up to pending flips and the left diagonals not yet merged).  A block is ONE complex multiply by the lane's merged phase and two
// stages x += (+-t) x[partner] (two v_fmac_f32_dpp each): 6 vector instructions where the gate-by-gate sweep of cpf_split.h spends
// 14 (four x / y rotations at 2 DPP + 1 packed, one merged diagonal).  Every state of the forward halves stays in the registers of
// the wave that made it; LDS sees the two states at the cut only.  The gradient halves then move ONE matrix: wave A un-applies
// blocks M-1 .. 0 from the cotangent w = f W with psi(s_b) at hand, wave B applies blocks M .. K-1 to psi' = conj(f) psi with
// W(s_b) at hand - the lane's own value from its register, the two partner lanes' by four v_mov_b32_dpp from theirs (no longer the
// sweep's live value, so the moves wait for nothing and fill the sweep's dependency stalls; reading them back from stored states
// cost a 512-byte LDS write per block in the forward halves and two reads per block under the reduction: config 2 4.88 -> 4.76 ms).
// At a slice the 7 generator sums of cpf_zyz_math.h come from three products per lane:
//     conj(w) psi            -> Pz of the block's two qubits (signed by the lane's bits), the |11> sum of the NEXT block's pair
//     conj(w) psi[partner]   -> Px (imaginary part), Py (real part, signed), once per qubit of the block
// (tools/zyz3_model.py is the executable specification, checked against tools/zyz_model.py and central differences).
//
// The sums do not cross the lanes in the sweeps at all: every lane writes its 7 shares to LDS (one 272-byte row per sum: 64 lanes
// + padding, conflict-free both ways), and each wave then adds ITS OWN rows with one lane per row - 16 ds_read_b128 and 33 packed
// adds per 64 sums where the DPP reduce-scatter of cpf_blocks.h costs 28 issue slots per block inside the sweep.
//
// A step is a dependent chain - angles -> tables -> sweeps -> sums -> gradients -> angles - and one wave's stream of dependent
// instructions issues at 5 - 9 clocks each (tools/ubench/issue_cost.hip), so what counts is the LONGER of the two waves' streams
// between barriers, not the instructions issued:
//   * the side lanes build their block's table themselves: two sides of a block are neighbouring lanes, so a quad_perm exchange of
//     the e-pairs replaces a trip through LDS, and each lane writes two of the four entries a sweep lane reads with one
//     ds_read_b128: (d.re, d.im, signed t_up, signed t_dn).  The common phase of a block's entries is left in (no loss sees it);
//   * the flip bits of all sides are ONE ballot (all sides sit in wave B), pending parities a popcount against a per-lane mask;
//   * T = sum conj(W) psi over the 64 lanes: six DPP stages and a v_readlane per component, both waves, identical arithmetic.  Two
//     v_mfma_f32_16x16x4_f32 against a register of ones (the matrix core as an adder tree, CPF_Z3_MFMA=1) are shorter for a wave on
//     its own and longer where two waves share a SIMD and its matrix pipe: config 2 4.65 against 4.61 ms, config 3's share 5.13
//     against 5.01, 256 samples 4.10 against 4.14;
//   * the penalty sum is K broadcast LDS reads in each wave (no cross-lane instruction);
//   * two lanes per item: the even lane owns angle slots 0 and 1, the odd lane slots 2 and 3, so that Adam and the sin / cos
//     polynomials - the longest stretch of both owner phases - run on one packed pair per lane instead of two; the even lane fetches
//     the other pair's (cos, sin) by DPP, does the rest as the item's main lane and hands two gradients back (config 2 4.72 -> 4.54 ms,
//     config 3's share 5.02 -> 4.78, same box; halving the work of one wave alone does nothing: 4.58 -> 4.57 / 4.54 as prototypes);
//   * everything that is not a block's side belongs to wave A, whose owner phase runs beside wave B's: the surface gates (chain
//     rule, Adam, sin / cos, the two variants S and F S of each surface matrix - wave B only publishes the three flip parities) and
//     the penalty of the CP angles, which wave B publishes right after its Adam update together with a step counter in LDS that
//     wave A waits for (config 2, same box: 5.2 -> 4.9 ms once the cut moved from 5 to 4 blocks: wave A's share of the sweeps
//     shrinks by what its owner phase grew).
// Both waves form the loss with the same instructions and track the best iterate for the angles they own.
//
// No branch on the lane may sit next to the step loop's back edge: with `if (lane < 3) { store }` as the last statement of wave A's
// loop the compiler split the loop by lanes (irreducible control flow repaired with a guard block), and the s_barrier at the loop's
// top then ran once per group of lanes - the two waves lost step, silently (losses off by 1e-2 from the second step on, for some
// cuts only).  Wave A's owner phase therefore sits at the top of the loop and stores without a branch (lanes that own nothing write
// to rows of shares that are dead at that point).
//
// Requires: n = 3, 1..3 rotation letters (any), CP or CZ blocks, loss kinds with a linear cotangent (HST, state preparation),
// 2 K <= 64 sides, K and the layer compile-time constants (instances: ahead of time for configs 2 / 3, cpf_jit.h otherwise).
#pragma once
#include "cpf_layered.h"
#include "cpf_zyz_math.h"

namespace cpf {

typedef __attribute__((address_space(3))) float z3_lds_f;     // a float in LDS (32-bit pointers: what the per-block store bases are kept as)
constexpr int ZYZ3_ROW = 68;                     // floats per row of shares: 64 lanes + 4 (16 lanes of one ds_read_b128 pass hit 16 different bank groups)

// floats of LDS per sample of an instance with K blocks (Zyz3Lds<K>::TOTAL)
__host__ __device__ constexpr int zyz3_lds_floats(int K) {
    const int rows = 10 + 7 * K, srows = 7 + 5 * K;
    const int pv = PEN_FLOATS + 8 * K + 4, tab = pv + 3 * ((K + 3) & ~3), surfm = tab + 16 * K, tot = surfm + 48 + 24, states = (tot + rows + 1 + 3) & ~3;
    return states + 128 * 2 + 128 + srows * ZYZ3_ROW;
}
__host__ __device__ constexpr bool zyz3_supported(int K, int n_rot) { return K >= 2 && 2 * K <= 64 && n_rot >= 1 && n_rot <= 3; }

template <int K>
struct Zyz3Lds {                                  // float offsets, per sample; [0, 28) = penalty table
    static constexpr int up4(int x) { return (x + 3) & ~3; }
    static constexpr int ROWS = 10 + 7 * K;       // sums: 9 of the surface slice, S11 of block 0, then 7 per block (Px Py Pz up, Px Py Pz dn, S11 of the next block)
    static constexpr int SROWS = 7 + 5 * K;       // rows of shares: Px Py of the three qubits and u at the surface slice, then 5 per block (Px Py up, Px Py dn, u)
    static constexpr int REC1 = PEN_FLOATS;       // (2 K + 1) x {l0, l1}: left factors of a side; the last entry is the identity ("no previous side")
    static constexpr int PV = REC1 + 8 * K + 4;   // K penalty values (padded to a multiple of 4)
    static constexpr int PTH = PV + up4(K);       // the K CP angles (wave B -> wave A, which evaluates the penalty)
    static constexpr int PSL = PTH + up4(K);      // r x the penalty's slope at each of them (wave A -> wave B's next Adam step)
    static constexpr int TAB = PSL + up4(K);      // K x 4 entries x {d.re, d.im, t_up signed, t_dn signed}
    static constexpr int SURFM = TAB + 16 * K;    // 3 x {S, F S} x 2x2 complex: both variants of each surface matrix
    static constexpr int MSF = SURFM + 48;        // the three flip parities (floats 0 / 1); [3] the step counter wave A waits for;
                                                  // [4] "this step's iterate is the best so far" (wave B decides, wave A follows); [7] scratch
                                                  // [5], [6] the barrier counts of the sample's waves A and B (compared after the last step)
    static constexpr int CPT = MSF + 8;           // 4 x {k_t, k_u, k_d, -}: the CP angle's chain rule by (ca | cb << 1) (zyz::chain_cp as a table)
    static constexpr int TOT = CPT + 16;          // the ROWS sums, one spare slot behind them for stores that mean nothing
    static constexpr int STATES = up4(TOT + ROWS + 1);       // 2 x 64 x (re, im): the two states at the cut, psi(s_{M-1}) and W(s_{M-1})
    static constexpr int SCR = STATES + 128 * 2;             // 2 x 64 floats nobody reads: where the row sums put what a row does not produce
    static constexpr int PART = SCR + 128;                   // SROWS x ZYZ3_ROW
    static constexpr int TOTAL = PART + SROWS * ZYZ3_ROW;
    static constexpr int row_blk(int b) { return 10 + 7 * b; }
    static constexpr int srow_blk(int b) { return 7 + 5 * b; }
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
// conj(d) * x twice: o stays (the state the gradient half reads), w is what the next block's in-place stages overwrite - one
// packed instruction more instead of the two v_mov_b32 a copy costs, and one link less in the sweep's dependent chain
__device__ __forceinline__ void z3_cmul_conj2(f2 d, f2 x, f2& o, f2& w) {
    asm("v_pk_mul_f32 %0, %2, %3 op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %1, %2, %3, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[1,0,0]\n\t"
        "v_pk_fma_f32 %0, %2, %3, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[1,0,0]" : "=&v"(o), "=&v"(w) : "v"(d), "v"(x));
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
// Both stages of a block in ONE statement: first on row qubit QA, then on QB.  (Between two asm statements the compiler puts a wait
// state of its own: it cannot see what the first one wrote.)  Wait states: two between the vector write of x - the complex multiply
// in front - and the first DPP read; one s_nop 0 between the stages (each half then has two instructions between its write and its
// next DPP read).
template <int QA, int QB, bool NEG, int LEAD = 2>
__device__ __forceinline__ void z3_stage2(f2& x, float ta, float tb) {
    static_assert(QA >= 0 && QA < 3 && QB >= 0 && QB < 3 && QA != QB, "row qubits 0 .. 2 sit on lane bits 0, 1, 3");
#define CPF_Z3_CTRL(Q) (Q == 0 ? "quad_perm:[1,0,3,2]" : (Q == 1 ? "quad_perm:[2,3,0,1]" : "row_ror:8"))
#define CPF_Z3_ST2(CA, CB, SGN, LEADNOP)                                                                                       \
    asm(LEADNOP "v_fmac_f32_dpp %0, %0, " SGN "%2 " CA " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                         \
        "v_fmac_f32_dpp %1, %1, " SGN "%2 " CA " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                                 \
        "s_nop 0\n\tv_fmac_f32_dpp %0, %0, " SGN "%3 " CB " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                    \
        "v_fmac_f32_dpp %1, %1, " SGN "%3 " CB " row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x.x), "+v"(x.y) : "v"(ta), "v"(tb));
#define CPF_Z3_ST2L(CA, CB, SGN) if constexpr (LEAD >= 2) { CPF_Z3_ST2(CA, CB, SGN, "s_nop 1\n\t") } else if constexpr (LEAD == 1) { CPF_Z3_ST2(CA, CB, SGN, "s_nop 0\n\t") } else { CPF_Z3_ST2(CA, CB, SGN, "") }
#define CPF_Z3_ST2S(CA, CB) if constexpr (NEG) { CPF_Z3_ST2L(CA, CB, "-") } else { CPF_Z3_ST2L(CA, CB, "") }
    if constexpr (QA == 0 && QB == 1) { CPF_Z3_ST2S("quad_perm:[1,0,3,2]", "quad_perm:[2,3,0,1]") }
    else if constexpr (QA == 0 && QB == 2) { CPF_Z3_ST2S("quad_perm:[1,0,3,2]", "row_ror:8") }
    else if constexpr (QA == 1 && QB == 0) { CPF_Z3_ST2S("quad_perm:[2,3,0,1]", "quad_perm:[1,0,3,2]") }
    else if constexpr (QA == 1 && QB == 2) { CPF_Z3_ST2S("quad_perm:[2,3,0,1]", "row_ror:8") }
    else if constexpr (QA == 2 && QB == 0) { CPF_Z3_ST2S("row_ror:8", "quad_perm:[1,0,3,2]") }
    else { CPF_Z3_ST2S("row_ror:8", "quad_perm:[2,3,0,1]") }
#undef CPF_Z3_ST2S
#undef CPF_Z3_ST2L
#undef CPF_Z3_ST2
#undef CPF_Z3_CTRL
}
// Both stages of a block on the live matrix x (row qubits QA then QB; NEG: un-applied) AND four partner values of a STORED state o
// that the products of this block (wave B) or of the next one (wave A) need: the moves read registers nobody is writing, so they
// stand where the stages' wait states would be - no s_nop in the block.  The caller guarantees two wait states between the last
// vector write of x and this statement's third instruction (the complex multiply in front and the two moves do).
//   pre = (o.re at the partner on FA, o.re at the partner on FB), pim likewise for o.im
#define CPF_Z3_C0 "quad_perm:[1,0,3,2]"
#define CPF_Z3_C1 "quad_perm:[2,3,0,1]"
#define CPF_Z3_C2 "row_ror:8"
#define CPF_Z3_DPPT " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define CPF_Z3_PAIRS6(M, ...)                                                                                               \
    if constexpr (M##A == 0 && M##B == 1) { CPF_Z3_BODY(__VA_ARGS__, CPF_Z3_C0, CPF_Z3_C1) }                                 \
    else if constexpr (M##A == 0 && M##B == 2) { CPF_Z3_BODY(__VA_ARGS__, CPF_Z3_C0, CPF_Z3_C2) }                            \
    else if constexpr (M##A == 1 && M##B == 0) { CPF_Z3_BODY(__VA_ARGS__, CPF_Z3_C1, CPF_Z3_C0) }                            \
    else if constexpr (M##A == 1 && M##B == 2) { CPF_Z3_BODY(__VA_ARGS__, CPF_Z3_C1, CPF_Z3_C2) }                            \
    else if constexpr (M##A == 2 && M##B == 0) { CPF_Z3_BODY(__VA_ARGS__, CPF_Z3_C2, CPF_Z3_C0) }                            \
    else { CPF_Z3_BODY(__VA_ARGS__, CPF_Z3_C2, CPF_Z3_C1) }
template <int QA, int QB, bool NEG, int FA, int FB>
__device__ __forceinline__ void z3_stage2_fetch(f2& x, float ta, float tb, f2 o, f2& pre, f2& pim) {
    static_assert(QA >= 0 && QA < 3 && QB >= 0 && QB < 3 && QA != QB && FA >= 0 && FA < 3 && FB >= 0 && FB < 3 && FA != FB, "row qubits 0 .. 2 sit on lane bits 0, 1, 3");
#define CPF_Z3_BODY(SGN, CA, CB, GA, GB)                                                                                     \
    asm("v_mov_b32_dpp %2, %6 " GA CPF_Z3_DPPT "v_mov_b32_dpp %4, %7 " GA CPF_Z3_DPPT                                        \
        "v_fmac_f32_dpp %0, %0, " SGN "%8 " CA CPF_Z3_DPPT "v_fmac_f32_dpp %1, %1, " SGN "%8 " CA CPF_Z3_DPPT                \
        "v_mov_b32_dpp %3, %6 " GB CPF_Z3_DPPT "v_mov_b32_dpp %5, %7 " GB CPF_Z3_DPPT                                        \
        "v_fmac_f32_dpp %0, %0, " SGN "%9 " CB CPF_Z3_DPPT "v_fmac_f32_dpp %1, %1, " SGN "%9 " CB " row_mask:0xf bank_mask:0xf bound_ctrl:1" \
        : "+v"(x.x), "+v"(x.y), "=&v"(pre.x), "=&v"(pre.y), "=&v"(pim.x), "=&v"(pim.y) : "v"(o.x), "v"(o.y), "v"(ta), "v"(tb));
#define CPF_Z3_STAGE_PAIRS(SGN)                                                                                              \
    if constexpr (QA == 0 && QB == 1) { CPF_Z3_PAIRS6(F, SGN, CPF_Z3_C0, CPF_Z3_C1) }                                        \
    else if constexpr (QA == 0 && QB == 2) { CPF_Z3_PAIRS6(F, SGN, CPF_Z3_C0, CPF_Z3_C2) }                                   \
    else if constexpr (QA == 1 && QB == 0) { CPF_Z3_PAIRS6(F, SGN, CPF_Z3_C1, CPF_Z3_C0) }                                   \
    else if constexpr (QA == 1 && QB == 2) { CPF_Z3_PAIRS6(F, SGN, CPF_Z3_C1, CPF_Z3_C2) }                                   \
    else if constexpr (QA == 2 && QB == 0) { CPF_Z3_PAIRS6(F, SGN, CPF_Z3_C2, CPF_Z3_C0) }                                   \
    else { CPF_Z3_PAIRS6(F, SGN, CPF_Z3_C2, CPF_Z3_C1) }
    if constexpr (NEG) { CPF_Z3_STAGE_PAIRS("-") } else { CPF_Z3_STAGE_PAIRS("") }
#undef CPF_Z3_STAGE_PAIRS
#undef CPF_Z3_BODY
}
// the four partner values alone (a stored state: no wait states needed)
template <int FA, int FB>
__device__ __forceinline__ void z3_fetch(f2 o, f2& pre, f2& pim) {
    static_assert(FA >= 0 && FA < 3 && FB >= 0 && FB < 3 && FA != FB, "row qubits 0 .. 2 sit on lane bits 0, 1, 3");
#define CPF_Z3_BODY(GA, GB)                                                                                                  \
    asm("v_mov_b32_dpp %0, %4 " GA CPF_Z3_DPPT "v_mov_b32_dpp %2, %5 " GA CPF_Z3_DPPT                                        \
        "v_mov_b32_dpp %1, %4 " GB CPF_Z3_DPPT "v_mov_b32_dpp %3, %5 " GB " row_mask:0xf bank_mask:0xf bound_ctrl:1"         \
        : "=&v"(pre.x), "=&v"(pre.y), "=&v"(pim.x), "=&v"(pim.y) : "v"(o.x), "v"(o.y));
    if constexpr (FA == 0 && FB == 1) { CPF_Z3_BODY(CPF_Z3_C0, CPF_Z3_C1) }
    else if constexpr (FA == 0 && FB == 2) { CPF_Z3_BODY(CPF_Z3_C0, CPF_Z3_C2) }
    else if constexpr (FA == 1 && FB == 0) { CPF_Z3_BODY(CPF_Z3_C1, CPF_Z3_C0) }
    else if constexpr (FA == 1 && FB == 2) { CPF_Z3_BODY(CPF_Z3_C1, CPF_Z3_C2) }
    else if constexpr (FA == 2 && FB == 0) { CPF_Z3_BODY(CPF_Z3_C2, CPF_Z3_C0) }
    else { CPF_Z3_BODY(CPF_Z3_C2, CPF_Z3_C1) }
#undef CPF_Z3_BODY
}
// the three row-qubit partners of a value an asm statement has just written (two wait states in front)
__device__ __forceinline__ void z3_partner3(f2 v, f2& p0, f2& p1, f2& p2) {
    asm("s_nop 1\n\t"
        "v_mov_b32_dpp %0, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_mov_b32_dpp %1, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_mov_b32_dpp %2, %6 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_mov_b32_dpp %3, %7 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_mov_b32_dpp %4, %6 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_mov_b32_dpp %5, %7 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1"
        : "=&v"(p0.x), "=&v"(p0.y), "=&v"(p1.x), "=&v"(p1.y), "=&v"(p2.x), "=&v"(p2.y) : "v"(v.x), "v"(v.y));
}
// One block of wave B's forward half in ONE statement: both stages un-applied from the working copy, which lives in the FIXED
// registers v[2:3] (the only way to name a register pair AND its halves in one asm statement: the DPP stages take the halves, the
// packed multiply the pair), then conj(d) x once into o (kept for the gradient half) and once back into the working copy.  One
// statement per block instead of two: between two dependent asm statements the compiler inserts a wait state of its own, and the
// tied halves of a free register pair made its allocator copy halves around (2 v_mov_b32 in every other block of round 6's first
// listing).  The previous statement wrote v[2:3] with its last instruction: two wait states in front.
template <int QA, int QB, bool LAST>
__device__ __forceinline__ f2 z3_unapply_block(float& L0, float& L1, f2 d, float ta, float tb) {
    static_assert(QA >= 0 && QA < 3 && QB >= 0 && QB < 3 && QA != QB, "row qubits 0 .. 2 sit on lane bits 0, 1, 3");
    f2 o, t;
#define CPF_Z3_BODY(TAIL, CA, CB)                                                                                            \
    asm("s_nop 1\n\t"                                                                                                        \
        "v_fmac_f32_dpp v2, v2, -%5 " CA CPF_Z3_DPPT "v_fmac_f32_dpp v3, v3, -%5 " CA CPF_Z3_DPPT                             \
        "s_nop 0\n\t"                                                                                                        \
        "v_fmac_f32_dpp v2, v2, -%6 " CB CPF_Z3_DPPT "v_fmac_f32_dpp v3, v3, -%6 " CB CPF_Z3_DPPT                             \
        "v_pk_mul_f32 %1, %4, v[2:3] op_sel_hi:[0,1]\n\t"                                                                    \
        "v_pk_fma_f32 %0, %4, v[2:3], %1 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[1,0,0]" TAIL                               \
        : "=&v"(o), "=&v"(t), "+{v2}"(L0), "+{v3}"(L1) : "v"(d), "v"(ta), "v"(tb));
#define CPF_Z3_TAILED(CA, CB)                                                                                                \
    if constexpr (LAST) { CPF_Z3_BODY("", CA, CB) }                                                                          \
    else { CPF_Z3_BODY("\n\tv_pk_fma_f32 v[2:3], %4, v[2:3], %1 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[1,0,0]", CA, CB) }
    if constexpr (QA == 0 && QB == 1) { CPF_Z3_TAILED(CPF_Z3_C0, CPF_Z3_C1) }
    else if constexpr (QA == 0 && QB == 2) { CPF_Z3_TAILED(CPF_Z3_C0, CPF_Z3_C2) }
    else if constexpr (QA == 1 && QB == 0) { CPF_Z3_TAILED(CPF_Z3_C1, CPF_Z3_C0) }
    else if constexpr (QA == 1 && QB == 2) { CPF_Z3_TAILED(CPF_Z3_C1, CPF_Z3_C2) }
    else if constexpr (QA == 2 && QB == 0) { CPF_Z3_TAILED(CPF_Z3_C2, CPF_Z3_C0) }
    else { CPF_Z3_TAILED(CPF_Z3_C2, CPF_Z3_C1) }
#undef CPF_Z3_TAILED
#undef CPF_Z3_BODY
    return o;
}
// The four products behind a block's Px / Py shares, halves of the live matrix v selected by op_sel (the compiler builds the
// broadcast of v.y with a v_mov_b32 and then splits the register pair: two moves per block in round 5's listing):
//   A (wave A: v = the cotangent w, partners of psi):  px =   v.x pim - v.y pre,  py = -(v.x pre + v.y pim)
//   B (wave B: v = psi', partners of W):               px =   v.y pre - v.x pim,  py =   v.x pre + v.y pim
template <bool WAVE_B>
__device__ __forceinline__ void z3_products(f2 v, f2 pre, f2 pim, f2& px, f2& py) {
    if constexpr (!WAVE_B)
        asm("v_pk_mul_f32 %0, %2, %3 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[1,0]\n\t"
            "v_pk_mul_f32 %1, %2, %4 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[1,0]\n\t"
            "v_pk_fma_f32 %0, %2, %4, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
            "v_pk_fma_f32 %1, %2, %3, %1 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]"
            : "=&v"(px), "=&v"(py) : "v"(v), "v"(pre), "v"(pim));
    else
        asm("v_pk_mul_f32 %0, %2, %4 op_sel:[0,0] op_sel_hi:[0,1] neg_lo:[1,0] neg_hi:[1,0]\n\t"
            "v_pk_mul_f32 %1, %2, %4 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
            "v_pk_fma_f32 %0, %2, %3, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
            "v_pk_fma_f32 %1, %2, %3, %1 op_sel:[0,0,0] op_sel_hi:[0,1,1]"
            : "=&v"(px), "=&v"(py) : "v"(v), "v"(pre), "v"(pim));
}
// ---- A whole block of the gradient half in ONE statement (round 6, second pass).  Between two dependent asm statements the compiler
// puts an s_nop 0 of its own (it cannot see what the first one wrote): a block made of four statements carried two of them on 19
// instructions.  The live matrix sits in the FIXED registers v[2:3] and the partner values in v[4:5] (re) / v[6:7] (im), so that one
// statement can use the halves (DPP) and the pairs (packed) of the same registers (see z3_unapply_block).
// Launches whose workgroups are all resident run 2048 waves in lock-step through phases of very different density (owner phases: a few
// lanes; sweeps: all of them), and the chip's clock follows the current swing (section "Round 6" of DESIGN.md).  STAGGER = G > 1: the
// workgroups start their step loops G-fold out of phase (group g sleeps g / G of a step once, in front of the loop), so that dense
// and sparse phases overlap across the chip - and on a SIMD shared by a wave A and a wave B of different samples.  Measured: whether
// the groups really sleep makes no difference (bit on / off: equal to 0.1 %), so the clock does not follow the swing but the mean power;
// what the few instructions in front of the loops do change is the generated code, slightly (709 + 516 instructions in the step loops
// instead of 707 + 512, a second exit) - and that is worth a steady 0.5 %, unexplained and reproducible on four boxes, with
// the four-statement gradient blocks on three boxes (4.00 - 4.02 against 4.02 - 4.04 ms) and -1 % with the fused ones.  Kept for that.
#ifndef CPF_Z3_STAGGER
#define CPF_Z3_STAGGER 4         // groups (0 / 1: none); the host asks for it with bit 1 of KArgs::prio (bit 0: wave B's issue priority)
#endif
#ifndef CPF_Z3_STAGGER_SHIFT
#define CPF_Z3_STAGGER_SHIFT 3   // group = (workgroup >> SHIFT) % G: workgroups go round the 8 XCDs first
#endif
#ifndef CPF_Z3_STAGGER_STEP
#define CPF_Z3_STAGGER_STEP 76   // one step in units of 64 clocks (config 2: 4900 clocks)
#endif
// Measured (profiles/r06_cfg2_levers_ab.txt section 13): 707 + 512 -> 691 + 505 instructions per sample-step and NOT faster - config 2's
// launch is power-limited: the fused form's best launches take 3.96 ms, but its sustained mean over 150 launches is 4.06 - 4.08 (the
// clock dips in a third of them) against 4.02 - 4.04 steady for the four-statement form, on three boxes.  Off by default.
#ifndef CPF_Z3_FUSED
#define CPF_Z3_FUSED 0           // 1 = one statement per gradient block, 0 = the four statements per block of the first pass
#endif
#ifndef CPF_Z3_FUSED_A
#define CPF_Z3_FUSED_A CPF_Z3_FUSED
#endif
#ifndef CPF_Z3_FUSED_B
#define CPF_Z3_FUSED_B CPF_Z3_FUSED
#endif
// wave A: shares of block B from the cotangent w = v[2:3] and psi(s_B)'s partner values v[4:7]; U = conj(w) o (its .y is the |11> share);
// then (unless LAST) the partner values of on = psi(s_{B-1}) on the NEXT block's qubits (FA, FB) arrive while the block's stages
// (QA then QB, un-applied: -t) run on w; then w <- conj(d) w.  No wait state needed: w was last written six instructions before its
// first DPP read, and two moves stand between the stages.
template <int QA, int QB, int FA, int FB, bool LAST>
__device__ __forceinline__ void z3_grad_block_a(float& w0, float& w1, float& pr0, float& pr1, float& pi0, float& pi1, f2 o, f2 on, f2 d, float ta, float tb,
                                                f2& PX, f2& PY, f2& U) {
    f2 t;
#define CPF_Z3_GA_HEAD                                                                                                         \
        "v_pk_mul_f32 %0, v[2:3], v[4:5] op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[1,0]\n\t"                           \
        "v_pk_mul_f32 %1, v[2:3], v[6:7] op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[1,0]\n\t"                           \
        "v_pk_fma_f32 %0, v[2:3], v[6:7], %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"                                             \
        "v_pk_fma_f32 %1, v[2:3], v[4:5], %1 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"               \
        "v_pk_mul_f32 %2, v[2:3], %10 op_sel_hi:[0,1]\n\t"                                                                     \
        "v_pk_fma_f32 %2, v[2:3], %10, %2 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[1,0,0]\n\t"
#define CPF_Z3_GA_TAIL                                                                                                         \
        "v_pk_mul_f32 %3, %13, v[2:3] op_sel_hi:[0,1]\n\t"                                                                     \
        "v_pk_fma_f32 v[2:3], %13, v[2:3], %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[1,0,0]"
#define CPF_Z3_GA_OPS                                                                                                          \
        : "=&v"(PX), "=&v"(PY), "=&v"(U), "=&v"(t), "+{v2}"(w0), "+{v3}"(w1), "+{v4}"(pr0), "+{v5}"(pr1), "+{v6}"(pi0), "+{v7}"(pi1)  \
        : "v"(o), "v"(on.x), "v"(on.y), "v"(d), "v"(ta), "v"(tb)
#define CPF_Z3_BODY(CA, CB, GA, GB)                                                                                            \
    if constexpr (LAST)                                                                                                        \
        asm(CPF_Z3_GA_HEAD                                                                                                     \
            "v_fmac_f32_dpp v2, v2, -%14 " CA CPF_Z3_DPPT "v_fmac_f32_dpp v3, v3, -%14 " CA CPF_Z3_DPPT "s_nop 0\n\t"          \
            "v_fmac_f32_dpp v2, v2, -%15 " CB CPF_Z3_DPPT "v_fmac_f32_dpp v3, v3, -%15 " CB CPF_Z3_DPPT                        \
            CPF_Z3_GA_TAIL CPF_Z3_GA_OPS);                                                                                     \
    else                                                                                                                       \
        asm(CPF_Z3_GA_HEAD                                                                                                     \
            "v_mov_b32_dpp v4, %11 " GA CPF_Z3_DPPT "v_mov_b32_dpp v6, %12 " GA CPF_Z3_DPPT                                    \
            "v_fmac_f32_dpp v2, v2, -%14 " CA CPF_Z3_DPPT "v_fmac_f32_dpp v3, v3, -%14 " CA CPF_Z3_DPPT                        \
            "v_mov_b32_dpp v5, %11 " GB CPF_Z3_DPPT "v_mov_b32_dpp v7, %12 " GB CPF_Z3_DPPT                                    \
            "v_fmac_f32_dpp v2, v2, -%15 " CB CPF_Z3_DPPT "v_fmac_f32_dpp v3, v3, -%15 " CB CPF_Z3_DPPT                        \
            CPF_Z3_GA_TAIL CPF_Z3_GA_OPS);
    if constexpr (QA == 0 && QB == 1) { CPF_Z3_PAIRS6(F, CPF_Z3_C0, CPF_Z3_C1) }
    else if constexpr (QA == 0 && QB == 2) { CPF_Z3_PAIRS6(F, CPF_Z3_C0, CPF_Z3_C2) }
    else if constexpr (QA == 1 && QB == 0) { CPF_Z3_PAIRS6(F, CPF_Z3_C1, CPF_Z3_C0) }
    else if constexpr (QA == 1 && QB == 2) { CPF_Z3_PAIRS6(F, CPF_Z3_C1, CPF_Z3_C2) }
    else if constexpr (QA == 2 && QB == 0) { CPF_Z3_PAIRS6(F, CPF_Z3_C2, CPF_Z3_C0) }
    else { CPF_Z3_PAIRS6(F, CPF_Z3_C2, CPF_Z3_C1) }
#undef CPF_Z3_BODY
#undef CPF_Z3_GA_OPS
#undef CPF_Z3_GA_TAIL
#undef CPF_Z3_GA_HEAD
}
// wave B: psi' = v[2:3] <- d psi', the partner values of o = W(s_B) on the block's qubits arrive in v[4:7] while its stages (QA then QB)
// run on psi' (the two moves in front of each stage are its wait states), then the four products and U = conj(o) psi'.
template <int QA, int QB>
__device__ __forceinline__ void z3_grad_block_b(float& p0, float& p1, f2 d, f2 o, float ta, float tb, f2& PX, f2& PY, f2& U) {
    f2 t;
    float q4, q5, q6, q7;                                      // v[4:7]: the partner values, dead after the statement
#define CPF_Z3_BODY(CA, CB)                                                                                                    \
    asm("v_pk_mul_f32 %3, %10, v[2:3] op_sel_hi:[0,1]\n\t"                                                                      \
        "v_pk_fma_f32 v[2:3], %10, v[2:3], %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"                              \
        "v_mov_b32_dpp v4, %11 " CA CPF_Z3_DPPT "v_mov_b32_dpp v6, %12 " CA CPF_Z3_DPPT                                          \
        "v_fmac_f32_dpp v2, v2, %13 " CA CPF_Z3_DPPT "v_fmac_f32_dpp v3, v3, %13 " CA CPF_Z3_DPPT                                \
        "v_mov_b32_dpp v5, %11 " CB CPF_Z3_DPPT "v_mov_b32_dpp v7, %12 " CB CPF_Z3_DPPT                                          \
        "v_fmac_f32_dpp v2, v2, %14 " CB CPF_Z3_DPPT "v_fmac_f32_dpp v3, v3, %14 " CB CPF_Z3_DPPT                              \
        "v_pk_mul_f32 %0, v[2:3], v[6:7] op_sel:[0,0] op_sel_hi:[0,1] neg_lo:[1,0] neg_hi:[1,0]\n\t"                           \
        "v_pk_mul_f32 %1, v[2:3], v[6:7] op_sel:[1,0] op_sel_hi:[1,1]\n\t"                                                     \
        "v_pk_fma_f32 %0, v[2:3], v[4:5], %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"                                             \
        "v_pk_fma_f32 %1, v[2:3], v[4:5], %1 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"                                             \
        "v_pk_mul_f32 %2, %15, v[2:3] op_sel_hi:[0,1]\n\t"                                                                     \
        "v_pk_fma_f32 %2, %15, v[2:3], %2 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[1,0,0]"                                     \
        : "=&v"(PX), "=&v"(PY), "=&v"(U), "=&v"(t), "+{v2}"(p0), "+{v3}"(p1), "=&{v4}"(q4), "=&{v5}"(q5), "=&{v6}"(q6), "=&{v7}"(q7)  \
        : "v"(d), "v"(o.x), "v"(o.y), "v"(ta), "v"(tb), "v"(o));
    if constexpr (QA == 0 && QB == 1) { CPF_Z3_BODY(CPF_Z3_C0, CPF_Z3_C1) }
    else if constexpr (QA == 0 && QB == 2) { CPF_Z3_BODY(CPF_Z3_C0, CPF_Z3_C2) }
    else if constexpr (QA == 1 && QB == 0) { CPF_Z3_BODY(CPF_Z3_C1, CPF_Z3_C0) }
    else if constexpr (QA == 1 && QB == 2) { CPF_Z3_BODY(CPF_Z3_C1, CPF_Z3_C2) }
    else if constexpr (QA == 2 && QB == 0) { CPF_Z3_BODY(CPF_Z3_C2, CPF_Z3_C0) }
    else { CPF_Z3_BODY(CPF_Z3_C2, CPF_Z3_C1) }
#undef CPF_Z3_BODY
}
// The penalty of one CP angle and its slope (penalty.py:44-71): cpf_device.h penalty_lds (since round 6 the compare-free form this
// kernel introduced serves every kernel).
__device__ __forceinline__ float z3_penalty(const float* tab, int pen_kind, float ang, float& slope) { return penalty_lds(tab, pen_kind, ang, slope); }
// the value held by the partner lane on row qubit Q (lane bits 0, 1, 3); every lane takes part
template <int Q>
__device__ __forceinline__ float z3_partner(float v) {
    constexpr int CTRL = Q == 0 ? 0xB1 : (Q == 1 ? 0x4E : 0x128);       // quad_perm [1,0,3,2] / [2,3,0,1] / row_ror:8
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int Q>
__device__ __forceinline__ int z3_partner(int v) {
    constexpr int CTRL = Q == 0 ? 0xB1 : (Q == 1 ? 0x4E : 0x128);
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
// both halves of a complex value at the partner lane on row qubit Q, wait states included: for values an inline-asm statement has
// just written (the compiler's hazard recogniser knows its own vector instructions only)
template <int Q>
__device__ __forceinline__ f2 z3_partner2(f2 v) {
    f2 o;
    if constexpr (Q == 0) asm("s_nop 1\n\tv_mov_b32_dpp %0, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_mov_b32_dpp %1, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=&v"(o.x), "=&v"(o.y) : "v"(v.x), "v"(v.y));
    else if constexpr (Q == 1) asm("s_nop 1\n\tv_mov_b32_dpp %0, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_mov_b32_dpp %1, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=&v"(o.x), "=&v"(o.y) : "v"(v.x), "v"(v.y));
    else asm("s_nop 1\n\tv_mov_b32_dpp %0, %2 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_mov_b32_dpp %1, %3 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=&v"(o.x), "=&v"(o.y) : "v"(v.x), "v"(v.y));
    return o;
}
// value of lane ^ 1 (quad_perm [1,0,3,2]); every lane takes part
__device__ __forceinline__ float z3_nb(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true)); }
__device__ __forceinline__ int z3_nb(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true); }

// Sum of a complex number over the 64 lanes, the same bits in every lane: the matrix core as an adder tree.
// v_mfma_f32_16x16x4_f32 with B = 1: D[i][j] = sum_k A[lane 16 k + i]; lane (g, j) receives rows i = 4 g .. 4 g + 3 in its four
// result registers.  Lanes are paired first (even lanes carry the real parts, odd lanes the imaginary parts), so one instruction
// serves both components; a second one adds the four lane groups.
#ifndef CPF_Z3_MFMA
#define CPF_Z3_MFMA 0            // A/B: 1 = the matrix core as an adder tree (below), 0 = six DPP stages + v_readlane (cpf_device.h wave_sum2)
#endif
#ifndef CPF_Z3_PRIO
#define CPF_Z3_PRIO 1            // issue priority of wave B in launches whose workgroups are all resident (zyz3_body; 0 = never raised)
#endif
// per-phase issue priorities of the two waves (0 .. 3) in such launches, as three decimal digits: owner / build, forward half, loss +
// gradient half.  A wave of one sample shares its SIMD with the other-role wave of another sample, both in the same phase: the wave the
// pair's barrier waits for should be served first.
#ifndef CPF_Z3_PRIO_A
#define CPF_Z3_PRIO_A 0
#endif
#ifndef CPF_Z3_PRIO_B
#define CPF_Z3_PRIO_B (111 * CPF_Z3_PRIO)
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
    // Round 6: ONE v_permlane32_swap folds the wave's halves - lanes 0..31 then hold re(l) + re(l + 32), lanes 32..63 im(l - 32) +
    // im(l) - so the four row stages and one row_bcast run on ONE register (5 DPP instructions instead of 12), and the totals are
    // read from lanes 31 (re) and 63 (im).  tools/micro/issue_cost2.hip: 61 against 74 cycles per SIMD with two waves on it.
    (void)odd;
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(re), __float_as_uint(im), false, false);
    float s = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf"
        : "+v"(s));
    re = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), 31));
    im = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), 63));
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

#ifndef CPF_Z3_EXECMASK
#define CPF_Z3_EXECMASK 1        // A/B: 1 = only the lanes that own something execute the owner phases (of wave A's 64 lanes 6 own a surface angle
                                 // and K evaluate a penalty, of wave B's 4 K hold a side) - not for speed in cycles but for POWER: the launch
                                 // runs into the board's power limit (~1000 W on this pool: once round 6 had made the step loops denser, sclk
                                 // dipped from 2.4 to 2.0 - 2.2 GHz for three launches in every ten), and idle lanes computing garbage burn
                                 // what useful ones burn.  Same box, 1400 back-to-back launches of config 2: 989 W, mean 4.16 ms (bimodal:
                                 // 4.07 / 4.45) without; 954 W, 4.05 ms steady with (profiles/r06_cfg2_levers_ab.txt).  A plain `if` on the
                                 // lane: setting EXEC by hand around the phases was tried first and is unsafe - the compiler peels the first
                                 // iteration and then computes the main loop's invariants inside the narrowed stretch
#endif
#ifndef CPF_Z3_ROWMASK
#define CPF_Z3_ROWMASK 1         // A/B: 1 = the lanes without a row of shares sit the row sums out (exec mask) instead of adding a row again:
                                 // fewer LDS reads and stores where four samples share a CU's LDS (config 2 4.20 -> 4.13 ms, config 3's share
                                 // 4.18 -> 4.20, same box); the barrier count checks that the step loops stayed whole
#endif
#ifndef CPF_Z3_HWSINCOS
#define CPF_Z3_HWSINCOS 0        // A/B: 1 = (cos, sin) on v_sin_f32 / v_cos_f32 behind a two-constant reduction to [-pi, pi]: 37 vector
                                 // instructions fewer per owner phase and wave, config 2 4.26 -> 4.15 ms, config 3's share 4.32 -> 4.19 (same
                                 // box; round 5's "no gain" was an artefact of its A/B tool) - and NOT adopted: max error 3.7e-7 against
                                 // fast_sincos2's 6e-8 (profiles/r05_hw_sincos_accuracy.txt) is a rounding of every ANGLE at the 1e-7 level,
                                 // which Adam's first steps amplify: short trajectories left the envelope the float32 oracle itself stays
                                 // in (test_short_adam_trajectory_matches_fp32_oracle 5.2e-5 against 3e-5, test_every_shipped_block_count)
#endif
__device__ __forceinline__ void z3_sincos2(f2 x, f2& sn, f2& cn) {
#if CPF_Z3_HWSINCOS
    const f2 t = x * splat(0.15915494309189535f);
    const f2 k = f2{rintf(t.x), rintf(t.y)};
    f2 r = pk_fma(-k, splat(6.28318548202514648f), x);
    r = pk_fma(-k, splat(-1.74845553146951715e-7f), r);
    const f2 rev = r * splat(0.15915494309189535f);
    sn = f2{__builtin_amdgcn_sinf(rev.x), __builtin_amdgcn_sinf(rev.y)};
    cn = f2{__builtin_amdgcn_cosf(rev.x), __builtin_amdgcn_cosf(rev.y)};
#else
    fast_sincos2(x, sn, cn);
#endif
}
struct Z3SideMap { int first[3], last[3]; };
template <int K, int... UD>
__host__ __device__ constexpr Z3SideMap z3_side_map() {
    constexpr int LL = sizeof...(UD) / 2;
    constexpr int ud[] = {UD...};
    Z3SideMap m = {{-1, -1, -1}, {-1, -1, -1}};
    for (int w = 0; w < 2 * K; w++) {
        const int q = ud[2 * ((w >> 1) % LL) + (w & 1)];
        if (m.first[q] < 0) m.first[q] = w;
        m.last[q] = w;
    }
    return m;
}
// ROT: rotation string and entangler (cpf_blocks.h rot_pack / rot_with_ent); KC: blocks; MC: blocks of wave A (1 .. KC-1);
// PAIRS: samples per workgroup (1 or 2); UD...: the layer (up0, dn0, up1, dn1, ...), block b sits on pair b mod len (topology.py:15-20).
// PAIRS = 2 (2 K <= 32 sides; a switch, no launch takes it by default): four waves, two samples, each with its own copy of the LDS
// map - and ONE owner of the sides: wave B of the first sample owns the block angles of both (lanes 0 .. 31 the first sample's
// sides, 32 .. 63 the second's; one lane per side).  At two samples per CU it beat one sample per workgroup while an owner lane
// carried all four slots of its side (4.35 against 4.70 ms) and loses to the two-lane layout (4.46 against 4.28).  The second
// sample's wave B only sweeps; each sample's wave A owns its own surface gates and penalty.
template <int ROT, int KC, int MC, int PAIRS, int... UD>
__device__ __forceinline__ void zyz3_body(const KArgs& a) {
    using L = Lay<3, 0>;
    using PL = PairList<UD...>;
    using Lds = Zyz3Lds<KC>;
    constexpr int LL = sizeof...(UD) / 2, K = KC, M = MC;
    constexpr int NR = rot_nr(ROT), ENT = rot_ent(ROT), NBP = 2 * NR + (ENT == 0 ? 1 : 0), NS = 9;
    // Two lanes per item where they fit (wave A always; wave B up to 16 blocks, one sample per workgroup): the even lane of a pair owns
    // the item's angle slots 0 and 1, the odd lane slots 2 and 3 - so Adam and the sin / cos polynomials run once per step on ONE
    // packed pair per lane instead of two (half of the owner phases' longest stretch; both waves have to shrink for the step to:
    // timing prototypes 4.58 -> 4.57 / 4.54 ms for one wave alone, 4.24 for both).  The even lane is the item's main lane: it fetches
    // its neighbour's (cos, sin) by DPP and does everything else as before; the odd lane gets its two gradients back the same way.
    constexpr bool S2 = PAIRS == 1 && 4 * K <= 64;
    constexpr int NITEM = (S2 ? 4 : 2) * K;                  // owner lanes of wave B: the sides (S2: two lanes per side)
    constexpr int SROWS = Lds::SROWS, SA = Lds::srow_blk(M);   // rows of shares [0, SA) are written and added by wave A, [SA, SROWS) by wave B
    constexpr int NRA = (SA + 63) / 64, NRB = (SROWS - SA + 63) / 64;
    static_assert(LL > 0 && NR >= 1 && NR <= 3 && ENT <= 1, "1..3 rotation letters, CP or CZ blocks");
    static_assert(M >= 1 && M < K && NITEM <= 64, "both waves sweep at least one block; one owner lane per item");
    static_assert(PAIRS == 1 || (PAIRS == 2 && NITEM <= 32), "two samples per workgroup: the sides of each in half a wave");
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
    bool lb[3];
#pragma unroll
    for (int q = 0; q < 3; q++) lb[q] = (lane >> L::row_lane_bit(q)) & 1;
    const bool odd = lane & 1;
    const int col = L::col_index(lane, 0), row = L::row_index(lane, 0);
    int kofs[LL];                                            // per layer position: the lane's table entry
    static_for([&](auto jj) __attribute__((always_inline)) {
        constexpr int J = decltype(jj)::value, UP = PL::v[2 * J], DN = PL::v[2 * J + 1];
        kofs[J] = 4 * (2 * (lb[UP] ? 1 : 0) + (lb[DN] ? 1 : 0));
    }, typename make_iseq<LL>::type{});
    f2* sp = reinterpret_cast<f2*>(smem + Lds::STATES) + lane;             // sp[0]: wave A's state at the cut, sp[64]: wave B's
    float* part = smem + Lds::PART + lane;                                 // row r at part[ZYZ3_ROW r]
    const float* tab = smem + Lds::TAB;
    penalty_table_init(smem, a, tid);
    if (tid == 0) *reinterpret_cast<float4*>(smem + Lds::REC1 + 8 * K) = make_float4(1.f, 0.f, 1.f, 0.f);
    if (tid < 3 * Lds::up4(K)) smem[Lds::PV + tid] = 0.f;
    if (tid < 8) smem[Lds::MSF + tid] = 0.f;                 // [3]: the step counter wave A waits for - LDS keeps what the last workgroup left
    if (tid < 4) {
        // gradient of the CP angle = k_t s11 + k_u pz_up + k_d pz_dn with (k_t, k_u, k_d) by where the entangler's phase sits in the held
        // frame (ca, cb): cpf_zyz_math.h chain_cp's four branches, times -2 (round 5 compiled them into nested exec-mask regions)
        const int ca = tid & 1, cb = tid >> 1;
        *reinterpret_cast<float4*>(smem + Lds::CPT + 4 * tid) =
            make_float4((ca == cb) ? -2.f : 2.f, cb ? 0.f : (ca ? 1.f : -1.f), ca ? 0.f : (cb ? 1.f : -1.f), 0.f);
    }

    // phase PH (0 owner / build, 1 forward half, 2 loss + gradient half) begins: set the wave's issue priority where it changes
    auto phase_prio = [&](auto code, auto ph) __attribute__((always_inline)) {
        constexpr int C = decltype(code)::value, PH = decltype(ph)::value;
        constexpr int d[3] = {(C / 100) % 10, (C / 10) % 10, C % 10};
        if constexpr (d[PH] != d[(PH + 2) % 3]) { if (a.prio & 1) __builtin_amdgcn_s_setprio(d[PH]); }
    };
    auto lds_fence = [&]() __attribute__((always_inline)) { __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); };
    // Every barrier a wave passes is counted (one scalar add); the two waves compare their counts after the last step and poison the
    // sample's results when they differ.  The step loops are only correct while the compiler keeps ONE barrier per phase and wave: a
    // loop it split by lanes ran its s_barrier once per group of lanes and the two waves silently lost step (see "No branch on the
    // lane ..." above) - with the count, every instance of every shape checks itself on every launch.
    // (readfirstlane: free where the count already sits in a scalar register - every instance but those with fewer than three blocks in
    // wave B, where the count passes a divergent join and hipcc's backend aborted on the "+s" operand: "illegal VGPR to SGPR copy")
    unsigned nbar = 0;
    auto barrier = [&]() __attribute__((always_inline)) { nbar = __builtin_amdgcn_readfirstlane(nbar); asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier\n\ts_add_u32 %0, %0, 1" : "+s"(nbar) :: "memory", "scc"); };
    // static structure: qubit of side w (block w >> 1, side w & 1)
    auto qubit_of = [&](int w) {
        int r = 0;
        static_for([&](auto jj) __attribute__((always_inline)) {
            constexpr int J = decltype(jj)::value;
            if ((w >> 1) % LL == J) r = (w & 1) ? PL::v[2 * J + 1] : PL::v[2 * J];
        }, typename make_iseq<LL>::type{});
        return r;
    };
    // One wave adds rows [r_lo, r_hi) of the shares (its own lanes wrote them: in order, no barrier), one lane per row, 64 rows a
    // round.  A row is added in eight parts - by the lane bits 0, 1, 3 of its entries, the row qubits' bits - because a u row
    // (u = Im conj(w) psi at every lane) stands for three signed sums: Pz of a qubit is u over the lanes with the qubit's bit clear
    // minus u over the others, and the |11> sum of a pair is u over the lanes with both bits set.  Writing u once instead of three
    // signed copies takes two of a block's seven 256-byte LDS stores out of the sweeps (config 2: 4.76 -> 4.66 ms); the parts cost the
    // adding lane nothing (the same 32 packed adds) and the combinations ~20 instructions per wave and step.
    struct RowK { int row, d_tot, dz0, dz1, dz2, ds; float w01, w02, w12; };
    auto row_consts = [&](int r_lo, int r_hi, int r) {
        RowK k;
        // results a row does not produce go to a scratch slot of this lane's own (round 5 sent them all to ONE spare address: up to 45
        // lanes of a store on one bank)
        const int sr = r_lo + lane + 64 * r, none = Lds::SCR + (r_lo == 0 ? 0 : 64) + lane;
        k.row = sr < r_hi ? sr : r_hi - 1;
        k.d_tot = k.dz0 = k.dz1 = k.dz2 = k.ds = none;
        k.w01 = k.w02 = k.w12 = 0.f;
        if (sr < r_hi) {
            const int b = sr < 7 ? -1 : (sr - 7) / 5, j = sr < 7 ? sr : (sr - 7) % 5;
            const bool is_u = sr < 7 ? j == 6 : j == 4;
            if (!is_u) {
                // a Px row is a plain sum over the lanes.  A Py row holds the UNSIGNED shares (round 6): its total carries the sign of
                // the lane's bit of the row's qubit, which the eight parts give for nothing - Py = (bit clear) - (bit set), the
                // z combination of that qubit - instead of one packed multiply per block in the sweeps
                const int dst = Lds::TOT + (b < 0 ? 3 * (j >> 1) + (j & 1) : Lds::row_blk(b) + 3 * (j >> 1) + (j & 1));
                if (j & 1) {
                    const int q = b < 0 ? (j >> 1) : qubit_of(2 * b + (j >> 1));
                    if (q == 0) k.dz0 = dst; else if (q == 1) k.dz1 = dst; else k.dz2 = dst;
                } else k.d_tot = dst;
            } else {
                if (b < 0) { k.dz0 = Lds::TOT + 2; k.dz1 = Lds::TOT + 5; k.dz2 = Lds::TOT + 8; }
                else {                                       // (selects, not a dynamically indexed local array: no private segment)
                    const int qu = qubit_of(2 * b), qd = qubit_of(2 * b + 1), du = Lds::TOT + Lds::row_blk(b) + 2, dd = Lds::TOT + Lds::row_blk(b) + 5;
                    k.dz0 = qu == 0 ? du : (qd == 0 ? dd : none);
                    k.dz1 = qu == 1 ? du : (qd == 1 ? dd : none);
                    k.dz2 = qu == 2 ? du : (qd == 2 ? dd : none);
                }
                if (b + 1 < K) {                             // the |11> sum of the NEXT block's pair
                    k.ds = Lds::TOT + (b < 0 ? 9 : Lds::row_blk(b) + 6);
                    const int qa = qubit_of(2 * (b + 1)), qb = qubit_of(2 * (b + 1) + 1), pr = (1 << qa) | (1 << qb);
                    k.w01 = pr == 3 ? 1.f : 0.f; k.w02 = pr == 5 ? 1.f : 0.f; k.w12 = pr == 6 ? 1.f : 0.f;
                }
            }
        }
        return k;
    };
    auto row_sums = [&](const RowK& k) __attribute__((always_inline)) {
        const float4* src = reinterpret_cast<const float4*>(smem + Lds::PART + ZYZ3_ROW * k.row);
        f2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f}, b0 = {0.f, 0.f}, b1 = {0.f, 0.f};      // lane bit 3 clear: (bits 1 0 = 00, 01), (10, 11); set: likewise
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const float4 v = src[q];
            if ((q >> 1) & 1) { b0 += f2{v.x, v.y}; b1 += f2{v.z, v.w}; }
            else { a0 += f2{v.x, v.y}; a1 += f2{v.z, v.w}; }
        }
        const f2 t0 = a0 + b0, t1 = a1 + b1;
        const float sa = (a0.x + a0.y) + (a1.x + a1.y), sb = (b0.x + b0.y) + (b1.x + b1.y);
        smem[k.d_tot] = sa + sb;
        smem[k.dz0] = (t0.x + t1.x) - (t0.y + t1.y);
        smem[k.dz1] = (t0.x + t0.y) - (t1.x + t1.y);
        smem[k.dz2] = sa - sb;
        smem[k.ds] = fmaf(k.w01, t1.y, fmaf(k.w02, b0.y + b1.y, k.w12 * (b1.x + b1.y)));
    };
    const int iters = a.T > 0 ? a.T : 1;
#ifdef CPF_STAMPS
    unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev = 0;
#endif
    // ---- the angles of an owner lane's item (four slots: a side's rotation letters and, slot 3, the CP angle; a qubit's three surface
    // angles), their Adam state and (cos, sin): wave B's side lanes, wave A's qubit lanes
    int pj[4] = {-1, -1, -1, -1};
    f2 th01 = {0.f, 0.f}, th23 = {0.f, 0.f}, mm01 = {0.f, 0.f}, mm23 = {0.f, 0.f}, vv01 = {0.f, 0.f}, vv23 = {0.f, 0.f};
    f2 g01 = {0.f, 0.f}, g23 = {0.f, 0.f}, nlr01 = {0.f, 0.f}, nlr23 = {0.f, 0.f};
    float cs[4][2] = {{1.f, 0.f}, {1.f, 0.f}, {1.f, 0.f}, {1.f, 0.f}};
    float best = 0.f, first = 0.f;
    // HBM is touched at the two ends of the run only, and in whole rows: the pair's 128 threads read the sample's P initial angles
    // contiguously into LDS (the rows of shares: free until the first gradient half) and copy them to output slot 0; an owner lane
    // takes its angles from there, and after the last step puts its best ones back for one contiguous store (instead of scattered
    // 4-byte accesses of the owner lanes; time-neutral, 4.37 / 4.33 ms).
    static_assert(Lds::SROWS * ZYZ3_ROW >= 9 + 7 * K, "the angles of a sample fit the scratch");
    {
        const bool live_pair = PAIRS == 1 || s_raw < a.nsamples;
        float* const ang = smem + Lds::PART;
        for (int j = tid; j < P; j += 128) {
            const float v = a.init[s_idx * P + j];
            ang[j] = v;
            if (live_pair) a.out_params[(s_idx * 2) * P + j] = v;          // slot 0: the initial angles; slot 1 is written once, after the last step
        }
        barrier();
    }
    auto load_angles = [&](const float* ang, int64_t sample) __attribute__((always_inline)) {
        float t0[4], nl[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            t0[k] = 0.f; nl[k] = 0.f;
            if (pj[k] >= 0) {
                t0[k] = ang[pj[k]];
                nl[k] = (a.frozen && a.frozen[sample * P + pj[k]]) ? 0.f : -a.lr;
            }
        }
        th01 = f2{t0[0], t0[1]}; th23 = f2{t0[2], t0[3]}; nlr01 = f2{nl[0], nl[1]}; nlr23 = f2{nl[2], nl[3]};
    };
    // Adam closes step `it - 1` (optax scale_by_adam, optimization.py:14-25: the same operations in the same order as the scalar form, per
    // half), then (cos, sin) of the half angles (slot 3: a CP angle, the full angle)
    // halfq: the lane owns slots 0 and 1 only (two lanes per item); sc01: what sin / cos take of slots 0, 1 (half angles; the full angle of a CP slot)
    // The bias corrections come as RECIPROCALS from the host (round 6: two v_rcp_f32 fewer per wave and step) and one step ahead:
    // the scalar load of step it + 1's pair is issued here, a whole step before its use (the knock-out of the load alone was 1 %).
    f2 ibc_next = {1.f, 1.f};
    { const float2 b0 = const_load_f2(a.ibias, 0); ibc_next = f2{b0.x, b0.y}; }
    auto adam_sincos = [&](int it, auto halfq, f2 sc01) __attribute__((always_inline)) {
        constexpr bool HALF = decltype(halfq)::value != 0;
        if (it > 0) {
            const f2 ibc = ibc_next;
            const f2 m0 = pk_fma(splat(a.b1), mm01, splat(a.omb1) * g01);
            const f2 v0 = pk_fma(splat(a.b2), vv01, splat(a.omb2) * (g01 * g01));
            mm01 = m0; vv01 = v0;
            const f2 h0 = v0 * splat(ibc.y);
            const f2 d0 = f2{__builtin_amdgcn_sqrtf(h0.x), __builtin_amdgcn_sqrtf(h0.y)} + splat(a.eps);
            const f2 u0 = (m0 * splat(ibc.x)) * f2{__builtin_amdgcn_rcpf(d0.x), __builtin_amdgcn_rcpf(d0.y)};
            th01 = pk_fma(nlr01, u0, th01);
            if constexpr (!HALF) {
                const f2 m1 = pk_fma(splat(a.b1), mm23, splat(a.omb1) * g23);
                const f2 v1 = pk_fma(splat(a.b2), vv23, splat(a.omb2) * (g23 * g23));
                mm23 = m1; vv23 = v1;
                const f2 h1 = v1 * splat(ibc.y);
                const f2 d1 = f2{__builtin_amdgcn_sqrtf(h1.x), __builtin_amdgcn_sqrtf(h1.y)} + splat(a.eps);
                const f2 u1 = (m1 * splat(ibc.x)) * f2{__builtin_amdgcn_rcpf(d1.x), __builtin_amdgcn_rcpf(d1.y)};
                th23 = pk_fma(nlr23, u1, th23);
            }
        }
        { const float2 bn = const_load_f2(a.ibias, it); ibc_next = f2{bn.x, bn.y}; }      // for step it + 1 (it <= T - 1: inside the table)
        f2 s01, c01;
        z3_sincos2(th01 * sc01, s01, c01);
        cs[0][0] = c01.x; cs[0][1] = s01.x; cs[1][0] = c01.y; cs[1][1] = s01.y;
        if constexpr (HALF) {                               // slots 2, 3: the neighbouring (odd) lane's pair
            cs[2][0] = z3_nb(c01.x); cs[2][1] = z3_nb(s01.x); cs[3][0] = z3_nb(c01.y); cs[3][1] = z3_nb(s01.y);
        } else {
            f2 s23, c23;
            z3_sincos2(th23 * f2{0.5f, 1.f}, s23, c23);
            cs[2][0] = c23.x; cs[2][1] = s23.x; cs[3][0] = c23.y; cs[3][1] = s23.y;
        }
    };
    // r x the sum of the K penalty values
    auto pen_reg = [&](const float4 (&pvq)[(K + 3) / 4]) __attribute__((always_inline)) {
        float pen_sum = 0.f;
        if (a.use_reg) {
#pragma unroll
            for (int k = 0; k < (K + 3) / 4; k++) pen_sum += (pvq[k].x + pvq[k].y) + (pvq[k].z + pvq[k].w);
        }
        return a.use_reg ? a.r * pen_sum : 0.f;
    };
    // The best iterate's angles stay in registers (selects, no branch and no store inside the step) and are written once, by
    // put_best, after the last step.  ONE wave decides which iterate is the best so far - wave B, in its loss phase - and publishes
    // the decision (Lds::MSF + 4); wave A follows it at the top of its next step, before Adam moves its angles (round 6: until then
    // both waves formed the regularised loss and compared, and out_params[:, 1] was a consistent iterate only as long as the two
    // copies of that arithmetic rounded alike).
    f2 bt01 = {0.f, 0.f}, bt23 = {0.f, 0.f};
    auto put_best = [&](float* ang) __attribute__((always_inline)) {          // the owner lane's best angles, back into the sample's row in LDS
        const float t0[4] = {bt01.x, bt01.y, bt23.x, bt23.y};
#pragma unroll
        for (int k = 0; k < 4; k++) if (pj[k] >= 0) ang[pj[k]] = t0[k];
    };
    auto hist_angles = [&](int it, int64_t sample, bool live) __attribute__((always_inline)) {      // keep_history: this step's angles of the lane's item
#ifndef CPF_STAMPS
        if (a.hist_params && live) {
            const float t0[4] = {th01.x, th01.y, th23.x, th23.y};
#pragma unroll
            for (int k = 0; k < 4; k++) if (pj[k] >= 0) a.hist_params[(sample * a.T + it) * P + pj[k]] = t0[k];
        }
#endif
    };
    // wave B: loss, the decision, its own best angles
    auto track = [&](int it, float tr, float ti, float reg, int64_t sample, bool live, bool writes_loss, float* take_dst) __attribute__((always_inline)) {
        const float loss = 1.f - fmaf(tr, tr, ti * ti) * a.inv_dim2;
        const float regloss = loss + reg;
        const bool take = it == 0 || regloss < best;         // strict: the first of equal losses stays (optimization.py:365-370 argmin)
        first = it == 0 ? regloss : first;
        best = take ? regloss : best;
        bt01 = take ? th01 : bt01; bt23 = take ? th23 : bt23;
        *take_dst = take ? 1.f : 0.f;                        // (the lanes that do not speak for a sample store into scratch)
        hist_angles(it, sample, live);
#ifndef CPF_STAMPS
        if (a.hist_loss && writes_loss && live) a.hist_loss[sample * a.T + it] = regloss;
#endif
    };

    if (wave == 0) {
        // =====================================================================================================================
        // wave A: blocks 0 .. M-1, the surface gates, the penalty
        // =====================================================================================================================
        int surf_off[3];                                     // element [r_q][c_q] of S_q; of F S_q 8 floats further
#pragma unroll
        for (int q = 0; q < 3; q++) surf_off[q] = Lds::SURFM + 16 * q + 4 * (lb[q] ? 1 : 0) + 2 * ((col >> (2 - q)) & 1);
        // Lanes 0 .. 2 own the surface gates Rz(t2) Rx(t1) Rz(t0) of the three qubits (main.py:122-124): chain rule, Adam, (cos, sin)
        // and both variants of the matrix (S and F S: which one the sweeps take is wave B's flip parity of the step) - behind the last
        // barrier of a step, while wave B does the same for the blocks.  Lanes 0 .. K-1 also evaluate the penalty of the CP angles
        // (penalty.py:44-71) that wave B publishes: ~35 instructions that would otherwise sit in wave B's stream.
        const bool live = PAIRS == 1 || s_raw < a.nsamples;
        const bool is_qubit = lane < 6 && !(lane & 1);     // the main lane of qubit lane >> 1; the odd lane beside it owns the third angle
        const int qa = lane < 6 ? lane >> 1 : 0;
        if (lane < 6) { if (lane & 1) pj[0] = 3 * qa + 2; else { pj[0] = 3 * qa; pj[1] = 3 * qa + 1; } }
        load_angles(smem + Lds::PART, s_idx);
        float my_flip = 0.f;
        // No branch on the lane here: every lane forms a matrix and stores it, the lanes that own no qubit into the rows of shares
        // (dead between the last barrier of a step and the gradient halves of the next).  A divergent region next to the loop's back
        // edge made the compiler split the loop by lanes - the barrier at its top then ran once per group of lanes.
        float* const pub = smem + (is_qubit ? Lds::SURFM + 16 * qa : Lds::PART + 16 * lane);
        static_assert(Lds::SROWS * ZYZ3_ROW >= 64 * 16, "scratch for the lanes without a qubit");
        unsigned flag_addr = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) float*)(smem + Lds::MSF + 3);      // LDS byte address
        asm volatile("" : "+v"(flag_addr));                  // (opaque: the compiler otherwise re-derives it, null check of the cast included, in every poll)
        float* const pen_dst = smem + (lane < K ? Lds::PV + lane : Lds::PART + 2 * lane);      // (the scratch of the matrices again: stored later, read never)
        static_assert(Lds::SROWS * ZYZ3_ROW >= 128 + (Lds::PSL - Lds::PV), "scratch for the lanes without a CP angle");
        auto publish_surface = [&]() __attribute__((always_inline)) {
            float sm[8];
            zyz::surface_matrix(sm, cs[0][0], cs[0][1], cs[1][0], cs[1][1], cs[2][0], cs[2][1], 0);
            float4* dst = reinterpret_cast<float4*>(pub);
            dst[0] = make_float4(sm[0], sm[1], sm[2], sm[3]);
            dst[1] = make_float4(sm[4], sm[5], sm[6], sm[7]);
            dst[2] = make_float4(-sm[4], -sm[5], -sm[6], -sm[7]);            // F S: row 0 = -row 1, row 1 = row 0
            dst[3] = make_float4(sm[0], sm[1], sm[2], sm[3]);
        };
        RowK rka[NRA];
#pragma unroll
        for (int r = 0; r < NRA; r++) rka[r] = row_consts(0, SA, r);
        // Where a block's five rows of shares start, per block, as LDS pointers kept in registers for the whole run (the empty asm
        // makes them opaque): ds_write2_b32 reaches 1020 bytes from its base, so the compiler otherwise re-derives a base with a
        // v_add_u32 in front of every second store of the gradient halves (round 5's listing: 35 of them per sample-step).
        z3_lds_f* pb[M];
        static_for([&](auto bb) __attribute__((always_inline)) {
            constexpr int B = decltype(bb)::value;
            pb[B] = (z3_lds_f*)(part + ZYZ3_ROW * Lds::srow_blk(B));
            asm volatile("" : "+v"(pb[B]));
        }, typename make_iseq<M>::type{});
        z3_lds_f* pb_surf = (z3_lds_f*)part;
        asm volatile("" : "+v"(pb_surf));
        // lanes 0 .. 5 own the surface angles, lanes 0 .. K-1 evaluate the penalty of a CP angle each: nobody else executes the owner phase
        const bool owner_a = !CPF_Z3_EXECMASK || lane < ((ENT == 0 && a.use_reg && K > 6) ? K : 6);
        barrier();                                           // constant records of the set-up (the first owner phase runs on all lanes: whatever
                                                             // the compiler computes once, in front of the loop, must not see a narrowed mask)
#ifdef CPF_STAMPS
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev) :: "memory");
#endif
        if ((a.prio & 1) && CPF_Z3_PRIO_A / 100 != 0) __builtin_amdgcn_s_setprio((CPF_Z3_PRIO_A / 100) % 10);
#if CPF_Z3_STAGGER > 1
        if (a.prio & 2) {
            static_for([&](auto gg) __attribute__((always_inline)) {
                constexpr int G = decltype(gg)::value;
                if (G > 0 && (int)((blockIdx.x >> CPF_Z3_STAGGER_SHIFT) % CPF_Z3_STAGGER) == G) __builtin_amdgcn_s_sleep(CPF_Z3_STAGGER_STEP * G / CPF_Z3_STAGGER);
            }, typename make_iseq<CPF_Z3_STAGGER>::type{});
        }
#endif
        for (int it = 0; it < iters; it++) {
            phase_prio(ic<CPF_Z3_PRIO_A>{}, ic<0>{});
            if (owner_a) {   // the surface gates' owner phase, beside wave B's: chain rule for the step just closed, Adam, (cos, sin), matrices
                const float* tot = smem + Lds::TOT + 3 * qa;
                float gs[3] = {0.f, 0.f, 0.f};
                if (it > 0) {
                    // wave B's verdict on the iterate of the step just closed (these are still its angles: Adam comes next)
                    const float tk = smem[Lds::MSF + 4];
                    zyz::chain_surface(gs, tot[0], tot[1], tot[2], my_flip != 0.f ? 1 : 0, cs[1][0], cs[1][1], cs[2][0], cs[2][1]);
                    bt01 = tk != 0.f ? th01 : bt01;
                }
                const float g2 = z3_nb(gs[2]);              // the third angle's gradient, for the odd lane
                g01 = (lane & 1) ? f2{g2, 0.f} : f2{gs[0], gs[1]};
                adam_sincos(it, ic<1>{}, splat(0.5f));
                publish_surface();
                if (ENT == 0 && a.use_reg) {
                    // the penalty of CP angle `lane` (penalty.py:44-71): its value for the loss, its slope for wave B's next Adam step.
                    // Wave B publishes the angles right after its Adam update and then a step counter; by the time the surface
                    // matrices are out it has long done so (its chain rule and Adam come first), so the wait is a formality - and it
                    // cannot deadlock: wave B's stores depend on nothing this wave does before the barrier below.
                    float seen;
                    do {
                        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(seen) : "v"(flag_addr) : "memory");
                    } while (__builtin_amdgcn_readfirstlane(__float_as_int(seen)) != __float_as_int((float)(it + 1)));
                    const float th_cp = smem[Lds::PTH + (lane < K ? lane : K - 1)];
                    float sl;
                    const float pv = z3_penalty(smem, a.pen_kind, th_cp, sl);
                    pen_dst[0] = pv;                         // no branch on the lane here either (see publish_surface)
                    pen_dst[Lds::PSL - Lds::PV] = a.r * sl;
                }
            }
            CPF_STAMP(0)
            barrier();                                       // tables and surface matrices of this step
            CPF_STAMP(1)
            phase_prio(ic<CPF_Z3_PRIO_A>{}, ic<1>{});
            float4 ent[M];
            f2 own[M + 1];
            const float4 msf = *reinterpret_cast<const float4*>(smem + Lds::MSF);
            float2 fa[3], fb[3];
#pragma unroll
            for (int q = 0; q < 3; q++) { fa[q] = *reinterpret_cast<const float2*>(smem + surf_off[q]); fb[q] = *reinterpret_cast<const float2*>(smem + surf_off[q] + 8); }
            static_for([&](auto bb) __attribute__((always_inline)) {
                constexpr int B = decltype(bb)::value;
                ent[B] = *reinterpret_cast<const float4*>(tab + 16 * B + kofs[B % LL]);
            }, typename make_iseq<M>::type{});
            my_flip = qa == 0 ? msf.x : (qa == 1 ? msf.y : msf.z);
            const f2 f0 = msf.x != 0.f ? f2{fb[0].x, fb[0].y} : f2{fa[0].x, fa[0].y};
            const f2 f1 = msf.y != 0.f ? f2{fb[1].x, fb[1].y} : f2{fa[1].x, fa[1].y};
            const f2 f2_ = msf.z != 0.f ? f2{fb[2].x, fb[2].y} : f2{fa[2].x, fa[2].y};
            f2 x = z3_cmul(f2_, z3_cmul(f1, f0));
            own[0] = x;
            static_for([&](auto bb) __attribute__((always_inline)) {
                constexpr int B = decltype(bb)::value, J = B % LL;
                x = z3_cmul(f2{ent[B].x, ent[B].y}, x);
                z3_stage2<PL::v[2 * J], PL::v[2 * J + 1], false>(x, ent[B].z, ent[B].w);
                own[B + 1] = x;
            }, typename make_iseq<M>::type{});
            sp[0] = x;                                       // psi(s_{M-1}): the only state the other wave reads
            hist_angles(it, s_idx, live);                    // (keep_history only)
            CPF_STAMP(2)                                  // forward half
            barrier();
            CPF_STAMP(3)
            phase_prio(ic<CPF_Z3_PRIO_A>{}, ic<2>{});
            const f2 w_c = sp[64];
            const f2 tt = z3_cmul_conj(w_c, x);              // T = sum conj(W) psi at the cut
            float tr = tt.x, ti = tt.y;
            z3_allsum(tr, ti, odd);
            f2 w = z3_cmul(f2{-a.inv_dim2 * tr, -a.inv_dim2 * ti}, w_c);       // the cotangent at the cut: f W, f = -T / N^2
            if (a.T == 0 || it + 1 == iters) break;
            CPF_STAMP(4)                                  // loss
            // A block of the gradient half: the partner lanes' psi(s_B) on the block's two qubits are at hand - fetched while the
            // PREVIOUS block's stages ran, straight from the partner lanes' registers (the state is no longer a sweep's live value) -
            // three complex products give the five shares, then the block is un-applied from the cotangent while the next block's
            // partner values arrive.  19 instructions, no s_nop.
            f2 pre, pim;                                     // (re up, re dn), (im up, im dn) of psi(s_B) at the partner lanes
            z3_fetch<PL::v[2 * ((M - 1) % LL)], PL::v[2 * ((M - 1) % LL) + 1]>(own[M], pre, pim);
#if CPF_Z3_FUSED_A
            {
                float w0 = w.x, w1 = w.y, pr0 = pre.x, pr1 = pre.y, pi0 = pim.x, pi1 = pim.y;       // in v[2:7] inside every block's statement
                static_for([&](auto bb) __attribute__((always_inline)) {
                    constexpr int B = M - 1 - decltype(bb)::value, J = B % LL, JN = (B > 0 ? B - 1 : 0) % LL;
                    constexpr int UP = PL::v[2 * J], DN = PL::v[2 * J + 1];
                    // Im of conj(w) psi[partner] and MINUS its real part for both qubits at once (the Py rows hold unsigned shares: the sign
                    // of the lane's bit is applied by the row sums, see row_consts); Im(conj(w) psi) at the lane
                    f2 PX, PY, U;
                    z3_grad_block_a<DN, UP, PL::v[2 * JN], PL::v[2 * JN + 1], B == 0>(w0, w1, pr0, pr1, pi0, pi1, own[B + 1], own[B], f2{ent[B].x, ent[B].y},
                                                                                       ent[B].w, ent[B].z, PX, PY, U);
                    pb[B][0] = PX.x;
                    pb[B][ZYZ3_ROW] = PY.x;
                    pb[B][2 * ZYZ3_ROW] = PX.y;
                    pb[B][3 * ZYZ3_ROW] = PY.y;
                    pb[B][4 * ZYZ3_ROW] = U.y;
                }, typename make_iseq<M>::type{});
                w = f2{w0, w1};
            }
#else
            static_for([&](auto bb) __attribute__((always_inline)) {
                constexpr int B = M - 1 - decltype(bb)::value, J = B % LL;
                constexpr int UP = PL::v[2 * J], DN = PL::v[2 * J + 1];
                const f2 o = own[B + 1];
                f2 PX, PY;
                z3_products<false>(w, pre, pim, PX, PY);
                pb[B][0] = PX.x;
                pb[B][ZYZ3_ROW] = PY.x;
                pb[B][2 * ZYZ3_ROW] = PX.y;
                pb[B][3 * ZYZ3_ROW] = PY.y;
                pb[B][4 * ZYZ3_ROW] = z3_cmul_conj(w, o).y;
                if constexpr (B > 0) {
                    constexpr int JN = (B - 1) % LL;
                    z3_stage2_fetch<DN, UP, true, PL::v[2 * JN], PL::v[2 * JN + 1]>(w, ent[B].w, ent[B].z, own[B], pre, pim);
                } else z3_stage2<DN, UP, true, 0>(w, ent[B].w, ent[B].z);
                w = z3_cmul_conj(f2{ent[B].x, ent[B].y}, w);
            }, typename make_iseq<M>::type{});
#endif
            {   // the surface slice s_-1: all three qubits, and the |11> sum of block 0.  The partner is taken of the cotangent (the sums run
                // over all lanes, so conj(w) psi[partner] and conj(w[partner]) psi add up to the same): ONE complex product per qubit
                // gives the Py share (real part; its sign comes with the row sums) and the Px share (imaginary part)
                const f2 o = own[0];
                f2 wq[3];
                z3_partner3(w, wq[0], wq[1], wq[2]);
                static_for([&](auto qq) __attribute__((always_inline)) {
                    constexpr int Q = decltype(qq)::value;
                    const f2 zq = z3_cmul_conj(wq[Q], o);
                    pb_surf[ZYZ3_ROW * (2 * Q + 0)] = zq.y;
                    pb_surf[ZYZ3_ROW * (2 * Q + 1)] = zq.x;
                }, typename make_iseq<3>::type{});
                pb_surf[ZYZ3_ROW * 6] = z3_cmul_conj(w, o).y;
            }
            lds_fence();
#pragma unroll
            for (int r = 0; r < NRA; r++) {
#if CPF_Z3_ROWMASK
                if (lane + 64 * r < SA) row_sums(rka[r]);
#else
                row_sums(rka[r]);
#endif
            }
            CPF_STAMP(5)                                  // gradient half + row sums
            barrier();
            CPF_STAMP(6)
            CPF_STAMP(7)
        }
#ifdef CPF_STAMPS
        if (blockIdx.x == 0 && pair == 0 && lane == 0 && a.hist_loss) for (int k = 0; k < 8; k++) a.hist_loss[k] = (float)stamp_sum[k];
#endif
#ifdef CPF_HWID
        // diagnostics build (tools/micro/hwid3.py, keep_history, T >= 4): HW_ID / XCC_ID of the sample's two waves in place of the
        // first four entries of its loss history - which waves share a SIMD
        if (a.hist_loss && lane == 0 && live) {
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            a.hist_loss[s_idx * a.T + 0] = (float)(hw & 0xffff); a.hist_loss[s_idx * a.T + 1] = (float)(xcc & 0xf);
        }
#endif
        barrier();                                           // wave B's verdict on the last iterate is out
        if (smem[Lds::MSF + 4] != 0.f) bt01 = th01;
        put_best(smem + Lds::PART);
        if (lane == 0) smem[Lds::MSF + 5] = __uint_as_float(nbar + 1);      // this wave's barrier count (the one below included), for wave B to compare
        barrier();                                           // wave B of the pair stores the row
        return;
    }

    // =========================================================================================================================
    // wave B: blocks M .. K-1, and the owner of every block angle: lane w < 2 K owns side w (block w >> 1, side w & 1)
    // =========================================================================================================================
#if CPF_Z3_PRIO_B
    // While every workgroup of the launch is resident (a.prio, set by the host) wave B gets the higher issue priority on the SIMD it
    // shares with a wave A of another sample (config 2, cut 4: 5.35 -> 4.74 ms; raising wave A's in its gradient half on top: 4.92).
    // In a launch of several rounds of workgroups the priority starves the other waves (2048 samples 11.6 -> 14.8 ms).
    if (a.prio & 1) __builtin_amdgcn_s_setprio((CPF_Z3_PRIO_B / 100) % 10);
#endif
    // the owner wave and, in it, the sample and the item of this lane
    const bool owner_wave = pair == 0;
    const bool upper = PAIRS > 1 && lane >= 32;
    const int il = PAIRS > 1 ? (lane & 31) : lane, lane0 = lane - il;
    float* osmem = smem_wg + (upper ? Lds::TOTAL : 0);
    const int64_t os_raw = (int64_t)blockIdx.x * PAIRS + (upper ? 1 : 0);
    const bool ovs = PAIRS == 1 || os_raw < a.nsamples;
    const int64_t os_idx = ovs ? os_raw : (int64_t)a.nsamples - 1;
    const float2 vt_ = a.target[row * 8 + col];
    const f2 vt = {vt_.x, vt_.y};
    // first / last side on each row qubit: compile-time constants of the layer and the block count
    constexpr Z3SideMap sm = z3_side_map<K, UD...>();
    const int lastside[3] = {sm.last[0], sm.last[1], sm.last[2]}, firstside[3] = {sm.first[0], sm.first[1], sm.first[2]};
    // S2: lanes 2 w and 2 w + 1 work for side w - the even one is its main lane (slots 0, 1 and everything else), the odd one owns slots 2, 3
    const int iw = S2 ? (il >> 1) : il;
    const bool odd2 = S2 && (il & 1);
    const bool is_owner = owner_wave && il < NITEM;          // a lane with angles of a side
    const bool is_side = is_owner && !odd2;                  // the side's main lane
    const int ib = iw >> 1, iside = iw & 1;
    int prev = -1, prev_dn = -1;                             // previous side on the side's qubit (side 0: also of the block's dn side)
    unsigned long long later = 0;                            // later sides on the side's qubit
    int first_q = -1;                                        // the qubit this side is the first one on (it publishes the qubit's flip parity)
    if (is_side) {
        const int q = qubit_of(iw);
        for (int w = iw - 1; w >= 0; w--) if (qubit_of(w) == q) { prev = w; break; }
        for (int w = iw + 1; w < 2 * K; w++) if (qubit_of(w) == q) later |= 1ull << (lane0 + (S2 ? 2 * w : w));   // bit = the main lane of the side
        if (iside == 0) { const int qd = qubit_of(iw + 1); for (int w = iw - 1; w >= 0; w--) if (qubit_of(w) == qd) { prev_dn = w; break; } }
#pragma unroll
        for (int q = 0; q < 3; q++) if (firstside[q] == iw) first_q = q;
    }
    if (is_owner) {                                          // global slot k: rotation letter k of the side; slot 3: the block's CP angle (side 0)
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int j = k < NR ? NS + NBP * ib + 2 * k + iside : ((k == 3 && ENT == 0 && iside == 0) ? NS + NBP * ib + 2 * NR : -1);
            if (!S2) pj[k] = j;
            else if ((k >> 1) == (odd2 ? 1 : 0)) pj[k & 1] = j;
        }
    }
    load_angles(osmem + Lds::PART, os_idx);
    const bool has_cp = ENT == 0 && is_owner && iside == 0 && (!S2 || odd2);       // the lane that holds the CP angle: slot 3, or slot 1 of the odd lane
    const f2 sc01 = (S2 && odd2) ? f2{0.5f, 1.f} : f2{0.5f, 0.5f};
    auto cp_angle = [&]() __attribute__((always_inline)) { return S2 ? th01.y : th23.y; };
    // the value held by the main lane of the block's other side
    auto side_nb = [&](auto v) __attribute__((always_inline)) {
        if constexpr (S2) return z3_partner<1>(v); else return z3_nb(v);
    };
    zyz::Side sd = zyz::Side{};
    int cacb = 0;                                            // even side lanes: where the entangler's phase sits
    float psl = 0.f;
    // what the chain rule reads (float offsets into the row totals)
    int rowP = 0, row_s11 = 0, row_pzu = 0, row_pzd = 0;
    if (is_side) {
        rowP = Lds::row_blk(ib) + 3 * iside;
        row_s11 = ib == 0 ? 9 : Lds::row_blk(ib - 1) + 6;
        auto pz_row = [&](int pw, int w) { return pw >= 0 ? Lds::row_blk(pw >> 1) + 3 * (pw & 1) + 2 : 3 * qubit_of(w) + 2; };   // Pz of the previous side, or of the surface state
        row_pzu = pz_row(prev, iw);
        row_pzd = iside == 0 ? pz_row(prev_dn, iw + 1) : 0;
    }
    const int prev_off = Lds::REC1 + 4 * (prev >= 0 ? prev : 2 * K);
    // final left factors: the last side on each row qubit, read at the start of every step
    int lfin_off[3];
#pragma unroll
    for (int q = 0; q < 3; q++) lfin_off[q] = Lds::REC1 + 4 * (lastside[q] >= 0 ? lastside[q] : 2 * K) + 2 * (lb[q] ? 1 : 0);

    if (owner_wave && il < 3 && firstside[il] < 0) osmem[Lds::MSF + il] = 0.f;      // a qubit no block touches is never flipped
    RowK rkb[NRB];
#pragma unroll
    for (int r = 0; r < NRB; r++) rkb[r] = row_consts(SA, SROWS, r);
    z3_lds_f* pb[K - M];                                     // where each block's rows of shares start (see wave A)
    static_for([&](auto bb) __attribute__((always_inline)) {
        constexpr int B = M + decltype(bb)::value;
        pb[B - M] = (z3_lds_f*)(part + ZYZ3_ROW * Lds::srow_blk(B));
        asm volatile("" : "+v"(pb[B - M]));
    }, typename make_iseq<K - M>::type{});
    const float th_cp0 = cp_angle();                         // the initial CP angle of a has_cp lane, for the epilogue's reg recompute
    // Where this lane's owner-phase stores go: the real slot where the lane owns one, its own 16 bytes of the rows of shares (dead
    // between the row sums and the next gradient half) where it does not - so that the owner and loss phases store without a branch on
    // the lane (round 5's listing: an s_and_saveexec / s_or pair around each of seven stores per step)
    float* const dummy = osmem + Lds::PART + 4 * il + (upper ? 128 : 0);
    float* const take_dst = osmem + Lds::MSF + (il == 0 ? 4 : 7);      // (stored in the LOSS phase, while wave A fills the rows of shares: a spare slot, not `dummy`)
    float* const pth_dst = has_cp ? osmem + Lds::PTH + ib : dummy;
    float* const flag_dst = il == 0 ? osmem + Lds::MSF + 3 : dummy;
    float* const rec_dst = is_side ? osmem + Lds::REC1 + 4 * iw : dummy;
    float* const msf_dst = first_q >= 0 ? osmem + Lds::MSF + first_q : dummy;
    float* const tab_dst = is_side ? osmem + Lds::TAB + 16 * ib + 8 * iside : osmem + Lds::PART + 256 + 8 * lane;
    const float* const psl_src = osmem + Lds::PSL + (has_cp ? ib : 0);
    static_assert(Lds::SROWS * ZYZ3_ROW >= 256 + 8 * 64, "scratch for the lanes without a table entry");
    // the lanes with angles of a side: nobody else executes the chain rule and the owner phase
    const bool owner_b = owner_wave && (!CPF_Z3_EXECMASK || is_owner);
    barrier();                                               // penalty table, constant records: written by wave A's lanes
#ifdef CPF_STAMPS
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev) :: "memory");
#endif
#ifdef CPF_HWID
    unsigned long long clk0 = 0, rt0 = 0;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(clk0), "=s"(rt0) :: "memory");
#endif
#if CPF_Z3_STAGGER > 1
    if (a.prio & 2) {
        static_for([&](auto gg) __attribute__((always_inline)) {
        constexpr int G = decltype(gg)::value;
        if (G > 0 && (int)((blockIdx.x >> CPF_Z3_STAGGER_SHIFT) % CPF_Z3_STAGGER) == G) __builtin_amdgcn_s_sleep(CPF_Z3_STAGGER_STEP * G / CPF_Z3_STAGGER);
        }, typename make_iseq<CPF_Z3_STAGGER>::type{});
    }
#endif
    for (int it = 0; it < iters; it++) {
        phase_prio(ic<CPF_Z3_PRIO_B>{}, ic<0>{});
        // ================= owner: the chain rule of the step just closed (into the owners' registers), Adam, (cos, sin); sides, parities,
        // e-pairs, tables.  At the TOP of the step, under one branch on the lane (the last statement in front of the loop's back edge
        // stays a barrier: see the head of this file).
        if (owner_b) {
        if (it > 0) {
            const float* tot = osmem + Lds::TOT;
            const float px = tot[rowP], py = tot[rowP + 1], pz = tot[rowP + 2];
            const float s11 = tot[row_s11], pzu = tot[row_pzu], pzd = tot[row_pzd];
            const float4 kcp = *reinterpret_cast<const float4*>(osmem + Lds::CPT + 4 * cacb);
            auto rrec = [&](int idx, float& c, float& sn, int& kind) __attribute__((always_inline)) {
                const int k = (idx - 1) >> 1;
                c = cs[k][0]; sn = cs[k][1]; kind = 0;
            };
            float gp[4] = {0.f, 0.f, 0.f, 0.f};
            zyz::chain_side<ROT & 0xFFFF>(gp, px, py, pz, sd, rrec, NR, iside);
            if (ENT == 0) gp[3] = fmaf(kcp.x, s11, fmaf(kcp.y, pzu, kcp.z * pzd));
            if constexpr (S2) {                           // slots 2, 3 go to the odd lane beside the main lane
                const float n2 = z3_nb(gp[2]), n3 = z3_nb(gp[3]);
                g01 = odd2 ? f2{n2, n3} : f2{gp[0], gp[1]};
            } else { g01 = f2{gp[0], gp[1]}; g23 = f2{gp[2], gp[3]}; }
        }
        if (it > 0 && has_cp && a.use_reg) { if constexpr (S2) g01.y += psl; else g23.y += psl; }
        adam_sincos(it, ic<S2 ? 1 : 0>{}, sc01);
        if (ENT == 0 && a.use_reg) {                          // wave A evaluates the penalty of the CP angles: the angles, then the step counter it waits for
            *pth_dst = cp_angle();
            lds_fence();
            *flag_dst = (float)(it + 1);
        }
        {
            auto rrec = [&](int idx, float& c, float& sn, int& kind) __attribute__((always_inline)) {
                const int k = (idx - 1) >> 1;
                c = cs[k][0]; sn = cs[k][1]; kind = 0;
            };
            sd = zyz::side_build<ROT & 0xFFFF>(rrec, NR, iside);
            *reinterpret_cast<float4*>(rec_dst) = make_float4(sd.l0.re, sd.l0.im, sd.l1.re, sd.l1.im);
            const unsigned long long fbits = __builtin_amdgcn_ballot_w64(is_side && sd.f != 0);
            sd.m = __popcll(fbits & later) & 1;
            *msf_dst = (float)(sd.m ^ sd.f);      // every flip on the qubit: those later than its first side, and that side's own
            lds_fence();
            const float4 pl = *reinterpret_cast<const float4*>(osmem + prev_off);
            const zyz::EPair e = zyz::side_epair(sd, true, zyz::c2(pl.x, pl.y), zyz::c2(pl.z, pl.w));      // "no previous side" reads the identity record
            // the block's other side is the neighbouring lane: exchange the e-pairs instead of a trip through LDS, then this lane
            // builds the two entries with a_up = iside (side 0: entries (0, 0), (0, 1); side 1: (1, 0), (1, 1))
            const f2 pe0 = {side_nb(e.e0.re), side_nb(e.e0.im)}, pe1 = {side_nb(e.e1.re), side_nb(e.e1.im)};
            const int pc = side_nb(e.c);
            const float psc = side_nb(sd.sc), pt = side_nb(sd.t);
            const float plc = side_nb(cs[3][0]), pls = side_nb(cs[3][1]);          // the CP angle belongs to side 0
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
            {
                float4* dst = reinterpret_cast<float4*>(tab_dst);
                dst[0] = make_float4(d0.x, d0.y, tus, -td);
                dst[1] = make_float4(d1.x, d1.y, tus, td);
            }
        }
        }
        CPF_STAMP(0)                                  // owner / build
        barrier();
        CPF_STAMP(1)
        phase_prio(ic<CPF_Z3_PRIO_B>{}, ic<1>{});

        // ================= forward half: Vd = conj(dfin) V = W(s_{K-1}), then blocks K-1 .. M un-applied.  The table entries are read
        // once and serve both sweeps; the states stay in registers for the gradient half.
        float4 ent[K - M];
        f2 own[K - M + 1];
        float4 pvq[(K + 3) / 4];
        float2 lf[3];
#pragma unroll
        for (int q = 0; q < 3; q++) lf[q] = *reinterpret_cast<const float2*>(smem + lfin_off[q]);
        static_for([&](auto bb) __attribute__((always_inline)) {
            constexpr int B = M + decltype(bb)::value;
            ent[B - M] = *reinterpret_cast<const float4*>(tab + 16 * B + kofs[B % LL]);
        }, typename make_iseq<K - M>::type{});
        f2 x, xs;                                            // W(s_b) as the gradient half will read it, and the copy the next block's stages overwrite
        z3_cmul_conj2(z3_cmul(f2{lf[2].x, lf[2].y}, z3_cmul(f2{lf[1].x, lf[1].y}, f2{lf[0].x, lf[0].y})), vt, x, xs);
        own[K - M] = x;
        {
            float L0 = xs.x, L1 = xs.y;                      // the working copy, in v[2:3] inside every block's statement: see z3_unapply_block
            static_for([&](auto bb) __attribute__((always_inline)) {
                constexpr int B = K - 1 - decltype(bb)::value, J = B % LL;
                x = z3_unapply_block<PL::v[2 * J + 1], PL::v[2 * J], B == M>(L0, L1, f2{ent[B - M].x, ent[B - M].y}, ent[B - M].w, ent[B - M].z);
                own[B - M] = x;                                              // W(s_{B-1})
            }, typename make_iseq<K - M>::type{});
        }
        sp[64] = x;                                   // W(s_{M-1}): the only state the other wave reads
        CPF_STAMP(2)                                  // forward half
        barrier();
        CPF_STAMP(3)
        phase_prio(ic<CPF_Z3_PRIO_B>{}, ic<2>{});

        // ================= loss: T = sum conj(W) psi at the cut (wave A forms the same T for its cotangent)
        const f2 psi_c = sp[0];
#pragma unroll
        for (int k = 0; k < (K + 3) / 4; k++) pvq[k] = reinterpret_cast<const float4*>(osmem + Lds::PV)[k];
        psl = *psl_src;
        const f2 tt = z3_cmul_conj(x, psi_c);
        float tr = tt.x, ti = tt.y;
        z3_allsum(tr, ti, odd);
        if (owner_wave) {
            float otr = tr, oti = ti;                 // T of the sample this lane's item belongs to
            if constexpr (PAIRS > 1) {                // the second sample's T: its wave pair forms it too, but not where the angles live
                const f2* sp1 = reinterpret_cast<const f2*>(smem_wg + Lds::TOTAL + Lds::STATES) + lane;
                const f2 t1 = z3_cmul_conj(sp1[64], sp1[0]);
                float tr1 = t1.x, ti1 = t1.y;
                z3_allsum(tr1, ti1, odd);
                otr = upper ? tr1 : tr; oti = upper ? ti1 : ti;
            }
            track(it, otr, oti, pen_reg(pvq), os_idx, ovs, il == 0, take_dst);
        }
        if (a.T == 0) break;
        if (it + 1 == iters) break;                   // the last step's gradient would move the angles past the last evaluated iterate
        CPF_STAMP(4)                                  // loss

        // ================= gradient half: blocks M .. K-1 applied to psi' = conj(f) psi (the scalar moves onto psi once), W(s_b) at hand
        {
            f2 pp = z3_cmul_conj(f2{-a.inv_dim2 * tr, -a.inv_dim2 * ti}, psi_c);
            if constexpr (CPF_Z3_FUSED_B != 0) {
            float p0 = pp.x, p1 = pp.y;                                           // in v[2:3] inside every block's statement
            static_for([&](auto bb) __attribute__((always_inline)) {
                constexpr int B = M + decltype(bb)::value, J = B % LL;
                constexpr int UP = PL::v[2 * J], DN = PL::v[2 * J + 1];
                // the roles swapped: Im / Re of conj(W[partner]) psi' (unsigned: the row sums sign the Py rows), Im(conj(W) psi') at the lane
                f2 PX, PY, U;
                z3_grad_block_b<UP, DN>(p0, p1, f2{ent[B - M].x, ent[B - M].y}, own[B - M + 1], ent[B - M].z, ent[B - M].w, PX, PY, U);
                pb[B - M][0] = PX.x;
                pb[B - M][ZYZ3_ROW] = PY.x;
                pb[B - M][2 * ZYZ3_ROW] = PX.y;
                pb[B - M][3 * ZYZ3_ROW] = PY.y;
                pb[B - M][4 * ZYZ3_ROW] = U.y;
            }, typename make_iseq<K - M>::type{});
            } else {
            static_for([&](auto bb) __attribute__((always_inline)) {
                constexpr int B = M + decltype(bb)::value, J = B % LL;
                constexpr int UP = PL::v[2 * J], DN = PL::v[2 * J + 1];
                pp = z3_cmul(f2{ent[B - M].x, ent[B - M].y}, pp);
                const f2 o = own[B - M + 1];                                      // W(s_B)
                f2 pre, pim, PX, PY;                                              // W(s_B) at the partner lanes: (re up, re dn), (im up, im dn)
                z3_stage2_fetch<UP, DN, false, UP, DN>(pp, ent[B - M].z, ent[B - M].w, o, pre, pim);
                // the roles swapped: Im / Re of conj(W[partner]) psi' (unsigned: the row sums sign the Py rows), Im(conj(W) psi') at the lane
                z3_products<true>(pp, pre, pim, PX, PY);
                pb[B - M][0] = PX.x;
                pb[B - M][ZYZ3_ROW] = PY.x;
                pb[B - M][2 * ZYZ3_ROW] = PX.y;
                pb[B - M][3 * ZYZ3_ROW] = PY.y;
                pb[B - M][4 * ZYZ3_ROW] = z3_cmul_conj(o, pp).y;
            }, typename make_iseq<K - M>::type{});
            }
        }
        lds_fence();
#pragma unroll
        for (int r = 0; r < NRB; r++) {
#if CPF_Z3_ROWMASK
            if (lane + 64 * r < SROWS - SA) row_sums(rkb[r]);
#else
            row_sums(rkb[r]);
#endif
        }
        CPF_STAMP(5)                                  // gradient half + row sums
        barrier();
        CPF_STAMP(6)

        CPF_STAMP(7)
    }
#ifdef CPF_STAMPS
    if (blockIdx.x == 0 && pair == 0 && lane == 0 && a.hist_loss) for (int k = 0; k < 8; k++) a.hist_loss[8 + k] = (float)stamp_sum[k];
#endif

#ifdef CPF_HWID
    if (a.hist_loss && lane == 0 && (PAIRS == 1 || s_raw < a.nsamples)) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        a.hist_loss[s_idx * a.T + 2] = (float)(hw & 0xffff); a.hist_loss[s_idx * a.T + 3] = (float)(xcc & 0xf);
        if (a.T >= 6) {                                      // shader clocks and 100 MHz ticks of this wave's step loops: the clock the launch ran at
            unsigned long long clk1, rt1;
            asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(clk1), "=s"(rt1) :: "memory");
            a.hist_loss[s_idx * a.T + 4] = (float)(clk1 - clk0); a.hist_loss[s_idx * a.T + 5] = (float)(rt1 - rt0);
        }
    }
#endif
    // ---- epilogue (optimization.py:365-370, cp_utils.py:144-164): reg recompute, loss = regloss - reg, argmin, CZ count, entry-loss flag
    barrier();                                               // wave A takes the verdict on the last iterate
    if (owner_wave) put_best(osmem + Lds::PART);
    if (lane == 0) smem[Lds::MSF + 6] = __uint_as_float(nbar + 1);          // this wave's barrier count, the one below included
    barrier();
    if (PAIRS == 1 || s_raw < a.nsamples) {
        const float* ang = smem + Lds::PART;
        for (int j = lane; j < P; j += 64) a.out_params[(s_idx * 2 + 1) * P + j] = ang[j];
    }
    if (owner_wave) {
        // every wave of the workgroup passed the same barriers or the run is void: a step loop the compiler split by lanes lets a wave
        // pass one more than once per phase (see the head of this file); the results are poisoned rather than silently wrong
        const bool in_step = __float_as_uint(osmem[Lds::MSF + 5]) == nbar && __float_as_uint(osmem[Lds::MSF + 6]) == nbar;
        if (!in_step) { first = __builtin_nanf(""); best = first; }
        const int bi = (best < first) ? 1 : 0;
        float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f;
        if (has_cp) {
            const float t0 = th_cp0, t1 = S2 ? bt01.y : bt23.y;      // the initial CP angle, the best one
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
            a.out_cz[os_idx] = in_step ? (int32_t)acc2 : -1;
            a.out_pass[os_idx] = ((bi ? l1 : l0) <= a.entry_loss) ? 1 : 0;
        }
    }
}

template <int ROT, int KC, int MC, int PAIRS, int... UD>
__global__ __launch_bounds__(128 * PAIRS, 1) void k_cp_zyz3(const KArgs a) {
    zyz3_body<ROT, KC, MC, PAIRS, UD...>(a);
}

}  // namespace cpf
