// cpflow_amd: specialised persistent kernel for 3-qubit templates (BASELINE configs 1-3).
//
// One wavefront = one sample.  The 8x8 unitary psi and its cotangent C sit one complex
// element per lane: row bits on lane bits {0,1,3} (all reachable with one DPP modifier:
// quad_perm / row_ror:8), column bits on lane bits {2,4,5} (columns never interact).
//
// What differs from the generic kernel (cpf_device.h):
//   * dispatch once per entangling BLOCK (6 ordered qubit pairs x the rotation string as
//     template parameters) instead of once per gate;
//   * the (cos, sin) of a block's angles live in one 64-byte LDS record, fetched with four
//     broadcast ds_read_b128 and prefetched one block ahead;
//   * the 7 per-angle gradient partials of a block are reduced together by a DPP
//     reduce-scatter (24 VALU ops per block instead of 7 full wave reductions);
//   * theta / Adam moments / penalty state stay in the owner lanes' registers for the
//     whole run (parameter j lives in lane j % 64, slot j / 64).
#pragma once
#include "cpf_device.h"

namespace cpf {

// rotation string packed as NR | kind0 << 4 | kind1 << 8 | kind2 << 12
__host__ __device__ constexpr int rot_pack(int nr, int k0, int k1, int k2) { return nr | (k0 << 4) | (k1 << 8) | (k2 << 12); }
__host__ __device__ constexpr int rot_nr(int rot) { return rot & 15; }
__host__ __device__ constexpr int rot_kind(int rot, int i) { return (rot >> (4 + 4 * i)) & 15; }

struct Fast3Layout {                  // LDS float offsets
    // floats [0, 28): penalty table (PEN_LO / PEN_SI of cpf_device.h)
    static constexpr int SURF_CS = 28;    // 9 x (c, s) = 18 floats, padded to 20
    static constexpr int BLK_CS = 48;     // + 16 * b : 7 x (c, s), [14] = pair code, [15] pad
    __host__ __device__ static constexpr int g_base(int K) { return BLK_CS + 16 * K; }
    // gradient slabs: [k][row] floats; surface 16 k, block 8 k, 4 rows each
    __host__ __device__ static constexpr int g_surf(int K) { return g_base(K); }
    __host__ __device__ static constexpr int g_blk(int K, int b) { return g_base(K) + 64 + 32 * b; }
    __host__ __device__ static constexpr int total(int K) { return g_base(K) + 64 + 32 * K; }
};

using L3 = Lay<3, 0>;

// reduce 8 per-lane values over the 64 lanes; afterwards dst[k*4 + row] (row = lane >> 4) holds the
// partial sum of d[k] over the 16 lanes of that row.  Stages: scatter over lane bits 0,1 (quad_perm),
// bit 2 (row_shl/shr:4), then add over bit 3 (row_ror:8).  Rows are added by the reader.
__device__ __forceinline__ void reduce8_rows(const float (&d)[8], float* dst, int lane, int dst_lane) {
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
    float e[4], f[2];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float keep = b0 ? d[2 * i + 1] : d[2 * i], give = b0 ? d[2 * i] : d[2 * i + 1];
        e[i] = keep + dppz<0xB1>(give);
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const float keep = b1 ? e[2 * i + 1] : e[2 * i], give = b1 ? e[2 * i] : e[2 * i + 1];
        f[i] = keep + dppz<0x4E>(give);
    }
    const float keep = b2 ? f[1] : f[0], give = b2 ? f[0] : f[1];
    float r = keep + xor_lane<2>(give);
    r += dppz<0x128>(r);
    dst[dst_lane] = r;       // dst_lane = (lane & 7) * 4 + (lane >> 4); lanes l and l^8 store the same value to the same word
}

template <int KIND, int Q>
__device__ __forceinline__ void f3_rot_fwd(State<L3>& st, float c, float s) {
    rot_apply<L3, KIND, Q>(st.pr, st.pi, c, s, st);
}
template <int KIND, int Q>
__device__ __forceinline__ float f3_rot_bwd(State<L3>& st, float c, float s) {
    const float d = rot_grad<L3, KIND, Q>(st);
    rot_apply<L3, KIND, Q>(st.pr, st.pi, c, -s, st);
    rot_apply<L3, KIND, Q>(st.cr, st.ci, c, -s, st);
    return d;
}
// one entangling block, forward: CP(up,dn,a[nb-1]) then R_g(up,a[2i]) R_g(dn,a[2i+1]) per letter
template <int UP, int DN, int ROT>
__device__ __forceinline__ void f3_block_fwd(State<L3>& st, const float (&r)[16]) {
    constexpr int NR = rot_nr(ROT), NB = 2 * NR + 1;
    constexpr int Q0 = UP < DN ? UP : DN, Q1 = UP < DN ? DN : UP;
    cp_apply<L3, Q0, Q1>(st.pr, st.pi, r[2 * (NB - 1)], r[2 * (NB - 1) + 1], st);
#pragma unroll
    for (int i = 0; i < NR; i++) {
        if (rot_kind(ROT, i) == OP_RX) {
            f3_rot_fwd<OP_RX, UP>(st, r[4 * i], r[4 * i + 1]); f3_rot_fwd<OP_RX, DN>(st, r[4 * i + 2], r[4 * i + 3]);
        } else if (rot_kind(ROT, i) == OP_RY) {
            f3_rot_fwd<OP_RY, UP>(st, r[4 * i], r[4 * i + 1]); f3_rot_fwd<OP_RY, DN>(st, r[4 * i + 2], r[4 * i + 3]);
        } else {
            f3_rot_fwd<OP_RZ, UP>(st, r[4 * i], r[4 * i + 1]); f3_rot_fwd<OP_RZ, DN>(st, r[4 * i + 2], r[4 * i + 3]);
        }
    }
}
// backward: gates in reverse order; d[k] = gradient share of block angle k
template <int UP, int DN, int ROT>
__device__ __forceinline__ void f3_block_bwd(State<L3>& st, const float (&r)[16], float (&d)[8]) {
    constexpr int NR = rot_nr(ROT), NB = 2 * NR + 1;
    constexpr int Q0 = UP < DN ? UP : DN, Q1 = UP < DN ? DN : UP;
#pragma unroll
    for (int k = 0; k < 8; k++) d[k] = 0.f;
#pragma unroll
    for (int i = NR - 1; i >= 0; i--) {
        if (rot_kind(ROT, i) == OP_RX) {
            d[2 * i + 1] = f3_rot_bwd<OP_RX, DN>(st, r[4 * i + 2], r[4 * i + 3]); d[2 * i] = f3_rot_bwd<OP_RX, UP>(st, r[4 * i], r[4 * i + 1]);
        } else if (rot_kind(ROT, i) == OP_RY) {
            d[2 * i + 1] = f3_rot_bwd<OP_RY, DN>(st, r[4 * i + 2], r[4 * i + 3]); d[2 * i] = f3_rot_bwd<OP_RY, UP>(st, r[4 * i], r[4 * i + 1]);
        } else {
            d[2 * i + 1] = f3_rot_bwd<OP_RZ, DN>(st, r[4 * i + 2], r[4 * i + 3]); d[2 * i] = f3_rot_bwd<OP_RZ, UP>(st, r[4 * i], r[4 * i + 1]);
        }
    }
    const float cc = r[2 * (NB - 1)], sc = r[2 * (NB - 1) + 1];
    d[NB - 1] = cp_grad<L3, Q0, Q1>(st);
    cp_apply<L3, Q0, Q1>(st.pr, st.pi, cc, -sc, st);
    cp_apply<L3, Q0, Q1>(st.cr, st.ci, cc, -sc, st);
}

__device__ __forceinline__ void load_rec(const float* p, float (&r)[16]) {
    const float4* q = reinterpret_cast<const float4*>(p);
    const float4 a = q[0], b = q[1], c = q[2], d = q[3];
    r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w; r[4] = b.x; r[5] = b.y; r[6] = b.z; r[7] = b.w;
    r[8] = c.x; r[9] = c.y; r[10] = c.z; r[11] = c.w; r[12] = d.x; r[13] = d.y; r[14] = d.z; r[15] = d.w;
}

#define CPF_F3_PAIRS(M) M(0, 1) M(1, 0) M(0, 2) M(2, 0) M(1, 2) M(2, 1)

// LAYER: the connectivity layer as a compile-time sequence of (up, dn) pairs
//   L | up0 << 4 | dn0 << 6 | up1 << 8 | dn1 << 10 | up2 << 12 | dn2 << 14       (L <= 3), or 0 = any layer.
// With a known layer the block loop runs layer by layer over straight-line code; with LAYER = 0 every block
// dispatches on its pair code through a branch tree, which measured 22 % slower on config 2
// (143 extra branches per step at ~19 cycles each, plus the register moves that merge the cases).
__host__ __device__ constexpr int layer_pack(int L, int u0, int d0, int u1, int d1, int u2, int d2) {
    return L | (u0 << 4) | (d0 << 6) | (u1 << 8) | (d1 << 10) | (u2 << 12) | (d2 << 14);
}
__host__ __device__ constexpr int layer_len(int layer) { return layer & 15; }
__host__ __device__ constexpr int layer_up(int layer, int i) { return (layer >> (4 + 4 * i)) & 3; }
__host__ __device__ constexpr int layer_dn(int layer, int i) { return (layer >> (6 + 4 * i)) & 3; }

template <int ROT, int PPT, int LAYER>
__global__ __launch_bounds__(64) void k_cp_fast3(const KArgs a) {
    constexpr int LL = layer_len(LAYER);
    constexpr int NR = rot_nr(ROT), NB = 2 * NR + 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x;
    const int P = a.P, K = (P - 9) / NB, nsurf = a.n_actual;       // nsurf = 3 (2-qubit problems use the generic kernel)
    const int64_t s_idx = blockIdx.x;
    const float* init = a.init + s_idx * P;
    const uint8_t* frozen = a.frozen ? a.frozen + s_idx * P : nullptr;
    float* out_best = a.out_params + (s_idx * 2 + 1) * P;

    State<L3> st;
#pragma unroll
    for (int i = 0; i < 3; i++) { st.lb[i] = (lane >> L3::row_lane_bit(i)) & 1; st.sg[i] = st.lb[i] ? 1.f : -1.f; }
    const int col = L3::col_index(lane, 0), row = L3::row_index(lane, 0);
    const float2 tgt = a.target[row * 8 + col];
    const float vr = tgt.x, vi = tgt.y;
    const bool diag = (row == col);
    const int dst_lane = (lane & 7) * 4 + (lane >> 4);

    // ---- owner state.  Owner slot o = lane + 64 * s holds parameter pj[s]; the K CP angles come first
    //      (o < K -> CP angle of block o), so that for K <= 64 only slot 0 evaluates the penalty.
    float th[PPT], mm[PPT], vv[PPT], psl[PPT], pval[PPT];
    int cs_off[PPT], g_off[PPT], pj[PPT];
    bool valid[PPT], is_cp[PPT], frz[PPT];
    penalty_table_init(smem, a, lane);
#pragma unroll
    for (int s = 0; s < PPT; s++) {
        const int o = lane + 64 * s;
        valid[s] = o < P;
        int j = 0;
        if (valid[s]) {
            if (o < K) j = 9 + NB * o + NB - 1;
            else if (o - K < 9) j = o - K;
            else { const int q = o - K - 9; j = 9 + NB * (q / (NB - 1)) + q % (NB - 1); }
        }
        pj[s] = j;
        const int jj = j;
        is_cp[s] = valid[s] && a.pkind[jj];
        frz[s] = valid[s] && frozen && frozen[jj];
        if (jj < 9) { cs_off[s] = Fast3Layout::SURF_CS + 2 * jj; g_off[s] = Fast3Layout::g_surf(K) + 4 * jj; }
        else {
            const int b = (jj - 9) / NB, k = (jj - 9) % NB;
            cs_off[s] = Fast3Layout::BLK_CS + 16 * b + 2 * k; g_off[s] = Fast3Layout::g_blk(K, b) + 4 * k;
        }
        th[s] = valid[s] ? init[jj] : 0.f;
        mm[s] = 0.f; vv[s] = 0.f; psl[s] = 0.f; pval[s] = 0.f;
        if (valid[s]) { a.out_params[(s_idx * 2) * P + jj] = th[s]; out_best[jj] = th[s]; }
    }
    const bool slot_has_cp[4] = {K > 0, K > 64, K > 128, K > 192};
    // block pair codes: record slot 14 (static for the whole run)
    for (int b = lane; b < K; b += 64) {
        const int4 op = a.prog[9 + b * NB];                      // the block's CP op {kind, lo, hi, param}; up/dn from the next op
        const int4 r0 = a.prog[9 + b * NB + 1], r1 = a.prog[9 + b * NB + 2];
        (void)op;
        smem[Fast3Layout::BLK_CS + 16 * b + 14] = __int_as_float(r0.y * 4 + r1.y);   // up * 4 + dn
        smem[Fast3Layout::BLK_CS + 16 * b + 15] = 0.f;
    }
    auto publish = [&](int s) {      // (cos, sin) of the owner's angle -> LDS; penalty value / slope of CP angles
        float sn, cn;
        sincosf(is_cp[s] ? th[s] : 0.5f * th[s], &sn, &cn);
        if (valid[s]) *reinterpret_cast<float2*>(smem + cs_off[s]) = make_float2(cn, sn);
        if (a.use_reg && slot_has_cp[s] && is_cp[s]) { float sl; pval[s] = penalty_lds(smem, a.pen_kind, th[s], sl); psl[s] = a.r * sl; }
    };
    __syncthreads();
#pragma unroll
    for (int s = 0; s < PPT; s++) publish(s);
    __syncthreads();

    float best = 0.f, first = 0.f;
    const int iters = a.T > 0 ? a.T : 1;
    for (int it = 0; it < iters; it++) {
        const float2 bc = const_load_f2(a.bias, a.T > 0 ? it : 0);   // scalar load, issued early, consumed by the Adam phase
        // ================= forward
        st.pr[0] = diag ? 1.f : 0.f; st.pi[0] = 0.f;
        {
            const float* r = smem + Fast3Layout::SURF_CS;
            float sr[20];
#pragma unroll
            for (int i = 0; i < 5; i++) { const float4 v = reinterpret_cast<const float4*>(r)[i]; sr[4*i] = v.x; sr[4*i+1] = v.y; sr[4*i+2] = v.z; sr[4*i+3] = v.w; }
            f3_rot_fwd<OP_RZ, 0>(st, sr[0], sr[1]);  f3_rot_fwd<OP_RX, 0>(st, sr[2], sr[3]);   f3_rot_fwd<OP_RZ, 0>(st, sr[4], sr[5]);
            f3_rot_fwd<OP_RZ, 1>(st, sr[6], sr[7]);  f3_rot_fwd<OP_RX, 1>(st, sr[8], sr[9]);   f3_rot_fwd<OP_RZ, 1>(st, sr[10], sr[11]);
            f3_rot_fwd<OP_RZ, 2>(st, sr[12], sr[13]); f3_rot_fwd<OP_RX, 2>(st, sr[14], sr[15]); f3_rot_fwd<OP_RZ, 2>(st, sr[16], sr[17]);
        }
        if (K > 0) {
            float rc[16];
            load_rec(smem + Fast3Layout::BLK_CS, rc);
            if constexpr (LL == 0) {
                for (int b = 0; b < K; b++) {
                    float rn[16];
                    load_rec(smem + Fast3Layout::BLK_CS + 16 * (b + 1 < K ? b + 1 : b), rn);     // prefetch next record
                    const int pair = __builtin_amdgcn_readfirstlane(__float_as_int(rc[14]));
                    switch (pair) {
#define CPF_F3_CASE(U, D) case U * 4 + D: f3_block_fwd<U, D, ROT>(st, rc); break;
                        CPF_F3_PAIRS(CPF_F3_CASE)
#undef CPF_F3_CASE
                        default: break;
                    }
#pragma unroll
                    for (int i = 0; i < 16; i++) rc[i] = rn[i];
                }
            } else {
                // block b = l * LL + i sits on pair i of the layer (topology.py:15-20)
#define CPF_F3_FWD_STEP(I)                                                                              \
                if (b < K) {                                                                            \
                    float rn[16];                                                                       \
                    load_rec(smem + Fast3Layout::BLK_CS + 16 * (b + 1 < K ? b + 1 : b), rn);            \
                    f3_block_fwd<layer_up(LAYER, I), layer_dn(LAYER, I), ROT>(st, rc);                  \
                    _Pragma("unroll") for (int i = 0; i < 16; i++) rc[i] = rn[i];                       \
                    b++;                                                                                \
                }
                for (int b = 0; b < K;) {
                    CPF_F3_FWD_STEP(0)
                    if constexpr (LL > 1) { CPF_F3_FWD_STEP(1) }
                    if constexpr (LL > 2) { CPF_F3_FWD_STEP(2) }
                }
#undef CPF_F3_FWD_STEP
            }
        }
        // ================= loss, penalty, best tracking
        float tr = wave_sum(fmaf(st.pr[0], vr, st.pi[0] * vi));
        float ti = wave_sum(fmaf(st.pi[0], vr, -st.pr[0] * vi));
        float reg = 0.f;
        if (a.use_reg) {
            float ps = 0.f;
#pragma unroll
            for (int s = 0; s < PPT; s++) ps += pval[s];
            reg = a.r * wave_sum(ps);
        }
        const float ab = sqrtf(fmaf(tr, tr, ti * ti));
        const float loss = 1.f - ab * ab * a.inv_dim2;
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
            if (lane == 0) a.hist_loss[s_idx * a.T + it] = regloss;
        }
        if (a.T == 0) break;

        // ================= backward sweep
        {
            const float fr = -a.inv_dim2 * tr, fi = -a.inv_dim2 * ti;
            st.cr[0] = fmaf(fr, vr, -fi * vi);
            st.ci[0] = fmaf(fr, vi, fi * vr);
        }
        if (K > 0) {
            float rc[16], d[8];
            load_rec(smem + Fast3Layout::BLK_CS + 16 * (K - 1), rc);
            if constexpr (LL == 0) {
                for (int b = K - 1; b >= 0; b--) {
                    float rn[16];
                    load_rec(smem + Fast3Layout::BLK_CS + 16 * (b > 0 ? b - 1 : 0), rn);
                    const int pair = __builtin_amdgcn_readfirstlane(__float_as_int(rc[14]));
                    switch (pair) {
#define CPF_F3_CASE(U, D) case U * 4 + D: f3_block_bwd<U, D, ROT>(st, rc, d); break;
                        CPF_F3_PAIRS(CPF_F3_CASE)
#undef CPF_F3_CASE
                        default: break;
                    }
                    reduce8_rows(d, smem + Fast3Layout::g_blk(K, b), lane, dst_lane);
#pragma unroll
                    for (int i = 0; i < 16; i++) rc[i] = rn[i];
                }
            } else {
#define CPF_F3_BWD_STEP(I)                                                                              \
                {                                                                                       \
                    float rn[16];                                                                       \
                    load_rec(smem + Fast3Layout::BLK_CS + 16 * (b > 0 ? b - 1 : 0), rn);                \
                    f3_block_bwd<layer_up(LAYER, I), layer_dn(LAYER, I), ROT>(st, rc, d);               \
                    reduce8_rows(d, smem + Fast3Layout::g_blk(K, b), lane, dst_lane);                   \
                    _Pragma("unroll") for (int i = 0; i < 16; i++) rc[i] = rn[i];                       \
                    b--;                                                                                \
                }
                int b = K - 1;
                // the truncated last layer first (its blocks sit on pairs 0 .. K % LL - 1), then full layers
                const int rem = K % LL;
                if constexpr (LL > 2) { if (rem > 2) CPF_F3_BWD_STEP(2) }
                if constexpr (LL > 1) { if (rem > 1) CPF_F3_BWD_STEP(1) }
                if (rem > 0) CPF_F3_BWD_STEP(0)
                while (b >= 0) {
                    if constexpr (LL > 2) { CPF_F3_BWD_STEP(2) }
                    if constexpr (LL > 1) { CPF_F3_BWD_STEP(1) }
                    CPF_F3_BWD_STEP(0)
                }
#undef CPF_F3_BWD_STEP
            }
        }
        {
            const float* r = smem + Fast3Layout::SURF_CS;
            float sr[20], d[8], d2[8];
#pragma unroll
            for (int i = 0; i < 5; i++) { const float4 v = reinterpret_cast<const float4*>(r)[i]; sr[4*i] = v.x; sr[4*i+1] = v.y; sr[4*i+2] = v.z; sr[4*i+3] = v.w; }
#pragma unroll
            for (int k = 0; k < 8; k++) d2[k] = 0.f;
            d2[0] = f3_rot_bwd<OP_RZ, 2>(st, sr[16], sr[17]); d[7] = f3_rot_bwd<OP_RX, 2>(st, sr[14], sr[15]); d[6] = f3_rot_bwd<OP_RZ, 2>(st, sr[12], sr[13]);
            d[5] = f3_rot_bwd<OP_RZ, 1>(st, sr[10], sr[11]);  d[4] = f3_rot_bwd<OP_RX, 1>(st, sr[8], sr[9]);   d[3] = f3_rot_bwd<OP_RZ, 1>(st, sr[6], sr[7]);
            d[2] = f3_rot_bwd<OP_RZ, 0>(st, sr[4], sr[5]);    d[1] = f3_rot_bwd<OP_RX, 0>(st, sr[2], sr[3]);   d[0] = f3_rot_bwd<OP_RZ, 0>(st, sr[0], sr[1]);
            reduce8_rows(d, smem + Fast3Layout::g_surf(K), lane, dst_lane);
            reduce8_rows(d2, smem + Fast3Layout::g_surf(K) + 32, lane, dst_lane);
        }
        lds_sync<1>();

        // ================= Adam (optax 0.1.1) + next step's (cos, sin) / penalty, owner lanes
        const float ibc1 = __builtin_amdgcn_rcpf(bc.x), ibc2 = __builtin_amdgcn_rcpf(bc.y);
#pragma unroll
        for (int s = 0; s < PPT; s++) {
            const float4 gp = *reinterpret_cast<const float4*>(smem + g_off[s]);
            float g = (gp.x + gp.y) + (gp.z + gp.w);
            if (a.use_reg && is_cp[s]) g += psl[s];
            if (valid[s] && !frz[s]) {
                const float m = fmaf(a.b1, mm[s], a.omb1 * g);
                const float v = fmaf(a.b2, vv[s], a.omb2 * (g * g));
                mm[s] = m; vv[s] = v;
                // hardware sqrt / rcp (1 ulp): the step error is ~1e-8, below the angle's own ulp
                th[s] = fmaf(-a.lr, (m * ibc1) * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v * ibc2) + a.eps), th[s]);
            }
            publish(s);
        }
        lds_sync<1>();
    }

    // ================= epilogue (optimization.py:365-370, cp_utils.py:144-164)
    const int bi = (best < first) ? 1 : 0;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f;
#pragma unroll
    for (int s = 0; s < PPT; s++) {
        if (is_cp[s]) {
            const int j = pj[s];
            const float t0 = a.out_params[(s_idx * 2) * P + j], t1 = out_best[j];
            float sl;
            if (a.use_reg) { acc0 += penalty_lds(smem, a.pen_kind, t0, sl); acc1 += penalty_lds(smem, a.pen_kind, t1, sl); }
            acc2 += cz_value(bi ? t1 : t0, a.thr_cp);
        }
    }
    acc0 = wave_sum(acc0); acc1 = wave_sum(acc1); acc2 = wave_sum(acc2);
    if (lane == 0) {
        const float reg0 = a.use_reg ? a.r * acc0 : 0.f, reg1 = a.use_reg ? a.r * acc1 : 0.f;
        const float l0 = first - reg0, l1 = best - reg1;
        a.out_regloss[s_idx * 2 + 0] = first; a.out_regloss[s_idx * 2 + 1] = best;
        a.out_reg[s_idx * 2 + 0] = reg0;      a.out_reg[s_idx * 2 + 1] = reg1;
        a.out_loss[s_idx * 2 + 0] = l0;       a.out_loss[s_idx * 2 + 1] = l1;
        a.out_cz[s_idx] = (int32_t)acc2;
        a.out_pass[s_idx] = ((bi ? l1 : l0) <= a.entry_loss) ? 1 : 0;
    }
    (void)nsurf;
}

}  // namespace cpf
