// cpflow_amd: building blocks of the layer-specialised kernel (cpf_layered.h):
//   rot_pack / rot_kind      rotation string as a template argument
//   reduce8_rows             DPP reduce-scatter of 8 per-lane gradient partials (16 VALU ops per block
//                            instead of 8 full wave reductions)
//   load_rec                 one 64-byte LDS record = the (cos, sin) of a block's 7 angles, read with four
//                            broadcast ds_read_b128
#pragma once
#include "cpf_device.h"

namespace cpf {

// rotation string packed as NR | kind0 << 4 | kind1 << 8 | kind2 << 12
__host__ __device__ constexpr int rot_pack(int nr, int k0, int k1, int k2) { return nr | (k0 << 4) | (k1 << 8) | (k2 << 12); }
__host__ __device__ constexpr int rot_nr(int rot) { return rot & 15; }
__host__ __device__ constexpr int rot_kind(int rot, int i) { return (rot >> (4 + 4 * i)) & 15; }
// bits 16-17: the block's entangler, 0 = CP(angle) (the block's last parameter), 1 = CZ, 2 = CX (no parameter).  The
// layered kernels treat CZ as a CP whose (cos, sin) record slot holds (-1, 0) for good: the sweeps are the same code, only
// the parameter <-> record-slot map differs (2 NR parameters per block instead of 2 NR + 1).  CX is a row permutation
// (cpf_device.h cx_apply) in the gate-by-gate sweeps; it has no merged-diagonal or two-ended form.
__host__ __device__ constexpr int rot_ent(int rot) { return (rot >> 16) & 3; }
__host__ __device__ constexpr int rot_with_ent(int rot, int ent) { return (rot & 0xFFFF) | (ent << 16); }

// reduce 8 per-lane values over the 64 lanes; afterwards dst[k*4 + row] (row = lane >> 4) holds the
// partial sum of d[k] over the 16 lanes of that row.  Rows are added by the reader.
// Reduce-scatter over lane bits 2 and 3 (8 -> 4 -> 2 values per lane), then bit 0 (2 -> 1), then a plain
// add over bit 1.  Bits 2 and 3 select DPP banks (4 consecutive lanes), so each half of those stages is ONE
// v_add_f32_dpp under a bank mask: lanes of banks 0,2 add their +4 neighbour's even value to their own,
// lanes of banks 1,3 their -4 neighbour's odd value — no v_cndmask, no separate lane moves (12 issue
// slots for the two stages instead of 22; the compiler cannot express the partial writes, hence asm).
// Was 10.7 % of config 2's kernel time with compiler-generated selects (14 calls per optimiser step).
__device__ __forceinline__ int reduce8_dst_lane(int lane) {
    const int k = 4 * (lane & 1) + 2 * ((lane >> 3) & 1) + ((lane >> 2) & 1);     // value index this lane ends up holding
    return k * 4 + (lane >> 4);
}
// SEVEN: d[7] is unused (a block has 7 angles).  The instruction that would add it is dropped; lanes of banks 1, 3
// then carry an undefined e3, which the later stages only ever combine with other "value 7" lanes (the
// reduce-scatter never mixes value indices), so it ends in dst[28..31] of the batch, which no owner reads.
template <bool SEVEN = false>
__device__ __forceinline__ void reduce8_rows(const float (&d)[8], float* dst, int lane, int dst_lane) {
    float e0, e1, e2, e3, f0, f1;
    // s_nop 1: the operands may come straight from a VALU write (2 wait states before a DPP read); inside the
    // block every DPP read is >= 3 instructions behind the write it depends on.
    if constexpr (SEVEN) {
        asm("s_nop 1\n\t"
            "v_add_f32_dpp %0, %6, %6 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %0, %7, %7 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %1, %8, %8 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %1, %9, %9 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %2, %10, %10 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %2, %11, %11 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %3, %12, %12 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %4, %0, %0 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
            "v_add_f32_dpp %4, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %5, %2, %2 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
            "v_add_f32_dpp %5, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xc"
            : "=&v"(e0), "=&v"(e1), "=&v"(e2), "=&v"(e3), "=&v"(f0), "=&v"(f1)
            : "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]), "v"(d[4]), "v"(d[5]), "v"(d[6]));
    } else {
        asm("s_nop 1\n\t"
            "v_add_f32_dpp %0, %6, %6 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %0, %7, %7 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %1, %8, %8 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %1, %9, %9 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %2, %10, %10 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %2, %11, %11 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %3, %12, %12 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %3, %13, %13 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_add_f32_dpp %4, %0, %0 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
            "v_add_f32_dpp %4, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %5, %2, %2 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
            "v_add_f32_dpp %5, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xc"
            : "=&v"(e0), "=&v"(e1), "=&v"(e2), "=&v"(e3), "=&v"(f0), "=&v"(f1)
            : "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]), "v"(d[4]), "v"(d[5]), "v"(d[6]), "v"(d[7]));
    }
    const bool b0 = lane & 1;
    const float keep = b0 ? f1 : f0, give = b0 ? f0 : f1;
    float r = keep + dppz<0xB1>(give);
    r += dppz<0x4E>(r);
    dst[dst_lane] = r;       // dst_lane = reduce8_dst_lane(lane); lanes l and l^2 store the same value to the same word
}

__device__ __forceinline__ void load_rec(const float* p, float (&r)[16]) {
#ifdef CPF_EXP_NORECLOAD          // timing experiment only (wrong results): what the per-block record reads cost
    for (int i = 0; i < 16; i++) r[i] = 0.5f + 0.01f * i;
    return;
#endif
    const float4* q = reinterpret_cast<const float4*>(p);
    const float4 a = q[0], b = q[1], c = q[2], d = q[3];
    r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w; r[4] = b.x; r[5] = b.y; r[6] = b.z; r[7] = b.w;
    r[8] = c.x; r[9] = c.y; r[10] = c.z; r[11] = c.w; r[12] = d.x; r[13] = d.y; r[14] = d.z; r[15] = d.w;
}


}  // namespace cpf
