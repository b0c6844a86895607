// cpflow_amd: building blocks of the layer-specialised kernel (cpf_layered.h):
//   rot_pack / rot_kind      rotation string as a template argument
//   reduce8_rows             DPP reduce-scatter of 8 per-lane gradient partials (24 VALU ops per block
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

// reduce 8 per-lane values over the 64 lanes; afterwards dst[k*4 + row] (row = lane >> 4) holds the
// partial sum of d[k] over the 16 lanes of that row.  Stages: scatter over lane bits 0,1 (quad_perm),
// bit 2 (row_shl/shr:4), then add over bit 3 (row_ror:8).  Rows are added by the reader.
// Measured cost on config 2: 10.7 % of the kernel time (14 calls per optimiser step).
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

__device__ __forceinline__ void load_rec(const float* p, float (&r)[16]) {
    const float4* q = reinterpret_cast<const float4*>(p);
    const float4 a = q[0], b = q[1], c = q[2], d = q[3];
    r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w; r[4] = b.x; r[5] = b.y; r[6] = b.z; r[7] = b.w;
    r[8] = c.x; r[9] = c.y; r[10] = c.z; r[11] = c.w; r[12] = d.x; r[13] = d.y; r[14] = d.z; r[15] = d.w;
}


}  // namespace cpf
