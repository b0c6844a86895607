// Micro-benchmark: saturated issue cost (cycles per wave64 instruction per SIMD) of the VALU forms the 3-qubit
// kernels use: plain v_fma_f32, v_mul_f32_dpp quad_perm / row_ror, v_add_f32_dpp with bank mask, v_pk_fma_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL> __device__ __forceinline__ float dppz(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, int reps, float c) {
    float v[8];
    for (int i = 0; i < 8; i++) v[i] = threadIdx.x * 0.01f + i;
    typedef float float2v __attribute__((ext_vector_type(2)));
    float2v p[4];
    for (int i = 0; i < 4; i++) p[i] = float2v{v[2 * i], v[2 * i + 1]};
    const float2v cc = {c, c};
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (MODE == 0) v[i] = fmaf(v[i], c, 0.5f);                         // 8 independent chains of v_fma
                else if (MODE == 1) v[i] = c * dppz<0xB1>(v[i]);                   // v_mul_f32_dpp quad_perm
                else if (MODE == 2) v[i] = c * dppz<0x128>(v[i]);                  // v_mul_f32_dpp row_ror:8
                else if (MODE == 3) v[i] = v[i] + dppz<0x4E>(v[i]);               // v_add_f32_dpp
                else if (MODE == 4) v[i] = c * v[i];                               // v_mul_f32
            }
            if (MODE == 5) {
#pragma unroll
                for (int i = 0; i < 4; i++) p[i] = __builtin_elementwise_fma(p[i], cc, cc);   // v_pk_fma_f32
#pragma unroll
                for (int i = 0; i < 4; i++) p[i] = __builtin_elementwise_fma(p[i], cc, cc);
            }
        }
    }
    float acc = 0.f;
    for (int i = 0; i < 8; i++) acc += v[i];
    for (int i = 0; i < 4; i++) acc += p[i].x + p[i].y;
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}
template <int MODE> void run(const char* name, int wps) {
    const int blocks = 256 * 4 * wps, reps = 4000;
    float* out; (void)hipMalloc(&out, blocks * 64 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int it = 0; it < 3; it++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, reps, 0.999f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double insts = 32.0 * reps * wps;          // per SIMD
    printf("%-28s waves/SIMD %d: %.3f ms -> %.2f ns per instruction per SIMD (x2.4 GHz = %.2f cycles)\n", name, wps, ms, ms * 1e6 / insts, ms * 1e6 / insts * 2.4);
    (void)hipFree(out);
}
int main() {
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma_f32 x8 chains", w); run<4>("v_mul_f32", w); run<1>("v_mul_f32_dpp quad_perm", w); run<2>("v_mul_f32_dpp row_ror:8", w);
        run<3>("v_add_f32_dpp quad_perm", w); run<5>("v_pk_fma_f32 (2 flop-lanes)", w);
    }
    return 0;
}
