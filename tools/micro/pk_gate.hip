// Micro-benchmark: the rotation gate with (re, im) kept as a register pair: 2 x v_mul_f32_dpp + 1 x v_pk_fma_f32
// against the scalar form 2 x v_mul_f32_dpp + 2 x v_fmac_f32; complex multiply as 2 packed ops against 4 scalar ones.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int CTRL> __device__ __forceinline__ float dppz(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
template <int CTRL> __device__ __forceinline__ void rx_pk(f2& x, f2 cs) {
    f2 t;
    t.x = cs.y * dppz<CTRL>(x.y);
    t.y = -cs.y * dppz<CTRL>(x.x);
    x = __builtin_elementwise_fma(f2{cs.x, cs.x}, x, t);
}
template <int CTRL> __device__ __forceinline__ void rx_sc(float& re, float& im, float c, float s) {
    const float tr = s * dppz<CTRL>(im), ti = s * dppz<CTRL>(re);
    const float nr = fmaf(c, re, tr), ni = fmaf(c, im, -ti);
    re = nr; im = ni;
}
// x *= rho (complex)
__device__ __forceinline__ void cmul_pk(f2& x, f2 rho) {
    f2 a = f2{rho.x, rho.x} * x;                                   // (rx re, rx im)
    x = __builtin_elementwise_fma(f2{-rho.y, rho.y}, f2{x.y, x.x}, a);  // (-ry im, ry re)
}
__device__ __forceinline__ void cmul_sc(float& re, float& im, float rx, float ry) {
    const float nr = fmaf(-ry, im, rx * re), ni = fmaf(ry, re, rx * im);
    re = nr; im = ni;
}
template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, const float* coef, int reps) {
    f2 x = {threadIdx.x * 0.01f, 1.f - threadIdx.x * 0.01f};
    float re = x.x, im = x.y;
    __shared__ float lds[8];
    if (threadIdx.x < 8) lds[threadIdx.x] = coef[threadIdx.x];
    __syncthreads();
    const f2 cs0 = {lds[0], lds[1]}, cs1 = {lds[2], lds[3]}, cs2 = {lds[4], lds[5]};          // VGPRs, as in the kernels
    f2 rho = {coef[6] + threadIdx.x * 1e-3f, coef[7]};
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (MODE == 0) { rx_pk<0xB1>(x, cs0); rx_pk<0x4E>(x, cs1); rx_pk<0x128>(x, cs2); cmul_pk(x, rho); }
            else { rx_sc<0xB1>(re, im, cs0.x, cs0.y); rx_sc<0x4E>(re, im, cs1.x, cs1.y); rx_sc<0x128>(re, im, cs2.x, cs2.y); cmul_sc(re, im, rho.x, rho.y); }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = x.x + x.y + re + im;
}
template <int MODE> void run(const char* name, int wps, float* coef) {
    const int blocks = 256 * 4 * wps, reps = 4000;
    float* out; (void)hipMalloc(&out, blocks * 64 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int it = 0; it < 3; it++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, coef, reps);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double groups = 4.0 * reps * wps;          // (3 gates + 1 complex multiply) per SIMD
    printf("%-8s waves/SIMD %d: %.3f ms -> %.2f ns per (3 gates + cmul) per SIMD\n", name, wps, ms, ms * 1e6 / groups);
    (void)hipFree(out);
}
int main() {
    float h[8] = {0.8f, 0.6f, 0.6f, 0.8f, 0.96f, 0.28f, 0.7071f, 0.7071f}, *coef;
    (void)hipMalloc(&coef, 32); (void)hipMemcpy(coef, h, 32, hipMemcpyHostToDevice);
    for (int w : {1, 2, 4}) { run<0>("packed", w, coef); run<1>("scalar", w, coef); }
    return 0;
}
