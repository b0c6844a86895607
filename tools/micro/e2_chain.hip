// Micro-benchmark: cost per SAMPLE-gate per SIMD of the 3-qubit rotation chain in two layouts.
//   E1: one amplitude per lane (64 lanes = one sample's 8x8 matrix), every qubit on lanes: 2 x v_mul_f32_dpp + v_pk_fma_f32
//       per gate, one dependent chain per wave; run with 2 waves per SIMD (what S = 1024 gives the two-ended kernel).
//   E2: two amplitudes per lane (32 lanes per sample, two samples per wave), one qubit in registers: gates on it are
//       2 x (v_pk_mul_f32 + v_pk_fma_f32), the others 2 x (2 x v_mul_f32_dpp + v_pk_fma_f32); two independent chains per
//       wave; run with 1 wave per SIMD (the same 1024 samples).
// Reported: shader cycles (s_memtime) per gate application per sample, per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int CTRL> __device__ __forceinline__ float dppz(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
template <int CTRL> __device__ __forceinline__ f2 rx_lane(f2 x, float c, float s) {
    f2 t;
    t.x = s * dppz<CTRL>(x.y);
    t.y = -s * dppz<CTRL>(x.x);
    return pk_fma(f2{c, c}, x, t);
}
template <int E2>
__global__ __launch_bounds__(64) void k(unsigned long long* out, float* sink, int reps, const float* cs) {
    f2 x[2];
    x[0] = f2{threadIdx.x * 0.01f, 0.5f}; x[1] = f2{0.25f, threadIdx.x * 0.02f};
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const float c = cs[2 * u], s = cs[2 * u + 1];          // uniform loads stand in for the LDS records
            if constexpr (!E2) {      // three gates, all on lane qubits
                x[0] = rx_lane<0xB1>(x[0], c, s); x[0] = rx_lane<0x4E>(x[0], c, s); x[0] = rx_lane<0x128>(x[0], c, s);
            } else {                  // three gates: two on lane qubits (both elements), one on the register qubit
                x[0] = rx_lane<0xB1>(x[0], c, s); x[1] = rx_lane<0xB1>(x[1], c, s);
                x[0] = rx_lane<0x4E>(x[0], c, s); x[1] = rx_lane<0x4E>(x[1], c, s);
                const f2 a = x[0], b = x[1];
                x[0] = pk_fma(f2{s, -s}, f2{b.y, b.x}, f2{c, c} * a);
                x[1] = pk_fma(f2{s, -s}, f2{a.y, a.x}, f2{c, c} * b);
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    sink[blockIdx.x * 64 + threadIdx.x] = x[0].x + x[0].y + x[1].x + x[1].y;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}
template <int E2> void run(int wps) {
    const int blocks = 256 * 4 * wps, reps = 2000;
    unsigned long long* out; float *sink, *cs;
    (void)hipMalloc(&out, blocks * 8); (void)hipMalloc(&sink, blocks * 64 * 4); (void)hipMalloc(&cs, 64);
    std::vector<float> h(16); for (int i = 0; i < 8; i++) { h[2 * i] = 0.9999f; h[2 * i + 1] = 0.01f; }
    (void)hipMemcpy(cs, h.data(), 64, hipMemcpyHostToDevice);
    float ms = 0; hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int it = 0; it < 3; it++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<E2>), dim3(blocks), dim3(64), 0, 0, out, sink, reps, cs);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<unsigned long long> c(blocks); (void)hipMemcpy(c.data(), out, blocks * 8, hipMemcpyDeviceToHost);
    std::sort(c.begin(), c.end());
    const double gates = 24.0 * reps, samples_per_wave = E2 ? 2 : 1;
    printf("%s waves/SIMD %d: %.1f cycles per gate per wave (median) -> %.2f cycles per sample-gate per SIMD; wall %.3f ms -> %.2f ns per sample-gate per SIMD\n",
           E2 ? "E2 (2 samples/wave)" : "E1 (1 sample/wave) ", wps, c[blocks / 2] / gates, c[blocks / 2] / gates * wps / (samples_per_wave * wps) ,
           ms, ms * 1e6 / (gates * samples_per_wave * wps));
}
int main() { for (int w : {1, 2, 4}) { run<0>(w); run<1>(w); } return 0; }
