// Micro-benchmark: VALU issue cost in SHADER CYCLES (s_memtime), not wall time at an assumed clock.
// For W waves per SIMD (W = 1, 2, 4, 8) and ILP independent chains per wave, every wave runs the same unrolled
// body and stamps s_memtime around it; reported: cycles per instruction per wave and per SIMD (= / W), and the
// shader clock (s_memtime ticks / s_memrealtime 100 MHz ticks).
//   fma       v_fma_f32                       pk        v_pk_fma_f32
//   dppmul    v_mul_f32_dpp quad_perm         gate      the Rx gate of cpf_device.h: 2 x v_mul_f32_dpp + v_pk_fma_f32
//   gate_reg  the register-resident Rx gate: v_pk_mul_f32 + v_pk_fma_f32 on an amplitude pair
//   perm32/16 v_permlane32_swap / v_permlane16_swap on a register pair (each followed by nothing: the swap chain itself)
// build: hipcc --offload-arch=gfx950 -O3 -o issue_rate issue_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int CTRL> __device__ __forceinline__ float dppz(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
enum { FMA = 0, PK = 1, DPPMUL = 2, GATE = 3, GATE_REG = 4, PERM32 = 5, PERM16 = 6 };
template <int MODE> __host__ __device__ constexpr int insts_per_op() { return MODE == GATE ? 3 : (MODE == GATE_REG ? 4 : 1); }

template <int MODE, int ILP>
__global__ __launch_bounds__(64) void k(unsigned long long* out, float* sink, int reps, float c, float s) {
    f2 x[ILP], y[ILP];
#pragma unroll
    for (int i = 0; i < ILP; i++) { x[i] = f2{threadIdx.x * 0.01f + i, 0.5f - i}; y[i] = f2{0.25f * i, 1.f}; }
    const f2 cc = {c, c}, ss = {s, -s};
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
#pragma unroll
            for (int i = 0; i < ILP; i++) {
                if constexpr (MODE == FMA) x[i].x = fmaf(x[i].x, c, s);
                else if constexpr (MODE == PK) x[i] = __builtin_elementwise_fma(x[i], cc, ss);
                else if constexpr (MODE == DPPMUL) x[i].x = c * dppz<0xB1>(x[i].x);
                else if constexpr (MODE == PERM32) {
                    auto r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(x[i].x), __float_as_uint(x[i].y), false, false);
                    x[i].x = __uint_as_float(r2[0]); x[i].y = __uint_as_float(r2[1]);
                } else if constexpr (MODE == PERM16) {
                    auto r2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(x[i].x), __float_as_uint(x[i].y), false, false);
                    x[i].x = __uint_as_float(r2[0]); x[i].y = __uint_as_float(r2[1]);
                }
                else if constexpr (MODE == GATE) {
                    f2 t;
                    t.x = s * dppz<0xB1>(x[i].y);
                    t.y = -s * dppz<0xB1>(x[i].x);
                    x[i] = __builtin_elementwise_fma(cc, x[i], t);
                } else {
                    const f2 a = x[i], b = y[i];
                    x[i] = __builtin_elementwise_fma(ss, f2{b.y, b.x}, cc * a);
                    y[i] = __builtin_elementwise_fma(ss, f2{a.y, a.x}, cc * b);
                }
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < ILP; i++) acc += x[i].x + x[i].y + y[i].x + y[i].y;
    sink[blockIdx.x * 64 + threadIdx.x] = acc;
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int MODE, int ILP>
void run(const char* name, int wps) {
    const int blocks = 256 * 4 * wps, reps = 500;
    unsigned long long* out; float* sink;
    (void)hipMalloc(&out, blocks * 16); (void)hipMalloc(&sink, blocks * 64 * 4);
    for (int it = 0; it < 2; it++) hipLaunchKernelGGL((k<MODE, ILP>), dim3(blocks), dim3(64), 0, 0, out, sink, reps, 0.999f, 0.01f);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(2 * blocks);
    (void)hipMemcpy(h.data(), out, blocks * 16, hipMemcpyDeviceToHost);
    std::vector<double> cyc(blocks), clk(blocks);
    for (int b = 0; b < blocks; b++) { cyc[b] = (double)h[2 * b]; clk[b] = (double)h[2 * b] / (double)h[2 * b + 1] * 0.1; }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    const double insts = 16.0 * reps * ILP * insts_per_op<MODE>();
    printf("%-9s ILP %d waves/SIMD %d: %.2f cycles/inst/wave (median), %.2f cycles/inst/SIMD, clock %.2f GHz\n", name, ILP, wps,
           cyc[blocks / 2] / insts, cyc[blocks / 2] / insts / wps, clk[blocks / 2]);
    (void)hipFree(out); (void)hipFree(sink);
}
template <int MODE> void sweep(const char* name) {
    for (int w : {1, 2, 4, 8}) { run<MODE, 1>(name, w); run<MODE, 2>(name, w); run<MODE, 4>(name, w); run<MODE, 8>(name, w); }
}
int main() {
    sweep<FMA>("fma"); sweep<PK>("pk"); sweep<DPPMUL>("dppmul"); sweep<PERM32>("perm32"); sweep<PERM16>("perm16"); sweep<GATE>("gate"); sweep<GATE_REG>("gate_reg");
    return 0;
}
