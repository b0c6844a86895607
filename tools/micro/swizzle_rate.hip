// Micro-benchmark: can the LDS crossbar (ds_swizzle_b32 / ds_bpermute_b32: no memory access, issued on the LDS pipe) carry the
// cross-lane moves that cost a double VALU slot as DPP?  Same harness as issue_rate.hip (W waves per SIMD resident and released
// together, cycles from s_memtime, wall-clock aggregate).
//   swz       ds_swizzle_b32 (lane ^ 1), ILP independent chains            bperm     ds_bpermute_b32 (lane ^ 32)
//   gate      the lane-qubit Rx gate as shipped: 2 x v_mul_f32_dpp + v_pk_fma_f32
//   gate_swz  the same gate with its partner fetched by 2 x ds_swizzle_b32 (+ v_pk_mul_f32 + v_pk_fma_f32)
//   mix       per operation one ds_swizzle_b32 on one chain and two v_pk_fma_f32 on another: do the two pipes overlap?
//   pk2       the two v_pk_fma_f32 of `mix` alone
// build: hipcc --offload-arch=gfx950 -O3 -o swizzle_rate swizzle_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int CTRL> __device__ __forceinline__ float dppz(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float swz1(float x) { return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(x), 0x041f)); }   // bit-mask mode: and 0x1f, or 0, xor 1
enum { SWZ = 0, BPERM = 1, GATE = 2, GATE_SWZ = 3, MIX = 4, PK2 = 5 };
template <int MODE> __host__ __device__ constexpr int vector_insts() { return MODE == GATE ? 3 : (MODE == GATE_SWZ ? 2 : (MODE == MIX || MODE == PK2 ? 2 : 0)); }
template <int MODE> __host__ __device__ constexpr int lds_insts() { return MODE == SWZ || MODE == BPERM || MODE == MIX ? 1 : (MODE == GATE_SWZ ? 2 : 0); }

template <int MODE, int ILP>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, float* sink, int reps, float c, float s) {
    extern __shared__ float keep_off[];               // 96 KiB: one workgroup per CU
    if (reps < 0) keep_off[threadIdx.x] = c;
    f2 x[ILP], y[ILP];
#pragma unroll
    for (int i = 0; i < ILP; i++) { x[i] = f2{threadIdx.x * 0.01f + i, 0.5f - i}; y[i] = f2{0.25f * i, 1.f}; }
    const f2 cc = {c, c}, ss = {s, -s};
    const int partner = ((threadIdx.x & 63) ^ 32) << 2;
    unsigned long long t0, t1, r0, r1;
    __syncthreads();
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
#pragma unroll
            for (int i = 0; i < ILP; i++) {
                if constexpr (MODE == SWZ) x[i].x = swz1(x[i].x);
                else if constexpr (MODE == BPERM) x[i].x = __int_as_float(__builtin_amdgcn_ds_bpermute(partner, __float_as_int(x[i].x)));
                else if constexpr (MODE == GATE) {
                    f2 t;
                    t.x = s * dppz<0xB1>(x[i].y);
                    t.y = -s * dppz<0xB1>(x[i].x);
                    x[i] = __builtin_elementwise_fma(cc, x[i], t);
                } else if constexpr (MODE == GATE_SWZ) {
                    const f2 p = {swz1(x[i].y), swz1(x[i].x)};
                    x[i] = __builtin_elementwise_fma(cc, x[i], ss * p);
                } else if constexpr (MODE == MIX) {
                    y[i].x = swz1(y[i].x);
                    x[i] = __builtin_elementwise_fma(x[i], cc, ss);
                    x[i] = __builtin_elementwise_fma(x[i], cc, ss);
                } else {
                    x[i] = __builtin_elementwise_fma(x[i], cc, ss);
                    x[i] = __builtin_elementwise_fma(x[i], cc, ss);
                }
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < ILP; i++) acc += x[i].x + x[i].y + y[i].x + y[i].y;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        out[2 * w] = t1 - t0; out[2 * w + 1] = r1 - r0;
    }
}

template <int MODE, int ILP>
void run(const char* name, int wps) {
    const int cus = 256, wpb = 4 * wps, waves = cus * wpb, reps = 300;
    const size_t lds = 96 * 1024;
    unsigned long long* out; float* sink;
    (void)hipMalloc(&out, waves * 16); (void)hipMalloc(&sink, (size_t)waves * 64 * 4);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE, ILP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0.f;
    for (int it = 0; it < 2; it++) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<MODE, ILP>), dim3(cus), dim3(64 * wpb), lds, 0, out, sink, reps, 0.999f, 0.01f);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<unsigned long long> h(2 * waves);
    (void)hipMemcpy(h.data(), out, waves * 16, hipMemcpyDeviceToHost);
    std::vector<double> cyc(waves), clk(waves);
    for (int b = 0; b < waves; b++) { cyc[b] = (double)h[2 * b]; clk[b] = (double)h[2 * b] / (double)h[2 * b + 1] * 0.1; }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    const double ops = 16.0 * reps * ILP, ghz = clk[waves / 2];
    const double wall = (double)ms * 1e-3 * ghz * 1e9 / (ops * wps);          // cycles per operation per SIMD from the wall clock
    printf("%-9s ILP %d waves/SIMD %d: %7.2f cycles/op/wave (median), %7.2f cycles/op/SIMD, wall-clock %7.2f cycles/op/SIMD  [%d vector + %d LDS-pipe instructions per op; "
           "LDS pipe: %.2f cycles per instruction per CU]  clock %.2f GHz\n", name, ILP, wps, cyc[waves / 2] / ops, cyc[waves / 2] / ops / wps, wall,
           vector_insts<MODE>(), lds_insts<MODE>(), lds_insts<MODE>() ? wall / 4 / lds_insts<MODE>() : 0.0, ghz);
    (void)hipFree(out); (void)hipFree(sink);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
}
template <int MODE> void sweep(const char* name) {
    for (int w : {1, 2, 4}) { run<MODE, 1>(name, w); run<MODE, 4>(name, w); run<MODE, 8>(name, w); }
}
int main() {
    sweep<SWZ>("swz"); sweep<BPERM>("bperm"); sweep<GATE>("gate"); sweep<GATE_SWZ>("gate_swz"); sweep<PK2>("pk2"); sweep<MIX>("mix");
    return 0;
}
