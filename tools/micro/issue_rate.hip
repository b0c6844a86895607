// Micro-benchmark: VALU issue cost in SHADER CYCLES (s_memtime), not wall time at an assumed clock.
// For W waves per SIMD (W = 1, 2, 4, 8) and ILP independent chains per wave, every wave runs the same unrolled
// body and stamps s_memtime around it; reported: cycles per instruction per wave and per SIMD (= / W), the shader clock
// (s_memtime ticks / s_memrealtime 100 MHz ticks), AND the wall-clock aggregate (instructions of the whole launch / HIP-event
// time / 1024 SIMDs / clock), which does not depend on how the waves were placed.
// Residency is built in, then verified: ONE workgroup of 4 W waves per CU (two of 16 waves for W = 8) (its waves go round-robin over the CU's four
// SIMDs: W per SIMD), a second workgroup kept off the CU by 96 KiB of LDS, all waves released together by a barrier, and
// every wave reports HW_ID / XCC_ID: the run prints how many (XCC, SE, CU, SIMD) slots hold exactly W waves.  (Round 2
// launched 1024 W single-wave workgroups, which the dispatcher does not spread W per SIMD within the timed window: its 4- and
// 8-wave rows were mis-normalised.)
//   fma       v_fma_f32                       pk        v_pk_fma_f32
//   dppmul    v_mul_f32_dpp quad_perm         gate      the Rx gate of cpf_device.h: 2 x v_mul_f32_dpp + v_pk_fma_f32
//   gate_reg  the register-resident Rx gate: v_pk_mul_f32 + v_pk_fma_f32 on an amplitude pair
//   perm32/16 v_permlane32_swap / v_permlane16_swap on a register pair (each followed by nothing: the swap chain itself)
// build: hipcc --offload-arch=gfx950 -O3 -o issue_rate issue_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <map>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int CTRL> __device__ __forceinline__ float dppz(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
enum { FMA = 0, PK = 1, DPPMUL = 2, GATE = 3, GATE_REG = 4, PERM32 = 5, PERM16 = 6 };
template <int MODE> __host__ __device__ constexpr int insts_per_op() { return MODE == GATE ? 3 : (MODE == GATE_REG ? 4 : 1); }

template <int MODE, int ILP>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, float* sink, unsigned* hwid, int reps, float c, float s) {
    extern __shared__ float keep_off[];               // 96 KiB: one workgroup per CU
    if (reps < 0) keep_off[threadIdx.x] = c;
    f2 x[ILP], y[ILP];
#pragma unroll
    for (int i = 0; i < ILP; i++) { x[i] = f2{threadIdx.x * 0.01f + i, 0.5f - i}; y[i] = f2{0.25f * i, 1.f}; }
    const f2 cc = {c, c}, ss = {s, -s};
    unsigned long long t0, t1, r0, r1;
    __syncthreads();
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
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[2 * w] = t1 - t0; out[2 * w + 1] = r1 - r0;
        hwid[2 * w] = hw; hwid[2 * w + 1] = xcc;
    }
}

template <int MODE, int ILP>
void run(const char* name, int wps) {
    // a workgroup holds at most 16 waves: 8 waves per SIMD = two 16-wave workgroups per CU (64 KiB of LDS each: a third does not fit)
    const int per_cu = wps > 4 ? 2 : 1, cus = 256 * per_cu, wpb = 4 * wps / per_cu, waves = cus * wpb, reps = 500;
    const size_t lds = (per_cu == 1 ? 96 : 64) * 1024;
    unsigned long long* out; float* sink; unsigned* hwid;
    (void)hipMalloc(&out, waves * 16); (void)hipMalloc(&sink, (size_t)waves * 64 * 4); (void)hipMalloc(&hwid, waves * 8);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE, ILP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0.f;
    for (int it = 0; it < 2; it++) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<MODE, ILP>), dim3(cus), dim3(64 * wpb), lds, 0, out, sink, hwid, reps, 0.999f, 0.01f);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<unsigned long long> h(2 * waves);
    std::vector<unsigned> hw(2 * waves);
    (void)hipMemcpy(h.data(), out, waves * 16, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hw.data(), hwid, waves * 8, hipMemcpyDeviceToHost);
    std::vector<double> cyc(waves), clk(waves);
    for (int b = 0; b < waves; b++) { cyc[b] = (double)h[2 * b]; clk[b] = (double)h[2 * b] / (double)h[2 * b + 1] * 0.1; }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    // residency: waves per (XCC, SE, CU, SIMD)
    std::map<unsigned, int> slots;
    for (int b = 0; b < waves; b++) slots[((hw[2 * b + 1] & 15) << 16) | (((hw[2 * b] >> 13) & 7) << 8) | (((hw[2 * b] >> 8) & 15) << 4) | ((hw[2 * b] >> 4) & 3)]++;
    int exact = 0;
    for (const auto& kv : slots) exact += kv.second == wps;
    const double insts = 16.0 * reps * ILP * insts_per_op<MODE>();
    const double ghz = clk[waves / 2];
    const double wall = (double)ms * 1e-3 * ghz * 1e9 / (insts * wps);        // cycles per instruction per SIMD from the wall clock (launch overhead included)
    printf("%-9s ILP %d waves/SIMD %d: %.2f cycles/inst/wave (median), %.2f cycles/inst/SIMD; wall-clock %.2f cycles/inst/SIMD (%.3f ms); "
           "%d SIMDs used, %d with exactly %d wave(s); clock %.2f GHz\n", name, ILP, wps, cyc[waves / 2] / insts, cyc[waves / 2] / insts / wps, wall, ms,
           (int)slots.size(), exact, wps, ghz);
    (void)hipFree(out); (void)hipFree(sink); (void)hipFree(hwid);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
}
template <int MODE> void sweep(const char* name) {
    for (int w : {1, 2, 4, 8}) { run<MODE, 1>(name, w); run<MODE, 2>(name, w); run<MODE, 4>(name, w); run<MODE, 8>(name, w); }
}
int main() {
    sweep<FMA>("fma"); sweep<PK>("pk"); sweep<DPPMUL>("dppmul"); sweep<PERM32>("perm32"); sweep<PERM16>("perm16"); sweep<GATE>("gate"); sweep<GATE_REG>("gate_reg");
    return 0;
}
