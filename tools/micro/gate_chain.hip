// Micro-benchmark: cycles per gate of the 3-qubit kernels' rotation pattern (2 x v_mul_f32_dpp + 2 x v_fmac_f32 per gate)
// for NCH independent matrices per wave and 1 / 2 / 4 / 8 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -o gate_chain gate_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../cpflow_amd/csrc/cpf_device.h"
using namespace cpf;
using L = Lay<3, 0>;

template <int NCH>
__global__ __launch_bounds__(64) void k_chain(float* out, unsigned long long* cyc, int reps, float c, float s) {
    State<L> st[NCH];
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < NCH; k++) {
        for (int i = 0; i < L::NLR; i++) { st[k].lb[i] = (lane >> L::row_lane_bit(i)) & 1; st[k].sg[i] = st[k].lb[i] ? 1.f : -1.f; }
        st[k].pr[0] = 0.01f * lane + k; st[k].pi[0] = 0.02f * lane - k;
    }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int g = 0; g < 4; g++) {                     // 12 gates per iteration: x0 y1 x2 y0 x1 y2 ...
#pragma unroll
            for (int k = 0; k < NCH; k++) rot_apply<L, OP_RX, 0>(st[k].pr, st[k].pi, c, s, st[k]);
#pragma unroll
            for (int k = 0; k < NCH; k++) rot_apply<L, OP_RY, 1>(st[k].pr, st[k].pi, c, s, st[k]);
#pragma unroll
            for (int k = 0; k < NCH; k++) rot_apply<L, OP_RX, 2>(st[k].pr, st[k].pi, c, s, st[k]);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; k++) acc += st[k].pr[0] + st[k].pi[0];
    out[blockIdx.x * 64 + lane] = acc;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NCH>
void run(int waves_per_simd) {
    const int blocks = 256 * 4 * waves_per_simd, reps = 2000;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, blocks * 64 * 4); hipMalloc(&cyc, blocks * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 2; it++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_chain<NCH>, dim3(blocks), dim3(64), 0, 0, out, cyc, reps, 0.8f, 0.6f);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += v; mean /= blocks;
    const double gates = 12.0 * reps * NCH;
    // s_memtime counts at 100 MHz on gfx9: convert with the event time instead
    printf("chains/wave %d waves/SIMD %d: %.3f ms  -> %.2f ns per gate per wave, %.2f cycles@2.4GHz per gate (per wave), SIMD cycles per instruction %.2f  (memtime ticks/gate %.3f)\n",
           NCH, waves_per_simd, ms, ms * 1e6 / gates, ms * 1e6 / gates * 2.4, ms * 1e6 * 2.4 / (gates * 4 * waves_per_simd), mean / gates);
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int w : {1, 2, 4, 8}) { run<1>(w); run<2>(w); run<4>(w); }
    return 0;
}
