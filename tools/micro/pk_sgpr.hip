// Micro-test: v_pk_fma_f32 with a SCALAR register pair as src0 - which dwords do the two halves read under op_sel /
// op_sel_hi, and do the neg modifiers apply?  (The dense-block kernel wants its block matrix G in SGPRs: the broadcast
// factor of each complex multiply-add would then cost no vector-register traffic at all.)
// Also: issue cost of the SGPR form against the VGPR form.
// build: hipcc --offload-arch=gfx950 -O3 -o pk_sgpr.bin pk_sgpr.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ void k_sem(const float* g, float* out) {
    // g[0], g[1] -> an aligned SGPR pair through a scalar (constant address space) load
    const f2 gs = *reinterpret_cast<const f2 __attribute__((address_space(4)))*>((const __attribute__((address_space(4))) void*)g);
    const f2 b = {3.f + threadIdx.x, 5.f + 2.f * threadIdx.x};
    f2 r0 = {0.f, 0.f}, r1 = {0.f, 0.f}, r2 = {0.f, 0.f}, r3 = {0.f, 0.f}, r4 = {0.f, 0.f};
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(r0) : "s"(gs), "v"(b));                                  // (lo, lo) * (b.x, b.y) ?
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(r1) : "s"(gs), "v"(b));                    // (hi, hi) * (b.x, b.y) ?
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[1,0,0]" : "+v"(r2) : "s"(gs), "v"(b));     // (hi, -hi) * (b.y, b.x) ?
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,1,1]" : "+v"(r3) : "s"(gs), "v"(b));                    // (lo, hi) * (b.x, b.y) ?
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "+v"(r4) : "s"(gs), "v"(b));     // (-hi, hi) * (b.y, b.x) ?
    float* o = out + 10 * threadIdx.x;
    o[0] = r0.x; o[1] = r0.y; o[2] = r1.x; o[3] = r1.y; o[4] = r2.x; o[5] = r2.y; o[6] = r3.x; o[7] = r3.y; o[8] = r4.x; o[9] = r4.y;
}

template <int SGPR>
__global__ __launch_bounds__(64) void k_rate(const float* g, unsigned long long* out, float* sink, int reps) {
    const f2 gs = *reinterpret_cast<const f2 __attribute__((address_space(4)))*>((const __attribute__((address_space(4))) void*)g);
    f2 gv = {g[threadIdx.x & 1], g[1]};
    f2 acc[4], b[4];
    for (int i = 0; i < 4; i++) { acc[i] = f2{0.f, 0.f}; b[i] = f2{1.f + threadIdx.x + i, 2.f - i}; }
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if constexpr (SGPR) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[1,0,0]" : "+v"(acc[i]) : "s"(gs), "v"(b[i]));
                else asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[1,0,0]" : "+v"(acc[i]) : "v"(gv), "v"(b[i]));
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float s = 0.f;
    for (int i = 0; i < 4; i++) s += acc[i].x + acc[i].y;
    sink[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

int main() {
    float hg[2] = {7.f, 11.f}, *dg, *dout, *sink;
    unsigned long long* dt;
    (void)hipMalloc(&dg, 8); (void)hipMalloc(&dout, 64 * 10 * 4); (void)hipMalloc(&sink, 2048 * 64 * 4); (void)hipMalloc(&dt, 2048 * 8);
    (void)hipMemcpy(dg, hg, 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_sem, dim3(1), dim3(64), 0, 0, dg, dout);
    std::vector<float> o(640);
    (void)hipMemcpy(o.data(), dout, 2560, hipMemcpyDeviceToHost);
    const int t = 2;                               // thread 2: b = (5, 9)
    const float bx = 3.f + t, by = 5.f + 2.f * t, lo = hg[0], hi = hg[1];
    const float* r = &o[10 * t];
    printf("g = (lo %g, hi %g), b = (%g, %g)\n", lo, hi, bx, by);
    printf("op_sel 0 / hi 0            : (%g, %g)  expect (lo bx, lo by) = (%g, %g)\n", r[0], r[1], lo * bx, lo * by);
    printf("op_sel 1 / hi 1            : (%g, %g)  expect (hi bx, hi by) = (%g, %g)\n", r[2], r[3], hi * bx, hi * by);
    printf("(hi, -hi) x (b.y, b.x)     : (%g, %g)  expect (%g, %g)\n", r[4], r[5], hi * by, -hi * bx);
    printf("op_sel 0 / hi 1 (lo, hi)   : (%g, %g)  expect (%g, %g)\n", r[6], r[7], lo * bx, hi * by);
    printf("(-hi, hi) x (b.y, b.x)     : (%g, %g)  expect (%g, %g)\n", r[8], r[9], -hi * by, hi * bx);
    for (int sg = 0; sg < 2; sg++) {
        const int blocks = 2048, reps = 300;
        for (int it = 0; it < 2; it++) {
            if (sg) hipLaunchKernelGGL(k_rate<1>, dim3(blocks), dim3(64), 0, 0, dg, dt, sink, reps);
            else hipLaunchKernelGGL(k_rate<0>, dim3(blocks), dim3(64), 0, 0, dg, dt, sink, reps);
        }
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks);
        (void)hipMemcpy(h.data(), dt, blocks * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        printf("%s src0: %.2f cycles per v_pk_fma_f32 per wave (median, 2 waves per SIMD, ILP 4)\n", sg ? "SGPR" : "VGPR", (double)h[blocks / 2] / (64.0 * reps));
    }
    return 0;
}
