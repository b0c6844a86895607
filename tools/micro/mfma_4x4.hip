// Micro-benchmark for the dense-block kernel on the matrix cores (cpf_mfma.h): v_mfma_f32_4x4x1_16b_f32.
//   1. operand / result lane maps, checked with exact integer data (asymmetric A and B);
//   2. issue cost in shader cycles (s_memtime) of a stream of independent 4x4x1 MFMAs, one wave per SIMD;
//   3. how many independent v_pk_fma_f32 fit between two MFMAs before the stream slows down (co-issue of the vector
//      and the matrix pipe from ONE wave) - the number that decides whether the environment accumulation can hide
//      under the three matrix applications of a block.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_4x4 mfma_4x4.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void k_layout(const float* a, const float* b, float* d) {
    const int l = threadIdx.x;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
    for (int r = 0; r < 4; r++) d[4 * l + r] = acc[r];
}

// NM MFMAs (independent accumulators) followed by NV packed FMAs (independent chains), repeated
template <int NM, int NV>
__global__ __launch_bounds__(64) void k_rate(unsigned long long* out, float* sink, int reps, float c) {
    f4 acc[8];
    f2 x[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { acc[i] = f4{0.f, 0.f, 0.f, 0.f}; x[i] = f2{threadIdx.x * 0.01f + i, 0.5f - i}; }
    const float av = threadIdx.x * 0.001f, bv = 1.f - threadIdx.x * 0.002f;
    const f2 cc = {c, c}, ss = {0.01f, -0.01f};
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int i = 0; i < NM; i++) acc[(u * NM + i) % 8] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, acc[(u * NM + i) % 8], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NV; i++) x[(u * NV + i) % 8] = __builtin_elementwise_fma(x[(u * NV + i) % 8], cc, ss);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + x[i].x + x[i].y;
    sink[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

// 8-wave workgroups (two waves on every SIMD of a CU: wave w sits on SIMD w % 4): waves 0..3 issue only MFMAs, waves 4..7 only
// packed FMAs (ROLES = 1), or all eight the same stream (ROLES = 0: mode 0 = MFMA only, 1 = FMA only).  Do the two pipes run
// side by side when the instructions come from DIFFERENT waves of a SIMD?
template <int ROLES>
__global__ __launch_bounds__(512) void k_pair(unsigned long long* out, float* sink, int reps, float c, int mode) {
    f4 acc[8];
    f2 x[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { acc[i] = f4{0.f, 0.f, 0.f, 0.f}; x[i] = f2{threadIdx.x * 0.01f + i, 0.5f - i}; }
    const float av = threadIdx.x * 0.001f, bv = 1.f - threadIdx.x * 0.002f;
    const f2 cc = {c, c}, ss = {0.01f, -0.01f};
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool do_mfma = ROLES ? (wave < 4) : (mode == 0);
    unsigned long long t0, t1;
    __syncthreads();
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    if (do_mfma) {
        for (int r = 0; r < reps; r++) {
#pragma unroll
            for (int u = 0; u < 32; u++) acc[u % 8] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, acc[u % 8], 0, 0, 0);
        }
    } else {
        for (int r = 0; r < reps; r++) {
#pragma unroll
            for (int u = 0; u < 64; u++) x[u % 8] = __builtin_elementwise_fma(x[u % 8], cc, ss);
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + x[i].x + x[i].y;
    sink[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}
template <int ROLES>
void run_pair(int mode, const char* what) {
    const int blocks = 256, reps = 200;
    unsigned long long* out; float* sink;
    (void)hipMalloc(&out, blocks * 64); (void)hipMalloc(&sink, blocks * 512 * 4);
    for (int it = 0; it < 2; it++) hipLaunchKernelGGL((k_pair<ROLES>), dim3(blocks), dim3(512), 0, 0, out, sink, reps, 0.999f, mode);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 8);
    (void)hipMemcpy(h.data(), out, blocks * 64, hipMemcpyDeviceToHost);
    std::vector<unsigned long long> lo, hi;
    for (int b = 0; b < blocks; b++) for (int w = 0; w < 8; w++) (w < 4 ? lo : hi).push_back(h[b * 8 + w]);
    std::sort(lo.begin(), lo.end()); std::sort(hi.begin(), hi.end());
    printf("two waves per SIMD, %s: waves 0-3 %.0f cycles, waves 4-7 %.0f cycles (median; per wave %d x 32 MFMA = %d x 64 v_pk_fma_f32 of work)\n", what,
           (double)lo[lo.size() / 2], (double)hi[hi.size() / 2], reps, reps);
    (void)hipFree(out); (void)hipFree(sink);
}

template <int NM, int NV>
void run(int wps) {
    const int blocks = 256 * 4 * wps, reps = 400;
    unsigned long long* out; float* sink;
    (void)hipMalloc(&out, blocks * 8); (void)hipMalloc(&sink, blocks * 64 * 4);
    for (int it = 0; it < 2; it++) hipLaunchKernelGGL((k_rate<NM, NV>), dim3(blocks), dim3(64), 0, 0, out, sink, reps, 0.999f);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    (void)hipMemcpy(h.data(), out, blocks * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double groups = 8.0 * reps;
    printf("waves/SIMD %d: %d mfma_4x4x1 + %d v_pk_fma_f32 per group: %.2f cycles per group per wave (median)\n", wps, NM, NV, (double)h[blocks / 2] / groups);
    (void)hipFree(out); (void)hipFree(sink);
}

int main() {
    // ---- layout: A value = 1 + lane, B value = 100 * (1 + lane): d[lane][r] tells which (a lane, b lane) pair met
    std::vector<float> a(64), b(64), d(256);
    for (int l = 0; l < 64; l++) { a[l] = 1.f + l; b[l] = 100.f * (1 + l) + 0.f; }
    float *da, *db, *dd;
    (void)hipMalloc(&da, 256); (void)hipMalloc(&db, 256); (void)hipMalloc(&dd, 1024);
    (void)hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice); (void)hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, da, db, dd);
    (void)hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++)
        for (int r = 0; r < 4; r++) {
            // expected: block = l / 4, j = l % 4, i = r:  D[block][i][j] = A[lane 4 block + i] * B[lane 4 block + j]
            const float want = a[4 * (l / 4) + r] * b[l];
            if (d[4 * l + r] != want) { if (bad < 8) printf("layout: lane %d reg %d = %g, expected %g\n", l, r, d[4 * l + r], want); bad++; }
        }
    printf("layout D[lane = 4 blk + j][reg i] = A[lane 4 blk + i] * B[lane 4 blk + j]: %s\n", bad ? "MISMATCH" : "confirmed");
    run_pair<0>(0, "all eight waves MFMA only");
    run_pair<0>(1, "all eight waves v_pk_fma_f32 only");
    run_pair<1>(0, "waves 0-3 MFMA, waves 4-7 v_pk_fma_f32");
    for (int w : {1, 2}) {
        run<1, 0>(w); run<2, 0>(w); run<4, 0>(w); run<0, 4>(w);
        run<4, 1>(w); run<4, 2>(w); run<4, 4>(w); run<4, 6>(w); run<4, 8>(w); run<2, 4>(w); run<1, 4>(w);
    }
    return 0;
}
