// Cycles per instruction pattern for ONE wave's dependent stream on a gfx950 SIMD - the quantity that bounds a step of
// cpf_zyz3.h (a step is one dependent chain; see DESIGN.md "what bounds a step").  Each pattern is unrolled REP times inside a
// loop of ITER trips, timed with s_memtime on wave 0 of block 0 while `waves per SIMD` identical waves run beside it.
//   build: hipcc --offload-arch=gfx950 -O3 -o /tmp/issue_cost tools/ubench/issue_cost.hip      run: /tmp/issue_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP 16
#define ITER 256

template <int P>
__device__ __forceinline__ void pattern(f2& x, f2& y, f2& z, f2& w, f2 d, float t) {
    if constexpr (P == 0) {            // dependent v_fma_f32
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x.x) : "v"(d.x), "v"(d.y));
    } else if constexpr (P == 1) {     // dependent v_pk_fma_f32
        asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(d));
    } else if constexpr (P == 2) {     // four independent v_fma_f32
        asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5"
                     : "+v"(x.x), "+v"(x.y), "+v"(y.x), "+v"(y.y) : "v"(d.x), "v"(d.y));
    } else if constexpr (P == 3) {     // four independent v_pk_fma_f32
        asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n\tv_pk_fma_f32 %1, %1, %4, %4\n\tv_pk_fma_f32 %2, %2, %4, %4\n\tv_pk_fma_f32 %3, %3, %4, %4"
                     : "+v"(x), "+v"(y), "+v"(z), "+v"(w) : "v"(d));
    } else if constexpr (P == 4) {     // the kernel's block: packed complex multiply, two DPP stages
        f2 o;
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]\n\t"
                     "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=&v"(o) : "v"(d), "v"(x));
        x = o;
        asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "v_fmac_f32_dpp %1, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x.x), "+v"(x.y) : "v"(t));
        asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %2 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "v_fmac_f32_dpp %1, %1, %2 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x.x), "+v"(x.y) : "v"(t));
    } else if constexpr (P == 5) {     // the same block with a scalar complex multiply (two levels of two independent instructions)
        float a, b;
        asm volatile("v_mul_f32 %0, %2, %4\n\tv_mul_f32 %1, %2, %5\n\t"
                     "v_fma_f32 %0, -%3, %5, %0\n\tv_fma_f32 %1, %3, %4, %1" : "=&v"(a), "=&v"(b) : "v"(d.x), "v"(d.y), "v"(x.x), "v"(x.y));
        x.x = a; x.y = b;
        asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "v_fmac_f32_dpp %1, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x.x), "+v"(x.y) : "v"(t));
        asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %2 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "v_fmac_f32_dpp %1, %1, %2 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x.x), "+v"(x.y) : "v"(t));
    } else if constexpr (P == 6) {     // one DPP stage alone (s_nop 1 + two v_fmac_f32_dpp)
        asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "v_fmac_f32_dpp %1, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x.x), "+v"(x.y) : "v"(t));
    } else if constexpr (P == 7) {     // the packed complex multiply alone
        f2 o;
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]\n\t"
                     "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=&v"(o) : "v"(d), "v"(x));
        x = o;
    } else if constexpr (P == 8) {     // the scalar complex multiply alone
        float a, b;
        asm volatile("v_mul_f32 %0, %2, %4\n\tv_mul_f32 %1, %2, %5\n\t"
                     "v_fma_f32 %0, -%3, %5, %0\n\tv_fma_f32 %1, %3, %4, %1" : "=&v"(a), "=&v"(b) : "v"(d.x), "v"(d.y), "v"(x.x), "v"(x.y));
        x.x = a; x.y = b;
    } else if constexpr (P == 9) {     // the block with its LDS store (as in the forward half)
        pattern<4>(x, y, z, w, d, t);
    } else if constexpr (P == 10) {    // dependent transcendental
        asm volatile("v_rcp_f32 %0, %0" : "+v"(x.x));
    } else if constexpr (P == 11) {    // v_mov_b32_dpp + v_fma_f32 (what the compiler emits for x += t * dpp(x)), both halves
        float a, b;
        asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "v_mov_b32_dpp %1, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                     "v_fmac_f32 %2, %0, %4\n\tv_fmac_f32 %3, %1, %4" : "=&v"(a), "=&v"(b), "+v"(x.x), "+v"(x.y) : "v"(t));
    }
}

template <int P>
__global__ __launch_bounds__(64) void k(float* out, unsigned long long* cyc, float t) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x;
    f2 x = {1.f + lane * 1e-3f, 0.5f}, y = x, z = x, w = x;
    const f2 d = {0.999f, 0.01f};
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int i = 0; i < ITER; i++) {
#pragma unroll
        for (int r = 0; r < REP; r++) {
            pattern<P>(x, y, z, w, d, t);
            if constexpr (P == 9) reinterpret_cast<f2*>(lds)[64 * r + lane] = x;
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    out[blockIdx.x * 64 + lane] = x.x + x.y + y.x + y.y + z.x + w.x;
    if (blockIdx.x == 0 && lane == 0) *cyc = t1 - t0;
}

template <int P>
void run(const char* name, int ninstr, float* out, unsigned long long* cyc) {
    for (int per_simd : {1, 2}) {
        const int grid = 256 * 4 * per_simd;
        unsigned long long best = ~0ull;
        for (int rep = 0; rep < 5; rep++) {
            hipLaunchKernelGGL(k<P>, dim3(grid), dim3(64), 16384, 0, out, cyc, 1e-3f);
            hipDeviceSynchronize();
            unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            if (c < best) best = c;
        }
        const double per = (double)best / (ITER * REP);
        printf("%-64s waves/SIMD %d  %7.2f cycles per pattern  %6.2f per vector instruction\n", name, per_simd, per, per / ninstr);
    }
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 2048 * 64 * 4); hipMalloc(&cyc, 8);
    run<0>("v_fma_f32, dependent", 1, out, cyc);
    run<1>("v_pk_fma_f32, dependent", 1, out, cyc);
    run<2>("4 x v_fma_f32, independent", 4, out, cyc);
    run<3>("4 x v_pk_fma_f32, independent", 4, out, cyc);
    run<10>("v_rcp_f32, dependent", 1, out, cyc);
    run<7>("complex multiply, packed (pk_mul, pk_fma)", 2, out, cyc);
    run<8>("complex multiply, scalar (2 mul, 2 fma)", 4, out, cyc);
    run<6>("DPP stage (s_nop 1, 2 x v_fmac_f32_dpp)", 2, out, cyc);
    run<11>("DPP stage as compiled (s_nop 1, 2 x v_mov_dpp, 2 x v_fmac)", 4, out, cyc);
    run<4>("block: packed multiply + 2 stages", 6, out, cyc);
    run<5>("block: scalar multiply + 2 stages", 8, out, cyc);
    run<9>("block: packed multiply + 2 stages + ds_write_b64", 6, out, cyc);
    return 0;
}
