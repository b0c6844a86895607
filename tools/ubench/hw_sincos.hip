// Accuracy of v_sin_f32 / v_cos_f32 (input in revolutions) against double precision, with a two-constant reduction of the angle
// to [-pi, pi] in front - would the hardware pair do for the kernels' (cos, sin) of half angles?  (cpf_device.h fast_sincos: 6e-8.)
//   build: hipcc --offload-arch=gfx950 -O3 -o /tmp/hw_sincos tools/ubench/hw_sincos.hip      run: /tmp/hw_sincos
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const float* x, float* s, float* c, int n, int mode) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r = x[i], rev;
    if (mode == 0) rev = r * 0.15915494309189535f;
    else {
        const float kk = rintf(r * 0.15915494309189535f);
        r = fmaf(-kk, 6.28318548202514648f, r);
        r = fmaf(-kk, -1.74845553146951715e-7f, r);
        rev = r * 0.15915494309189535f;
    }
    s[i] = __builtin_amdgcn_sinf(rev);
    c[i] = __builtin_amdgcn_cosf(rev);
}
int main() {
    const int n = 1 << 22;
    std::vector<float> x(n), s(n), c(n);
    float *dx, *ds, *dc;
    hipMalloc(&dx, n * 4); hipMalloc(&ds, n * 4); hipMalloc(&dc, n * 4);
    for (float range : {3.1415926f, 12.f, 40.f}) {
        for (int i = 0; i < n; i++) x[i] = range * (2.0f * i / (n - 1) - 1.0f);
        hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
        for (int mode = 0; mode < 2; mode++) {
            hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, ds, dc, n, mode);
            hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost); hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost);
            double es = 0, ec = 0, rs = 0, rc = 0, en = 0;
            for (int i = 0; i < n; i++) {
                const double a = sin((double)x[i]), b = cos((double)x[i]);
                es = fmax(es, fabs(s[i] - a)); ec = fmax(ec, fabs(c[i] - b));
                rs += (s[i] - a) * (s[i] - a); rc += (c[i] - b) * (c[i] - b);
                en = fmax(en, fabs((double)s[i] * s[i] + (double)c[i] * c[i] - 1.0));
            }
            printf("|x| <= %5.2f  %-28s max |sin err| %.2e  max |cos err| %.2e  rms %.2e / %.2e  max |s^2 + c^2 - 1| %.2e\n", range,
                   mode ? "reduced to [-pi, pi] first" : "x / 2 pi straight", es, ec, sqrt(rs / n), sqrt(rc / n), en);
        }
    }
    return 0;
}
