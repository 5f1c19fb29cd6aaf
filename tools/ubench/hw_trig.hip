// Accuracy of the hardware v_sin_f32 / v_cos_f32 (argument in revolutions) over the light-sampling range [0, 4).
// Build: hipcc --offload-arch=gfx950 -O3 hw_trig.hip -o hw_trig ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void k(const float* x, float* s, float* c, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { s[i] = __builtin_amdgcn_sinf(x[i]); c[i] = __builtin_amdgcn_cosf(x[i]); }
}

int main() {
    const int n = 1 << 24;
    std::vector<float> x(n), s(n), c(n);
    // the kernel's arguments are (float)u32 * 2^-30; take a dense, slightly irregular sample of them
    unsigned v = 12345u;
    for (int i = 0; i < n; i++) {
        v ^= v << 13; v ^= v >> 17; v ^= v << 5;
        x[i] = (i & 1) ? (float)v * (1.0f / 1073741824.0f) : (float)i * (4.0f / n);
    }
    float *dx, *ds, *dc;
    CHK(hipMalloc(&dx, n * 4)); CHK(hipMalloc(&ds, n * 4)); CHK(hipMalloc(&dc, n * 4));
    CHK(hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice));
    k<<<n / 256, 256>>>(dx, ds, dc, n);
    CHK(hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost));
    CHK(hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost));
    double max_abs_s = 0, max_abs_c = 0, max_ulp_s = 0, max_ulp_c = 0, sum_abs = 0;
    float arg_s = 0, arg_c = 0;
    for (int i = 0; i < n; i++) {
        const double t = 2.0 * M_PI * (double)x[i];
        const double rs = sin(t), rc = cos(t);
        const double es = fabs((double)s[i] - rs), ec = fabs((double)c[i] - rc);
        sum_abs += es;
        if (es > max_abs_s) { max_abs_s = es; arg_s = x[i]; }
        if (ec > max_abs_c) { max_abs_c = ec; arg_c = x[i]; }
        const double us = es / (double)(nextafterf(fabsf((float)rs), INFINITY) - fabsf((float)rs));
        const double uc = ec / (double)(nextafterf(fabsf((float)rc), INFINITY) - fabsf((float)rc));
        if (fabs(rs) > 1e-3 && us > max_ulp_s) max_ulp_s = us;
        if (fabs(rc) > 1e-3 && uc > max_ulp_c) max_ulp_c = uc;
    }
    printf("v_sin_f32: max abs err %.3e (at %.9g rev), mean abs %.3e, max ulp (|sin|>1e-3) %.1f\n", max_abs_s, arg_s, sum_abs / n, max_ulp_s);
    printf("v_cos_f32: max abs err %.3e (at %.9g rev), max ulp (|cos|>1e-3) %.1f\n", max_abs_c, arg_c, max_ulp_c);
    return 0;
}
