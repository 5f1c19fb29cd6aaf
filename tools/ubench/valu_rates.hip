// Micro-benchmark: issue cost of the VALU instructions the trace kernel is made of (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float float2_ __attribute__((ext_vector_type(2)));
#define ITER 4096
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, float seed) {
    float a[8];
    float2_ p[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = seed + i + threadIdx.x; p[i] = (float2_){a[i], a[i] + 0.5f}; }
    double dd = seed;
    const float m = 1.0000001f, c = 1e-7f;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (MODE == 0) a[i] = __builtin_fmaf(a[i], m, c);                         // v_fma_f32
            if (MODE == 1) p[i] = __builtin_elementwise_fma(p[i], (float2_){m, m}, (float2_){c, c}); // v_pk_fma_f32
            if (MODE == 2) a[i] = __builtin_amdgcn_sqrtf(a[i]) + 1.0f;               // v_sqrt_f32 + v_add
            if (MODE == 3) a[i] = __builtin_amdgcn_rcpf(a[i]) + 1.0f;                // v_rcp_f32 + v_add
            if (MODE == 4) a[i] = a[i] * m;                                          // v_mul_f32
            if (MODE == 5) a[i] = (a[i] > 3.0f) ? a[i] - 1.0f : a[i] + c;            // cmp + cndmask-ish
            if (MODE == 7) a[i] = __builtin_amdgcn_sinf(a[i]) + 1.0f;                // v_sin_f32 + v_add
            if (MODE == 8) a[i] = __builtin_amdgcn_rsqf(a[i]) + 1.0f;                // v_rsq_f32 + v_add
        }
        if (MODE == 6) { dd = dd * 1.0000001 + 1e-9; }                               // v_fma_f64 chain (1 per iter)
    }
    float s = (float)dd;
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, double ops_per_iter, int blocks_per_cu) {
    int grid = 256 * blocks_per_cu;
    float* d; CHK(hipMalloc(&d, grid * 256 * 4));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, 1.0f);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, 1.0f);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    // wave-instructions per SIMD: each block = 4 waves = one per SIMD
    double winst_per_simd = (double)blocks_per_cu * ITER * ops_per_iter;
    double ns_per_inst = ms * 1e6 / winst_per_simd;
    printf("%-28s waves/SIMD %d: %.3f ms, %.3f ns per wave-instruction per SIMD (= %.2f cycles @2.4GHz)\n", name, blocks_per_cu, ms,
           ns_per_inst, ns_per_inst * 2.4);
    CHK(hipFree(d));
}

int main() {
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma_f32 x8", 8, w);
        run<1>("v_pk_fma_f32 x8", 8, w);
        run<2>("v_sqrt_f32+v_add x8", 16, w);
        run<3>("v_rcp_f32+v_add x8", 16, w);
        run<4>("v_mul_f32 x8", 8, w);
        run<5>("cmp+sub/add+cndmask x8", 32, w);
        run<6>("v_fma_f64 x1 (dependent)", 1, w);
        run<7>("v_sin_f32+v_add x8", 16, w);
        run<8>("v_rsq_f32+v_add x8", 16, w);
    }
    return 0;
}
