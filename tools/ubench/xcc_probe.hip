// Which XCD / CU / SIMD / wave slot does a workgroup run on?  (s_getreg XCC_ID, HW_ID on gfx950)
// Checks what the tree-parallel tail's slot allocator relies on: XCC_ID in 0..7, and reports whether blocks b and b + 8 share it
// and whether (XCC_ID, HW_ID[15:0]) is unique among the waves resident at the same time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <set>
#include <vector>
__global__ void probe(unsigned* out, unsigned long long* t) {
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));      // HW_REG_XCC_ID[3:0]
    const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));       // HW_REG_HW_ID[31:0]
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_sleep(100);
    unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    while (t1 - t0 < 2000) t1 = __builtin_amdgcn_s_memrealtime();                     // ~20 us: the grid's waves overlap in time
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; t[2 * blockIdx.x] = t0; t[2 * blockIdx.x + 1] = t1; }
}
int main() {
    const int n = 4096;
    unsigned* d; unsigned long long* dt;
    hipMalloc(&d, n * 8); hipMalloc(&dt, n * 16);
    hipLaunchKernelGGL(probe, dim3(n), dim3(64), 0, 0, d, dt);
    std::vector<unsigned> h(2 * n); std::vector<unsigned long long> ht(2 * n);
    hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost); hipMemcpy(ht.data(), dt, n * 16, hipMemcpyDeviceToHost);
    std::set<unsigned> xccs; int mism = 0;
    for (int b = 0; b < n; b++) { xccs.insert(h[2 * b]); if (b >= 8 && h[2 * b] != h[2 * (b - 8)]) mism++; }
    printf("distinct XCC_ID values: %zu (", xccs.size()); for (unsigned x : xccs) printf("%u ", x); printf("); blocks b, b-8 on different XCCs: %d of %d\n", mism, n - 8);
    // uniqueness of (xcc, hw[15:0]) among waves that overlap in time
    int clash = 0;
    for (int a = 0; a < n; a++) for (int b = a + 1; b < n; b++)
        if (h[2 * a] == h[2 * b] && (h[2 * a + 1] & 0xFFFF) == (h[2 * b + 1] & 0xFFFF) && ht[2 * a] < ht[2 * b + 1] && ht[2 * b] < ht[2 * a + 1]) clash++;
    printf("concurrent waves with equal (XCC_ID, HW_ID[15:0]): %d\n", clash);
    for (int b = 0; b < 12; b++) printf("block %d: xcc %u hw_id 0x%08x (wave %u simd %u cu %u sh %u se %u)\n", b, h[2 * b], h[2 * b + 1], h[2 * b + 1] & 15, (h[2 * b + 1] >> 4) & 3,
                                        (h[2 * b + 1] >> 8) & 15, (h[2 * b + 1] >> 12) & 1, (h[2 * b + 1] >> 13) & 7);
    return 0;
}
