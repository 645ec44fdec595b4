// Probe: which XCD / SE / CU does workgroup b of a grid of single-wave persistent workers land on?  (tools only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ __launch_bounds__(64, 8) void k_probe(unsigned *out, int spin) {
    __shared__ float lds[1168];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    float a = threadIdx.x;
    for (int i = 0; i < spin; ++i) { a = a * 1.0001f + 0.5f; lds[(threadIdx.x + i) % 1168] = a; }
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
    if (a == 123.f) out[0] = lds[5];
}
int main() {
    const int n = 8192;
    unsigned *d; hipMalloc(&d, n * 8);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_probe, dim3(n), dim3(64), 0, 0, d, 20000);
        hipDeviceSynchronize();
    }
    std::vector<unsigned> h(2 * n); hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, int> percu; int xcd_ok = 0;
    for (int b = 0; b < n; ++b) {
        unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
        unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7, simd = (hw >> 4) & 3, wave = hw & 0xf;
        if (xcc == (unsigned)(b % 8)) ++xcd_ok;
        percu[(xcc << 8) | (se << 5) | (sh << 4) | cu]++;
        if (b < 96 || (b % 8 == 0 && b < 1200)) printf("b %5d xcc %u se %u sh %u cu %2u simd %u wave %2u raw %08x\n", b, xcc, se, sh, cu, simd, wave, hw);
    }
    printf("blocks on XCD b%%8: %d of %d; distinct CUs %zu\n", xcd_ok, n, percu.size());
    std::map<int, int> hist; for (auto &kv : percu) hist[kv.second]++;
    for (auto &kv : hist) printf("  CUs with %d workers: %d\n", kv.first, kv.second);
    return 0;
}
