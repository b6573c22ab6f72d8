// Which XCD does workgroup b of a launch land on?  (hipcc --offload-arch=gfx950 -O2 xcc_placement.hip -o xcc_placement.bin)
// Launches the gradient step's grid shape (256 + 140 workgroups of 512 threads, 98 KB of LDS each) and records HW_REG_XCC_ID / HW_REG_HW_ID per block.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(512) void probe(unsigned *out, int spin) {
    extern __shared__ float S[];
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hwid; }
    // stay resident for a while, like the real workgroups (the first 256 fill the chip, the rest follow)
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin * (blockIdx.x < 128 ? 1 : 2)) __builtin_amdgcn_s_sleep(8);
    S[threadIdx.x] = (float)xcc;
}
int main() {
    const int grids[] = {256, 396, 140, 1024};
    hipFuncSetAttribute(reinterpret_cast<const void *>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    for (int g : grids) {
        unsigned *d; hipMalloc(&d, 8 * g);
        int bad_total = 0;
        for (int rep = 0; rep < 20; ++rep) {
            hipLaunchKernelGGL(probe, dim3(g), dim3(512), 98 * 1024, 0, d, 1000);      // 10 us (first half), 20 us
            std::vector<unsigned> h(2 * g);
            hipMemcpy(h.data(), d, 8 * g, hipMemcpyDeviceToHost);
            int bad = 0;
            for (int b = 0; b < g; ++b) bad += (int)((h[2 * b] & 15u) != (unsigned)(b % 8));
            bad_total += bad;
            if (rep == 0) {
                printf("grid %4d: xcc of blocks 0..23:", g);
                for (int b = 0; b < 24 && b < g; ++b) printf(" %u", h[2 * b] & 15u);
                printf("\n           xcc of blocks %d..%d:", g - 16, g - 1);
                for (int b = g - 16; b < g; ++b) printf(" %u", h[2 * b] & 15u);
                printf("\n");
            }
        }
        printf("grid %4d: blocks with xcc != block %% 8 over 20 launches: %d\n", g, bad_total);
        hipFree(d);
    }
    return 0;
}
