// Issue-rate micro-benchmark for the f32 MFMA shapes on gfx950: cycles per instruction for chains that alternate
// between NACC independent accumulators, one wave per SIMD and two waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 scripts/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void k16(float *out, long long *cyc, int iters) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NACC>
__global__ void k32(float *out, long long *cyc, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <typename K>
void run(const char *name, K kern, int nacc, int threads, int flop_per_instr) {
    float *out; long long *cyc;
    hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 8);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<256, threads>>>(out, cyc, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kern<<<256, threads>>>(out, cyc, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double instr_per_wave = (double)iters * 16 * nacc;
    const double waves_per_simd = threads / 64 / 4.0;
    const double tflops = 256.0 * (threads / 64) * instr_per_wave * flop_per_instr / (ms * 1e-3) / 1e12;
    printf("%-22s acc=%d waves/SIMD=%.0f : %.1f s_memtime ticks per MFMA per wave (100 MHz ticks x24 ~ cycles), %.3f ms, %.1f TFLOP/s\n",
           name, nacc, waves_per_simd, (double)c / instr_per_wave, ms, tflops);
    hipFree(out); hipFree(cyc);
}
int main() {
    run("16x16x4 f32", k16<1>, 1, 256, 2048); run("16x16x4 f32", k16<2>, 2, 256, 2048); run("16x16x4 f32", k16<4>, 4, 256, 2048);
    run("16x16x4 f32", k16<2>, 2, 512, 2048); run("16x16x4 f32", k16<4>, 4, 512, 2048);
    run("32x32x2 f32", k32<1>, 1, 256, 4096); run("32x32x2 f32", k32<2>, 2, 256, 4096);
    run("32x32x2 f32", k32<2>, 2, 512, 4096);
    return 0;
}
