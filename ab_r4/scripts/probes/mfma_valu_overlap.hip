// Probe (MI355X): how much VALU work hides under v_mfma_f32_16x16x32_f16 (4 passes, 16 cycles)?
//  (a) one wave per SIMD, K plain VALU instructions between consecutive independent MFMAs, K = 0..6
//  (b) two waves per SIMD: one MFMA-only, one VALU-only (alone and together)
//  (c) two waves per SIMD, both running the interleaved stream of (a)
// build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int K>
__device__ __forceinline__ void body(f32x4 (&acc)[8], f16x8 a, f16x8 b, float (&v)[8], float m) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[j], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[(j + k) & 7]) : "v"(m));
    }
}

// MODE 0: every wave runs MFMA + K fillers; MODE 1: waves 0-3 MFMA only, waves 4-7 (if present) K*8 VALU only per iteration
// (d) dependent-accumulator distance: the 8 MFMAs of an iteration cycle over NACC accumulators
template <int K, int NACC>
__global__ __launch_bounds__(512) void kern_dist(float *out, int iters) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 1e-3f); b[i] = (_Float16)1.0f; }
    f32x4 acc[8];
    float v[8];
    for (int j = 0; j < 8; ++j) { acc[j] = (f32x4){0, 0, 0, 0}; v[j] = threadIdx.x * 0.001f + j; }
    const float m = 0.999f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc[j % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[j % NACC], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[(j + k) & 7]) : "v"(m));
        }
    }
    float s = 0;
    for (int j = 0; j < 8; ++j) s += acc[j][0] + v[j];
    if (s == 12345.f) out[0] = s;
}
template <int K, int NACC>
static float run_dist(int iters, float *dout) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        kern_dist<K, NACC><<<256, 512>>>(dout, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    return ms;
}

// (e) the same matrix work as 32x32x16 MFMAs (8 passes, 2 x the FLOPs of a 16x16x32): one MFMA + K2 fillers
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int K2>
__global__ __launch_bounds__(512) void kern32(float *out, int iters) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 1e-3f); b[i] = (_Float16)1.0f; }
    f32x16 acc[4];
    float v[8];
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    for (int j = 0; j < 8; ++j) v[j] = threadIdx.x * 0.001f + j;
    const float m = 0.999f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < K2; ++k) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[(j + k) & 7]) : "v"(m));
        }
    }
    float s = 0;
    for (int j = 0; j < 4; ++j) s += acc[j][0];
    for (int j = 0; j < 8; ++j) s += v[j];
    if (s == 12345.f) out[0] = s;
}
template <int K2>
static float run32(int iters, float *dout) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        kern32<K2><<<256, 512>>>(dout, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    return ms;
}

template <int K, int MODE>
__global__ __launch_bounds__(512) void kern(float *out, int iters, int valu_only_first) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 1e-3f); b[i] = (_Float16)1.0f; }
    f32x4 acc[8];
    float v[8];
    for (int j = 0; j < 8; ++j) { acc[j] = (f32x4){0, 0, 0, 0}; v[j] = threadIdx.x * 0.001f + j; }
    const float m = 0.999f;
    const int wave = threadIdx.x >> 6;
    if (MODE == 0) {
        for (int it = 0; it < iters; ++it) body<K>(acc, a, b, v, m);
    } else {
        const bool valu_wave = valu_only_first ? (wave < 4 && blockDim.x > 256) || (blockDim.x == 256 && valu_only_first == 2) : wave >= 4;
        if (!valu_wave) { for (int it = 0; it < iters; ++it) body<0>(acc, a, b, v, m); }
        else {
            for (int it = 0; it < iters; ++it)
#pragma unroll
                for (int k = 0; k < 8 * K; ++k) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[k & 7]) : "v"(m));
        }
    }
    float s = 0;
    for (int j = 0; j < 8; ++j) s += acc[j][0] + v[j];
    if (s == 12345.f) out[0] = s;
}

template <int K, int MODE>
static float run(int threads, int iters, int flag, float *dout) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        kern<K, MODE><<<256, threads>>>(dout, iters, flag);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    return ms;
}

int main() {
    float *dout; (void)hipMalloc(&dout, 64);
    const int iters = 20000;
    const double per = 1e6 / (iters * 8.0);   // ms -> ns per MFMA slot
    printf("(a) ONE wave per SIMD (256 threads/CU): ns per [MFMA + K v_fma_f32]   (16 cycles at 2.4 GHz = 6.7 ns)\n");
    printf("   K=0 %.2f  K=1 %.2f  K=2 %.2f  K=3 %.2f  K=4 %.2f  K=5 %.2f  K=6 %.2f  K=8 %.2f\n", run<0, 0>(256, iters, 0, dout) * per, run<1, 0>(256, iters, 0, dout) * per,
           run<2, 0>(256, iters, 0, dout) * per, run<3, 0>(256, iters, 0, dout) * per, run<4, 0>(256, iters, 0, dout) * per, run<5, 0>(256, iters, 0, dout) * per,
           run<6, 0>(256, iters, 0, dout) * per, run<8, 0>(256, iters, 0, dout) * per);
    printf("(c) TWO waves per SIMD (512 threads/CU), both interleaved: ns per [MFMA + K v_fma_f32] per SIMD (two waves' slots)\n");
    printf("   K=0 %.2f  K=1 %.2f  K=2 %.2f  K=3 %.2f  K=4 %.2f  K=6 %.2f  K=8 %.2f\n", run<0, 0>(512, iters, 0, dout) * per / 2, run<1, 0>(512, iters, 0, dout) * per / 2,
           run<2, 0>(512, iters, 0, dout) * per / 2, run<3, 0>(512, iters, 0, dout) * per / 2, run<4, 0>(512, iters, 0, dout) * per / 2, run<6, 0>(512, iters, 0, dout) * per / 2,
           run<8, 0>(512, iters, 0, dout) * per / 2);
    printf("(b) split roles, per iteration of 8 MFMAs (wave A) and 8K v_fma (wave B): ms total\n");
    printf("   MFMA-only wave alone (256 thr): %.3f ms\n", run<0, 0>(256, iters, 0, dout));
    printf("   VALU-only wave alone, K=2: %.3f  K=3: %.3f  K=4: %.3f ms\n", run<2, 1>(256, iters, 2, dout), run<3, 1>(256, iters, 2, dout), run<4, 1>(256, iters, 2, dout));
    printf("   together (MFMA waves 0-3 older, VALU waves 4-7), K=2: %.3f  K=3: %.3f  K=4: %.3f ms\n", run<2, 1>(512, iters, 0, dout), run<3, 1>(512, iters, 0, dout), run<4, 1>(512, iters, 0, dout));
    printf("   together (VALU waves 0-3 older, MFMA waves 4-7), K=2: %.3f  K=3: %.3f  K=4: %.3f ms\n", run<2, 1>(512, iters, 1, dout), run<3, 1>(512, iters, 1, dout), run<4, 1>(512, iters, 1, dout));
    printf("(d) TWO waves per SIMD, dependent-accumulator distance (ns per MFMA slot per SIMD):\n");
    printf("   K=0: dist 8 %.2f  4 %.2f  2 %.2f  1 %.2f\n", run_dist<0, 8>(iters, dout) * per / 2, run_dist<0, 4>(iters, dout) * per / 2, run_dist<0, 2>(iters, dout) * per / 2, run_dist<0, 1>(iters, dout) * per / 2);
    printf("   K=2: dist 8 %.2f  4 %.2f  2 %.2f  1 %.2f\n", run_dist<2, 8>(iters, dout) * per / 2, run_dist<2, 4>(iters, dout) * per / 2, run_dist<2, 2>(iters, dout) * per / 2, run_dist<2, 1>(iters, dout) * per / 2);
    printf("   K=3: dist 8 %.2f  4 %.2f  2 %.2f  1 %.2f\n", run_dist<3, 8>(iters, dout) * per / 2, run_dist<3, 4>(iters, dout) * per / 2, run_dist<3, 2>(iters, dout) * per / 2, run_dist<3, 1>(iters, dout) * per / 2);
    {
        const double per32 = 1e6 / (iters * 4.0) / 2;     // ns per 32x32x16 MFMA per SIMD (two waves) = two 16x16x32 slots of matrix work
        printf("(e) TWO waves per SIMD, v_mfma_f32_32x32x16_f16 + K2 fillers: ns per MFMA (= 2 slots of (c)): K2=0 %.2f  2 %.2f  4 %.2f  6 %.2f  8 %.2f  10 %.2f\n",
               run32<0>(iters, dout) * per32, run32<2>(iters, dout) * per32, run32<4>(iters, dout) * per32, run32<6>(iters, dout) * per32, run32<8>(iters, dout) * per32,
               run32<10>(iters, dout) * per32);
    }
    return 0;
}
