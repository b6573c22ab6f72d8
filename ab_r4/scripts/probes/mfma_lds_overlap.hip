// Probe (MI355X): cost of feeding v_mfma_f32_16x16x32_f16 its A operand from LDS (one or two ds_read_b128 per MFMA pair), two
// waves per SIMD, next to K plain VALU fillers per MFMA.  The A operand of MFMA i + 8 is the data read at MFMA i.
// build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// DS = ds_read_b128 per 8 MFMAs (0, 4 = one per two MFMAs, 8 = one per MFMA, 16 = two per MFMA)
template <int K, int DS>
__global__ __launch_bounds__(512) void kern(float *out, int iters) {
    extern __shared__ u32x4 lds[];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (u32x4){(unsigned)i, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    __syncthreads();
    f16x8 b;
    for (int i = 0; i < 8; ++i) b[i] = (_Float16)1.0f;
    f32x4 acc[8];
    float v[8];
    f16x8 a[16];
    for (int j = 0; j < 8; ++j) { acc[j] = (f32x4){0, 0, 0, 0}; v[j] = threadIdx.x * 0.001f + j; }
    for (int j = 0; j < 16; ++j) a[j] = __builtin_bit_cast(f16x8, lds[(threadIdx.x & 63) + 64 * j]);
    const float m = 0.999f;
    int base = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(DS == -8 ? a[8 * (it & 1) + j] : a[j], b, acc[j], 0, 0, 0);
            if (DS == -8) { if (j == 0) { for (int q = 0; q < 8; ++q) a[8 * ((it + 1) & 1) + q] = __builtin_bit_cast(f16x8, lds[base + 64 * (q + 8 * (it & 7))]); } }
            else if (DS == 16) { a[j] = __builtin_bit_cast(f16x8, lds[base + 64 * (j + 8 * (it & 7))]); a[j + 8] = __builtin_bit_cast(f16x8, lds[base + 64 * (j + 64 + 8 * (it & 7))]); }
            else if (DS == 8) a[j] = __builtin_bit_cast(f16x8, lds[base + 64 * (j + 8 * (it & 7))]);
            else if (DS == 4 && (j & 1) == 0) a[j] = __builtin_bit_cast(f16x8, lds[base + 64 * (j + 8 * (it & 7))]);
#pragma unroll
            for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[(j + k) & 7]) : "v"(m));
            __builtin_amdgcn_sched_barrier(0);
        }
        if (DS == 16) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j + 8], b, acc[j], 0, 0, 0);
        }
    }
    float s = 0;
    for (int j = 0; j < 8; ++j) s += acc[j][0] + v[j];
    if (s == 12345.f) out[0] = s;
}

// bursty variant: 8 ds_read_b128 back to back once per 8 MFMAs, consumed 8..15 MFMAs later (static double buffer)
template <int K>
__global__ __launch_bounds__(512) void kern_burst(float *out, int iters) {
    extern __shared__ u32x4 lds[];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (u32x4){(unsigned)i, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    __syncthreads();
    f16x8 b;
    for (int i = 0; i < 8; ++i) b[i] = (_Float16)1.0f;
    f32x4 acc[8];
    float v[8];
    f16x8 a0[8], a1[8];
    for (int j = 0; j < 8; ++j) { acc[j] = (f32x4){0, 0, 0, 0}; v[j] = threadIdx.x * 0.001f + j; }
    const int base = threadIdx.x & 63;
    for (int j = 0; j < 8; ++j) a0[j] = __builtin_bit_cast(f16x8, lds[base + 64 * j]);
    const float m = 0.999f;
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int q = 0; q < 8; ++q) a1[q] = __builtin_bit_cast(f16x8, lds[base + 64 * (q + 8 * (it & 7))]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0[j], b, acc[j], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[(j + k) & 7]) : "v"(m));
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) a0[q] = __builtin_bit_cast(f16x8, lds[base + 64 * (q + 8 * ((it + 1) & 7))]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[j], b, acc[j], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[(j + k) & 7]) : "v"(m));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int j = 0; j < 8; ++j) s += acc[j][0] + v[j];
    if (s == 12345.f) out[0] = s;
}
template <int K>
static float run_burst(int iters, float *dout) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern_burst<K>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        kern_burst<K><<<256, 512, 131072>>>(dout, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    return ms;
}

template <int K, int DS>
static float run(int iters, float *dout) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern<K, DS>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        kern<K, DS><<<256, 512, 131072>>>(dout, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    return ms;
}

int main() {
    float *dout; (void)hipMalloc(&dout, 64);
    const int iters = 20000;
    const double per = 1e6 / (iters * 8.0) / 2;   // ns per MFMA per SIMD (two waves)
    printf("two waves per SIMD; ns per MFMA slot (7.0 = matrix pipe bound); columns: K = VALU fillers per MFMA\n");
    printf("no LDS reads            K=0 %.2f  K=1 %.2f  K=2 %.2f  K=3 %.2f\n", run<0, 0>(iters, dout) * per, run<1, 0>(iters, dout) * per, run<2, 0>(iters, dout) * per, run<3, 0>(iters, dout) * per);
    printf("1 ds_read_b128 / 2 MFMA K=0 %.2f  K=1 %.2f  K=2 %.2f  K=3 %.2f\n", run<0, 4>(iters, dout) * per, run<1, 4>(iters, dout) * per, run<2, 4>(iters, dout) * per, run<3, 4>(iters, dout) * per);
    printf("1 ds_read_b128 / MFMA   K=0 %.2f  K=1 %.2f  K=2 %.2f  K=3 %.2f\n", run<0, 8>(iters, dout) * per, run<1, 8>(iters, dout) * per, run<2, 8>(iters, dout) * per, run<3, 8>(iters, dout) * per);
    printf("8 ds_read_b128 in one burst / 8 MFMA, used 8..15 MFMAs later K=0 %.2f  K=2 %.2f  K=3 %.2f\n", run_burst<0>(iters, dout) * per, run_burst<2>(iters, dout) * per, run_burst<3>(iters, dout) * per);
    printf("2 ds_read_b128 / 2 MFMA (16 MFMAs per iteration) K=0 %.2f  K=2 %.2f\n", run<0, 16>(iters, dout) * per / 2, run<2, 16>(iters, dout) * per / 2);
    return 0;
}
