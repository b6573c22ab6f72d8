// Probe (MI355X): is a 2-piece f16 split (hi = RNE(x), lo = RNE(x - hi)) with three v_mfma_f32_16x16x32_f16 products
// (hi.hi + hi.lo + lo.hi) as accurate as the exact-f32 v_mfma_f32_16x16x4_f32 for the IQN layer shapes?  Also: are f16
// subnormal MFMA inputs kept, and what do the two instruction streams sustain?   build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int K = 224, M = 16, N = 16;

__device__ __forceinline__ f16x2 cvt2(float x, float y) {
    f32x2 v = {x, y};
    return __builtin_convertvector(v, f16x2);     // RNE; v_cvt_pk_f16_f32 on gfx950
}

// C = A[16][K] * B[K][16], A and B row-major f32 in global memory, one wave
__global__ void gemm_f32(const float *A, const float *B, float *C) {
    const int l = threadIdx.x, g = l >> 4, c = l & 15;
    f32x4 acc = {0, 0, 0, 0};
    for (int k = 0; k < K; k += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[c * K + k + g], B[(k + g) * N + c], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) C[(4 * g + r) * N + c] = acc[r];
}

template <bool RTZ>
__global__ void gemm_split(const float *A, const float *B, float *C, float sa, float sb) {
    const int l = threadIdx.x, g = l >> 4, c = l & 15;
    f32x4 acc = {0, 0, 0, 0};
    for (int k = 0; k < K; k += 32) {
        f16x8 ah, al, bh, bl;
        for (int i = 0; i < 8; i += 2) {
            float a0 = A[c * K + k + 8 * g + i] * sa, a1 = A[c * K + k + 8 * g + i + 1] * sa;
            float b0 = B[(k + 8 * g + i) * N + c] * sb, b1 = B[(k + 8 * g + i + 1) * N + c] * sb;
            f16x2 h, q;
            if (RTZ) h = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(a0, a1)); else h = cvt2(a0, a1);
            ah[i] = h[0]; ah[i + 1] = h[1];
            float r0 = fmaf((float)h[0], -1.0f, a0), r1 = fmaf((float)h[1], -1.0f, a1);
            if (RTZ) q = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(r0, r1)); else q = cvt2(r0, r1);
            al[i] = q[0]; al[i + 1] = q[1];
            if (RTZ) h = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(b0, b1)); else h = cvt2(b0, b1);
            bh[i] = h[0]; bh[i + 1] = h[1];
            r0 = fmaf((float)h[0], -1.0f, b0); r1 = fmaf((float)h[1], -1.0f, b1);
            if (RTZ) q = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(r0, r1)); else q = cvt2(r0, r1);
            bl[i] = q[0]; bl[i + 1] = q[1];
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
    }
    const float inv = 1.0f / (sa * sb);
    for (int r = 0; r < 4; ++r) C[(4 * g + r) * N + c] = acc[r] * inv;
}

// subnormal probe: A = 2^-20 (an f16 subnormal), B = 1 -> sum over k of 32 terms = 2^-15 if inputs are kept, 0 if flushed
__global__ void denorm_probe(float *out) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)9.5367431640625e-07f; b[i] = (_Float16)1.0f; }
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = acc[0]; out[1] = (float)a[0]; }
    // conversion of a value that lands in the subnormal f16 range
    f16x2 h = cvt2(3.0e-6f, -2.0e-7f);
    if (threadIdx.x == 0) { out[2] = (float)h[0]; out[3] = (float)h[1]; }
}

// issue-rate probes: `iters` x 8 independent accumulators per wave, 8 waves per block, one block per CU
__global__ __launch_bounds__(512) void rate_f16(float *out, int iters) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 1e-3f); b[i] = (_Float16)1.0f; }
    f32x4 acc[8];
    for (int j = 0; j < 8; ++j) acc[j] = (f32x4){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[j], 0, 0, 0);
    float s = 0;
    for (int j = 0; j < 8; ++j) s += acc[j][0];
    if (s == 12345.f) out[0] = s;
}
__global__ __launch_bounds__(512) void rate_f32(float *out, int iters) {
    float a = threadIdx.x * 1e-3f, b = 1.0f;
    f32x4 acc[8];
    for (int j = 0; j < 8; ++j) acc[j] = (f32x4){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
    float s = 0;
    for (int j = 0; j < 8; ++j) s += acc[j][0];
    if (s == 12345.f) out[0] = s;
}

static double urand() { return (double)rand() / RAND_MAX; }
static double nrand() { return std::sqrt(-2.0 * std::log(urand() + 1e-300)) * std::cos(6.283185307179586 * urand()); }

static void report(const char *name, const std::vector<float> &C, const std::vector<double> &R) {
    double maxref = 0, maxerr = 0, se = 0, sr = 0;
    for (int i = 0; i < M * N; ++i) { maxref = fmax(maxref, fabs(R[i])); maxerr = fmax(maxerr, fabs(C[i] - R[i])); se += (C[i] - R[i]) * (C[i] - R[i]); sr += R[i] * R[i]; }
    printf("  %-34s max|err| / max|C| = %.3e    rms err / rms C = %.3e\n", name, maxerr / maxref, std::sqrt(se / sr));
}

int main() {
    srand(7);
    float *dA, *dB, *dC, *dout;
    hipMalloc(&dA, M * K * 4); hipMalloc(&dB, K * N * 4); hipMalloc(&dC, M * N * 4); hipMalloc(&dout, 64);
    const char *cases[] = {"weights N(0,0.1), activations relu(N(0,1))", "weights N(0,0.1), activations 1e3*relu(N(0,1))", "wide range: |a| log-uniform 1e-6..1"};
    for (int cs = 0; cs < 3; ++cs) {
        std::vector<float> A(M * K), B(K * N), C(M * N);
        std::vector<double> R(M * N);
        float amax = 0, bmax = 0;
        for (auto &v : A) { v = cs == 2 ? (float)(std::pow(10.0, -6 * urand()) * (urand() < 0.5 ? -1 : 1)) : (float)(0.1 * nrand()); amax = fmaxf(amax, fabsf(v)); }
        for (auto &v : B) { double z = nrand(); v = (float)((z > 0 ? z : 0) * (cs == 1 ? 1e3 : 1.0)); bmax = fmaxf(bmax, fabsf(v)); }
        for (int i = 0; i < M; ++i) for (int j = 0; j < N; ++j) { double s = 0; for (int k = 0; k < K; ++k) s += (double)A[i * K + k] * (double)B[k * N + j]; R[i * N + j] = s; }
        hipMemcpy(dA, A.data(), M * K * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), K * N * 4, hipMemcpyHostToDevice);
        // power-of-two scales that put the largest element just below 2^15, and a deliberately conservative one (2^10 lower)
        const float sa = std::exp2(std::floor(std::log2(32768.0 / amax))), sb = std::exp2(std::floor(std::log2(32768.0 / bmax)));
        printf("case %d: %s  (K = %d; scales 2^%d, 2^%d)\n", cs, cases[cs], K, (int)std::log2(sa), (int)std::log2(sb));
        // float32 CPU dot product for scale
        for (int i = 0; i < M; ++i) for (int j = 0; j < N; ++j) { float s = 0; for (int k = 0; k < K; ++k) s = fmaf(A[i * K + k], B[k * N + j], s); C[i * N + j] = s; }
        report("CPU float32 fmaf chain", C, R);
        gemm_f32<<<1, 64>>>(dA, dB, dC); hipMemcpy(C.data(), dC, M * N * 4, hipMemcpyDeviceToHost); report("v_mfma_f32_16x16x4_f32", C, R);
        gemm_split<false><<<1, 64>>>(dA, dB, dC, sa, sb); hipMemcpy(C.data(), dC, M * N * 4, hipMemcpyDeviceToHost); report("f16 split x3, RNE, scaled to 2^15", C, R);
        gemm_split<false><<<1, 64>>>(dA, dB, dC, sa, sb / 1024.f); hipMemcpy(C.data(), dC, M * N * 4, hipMemcpyDeviceToHost); report("f16 split x3, RNE, B scale 2^-10 lower", C, R);
        gemm_split<false><<<1, 64>>>(dA, dB, dC, sa, sb / 1048576.f); hipMemcpy(C.data(), dC, M * N * 4, hipMemcpyDeviceToHost); report("f16 split x3, RNE, B scale 2^-20 lower", C, R);
        gemm_split<true><<<1, 64>>>(dA, dB, dC, sa, sb); hipMemcpy(C.data(), dC, M * N * 4, hipMemcpyDeviceToHost); report("f16 split x3, RTZ, scaled to 2^15", C, R);
    }
    float out[4];
    denorm_probe<<<1, 64>>>(dout); hipMemcpy(out, dout, 16, hipMemcpyDeviceToHost);
    printf("subnormal inputs: 32 x (2^-20 * 1) = %.6e (kept -> 3.051758e-05, flushed -> 0); a as float %.6e; cvt(3.0e-6, -2.0e-7) -> %.6e %.6e\n", out[0], out[1], out[2], out[3]);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int which = 0; which < 2; ++which) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (which == 0) rate_f16<<<256, 512>>>(dout, iters); else rate_f32<<<256, 512>>>(dout, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double n_mfma = 256.0 * 8 * iters * 8, flop = n_mfma * (which == 0 ? 16384.0 : 2048.0);
        printf("%s: %.3f ms, %.1f TFLOP/s, %.2f ns per MFMA per SIMD-pair-of-waves\n", which == 0 ? "v_mfma_f32_16x16x32_f16" : "v_mfma_f32_16x16x4_f32 ", ms, flop / ms * 1e-9, ms * 1e6 / (iters * 8 * 2));
    }
    return 0;
}
