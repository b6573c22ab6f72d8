// Probe (MI355X): issue cost of the act kernel's vector instructions.  Each kernel runs a long stream of INDEPENDENT copies of one
// instruction (8 destination registers in rotation) in every wave; time per instruction and wave -> cycles at the clock the
// v_fma_f32 stream implies (taken as 4 cycles per wave64 instruction).  W = 1 or 2 waves per SIMD (256 or 512 threads per workgroup,
// one workgroup per CU).  Also: the same streams with one v_mfma_f32_16x16x32_f16 per K instructions (K = 3), i.e. what an
// instruction costs when it sits between matrix instructions.
// build: hipcc --offload-arch=gfx950 -O3 valu_cost.hip -o valu_cost.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define OPS(X)                                                                                                              \
    X(0, "v_fma_f32", "v_fma_f32 %0, %1, %2, %0", "+v")                                                                       \
    X(1, "v_max_i32", "v_max_i32 %0, %1, %0", "+v")                                                                            \
    X(2, "v_cvt_pk_f16_f32", "v_cvt_pk_f16_f32 %0, %1, %2", "=v")                                                              \
    X(3, "v_fma_mix_f32", "v_fma_mix_f32 %0, 1.0, %1, -%2 op_sel_hi:[0,0,1]", "=v")                                           \
    X(4, "v_fma_mixlo_f16", "v_fma_mixlo_f16 %0, 1.0, %1, -%2 op_sel_hi:[0,0,1]", "+v")                                       \
    X(5, "v_pk_mul_f32", "v_pk_mul_f32 %0, %1, %2", "=v")                                                                      \
    X(6, "v_pk_fma_f32", "v_pk_fma_f32 %0, %1, %2, %0", "+v")                                                                  \
    X(7, "v_cos_f32", "v_cos_f32 %0, %1", "=v")                                                                                \
    X(8, "v_add_f32_dpp", "v_add_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1", "=v")       \
    X(9, "v_mul_f32", "v_mul_f32 %0, %1, %2", "=v")                                                                            \
    X(10, "v_cvt_pkrtz_f16_f32", "v_cvt_pkrtz_f16_f32 %0, %1, %2", "=v")                                                      \
    X(11, "v_pk_max_f16", "v_pk_max_f16 %0, %1, %2", "=v")                                                                     \
    X(12, "v_readlane_b32 (to sgpr)", "v_readlane_b32 s20, %1, 3", "=v")                                                      \
    X(13, "v_mov_b32", "v_mov_b32 %0, %1", "=v")                                                                              \
    X(14, "s_nop 0", "s_nop 0 ; %0 %1 %2", "+v")                                                                               \
    X(15, "s_waitcnt lgkmcnt(0)", "s_waitcnt lgkmcnt(0) ; %0 %1 %2", "+v")                                                     \
    X(16, "s_mov_b32", "s_mov_b32 s20, 7 ; %0 %1 %2", "+v")

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int OP, int K>
__global__ void kern(float *out, int iters) {
    f32x2 d[8];
    f32x2 a = {threadIdx.x * 1e-3f + 0.25f, 0.5f}, b = {0.999f, 1.001f};
    for (int j = 0; j < 8; ++j) d[j] = (f32x2){(float)j, 1.0f};
    f16x8 ma, mb;
    for (int i = 0; i < 8; ++i) { ma[i] = (_Float16)(threadIdx.x * 1e-3f); mb[i] = (_Float16)1.0f; }
    f32x4 acc[4];
    for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 48; ++u) {
            if (K > 0 && u % K == 0) acc[(u / K) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ma, mb, acc[(u / K) & 3], 0, 0, 0);
#define X(ID, NAME, ASM, CON)                                                                                              \
            if (OP == ID) {                                                                                                \
                if (ID == 5 || ID == 6) asm volatile(ASM : CON(d[u & 7]) : "v"(a), "v"(b));                                  \
                else asm volatile(ASM : CON(d[u & 7].x) : "v"(a.x), "v"(b.x));                                               \
            }
            OPS(X)
#undef X
        }
    }
    float s = 0;
    for (int j = 0; j < 8; ++j) s += d[j].x + d[j].y;
    for (int j = 0; j < 4; ++j) s += acc[j][0];
    if (s == 12345.678f) out[0] = s;
}

template <int OP, int K>
static float run(int threads, int iters, float *dout) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        kern<OP, K><<<256, threads>>>(dout, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    return ms;
}

int main() {
    float *dout; (void)hipMalloc(&dout, 64);
    const int iters = 4000;
    const double n_inst = 48.0 * iters;
    printf("# ns per instruction and wave (one workgroup per CU; W waves per SIMD); cycles = relative to v_fma_f32 at W = 1 taken as 4\n");
    float ref1 = 0;
#define X(ID, NAME, ASM, CON)                                                                                              \
    {                                                                                                                      \
        const float t1 = run<ID, 0>(256, iters, dout), t2 = run<ID, 0>(512, iters, dout), t3 = run<ID, 3>(512, iters, dout);   \
        if (ID == 0) ref1 = t1;                                                                                            \
        const double c = 4.0 / (ref1 * 1e6 / n_inst);                                                                      \
        printf("%-26s W=1 %6.2f ns (%5.1f cyc)   W=2 %6.2f ns per wave-instr (%5.1f cyc of the SIMD per instr)   W=2 with an MFMA every 3: %6.2f ns per MFMA+3 group (%5.1f cyc)\n", NAME, \
               t1 * 1e6 / n_inst, t1 * 1e6 / n_inst * c, t2 * 1e6 / n_inst, t2 * 1e6 / n_inst * c / 2, t3 * 1e6 / (n_inst / 3) / 2, t3 * 1e6 / (n_inst / 3) * c / 2); \
    }
    OPS(X)
#undef X
    {
        const float t = run<13, 1>(512, iters, dout);      // MFMA after every v_mov: ~ MFMA-bound
        printf("MFMA + 1 v_mov, W=2: %6.2f ns per MFMA of the SIMD\n", t * 1e6 / n_inst / 2);
        // one wave per SIMD: how fast does a wave's own stream of MFMA + K plain instructions run without a partner?
        const float a1 = run<13, 1>(256, iters, dout), a2 = run<1, 2>(256, iters, dout), a3 = run<1, 3>(256, iters, dout), a4 = run<1, 4>(256, iters, dout);
        const float b2 = run<1, 2>(512, iters, dout), b4 = run<1, 4>(512, iters, dout);
        printf("ONE wave per SIMD, MFMA + K x v_max_i32: K=1 %6.2f  K=2 %6.2f  K=3 %6.2f  K=4 %6.2f ns per MFMA;   two waves: K=2 %6.2f  K=4 %6.2f ns per MFMA of the SIMD\n",
               a1 * 1e6 / n_inst, a2 * 1e6 / (n_inst / 2), a3 * 1e6 / (n_inst / 3), a4 * 1e6 / (n_inst / 4), b2 * 1e6 / (n_inst / 2) / 2, b4 * 1e6 / (n_inst / 4) / 2);
    }
    return 0;
}
