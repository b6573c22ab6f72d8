// iqn_train.hip -- fused IQN gradient step for gfx950 (MI355X): forward of the target and the local network,
// quantile-Huber TD loss, backward, gradient-norm clip and Adam in three launches.
//
// Replaces, for one optimizer step of IQNAgent.train (thirdparty/IQN/agent.py:269-304) on a batch drawn from the
// device replay ring:
//     Q_targets_next = target(next_states, 8 taus).max over actions            (agent.py:279-281)
//     Q_targets      = r + gamma * Q_targets_next * (1 - done)                 (:283)
//     Q_expected     = local(states, 8 taus).gather(action)                    (:285-286)
//     td[b,i,j]      = Q_targets[b,j] - Q_expected[b,i];  Huber(kappa = 1)     (:289-292, 401-407)
//     loss           = (|tau_i - 1[td < 0]| * huber).sum(i).mean(j).mean(b)    (:293-295)
//     backward; clip_grad_norm_(0.5); Adam(lr 1e-4)                            (:298-301)
// with the network of thirdparty/IQN/model.py:160-186 (linear encoders without activation, cos embedding,
// Hadamard product, three more linear layers).
//
// Why kernels: in PyTorch the step is ~150 tiny dependent kernels (forward x2, autograd, clip, Adam); even replayed
// from a hipGraph it takes ~540 us, all launch latency -- the arithmetic is 0.5 GFLOP.  Here (round 3 structure):
//   iqn_train_fwdbwd  2 x (batch / 2) workgroups of 512 threads, one per CU, in TWO ROLES.  Workgroups [0, batch/2) are
//                     TARGET workgroups: each runs the target network on the next_states of 2 batch elements (16 (sample,
//                     tau) rows = one MFMA M tile) and publishes the 16 TD targets as self-tagged 8-byte granules.
//                     Workgroups [batch/2, batch) are LOCAL workgroups: local network forward on the states of the same 2
//                     elements on all eight waves, pick up the 16 TD targets (they are ready by then: both roles run the
//                     same forward at the same time on different CUs), loss gradient, the whole backward out of LDS, partial
//                     parameter gradient [35 785] to HBM.  The batch's ring rows come from a keyed pseudo-random PERMUTATION
//                     of [0, ring_size) (slot k -> row perm(k): distinct by construction, O(1) per slot), so no workgroup
//                     has to look at another slot's draw.  Every weight operand of the forward, and the two transposed ones
//                     of the backward, is requested into registers before the first barrier: the phases of the chain no
//                     longer start with an L2 round trip.
//   iqn_grad_reduce   sums the partials in a fixed order (deterministic, no float atomics), all loads of a thread in flight
//                     at once -> flat gradient, loss, per-block sums of squares of the reduced gradient;
//   iqn_adam          global norm (from the block sums, or -- after an all-reduce rewrote the gradient -- recomputed by every
//                     block in one fixed order), clip coefficient, Adam update (torch.optim.Adam arithmetic), one flat pass.
// Between the last two the caller may all-reduce the flat gradient (shared learner over RCCL); the 1 / world_size average
// is folded into iqn_adam (`grad_scale`).
//
// MFMA mapping: exact-f32 v_mfma_f32_16x16x4_f32 throughout (the reference trains in float32).  Every product is a
// 16x16 output tile accumulated over K in blocks of 16: lane l = (i = l & 15, g = l >> 4) feeds A[i][k] and B[k][i]
// for the four k = k0 + 4g + s, s = 0..3 of its block -- the k order of a dot product is free, and this one makes
// the lane's four values CONTIGUOUS whenever the operand is stored with k as the fast index (one 16-byte load feeds
// four MFMAs).  Operands stored the other way (k strided) take four scalar loads.  With that, the same primitive
// covers the forward (activations x W^T), the data gradients (dY x W) and the weight gradients (dY^T x X, contracted
// over the 16 rows of the workgroup) without any transposed copy of weights or activations.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <mutex>

#include "marinenav_hip.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned long long gu64;   // global, for agent-scope (sc1) granule accesses

constexpr int OBS = MN_OBS_DIM;  // 26
constexpr int F = 208;           // feature width 16 + 16 + 176
constexpr int NC = 64;           // cos embedding inputs
constexpr int H = 64;            // hidden width
constexpr int NA = 9;            // actions
constexpr int NQ = 8;            // taus per sample in training (agent.py:61 N = 8)
constexpr int BE = 2;            // batch elements per workgroup
constexpr int ROWS = BE * NQ;    // 16 = one MFMA M tile
// flat parameter vector = ObsEncoder.named_parameters() order (model.py:120-136)
constexpr int O_VW = 0, O_VB = 32, O_GW = 48, O_GB = 80, O_SW = 96, O_SB = 3968, O_W1 = 4144, O_B1 = 17456,
              O_W2 = 17664, O_B2 = 30976, O_W3 = 31040, O_B3 = 35136, O_W4 = 35200, O_B4 = 35776, P_TOTAL = 35785;
constexpr int P_PAD = 35788;     // row stride of the per-workgroup partial gradients: 16-byte aligned rows
constexpr int LDC = 68;          // row stride of the 64-wide LDS activations (16-byte aligned rows, bank skew)
constexpr int LDF = 212;         // row stride of the 208-wide LDS activations
// LDS layout (floats); every 2-D block starts 16-byte aligned.  Forward pass (kept for the backward in a local workgroup):
constexpr int S_C = 0;                       // [16][LDC]  cos features
constexpr int S_H1 = S_C + ROWS * LDC;       // [16][LDF]  relu(cos W1^T + b1)
constexpr int S_X = S_H1 + ROWS * LDF;       // [16][LDF]  h1 * features
constexpr int S_DX = S_X + ROWS * LDF;       // [16][LDF]  dL/dx, then dL/d(pre-activation of h1)
constexpr int S_H2 = S_DX + ROWS * LDF;      // [16][LDC]
constexpr int S_H3 = S_H2 + ROWS * LDC;      // [16][LDC]
constexpr int S_DH2 = S_H3 + ROWS * LDC;     // [16][LDC]
constexpr int S_DH3 = S_DH2 + ROWS * LDC;    // [16][LDC]
constexpr int S_FEAT = S_DH3 + ROWS * LDC;   // [2][208]   encoder outputs
constexpr int S_DF = S_FEAT + BE * F;        // [2][208]   dL/dfeatures
constexpr int S_OBS = S_DF + BE * F;         // [2][28]    states (local role) / next_states (target role)
constexpr int S_Q = S_OBS + BE * 28;         // [16][12]   quantile values
constexpr int S_QT = S_Q + ROWS * 12;        // [16] TD targets
constexpr int S_G = S_QT + ROWS;             // [16] dL/dQ_expected
constexpr int S_TAU = S_G + ROWS;            // [16] this role's taus
constexpr int S_MISC = S_TAU + ROWS;         // rew[2], done[2], loss terms[16]
constexpr int S_LOCAL = (S_MISC + 2 * BE + ROWS + 3) / 4 * 4;
// second set of forward buffers: only used when a local workgroup has to run the target forward itself (mode 1, or the
// TD targets of its target workgroup did not arrive in time); the local pass's activations must survive for the backward
constexpr int T_C = S_LOCAL;
constexpr int T_X = T_C + ROWS * LDC;
constexpr int T_H2 = T_X + ROWS * LDF;
constexpr int T_H3 = T_H2 + ROWS * LDC;
constexpr int T_FEAT = T_H3 + ROWS * LDC;
constexpr int T_OBS = T_FEAT + BE * F;
constexpr int T_Q = T_OBS + BE * 28;
constexpr int T_TAU = T_Q + ROWS * 12;
constexpr int S_TOTAL = T_TAU + ROWS;
constexpr int LDS_BYTES = S_TOTAL * 4;
constexpr int THREADS = 512;
constexpr int NT1 = F / 16;      // 13 column tiles of layer 1

struct PassBufs { float *c, *h1, *x, *h2, *h3, *feat, *q; };

// Phase stamps of ONE target and ONE local workgroup (100 MHz wall clock), only in the profiling build
// (-DMN_TRAIN_PHASES: scripts/train_phase_timing.py compiles its own copy of this file; the shipped library has no stamps).
#ifdef MN_TRAIN_PHASES
__device__ unsigned long long g_phase[2][32];
__device__ unsigned long long g_phase2[2][2][8];      // [reduce, adam][block 0, a middle block][stamp]
__device__ unsigned long long g_wgt[1024][4];         // every forward / backward workgroup: start, end, row acknowledged, group flags seen
__device__ unsigned long long g_phase3[3][8];         // reduction + Adam blocks (fused launch or third role): first, middle, last block
#define PH2(kern, k) do { if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2)) g_phase2[kern][blockIdx.x == 0 ? 0 : 1][k] = wall_clock64(); } while (0)
#define PH3(k) do { if (threadIdx.x == 0 && (vb == 0 || vb == nvb / 2 || vb == nvb - 1)) g_phase3[vb == 0 ? 0 : (vb == nvb - 1 ? 2 : 1)][k] = wall_clock64(); } while (0)
#define PH(k) do { if (threadIdx.x == 0 && (blockIdx.x == 0 || (int)blockIdx.x == ph_local)) g_phase[blockIdx.x == 0 ? 0 : 1][k] = wall_clock64(); } while (0)
#else
#define PH(k) do { } while (0)
#define PH2(kern, k) do { } while (0)
#define PH3(k) do { } while (0)
#endif

// The thread index, opaque to the optimiser at every use.  The fused kernel runs its steps in a loop; with plain threadIdx.x every per-thread address and index of a
// step is loop-invariant, gets hoisted in front of the loop and lives across the whole body (measured: 1 KB of scratch per lane).  Derived again where it is used, it
// costs a handful of vector instructions per call site.
__device__ __forceinline__ int tid_now() {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}

// ---- parameter loads ------------------------------------------------------------------------------------------------------
// A flat parameter vector as the forward / backward kernel reads it (element index -> value).
struct ParamView {
    const float *p;
    __device__ __forceinline__ explicit ParamView(const float *P) : p(P) {}
    __device__ __forceinline__ float f1(int i) const { return p[i]; }
    __device__ __forceinline__ float2 f2(int i) const { return *reinterpret_cast<const float2 *>(p + i); }      // i even
    __device__ __forceinline__ float4 f4(int i) const { return *reinterpret_cast<const float4 *>(p + i); }      // i a multiple of 4
};

// ---- MFMA tile primitives --------------------------------------------------------------------------------------------
// B operand of a 16x16 tile over K = 16 * NB, element (k, n), requested into registers (the loads are issued here; nothing waits).  W = parameter offset `off`.
//   k-contiguous (nn.Linear weight used as W^T in the forward: element (k, n) at W[n * ldb + k]): one 16-byte load per block.
template <int NB>
__device__ __forceinline__ void load_b_kcontig(float (&b)[NB][4], const ParamView &P, int off, int ldb, int row = -1) {
    const int lane = tid_now() & 63, i = row < 0 ? lane & 15 : row, g = lane >> 4;
#pragma unroll
    for (int kk = 0; kk < NB; ++kk) {
        const float4 v = P.f4(off + i * ldb + kk * 16 + 4 * g);
        b[kk][0] = v.x; b[kk][1] = v.y; b[kk][2] = v.z; b[kk][3] = v.w;
    }
}
//   the same, or -- when `real` is false -- NB requests of the one 16-byte word at the head of the vector (a wave that has no such tile)
template <int NB>
__device__ __forceinline__ void load_b_kcontig_if(float (&b)[NB][4], const ParamView &P, int off, int ldb, bool real) {
    const int lane = tid_now() & 63, i = lane & 15, g = lane >> 4;
#pragma unroll
    for (int kk = 0; kk < NB; ++kk) {
        const float4 v = P.f4(real ? off + i * ldb + kk * 16 + 4 * g : 0);
        b[kk][0] = v.x; b[kk][1] = v.y; b[kk][2] = v.z; b[kk][3] = v.w;
    }
}
//   k-strided (the same weight used untransposed in the backward: element (k, n) at W[k * ldb + n]): four scalar loads per block.
template <int NB>
__device__ __forceinline__ void load_b_kstrided(float (&b)[NB][4], const ParamView &P, int off, int ldb) {
    const int lane = tid_now() & 63, i = lane & 15, g = lane >> 4;
#pragma unroll
    for (int kk = 0; kk < NB; ++kk)
#pragma unroll
        for (int s = 0; s < 4; ++s) b[kk][s] = P.f1(off + (kk * 16 + 4 * g + s) * ldb + i);
}
// acc += A . B with A (rows x k, k contiguous, 16-byte aligned rows) in LDS and B in registers.
template <int NB>
__device__ __forceinline__ f32x4 mma_a_lds(const float *A, int lda, const float (&b)[NB][4], f32x4 acc) {
    const int lane = tid_now() & 63, i = lane & 15, g = lane >> 4;
    float a[NB][4];
#pragma unroll
    for (int kk = 0; kk < NB; ++kk) {
        const float4 v = *reinterpret_cast<const float4 *>(A + i * lda + kk * 16 + 4 * g);
        a[kk][0] = v.x; a[kk][1] = v.y; a[kk][2] = v.z; a[kk][3] = v.w;
    }
#pragma unroll
    for (int kk = 0; kk < NB; ++kk)
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kk][s], b[kk][s], acc, 0, 0, 0);
    return acc;
}
// NT tiles of C = A^T . B over the workgroup's 16 rows that share ONE A tile (16 columns of A) and take B column tiles
// nk0, nk0 + step, ...: every LDS operand of the wave is requested before the first MFMA, every MFMA issued before the first
// store (one tile at a time -- read, 4 MFMAs, store -- is a 750-cycle latency chain per tile).  Tiles beyond nk_end are computed
// on a clamped copy and not stored.  st(nk, acc): acc[r] = C[4 g + r][i] of tile nk.
template <int NT, typename Store>
__device__ __forceinline__ void rows_gemm_fixed_a(const float *A, int lda, const float *B, int ldb, int nk0, int step, int nk_end,
                                                  Store st) {
    const int lane = tid_now() & 63, i = lane & 15, g = lane >> 4;
    float a[4], b[NT][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) a[s] = A[(4 * g + s) * lda + i];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int nk = min(nk0 + j * step, nk_end - 1);
#pragma unroll
        for (int s = 0; s < 4; ++s) b[j][s] = B[(4 * g + s) * ldb + nk * 16 + i];
    }
    f32x4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[j][s], acc[j], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int nk = nk0 + j * step;
        if (nk < nk_end) st(nk, acc[j]);
    }
}

// The same with ONE B tile shared and A column tiles mk0, mk0 + step, ...: st(mk, acc) with acc[r] = C[4 g + r][i] of A tile mk.
// (Used transposed for the weight gradients: with A = the layer's INPUT tile and B = the output-gradient tile a lane ends up with four
// consecutive input columns of one output row -- one 16-byte store instead of four scattered 4-byte ones.)
template <int NT, typename Store>
__device__ __forceinline__ void rows_gemm_fixed_b(const float *A, int lda, const float *B, int ldb, int mk0, int step, int mk_end,
                                                  Store st) {
    const int lane = tid_now() & 63, i = lane & 15, g = lane >> 4;
    float b[4], a[NT][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) b[s] = B[(4 * g + s) * ldb + i];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int mk = min(mk0 + j * step, mk_end - 1);
#pragma unroll
        for (int s = 0; s < 4; ++s) a[j][s] = A[(4 * g + s) * lda + mk * 16 + i];
    }
    f32x4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][s], b[s], acc[j], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int mk = mk0 + j * step;
        if (mk < mk_end) st(mk, acc[j]);
    }
}

// 16-byte store of a partial-gradient tile row.  MN_PSTORE: 0 = plain (stays dirty in this XCD's L2 until the kernel's end: the
// 18 MB then drain at the launch boundary), 1 = sc1 (write-through at agent scope), 2 = nt (streaming, the default), 3 = sc0 sc1.
// Measured per gradient step (scripts/train_variants.py, one GPU, alternating): plain 40.8 us, sc1 39.1, sc0 sc1 39.1, nt 37.0-37.2.
// All four are ordinary stores as far as visibility goes: the reduction kernel is a later launch.
#ifndef MN_PSTORE
#define MN_PSTORE 2
#endif

typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
#ifndef MN_HIER_STORE
#define MN_HIER_STORE 0
#endif
__device__ __forceinline__ void pstore4(float *base, __amdgpu_buffer_rsrc_t rsrc, int float_off, const f32x4 &v, bool wt = false, bool keep = false) {
    if (keep) {    // XCD-grouped one-launch step: the row is read back through this XCD's L2 a few microseconds from now -- an ordinary store
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4s, v), rsrc, float_off * 4, 0, MN_HIER_STORE);
        return;
    }
    if (wt) {      // one-launch step: write-through at agent scope (sc1) -- the row is read by other XCDs' workgroups of THIS launch, and the
                   // "row complete" word behind it then needs no L2 write-back, only the stores' acknowledgements (s_waitcnt vmcnt(0))
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4s, v), rsrc, float_off * 4, 0, 16);
        return;
    }
#if MN_PSTORE == 0
    *reinterpret_cast<f32x4 *>(base + float_off) = v;
#elif MN_PSTORE == 1
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4s, v), rsrc, float_off * 4, 0, 16);
#elif MN_PSTORE == 2
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4s, v), rsrc, float_off * 4, 0, 2);
#else
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4s, v), rsrc, float_off * 4, 0, 17);
#endif
}

// the small rest of a partial row (biases, output layer, encoders: 15 % of it) goes out as plain stores: non-temporal 4-byte
// stores measured slower (39.4 vs 37.0 us per step), as did non-temporal loads in the reduction (42.8)
__device__ __forceinline__ void pstore1(float *p, float v, bool wt = false) {
    if (wt) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

// ---- arithmetic shared by the launch forms ----------------------------------------------------------------------------------
// This file is compiled with -ffp-contract=off (csrc/Makefile) and every fused multiply-add it wants is WRITTEN OUT: left to the compiler, `a * b + c * d` was
// contracted one way in iqn_adam and another way in reduce_adam_body<1> once an unrelated branch was added there (round 5) -- the bit-identity of the launch forms
// must not hang on that.  Same lesson as the env kernels (mn_device.h).
__device__ __forceinline__ float sumsq4(float a, float b, float c, float d) { return fmaf(d, d, fmaf(c, c, fmaf(b, b, a * a))); }
// torch.optim.Adam's single-tensor update on one element: m.lerp_(g, 1 - b1); v.mul_(b2).addcmul_(g, g, value = 1 - b2); p.addcdiv_(m, sqrt(v) / bc2_sqrt + eps, value = -step_size)
__device__ __forceinline__ void adam_update(float g, float &m, float &v, float &p, float w1, float b2f, float w2, float step_size, float bc2_sqrt, float eps) {
    m = fmaf(g - m, w1, m);
    v = fmaf(w2, g * g, v * b2f);
    p = fmaf(-step_size, m / (sqrtf(v) / bc2_sqrt + eps), p);
}

// ---- the batch draw ----------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t x) {   // splitmix64 finaliser (Steele, Lea, Flood 2014)
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ uint32_t mix32(uint32_t x) {   // murmur3 finaliser
    x ^= x >> 16; x *= 0x85EBCA6Bu;
    x ^= x >> 13; x *= 0xC2B2AE35u;
    return x ^ (x >> 16);
}

constexpr int MAX_BATCH = 1024;
__device__ __forceinline__ uint64_t sample_base(const uint64_t *__restrict__ state) {
    return mix64(state[0] + 0x9E3779B97F4A7C15ull * (state[1] + 1));
}
// tau draw e of the step (e in [0, 2 * batch * 8): target network's first, model.py:149): 24-bit uniform in [0, 1), like torch.rand
__device__ __forceinline__ float sample_tau(uint64_t base, int e) {
    const uint64_t x = mix64(base ^ (0xD1B54A32D192ED03ull * (uint64_t)(e + 1)));
    return (float)(x >> 40) * (1.0f / 16777216.0f);
}
// ReplayBuffer.sample (replay_buffer.py:42-47: random.sample(memory, k) = k DISTINCT uniform rows): slot k of the batch reads ring
// row perm(k), where perm is a pseudo-random permutation of [0, n) keyed by the step's `base` -- a 4-round balanced Feistel network
// on the smallest even-width power-of-two domain >= n (Luby-Rackoff: three rounds of a good round function already give a
// pseudo-random permutation), restricted to [0, n) by cycle walking (domain < 4 n: fewer than four evaluations expected).  The
// first `batch` images of a uniformly random permutation ARE a uniform sample without replacement; distinctness holds by
// construction, so a slot is O(1) and independent of the others -- every workgroup evaluates just its own two slots (round 2 ran
// a draw-and-redraw loop over the whole batch in every workgroup: 4-6 us of the kernel).
__device__ __forceinline__ uint32_t perm_row(uint64_t base, uint32_t n, uint32_t k) {
    const int bits = n > 1 ? 32 - __builtin_clz(n - 1) : 1;
    const int half = (bits + 1) >> 1;
    const uint32_t mask = (1u << half) - 1u;
    uint32_t rk[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) rk[r] = (uint32_t)(mix64(base + 0xA24BAED4963EE407ull * (uint64_t)(r + 1)) >> 32);
    uint32_t x = k;
    do {
        uint32_t L = x >> half, R = x & mask;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t t = L ^ (mix32(R + rk[r]) & mask);
            L = R;
            R = t;
        }
        x = (L << half) | R;
    } while (x >= n);
    return x;
}

// ---- forward pass ------------------------------------------------------------------------------------------------------
// Weight operands of one forward pass, per wave (tile assignment: layer 1 tiles {wave, wave + 8}, layers 2 / 3 tile `wave` on
// waves 0-3, output layer on wave 4), plus this thread's encoder row.
struct FwdWeights {
    float w1a[4][4], w1b[4][4];   // cos_embedding tiles wave, wave + 8 (K = 64)
    float w2[13][4];              // hidden_layer tile (K = 208), waves 0-3
    float w34[4][4];              // hidden_layer_2 tile (K = 64) on waves 0-3, output_layer (9 of 16 columns) on waves 4-7
    float enc[22];                // encoder row of feature (tid % 208): 2 (velocity / goal) or 22 (sonar) weights
    float enc_b, b1a, b1b, b2, b34;
};

// All loads are unconditional and in one straight line (clamped indices instead of branches): a divergent branch around a load
// makes hipcc wait for every outstanding load at the join, which would turn the prefetch into a chain of round trips.
__device__ __forceinline__ void prefetch_forward(FwdWeights &w, const ParamView &P) {
    const int tid = tid_now(), wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, i = lane & 15;
    {   // encoders: thread t < 416 computes feature o = t % 208 of batch element t / 208 (model.py:170-173); a velocity / goal
        // feature uses the first two of the 22 values it loads (they stay inside the flat parameter vector)
        const int o = tid % F;
        const int e = o < 16 ? 0 : 1, oo = o - 16 * e;
        const int woff = o < 32 ? (e ? O_GW : O_VW) + oo * 2 : O_SW + (o - 32) * 22;   // even: 8-byte aligned rows
        const int boff = o < 32 ? (e ? O_GB : O_VB) + oo : O_SB + o - 32;
        const int row = tid < BE * F ? woff : 0;      // threads >= 416: one shared line
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const float2 v = P.f2(row + 2 * k);
            w.enc[2 * k] = v.x; w.enc[2 * k + 1] = v.y;
        }
        w.enc_b = P.f1(boff);
    }
    // A wave without such a tile requests ONE 16-byte word instead (every lane the same address: a single cache-line request), so that
    // the request stream stays branch-free without fetching the tile twice
    const int t1b = min(wave + 8, NT1 - 1);
    const bool has1b = wave + 8 < NT1;
    load_b_kcontig<4>(w.w1a, P, O_W1 + wave * 16 * NC, NC);
    w.b1a = P.f1(O_B1 + wave * 16 + i);
    load_b_kcontig_if<4>(w.w1b, P, O_W1 + t1b * 16 * NC, NC, has1b);
    w.b1b = P.f1(has1b ? O_B1 + t1b * 16 + i : 0);
}
// ... and the operands of layers 2-4 (60 % of the bytes), requested right AFTER the first barrier: a wave cannot write its
// transitions to LDS before it has ISSUED every request in front of that write, and issuing 170 KB per CU takes 2.7 us of the load
// path's time (37.0 -> 36.55 us per step).
__device__ __forceinline__ void prefetch_forward_late(FwdWeights &w, const ParamView &P) {
    const int tid = tid_now(), wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, i = lane & 15;
    const int t23 = wave & 3;
    const bool l3 = wave < 4;
    load_b_kcontig_if<13>(w.w2, P, O_W2 + t23 * 16 * F, F, l3);
    w.b2 = P.f1(l3 ? O_B2 + t23 * 16 + i : 0);
    // waves 0-3: their hidden_layer_2 tile; waves 4-7: the output layer (9 of 16 columns: columns 9..15 read row 8 again and are
    // never stored) -- one load sequence, the address selects
    load_b_kcontig<4>(w.w34, P, l3 ? O_W3 + t23 * 16 * H : O_W4, H, l3 ? i : min(i, NA - 1));
    w.b34 = P.f1(l3 ? O_B3 + t23 * 16 + i : O_B4 + min(i, NA - 1));
}

// Transposed weight operands of the backward (dh2 = dh3 . W3, dx = dh2 . W2: element (k, n) at W[k * ld + n]), per wave: requested
// in the middle of the local forward pass, as soon as the 52 registers of the hidden_layer tile are free.
struct BwdWeights {
    float w3t[4][4];              // hidden_layer_2 columns of dh2 tile wave & 3 (waves 0-3 use it)
    float w2ta[4][4], w2tb[4][4]; // hidden_layer columns of dx tiles wave, wave + 8
};
__device__ __forceinline__ void prefetch_backward(BwdWeights &bw, const ParamView &PL) {
    const int wave = __builtin_amdgcn_readfirstlane(tid_now() >> 6);
    load_b_kstrided<4>(bw.w3t, PL, O_W3 + (wave & 3) * 16, H);
    load_b_kstrided<4>(bw.w2ta, PL, O_W2 + wave * 16, F);
    load_b_kstrided<4>(bw.w2tb, PL, O_W2 + min(wave + 8, NT1 - 1) * 16, F);
}

// model.py:160-186 for the 16 rows of this workgroup on all eight waves: `obs` [2][28], `tau` [16] in LDS (visible: the caller
// placed a barrier after writing them).  Leaves cos, (h1,) x, h2, h3, features and q in LDS; ends with a barrier.
template <bool BWD_PREFETCH>
__device__ __forceinline__ void forward_pass(const PassBufs &Bf, const FwdWeights &w, const float *obs, const float *tau,
                                             BwdWeights *bw, const ParamView &PL, int ph_local = -1) {
    const int tid = tid_now(), wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, i = lane & 15, g = lane >> 4;
    {   // encoders: three linear maps, no activation (model.py:170-173).  Branch-free (a branch here would make hipcc wait for
        // every outstanding weight request): threads >= 416 compute a copy of batch element 1's feature into a dead LDS slot
        const int be = min(tid / F, BE - 1), o = tid % F;
        const bool small = o < 32;
        const float *in = obs + be * 28 + (o < 16 ? 0 : (small ? 2 : 4));
        float acc = w.enc_b;
#pragma unroll
        for (int k = 0; k < 22; ++k) acc = fmaf(w.enc[k], (k < 2 || !small) ? in[k] : 0.f, acc);
        float *dst = tid < BE * F ? Bf.feat + tid : Bf.x + tid;      // x is written by layer 1, after the barrier
        *dst = acc;
    }
    // cos(tau * pi * i), pis = float32(pi * i) (model.py:130,149-155)
#pragma unroll
    for (int e = 0; e < ROWS * NC / THREADS; ++e) {
        const int t = tid + THREADS * e, r = t >> 6, c = t & 63;
        Bf.c[r * LDC + c] = cosf(tau[r] * (float)(M_PI * (double)c));
    }
    __syncthreads();
    PH(2);   /* encoders + cos done */
    // h1 = relu(cos W1^T + b1); x = h1 * features   (13 column tiles: waves 0-4 take two, 5-7 one)
    {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = mma_a_lds<4>(Bf.c, LDC, w.w1a, acc);
        const int o = wave * 16 + i;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * g + r;
            const float v = fmaxf(acc[r] + w.b1a, 0.f);
            if (Bf.h1) Bf.h1[row * LDF + o] = v;
            Bf.x[row * LDF + o] = v * Bf.feat[(row >> 3) * F + o];
        }
    }
    if (wave + 8 < NT1) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = mma_a_lds<4>(Bf.c, LDC, w.w1b, acc);
        const int o = (wave + 8) * 16 + i;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * g + r;
            const float v = fmaxf(acc[r] + w.b1b, 0.f);
            if (Bf.h1) Bf.h1[row * LDF + o] = v;
            Bf.x[row * LDF + o] = v * Bf.feat[(row >> 3) * F + o];
        }
    }
    __syncthreads();
    PH(3);   /* layer 1 */
    if (wave < 4) {   // h2 = relu(x W2^T + b2): one 16-column tile per wave, K = 208
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = mma_a_lds<13>(Bf.x, LDF, w.w2, acc);
        const int o = wave * 16 + i;
#pragma unroll
        for (int r = 0; r < 4; ++r) Bf.h2[(4 * g + r) * LDC + o] = fmaxf(acc[r] + w.b2, 0.f);
    }
    if constexpr (BWD_PREFETCH) prefetch_backward(*bw, PL);
    __syncthreads();
    PH(4);   /* layer 2 */
    if (wave < 4) {   // h3 = relu(h2 W3^T + b3)
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = mma_a_lds<4>(Bf.h2, LDC, w.w34, acc);
        const int o = wave * 16 + i;
#pragma unroll
        for (int r = 0; r < 4; ++r) Bf.h3[(4 * g + r) * LDC + o] = fmaxf(acc[r] + w.b34, 0.f);
    }
    __syncthreads();
    PH(5);   /* layer 3 */
    if (wave == 4) {   // q = h3 W4^T + b4 (9 of 16 columns)
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = mma_a_lds<4>(Bf.h3, LDC, w.w34, acc);
        if (i < NA) {
#pragma unroll
            for (int r = 0; r < 4; ++r) Bf.q[(4 * g + r) * 12 + i] = acc[r] + w.b34;
        }
    }
    __syncthreads();
    PH(6);   /* output layer */
}

// TD targets r + gamma * max_a Q_target(next, tau_j) * (1 - done) (agent.py:281-283) of rows 0..15 from q [16][12] in LDS.
__device__ __forceinline__ float td_target(const float *q, int row, float rew, float done, float gamma) {
    float m = q[row * 12];
#pragma unroll
    for (int a = 1; a < NA; ++a) m = fmaxf(m, q[row * 12 + a]);
    return fmaf(gamma * m, 1.f - done, rew);
}

struct BatchArgs {
    const float *ring_s, *ring_ns, *ring_r, *ring_d;
    const int64_t *ring_a;
    const int64_t *idx;                 // given batch (idx, taus) ...
    const float *taus_t, *taus_l;
    const uint64_t *rng_state;          // ... or drawn from {seed, call counter}
    int64_t ring_n;
    int64_t *idx_out;
    float *taus_out;
};

// Workspace layout (floats): [n_part][P_PAD] partial gradients | [n_part] loss partials (padded to 4) | [N_RED] block sums of
// squares (padded to 4) | [n_part][16] TD-target granules (u64) | epoch (u64), tickets, staging tag | [batch][72] the NEXT step's
// batch, staged by iqn_grad_reduce.  The caller zero-fills the workspace once, before the first call, and passes it unchanged
// afterwards (granule tags, epoch, tickets and the staged batch live there).
#ifndef MN_RED_COLS
#define MN_RED_COLS 32
#endif
#ifndef MN_RED_SEG
#define MN_RED_SEG 8
#endif
constexpr int RED_COLS = MN_RED_COLS;                          // float4 columns per reduction block
constexpr int N_COLS = P_PAD / 4;                              // 8947
constexpr int N_RED = (N_COLS + RED_COLS - 1) / RED_COLS;      // 280 reduction blocks
constexpr int RED_SEG = MN_RED_SEG;                            // partial-sum segments per column (fixed combination order)
__host__ __device__ constexpr int64_t pad4(int64_t x) { return (x + 3) / 4 * 4; }
__host__ __device__ constexpr int64_t ws_loss(int n_part) { return (int64_t)n_part * P_PAD; }
__host__ __device__ constexpr int64_t ws_sq(int n_part) { return ws_loss(n_part) + pad4(n_part); }
__host__ __device__ constexpr int64_t ws_tdq(int n_part) { return ws_sq(n_part) + pad4(N_RED); }      // (NOT rounded up to a cache line: shifting everything behind it by 32 bytes cost the stand-alone reduction 6 us -- measured, round 5)
__host__ __device__ constexpr int64_t ws_epoch(int n_part) { return ws_tdq(n_part) + 2 * (int64_t)n_part * ROWS; }
// epoch block: [0..1] epoch (u64), [2] Adam ticket (u32), [3] reduce ticket (u32), [4..7] staging tag {call counter, ring rows} (2 x u64),
// [8] WS_MAGIC (written by mn_iqn_train_workspace_init: the reduction and Adam kernels refuse a workspace without it), [9] count of local workgroups of
// XCD-grouped one-launch steps that did not run on the XCD of their group's first workgroup (u32, diagnostic: their rows took the slow way through memory),
// [10..11] device pointer (u64) of this rank's gradient mailbox, 0 = none (mn_xchg_attach: the reduction kernel publishes into it)
// [12] count (u32) of reduction + Adam blocks of which a bounded wait ran out, ever (hand-off inside the launch, or a peer's granules in the shared learner's
// exchange): such a block leaves moments and parameters untouched and writes NaN into its piece of the gradient; the caller reads the word at its
// evaluation points and raises (mn_iqn_train_workspace_status_word), [13..15] reserved
// [16 ..] N_RED self-tagged norm partials (u64) of the fused reduction + Adam launch (iqn_grad_reduce_adam)
__host__ __device__ constexpr int64_t ws_xsq(int n_part) { return ws_epoch(n_part) + 16; }
// then, for the one-launch step:
//   ws_done   n_part "row complete" words (u64 {step tag, flags}, agent scope: workgroup w's partial-gradient row is final; flag bit 0 = the row was written
//             through to memory because the workgroup did not run on XCD (block index % 8))
//   ws_gdone  n_part u32 step tags of the UNGROUPED form (workgroup w's row and loss partial are in memory)
//   ws_lflag  8 x 64 u32 step tags, one 256-byte block per XCD group: the same "row complete" news for the workgroups of the same XCD, through that
//             XCD's L2 (ordinary store, sc0 load) -- a third of the latency of the word in memory
//   ws_lossq  n_part self-tagged loss partials (u64 {step tag, value}) of the grouped form
//   ws_xcc    n_part u64 {step tag, XCC_ID}: where each local workgroup runs, published when it starts (the first workgroup of a group defines the group's XCD)
//   ws_grp    the eight group rows (sum of the rows w = x mod 8) as self-tagged 8-byte granules {step tag, value}: [8][P_PAD] u64 -- the data is the flag
__host__ __device__ constexpr int64_t ws_done(int n_part) { return ws_xsq(n_part) + 2 * N_RED; }
__host__ __device__ constexpr int64_t ws_gdone(int n_part) { return ws_done(n_part) + pad4(2 * n_part); }
__host__ __device__ constexpr int64_t ws_lflag(int n_part) { return ws_gdone(n_part) + pad4(n_part); }
__host__ __device__ constexpr int64_t ws_lossq(int n_part) { return ws_lflag(n_part) + 8 * 64; }
__host__ __device__ constexpr int64_t ws_xcc(int n_part) { return ws_lossq(n_part) + pad4(2 * n_part); }
__host__ __device__ constexpr int64_t ws_grp(int n_part) { return ws_xcc(n_part) + pad4(2 * n_part); }
__host__ __device__ constexpr int64_t ws_stage(int n_part) { return ws_grp(n_part) + 2 * (int64_t)RED_SEG * P_PAD; }
constexpr uint32_t WS_MAGIC = 0x4D4E5753u;      // "MNWS"
constexpr int STG = 72;   // floats per staged batch slot: state[26] | next_state[26] | action | reward | done | pad | taus_target[8] | taus_local[8]
__host__ __device__ constexpr int64_t ws_total(int n_part) { return ws_stage(n_part) + (int64_t)n_part * BE * STG; }

constexpr int MODE_TWO_ROLES = 0, MODE_LOCAL_ONLY = 1;

// tag of the gradient a step publishes: never 0 (mailboxes start zero-filled), consecutive steps differ, equal on every rank of a shared
// learner (all ranks have completed the same number of steps)
__host__ __device__ __forceinline__ uint32_t xchg_tag(uint64_t epoch_after_step) { return (uint32_t)(epoch_after_step % 0xFFFFFFFFull) + 1u; }


// the caller's copies of a batch drawn in the launch (inspection, tests): by workgroup 0, after its real work
__device__ __forceinline__ void write_batch_copies(const BatchArgs &ba, uint64_t base, int batch) {
    if (!ba.rng_state) return;
    if (ba.idx_out)
        for (int k = tid_now(); k < batch; k += THREADS) ba.idx_out[k] = perm_row(base, (uint32_t)ba.ring_n, (uint32_t)k);
    if (ba.taus_out)
        for (int e = tid_now(); e < 2 * batch * NQ; e += THREADS) ba.taus_out[e] = sample_tau(base, e);
}

constexpr int N_ADAM = (P_TOTAL + 255) / 256;   // 140 blocks of 256 parameters
constexpr int RED_MAX_PER = 128 / RED_SEG;   // covers batch <= 256 with every load in flight; larger batches loop

constexpr int XCHG_MAX_RANKS = 8;
struct XchgPeers { const gu64 *mb[XCHG_MAX_RANKS]; };      // by rank (this rank's own included)
// the shared learner's exchange as the reduction + Adam blocks see it: a small record in DEVICE memory (mn_xchg keeps it current) that the kernels get a pointer
// to -- as a by-value kernel argument its eleven pointers sat in scalar registers through the whole forward / backward kernel (126 spilled, scratch)
struct XchgArgs {
    XchgPeers peers;
    gu64 *own;             // this rank's mailbox (writable; a field of its own: indexing peers.mb[] with a run-time rank would put the struct into scratch)
    int world;
    unsigned *status;      // device word: blocks whose gather ran into the bound
    uint64_t bound;        // ticks of the 100 MHz counter a gather waits for a peer's granules
};
// e[0..3] = sum over ranks, IN RANK ORDER, of granules q .. q + 3 of the step tagged `tag`.  All ranks' granules are requested together
// (independent system-scope loads in flight over the fabric at once, not one round trip per peer); a pass that finds a stale tag is repeated
// as a whole.  Returns true if the bound (ticks of the 100 MHz counter: mn_xchg_set_timeout_ms) was hit.
__device__ __forceinline__ bool xchg_gather4(const XchgPeers &peers, int world, uint32_t tag, int q, float (&e)[4], uint64_t bound = 200000000ull) {
    uint64_t x[XCHG_MAX_RANKS][4];
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    bool late = false;
    for (;;) {
        bool ok = true;
#pragma unroll
        for (int r = 0; r < XCHG_MAX_RANKS; ++r)
            if (r < world) {
                const gu64 *src = peers.mb[r] + (size_t)(tag & 1u) * P_PAD + q;
#pragma unroll
                for (int k = 0; k < 4; ++k) x[r][k] = __hip_atomic_load(src + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
#pragma unroll
        for (int r = 0; r < XCHG_MAX_RANKS; ++r)
            if (r < world)
#pragma unroll
                for (int k = 0; k < 4; ++k) ok = ok && (uint32_t)(x[r][k] >> 32) == tag;
        if (ok) break;
        if (__builtin_amdgcn_s_memrealtime() - t0 > bound) { late = true; break; }
        __builtin_amdgcn_s_sleep(8);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) e[k] = 0.f;
#pragma unroll
    for (int r = 0; r < XCHG_MAX_RANKS; ++r)
        if (r < world)
#pragma unroll
            for (int k = 0; k < 4; ++k) e[k] += __uint_as_float((uint32_t)x[r][k]);
    return late;
}
// ---- The reduction AND the optimizer step in one launch (round 4; mn_iqn_train_step*): a gradient step is TWO launches ------------------
// Adam block b (of iqn_adam's 140, here with 512 threads) does iqn_grad_reduce's work for ITS 64 float4 columns -- thread (cx, seg) sums
// segment seg of the partials of column 64 b + cx with every load in flight, the eight segment sums are combined in iqn_grad_reduce's order --
// forms the two norm partials those columns make up (iqn_grad_reduce's blocks 2 b and 2 b + 1: same grouping, same order), publishes them as
// self-tagged granules in the workspace, and runs iqn_adam's body for its 256 parameters on the 280 partials it polls: the partials are the
// grid-wide dependency, no barrier, no third launch.  Loss, staging of the next batch, generator counter and
// hand-off epoch as in iqn_grad_reduce.  Every sum in the order of the three-launch path: BIT-IDENTICAL to iqn_grad_reduce + iqn_adam.
// All 140 blocks are resident together (the polls are bounded anyway: ~2 s, then the loss is NaN).
constexpr int RA_COLS = 64, RA_BT = RA_COLS * RED_SEG;      // 512 threads: one wavefront per segment
static_assert(RA_COLS == 2 * RED_COLS && N_RED == 2 * ((P_TOTAL + 255) / 256), "an Adam block's 64 columns are two of the reduction's column groups");
// `vb` of `nvb` = the block's index among the reduction blocks (the stand-alone launch: vb; as the third role of the forward / backward
// launch: vb - its workgroups).  `done` != NULL (third role): the partial gradients are being written by workgroups of the SAME launch;
// done[w] == done_tag says workgroup w's partial row and loss partial are complete -- each wave waits for the 16 rows of its segment.
// (defined with group_reduce below) wavefront 0 waits for the "row complete" news of the rows w = x (mod 8); returns the mask of rows that were written
// through to memory instead of into this XCD's L2, *late = a bounded wait ran out
__device__ __forceinline__ unsigned long long group_rows_wait(float *__restrict__ ws, int n_part, int x, uint32_t tag, int tid, bool *late);
// cache policy of the loads that read a group's partial rows back through the XCD's L2.  1 = sc0, 16 = sc1 (bypasses this CU's vector cache, still served by the L2).
#ifndef MN_ROW_AUX
#define MN_ROW_AUX 16
#endif
constexpr int ROW_AUX = MN_ROW_AUX;
__device__ __forceinline__ void group_put(const __amdgpu_buffer_rsrc_t &grp, uint32_t tag, int c, float4 acc);

// The counters of the step a reduction + Adam block works on.  The stand-alone launch reads them from memory; the fused launch, which may run several steps,
// reads them once when it starts and counts itself (nothing in memory moves until its last step).
struct StepCtx {
    uint64_t epoch;      // hand-off epoch the step starts with (tags derive from it)
    int32_t adam_step;   // optimizer steps done before this one
    uint64_t rs0, ctr;   // generator: seed, call counter of THIS step's batch (unused without rng_state)
    bool last;           // the launch's last step: tickets are taken, the counters in memory advance to behind this step, the next batch is staged
    bool fused;          // role of the forward / backward launch: the eight XCD group rows arrive as self-tagged granules (group_reduce)
};

// Physical block `pb` of `n_phys` runs the virtual blocks vb = pb, pb + n_phys, ... < nvb (at most VPB of them; VPB = 1 and n_phys = nvb everywhere but in the
// fused launch, whose resident blocks cover the 140 virtual ones two each).
template <int VPB = 1>
__device__ __forceinline__ void reduce_adam_body(const int pb, const int n_phys, const int nvb, float *__restrict__ ws, int n_part, float *__restrict__ grad,
                                                 float *__restrict__ loss_out, uint64_t *__restrict__ rng_state, const BatchArgs &ba, int prefetch_next,
                                                 float *__restrict__ params, float *__restrict__ m, float *__restrict__ v, int32_t *__restrict__ step,
                                                 double lr, double b1, double b2, double eps_d, double max_norm_d, const StepCtx sc,
                                                 const XchgArgs *xa = nullptr, float grad_scale = 1.0f) {
    const XchgPeers *peers = xa ? &xa->peers : nullptr;
    const int world = xa ? xa->world : 1;
    const bool grouped = sc.fused;
    const uint32_t done_tag = (uint32_t)(sc.epoch % 0xFFFFFFFFull) + 1u;      // = the forward / backward workgroups' hand-off tag of this step
    const uint32_t *done = sc.fused ? reinterpret_cast<const uint32_t *>(ws + ws_gdone(n_part)) : nullptr;
    __shared__ float4 red[VPB][RED_SEG][RA_COLS];
    __shared__ float sq[VPB][RA_COLS];
    __shared__ float gsh[VPB][4 * RA_COLS];
    __shared__ float nred[4];
    __shared__ float s_bc[2];
    __shared__ int s_lateflag;      // block-wide OR of `late`: written only by a thread whose bounded wait ran out, read behind barriers that are there anyway
                                    // (__syncthreads_or is a workgroup reduction through LDS: ~0.5 us each, three per step)
    // (seg = the wavefront's index: said so explicitly -- behind tid_now() the compiler no longer knows that it is wave-uniform, built the group row's buffer descriptor per
    // lane and wrapped every granule load in a waterfall loop)
    const int tid = tid_now(), cx = tid % RA_COLS, seg = __builtin_amdgcn_readfirstlane(tid / RA_COLS);
    const int vb = pb;      // (block 0 = virtual block 0: the loss; PH3 stamps)
    int vbs[VPB], col[VPB], p[VPB];
    bool on[VPB];
#pragma unroll
    for (int j = 0; j < VPB; ++j) {
        vbs[j] = pb + j * n_phys;
        on[j] = vbs[j] < nvb;      // (block-uniform; see the norm-partial store below for what hipcc does with it)
        col[j] = vbs[j] * RA_COLS + cx;
        p[j] = vbs[j] * 256 + tid;
    }
    if (*reinterpret_cast<const uint32_t *>(ws + ws_epoch(n_part) + 8) != WS_MAGIC) {      // see iqn_grad_reduce
        if (vb == 0 && tid == 0) loss_out[0] = __builtin_nanf("");
        return;
    }
    PH3(0);
    if (tid_now() == 0) s_lateflag = 0;
    __syncthreads();
    const uint32_t tag = xchg_tag(sc.epoch + 1);      // the epoch this step ends with
    // this thread's Adam operands first (threads 0 .. 255 own one parameter each): their latency overlaps the reduction
    float mp[VPB], vp[VPB], pp[VPB];
#pragma unroll
    for (int j = 0; j < VPB; ++j) {
        mp[j] = vp[j] = pp[j] = 0.f;
        if (on[j] && tid < 256 && p[j] < P_TOTAL) { mp[j] = m[p[j]]; vp[j] = v[p[j]]; pp[j] = params[p[j]]; }
    }
    const int t_step = sc.adam_step + 1;
    // staging of the NEXT step's batch (see iqn_grad_reduce): a pure copy, any thread mapping does
    constexpr int SPB = RA_BT / STG;
    const int batch = n_part * BE;
    int st_slot = -1, st_e = 0;
    float st_v = 0.f;
    auto staged_value = [&](uint64_t base_n, int slot, int e) -> float {
        const int64_t row = perm_row(base_n, (uint32_t)ba.ring_n, (uint32_t)slot);
        if (e < OBS) return ba.ring_s[row * OBS + e];
        if (e < 2 * OBS) return ba.ring_ns[row * OBS + e - OBS];
        if (e == 2 * OBS) return (float)ba.ring_a[row];
        if (e == 2 * OBS + 1) return ba.ring_r[row];
        if (e == 2 * OBS + 2) return ba.ring_d[row];
        if (e >= 56) return sample_tau(base_n, (e < 64 ? 0 : batch * NQ) + slot * NQ + (e & 7));
        return 0.f;
    };
    const bool stager = prefetch_next && sc.last && rng_state && vb >= 1 && tid / STG < SPB;
    if (stager) {      // this block's first SPB slots, requested now (blocks 1 .. n_phys - 1 share the batch; more passes, if any, further down)
        st_e = tid % STG;
        const int slot = (vb - 1) * SPB + tid / STG;
        if (slot < batch) st_slot = slot;
    }
    const uint64_t base_n = stager ? mix64(sc.rs0 + 0x9E3779B97F4A7C15ull * (sc.ctr + 2)) : 0;      // = sample_base of the call after this step's
    if (st_slot >= 0) st_v = staged_value(base_n, st_slot, st_e);
    // Segment `seg` of a column = the rows w = seg (mod RED_SEG), ascending (round 4; was: RED_SEG contiguous blocks of rows) -- the rows whose
    // workgroups share an XCD (block index % 8), which is what lets the one-launch step sum a segment inside that XCD's L2 (`grouped`: the eight
    // segment sums are already formed, in ws_grp; one row per segment is left to read).  Same order in every path: all of them stay bit-identical.
    bool late = false;
    PH3(1);
    float lpart = 0.f;      // block 0 sums the loss with iqn_grad_reduce's 256-thread shape
    float4 acc[VPB];
#pragma unroll
    for (int j = 0; j < VPB; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (grouped) {
        // XCD-grouped one-launch step: the eight segment sums arrive as self-tagged granules (group_reduce) -- thread (cx, seg) polls the four of
        // column `col` of group row `seg`, block 0 the n_part loss partials.  No flag, no fence: the data is the flag.
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
        // The buddy: reduction block pb sits (like local workgroup pb of the launch) on XCD pb mod 8 with nothing to do until the granules arrive -- it takes the
        // second half of local workgroup (pb mod 8) + 8 (pb / 8)'s share of its group's row sum, so that a CU moves 72 KB out of the L2 instead of 143 (the L2 ->
        // CU path, 64 B per clock, is what a share takes).  Only when there is exactly one reduction block per local workgroup and 16 rows per group (batch
        // 256), the block really is on its group's XCD, and no row of the group went through memory; it says so in ws_gdone (unused by the grouped form
        // otherwise) when it starts, and the local workgroup halves its share only if it reads that.  A local workgroup that looked too early does the
        // whole share: the same granules twice, bit for bit.
        if (done && n_phys == n_part && n_part == 128) {
            __shared__ int s_buddy;
            const int bx = pb & 7;
            if (tid == 0) {
                unsigned xcc;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                const uint64_t lw = __hip_atomic_load((const gu64 *)(ws + ws_xcc(n_part)) + bx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool ok = (uint32_t)(lw >> 32) == done_tag && (unsigned)(lw & 15u) == (xcc & 15u);
                if (ok) __hip_atomic_store(const_cast<uint32_t *>(done) + pb, done_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_buddy = ok ? 1 : 0;
            }
            __syncthreads();
            if (s_buddy) {
                bool glate = false;
                const unsigned long long gmis = group_rows_wait(ws, n_part, bx, done_tag, tid, &glate);
                if (gmis == 0 && !glate) {      // (glate: the block polls the granules below like any other; whoever waited for the row in vain raises the status word)
                    constexpr int PER = (N_COLS + 15) / 16, HALF = PER / 2;      // 560 columns per local workgroup, 280 of them here
                    const int c = (pb >> 3) * PER + HALF + tid, c1 = min(N_COLS, (pb >> 3) * PER + PER);
                    if (tid < PER - HALF && c < c1) {
                        const __amdgpu_buffer_rsrc_t rows = __builtin_amdgcn_make_buffer_rsrc(ws, 0, n_part * P_PAD * 4, 0x00020000);
                        const __amdgpu_buffer_rsrc_t grp = __builtin_amdgcn_make_buffer_rsrc(ws + ws_grp(n_part) + 2 * (size_t)bx * P_PAD, 0, P_PAD * 8, 0x00020000);
                        float4 t[16];
                        // (sc1: past this CU's vector cache, served by the XCD's L2, which has the line)
#pragma unroll
                        for (int u = 0; u < 16; ++u) t[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rows, ((bx + 8 * u) * N_COLS + c) * 16, 0, ROW_AUX));
                        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                        for (int u = 0; u < 16; ++u) { a.x += t[u].x; a.y += t[u].y; a.z += t[u].z; a.w += t[u].w; }
                        group_put(grp, done_tag, c, a);
                    }
                }
            }
        }
        // (Measured and dropped, +0.6 .. +0.9 us per step each: one wavefront per block first watching the "row complete" words, or "share issued" hints
        // stored behind the granule stores, before everybody looks for the granules.)
        if (vb == 0 && tid < 256) {
            const gu64 *lq = (const gu64 *)(ws + ws_lossq(n_part));
            for (int wq = tid; wq < n_part; wq += 256) {
                uint64_t x;
                for (;;) {
                    x = __hip_atomic_load(lq + wq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((uint32_t)(x >> 32) == done_tag) break;
                    if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) { late = true; break; }
                    __builtin_amdgcn_s_sleep(8);
                }
                lpart += __uint_as_float((uint32_t)x);
            }
        }
        {
#ifndef MN_TAIL_NAP
#define MN_TAIL_NAP 700
#endif
            // Nothing can arrive for a while: these blocks start when the target workgroups end, i.e. when the TD targets are out, and the backward pass behind
            // those takes >= 9 us.  65 000 threads polling 2.3 MB of granules through that time is memory traffic next to the backward pass (on this chip it
            // costs the step nothing measurable -- 33.4-33.6 us with naps of 0 / 5 / 7 / 9 us -- but it is traffic other streams' kernels would see).
            while (__builtin_amdgcn_s_memrealtime() - t0 < (uint64_t)MN_TAIL_NAP) __builtin_amdgcn_s_sleep(32);
            bool want[VPB];
            uint64_t x[VPB][4];
            const __amdgpu_buffer_rsrc_t grp = __builtin_amdgcn_make_buffer_rsrc(ws + ws_grp(n_part) + 2 * (size_t)seg * P_PAD, 0, P_PAD * 8, 0x00020000);
#pragma unroll
            for (int j = 0; j < VPB; ++j) {
                want[j] = on[j] && col[j] < N_COLS;
#pragma unroll
                for (int q = 0; q < 4; ++q) x[j][q] = 0;
            }
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int j = 0; j < VPB; ++j)
                    if (on[j]) {      // (block-uniform) two granules per 16-byte sc1 load; each granule vouches for itself
                        const int c = want[j] ? col[j] : 0;
                        const u32x4s lo = __builtin_amdgcn_raw_buffer_load_b128(grp, c * 32, 0, 16), hi = __builtin_amdgcn_raw_buffer_load_b128(grp, c * 32 + 16, 0, 16);
                        x[j][0] = ((uint64_t)lo[1] << 32) | lo[0]; x[j][1] = ((uint64_t)lo[3] << 32) | lo[2];
                        x[j][2] = ((uint64_t)hi[1] << 32) | hi[0]; x[j][3] = ((uint64_t)hi[3] << 32) | hi[2];
                    }
#pragma unroll
                for (int j = 0; j < VPB; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) ok = ok && (!want[j] || (uint32_t)(x[j][q] >> 32) == done_tag);
                if (ok) break;
                if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) { late = true; break; }
                __builtin_amdgcn_s_sleep(4);
            }
            // (0 + G: the accumulator of the ungrouped paths starts at zero and adds its rows to it; x + 0 = x exactly, also for -0 since the sum of
            // the group's rows was itself formed as 0 + ...)
#pragma unroll
            for (int j = 0; j < VPB; ++j)
                if (want[j])
                    acc[j] = make_float4(__uint_as_float((uint32_t)x[j][0]), __uint_as_float((uint32_t)x[j][1]), __uint_as_float((uint32_t)x[j][2]), __uint_as_float((uint32_t)x[j][3]));
        }
    } else {
    // stand-alone launch: the rows are an earlier launch's
    if (vb == 0 && tid < 256)
        for (int wq = tid; wq < n_part; wq += 256) lpart += ws[ws_loss(n_part) + wq];
#pragma unroll
    for (int j = 0; j < VPB; ++j)
    if (on[j] && col[j] < N_COLS) {
        const float4 *src = reinterpret_cast<const float4 *>(ws) + col[j];
        for (int wb = seg; wb < n_part; wb += RED_SEG * RED_MAX_PER) {
            float4 t[RED_MAX_PER];
#pragma unroll
            for (int u = 0; u < RED_MAX_PER; ++u)
                t[u] = wb + RED_SEG * u < n_part ? src[(size_t)(wb + RED_SEG * u) * N_COLS] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < RED_MAX_PER; ++u) { acc[j].x += t[u].x; acc[j].y += t[u].y; acc[j].z += t[u].z; acc[j].w += t[u].w; }
        }
    }
    }
#pragma unroll
    for (int j = 0; j < VPB; ++j) red[j][seg][cx] = acc[j];
    if (late) s_lateflag = 1;
    __syncthreads();      // (a bounded wait that ran out anywhere in the block poisons the block's output: s_lateflag, read further down)
    PH3(2);
    // This block's ticket, taken HERE -- behind a barrier that every read of the epoch, the Adam step and the generator's counter above sits in front of -- and
    // looked at only at the very end: the round trip of the atomic (~1 us) runs under the norm exchange instead of between Adam and the end of the launch.
    unsigned ticket_old = 0;
    if (tid == 0 && sc.last) ticket_old = __hip_atomic_fetch_add(reinterpret_cast<unsigned *>(ws + ws_epoch(n_part) + 3), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (seg == 0)
#pragma unroll
    for (int j = 0; j < VPB; ++j) {
        float4 s = red[j][0][cx];
#pragma unroll
        for (int q = 1; q < RED_SEG; ++q) { s.x += red[j][q][cx].x; s.y += red[j][q][cx].y; s.z += red[j][q][cx].z; s.w += red[j][q][cx].w; }
        float ss = 0.f;
        float e[4] = {0.f, 0.f, 0.f, 0.f};
        if (on[j] && col[j] < N_COLS) {
            e[0] = s.x; e[1] = s.y; e[2] = s.z; e[3] = s.w;
            ss = sumsq4(s.x, s.y, s.z, s.w);   // padding columns are zeros
        }
        if (peers) {      // shared learner, one-shot exchange IN this launch (mn_iqn_train_step_xchg): publish this rank's columns, gather every rank's
            if (on[j] && col[j] < N_COLS) {
                gu64 *dst = xa->own + (size_t)(tag & 1u) * P_PAD + 4 * col[j];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    __hip_atomic_store(dst + k, ((uint64_t)tag << 32) | (uint64_t)__float_as_uint(e[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (xchg_gather4(*peers, world, tag, 4 * col[j], e, xa->bound)) {
                    late = true;
                    atomicAdd(xa->status, 1u);      // (mn_xchg_status: a peer's granules did not arrive)
                }
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (4 * col[j] + k >= P_TOTAL) e[k] = 0.f;
            }
            float sc[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) sc[k] = e[k] * grad_scale;
            ss = sumsq4(sc[0], sc[1], sc[2], sc[3]);      // iqn_grad_sumsq's expression
        }
        sq[j][cx] = ss;
#pragma unroll
        for (int k = 0; k < 4; ++k) gsh[j][4 * cx + k] = e[k];
    }
    if (late) s_lateflag = 1;      // (the exchange's gather may have run into its bound in some thread)
    __syncthreads();
    gu64 *xsq = (gu64 *)(ws + ws_xsq(n_part));
    if (tid == 0 || tid == RED_COLS)
#pragma unroll
        for (int j = 0; j < VPB; ++j) {
            if (!on[j]) continue;
            // ... and by the index itself: `on[j]` alone is NOT safe here.  hipcc keeps it as a lane mask, re-forms its negation inside the divergent granule poll above (under
            // that loop's EXEC: v_cmp_ne of a v_cndmask of the mask) and uses THAT register for this test -- lanes 0 / 32 that left the poll before its last iteration
            // read 0 = "on" (ROCm 7.2; found in round 5 with a sentinel region: the store then zeroed the TD-target granules that follow the norm partials in the
            // workspace).  A scalar compare of the index cannot be merged with that mask.
            if (2 * vbs[j] + 1 >= N_RED) continue;
            float t = 0.f;
            for (int k = 0; k < RED_COLS; ++k) t += sq[j][tid + k];
            ws[ws_sq(n_part) + 2 * vbs[j] + (tid >> 5)] = t;      // (also where the three-launch path keeps them)
            __hip_atomic_store(xsq + 2 * vbs[j] + (tid >> 5), ((uint64_t)tag << 32) | (uint64_t)__float_as_uint(t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    if (vb == 0) {      // the loss, in iqn_grad_reduce's order
        __shared__ float lw[4];
        float l = lpart;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) l += __shfl_xor(l, off, 64);
        if (tid < 256 && (tid & 63) == 0) lw[tid >> 6] = l;
        __syncthreads();
        if (tid == 0) {
            float t = 0.f;
            for (int k = 0; k < 4; ++k) t += lw[k];
            *loss_out = t;
        }
    }
    if (st_slot >= 0) ws[ws_stage(n_part) + (size_t)st_slot * STG + st_e] = st_v;
    if (stager && n_phys > 1)      // batches with more slots than one pass of the blocks covers (few blocks: the one-launch step's 70 for batch 512)
        for (int j = vb - 1 + (n_phys - 1); j * SPB < batch; j += n_phys - 1) {
            const int slot = j * SPB + tid / STG;
            if (slot < batch)
                ws[ws_stage(n_part) + (size_t)slot * STG + st_e] = staged_value(base_n, slot, st_e);
        }
    // Adam's bias corrections (two float64 pow: ~500 instructions) by a thread of the last wave, which has nothing else to do while waves 0-3 poll the
    // norm partials -- not behind the poll, and not in front of the barrier the partials are published behind; read after the barrier that follows the poll
    if (tid == RA_BT - 1) {
        s_bc[0] = (float)(lr / (1.0 - pow(b1, (double)t_step)));
        s_bc[1] = (float)sqrt(1.0 - pow(b2, (double)t_step));
    }
    // ---- clip + Adam (iqn_adam's body) on the 280 partials, polled: the data is the flag
    float part = 0.f;
    if (tid < 256)
        for (int c = tid; c < N_RED; c += 256) {
            uint64_t x;
            const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
            for (;;) {
                x = __hip_atomic_load(xsq + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((uint32_t)(x >> 32) == tag) break;
                if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) { late = true; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            part += __uint_as_float((uint32_t)x);
        }
    PH3(3);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
    if (tid < 256 && (tid & 63) == 0) nred[tid >> 6] = part;
    if (late) s_lateflag = 1;
    __syncthreads();
    late = s_lateflag != 0;
    const float sumsq = (nred[0] + nred[1]) + (nred[2] + nred[3]);
    const float norm = sqrtf(sumsq);
    const float coef = fminf((float)max_norm_d / (norm + 1e-6f), 1.f);
    const float step_size = s_bc[0], bc2_sqrt = s_bc[1];
    const float wm = (float)(1.0 - b1), b2f = (float)b2, wv = (float)(1.0 - b2), eps = (float)eps_d;
    // A bounded wait ran out somewhere in this block (a hand-off inside the launch, or a peer's granules): what it holds is not this step's gradient.
    // Moments and parameters stay as they are, its piece of the gradient is NaN, and the workspace's status word counts the block -- the caller looks at
    // the word at its evaluation points and raises (a shared learner's replicas would otherwise drift apart silently).
    if (late && tid == 0) atomicAdd(reinterpret_cast<unsigned *>(ws + ws_epoch(n_part) + 12), 1u);
#pragma unroll
    for (int j = 0; j < VPB; ++j)
    if (on[j] && tid < 256 && p[j] < P_TOTAL) {
        float gq = gsh[j][tid] * grad_scale;
        gq *= coef;
        if (late) { grad[p[j]] = __builtin_nanf(""); continue; }
        grad[p[j]] = gq;      // the (clipped) gradient, as iqn_adam leaves it
        float mm = mp[j], vv = vp[j], pn = pp[j];
        adam_update(gq, mm, vv, pn, wm, b2f, wv, step_size, bc2_sqrt, eps);
        m[p[j]] = mm;
        v[p[j]] = vv;
        params[p[j]] = pn;
    }
    // the block that took the LAST ticket advances the generator's call counter, the hand-off epoch and the Adam step, and tags the staged batch
    // (every block read all of them before taking its ticket; nothing in this launch reads them after that)
    PH3(4);
    if (tid == 0 && sc.last) {
        unsigned *ticket = reinterpret_cast<unsigned *>(ws + ws_epoch(n_part) + 3);
        if (ticket_old == (unsigned)n_phys - 1u) {
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *reinterpret_cast<uint64_t *>(ws + ws_epoch(n_part)) = sc.epoch + 1;
            *step = sc.adam_step + 1;
            uint64_t *stg_tag = reinterpret_cast<uint64_t *>(ws + ws_epoch(n_part) + 4);
            if (rng_state) {
                const uint64_t c = sc.ctr + 1;
                rng_state[1] = c;
                stg_tag[0] = c;
                stg_tag[1] = prefetch_next ? (uint64_t)ba.ring_n : 0;
            } else {
                stg_tag[1] = 0;
            }
        }
    }
}

__global__ __launch_bounds__(RA_BT) void iqn_grad_reduce_adam(float *__restrict__ ws, int n_part, float *__restrict__ grad, float *__restrict__ loss_out,
                                                              uint64_t *__restrict__ rng_state, BatchArgs ba, int prefetch_next,
                                                              float *__restrict__ params, float *__restrict__ m, float *__restrict__ v,
                                                              int32_t *__restrict__ step, double lr, double b1, double b2, double eps_d, double max_norm_d) {
    const StepCtx sc = {*reinterpret_cast<const uint64_t *>(ws + ws_epoch(n_part)), *step, rng_state ? rng_state[0] : 0ull, rng_state ? rng_state[1] : 0ull, true, false};
    reduce_adam_body<1>(blockIdx.x, gridDim.x, gridDim.x, ws, n_part, grad, loss_out, rng_state, ba, prefetch_next, params, m, v, step, lr, b1, b2, eps_d, max_norm_d, sc);
}

// ... and with the shared learner's one-shot gradient exchange inside (mn_iqn_train_step_xchg): two launches per step for a shared learner too
__global__ __launch_bounds__(RA_BT) void iqn_grad_reduce_adam_xchg(float *__restrict__ ws, int n_part, float *__restrict__ grad, float *__restrict__ loss_out,
                                                                   uint64_t *__restrict__ rng_state, BatchArgs ba, int prefetch_next,
                                                                   float *__restrict__ params, float *__restrict__ m, float *__restrict__ v,
                                                                   int32_t *__restrict__ step, double lr, double b1, double b2, double eps_d, double max_norm_d,
                                                                   const XchgArgs *__restrict__ xa, float grad_scale) {
    const StepCtx sc = {*reinterpret_cast<const uint64_t *>(ws + ws_epoch(n_part)), *step, rng_state ? rng_state[0] : 0ull, rng_state ? rng_state[1] : 0ull, true, false};
    reduce_adam_body<1>(blockIdx.x, gridDim.x, gridDim.x, ws, n_part, grad, loss_out, rng_state, ba, prefetch_next, params, m, v, step, lr, b1, b2, eps_d, max_norm_d, sc,
                        xa, grad_scale);
}

// One-launch step, XCD-grouped (StepTail::hier): the local workgroups of group x = the rows w = x (mod 8) = the workgroups the dispatcher put on XCD x.
// After its own row is complete a workgroup waits for the rows of its group, takes an equal share of the 8 947 16-byte columns and sums the group's
// rows for them in ascending row order -- reading through the XCD's own L2 (sc0 loads: past this CU's vector cache, served by the L2 that acknowledged
// the writers' stores) -- and writes that piece of the group row to ws_grp as self-tagged granules {step tag, value}, through to memory, where the
// reduction + Adam blocks of every XCD poll them: the data is the flag.  The 18 MB of partial gradients are read back from the L2 they were written to -- no other XCD waits for them; they drain to memory
// when the launch ends, like any dirty line (PMC: 23 MB written per launch in either form) -- and 2.3 MB of granules cross.
// "Row complete" travels twice: as a word in memory (agent scope, with the "written through" flag), and -- from workgroups that are where they should
// be -- as a word in the XCD's L2 (ordinary store; read with an atomic OR of 0, which the L2 executes: an sc0 load may be served by this CU's own vector
// cache, and was 0.8 us slower to notice), which is what the group normally sees first.  A row whose workgroup was NOT on XCD x was
// written through and is read with sc1 loads; such a workgroup takes no share (it cannot see the others' rows) unless the whole group is like that.
// The arithmetic does not depend on any of this.
__device__ __forceinline__ unsigned long long group_rows_wait(float *__restrict__ ws, int n_part, int x, uint32_t tag, int tid, bool *late_out) {
    __shared__ unsigned long long s_mis;
    __shared__ int s_late;
    const int gsz = n_part >> 3;      // n_part % 8 == 0, gsz <= 64 (MAX_BATCH / BE / 8)
    const gu64 *done = (const gu64 *)(ws + ws_done(n_part));
    if (tid < 64) {
        bool have = tid >= gsz, mis_row = false, late = false;
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
        for (int it = 0;; ++it) {
            if (!have) {
                uint32_t lf;      // an atomic OR of 0 with return: executed by the L2, whatever this CU's vector cache holds
                {
                    uint32_t *lp = reinterpret_cast<uint32_t *>(ws + ws_lflag(n_part) + 64 * x) + tid;
                    const uint32_t zero = 0;
                    asm volatile("global_atomic_or %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(lf) : "v"(lp), "v"(zero) : "memory");
                }
                if (lf == tag) have = true;
                else if ((it & 3) == 3) {                                                                           // every fourth look: the word in memory
                    const uint64_t v = __hip_atomic_load(done + x + 8 * tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((uint32_t)(v >> 32) == tag) { have = true; mis_row = (v & 1u) != 0; }
                }
            }
            if (__all(have)) break;
            if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) { late = true; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        const unsigned long long mis = __ballot(mis_row);
        if (tid == 0) { s_mis = mis; s_late = late; }
    }
    __syncthreads();
    *late_out = s_late != 0;
    return s_mis;
}

// column c of a group row: four self-tagged granules, two per 16-byte store, written through (sc1)
__device__ __forceinline__ void group_put(const __amdgpu_buffer_rsrc_t &grp, uint32_t tag, int c, float4 acc) {
    const u32x4s lo = {__float_as_uint(acc.x), tag, __float_as_uint(acc.y), tag}, hi = {__float_as_uint(acc.z), tag, __float_as_uint(acc.w), tag};
    __builtin_amdgcn_raw_buffer_store_b128(lo, grp, c * 32, 0, 16);
    __builtin_amdgcn_raw_buffer_store_b128(hi, grp, c * 32 + 16, 0, 16);
}

__device__ __forceinline__ void group_reduce(float *__restrict__ ws, int n_part, int part, uint32_t tag, int tid) {
    static_assert(RED_SEG == 8, "one segment per XCD");
    const int x = part & 7, gi = part >> 3, gsz = n_part >> 3;
    // the buddy's word (see reduce_adam_body), requested now, looked at behind the wait
    uint32_t buddy_word = 0;
    if (n_part == 128) buddy_word = __hip_atomic_load(reinterpret_cast<const uint32_t *>(ws + ws_gdone(n_part)) + part, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool late_w = false;
    const unsigned long long s_mis = group_rows_wait(ws, n_part, x, tag, tid, &late_w);
    const int s_late = late_w ? 1 : 0;
#ifdef MN_TRAIN_PHASES
    if (tid_now() == 0) g_wgt[blockIdx.x][3] = wall_clock64();
#endif
    const unsigned long long all = gsz == 64 ? ~0ull : ((1ull << gsz) - 1ull), mis = s_mis & all, well = ~mis & all;
    const unsigned long long takers = well ? well : all;      // nobody where it should be: every row is in memory, everybody can read them
    if (!((takers >> gi) & 1ull)) return;
    const int rank = __popcll(takers & ((1ull << gi) - 1ull)), n_takers = __popcll(takers);
    const int per = (N_COLS + n_takers - 1) / n_takers, c0 = rank * per, c1 = min(N_COLS, c0 + per);
    const __amdgpu_buffer_rsrc_t rows = __builtin_amdgcn_make_buffer_rsrc(ws, 0, n_part * P_PAD * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t grp = __builtin_amdgcn_make_buffer_rsrc(ws + ws_grp(n_part) + 2 * (size_t)x * P_PAD, 0, P_PAD * 8, 0x00020000);
    const bool poison = s_late != 0;      // a row never arrived (bounded wait): the step must not look valid
    auto put = [&](int c, float4 acc) {      // column c of the group row: four self-tagged granules, two per 16-byte store, written through (sc1)
        if (poison) acc.x = __builtin_nanf("");
        const u32x4s lo = {__float_as_uint(acc.x), tag, __float_as_uint(acc.y), tag}, hi = {__float_as_uint(acc.z), tag, __float_as_uint(acc.w), tag};
        __builtin_amdgcn_raw_buffer_store_b128(lo, grp, c * 32, 0, 16);
        __builtin_amdgcn_raw_buffer_store_b128(hi, grp, c * 32 + 16, 0, 16);
    };
#ifndef MN_GROUP_FAST
#define MN_GROUP_FAST 1
#endif
    if (MN_GROUP_FAST && mis == 0 && gsz == 16 && buddy_word == tag) {
        // ... and the XCD's reduction block number `part` has taken the second half of this share (reduce_adam_body): 280 columns, one per thread, one round of loads
        constexpr int HALF = ((N_COLS + 15) / 16) / 2;
        const int ca = c0 + (tid < HALF ? tid : 0);      // (threads beyond the half all read column c0 and drop it: no branch between the loads)
        float4 t[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) t[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rows, ((x + 8 * u) * N_COLS + ca) * 16, 0, ROW_AUX));
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 16; ++u) { a.x += t[u].x; a.y += t[u].y; a.z += t[u].z; a.w += t[u].w; }
        if (tid < HALF && ca < c1) put(ca, a);
        return;
    }
    if (MN_GROUP_FAST && mis == 0 && gsz == 16) {
        // The case that runs (batch 256, every workgroup on its XCD): straight-line code, all of a thread's loads in flight before its first add.  With 16
        // takers a share is 560 columns -- 48 threads have a second one (the others all read column c0 again and drop it: no branch between the loads, which
        // would make the compiler wait for each load on its own, as it does in the general loop below).
        const int ca = c0 + tid, cb = c0 + THREADS + tid, cb_eff = cb < c1 ? cb : c0;      // (c0 for all of them: one request per wave)
        float4 t[2][16];
#pragma unroll
        for (int u = 0; u < 16; ++u) t[0][u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rows, ((x + 8 * u) * N_COLS + ca) * 16, 0, ROW_AUX));
#pragma unroll
        for (int u = 0; u < 16; ++u) t[1][u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rows, ((x + 8 * u) * N_COLS + cb_eff) * 16, 0, ROW_AUX));
        float4 acc[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int u = 0; u < 16; ++u) { acc[h].x += t[h][u].x; acc[h].y += t[h][u].y; acc[h].z += t[h][u].z; acc[h].w += t[h][u].w; }
        if (ca < c1) put(ca, acc[0]);
        if (cb < c1) put(cb, acc[1]);
        for (int c = c0 + 2 * THREADS + tid; c < c1; c += THREADS) {      // (never with 16 takers)
            float4 a2 = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int u = 0; u < 16; ++u) {
                const float4 v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rows, ((x + 8 * u) * N_COLS + c) * 16, 0, ROW_AUX));
                a2.x += v.x; a2.y += v.y; a2.z += v.z; a2.w += v.w;
            }
            put(c, a2);
        }
        return;
    }
    if (MN_GROUP_FAST && mis == 0 && (gsz & 7) == 0) {      // batches 128, 384, 512, ...: eight rows' loads in flight at a time
        for (int c = c0 + tid; c < c1; c += THREADS) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int i0 = 0; i0 < gsz; i0 += 8) {
                float4 t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rows, ((x + 8 * (i0 + u)) * N_COLS + c) * 16, 0, ROW_AUX));
#pragma unroll
                for (int u = 0; u < 8; ++u) { acc.x += t[u].x; acc.y += t[u].y; acc.z += t[u].z; acc.w += t[u].w; }
            }
            put(c, acc);
        }
        return;
    }
    for (int c = c0 + tid; c < c1; c += THREADS) {      // any group size, any mix of rows in this L2 (sc0) and rows in memory (sc1)
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = 0; i < gsz; ++i) {
            const int off = ((x + 8 * i) * N_COLS + c) * 16;
            const float4 v = (mis >> i) & 1ull ? __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rows, off, 0, 16))
                                               : __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rows, off, 0, ROW_AUX));
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        put(c, acc);
    }
}

// ---- The fused step (round 5): reduction + clip + Adam inside the forward / backward launch ---------------------------------------------------------------------
// Every TARGET workgroup is also a reduction + Adam block (reduce_adam_body): target workgroup w runs the target network on the next_states of its two batch
// elements -- nothing in it depends on the local network -- and then does reduction + Adam work on the CU it sits on anyway (round 4 launched those blocks
// behind the forward / backward workgroups, onto the CUs the target workgroups had vacated).  Always XCD-grouped (the rows of an XCD's workgroups are summed
// inside its L2); batches whose half is not a multiple of 8 take two launches.
// (Round 5 also ran G steps in ONE persistent launch of this kernel -- mn_iqn_train_steps, bit-identical, 33.4 us per step against 32.4 for G launches: the hand-off
// of freshly written parameters to 128 workgroups on other XCDs cost what the launch boundary did.  Removed in round 6; profiles/r05_train_step_launches.txt.)
struct StepTail {
    int n_virtual;      // != 0: the fused step; the N_ADAM virtual reduction + Adam blocks run on ...
    int n_extra;        // ... the target workgroups + this many workgroups behind the forward / backward ones that do nothing else (small batches: 2 x (n_part + n_extra) >= N_ADAM)
    int misplace;       // test hook: pretend these local workgroups did not land on XCD (block index % 8): 1 = every fifth, 2 = all, 3 = all of group 3
    int prefetch_next;  // the step stages the batch of the call after this launch
    float *grad, *loss_out, *params, *m, *v;
    int32_t *step;
    uint64_t *rng_state;
    double lr, b1, b2, eps, max_norm;
    const XchgArgs *xa; // shared learner (non-NULL): the one-shot gradient exchange inside the reduction + Adam role, of xa_scale x the sum over ranks
    float xa_scale;
};

// XCHG: the instantiation whose reduction + Adam role carries the shared learner's exchange; the single learner's kernel is compiled without that code (its
// mailbox pointers cost 75 more spilled scalar registers in a kernel that has none to spare).  FUSED: the fused step above; false = forward / backward only
// (two- and three-launch forms; `tail` unused): one step, ordinary parameter loads.
// The kernel's arguments as ONE struct: the fused kernel re-reads them from the kernarg segment in every step of its loop (through a pointer the optimiser cannot see
// through) instead of holding ~90 loop-invariant scalar registers across the whole body -- with those held, hipcc spilled 350-420 scalar registers into vector-register
// lanes and 80-1 100 bytes per lane to scratch, and values restored from those spills were sporadically WRONG (a reduction block's `on[1]` came back true: its norm-partial
// store then zeroed other workgroups' TD-target granules; round 5, found with a sentinel region).  No scratch, few spills: what the kernel is built to.
struct TrainArgs {
    BatchArgs ba;
    const float *PL, *PT;
    float *ws;
    int batch;
    float gamma;
    int mode, use_staged;
    StepTail tail;
};
typedef const __attribute__((address_space(4))) TrainArgs *TrainArgsK;
// (field by field: a struct in the constant address space cannot be copied as a whole)
__device__ __forceinline__ BatchArgs ld_batch_args(TrainArgsK A) {
    return BatchArgs{A->ba.ring_s, A->ba.ring_ns, A->ba.ring_r, A->ba.ring_d, A->ba.ring_a, A->ba.idx, A->ba.taus_t, A->ba.taus_l, A->ba.rng_state, A->ba.ring_n,
                     A->ba.idx_out, A->ba.taus_out};
}
__device__ __forceinline__ StepTail ld_step_tail(TrainArgsK A) {
    return StepTail{A->tail.n_virtual, A->tail.n_extra, A->tail.misplace, A->tail.prefetch_next, A->tail.grad, A->tail.loss_out, A->tail.params, A->tail.m,
                    A->tail.v, A->tail.step, A->tail.rng_state, A->tail.lr, A->tail.b1, A->tail.b2, A->tail.eps, A->tail.max_norm, A->tail.xa, A->tail.xa_scale};
}

template <bool XCHG, bool FUSED>
__global__ __launch_bounds__(THREADS) void iqn_train_fwdbwd(const TrainArgs args) {
    extern __shared__ __align__(16) float S[];
    __shared__ int s_act[BE];
    __shared__ int s_got;
    // ---- the counters this launch starts from; nothing in memory moves before its last step's reduction + Adam blocks have all taken their ticket
    uint64_t epoch0, rs0 = 0, rs1 = 0, tg0 = 0, tg1 = 0;      // scalar state first (generator state, staging tag): issued before the weight requests flood the memory pipeline
    int32_t step0 = 0;
    {
        const int n_part = args.batch / BE;
        epoch0 = *reinterpret_cast<const uint64_t *>(args.ws + ws_epoch(n_part));
        if (args.ba.rng_state) {
            const uint64_t *stg_tag = reinterpret_cast<const uint64_t *>(args.ws + ws_epoch(n_part) + 4);
            rs0 = args.ba.rng_state[0]; rs1 = args.ba.rng_state[1];
            tg0 = stg_tag[0]; tg1 = stg_tag[1];
        }
        if (FUSED) step0 = *args.tail.step;
    }
#ifdef MN_TRAIN_PHASES
    if (tid_now() == 0) g_wgt[blockIdx.x][0] = wall_clock64();
#endif

    // Iteration 0: the forward / backward pass (target workgroups: the target forward pass); iteration 1 (fused step): target and extra workgroups run the reduction + Adam role.
    for (int k = 0; k <= (FUSED ? 1 : 0); ++k) {
    // (One step's scalar and address arithmetic must not be hoisted out of the loop -- hundreds of values would then live across the whole body: the thread index is
    // opaque at every use (tid_now), the block index and the kernel arguments -- re-read from the kernarg segment -- are made opaque per iteration, and everything
    // derived from them, the workgroup's role included, is derived again.)
    const int tid = tid_now();
    TrainArgsK A = (TrainArgsK)__builtin_amdgcn_kernarg_segment_ptr();
    int bid = blockIdx.x;
    if (FUSED) asm volatile("" : "+s"(A), "+s"(bid));
    const BatchArgs ba = FUSED ? ld_batch_args(A) : args.ba;
    const StepTail tail = FUSED ? ld_step_tail(A) : args.tail;
    float *const ws = FUSED ? A->ws : args.ws;
    const float *const PL = FUSED ? A->PL : args.PL, *const PT = FUSED ? A->PT : args.PT;
    const int batch = FUSED ? A->batch : args.batch, mode = FUSED ? A->mode : args.mode;
    const int n_part = batch / BE;
    const bool two_roles = mode == MODE_TWO_ROLES;      // (FUSED: always)
    const int n_fwd = two_roles ? 2 * n_part : n_part;
    const bool is_extra = FUSED && bid >= n_fwd;            // reduction + Adam blocks only
    const bool is_target = two_roles && bid < n_part;       // target workgroups come FIRST in dispatch order: nothing they need is produced by a local workgroup
    const int part = is_extra ? 0 : (two_roles && !is_target ? bid - n_part : bid);
    const int b0 = part * BE;
    const int pb = is_extra ? n_part + (bid - n_fwd) : bid, n_phys = n_part + tail.n_extra;      // reduction + Adam role: physical block pb of n_phys
#ifdef MN_TRAIN_PHASES
    const int ph_local = two_roles ? n_part : 0;
#else
    const int ph_local = -1;
#endif
    PH(0);
    const float gamma = FUSED ? A->gamma : args.gamma;
    const int use_staged = FUSED ? A->use_staged : args.use_staged;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, i = lane & 15, g = lane >> 4;
    const ParamView VL(PL);
    const float *stage = ws + ws_stage(n_part);
    const int st_slot = min(tid / STG, BE - 1), st_e = tid % STG;
    float *out = ws + (size_t)part * P_PAD;
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, P_PAD * 4, 0x00020000);
    if (k < 1 && !is_extra) {
    // hand-off tag of this step: never 0 (the workspace starts zero-filled), different from the neighbouring steps' and launches' tags
    const uint32_t tag = (uint32_t)((epoch0 + (uint64_t)k) % 0xFFFFFFFFull) + 1u;
    gu64 *granules = (gu64 *)(ws + ws_tdq(n_part)) + (size_t)part * ROWS;
    if (FUSED && !is_target && tid == 0) {      // where this local workgroup runs (see "local workgroup" below)
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        __hip_atomic_store((gu64 *)(ws + ws_xcc(n_part)) + part, ((uint64_t)tag << 32) | (uint64_t)(xcc & 15u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    // ---- The first requests of a step: (a) -- first step of a launch -- this workgroup's two batch slots as the previous launch's reduction blocks STAGED them
    // (transitions and taus, 72 floats per slot, at an address that depends on nothing but the kernel arguments), (b) every weight
    // operand of the forward pass.  One round trip instead of three dependent ones (generator state -> ring rows -> transitions).
    // (Measured and dropped: gathering the batch in FRONT of that wait -- 1.6 us earlier requests, but two sites that define the 140 weight registers cost 48 bytes of
    // scratch per lane and the step got 1 us longer.)
    const bool try_staged = use_staged && k == 0;
    float st_v = 0.f;
    if (try_staged) st_v = stage[(b0 + st_slot) * STG + st_e];      // kernel argument: a scalar branch
    FwdWeights w;
    const ParamView V(is_target ? PT : PL);
    prefetch_forward(w, V);
    PH(14);  /* all requests issued */
    // the staged batch is this step's batch iff it was drawn for this call counter from a ring of this many rows
    uint64_t base = 0;
    bool staged = false;
    if (ba.rng_state) {
        base = mix64(rs0 + 0x9E3779B97F4A7C15ull * (rs1 + (uint64_t)k + 1));      // = sample_base of the generator at call counter rs1 + k
        staged = try_staged && tg0 == rs1 && tg1 == (uint64_t)ba.ring_n;
    }
    PH(15);  /* generator state / staging tag read */
    int64_t row0 = 0, row1 = 0;      // (BE = 2; selects instead of an indexed array, which would live in scratch)
    if (staged) {      // uniform
        if (tid < BE * STG) {
            const int o = st_slot * 28;
            if (st_e < OBS) { if (!is_target) S[S_OBS + o + st_e] = st_v; }
            else if (st_e < 2 * OBS) { if (is_target) S[S_OBS + o + st_e - OBS] = st_v; else if (!two_roles) S[T_OBS + o + st_e - OBS] = st_v; }
            else if (st_e == 2 * OBS) s_act[st_slot] = (int)st_v;
            else if (st_e == 2 * OBS + 1) S[S_MISC + st_slot] = st_v;
            else if (st_e == 2 * OBS + 2) S[S_MISC + BE + st_slot] = st_v;
            else if (st_e >= 56 && st_e < 64) { if (is_target) S[S_TAU + st_slot * NQ + st_e - 56] = st_v; else if (!two_roles) S[T_TAU + st_slot * NQ + st_e - 56] = st_v; }
            else if (st_e >= 64) { if (!is_target) S[S_TAU + st_slot * NQ + st_e - 64] = st_v; }
        }
    } else {
        // ---- no (valid) staged batch: the batch rows of this workgroup (scalar arithmetic, or two loads in the given-batch form) ...
        float tau_t = 0.f, tau_l = 0.f;
        const int e_t = b0 * NQ + (tid & (ROWS - 1));
        if (ba.rng_state) {
            row0 = perm_row(base, (uint32_t)ba.ring_n, (uint32_t)b0);
            row1 = perm_row(base, (uint32_t)ba.ring_n, (uint32_t)(b0 + 1));
            // taus: target draws first (model.py:149 is called for the target network first, agent.py:279-286)
            tau_t = sample_tau(base, e_t);
            tau_l = sample_tau(base, batch * NQ + e_t);
        } else {
            row0 = ba.idx[b0];
            row1 = ba.idx[b0 + 1];
            tau_t = ba.taus_t[e_t];
            tau_l = ba.taus_l[e_t];
        }
        // ---- ... then the transitions, in one straight line without branches (all threads load, clamped -- the few that matter
        // store to LDS below)
        const int g_be = (tid / OBS) & 1, g_k = tid % OBS;
        const int64_t g_row = g_be ? row1 : row0, m_row = (tid & 1) ? row1 : row0;
        const float g_obs = (is_target ? ba.ring_ns : ba.ring_s)[g_row * OBS + g_k];
        const float g_tobs = ba.ring_ns[g_row * OBS + g_k];          // only kept when this workgroup runs both networks
        const int g_act = (int)ba.ring_a[m_row];
        const float g_rew = ba.ring_r[m_row], g_done = ba.ring_d[m_row];
        if (tid < ROWS) {
            S[S_TAU + tid] = is_target ? tau_t : tau_l;
            if (!two_roles) S[T_TAU + tid] = tau_t;
        }
        // gathered transitions -> LDS (replay_buffer.py:42-57)
        if (tid < BE * OBS) {
            S[S_OBS + g_be * 28 + g_k] = g_obs;
            if (!two_roles) S[T_OBS + g_be * 28 + g_k] = g_tobs;
        }
        if (tid < BE) {
            s_act[tid] = g_act;
            S[S_MISC + tid] = g_rew;
            S[S_MISC + BE + tid] = g_done;
        }
    }
    PH(16);  /* wave 0 has its transitions in LDS */
    __syncthreads();
    prefetch_forward_late(w, V);
    PH(1);   /* draw + gather + weight requests */

    // output-layer row of the action taken, for dh3 (element tid + 512 e of the [16][64] tile: row 8 e + (tid >> 6), column tid & 63,
    // i.e. batch element e): two more early requests
    float w4row[2] = {0.f, 0.f};
    if (!is_target) {
#pragma unroll
        for (int e = 0; e < 2; ++e) w4row[e] = VL.f1(O_W4 + s_act[e] * H + (tid & 63));
    }
    const PassBufs Bl = {S + S_C, is_target ? nullptr : S + S_H1, S + S_X, S + S_H2, S + S_H3, S + S_FEAT, S + S_Q};
    BwdWeights bw;
    if (is_target) forward_pass<false>(Bl, w, S + S_OBS, S + S_TAU, nullptr, VL, ph_local);
    else forward_pass<true>(Bl, w, S + S_OBS, S + S_TAU, &bw, VL, ph_local);

    if (is_target) {
        // publish the 16 TD targets: one self-tagged 8-byte granule each ({epoch, value}, agent-scope store: the data is the flag)
        if (tid < ROWS) {
            const int be = tid >> 3;
            const float v = td_target(S + S_Q, tid, S[S_MISC + be], S[S_MISC + BE + be], gamma);
            __hip_atomic_store(granules + tid, ((uint64_t)tag << 32) | (uint64_t)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        PH(7);   /* granules published */
        if (bid == 0) write_batch_copies(ba, base, batch);
#ifdef MN_TRAIN_PHASES
        if (tid_now() == 0) g_wgt[blockIdx.x][1] = wall_clock64();
#endif
    } else {

    // ---- local workgroup
    // The rows w = x (mod 8) are summed inside one XCD's L2, so their workgroups have to share an XCD.  The dispatcher deals
    // workgroups out to the XCDs round-robin (scripts/probes/xcc_placement.hip) -- from XCD 0 in a fresh process, from another one after other streams were
    // in use -- so block index % 8 names a set of workgroups on ONE XCD, not which.  Each local workgroup publishes the XCD it runs on; a group's XCD is
    // that of its first workgroup, whose word the others read here (long before they need it, behind the TD targets).
    uint64_t lead_word = 0;
    unsigned my_xcc = 0;
    if (FUSED) {
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(my_xcc));
        my_xcc &= 15u;
        if (tid == 0 && (part >> 3) != 0) lead_word = __hip_atomic_load((const gu64 *)(ws + ws_xcc(n_part)) + (part & 7), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- TD targets: from the target workgroup of the same two batch elements (ready by now -- it ran the same forward at the same
    // time on another CU), or computed here (mode 1; or the granules did not arrive within the bound, which in-order workgroup
    // dispatch makes impossible -- kept so that a wait can never hang the device)
    if (two_roles) {
        if (wave == 0) {
            bool ok = false;
            float v = 0.f;
            const uint64_t t0 = __builtin_readcyclecounter();
            for (;;) {
                uint64_t x = (uint64_t)tag << 32;
                if (lane < ROWS) x = __hip_atomic_load(granules + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = (uint32_t)(x >> 32) == tag;
                v = __uint_as_float((uint32_t)x);
                if (__all(ok)) break;
                if (__builtin_readcyclecounter() - t0 > 400000000ull) break;      // ~0.2 s of shader clocks
                __builtin_amdgcn_s_sleep(2);
            }
            const bool all_ok = __all(ok);
            if (all_ok && lane < ROWS) S[S_QT + lane] = v;
            if (lane == 0) s_got = all_ok ? 1 : 0;
            if (!all_ok && lane == 0) atomicAdd(reinterpret_cast<unsigned *>(ws + ws_epoch(n_part) + 13), 1u);      // (diagnostic: TD targets computed here after a wait in vain)
        }
        __syncthreads();
    }
    if (!two_roles || !s_got) {
        if (two_roles) {   // late fallback: the target side's inputs, from the staged slots or gathered now
            const int be = min(tid / OBS, BE - 1), kk = tid % OBS;
            if (staged) {
                if (tid < BE * OBS) S[T_OBS + be * 28 + kk] = stage[(b0 + be) * STG + OBS + kk];
                if (tid < ROWS) S[T_TAU + tid] = stage[(b0 + (tid >> 3)) * STG + 56 + (tid & 7)];
            } else {
                if (tid < BE * OBS) S[T_OBS + be * 28 + kk] = ba.ring_ns[(be ? row1 : row0) * OBS + kk];
                if (tid < ROWS) {
                    const int e_t = b0 * NQ + tid;
                    S[T_TAU + tid] = ba.rng_state ? sample_tau(base, e_t) : ba.taus_t[e_t];
                }
            }
            __syncthreads();
        }
        FwdWeights wt;
        const ParamView VT(PT);
        prefetch_forward(wt, VT);
        prefetch_forward_late(wt, VT);
        const PassBufs Bt = {S + T_C, nullptr, S + T_X, S + T_H2, S + T_H3, S + T_FEAT, S + T_Q};
        forward_pass<false>(Bt, wt, S + T_OBS, S + T_TAU, nullptr, VT, -2);
        if (tid < ROWS) {
            const int be = tid >> 3;
            S[S_QT + tid] = td_target(S + T_Q, tid, S[S_MISC + be], S[S_MISC + BE + be], gamma);
        }
        __syncthreads();
    }

    PH(8);   /* TD targets in LDS (hand-off wait, or own target forward) */
    // ---- quantile-Huber loss and dL/dQ_expected (agent.py:289-295, 401-407); every thread evaluates the (cheap) gradient of
    // the row its dh3 elements belong to, so the loss phase and the output-layer backward share one barrier interval
    // Fused step: the row is read by the workgroups of its group (block index % 8), which share an XCD, through that XCD's L2: ordinary stores.  A workgroup
    // that is NOT on its group's XCD (never observed) writes its row through to memory.
    bool wellplaced = false;
    if (FUSED) {
        __shared__ int s_well;
        if (tid == 0) {
            const gu64 *lw = (const gu64 *)(ws + ws_xcc(n_part)) + (part & 7);
            bool well = (part >> 3) == 0;      // the group's first workgroup is where the group is
            if (!well) {
                const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
                while ((uint32_t)(lead_word >> 32) != tag && __builtin_amdgcn_s_memrealtime() - t0 < 100000ull)      // (1 ms; then: not with the group)
                    lead_word = __hip_atomic_load(lw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                well = (uint32_t)(lead_word >> 32) == tag && (unsigned)(lead_word & 15u) == my_xcc;
            }
            const int gi = part >> 3;
            if (tail.misplace == 1 && gi % 5 == 0 && gi) well = false;
            if ((tail.misplace == 2 && gi) || (tail.misplace == 3 && (part & 7) == 3 && gi)) well = false;
            s_well = well ? 1 : 0;
        }
        __syncthreads();
        wellplaced = s_well != 0;
    }
    if (FUSED && !wellplaced && tid == 0)      // epoch block word [9]: local workgroups that found themselves on another XCD, ever (diagnostic)
        __hip_atomic_fetch_add(reinterpret_cast<unsigned *>(ws + ws_epoch(n_part) + 9), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool wt = FUSED && !wellplaced;
    const bool keep = FUSED && wellplaced;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int t = tid + THREADS * e, r = t >> 6, kq = t & 63, be = r >> 3;
        const float qe = S[S_Q + r * 12 + s_act[be]], tau = S[S_TAU + r];
        float lsum = 0.f, gsum = 0.f;
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const float td = S[S_QT + be * NQ + j] - qe, ad = fabsf(td);
            const float hub = ad <= 1.f ? 0.5f * td * td : ad - 0.5f;
            const float wq = fabsf(tau - (td < 0.f ? 1.f : 0.f));
            lsum = fmaf(wq, hub, lsum);
            gsum = fmaf(wq, fminf(fmaxf(td, -1.f), 1.f), gsum);
        }
        const float scale = 1.f / (float)(batch * NQ);
        const float gr = -gsum * scale;
        if (kq == 0) {
            S[S_G + r] = gr;
            S[S_MISC + 2 * BE + r] = lsum * scale;
        }
        // output layer backward: only the taken action's row carries gradient
        S[S_DH3 + r * LDC + kq] = S[S_H3 + r * LDC + kq] > 0.f ? gr * w4row[e] : 0.f;
    }
    __syncthreads();
    PH(9);   /* loss + dh3 */
    if (tid == 0) {
        float l = 0.f;
        for (int r = 0; r < ROWS; ++r) l += S[S_MISC + 2 * BE + r];
        if (FUSED)      // self-tagged, polled by reduction + Adam block 0
            __hip_atomic_store((gu64 *)(ws + ws_lossq(n_part)) + part, ((uint64_t)tag << 32) | (uint64_t)__float_as_uint(l), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else ws[ws_loss(n_part) + part] = l;
    }

    // ---- backward (all 8 waves)
    if (wave < 4) {   // dh2 = (dh3 W3) * [h2 > 0]
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = mma_a_lds<4>(S + S_DH3, LDC, bw.w3t, acc);
        const int c = wave * 16 + i;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rr = 4 * g + r;
            S[S_DH2 + rr * LDC + c] = S[S_H2 + rr * LDC + c] > 0.f ? acc[r] : 0.f;
        }
    } else {
        // dW3 = dh3^T h2 : 4 x 4 tiles, K = the 16 rows, computed transposed (h2^T dh3): wave 4 + mo takes output rows 16 mo .. and
        // ends up with four consecutive input columns per lane and tile
        const int mo = wave - 4;
        rows_gemm_fixed_b<4>(S + S_H2, LDC, S + S_DH3 + mo * 16, LDC, 0, 1, 4, [&](int nk, const f32x4 &acc) {
            pstore4(out, out_rsrc, O_W3 + (mo * 16 + i) * H + nk * 16 + 4 * g, acc, wt, keep);
        });
    }
    for (int e = tid; e < NA * H + NA + H; e += THREADS) {   // dW4, db4, db3
        float v = 0.f;
        if (e < NA * H) {
            const int a = e >> 6, kq = e & 63;
            for (int r = 0; r < ROWS; ++r)
                if (s_act[r >> 3] == a) v = fmaf(S[S_G + r], S[S_H3 + r * LDC + kq], v);
            pstore1(out + O_W4 + e, v, wt);
        } else if (e < NA * H + NA) {
            const int a = e - NA * H;
            for (int r = 0; r < ROWS; ++r)
                if (s_act[r >> 3] == a) v += S[S_G + r];
            pstore1(out + O_B4 + a, v, wt);
        } else {
            const int kq = e - NA * H - NA;
            for (int r = 0; r < ROWS; ++r) v += S[S_DH3 + r * LDC + kq];
            pstore1(out + O_B3 + kq, v, wt);
        }
    }
    __syncthreads();
    PH(10);  /* dh2, dW3, dW4 */
    {   // dx = dh2 W2 : 13 column tiles, first on every wave (the chain continues through them) ...
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = mma_a_lds<4>(S + S_DH2, LDC, bw.w2ta, acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) S[S_DX + (4 * g + r) * LDF + wave * 16 + i] = acc[r];
        if (wave + 8 < NT1) {
            f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
            acc2 = mma_a_lds<4>(S + S_DH2, LDC, bw.w2tb, acc2);
#pragma unroll
            for (int r = 0; r < 4; ++r) S[S_DX + (4 * g + r) * LDF + (wave + 8) * 16 + i] = acc2[r];
        }
    }
    {   // ... then dW2 = dh2^T x : 4 x 13 tiles, computed transposed (x^T dh2; the dh2 tile 16 (w & 3) .. shared by the wave's 6-7 tiles):
        // acc[r] = dW2[16 mo + i][16 nk + 4 g + r] -> one 16-byte store per lane and tile
        const int mo = wave & 3;
        rows_gemm_fixed_b<7>(S + S_X, LDF, S + S_DH2 + mo * 16, LDC, wave >> 2, 2, NT1, [&](int nk, const f32x4 &acc) {
            pstore4(out, out_rsrc, O_W2 + (mo * 16 + i) * F + nk * 16 + 4 * g, acc, wt, keep);
        });
    }
    if (tid < H) {
        float v = 0.f;
        for (int r = 0; r < ROWS; ++r) v += S[S_DH2 + r * LDC + tid];
        pstore1(out + O_B2 + tid, v, wt);
    }
    __syncthreads();
    PH(11);  /* dx, dW2 */
    // Hadamard product: d(features) = sum over the sample's 8 rows of dx * h1;  d(pre-h1) = dx * features * [h1 > 0]
    for (int t = tid; t < BE * F; t += THREADS) {
        const int be = t / F, o = t - be * F;
        const float f = S[S_FEAT + t];
        float df = 0.f;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int r = be * NQ + q;
            const float d = S[S_DX + r * LDF + o], h = S[S_H1 + r * LDF + o];
            df = fmaf(d, h, df);
            S[S_DX + r * LDF + o] = h > 0.f ? d * f : 0.f;
        }
        S[S_DF + t] = df;
    }
    __syncthreads();
    PH(12);  /* Hadamard */
    {   // dW1 = dh1^T cos : 13 x 4 tiles, computed transposed (cos^T dh1, the cos tile 16 (w & 3) .. shared by the wave's 6-7 tiles):
        // acc[r] = dW1[16 mo + i][16 nk + 4 g + r] -> one 16-byte store per lane and tile
        const int nk = wave & 3;
        rows_gemm_fixed_a<7>(S + S_C + nk * 16, LDC, S + S_DX, LDF, wave >> 2, 2, NT1, [&](int mo, const f32x4 &acc) {
            pstore4(out, out_rsrc, O_W1 + (mo * 16 + i) * NC + nk * 16 + 4 * g, acc, wt, keep);
        });
    }
    if (tid < F) {
        float v = 0.f;
        for (int r = 0; r < ROWS; ++r) v += S[S_DX + r * LDF + tid];
        pstore1(out + O_B1 + tid, v, wt);
    } else if (tid >= 256 && tid < 256 + F) {
        // encoders: dW = df^T obs, db = sum df
        const int o = tid - 256;
        const float d0 = S[S_DF + o], d1 = S[S_DF + F + o];
        const float *x0 = S + S_OBS, *x1 = S + S_OBS + 28;
        if (o < 16) {
            for (int kq = 0; kq < 2; ++kq) pstore1(out + O_VW + o * 2 + kq, fmaf(d1, x1[kq], d0 * x0[kq]), wt);
            pstore1(out + O_VB + o, d0 + d1, wt);
        } else if (o < 32) {
            for (int kq = 0; kq < 2; ++kq) pstore1(out + O_GW + (o - 16) * 2 + kq, fmaf(d1, x1[2 + kq], d0 * x0[2 + kq]), wt);
            pstore1(out + O_GB + o - 16, d0 + d1, wt);
        } else {
            pstore1(out + O_SB + o - 32, d0 + d1, wt);
        }
    }
    {   // sensor encoder dW [176 x 22] = 968 contiguous 16-byte pieces, one or two per thread (round 4: was 22 four-byte stores per lane at an
        // 88-byte stride -- 22 partial lines per lane, which write-through stores send to memory one by one); element (o, k) as before
        const float *x0 = S + S_OBS, *x1 = S + S_OBS + 28;
        for (int q = tid; q < 176 * 22 / 4; q += THREADS) {
            f32x4 v;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int e = 4 * q + c, o = 32 + e / 22, kq = e % 22;
                v[c] = fmaf(S[S_DF + F + o], x1[4 + kq], S[S_DF + o] * x0[4 + kq]);
            }
            pstore4(out, out_rsrc, O_SW + 4 * q, v, wt, keep);
        }
    }
    if (tid < P_PAD - P_TOTAL) pstore1(out + P_TOTAL + tid, 0.f, wt);   // row padding: read (as zeros) by the reduction's 16-byte loads
    PH(13);  /* dW1, encoder gradients issued */
    if (!two_roles && bid == 0) write_batch_copies(ba, base, batch);
    if (FUSED) {      // this workgroup's row (and loss partial) is final
        // Its stores are acknowledged -- by memory if they were write-through ones, by this XCD's L2 otherwise -- once vmcnt is 0; nothing of a
        // written-through row sits dirty in an L2: no __threadfence() (= an L2 write-back per workgroup, which made the first one-launch form 2.5 x slower)
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        PH(17);
#ifdef MN_TRAIN_PHASES
        if (tid_now() == 0) g_wgt[blockIdx.x][2] = wall_clock64();
#endif
        if (tid == 0) {
            if (wellplaced) *reinterpret_cast<volatile uint32_t *>(ws + ws_lflag(n_part) + 64 * (part & 7) + (part >> 3)) = tag;      // for this XCD's L2
            __hip_atomic_store((gu64 *)(ws + ws_done(n_part)) + part, ((uint64_t)tag << 32) | (wellplaced ? 0u : 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        group_reduce(ws, n_part, part, tag, tid);      // ... and this workgroup's share of its XCD group's row sum (self-tagged: nothing to wait for behind it)
        PH(18);
    }
#ifdef MN_TRAIN_PHASES
    __builtin_amdgcn_s_waitcnt(0);
    if (tid_now() == 0) g_wgt[blockIdx.x][1] = wall_clock64();
#endif
    }      // local workgroup
    }      // forward / backward of step k
    if (!FUSED) break;
    if ((is_target || is_extra) && k >= 1) {
        // ---- reduction + clip + Adam of step k - 1
        const int s = k - 1;
        const StepCtx sc = {epoch0 + (uint64_t)s, step0 + s, rs0, rs1 + (uint64_t)s, true, true};
        reduce_adam_body<2>(pb, n_phys, tail.n_virtual, ws, n_part, tail.grad, tail.loss_out + s, tail.rng_state, ba, tail.prefetch_next, tail.params, tail.m, tail.v,
                            tail.step, tail.lr, tail.b1, tail.b2, tail.eps, tail.max_norm, sc, XCHG ? tail.xa : nullptr, XCHG ? tail.xa_scale : 1.0f);
    }
    }      // k
}

// grad[p] = sum over workgroups of partial[wg][p].  One thread = one float4 column of one of RED_SEG contiguous segments of the
// partials: its (up to) n_part / 8 loads are all in flight before the first add (the round-2 kernel did four dependent rounds of
// eight), summed in index order; the eight segment sums are combined in a fixed order -> deterministic.  Also: this block's sum of
// squares of the reduced gradient (iqn_adam's norm), the loss (block 0), the generator's call counter and the hand-off epoch.
__global__ __launch_bounds__(RED_COLS *RED_SEG) void iqn_grad_reduce(float *__restrict__ ws, int n_part, float *__restrict__ grad,
                                                                       float *__restrict__ loss_out, uint64_t *__restrict__ rng_state,
                                                                       BatchArgs ba, int prefetch_next, uint64_t mailbox) {
    __shared__ float4 red[RED_SEG][RED_COLS];
    __shared__ float sq[RED_COLS];
    const int cx = tid_now() % RED_COLS, seg = tid_now() / RED_COLS;
    const int col = blockIdx.x * RED_COLS + cx;
    PH2(0, 0);
    constexpr int BT = RED_COLS * RED_SEG;
    // A workspace that mn_iqn_train_workspace_init never saw holds garbage tickets / epoch / tags: the counters would never advance and the
    // learner would silently repeat one batch.  Fail loudly instead: NaN loss, gradient untouched, Adam refuses too.
    if (*reinterpret_cast<const uint32_t *>(ws + ws_epoch(n_part) + 8) != WS_MAGIC) {
        if (blockIdx.x == 0 && tid_now() == 0) loss_out[0] = __builtin_nanf("");
        return;
    }
    float lpart = 0.f;      // block 0 sums the loss: its partials are requested now, summed at the end
    if (blockIdx.x == 0)
        for (int wq = tid_now(); wq < n_part; wq += BT) lpart += ws[ws_loss(n_part) + wq];
    // ---- staging of the NEXT step's batch (prefetch_next; blocks 1..): slot k of call counter + 1 -- ring row perm(k), its transition,
    // its 16 taus -- goes to a fixed address, so the next forward / backward launch starts with one round trip instead of three.
    // Requested first: the loads ride on the reduction's own memory latency.  The ring must not change before that launch uses it
    // (the caller passes use_staged only then); the tag written below ties the slots to {call counter, ring rows}.
    constexpr int SPB = BT / STG;      // staged slots per block
    const int batch = n_part * BE;
    int st_slot = -1, st_e = 0;
    float st_v = 0.f;
    auto staged_value = [&](uint64_t base_n, int slot, int e) -> float {
        const int64_t row = perm_row(base_n, (uint32_t)ba.ring_n, (uint32_t)slot);
        if (e < OBS) return ba.ring_s[row * OBS + e];
        if (e < 2 * OBS) return ba.ring_ns[row * OBS + e - OBS];
        if (e == 2 * OBS) return (float)ba.ring_a[row];
        if (e == 2 * OBS + 1) return ba.ring_r[row];
        if (e == 2 * OBS + 2) return ba.ring_d[row];
        if (e >= 56) return sample_tau(base_n, (e < 64 ? 0 : batch * NQ) + slot * NQ + (e & 7));
        return 0.f;
    };
    const bool stager = prefetch_next && rng_state && blockIdx.x >= 1 && (int)tid_now() / STG < SPB;
    if (stager) {      // this block's first SPB slots, requested now (more passes, for batches > 837, at the store below)
        st_e = tid_now() % STG;
        const int slot = ((int)blockIdx.x - 1) * SPB + tid_now() / STG;
        if (slot < batch) st_slot = slot;
    }
    const uint64_t base_n = stager ? mix64(rng_state[0] + 0x9E3779B97F4A7C15ull * (rng_state[1] + 2)) : 0;   // = sample_base after this step's increment (read once: see the ticket)
    const uint64_t epoch0 = *reinterpret_cast<const uint64_t *>(ws + ws_epoch(n_part));
    if (st_slot >= 0) st_v = staged_value(base_n, st_slot, st_e);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);      // segment seg = the rows w = seg (mod RED_SEG), ascending (see reduce_adam_body)
    if (col < N_COLS) {
        const float4 *src = reinterpret_cast<const float4 *>(ws) + col;
        for (int wb = seg; wb < n_part; wb += RED_SEG * RED_MAX_PER) {
            float4 t[RED_MAX_PER];
#pragma unroll
            for (int u = 0; u < RED_MAX_PER; ++u)
                t[u] = wb + RED_SEG * u < n_part ? src[(size_t)(wb + RED_SEG * u) * N_COLS] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < RED_MAX_PER; ++u) { v.x += t[u].x; v.y += t[u].y; v.z += t[u].z; v.w += t[u].w; }
        }
    }
    PH2(0, 1);   /* this wave's partial sums formed (all loads back) */
    red[seg][cx] = v;
    __syncthreads();
    PH2(0, 2);
    // this block's ticket, behind a barrier that its reads of the epoch and the generator's counter sit in front of, looked at only at the end (see reduce_adam_body)
    unsigned ticket_old = 0;
    if (tid_now() == 0) ticket_old = __hip_atomic_fetch_add(reinterpret_cast<unsigned *>(ws + ws_epoch(n_part) + 3), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (seg == 0) {
        float4 s = red[0][cx];
#pragma unroll
        for (int q = 1; q < RED_SEG; ++q) { s.x += red[q][cx].x; s.y += red[q][cx].y; s.z += red[q][cx].z; s.w += red[q][cx].w; }
        float ss = 0.f;
        if (col < N_COLS) {
            const int p = col * 4;
            if (p + 3 < P_TOTAL) *reinterpret_cast<float4 *>(grad + p) = s;   // grad is 16-byte aligned, p a multiple of 4
            else {
                const float e[4] = {s.x, s.y, s.z, s.w};
                for (int k = 0; k < 4; ++k)
                    if (p + k < P_TOTAL) grad[p + k] = e[k];
            }
            ss = sumsq4(s.x, s.y, s.z, s.w);   // padding columns are zeros
            // one-shot exchange of a shared learner (mn_xchg_*): the reduced gradient also goes to this rank's mailbox as self-tagged
            // 8-byte granules {step tag, value}, system scope -- the peers' gather kernels poll them, the data is the flag
            const uint64_t mb = mailbox ? mailbox : *reinterpret_cast<const uint64_t *>(ws + ws_epoch(n_part) + 10);      // (argument, or mn_xchg_attach's word)
            if (mb) {
                const uint32_t tag = xchg_tag(epoch0 + 1);     // the epoch this step ends with
                gu64 *dst = reinterpret_cast<gu64 *>(mb) + (size_t)(tag & 1u) * P_PAD + p;
                const float e[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    __hip_atomic_store(dst + k, ((uint64_t)tag << 32) | (uint64_t)__float_as_uint(e[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        sq[cx] = ss;
    }
    __syncthreads();
    if (tid_now() == 0) {
        float t = 0.f;
        for (int k = 0; k < RED_COLS; ++k) t += sq[k];
        ws[ws_sq(n_part) + blockIdx.x] = t;
    }
    if (blockIdx.x == 0) {      // the loss: every thread one partial (a single thread summing 128 dependent loads cost 11 us), fixed tree
        __shared__ float lw[BT / 64];
        float l = lpart;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) l += __shfl_xor(l, off, 64);
        if ((tid_now() & 63) == 0) lw[tid_now() >> 6] = l;
        __syncthreads();
        if (tid_now() == 0) {
            float t = 0.f;
            for (int k = 0; k < BT / 64; ++k) t += lw[k];
            *loss_out = t;
        }
    }
    PH2(0, 3);
    if (st_slot >= 0) ws[ws_stage(n_part) + (size_t)st_slot * STG + st_e] = st_v;
    if (stager && gridDim.x > 1)
        for (int j = (int)blockIdx.x - 1 + ((int)gridDim.x - 1); j * SPB < batch; j += (int)gridDim.x - 1) {
            const int slot = j * SPB + tid_now() / STG;
            if (slot < batch)
                ws[ws_stage(n_part) + (size_t)slot * STG + st_e] = staged_value(base_n, slot, st_e);
        }
    // the block that finishes LAST advances the generator's call counter (the batch of this step was drawn by iqn_train_fwdbwd; the
    // staging blocks above read the old value) and the hand-off epoch, and tags the staged batch
    if (tid_now() == 0) {
        unsigned *ticket = reinterpret_cast<unsigned *>(ws + ws_epoch(n_part) + 3);
        if (ticket_old == gridDim.x - 1) {
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *reinterpret_cast<uint64_t *>(ws + ws_epoch(n_part)) = epoch0 + 1;
            uint64_t *stg_tag = reinterpret_cast<uint64_t *>(ws + ws_epoch(n_part) + 4);
            if (rng_state) {
                const uint64_t c = rng_state[1] + 1;
                rng_state[1] = c;
                stg_tag[0] = c;
                stg_tag[1] = prefetch_next ? (uint64_t)ba.ring_n : 0;      // 0 rows: never a valid ring
            } else {
                stg_tag[1] = 0;
            }
        }
    }
}



// clip_grad_norm_(max_norm) (torch/nn/utils/clip_grad.py: coef = min(1, max_norm / (norm + 1e-6))) followed by
// torch.optim.Adam's update: m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
// p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps).  `step` lives on the device (hipGraph-capturable): this launch
// computes with t = *step + 1; the advanced value is stored by whichever block finishes last (a ticket counter in the workspace),
// so a block that is dispatched late -- another stream's kernel may hold the CUs -- still reads the old value.
// The sum of squares comes from per-block partial sums: iqn_grad_reduce's, or -- after an all-reduce rewrote the gradient --
// iqn_grad_sumsq's, which forms the same partial sums in the same order (so an exchange that returns the gradient unchanged, e.g.
// an all-reduce over one rank, leaves the step bit-identical).  The gradient is multiplied by grad_scale first (1 / world_size
// after an all-reduce(SUM); exactly 1.0f otherwise).
__global__ __launch_bounds__(RED_COLS) void iqn_grad_sumsq(const float *__restrict__ grad, float *__restrict__ blocksq, float grad_scale) {
    __shared__ float sq[RED_COLS];
    const int q = (blockIdx.x * RED_COLS + tid_now()) * 4;
    float e[4] = {0.f, 0.f, 0.f, 0.f};
    if (q + 3 < P_TOTAL) {
        const float4 x = *reinterpret_cast<const float4 *>(grad + q);
        e[0] = x.x; e[1] = x.y; e[2] = x.z; e[3] = x.w;
    } else {
        for (int k = 0; k < 4; ++k)
            if (q + k < P_TOTAL) e[k] = grad[q + k];
    }
    for (int k = 0; k < 4; ++k) e[k] *= grad_scale;
    sq[tid_now()] = sumsq4(e[0], e[1], e[2], e[3]);
    __syncthreads();
    if (tid_now() == 0) {
        float t = 0.f;
        for (int k = 0; k < RED_COLS; ++k) t += sq[k];
        blocksq[blockIdx.x] = t;
    }
}

// One-shot gradient exchange of a shared learner (one 143 KB bucket, latency-bound): every rank's reduction kernel has
// published its reduced gradient into its own mailbox (above); this kernel reads the mailboxes of ALL ranks -- its own and, through
// IPC-mapped pointers, the peers' -- and forms grad = sum over ranks IN RANK ORDER (same bits on every rank; for two ranks the same sum
// an all-reduce gives), plus iqn_grad_sumsq's per-block sums of squares of grad_scale * grad in the same shape and order, so that
// mn_iqn_train_adam needs no second pass.  No collective launch, no barrier: a granule carries its step tag, a reader polls until the tag
// is the current step's (bounded: ~2 s of the 100 MHz counter, then the status word is raised and the step continues with what is there --
// a wait can never hang the device).  Two slots alternate with the step parity: a rank publishes step k + 2 into slot k & 1 only after its
// gather of step k + 1, which needed every peer's step k + 1, which every peer published after ITS gather of step k.
__global__ __launch_bounds__(RED_COLS) void iqn_grad_gather(XchgPeers peers, int world, const float *__restrict__ ws, int n_part,
                                                            float *__restrict__ grad, float *__restrict__ blocksq, float grad_scale,
                                                            unsigned *__restrict__ status, uint64_t bound) {
    __shared__ float sq[RED_COLS];
    const int q = (blockIdx.x * RED_COLS + tid_now()) * 4;
    const uint32_t tag = xchg_tag(*reinterpret_cast<const uint64_t *>(ws + ws_epoch(n_part)));      // (the reduction kernel advanced the epoch)
    float e[4] = {0.f, 0.f, 0.f, 0.f};
    if (q < P_PAD) {
        bool late = false;
        for (int r = 0; r < world; ++r) {      // (rank after rank: this kernel is the measurement / fallback form; the fused launch below requests all ranks at once)
            const gu64 *src = peers.mb[r] + (size_t)(tag & 1u) * P_PAD + q;
            uint64_t x[4];
            const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    x[k] = __hip_atomic_load(src + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    ok = ok && (uint32_t)(x[k] >> 32) == tag;
                }
                if (ok) break;
                if (__builtin_amdgcn_s_memrealtime() - t0 > bound) { late = true; break; }
                __builtin_amdgcn_s_sleep(8);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) e[k] += __uint_as_float((uint32_t)x[k]);
        }
        if (late) {      // a peer's granules never came: the step must not look valid (mn_xchg_status counts it; the gradient and with it the norm are NaN)
            atomicAdd(status, 1u);
            e[0] = __builtin_nanf("");
        }
        if (q + 3 < P_TOTAL) *reinterpret_cast<float4 *>(grad + q) = make_float4(e[0], e[1], e[2], e[3]);
        else
            for (int k = 0; k < 4; ++k)
                if (q + k < P_TOTAL) grad[q + k] = e[k];
                else e[k] = 0.f;
    }
    for (int k = 0; k < 4; ++k) e[k] *= grad_scale;
    sq[tid_now()] = sumsq4(e[0], e[1], e[2], e[3]);
    __syncthreads();
    if (tid_now() == 0) {
        float t = 0.f;
        for (int k = 0; k < RED_COLS; ++k) t += sq[k];
        blocksq[blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(256) void iqn_adam(float *__restrict__ params, float *__restrict__ grad, float *__restrict__ m,
                                                float *__restrict__ v, const float *__restrict__ blocksq, int32_t *__restrict__ step,
                                                unsigned *__restrict__ ticket, double lr, double b1, double b2, double eps_d,
                                                double max_norm_d, float grad_scale) {
    __shared__ float red[4];
    __shared__ float s_bc[2];
    const int p = blockIdx.x * 256 + tid_now();
    PH2(1, 0);
    if (ticket[6] != WS_MAGIC) return;      // (ticket = epoch block + 2) workspace not initialised: see iqn_grad_reduce
    // this thread's operands first: their latency overlaps the norm
    float gq = 0.f, mp = 0.f, vp = 0.f, pp = 0.f;
    if (p < P_TOTAL) { gq = grad[p] * grad_scale; mp = m[p]; vp = v[p]; pp = params[p]; }
    float part = 0.f;
    for (int c = tid_now(); c < N_RED; c += 256) part += blocksq[c];
    int t_step = 0;
    unsigned ticket_old = 0;
    if (tid_now() == 255) {
        t_step = *step + 1;
        // python-float (double) scalars of torch's Adam, rounded to float32 where the tensor kernels consume them
        s_bc[0] = (float)(lr / (1.0 - pow(b1, (double)t_step)));
        // this block's ticket, taken once its read of the step counter has been consumed and looked at only at the end: the atomic's round trip (~1 us)
        // runs under the second pow and the norm instead of behind the parameter stores
        ticket_old = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_bc[1] = (float)sqrt(1.0 - pow(b2, (double)t_step));
    }
    // wave sums in a fixed order (DPP row / bank shuffles), then the four wave sums
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
    if ((tid_now() & 63) == 0) red[tid_now() >> 6] = part;
    __syncthreads();
    PH2(1, 1);   /* norm partials summed, bias corrections computed */
    const float sumsq = (red[0] + red[1]) + (red[2] + red[3]);
    const float norm = sqrtf(sumsq);
    const float coef = fminf((float)max_norm_d / (norm + 1e-6f), 1.f);
    const float step_size = s_bc[0], bc2_sqrt = s_bc[1];
    const float w1 = (float)(1.0 - b1), b2f = (float)b2, w2 = (float)(1.0 - b2), eps = (float)eps_d;
    const bool bad = !(sumsq == sumsq);      // NaN norm: the exchange in front of this launch timed out (iqn_grad_gather) -- nothing is updated, the status word counts it
    if (bad && blockIdx.x == 0 && tid_now() == 0) atomicAdd(ticket + 10, 1u);      // (ticket = epoch block + 2; + 10 = the workspace's status word)
    if (p < P_TOTAL && bad) grad[p] = __builtin_nanf("");
    if (p < P_TOTAL && !bad) {
        gq *= coef;
        grad[p] = gq;
        adam_update(gq, mp, vp, pp, w1, b2f, w2, step_size, bc2_sqrt, eps);
        m[p] = mp;
        v[p] = vp;
        params[p] = pp;
    }
    PH2(1, 2);
    // the block with the LAST ticket stores the advanced counter: every block's thread 255 read it before taking its ticket
    if (tid_now() == 255 && ticket_old == gridDim.x - 1) {
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *step = t_step;
    }
}

// ReplayBuffer.sample (replay_buffer.py:42-47: random.sample = uniform WITHOUT replacement) plus the 2 x batch x 8
// tau draws of the step (model.py:149), as a stand-alone launch: the batch iqn_train_fwdbwd draws for itself when it is given
// the generator state instead of index / tau buffers.  Counter-based: state = {seed, call counter} on the device, advanced here.
__global__ __launch_bounds__(256) void iqn_sample_kernel(int64_t n, int batch, uint64_t *__restrict__ state,
                                                         int64_t *__restrict__ idx, float *__restrict__ taus, int n_taus) {
    const uint64_t ctr = state[1], base = sample_base(state);
    __syncthreads();
    for (int e = tid_now(); e < n_taus; e += 256) taus[e] = sample_tau(base, e);
    for (int k = tid_now(); k < batch; k += 256) idx[k] = perm_row(base, (uint32_t)n, (uint32_t)k);
    if (tid_now() == 0) state[1] = ctr + 1;
}

int g_train_mode = MODE_TWO_ROLES;

}  // namespace

#include "iqn_train_host.h"      // launch plan, launchers, C-ABI entry points, mn_xchg_*
