// iqn_train.hip -- fused IQN gradient step for gfx950 (MI355X): forward of the target and the local network,
// quantile-Huber TD loss, backward, gradient-norm clip and Adam in four kernels.
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
// from a hipGraph it takes ~540 us, all launch latency -- the arithmetic is 0.5 GFLOP.  Here
//   iqn_train_fwdbwd  one 512-thread workgroup per 2 batch elements (= 16 (sample, tau) rows = one MFMA M-tile):
//                     gathers its transitions from the ring, runs the target forward on four waves and the local
//                     forward on the other four at the same time, then the loss gradient and the whole backward
//                     out of LDS on all eight, and writes its partial parameter gradient [35 785] to HBM;
//   iqn_grad_reduce   sums the partials in a fixed order (deterministic, no float atomics) -> flat gradient, loss;
//   iqn_sumsq         per-block sums of squares of the (possibly all-reduced) gradient, advances the step counter;
//   iqn_adam          global norm, clip coefficient, Adam update (torch.optim.Adam arithmetic), one flat pass.
// Between the last two the caller may all-reduce the flat gradient (shared learner over RCCL).
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

#include <mutex>

#include "marinenav_hip.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

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
constexpr int LDC = 68;          // row stride of the 64-wide LDS activations (16-byte aligned rows, bank skew)
constexpr int LDF = 212;         // row stride of the 208-wide LDS activations
// LDS layout (floats); every 2-D block starts 16-byte aligned.  Local-network pass (kept for the backward):
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
constexpr int S_OBS = S_DF + BE * F;         // [2 which][2][28]  states / next_states
constexpr int S_Q = S_OBS + 2 * BE * 28;     // [16][12]   quantile values
constexpr int S_QT = S_Q + ROWS * 12;        // [16] TD targets
constexpr int S_G = S_QT + ROWS;             // [16] dL/dQ_expected
constexpr int S_TAU = S_G + ROWS;            // [2][16]    0: target taus, 1: local taus
constexpr int S_MISC = S_TAU + 2 * ROWS;     // rew[2], done[2], loss terms[16]
// target-network pass (runs concurrently on the other four waves; nothing of it is kept but q)
constexpr int T_C = (S_MISC + 2 * BE + ROWS + 3) / 4 * 4;
constexpr int T_X = T_C + ROWS * LDC;
constexpr int T_H2 = T_X + ROWS * LDF;
constexpr int T_H3 = T_H2 + ROWS * LDC;
constexpr int T_FEAT = T_H3 + ROWS * LDC;
constexpr int T_Q = T_FEAT + BE * F;
constexpr int S_TOTAL = T_Q + ROWS * 12;
constexpr int LDS_BYTES = S_TOTAL * 4;
constexpr int THREADS = 512;

struct PassBufs { float *c, *h1, *x, *h2, *h3, *feat, *q; };

// One 16x16 output tile of C = A . B over K (a multiple of 16; compile-time so that every operand load of the tile is
// issued before the first MFMA), accumulated into `acc`.
//   AK: A element (m, k) lives at A[m * lda + k] (k contiguous, 16-byte aligned rows);  else at A[k * lda + m].
//   BK: B element (k, n) lives at B[n * ldb + k] (k contiguous);                         else at B[k * ldb + n].
// Rows n >= n_valid of a k-contiguous B are read as zeros (the 9-row output layer).
template <bool AK, bool BK, int K>
__device__ __forceinline__ f32x4 tile_gemm(const float *A, int lda, const float *B, int ldb, f32x4 acc, int n_valid = 16) {
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
    constexpr int NB = K / 16;
    float a[NB][4], b[NB][4];
#pragma unroll
    for (int kk = 0; kk < NB; ++kk) {
        const int kb = kk * 16 + 4 * g;
        if (AK) {
            const float4 v = *reinterpret_cast<const float4 *>(A + i * lda + kb);
            a[kk][0] = v.x; a[kk][1] = v.y; a[kk][2] = v.z; a[kk][3] = v.w;
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) a[kk][s] = A[(kb + s) * lda + i];
        }
        if (BK) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < n_valid) v = *reinterpret_cast<const float4 *>(B + i * ldb + kb);
            b[kk][0] = v.x; b[kk][1] = v.y; b[kk][2] = v.z; b[kk][3] = v.w;
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) b[kk][s] = B[(kb + s) * ldb + i];
        }
    }
#pragma unroll
    for (int kk = 0; kk < NB; ++kk)
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kk][s], b[kk][s], acc, 0, 0, 0);
    return acc;
}

__device__ __forceinline__ uint64_t mix64(uint64_t x) {   // splitmix64 finaliser (Steele, Lea, Flood 2014)
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    return x ^ (x >> 31);
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
// The batch's `batch` distinct ring rows into val[] (LDS), by ALL threads of the workgroup (any workgroup size: a slot's draws
// depend only on the slot and its attempt number, and every round redraws all clashing slots at once).  Every slot draws
// uniformly from [0, n); a slot whose value is also held by a lower slot redraws, until all are distinct.  A slot only ever
// rejects values that end up owned by a lower slot, so slot k's value is uniform over what slots < k left: exactly sequential
// sampling without replacement (replay_buffer.py:42-47, random.sample).
template <int NTHREADS>
__device__ __forceinline__ void sample_rows(int *val, int *flag, int64_t n, int batch, uint64_t base) {
    constexpr int PER = MAX_BATCH / NTHREADS;
    const int padded = (batch + 3) & ~3;
    int attempt[PER];
    for (int k = threadIdx.x, q = 0; k < padded; k += NTHREADS, ++q) {
        attempt[q] = 0;
        const uint64_t x = mix64(base + 0xA24BAED4963EE407ull * (uint64_t)(k + 1));
        val[k] = k < batch ? (int)__umul64hi(x, (uint64_t)n) : -1;
        flag[k] = 0;
    }
    __syncthreads();
    for (;;) {
        // does a LOWER slot hold slot k's value?  Two work items per slot, each scanning half of [0, k) four candidates per LDS
        // read; only the last, partial group needs index masks.  (One item per slot scanning all of [0, k) with masks on every
        // candidate was 2 us per pass -- VALU-bound -- in every one of the 128 workgroups.)
        for (int w = threadIdx.x; w < 2 * batch; w += NTHREADS) {
            const int k = w >> 1, part = w & 1, v = val[k];
            const int jmax = k & ~3, mid = (jmax >> 1) & ~3;
            const int lo = part ? mid : 0, hi = part ? jmax : mid;
            bool c = false;
            for (int j = lo; j < hi; j += 4) {
                const int4 q4 = *reinterpret_cast<const int4 *>(&val[j]);
                c |= (q4.x == v) | (q4.y == v) | (q4.z == v) | (q4.w == v);
            }
            if (part) {
                const int4 q4 = *reinterpret_cast<const int4 *>(&val[jmax]);
                c |= ((q4.x == v) & (jmax < k)) | ((q4.y == v) & (jmax + 1 < k)) | ((q4.z == v) & (jmax + 2 < k));
            }
            if (c) flag[k] = 1;
        }
        __syncthreads();
        int clash = 0;
        bool redo[PER];
        for (int k = threadIdx.x, q = 0; k < batch; k += NTHREADS, ++q) {
            redo[q] = flag[k] != 0;
            clash |= redo[q];
        }
        if (!__syncthreads_or(clash)) break;
        for (int k = threadIdx.x, q = 0; k < batch; k += NTHREADS, ++q) {
            if (!redo[q]) continue;
            ++attempt[q];
            const uint64_t x = mix64(base + 0xA24BAED4963EE407ull * (uint64_t)(k + 1) + 0x9FB21C651E98DF25ull * (uint64_t)attempt[q]);
            val[k] = (int)__umul64hi(x, (uint64_t)n);
            flag[k] = 0;
        }
        __syncthreads();
    }
}

// model.py:160-186 for the 16 rows of this workgroup, executed by ONE HALF of the workgroup (4 waves; the other half
// runs the other network at the same time, so every __syncthreads() here is reached by all 512 threads):
// `obs` [2][28], `tau` [16] in LDS, parameters `P` in HBM/L2.  Leaves cos, (h1,) x, h2, h3, features and q in LDS.
__device__ __forceinline__ void forward_pass(const PassBufs &Bf, const float *__restrict__ P, const float *obs, const float *tau) {
    const int tid = threadIdx.x & 255, wave = tid >> 6, lane = tid & 63, i = lane & 15, g = lane >> 4;
    // encoders: three linear maps, no activation (model.py:170-173)
    for (int t = tid; t < BE * F; t += 256) {
        const int be = t / F, o = t - be * F;
        float acc;
        if (o < 32) {
            const int e = o < 16 ? 0 : 1, oo = o - 16 * e;
            const float *w = P + (e ? O_GW : O_VW) + oo * 2, *in = obs + be * 28 + 2 * e;
            acc = fmaf(w[1], in[1], fmaf(w[0], in[0], P[(e ? O_GB : O_VB) + oo]));
        } else {
            const float *w = P + O_SW + (o - 32) * 22, *in = obs + be * 28 + 4;
            acc = P[O_SB + o - 32];
#pragma unroll
            for (int k = 0; k < 22; ++k) acc = fmaf(w[k], in[k], acc);
        }
        Bf.feat[t] = acc;
    }
    // cos(tau * pi * i), pis = float32(pi * i) (model.py:130,149-155)
    for (int t = tid; t < ROWS * NC; t += 256) {
        const int r = t >> 6, c = t & 63;
        Bf.c[r * LDC + c] = cosf(tau[r] * (float)(M_PI * (double)c));
    }
    __syncthreads();
    // h1 = relu(cos W1^T + b1); x = h1 * features   (13 column tiles over 4 waves)
    for (int tile = wave; tile < F / 16; tile += 4) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = tile_gemm<true, true, NC>(Bf.c, LDC, P + O_W1 + tile * 16 * NC, NC, acc);
        const int o = tile * 16 + i;
        const float bias = P[O_B1 + o];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * g + r;
            const float v = fmaxf(acc[r] + bias, 0.f);
            if (Bf.h1) Bf.h1[row * LDF + o] = v;
            Bf.x[row * LDF + o] = v * Bf.feat[(row >> 3) * F + o];
        }
    }
    __syncthreads();
    {   // h2 = relu(x W2^T + b2): one 16-column tile per wave, K = 208
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = tile_gemm<true, true, F>(Bf.x, LDF, P + O_W2 + wave * 16 * F, F, acc);
        const int o = wave * 16 + i;
        const float bias = P[O_B2 + o];
#pragma unroll
        for (int r = 0; r < 4; ++r) Bf.h2[(4 * g + r) * LDC + o] = fmaxf(acc[r] + bias, 0.f);
    }
    __syncthreads();
    {   // h3 = relu(h2 W3^T + b3)
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = tile_gemm<true, true, H>(Bf.h2, LDC, P + O_W3 + wave * 16 * H, H, acc);
        const int o = wave * 16 + i;
        const float bias = P[O_B3 + o];
#pragma unroll
        for (int r = 0; r < 4; ++r) Bf.h3[(4 * g + r) * LDC + o] = fmaxf(acc[r] + bias, 0.f);
    }
    __syncthreads();
    if (wave == 0) {   // q = h3 W4^T + b4 (9 of 16 columns)
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = tile_gemm<true, true, H>(Bf.h3, LDC, P + O_W4, H, acc, NA);
        if (i < NA) {
            const float bias = P[O_B4 + i];
#pragma unroll
            for (int r = 0; r < 4; ++r) Bf.q[(4 * g + r) * 12 + i] = acc[r] + bias;
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(THREADS) void iqn_train_fwdbwd(const float *__restrict__ ring_s, const float *__restrict__ ring_ns,
                                                            const int64_t *__restrict__ ring_a, const float *__restrict__ ring_r,
                                                            const float *__restrict__ ring_d, const int64_t *__restrict__ idx,
                                                            const float *__restrict__ taus_t, const float *__restrict__ taus_l,
                                                            const float *__restrict__ PL, const float *__restrict__ PT,
                                                            float *__restrict__ partial, float *__restrict__ loss_partial,
                                                            int batch, float gamma, const uint64_t *__restrict__ rng_state,
                                                            int64_t ring_n, int64_t *__restrict__ idx_out, float *__restrict__ taus_out) {
    extern __shared__ __align__(16) float S[];
    __shared__ int s_act[BE];
    __shared__ int64_t s_row[BE];
    __shared__ __align__(16) int s_val[MAX_BATCH];
    __shared__ int s_flag[MAX_BATCH];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 15, g = lane >> 4;
    const int b0 = blockIdx.x * BE;
    float *out = partial + (size_t)blockIdx.x * P_TOTAL;

    // The batch: either given (idx, taus_t, taus_l), or drawn here from the generator state -- EVERY workgroup runs the whole
    // (cheap, deterministic) draw of the batch's rows and keeps its own two, which saves the separate sampling launch (8-11 us
    // of a 57 us gradient step).  The state's call counter is advanced by iqn_grad_reduce, after all workgroups have read it.
    if (rng_state) {
        const uint64_t base = sample_base(rng_state);
        sample_rows<THREADS>(s_val, s_flag, ring_n, batch, base);
        if (tid < BE) s_row[tid] = s_val[b0 + tid];
        if (tid < 2 * ROWS) S[S_TAU + tid] = sample_tau(base, (tid < ROWS ? 0 : batch * NQ) + b0 * NQ + (tid & (ROWS - 1)));
        if (blockIdx.x == 0) {      // the caller's copies (inspection, tests)
            if (idx_out) for (int k = tid; k < batch; k += THREADS) idx_out[k] = s_val[k];
            if (taus_out) for (int e = tid; e < 2 * batch * NQ; e += THREADS) taus_out[e] = sample_tau(base, e);
        }
    } else {
        if (tid < BE) s_row[tid] = idx[b0 + tid];
        if (tid < 2 * ROWS) S[S_TAU + tid] = (tid < ROWS ? taus_t : taus_l)[b0 * NQ + (tid & (ROWS - 1))];
    }
    __syncthreads();
    // gather this workgroup's transitions from the replay ring (replay_buffer.py:42-57)
    for (int t = tid; t < 2 * BE * OBS; t += THREADS) {
        const int which = t / (BE * OBS), rem = t - which * (BE * OBS), be = rem / OBS, k = rem - be * OBS;
        S[S_OBS + which * (BE * 28) + be * 28 + k] = (which ? ring_ns : ring_s)[s_row[be] * OBS + k];
    }
    if (tid < BE) {
        const int64_t row = s_row[tid];
        s_act[tid] = (int)ring_a[row];
        S[S_MISC + tid] = ring_r[row];
        S[S_MISC + BE + tid] = ring_d[row];
    }
    __syncthreads();

    // ---- waves 0-3: local network on states; waves 4-7: target network on next_states (agent.py:279-286).
    // ONE call from uniform control flow -- the half a thread belongs to only selects its buffers / parameters / inputs,
    // so all 512 threads reach the same five barrier sites inside forward_pass.
    {
        const bool tgt = tid >= 256;
        const PassBufs B = {S + (tgt ? T_C : S_C), tgt ? nullptr : S + S_H1, S + (tgt ? T_X : S_X), S + (tgt ? T_H2 : S_H2),
                            S + (tgt ? T_H3 : S_H3), S + (tgt ? T_FEAT : S_FEAT), S + (tgt ? T_Q : S_Q)};
        forward_pass(B, tgt ? PT : PL, S + S_OBS + (tgt ? BE * 28 : 0), S + S_TAU + (tgt ? 0 : ROWS));
    }
    // TD targets: r + gamma * max_a Q_target(next, tau_j) * (1 - done)   (agent.py:281-283)
    if (tid < ROWS) {
        float m = S[T_Q + tid * 12];
        for (int a = 1; a < NA; ++a) m = fmaxf(m, S[T_Q + tid * 12 + a]);
        const int be = tid >> 3;
        S[S_QT + tid] = S[S_MISC + be] + gamma * m * (1.f - S[S_MISC + BE + be]);
    }
    __syncthreads();

    // ---- quantile-Huber loss and dL/dQ_expected (agent.py:289-295, 401-407)
    if (tid < ROWS) {
        const int be = tid >> 3;
        const float qe = S[S_Q + tid * 12 + s_act[be]], tau = S[S_TAU + ROWS + tid];
        float lsum = 0.f, gsum = 0.f;
        for (int j = 0; j < NQ; ++j) {
            const float td = S[S_QT + be * NQ + j] - qe, ad = fabsf(td);
            const float hub = ad <= 1.f ? 0.5f * td * td : ad - 0.5f;
            const float w = fabsf(tau - (td < 0.f ? 1.f : 0.f));
            lsum += w * hub;
            gsum += w * fminf(fmaxf(td, -1.f), 1.f);
        }
        const float scale = 1.f / (float)(batch * NQ);
        S[S_G + tid] = -gsum * scale;
        S[S_MISC + 2 * BE + tid] = lsum * scale;
    }
    __syncthreads();
    if (tid == 0) {
        float l = 0.f;
        for (int r = 0; r < ROWS; ++r) l += S[S_MISC + 2 * BE + r];
        loss_partial[blockIdx.x] = l;
    }

    // ---- backward (all 8 waves).  Output layer: only the taken action's row carries gradient.
    for (int t = tid; t < ROWS * H; t += THREADS) {
        const int r = t >> 6, k = t & 63;
        S[S_DH3 + r * LDC + k] = S[S_H3 + r * LDC + k] > 0.f ? S[S_G + r] * PL[O_W4 + s_act[r >> 3] * H + k] : 0.f;
    }
    for (int e = tid; e < NA * H + NA; e += THREADS) {
        float v = 0.f;
        if (e < NA * H) {
            const int a = e >> 6, k = e & 63;
            for (int r = 0; r < ROWS; ++r)
                if (s_act[r >> 3] == a) v += S[S_G + r] * S[S_H3 + r * LDC + k];
            out[O_W4 + e] = v;
        } else {
            const int a = e - NA * H;
            for (int r = 0; r < ROWS; ++r)
                if (s_act[r >> 3] == a) v += S[S_G + r];
            out[O_B4 + a] = v;
        }
    }
    __syncthreads();
    if (wave < 4) {   // dh2 = (dh3 W3) * [h2 > 0]
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = tile_gemm<true, false, H>(S + S_DH3, LDC, PL + O_W3 + wave * 16, H, acc);
        const int c = wave * 16 + i;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * g + r;
            S[S_DH2 + row * LDC + c] = S[S_H2 + row * LDC + c] > 0.f ? acc[r] : 0.f;
        }
    }
    for (int tt = wave; tt < 16; tt += 8) {   // dW3 = dh3^T h2 : 4 x 4 tiles, K = the 16 rows
        const int mo = tt >> 2, nk = tt & 3;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = tile_gemm<false, false, ROWS>(S + S_DH3 + mo * 16, LDC, S + S_H2 + nk * 16, LDC, acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) out[O_W3 + (mo * 16 + 4 * g + r) * H + nk * 16 + i] = acc[r];
    }
    if (tid < H) {
        float v = 0.f;
        for (int r = 0; r < ROWS; ++r) v += S[S_DH3 + r * LDC + tid];
        out[O_B3 + tid] = v;
    }
    __syncthreads();
    for (int job = wave; job < 5 * (F / 16); job += 8) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (job < F / 16) {   // dx = dh2 W2 : 13 column tiles
            acc = tile_gemm<true, false, H>(S + S_DH2, LDC, PL + O_W2 + job * 16, F, acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) S[S_DX + (4 * g + r) * LDF + job * 16 + i] = acc[r];
        } else {              // dW2 = dh2^T x : 4 x 13 tiles
            const int tt = job - F / 16, mo = tt / (F / 16), nk = tt - mo * (F / 16);
            acc = tile_gemm<false, false, ROWS>(S + S_DH2 + mo * 16, LDC, S + S_X + nk * 16, LDF, acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) out[O_W2 + (mo * 16 + 4 * g + r) * F + nk * 16 + i] = acc[r];
        }
    }
    if (tid < H) {
        float v = 0.f;
        for (int r = 0; r < ROWS; ++r) v += S[S_DH2 + r * LDC + tid];
        out[O_B2 + tid] = v;
    }
    __syncthreads();
    // Hadamard product: d(features) = sum over the sample's 8 rows of dx * h1;  d(pre-h1) = dx * features * [h1 > 0]
    for (int t = tid; t < BE * F; t += THREADS) {
        const int be = t / F, o = t - be * F;
        const float f = S[S_FEAT + t];
        float df = 0.f;
        for (int q = 0; q < NQ; ++q) {
            const int r = be * NQ + q;
            const float d = S[S_DX + r * LDF + o], h = S[S_H1 + r * LDF + o];
            df = fmaf(d, h, df);
            S[S_DX + r * LDF + o] = h > 0.f ? d * f : 0.f;
        }
        S[S_DF + t] = df;
    }
    __syncthreads();
    for (int tt = wave; tt < 4 * (F / 16); tt += 8) {   // dW1 = dh1^T cos : 13 x 4 tiles
        const int mo = tt >> 2, nk = tt & 3;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = tile_gemm<false, false, ROWS>(S + S_DX + mo * 16, LDF, S + S_C + nk * 16, LDC, acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) out[O_W1 + (mo * 16 + 4 * g + r) * NC + nk * 16 + i] = acc[r];
    }
    if (tid < F) {
        float v = 0.f;
        for (int r = 0; r < ROWS; ++r) v += S[S_DX + r * LDF + tid];
        out[O_B1 + tid] = v;
    } else if (tid >= 256 && tid < 256 + F) {
        // encoders: dW = df^T obs, db = sum df
        const int o = tid - 256;
        const float d0 = S[S_DF + o], d1 = S[S_DF + F + o];
        const float *x0 = S + S_OBS, *x1 = S + S_OBS + 28;
        if (o < 16) {
            for (int k = 0; k < 2; ++k) out[O_VW + o * 2 + k] = d0 * x0[k] + d1 * x1[k];
            out[O_VB + o] = d0 + d1;
        } else if (o < 32) {
            for (int k = 0; k < 2; ++k) out[O_GW + (o - 16) * 2 + k] = d0 * x0[2 + k] + d1 * x1[2 + k];
            out[O_GB + o - 16] = d0 + d1;
        } else {
            for (int k = 0; k < 22; ++k) out[O_SW + (o - 32) * 22 + k] = d0 * x0[4 + k] + d1 * x1[4 + k];
            out[O_SB + o - 32] = d0 + d1;
        }
    }
}

// grad[p] = sum over workgroups of partial[wg][p]: four quarter sums (one per thread row, partials in index order,
// 8 loads in flight) combined in a fixed order -> deterministic.  Block 0 also sums the loss.
__global__ __launch_bounds__(1024) void iqn_grad_reduce(const float *__restrict__ partial, const float *__restrict__ loss_partial,
                                                        int n_part, float *__restrict__ grad, float *__restrict__ loss_out,
                                                        uint64_t *__restrict__ rng_state) {
    __shared__ float red[4][256];
    if (rng_state && blockIdx.x == 0 && threadIdx.x == 0) rng_state[1] += 1;   // the batch of this step was drawn by iqn_train_fwdbwd
    const int px = threadIdx.x & 255, seg = threadIdx.x >> 8;
    const int p = blockIdx.x * 256 + px;
    const int per = (n_part + 3) / 4, w0 = seg * per, w1 = min(n_part, w0 + per);
    float v = 0.f;
    if (p < P_TOTAL) {
#pragma unroll 8
        for (int w = w0; w < w1; ++w) v += partial[(size_t)w * P_TOTAL + p];
    }
    red[seg][px] = v;
    __syncthreads();
    if (seg == 0 && p < P_TOTAL) grad[p] = ((red[0][px] + red[1][px]) + red[2][px]) + red[3][px];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        float l = 0.f;
        for (int w = 0; w < n_part; ++w) l += loss_partial[w];
        *loss_out = l;
    }
}

constexpr int N_SQ = (P_TOTAL + 255) / 256;   // 140 blocks of 256 parameters

// Per-block sum of squares of the (possibly all-reduced) gradient; block 0 advances the optimizer step counter.
__global__ __launch_bounds__(256) void iqn_sumsq(const float *__restrict__ grad, float *__restrict__ blocksq, int32_t *__restrict__ step) {
    __shared__ float red[256];
    const int p = blockIdx.x * 256 + threadIdx.x;
    const float gq = p < P_TOTAL ? grad[p] : 0.f;
    red[threadIdx.x] = gq * gq;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        blocksq[blockIdx.x] = red[0];
        if (blockIdx.x == 0) *step += 1;
    }
}

// clip_grad_norm_(max_norm) (torch/nn/utils/clip_grad.py: coef = min(1, max_norm / (norm + 1e-6))) followed by
// torch.optim.Adam's update: m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
// p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps).  `step` lives on the device (hipGraph-capturable).
__global__ __launch_bounds__(256) void iqn_adam(float *__restrict__ params, float *__restrict__ grad, float *__restrict__ m,
                                                float *__restrict__ v, const float *__restrict__ blocksq,
                                                const int32_t *__restrict__ step, double lr, double b1, double b2,
                                                double eps_d, double max_norm_d) {
    __shared__ float red[256];
    red[threadIdx.x] = threadIdx.x < N_SQ ? blocksq[threadIdx.x] : 0.f;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    const float norm = sqrtf(red[0]);
    const float coef = fminf((float)max_norm_d / (norm + 1e-6f), 1.f);
    const int t = *step;   // already advanced by iqn_sumsq
    // python-float (double) scalars of torch's Adam, rounded to float32 where the tensor kernels consume them
    const float step_size = (float)(lr / (1.0 - pow(b1, (double)t)));
    const float bc2_sqrt = (float)sqrt(1.0 - pow(b2, (double)t));
    const float w1 = (float)(1.0 - b1), b2f = (float)b2, w2 = (float)(1.0 - b2), eps = (float)eps_d;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p < P_TOTAL) {
        const float gq = grad[p] * coef;
        grad[p] = gq;
        const float mm = m[p] + (gq - m[p]) * w1;                 // lerp, as torch's _single_tensor_adam
        const float vv = v[p] * b2f + w2 * (gq * gq);
        m[p] = mm;
        v[p] = vv;
        params[p] -= step_size * (mm / (sqrtf(vv) / bc2_sqrt + eps));
    }
}

// ReplayBuffer.sample (replay_buffer.py:42-47: random.sample = uniform WITHOUT replacement) plus the 2 x batch x 8
// tau draws of the step (model.py:149), one workgroup (the stand-alone form of what iqn_train_fwdbwd does in its prologue when
// it is given the generator state instead of index / tau buffers).  Counter-based RNG: value = mix64(seed, call counter, slot,
// attempt); state = {seed, counter} on the device, advanced by the kernel (so the launch arguments never change).
__global__ __launch_bounds__(256) void iqn_sample_kernel(int64_t n, int batch, uint64_t *__restrict__ state,
                                                         int64_t *__restrict__ idx, float *__restrict__ taus, int n_taus) {
    __shared__ __align__(16) int val[MAX_BATCH];
    __shared__ int flag[MAX_BATCH];
    const uint64_t ctr = state[1], base = sample_base(state);
    for (int e = threadIdx.x; e < n_taus; e += 256) taus[e] = sample_tau(base, e);
    sample_rows<256>(val, flag, n, batch, base);
    for (int k = threadIdx.x; k < batch; k += 256) idx[k] = val[k];
    if (threadIdx.x == 0) state[1] = ctr + 1;
}

}  // namespace

extern "C" int mn_iqn_sample(int64_t ring_size, int32_t batch, uint64_t *rng_state_dev, int64_t *idx_out, float *taus_out,
                             int32_t n_taus_total, void *stream) {
    if (!rng_state_dev || !idx_out || (n_taus_total > 0 && !taus_out) || n_taus_total < 0) return MN_ERR_INVALID;
    if (batch <= 0 || batch > MAX_BATCH || ring_size < batch || ring_size > 0x7fffffff) return MN_ERR_INVALID;
    hipLaunchKernelGGL(iqn_sample_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, ring_size, batch, rng_state_dev, idx_out,
                       taus_out, n_taus_total);
    return hipGetLastError() == hipSuccess ? MN_OK : MN_ERR_HIP;
}

extern "C" int64_t mn_iqn_train_workspace_floats(int32_t batch) {
    if (batch <= 0 || batch % BE) return -1;
    return (int64_t)(batch / BE) * (P_TOTAL + 1) + N_SQ;
}

static int launch_grad(const float *ring_states, const float *ring_next_states, const int64_t *ring_actions,
                       const float *ring_rewards, const float *ring_dones, const int64_t *idx_dev, const float *taus_target_dev,
                       const float *taus_local_dev, const float *params_local, const float *params_target, float *workspace,
                       float *grad_out, float *loss_out, int32_t batch, int32_t num_taus, float gamma, uint64_t *rng_state_dev,
                       int64_t ring_size, int64_t *idx_out, float *taus_out, void *stream) {
    if (!ring_states || !ring_next_states || !ring_actions || !ring_rewards || !ring_dones || !params_local || !params_target ||
        !workspace || !grad_out || !loss_out)
        return MN_ERR_INVALID;
    if (rng_state_dev ? (batch > MAX_BATCH || ring_size < batch || ring_size > 0x7fffffff) : (!idx_dev || !taus_target_dev || !taus_local_dev))
        return MN_ERR_INVALID;
    if (batch <= 0 || batch % BE || num_taus != NQ) return MN_ERR_INVALID;
    {   // raise the dynamic-LDS limit once per device; guarded so that concurrent first calls from two threads are safe
        static std::mutex mu;
        static bool attr_set[64] = {false};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return MN_ERR_HIP;
        std::lock_guard<std::mutex> lock(mu);
        if (!attr_set[dev]) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(iqn_train_fwdbwd), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    LDS_BYTES) != hipSuccess)
                return MN_ERR_HIP;
            attr_set[dev] = true;
        }
    }
    const int n_part = batch / BE;
    float *partial = workspace, *loss_partial = workspace + (size_t)n_part * P_TOTAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(iqn_train_fwdbwd, dim3(n_part), dim3(THREADS), LDS_BYTES, s, ring_states, ring_next_states, ring_actions,
                       ring_rewards, ring_dones, idx_dev, taus_target_dev, taus_local_dev, params_local, params_target, partial,
                       loss_partial, batch, gamma, (const uint64_t *)rng_state_dev, ring_size, idx_out, taus_out);
    hipLaunchKernelGGL(iqn_grad_reduce, dim3(N_SQ), dim3(1024), 0, s, partial, loss_partial, n_part, grad_out,
                       loss_out, rng_state_dev);
    return hipGetLastError() == hipSuccess ? MN_OK : MN_ERR_HIP;
}

extern "C" int mn_iqn_train_grad(const float *ring_states, const float *ring_next_states, const int64_t *ring_actions,
                                 const float *ring_rewards, const float *ring_dones, const int64_t *idx_dev,
                                 const float *taus_target_dev, const float *taus_local_dev, const float *params_local,
                                 const float *params_target, float *workspace, float *grad_out, float *loss_out,
                                 int32_t batch, int32_t num_taus, float gamma, void *stream) {
    return launch_grad(ring_states, ring_next_states, ring_actions, ring_rewards, ring_dones, idx_dev, taus_target_dev, taus_local_dev,
                       params_local, params_target, workspace, grad_out, loss_out, batch, num_taus, gamma, nullptr, 0, nullptr, nullptr,
                       stream);
}

extern "C" int mn_iqn_train_grad_sampled(const float *ring_states, const float *ring_next_states, const int64_t *ring_actions,
                                         const float *ring_rewards, const float *ring_dones, int64_t ring_size,
                                         uint64_t *rng_state_dev, int64_t *idx_out, float *taus_out, const float *params_local,
                                         const float *params_target, float *workspace, float *grad_out, float *loss_out,
                                         int32_t batch, int32_t num_taus, float gamma, void *stream) {
    if (!rng_state_dev) return MN_ERR_INVALID;
    return launch_grad(ring_states, ring_next_states, ring_actions, ring_rewards, ring_dones, nullptr, nullptr, nullptr, params_local,
                       params_target, workspace, grad_out, loss_out, batch, num_taus, gamma, rng_state_dev, ring_size, idx_out, taus_out,
                       stream);
}

extern "C" int mn_iqn_train_adam(float *params, float *grad, float *exp_avg, float *exp_avg_sq, int32_t *step_dev,
                                 float *workspace, int32_t batch, double lr, double beta1, double beta2, double eps,
                                 double max_norm, void *stream) {
    if (!params || !grad || !exp_avg || !exp_avg_sq || !step_dev || !workspace || batch <= 0 || batch % BE) return MN_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    float *blocksq = workspace + (size_t)(batch / BE) * (P_TOTAL + 1);
    hipLaunchKernelGGL(iqn_sumsq, dim3(N_SQ), dim3(256), 0, s, grad, blocksq, step_dev);
    hipLaunchKernelGGL(iqn_adam, dim3(N_SQ), dim3(256), 0, s, params, grad, exp_avg, exp_avg_sq, blocksq, step_dev, lr, beta1,
                       beta2, eps, max_norm);
    return hipGetLastError() == hipSuccess ? MN_OK : MN_ERR_HIP;
}
