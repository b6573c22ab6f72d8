// iqn_act.hip -- fused IQN action-value kernel for gfx950 (MI355X).
//
// Replaces, for inference on a whole vector of environments, ObsEncoder.forward + the mean over the
// K = 32 quantile samples of ObsEncoder.get_qvals (thirdparty/IQN/model.py:141-191):
//     cos   = cos(tau * pi * [0..63])                                  (model.py:149-155)
//     x     = relu(cos @ W1^T + b1) * features                         (:177-181, Hadamard)
//     x     = relu(x @ W2^T + b2) ; x = relu(x @ W3^T + b3) ; q = x @ W4^T + b4   (:183-185)
//     Q     = mean over the K taus                                      (:188-191)
// The three linear observation encoders (:170-173, a block-diagonal 26 -> 208 map, <1 % of the FLOPs)
// run on the VALU inside the kernel (each lane computes 3-4 of the 208 features from the wave-uniform
// observation row and parks them in a per-wave LDS buffer), so the only inputs are the raw
// observations and the taus.  An optional epilogue does the
// argmax and the epsilon-greedy choice of IQNAgent.act (agent.py:199-203).
//
// Why a kernel: in eager PyTorch this path is ~95 % of a training vector step at 65 536 envs and is
// bound by elementwise traffic -- the [n*32, 208] activation is written and re-read five times
// (profiles/r01_full_loop_kernel_stats_v1.txt).  Here a wavefront owns one environment (32 tau rows)
// at a time and carries it through all four layers in registers; nothing but observations, taus and
// the 9 Q-values / the action touches HBM.
//
// MFMA mapping: exact-f32 v_mfma_f32_16x16x4_f32 (the reference is float32; no reduced precision).
// Every layer is computed TRANSPOSED, H^T = W . X^T: the weights are the A operand (16 output
// features x 4 k), the activations the B operand (4 k x 16 tau rows) and the C tile is
// [16 features x 16 taus] with lane l holding column (l & 15) and rows 4*(l >> 4) + r.  Because the
// k order of a dot product is free, MFMA step (t, r) of the NEXT layer is defined to consume input
// features {16t + 4g + r : g = 0..3} -- which is exactly register r of C tile t in lane group g.  So a
// layer's accumulator registers ARE the next layer's B operands: no LDS round trip, no shuffles.
// The weights are permuted into that order once per call (iqn_pack_kernel) and copied to LDS per workgroup
// (155 KiB of the 160 KiB incl. the encoders: one 512-thread workgroup per CU, 2 waves per SIMD so one wave's bias /
// ReLU / cos VALU work runs under the other's MFMAs); each ds_read_b128 feeds 4 k-steps x 2 tau
// tiles = 8 MFMAs.  Layers 1 and 2 are fused over the 13 feature tiles of the 208-wide activation,
// so the live state is 32 accumulator + 32 cos registers per lane.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>

#include "marinenav_hip.h"

#define MN_IQN_VARIANT_DEFAULT 2

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int K_TAUS = 32;      // model.py:118
constexpr int N_COS = 64;       // model.py:130
constexpr int F = 208;          // 16 + 16 + 176 feature width
constexpr int H = 64;           // hidden width
constexpr int A_OUT = 9;        // actions
constexpr int T1 = F / 16;      // 13 feature tiles
// LDS layout (floats)
constexpr int OFF_W1 = 0;                         // [13 t][4 m4][64 lanes][4]
constexpr int OFF_W2 = OFF_W1 + T1 * 4 * 64 * 4;  // [4 mt][13 t][64][4]
constexpr int OFF_W3 = OFF_W2 + 4 * T1 * 64 * 4;  // [4 mt][4 t2][64][4]
constexpr int OFF_W4 = OFF_W3 + 4 * 4 * 64 * 4;   // [4 t2][64][4]
constexpr int OFF_B1 = OFF_W4 + 4 * 64 * 4;       // [208]
constexpr int OFF_B2 = OFF_B1 + F;                // [64]
constexpr int OFF_B3 = OFF_B2 + H;                // [64]
constexpr int OFF_B4 = OFF_B3 + H;                // [16]
constexpr int OBS = MN_OBS_DIM;                   // 26
constexpr int OBS4 = 7;                           // 26 inputs padded to 7 float4
constexpr int OFF_WE = OFF_B4 + 16;               // [7 i4][208 f][4]: block-diagonal encoder weights
constexpr int OFF_BE = OFF_WE + OBS4 * F * 4;     // [208] encoder biases
constexpr int OFF_FB = OFF_BE + F;                // [8 waves][208] per-wave feature buffer
constexpr int LDS_FLOATS = OFF_FB + 8 * F;

__device__ __forceinline__ f32x4 mfma(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

__device__ __forceinline__ f32x4 relu4(f32x4 v) {
    f32x4 r;
    r.x = v.x > 0.f ? v.x : 0.f; r.y = v.y > 0.f ? v.y : 0.f; r.z = v.z > 0.f ? v.z : 0.f; r.w = v.w > 0.f ? v.w : 0.f;
    return r;
}

// sum over the 16 lanes of a row (lanes sharing l >> 4)
__device__ __forceinline__ float row_sum16(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad xor 1
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad xor 2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
    return v;
}

struct IqnWeights {   // device pointers, nn.Linear layout [out][in]
    const float *ve_w, *ve_b, *ge_w, *ge_b, *se_w, *se_b;   // velocity / goal / sensor encoders
    const float *W1, *b1, *W2, *b2, *W3, *b3, *W4, *b4;     // cos_embedding, hidden_layer, hidden_layer_2, output_layer
};

// block-diagonal encoder weight: feature f (0..207) x observation input i (0..25)  (model.py:126-128,170-173)
__device__ __forceinline__ float enc_weight(const IqnWeights &w, int f, int i) {
    if (f < 16) return (i < 2) ? w.ve_w[f * 2 + i] : 0.f;
    if (f < 32) return (i >= 2 && i < 4) ? w.ge_w[(f - 16) * 2 + (i - 2)] : 0.f;
    return (i >= 4 && i < OBS) ? w.se_w[(f - 32) * 22 + (i - 4)] : 0.f;
}

// Value of element i of the LDS weight image (floats [0, OFF_FB)): weights permuted into MFMA A-fragment
// order (nn.Linear stores [out][in]), biases, block-diagonal encoder.
__device__ __forceinline__ float pack_element(const IqnWeights &w, int i) {
    if (i < OFF_B1) {
        const int j = i & 3, l = (i >> 2) & 63, g = l >> 4, row = l & 15;
        if (i < OFF_W2) {            // W1p[t][m4][l][j] = W1[16t + row][4*(4*m4 + j) + g]
            const int q = i >> 8, m4 = q & 3, t = q >> 2;
            return w.W1[(16 * t + row) * N_COS + 4 * (4 * m4 + j) + g];
        } else if (i < OFF_W3) {     // W2p[mt][t][l][r] = W2[16mt + row][16t + 4g + r]
            const int q = (i - OFF_W2) >> 8, t = q % T1, mt = q / T1;
            return w.W2[(16 * mt + row) * F + 16 * t + 4 * g + j];
        } else if (i < OFF_W4) {     // W3p[mt][t2][l][r] = W3[16mt + row][16t2 + 4g + r]
            const int q = (i - OFF_W3) >> 8, t2 = q & 3, mt = q >> 2;
            return w.W3[(16 * mt + row) * H + 16 * t2 + 4 * g + j];
        }                            // W4p[t2][l][r] = W4[row][16t2 + 4g + r] (rows >= 9 are zero)
        const int t2 = (i - OFF_W4) >> 8;
        return row < A_OUT ? w.W4[row * H + 16 * t2 + 4 * g + j] : 0.f;
    }
    if (i < OFF_B2) return w.b1[i - OFF_B1];
    if (i < OFF_B3) return w.b2[i - OFF_B2];
    if (i < OFF_B4) return w.b3[i - OFF_B3];
    if (i < OFF_WE) return (i - OFF_B4) < A_OUT ? w.b4[i - OFF_B4] : 0.f;
    if (i < OFF_BE) {                // WEp[i4][f][c] = Wenc[f][4*i4 + c] (block-diagonal 208 x 26, zero elsewhere / padding)
        const int k = i - OFF_WE, c = k & 3, f = (k >> 2) % F, i4 = (k >> 2) / F;
        const int inp = 4 * i4 + c;
        return inp < OBS ? enc_weight(w, f, inp) : 0.f;
    }
    const int f = i - OFF_BE;
    return f < 16 ? w.ve_b[f] : (f < 32 ? w.ge_b[f - 16] : w.se_b[f - 32]);
}

// Builds the 149 KiB LDS image once per call in global memory, so that each of the 256 workgroups of the
// act kernel fills its LDS with a straight 16-byte coalesced copy instead of a 38 K-element gather.
__global__ __launch_bounds__(256) void iqn_pack_kernel(IqnWeights w, float *__restrict__ packed) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < OFF_FB) packed[i] = pack_element(w, i);
}

// Counter-based uniform draws: draw number `idx` of call `ctr` is a double murmur3-fmix32 of the index under two 32-bit
// keys derived from (seed, ctr) -- no generator state per element, any element can be produced by any thread.
__device__ __forceinline__ uint32_t fmix32(uint32_t x) {
    x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ float u01(uint32_t idx, uint32_t k0, uint32_t k1) {      // 24-bit uniform in [0, 1), like torch.rand
    return (float)(fmix32(fmix32(idx ^ k0) + k1) >> 8) * (1.0f / 16777216.0f);
}

constexpr int PACK_BLOCKS = (OFF_FB + 255) / 256;

// The random numbers of one act call: blocks [pack_blocks, gridDim.x) fill draws[0 .. 32 n) with tau = U[0,1) * cvar
// (model.py:149-153; per-row cvar if cvar_row) and draws[32 n .. 33 n) with the exploration uniforms of IQNAgent.act
// (agent.py:199).
__device__ __forceinline__ void draw_block(const uint64_t *__restrict__ rng_state, float *__restrict__ draws, int n,
                                           const float *__restrict__ cvar_row, float cvar, int pack_blocks) {
    const uint64_t base = mix64(rng_state[0] + 0x9E3779B97F4A7C15ull * (rng_state[1] + 1));
    const uint32_t k0 = (uint32_t)base, k1 = (uint32_t)(base >> 32);
    const long total4 = ((long)n * (K_TAUS + 1) + 3) / 4;          // float4 groups
    const long stride = (long)((int)gridDim.x - pack_blocks) * 256;
    for (long q = (long)((int)blockIdx.x - pack_blocks) * 256 + threadIdx.x; q < total4; q += stride) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long idx = 4 * q + j;
            float u = u01((uint32_t)idx, k0, k1);
            if (idx < (long)n * K_TAUS) u *= cvar_row ? cvar_row[idx / K_TAUS] : cvar;
            v[j] = u;
        }
        if (4 * q + 3 < (long)n * (K_TAUS + 1)) *reinterpret_cast<float4 *>(draws + 4 * q) = make_float4(v[0], v[1], v[2], v[3]);
        else
            for (int j = 0; j < 4 && 4 * q + j < (long)n * (K_TAUS + 1); ++j) draws[4 * q + j] = v[j];
    }
}

// The same weight image PLUS the random numbers of the call in one launch: blocks [0, PACK_BLOCKS) pack, the others
// fill draws[0 .. 32 n) with tau = U[0,1) * cvar (model.py:149-153; per-row cvar if cvar_row) and draws[32 n .. 33 n)
// with the exploration uniforms of IQNAgent.act (agent.py:199).  rng_state = {seed, call counter}; the counter is
// advanced by the act kernel that follows in the stream.
__global__ __launch_bounds__(256) void iqn_prep_kernel(IqnWeights w, float *__restrict__ packed, const uint64_t *__restrict__ rng_state,
                                                       float *__restrict__ draws, int n, const float *__restrict__ cvar_row,
                                                       float cvar, int pack_blocks) {
    // pack_blocks = PACK_BLOCKS when the cached weight image is stale (mn_iqn_weights_changed), else 0
    if ((int)blockIdx.x < pack_blocks) {
        const int i = blockIdx.x * blockDim.x + threadIdx.x;
        if (i < OFF_FB) packed[i] = pack_element(w, i);
        return;
    }
    draw_block(rng_state, draws, n, cvar_row, cvar, pack_blocks);
}

// QUANT = false: the training / acting hot path (tau-mean before the linear output layer, 960 MFMAs per env).
// QUANT = true : IQNAgent.act_eval (agent.py:217-236): the output layer runs per tau on the matrix pipe (+32 MFMAs on a
//                padded 16-row tile), the [n][32][9] quantile values are written out and Q is their mean.
template <bool QUANT>
__global__ __launch_bounds__(512, 2) void iqn_qvals_kernel(const float *__restrict__ obs, const float *__restrict__ taus,
                                                           const float *__restrict__ packed, float *__restrict__ qvals,
                                                           const float *__restrict__ explore_u, float eps,
                                                           int32_t *__restrict__ actions, int n,
                                                           uint64_t *__restrict__ rng_state, float *__restrict__ quantiles) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    if (rng_state && blockIdx.x == 0 && tid == 0) rng_state[1] += 1;   // the draws of this call were made by iqn_prep_kernel
    {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(packed);
        f32x4 *dst = reinterpret_cast<f32x4 *>(lds);
        for (int i = tid; i < OFF_FB / 4; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();

    const int lane = tid & 63, g = lane >> 4, col = lane & 15;
    const int wave = tid >> 6, waves_per_block = blockDim.x >> 6;
    const f32x4 *ldsv = reinterpret_cast<const f32x4 *>(lds);

    // cos(tau * pi * k) = cos(2 pi * (tau * k / 2)), k = 4m + g: the phase in REVOLUTIONS is tau * (k/2),
    // one exact-ish multiply; v_fract + v_cos_f32 replace libm's ~35-instruction range reduction.  The
    // reference rounds tau * float32(pi k) before its cos (model.py:130,155), so the two already
    // differ by ~1e-5 rad of input rounding at k = 63; that noise dominates either cos error.
    float hk[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) hk[m] = 0.5f * (float)(4 * m + g);

    // one environment (32 tau rows = NT = 2 column tiles) per wave iteration
    constexpr int NT = 2;
    for (int e = blockIdx.x * waves_per_block + wave; e < n; e += gridDim.x * waves_per_block) {
        float tau[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) tau[nt] = taus[(size_t)e * K_TAUS + 16 * nt + col];
        // layer-1 B operands: cos(tau * pis[k]) for k = 4m + g  (model.py:155)
        float cb[16][NT];
#pragma unroll
        for (int m = 0; m < 16; ++m)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) cb[m][nt] = __builtin_amdgcn_cosf(__builtin_amdgcn_fractf(tau[nt] * hk[m]));

        // ---- observation encoders (model.py:170-173): lane l computes features l, l+64, l+128, l+192 from
        // the 26 inputs (wave-uniform -> scalar loads) and parks them in this wave's LDS buffer, from
        // where every lane later reads the float4 {16t + 4g + r} it needs for the Hadamard product
        {
            const float *orow = obs + (size_t)__builtin_amdgcn_readfirstlane(e) * OBS;
            float ov[OBS4 * 4];
#pragma unroll
            for (int i = 0; i < OBS4 * 4; ++i) ov[i] = i < OBS ? orow[i] : 0.f;
            float *fb = lds + OFF_FB + wave * F;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int f = lane + 64 * j;
                if (f < F) {
                    float a = lds[OFF_BE + f];
#pragma unroll
                    for (int i4 = 0; i4 < OBS4; ++i4) {
                        const f32x4 wv = ldsv[(OFF_WE >> 2) + i4 * F + f];
                        a += wv[0] * ov[4 * i4] + wv[1] * ov[4 * i4 + 1] + wv[2] * ov[4 * i4 + 2] + wv[3] * ov[4 * i4 + 3];
                    }
                    fb[f] = a;
                }
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        const f32x4 *fbv = reinterpret_cast<const f32x4 *>(lds + OFF_FB + wave * F) + g;   // + 4*t per tile

        f32x4 acc2[4][NT];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc2[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

        // ---- layers 1 + 2 fused over the 13 feature tiles, software-pipelined: the layer-1 MFMAs of
        // tile t+1 are issued BEFORE the bias / ReLU / Hadamard epilogue of tile t, so the wave's own VALU
        // work sits in the shadow of its own MFMAs (in-order issue would otherwise drain the matrix pipe
        // at every tile boundary) -------------------------------------------------------------------
        f32x4 acc1[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc1[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m4 = 0; m4 < 4; ++m4) {
            const f32x4 a = ldsv[(OFF_W1 >> 2) + m4 * 64 + lane];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc1[nt] = mfma(a[j], cb[4 * m4 + j][nt], acc1[nt]);
        }
#pragma unroll
        for (int t = 0; t < T1; ++t) {
            f32x4 nxt[NT];
            if (t + 1 < T1) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) nxt[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int m4 = 0; m4 < 4; ++m4) {
                    const f32x4 a = ldsv[(OFF_W1 >> 2) + ((t + 1) * 4 + m4) * 64 + lane];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) nxt[nt] = mfma(a[j], cb[4 * m4 + j][nt], nxt[nt]);
                }
            }
            const f32x4 fv = fbv[4 * t];                             // features[e][16t + 4g + r]
            const f32x4 bias = ldsv[(OFF_B1 >> 2) + 4 * t + g];      // b1[16t + 4g + r]
            f32x4 h1[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) h1[nt] = relu4(acc1[nt] + bias) * fv;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const f32x4 a = ldsv[(OFF_W2 >> 2) + (mt * T1 + t) * 64 + lane];
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc2[mt][nt] = mfma(a[r], h1[nt][r], acc2[mt][nt]);
            }
            if (t + 1 < T1) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc1[nt] = nxt[nt];
            }
        }
        // ---- layer 2 epilogue, layer 3 ---------------------------------------------------------------
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const f32x4 bias = ldsv[(OFF_B2 >> 2) + 4 * mt + g];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc2[mt][nt] = relu4(acc2[mt][nt] + bias);
        }
        f32x4 acc3[4][NT];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc3[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t2 = 0; t2 < 4; ++t2) {
                const f32x4 a = ldsv[(OFF_W3 >> 2) + (mt * 4 + t2) * 64 + lane];
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc3[mt][nt] = mfma(a[r], acc2[t2][nt][r], acc3[mt][nt]);
            }
            const f32x4 bias = ldsv[(OFF_B3 >> 2) + 4 * mt + g];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc3[mt][nt] = relu4(acc3[mt][nt] + bias);
        }
        // ---- layer 4 + mean over the 32 taus (model.py:185,190).  The output layer is linear, so
        // mean_tau(W4 h3(tau) + b4) = W4 mean_tau(h3(tau)) + b4: the tau mean is taken FIRST (DPP row sums of the
        // layer-3 accumulators) and the 9 x 64 output layer becomes one small VALU mat-vec per environment
        // instead of 32 MFMAs on a padded 16-row tile (3 % of the kernel's matrix work).
        // After row_sum16 every lane of row group g holds sum_tau h3[16mt + 4g + r]; lane (g, col) then forms the
        // part of action `col` that comes from its 16 features (W4p[mt][lane][r] = W4[col][16mt + 4g + r], zero rows
        // for col >= 9) and the four row groups are added with two cross-row shuffles.
        float qv;
        if constexpr (!QUANT) {
            float part = 0.f;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const f32x4 a = ldsv[(OFF_W4 >> 2) + mt * 64 + lane];
#pragma unroll
                for (int r = 0; r < 4; ++r) part = fmaf(a[r], row_sum16(acc3[mt][0][r] + acc3[mt][1][r]), part);
            }
            part += __shfl_xor(part, 16);
            part += __shfl_xor(part, 32);
            qv = part * (1.0f / K_TAUS) + lds[OFF_B4 + col];     // Q(s, action = col), valid for col < 9
        } else {
            // quantile values Z(tau, a) = W4 h3(tau) + b4 (model.py:185): C tile [16 padded actions x 16 taus] per tau tile;
            // lane (g, col) holds actions 4g + r of tau 16 nt + col
            f32x4 acc4[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc4[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t2 = 0; t2 < 4; ++t2) {
                const f32x4 a = ldsv[(OFF_W4 >> 2) + t2 * 64 + lane];
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc4[nt] = mfma(a[r], acc3[t2][nt][r], acc4[nt]);
            }
            const f32x4 b4 = ldsv[(OFF_B4 >> 2) + g];
            float mine = 0.f;      // lane `a` (< 9) ends up with Q(s, a) = mean over the 32 taus
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int a_idx = 4 * g + r;
                float sum = 0.f;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float z = acc4[nt][r] + b4[r];
                    if (a_idx < A_OUT) quantiles[((size_t)e * K_TAUS + 16 * nt + col) * A_OUT + a_idx] = z;
                    sum += z;
                }
                sum = row_sum16(sum);                    // over the 16 tau columns of the row group
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) {         // hand action 4 gg + r to lane (4 gg + r)
                    const float v = __shfl(sum, 16 * gg);
                    if (lane == 4 * gg + r) mine = v;
                }
            }
            qv = mine * (1.0f / K_TAUS);
        }
        if (qvals && lane < A_OUT) qvals[(size_t)e * A_OUT + lane] = qv;
        // ---- IQNAgent.act epilogue (agent.py:199-203): argmax, epsilon-greedy ------------------------
        if (actions) {
            // lane a holds action a; gather the 9 values (first maximum wins, like np.argmax)
            float best = -INFINITY;
            int arg = 0;
#pragma unroll
            for (int a = 0; a < A_OUT; ++a) {
                const float v = __shfl(qv, a);
                if (v > best) { best = v; arg = a; }
            }
            if (lane == 0) {
                int act = arg;
                if (explore_u && eps > 0.f) {
                    const float u = explore_u[e];            // greedy iff u > eps (agent.py:200)
                    if (!(u > eps)) { act = (int)(u / eps * (float)A_OUT); act = act > A_OUT - 1 ? A_OUT - 1 : act; }
                }
                actions[e] = act;
            }
        }
    }
}


#include "iqn_act_split.h"
#include "iqn_act_tiled.h"

// ---- what clock does THIS GPU sustain under f16 matrix load?  (mn_probe_mfma_clock; round 4)
// The same act binary runs 10-12 % slower on some boxes of the pool (304-318 us vs 352-367 us per 65 536-env launch) while the
// exact-f32 kernel does not move.  This probe separates a slow box from a slow kernel: a pure stream of v_mfma_f32_16x16x32_f16 -- the act
// kernel's matrix instruction -- from two waves per SIMD on every CU.  The instruction occupies the SIMD's matrix pipe for 16 cycles
// (4 passes), so with the pipe saturated   effective clock = 16 x (matrix instructions per SIMD) / elapsed time.
// Wave 0 of every workgroup also brackets its loop with s_memtime (shader-clock ticks) and s_memrealtime (constant 100 MHz).
__global__ __launch_bounds__(512) void mfma_clock_probe_kernel(int iters, unsigned long long *__restrict__ stamps, float *__restrict__ sink) {
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    const int lane = threadIdx.x & 63;
    h8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (lane + i)); b[i] = (_Float16)(0.002f * (lane - i)); }
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[u & 3], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    float s = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (s == 12345.678f) sink[0] = s;      // keeps the accumulators alive
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

}  // namespace

// C-ABI ----------------------------------------------------------------------------------------------
#include <vector>

extern "C" int mn_probe_mfma_clock(double target_ms, double *out, void *stream) {
    if (!out || !(target_ms > 0.0) || target_ms > 2000.0) return MN_ERR_INVALID;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return MN_ERR_NO_DEVICE;
    const int n_cu = prop.multiProcessorCount;
    unsigned long long *stamps = nullptr;
    float *sink = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&stamps), 2 * n_cu * sizeof(unsigned long long)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&sink), sizeof(float)) != hipSuccess) { (void)hipFree(stamps); return MN_ERR_ALLOC; }
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    int rc = MN_OK;
    int iters = 2000;
    float ms = 0.f;
    for (int pass = 0; pass < 2 && rc == MN_OK; ++pass) {      // pass 0 calibrates the loop count, pass 1 is the measurement
        (void)hipEventRecord(e0, s);
        hipLaunchKernelGGL(mfma_clock_probe_kernel, dim3(n_cu), dim3(512), 0, s, iters, stamps, sink);
        (void)hipEventRecord(e1, s);
        if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || !(ms > 0.f)) { rc = MN_ERR_HIP; break; }
        if (pass == 0) {
            double scaled = iters * target_ms / ms;
            iters = scaled > 5e7 ? 50000000 : (scaled < 100 ? 100 : (int)scaled);
        }
    }
    if (rc == MN_OK) {
        std::vector<unsigned long long> h(2 * n_cu);
        if (hipMemcpy(h.data(), stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) rc = MN_ERR_HIP;
        else {
            double ct = 0, rt = 0;
            for (int i = 0; i < n_cu; ++i) { ct += (double)h[2 * i]; rt += (double)h[2 * i + 1]; }
            const double per_simd = 2.0 * iters * 16.0;      // two waves per SIMD, 16 matrix instructions per loop iteration
            out[0] = ms;
            out[1] = 16.0 * per_simd / (ms * 1e-3) / 1e9;     // GHz the matrix pipe ran at, if saturated
            out[2] = rt > 0 ? ct / rt * 0.1 : 0.0;            // GHz by the wave's own counters: shader ticks per 100 MHz tick
            out[3] = per_simd * 4.0 * n_cu * 16384.0 / (ms * 1e-3) / 1e12;      // sustained f16 TFLOP/s of the whole chip (2 x 16 x 16 x 32 FLOP each)
            out[4] = (double)n_cu;
        }
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(stamps); (void)hipFree(sink);
    return rc;
}

// Per-caller state of the act path: the permuted LDS weight image (cached between calls until the caller says the
// weights changed), the profiling events.  One context per agent / per stream: two contexts never share a buffer, so
// agents acting on different streams of one device cannot race on the image.
struct mn_iqn_ctx {
    int device = -1;
    int n_cu = 0;
    float *packed = nullptr;       // weight image of the exact-f32 16x16x4 kernel (variant 0)
    uint32_t *packed_sp = nullptr; // weight image of the split-f16 kernel (iqn_act_split.h)
    float *consts_sp = nullptr;    // its scale / bound constants
    float *h1_sp = nullptr;        // the launch's layer-1 constant [32 taus x 208] of the shared-tau kernels (mn_iqn_set_tau_mode) + 32 block maxima
    uint32_t *timg = nullptr;      // tiled shared-tau kernel (iqn_act_tiled.h): T = W2 h1 as hi / lo f16 pairs, and its auxiliary float block
    float *taux = nullptr;
    int tau_mode = 0;              // 0 = every environment its own 32 taus (the reference's per-call draw), 1 = one set of 32 per launch
    bool dirty = true, dirty_sp = true;
    int variant = MN_IQN_VARIANT_DEFAULT;   // mn_iqn_set_variant
    int max_blocks = 0;                     // mn_iqn_set_grid: 0 = one persistent workgroup per CU
    sp::LateRows late = {};                 // mn_iqn_set_late_rows: consumed by the next launch
    uint32_t *late_status = nullptr;        // waits of late rows that ran out (device word)
    volatile uint32_t *late_status_host = nullptr;      // ... and the host-mapped copy the kernel keeps of it (mn_iqn_late_timeouts_peek: no synchronisation)
    uint32_t *late_status_host_dev = nullptr;
    uint64_t late_bound_ticks = sp::LATE_BOUND_TICKS;    // mn_iqn_set_late_bound_ms
    std::vector<hipEvent_t> ev;
    int prof_max = 0, prof_n = 0;
};

extern "C" int mn_iqn_set_grid(mn_iqn_ctx *c, int32_t max_workgroups) {
    if (!c || max_workgroups < 0) return MN_ERR_INVALID;
    c->max_blocks = max_workgroups;
    return MN_OK;
}

extern "C" int mn_iqn_create(mn_iqn_ctx **out) {
    if (!out) return MN_ERR_INVALID;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return MN_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return MN_ERR_HIP;
    // per-device function attribute; setting it again for another context is harmless and has no shared host state
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(iqn_qvals_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            LDS_FLOATS * (int)sizeof(float)) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void *>(iqn_qvals_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            LDS_FLOATS * (int)sizeof(float)) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void *>(sp::iqn_qvals_split_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            sp::LDS_FLOATS * (int)sizeof(float)) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void *>(sp::iqn_qvals_split_kernel<false, false, sp::WAVES, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            sp::LDS_FLOATS * (int)sizeof(float)) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void *>(sp::iqn_qvals_split_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            sp::LDS_FLOATS * (int)sizeof(float)) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void *>(sp::iqn_qvals_tiled_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            sp::TL_FLOATS * (int)sizeof(float)) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void *>(sp::iqn_qvals_split_kernel<false, true, 8>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            sp::OFF_FB * (int)sizeof(float)) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void *>(sp::iqn_qvals_split_kernel<true, true, 8>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            sp::OFF_FB * (int)sizeof(float)) != hipSuccess)
        return MN_ERR_HIP;
    mn_iqn_ctx *c = new mn_iqn_ctx();
    c->device = dev;
    c->n_cu = prop.multiProcessorCount;
    if (hipMalloc(reinterpret_cast<void **>(&c->packed), OFF_FB * sizeof(float)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&c->packed_sp), sp::OFF_FB * sizeof(uint32_t)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&c->consts_sp), sp::N_CONST_BUF * sizeof(float)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&c->h1_sp), (sp::H1_FLOATS + 32) * sizeof(float)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&c->timg), sp::T_WORDS * sizeof(uint32_t)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&c->taux), sp::TA_FLOATS * sizeof(float)) != hipSuccess ||
        hipMemset(c->consts_sp, 0, sp::N_CONST_BUF * sizeof(float)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        (void)hipFree(c->packed);
        (void)hipFree(c->packed_sp);
        (void)hipFree(c->consts_sp);
        (void)hipFree(c->h1_sp);
        (void)hipFree(c->timg);
        (void)hipFree(c->taux);
        delete c;
        return MN_ERR_ALLOC;
    }
    *out = c;
    return MN_OK;
}

extern "C" int mn_iqn_destroy(mn_iqn_ctx *c) {
    if (!c) return MN_ERR_INVALID;
    int cur = -1;
    const bool moved = hipGetDevice(&cur) == hipSuccess && cur != c->device;
    if (moved) (void)hipSetDevice(c->device);
    for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
    (void)hipFree(c->packed);
    (void)hipFree(c->packed_sp);
    (void)hipFree(c->consts_sp);
    (void)hipFree(c->h1_sp);
    (void)hipFree(c->timg);
    (void)hipFree(c->taux);
    (void)hipFree(c->late_status);
    if (c->late_status_host) (void)hipHostFree((void *)c->late_status_host);
    if (moved) (void)hipSetDevice(cur);
    delete c;
    return MN_OK;
}

extern "C" int mn_iqn_weights_changed(mn_iqn_ctx *c) {
    if (!c) return MN_ERR_INVALID;
    c->dirty = true;
    c->dirty_sp = true;
    return MN_OK;
}

extern "C" int mn_iqn_set_variant(mn_iqn_ctx *c, int32_t variant) {
    if (!c || (variant != 0 && variant != 2)) return MN_ERR_INVALID;      // (1 and 3 were the 32x32 re-layouts of the two kernels: measured slower, removed in round 6)
    c->variant = variant;
    return MN_OK;
}

extern "C" int mn_iqn_set_tau_mode(mn_iqn_ctx *c, int32_t mode) {
    if (!c || mode < 0 || mode > 3) return MN_ERR_INVALID;
    c->tau_mode = mode;
    return MN_OK;
}

extern "C" int mn_iqn_profile_begin(mn_iqn_ctx *c, int32_t max_launches) {
    if (!c || max_launches < 0 || max_launches > 65536) return MN_ERR_INVALID;
    while ((int)c->ev.size() < 2 * max_launches) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return MN_ERR_HIP;
        c->ev.push_back(e);
    }
    c->prof_max = max_launches;
    c->prof_n = 0;
    return MN_OK;
}

extern "C" int mn_iqn_profile_end(mn_iqn_ctx *c, void *stream, double *mean_ms, int32_t *launches) {
    if (!c) return MN_ERR_INVALID;
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return MN_ERR_HIP;
    double sum = 0.0;
    for (int i = 0; i < c->prof_n; ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->ev[2 * i], c->ev[2 * i + 1]) != hipSuccess) return MN_ERR_HIP;
        sum += ms;
    }
    if (mean_ms) *mean_ms = c->prof_n ? sum / c->prof_n : 0.0;
    if (launches) *launches = c->prof_n;
    c->prof_max = 0;
    c->prof_n = 0;
    return MN_OK;
}

// The acting kernel that takes late rows: split-f16, per-environment taus, no quantile capture, at most 64 rows per wavefront.
static bool late_rows_supported(const mn_iqn_ctx *c, int n, bool quantiles) {
    if (c->variant != 2 || c->tau_mode != 0 || quantiles || n <= 0) return false;
    int blocks = (n + 7) / 8;
    const int cap = c->max_blocks > 0 ? c->max_blocks : c->n_cu;
    if (blocks > cap) blocks = cap;
    return (long)n <= 64L * 8L * blocks;
}

// Rows of the next act launch whose observation is still being written by a reset launch on another stream (mn_reset_done_async):
// `mask_dev` [n] (the step's done flags), `flags_dev` [n] / `tick` from mn_reset_done_async.  Returns MN_OK if the next launch of `n` rows will
// take them (then launch it with nothing in between), 1 if this context's current form cannot -- the caller then joins the reset (mn_reset_join) first.
extern "C" int mn_iqn_set_late_rows(mn_iqn_ctx *c, const uint8_t *mask_dev, const uint32_t *flags_dev, uint32_t tick, int32_t n) {
    if (!c) return MN_ERR_INVALID;
    c->late = sp::LateRows{};
    if (!mask_dev && !flags_dev) return MN_OK;      // clear
    if (!mask_dev || !flags_dev) return MN_ERR_INVALID;
    if (!late_rows_supported(c, n, false)) return 1;
    if (!c->late_status) {      // (all or nothing: the context only keeps the words once every allocation has succeeded)
        uint32_t *st = nullptr;
        void *hp = nullptr, *hd = nullptr;
        if (hipMalloc(reinterpret_cast<void **>(&st), sizeof(uint32_t)) != hipSuccess) return MN_ERR_ALLOC;
        if (hipMemset(st, 0, sizeof(uint32_t)) != hipSuccess || hipHostMalloc(&hp, sizeof(uint32_t), hipHostMallocMapped) != hipSuccess) { (void)hipFree(st); return MN_ERR_ALLOC; }
        *(volatile uint32_t *)hp = 0u;
        if (hipHostGetDevicePointer(&hd, hp, 0) != hipSuccess) { (void)hipFree(st); (void)hipHostFree(hp); return MN_ERR_HIP; }
        c->late_status = st; c->late_status_host = (volatile uint32_t *)hp; c->late_status_host_dev = (uint32_t *)hd;
    }
    c->late = sp::LateRows{mask_dev, flags_dev, tick, c->late_status, c->late_status_host_dev, c->late_bound_ticks};
    return MN_OK;
}

// Waits for a late row that ran out since the context was made (0 in a healthy run; anything else: actions were computed on unfinished observations).
extern "C" int mn_iqn_late_timeouts(mn_iqn_ctx *c, void *stream, uint32_t *out) {
    if (!c || !out) return MN_ERR_INVALID;
    *out = 0;
    if (!c->late_status) return MN_OK;
    if (hipMemcpyAsync(out, c->late_status, sizeof(uint32_t), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess) return MN_ERR_HIP;
    return hipStreamSynchronize((hipStream_t)stream) == hipSuccess ? MN_OK : MN_ERR_HIP;
}

// The same count as the launches executed so far have left it in a host-mapped word: no synchronisation, so a training loop can look every few
// vector steps (a launch still in flight is not counted yet).
extern "C" int mn_iqn_late_timeouts_peek(mn_iqn_ctx *c, uint32_t *out) {
    if (!c || !out) return MN_ERR_INVALID;
    *out = c->late_status_host ? *c->late_status_host : 0u;
    return MN_OK;
}

// Bound of a late row's wait in milliseconds (default 500; tests use a short one).  Applies to launches armed after the call.
extern "C" int mn_iqn_set_late_bound_ms(mn_iqn_ctx *c, double ms) {
    if (!c || !(ms > 0.0) || ms > 60000.0) return MN_ERR_INVALID;
    c->late_bound_ticks = (uint64_t)(ms * 1.0e5);      // 100 MHz counter
    return MN_OK;
}

static int launch_act(mn_iqn_ctx *c, const float *obs_dev, const float *taus_dev, const float *const *weights, float *qvals_dev,
                      const float *explore_u_dev, float eps, int32_t *actions_dev, float *quantiles_dev, int32_t n,
                      int32_t num_taus, uint64_t *rng_state_dev, float *draws_dev, const float *cvar_row_dev, float cvar,
                      void *stream) {
    if (!c || !obs_dev || !weights || (!qvals_dev && !actions_dev && !quantiles_dev)) return MN_ERR_INVALID;
    if (rng_state_dev ? !draws_dev : !taus_dev) return MN_ERR_INVALID;
    if (rng_state_dev && (long)n * (K_TAUS + 1) >= (1L << 32)) return MN_ERR_INVALID;   // 32-bit draw index
    for (int i = 0; i < 14; ++i) if (!weights[i]) return MN_ERR_INVALID;
    if (n <= 0 || num_taus != K_TAUS) return MN_ERR_INVALID;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev != c->device) return MN_ERR_INVALID;   // context lives on another device
    const IqnWeights w = {weights[0], weights[1], weights[2], weights[3], weights[4], weights[5], weights[6],
                          weights[7], weights[8], weights[9], weights[10], weights[11], weights[12], weights[13]};
    int blocks = (n + 7) / 8;
    const int cap = c->max_blocks > 0 ? c->max_blocks : c->n_cu;
    if (blocks > cap) blocks = cap;
    hipStream_t s = (hipStream_t)stream;
    // late rows (mn_iqn_set_late_rows) belong to THIS launch only; a form that cannot honour them must never see them (the caller joined the reset instead)
    const sp::LateRows late = c->late;
    c->late = sp::LateRows{};
    if (late.mask && !late_rows_supported(c, n, quantiles_dev != nullptr)) return MN_ERR_INVALID;
    const bool prof = c->prof_n < c->prof_max;
    if (prof) (void)hipEventRecord(c->ev[2 * c->prof_n], s);
    // variants (mn_iqn_set_variant): 0 = exact-f32 16x16x4 kernel, 2 = split-f16 kernel (iqn_act_split.h); each has a quantile-capture form (act_eval)
    const bool use_sp = c->variant == 2;
    if (c->tau_mode != 0) {
        // Launch-shared taus: ONE set of 32 quantile fractions for every environment of the launch (iqn_act_split.h, stage_sh).  Only the
        // split-f16 kernel has this form; per-row CVaR (adaptive policies) needs per-environment taus.
        if (c->variant != 2 || cvar_row_dev) return MN_ERR_INVALID;
        const int pack_blocks = c->dirty_sp ? sp::PACK_BLOCKS : 0;
        if (pack_blocks) hipLaunchKernelGGL(sp::iqn_split_consts_kernel, dim3(sp::CONST_BLOCKS), dim3(256), 0, s, w, c->consts_sp);
        int rng_blocks = 0;
        if (rng_state_dev) {
            rng_blocks = (int)(((long)n + K_TAUS + 255) / 256);
            if (rng_blocks > 8 * c->n_cu) rng_blocks = 8 * c->n_cu;
        }
        hipLaunchKernelGGL(sp::iqn_shared_prep_kernel, dim3(pack_blocks + sp::H1_BLOCKS + rng_blocks), dim3(256), 0, s, w, (const float *)c->consts_sp,
                           c->packed_sp, (const uint64_t *)rng_state_dev, draws_dev, n, rng_state_dev ? nullptr : taus_dev, cvar, pack_blocks, c->h1_sp);
        if (rng_state_dev) explore_u_dev = eps > 0.f ? draws_dev + K_TAUS : nullptr;
        c->dirty_sp = false;
        if (!quantiles_dev && ((c->tau_mode == 1 && n >= sp::TILED_MIN_ENVS) || c->tau_mode == 3)) {
            // large batch: the MFMA columns are environments (iqn_act_tiled.h): T = W2 h1 built once, 32 environments per wavefront
            hipLaunchKernelGGL(sp::iqn_tiled_prep_kernel, dim3(sp::T_PREP_BLOCKS + sp::TA_PREP_BLOCKS), dim3(256), 0, s, w, (const float *)c->consts_sp,
                               (const float *)c->h1_sp, c->timg, c->taux);
            hipLaunchKernelGGL(sp::iqn_qvals_tiled_kernel, dim3((n + 255) / 256), dim3(512), sp::TL_FLOATS * sizeof(float), s, obs_dev,
                               (const uint32_t *)c->packed_sp, (const uint32_t *)c->timg, (const float *)c->taux, qvals_dev, explore_u_dev, eps,
                               actions_dev, n, rng_state_dev);
            if (prof) { (void)hipEventRecord(c->ev[2 * c->prof_n + 1], s); ++c->prof_n; }
            return hipGetLastError() == hipSuccess ? MN_OK : MN_ERR_HIP;
        }
        if (quantiles_dev)
            hipLaunchKernelGGL((sp::iqn_qvals_split_kernel<true, true, 8>), dim3(blocks), dim3(512), sp::OFF_FB * sizeof(float), s, obs_dev, (const float *)nullptr,
                               (const uint32_t *)c->packed_sp, qvals_dev, explore_u_dev, eps, actions_dev, n, rng_state_dev, quantiles_dev, (const float *)c->h1_sp);
        else      // (12 waves per workgroup -- three per SIMD, the kernel needs 153 registers -- measured: 202-204 us against 203, no gain)
            hipLaunchKernelGGL((sp::iqn_qvals_split_kernel<false, true, 8>), dim3(blocks), dim3(512), sp::OFF_FB * sizeof(float), s, obs_dev, (const float *)nullptr,
                               (const uint32_t *)c->packed_sp, qvals_dev, explore_u_dev, eps, actions_dev, n, rng_state_dev, (float *)nullptr, (const float *)c->h1_sp);
        if (prof) { (void)hipEventRecord(c->ev[2 * c->prof_n + 1], s); ++c->prof_n; }
        return hipGetLastError() == hipSuccess ? MN_OK : MN_ERR_HIP;
    }
    if (use_sp) {
        bool &dirty_s = c->dirty_sp;
        uint32_t *image = c->packed_sp;
        const int pack_blocks = dirty_s ? sp::PACK_BLOCKS : 0;
        if (pack_blocks) hipLaunchKernelGGL(sp::iqn_split_consts_kernel, dim3(sp::CONST_BLOCKS), dim3(256), 0, s, w, c->consts_sp);
        if (rng_state_dev) {
            long groups = ((long)n * (K_TAUS + 1) + 3) / 4;
            int rng_blocks = (int)((groups + 255) / 256);
            if (rng_blocks > 8 * c->n_cu) rng_blocks = 8 * c->n_cu;
            hipLaunchKernelGGL(sp::iqn_split_prep_kernel, dim3(pack_blocks + rng_blocks), dim3(256), 0, s, w, (const float *)c->consts_sp,
                               image, (const uint64_t *)rng_state_dev, draws_dev, n, cvar_row_dev, cvar, pack_blocks);
            taus_dev = draws_dev;
            explore_u_dev = eps > 0.f ? draws_dev + (size_t)n * K_TAUS : nullptr;
        } else if (pack_blocks) {
            hipLaunchKernelGGL(sp::iqn_split_pack_kernel, dim3(sp::PACK_BLOCKS), dim3(256), 0, s, w, (const float *)c->consts_sp, image);
        }
        dirty_s = false;
        if (quantiles_dev)
            hipLaunchKernelGGL(sp::iqn_qvals_split_kernel<true>, dim3(blocks), dim3(512), sp::LDS_FLOATS * sizeof(float), s, obs_dev, taus_dev,
                               (const uint32_t *)image, qvals_dev, explore_u_dev, eps, actions_dev, n, rng_state_dev, quantiles_dev, (const float *)nullptr);
        else if (late.mask)
            hipLaunchKernelGGL((sp::iqn_qvals_split_kernel<false, false, sp::WAVES, true>), dim3(blocks), dim3(512), sp::LDS_ACT_FLOATS * sizeof(float), s, obs_dev, taus_dev,
                               (const uint32_t *)image, qvals_dev, explore_u_dev, eps, actions_dev, n, rng_state_dev, (float *)nullptr, (const float *)nullptr, late);
        else
            hipLaunchKernelGGL(sp::iqn_qvals_split_kernel<false>, dim3(blocks), dim3(512), sp::LDS_ACT_FLOATS * sizeof(float), s, obs_dev, taus_dev,
                               (const uint32_t *)image, qvals_dev, explore_u_dev, eps, actions_dev, n, rng_state_dev, (float *)nullptr, (const float *)nullptr, sp::LateRows{});
        if (prof) { (void)hipEventRecord(c->ev[2 * c->prof_n + 1], s); ++c->prof_n; }
        return hipGetLastError() == hipSuccess ? MN_OK : MN_ERR_HIP;
    }
    bool &dirty = c->dirty;
    float *packed = c->packed;
    const int pack_blocks = dirty ? PACK_BLOCKS : 0;
    if (rng_state_dev) {
        long groups = ((long)n * (K_TAUS + 1) + 3) / 4;
        int rng_blocks = (int)((groups + 255) / 256);
        if (rng_blocks > 8 * c->n_cu) rng_blocks = 8 * c->n_cu;
        hipLaunchKernelGGL(iqn_prep_kernel, dim3(pack_blocks + rng_blocks), dim3(256), 0, s, w, packed,
                           (const uint64_t *)rng_state_dev, draws_dev, n, cvar_row_dev, cvar, pack_blocks);
        taus_dev = draws_dev;
        explore_u_dev = eps > 0.f ? draws_dev + (size_t)n * K_TAUS : nullptr;
    } else if (pack_blocks) {
        hipLaunchKernelGGL(iqn_pack_kernel, dim3(PACK_BLOCKS), dim3(256), 0, s, w, packed);
    }
    dirty = false;
    if (quantiles_dev)
        hipLaunchKernelGGL(iqn_qvals_kernel<true>, dim3(blocks), dim3(512), LDS_FLOATS * sizeof(float), s, obs_dev, taus_dev,
                           packed, qvals_dev, explore_u_dev, eps, actions_dev, n, rng_state_dev, quantiles_dev);
    else
        hipLaunchKernelGGL(iqn_qvals_kernel<false>, dim3(blocks), dim3(512), LDS_FLOATS * sizeof(float), s, obs_dev, taus_dev,
                           packed, qvals_dev, explore_u_dev, eps, actions_dev, n, rng_state_dev, nullptr);
    if (prof) { (void)hipEventRecord(c->ev[2 * c->prof_n + 1], s); ++c->prof_n; }
    return hipGetLastError() == hipSuccess ? MN_OK : MN_ERR_HIP;
}

extern "C" int mn_iqn_refresh(mn_iqn_ctx *c, const float *const *weights, void *stream) {
    if (!c || !weights) return MN_ERR_INVALID;
    for (int i = 0; i < 14; ++i) if (!weights[i]) return MN_ERR_INVALID;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev != c->device) return MN_ERR_INVALID;
    const IqnWeights w = {weights[0], weights[1], weights[2], weights[3], weights[4], weights[5], weights[6],
                          weights[7], weights[8], weights[9], weights[10], weights[11], weights[12], weights[13]};
    hipStream_t s = (hipStream_t)stream;
    if (c->variant == 2) {
        if (c->dirty_sp) {
            hipLaunchKernelGGL(sp::iqn_split_consts_kernel, dim3(sp::CONST_BLOCKS), dim3(256), 0, s, w, c->consts_sp);
            hipLaunchKernelGGL(sp::iqn_split_pack_kernel, dim3(sp::PACK_BLOCKS), dim3(256), 0, s, w, (const float *)c->consts_sp, c->packed_sp);
            c->dirty_sp = false;
        }
    } else if (c->dirty) {
        hipLaunchKernelGGL(iqn_pack_kernel, dim3(PACK_BLOCKS), dim3(256), 0, s, w, c->packed);
        c->dirty = false;
    }
    return hipGetLastError() == hipSuccess ? MN_OK : MN_ERR_HIP;
}

extern "C" int mn_iqn_act(mn_iqn_ctx *c, const float *obs_dev, const float *taus_dev, const float *const *weights,
                          float *qvals_dev, const float *explore_u_dev, float eps, int32_t *actions_dev,
                          float *quantiles_dev, int32_t n, int32_t num_taus, void *stream) {
    return launch_act(c, obs_dev, taus_dev, weights, qvals_dev, explore_u_dev, eps, actions_dev, quantiles_dev, n, num_taus,
                      nullptr, nullptr, nullptr, 1.0f, stream);
}

extern "C" int mn_iqn_act_rng(mn_iqn_ctx *c, const float *obs_dev, const float *const *weights, uint64_t *rng_state_dev,
                              float *draws_dev, const float *cvar_row_dev, float cvar, float eps, int32_t *actions_dev,
                              float *qvals_dev, float *quantiles_dev, int32_t n, int32_t num_taus, void *stream) {
    if (!rng_state_dev) return MN_ERR_INVALID;
    return launch_act(c, obs_dev, nullptr, weights, qvals_dev, nullptr, eps, actions_dev, quantiles_dev, n, num_taus,
                      rng_state_dev, draws_dev, cvar_row_dev, cvar, stream);
}
