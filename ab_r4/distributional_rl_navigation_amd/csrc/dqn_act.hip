// dqn_act.hip -- the DQN baseline's greedy policy for a whole vector of environments in ONE launch (gfx950).
//
// The reference's DQN agent (run_experiments.py:74-98, 367-376) is its stable-baselines3 fork's `ObsEncoderPolicy`
// (thirdparty/stable_baselines3/common/torch_layers.py:96-135 + dqn/policies.py:48-73): the 26 -> (16 | 16 | 176) encoders WITHOUT
// activation, 208 -> 64 -> 64 -> 9 with ReLUs between (the "features extractor" already ends in 9 values), then sb3's default Q head
// 9 -> 64 -> 64 -> 9, and `argmax`.  Six dense layers per observation, 23 k multiply-adds -- 1 % of the IQN act kernel's work -- which
// eager PyTorch spends eleven small launches on per policy step.  Here a wavefront owns a tile of 16 ENVIRONMENTS (the MFMA's 16 columns)
// and carries it through all layers in registers, exact float32 (v_mfma_f32_16x16x4_f32):
//   * every layer transposed, H^T = W . X^T: weights are the A operand (16 output features x 4 k), activations the B operand (4 k x 16
//     envs), C tile = [16 features x 16 envs] with lane (g, col) holding features 4 g + r of env col;
//   * the k order of a dot product is free, so MFMA step (t, r) of the NEXT layer is defined to consume input features {16 t + 4 g + r}
//     = register r of C tile t in lane group g: a layer's accumulators ARE the next layer's B operands (same trick as iqn_act.hip);
//   * the three encoders are one block-diagonal 208 x 32 matrix (26 inputs zero-padded to 32), so they run on the matrix pipe too;
//   * all weights (125 KB, permuted into A-fragment order by dqn_pack_kernel into a caller-owned image) live in LDS; one ds_read_b128
//     feeds four MFMAs;
//   * epilogue: Q [n][9] (optional) and the greedy action = first maximum, like argmax.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "marinenav_hip.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int OBS = MN_OBS_DIM;      // 26
constexpr int F = 208, H = 64, A = 9;
// stages: the encoders (one block-diagonal matrix) + hidden_layer + hidden_layer_2 + output_layer + q_net.0 + q_net.2 + q_net.4, each as
// (M tiles of 16 outputs, K tiles of 16 inputs).  LDS image (floats): per stage [mt][kt][64 lanes][4 r] weights, then the biases padded
// to multiples of 16
constexpr int N_LAYERS = 7;
constexpr int LM[N_LAYERS] = {13, 4, 4, 1, 4, 4, 1}, LK[N_LAYERS] = {2, 13, 4, 4, 1, 4, 4};
constexpr int lw_off(int l) { int o = 0; for (int i = 0; i < l; ++i) o += LM[i] * LK[i] * 256; return o; }
constexpr int OFF_BIAS = lw_off(N_LAYERS);
constexpr int lb_off(int l) { int o = OFF_BIAS; for (int i = 0; i < l; ++i) o += LM[i] * 16; return o; }
constexpr int IMAGE_FLOATS = lb_off(N_LAYERS);
static_assert(IMAGE_FLOATS * 4 <= 160 * 1024, "the DQN weight image must fit the CU's LDS");
static_assert(IMAGE_FLOATS % 4 == 0, "16-byte copy");

struct DqnWeights {      // device pointers, nn.Linear layout [out][in]; sb3 names in the comment
    const float *ve_w, *ve_b, *ge_w, *ge_b, *se_w, *se_b;      // q_net.features_extractor.{velocity,goal,sensor}_encoder
    const float *h_w, *h_b, *h2_w, *h2_b, *o_w, *o_b;          // ... .hidden_layer, .hidden_layer_2, .output_layer
    const float *q0_w, *q0_b, *q2_w, *q2_b, *q4_w, *q4_b;      // q_net.q_net.{0,2,4}
};

// W_l[out][in] of layer l with the encoders as one block-diagonal [208][32] matrix; zero outside the true shape
__device__ __forceinline__ float layer_weight(const DqnWeights &w, int l, int out, int in) {
    switch (l) {
        case 0:
            if (out < 16) return in < 2 ? w.ve_w[out * 2 + in] : 0.f;
            if (out < 32) return (in >= 2 && in < 4) ? w.ge_w[(out - 16) * 2 + (in - 2)] : 0.f;
            return (out < F && in >= 4 && in < OBS) ? w.se_w[(out - 32) * 22 + (in - 4)] : 0.f;
        case 1: return (out < H && in < F) ? w.h_w[out * F + in] : 0.f;
        case 2: return (out < H && in < H) ? w.h2_w[out * H + in] : 0.f;
        case 3: return (out < A && in < H) ? w.o_w[out * H + in] : 0.f;
        case 4: return (out < H && in < A) ? w.q0_w[out * A + in] : 0.f;
        case 5: return (out < H && in < H) ? w.q2_w[out * H + in] : 0.f;
        default: return (out < A && in < H) ? w.q4_w[out * H + in] : 0.f;
    }
}
__device__ __forceinline__ float layer_bias(const DqnWeights &w, int l, int out) {
    switch (l) {
        case 0: return out < 16 ? w.ve_b[out] : (out < 32 ? w.ge_b[out - 16] : (out < F ? w.se_b[out - 32] : 0.f));
        case 1: return out < H ? w.h_b[out] : 0.f;
        case 2: return out < H ? w.h2_b[out] : 0.f;
        case 3: return out < A ? w.o_b[out] : 0.f;
        case 4: return out < H ? w.q0_b[out] : 0.f;
        case 5: return out < H ? w.q2_b[out] : 0.f;
        default: return out < A ? w.q4_b[out] : 0.f;
    }
}

// image[lw_off(l) + ((mt KT + t) 64 + lane) 4 + r] = W_l[16 mt + (lane & 15)][16 t + 4 (lane >> 4) + r]
__global__ __launch_bounds__(256) void dqn_pack_kernel(DqnWeights w, float *__restrict__ image) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= IMAGE_FLOATS) return;
    if (i >= OFF_BIAS) {
        int l = 0, k = i - OFF_BIAS;
        while (k >= LM[l] * 16) { k -= LM[l] * 16; ++l; }
        image[i] = layer_bias(w, l, k);
        return;
    }
    int l = 0, k = i;
    while (k >= LM[l] * LK[l] * 256) { k -= LM[l] * LK[l] * 256; ++l; }
    const int r = k & 3, lane = (k >> 2) & 63, q = k >> 8, t = q % LK[l], mt = q / LK[l];
    image[i] = layer_weight(w, l, 16 * mt + (lane & 15), 16 * t + 4 * (lane >> 4) + r);
}

// out[mt] = act(W_l in + b_l) for one 16-env tile: in[t][r] = input feature 16 t + 4 g + r of env col
template <int L, bool RELU, int MT, int KT>
__device__ __forceinline__ void dense(const float *__restrict__ lds, int lane, const f32x4 (&in)[KT], f32x4 (&out)[MT]) {
    const f32x4 *w4 = reinterpret_cast<const f32x4 *>(lds + lw_off(L)) + lane;
    const f32x4 *b4 = reinterpret_cast<const f32x4 *>(lds + lb_off(L)) + (lane >> 4);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        f32x4 acc = b4[4 * mt];      // bias[16 mt + 4 g + r]: the accumulator's initial value
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            const f32x4 a = w4[(mt * KT + t) * 64];
#pragma unroll
            for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], in[t][r], acc, 0, 0, 0);
        }
        if (RELU) { acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f); }
        out[mt] = acc;
    }
}

__global__ __launch_bounds__(512) void dqn_qvals_kernel(const float *__restrict__ obs, const float *__restrict__ image, float *__restrict__ qvals,
                                                        int32_t *__restrict__ actions, int n) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(image);
        f32x4 *dst = reinterpret_cast<f32x4 *>(lds);
        for (int i = threadIdx.x; i < IMAGE_FLOATS / 4; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, g = lane >> 4, col = lane & 15, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const int n_tiles = (n + 15) / 16;
    for (int tile = blockIdx.x * waves + wave; tile < n_tiles; tile += gridDim.x * waves) {
        const int e = tile * 16 + col;
        const bool live = e < n;
        // observation row of env `col`, features 16 t + 4 g + r (26 inputs, zero-padded to 32)
        f32x4 x0[2];
        const float *row = obs + (size_t)(live ? e : 0) * OBS;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = 16 * t + 4 * g + r;
                x0[t][r] = (live && k < OBS) ? row[k < OBS ? k : 0] : 0.f;
            }
        f32x4 f[13], h1[4], h2[4], o[1], q1[4], q2[4], q[1];
        dense<0, false, 13, 2>(lds, lane, x0, f);        // the three encoders, no activation (torch_layers.py:125-128)
        dense<1, true, 4, 13>(lds, lane, f, h1);         // hidden_layer + ReLU
        dense<2, true, 4, 4>(lds, lane, h1, h2);         // hidden_layer_2 + ReLU
        dense<3, false, 1, 4>(lds, lane, h2, o);         // output_layer: the extractor's 9 "features" (rows 9..15 are zero)
        dense<4, true, 4, 1>(lds, lane, o, q1);          // q_net.0 + ReLU
        dense<5, true, 4, 4>(lds, lane, q1, q2);         // q_net.2 + ReLU
        dense<6, false, 1, 4>(lds, lane, q2, q);         // q_net.4: Q(s, a), lane (g, col) holds actions 4 g + r of env col
        if (qvals && live)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * g + r < A) qvals[(size_t)e * A + 4 * g + r] = q[0][r];
        if (actions) {      // first maximum over the 9 actions: in-lane over r, then across the four lane groups
            float best = -INFINITY;
            int arg = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int a_idx = 4 * g + r;
                const float v = a_idx < A ? q[0][r] : -INFINITY;
                if (v > best) { best = v; arg = a_idx; }
            }
#pragma unroll
            for (int off = 16; off < 64; off <<= 1) {
                const float ob = __shfl_xor(best, off);
                const int oa = __shfl_xor(arg, off);
                if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
            }
            if (live && g == 0) actions[e] = arg;
        }
    }
}

}  // namespace

extern "C" int64_t mn_dqn_image_floats(void) { return IMAGE_FLOATS; }

extern "C" int mn_dqn_act(const float *obs_dev, const float *const *weights, float *image_dev, int32_t repack, float *qvals_dev,
                          int32_t *actions_dev, int32_t n, void *stream) {
    if (!obs_dev || !weights || !image_dev || (!qvals_dev && !actions_dev) || n <= 0) return MN_ERR_INVALID;
    for (int i = 0; i < 18; ++i) if (!weights[i]) return MN_ERR_INVALID;
    static bool attr_set[64] = {false};
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64 || hipGetDeviceProperties(&prop, dev) != hipSuccess) return MN_ERR_NO_DEVICE;
    if (!attr_set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(dqn_qvals_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                IMAGE_FLOATS * (int)sizeof(float)) != hipSuccess) return MN_ERR_HIP;
        attr_set[dev] = true;
    }
    const DqnWeights w = {weights[0], weights[1], weights[2], weights[3], weights[4], weights[5], weights[6], weights[7], weights[8],
                          weights[9], weights[10], weights[11], weights[12], weights[13], weights[14], weights[15], weights[16], weights[17]};
    hipStream_t s = (hipStream_t)stream;
    if (repack) hipLaunchKernelGGL(dqn_pack_kernel, dim3((IMAGE_FLOATS + 255) / 256), dim3(256), 0, s, w, image_dev);
    const int n_tiles = (n + 15) / 16;
    int blocks = (n_tiles + 7) / 8;
    if (blocks > prop.multiProcessorCount) blocks = prop.multiProcessorCount;
    hipLaunchKernelGGL(dqn_qvals_kernel, dim3(blocks), dim3(512), IMAGE_FLOATS * sizeof(float), s, obs_dev, (const float *)image_dev, qvals_dev,
                       actions_dev, n);
    return hipGetLastError() == hipSuccess ? MN_OK : MN_ERR_HIP;
}
