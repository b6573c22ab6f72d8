// iqn_act_split.h -- the IQN action-value network of iqn_act.hip on the f16 matrix pipe at float32 accuracy.
// Included by iqn_act.hip inside its anonymous namespace (uses IqnWeights, row_sum16, draw_block, the layer constants).
//
// Why: the exact-f32 MFMA (v_mfma_f32_16x16x4_f32) runs at the f32 VECTOR rate, 1/16 of the f16 rate, and the act kernel
// built on it sits at 84 % of that peak -- the only way to go substantially faster is to leave that
// pipe.  Here every f32 operand x is split into two f16 pieces, hi = RNE16(x), lo = RNE16(x - hi) (the subtraction is exact,
// so x = hi + lo + e with |e| <= 2^-22 |x| in the worst case and ~2^-24.4 |x| rms, as long as lo stays above the f16 subnormal
// floor), and a product of two operands is accumulated as lo.hi + hi.lo + hi.hi on v_mfma_f32_16x16x32_f16: f16 x f16 products
// are exact in the f32 accumulator, the dropped lo.lo term and the two e terms are each <= 2^-22 relative (worst case; random in
// sign, ~2^-24 typical).  That is the error class of one f32 rounding per product -- whose accumulated effect over a K = 64..224
// dot product is what the exact-f32 MFMA has too; measured against a float64 reference the three-product scheme is as close as the exact-f32 MFMA
// on these layer shapes (profiles/r02_f16_split_probe.txt: rms error 1.7e-7 vs 1.95e-7 relative, K = 224), f16 subnormal
// inputs are not flushed by the matrix pipe, and three f16 MFMAs cost 3/16 of the f32 instruction they replace
// (2057 vs 145.5 TFLOP/s sustained in the same probe).
//
// Range.  f16 tops out at 65504, so operands are scaled by powers of two (exact) before the split and the accumulators
// are unscaled in the layer epilogues (layers 2, 3: one v_pk_fma that also adds the bias; layer 1: its bias, pre-scaled, is
// the accumulators' initial value and its 2^-k1 is folded into the Hadamard multiplier, so that epilogue is ReLU + multiply):
//   * weights of layer l: 2^k_l with max |W_l| 2^k_l in [2^14, 2^15)                    (static, part of the weight image);
//   * cos embedding: in [-1, 1], not scaled (an f16 pair keeps an absolute error of 2^-25 for ANY |x| <= 1, i.e. 2^-25 of
//     the largest element, which is what a dot product's error is measured against);
//   * hidden activations: one power of two PER LAYER AND ENVIRONMENT, S_l, chosen from a guaranteed bound, not from the data:
//         |h1_j| <= B1_j |f_j|,  B1_j = sum_k |W1_jk| + |b1_j|     (|cos| <= 1),      m1 = max_j B1_j |f_j|   (per env)
//         |h2_i| <= M2 = R2 m1 + beta2,  R2 = max_i sum_j |W2_ij|,  beta2 = max |b2|;   |h3| <= M3 = R3 M2 + beta3
//     and S_l = 2^(14 - floor(log2 M_l)), so S_l |h_l| < 2^15 always: no overflow for any weights and any observation.  Layer
//     l + 1 receives S_l h_l; its accumulator (which carries S_l 2^k) is brought to S_(l+1) by the per-environment factor
//     2^-k S_(l+1) / S_l in the v_fma that adds the bias S_(l+1) b -- no extra instruction.  Each bound is conservative only by
//     its own row-sum slack (2^3..2^7 here), which costs nothing as long as the largest activation of a layer stays above 2^-1
//     after scaling (17 binades of slack), see the probe's "scale 2^-10 lower" rows and tests/test_act_split_scheme_cpu.py.
// The observation encoders, the tau-mean and the 9 x 64 output layer stay f32 VALU work as in the exact kernel.
//
// Schedule.  With the matrix work cut 4.7 x the kernel is bound by instruction issue: a SIMD issues about one instruction per
// 5 cycles in total over its two waves once VALU / LDS work is mixed with MFMAs (scripts/probes/mfma_valu_overlap.hip), and a
// wave's other instructions only hide under its MFMAs when they sit BETWEEN them.  The layers are therefore written as
// hand-interleaved pipeline stages (stage<B>, tail): every slot is one MFMA plus a few VALU / LDS instructions of an
// independent piece of work, pinned with sched_barrier(0).  What remains is the instruction count itself (per environment:
// 372 MFMA + ~960 VALU + ~195 LDS + ~170 scalar), ~0.15 us per instruction per 65 536-env launch.
//
// Layout.  Same transposed, register-chained scheme as the exact kernel: weights are the A operand (16 output features x
// 32 k, lane (g, row) holds k slots 8 g + i), activations the B operand (32 k x 16 taus), C tile [16 features x 16 taus]
// with lane (g, col) holding features 4 g + r.  K block b of the NEXT layer consumes C tiles 2b and 2b + 1: k slot
// (g, i) := feature 16 (2b + (i >> 2)) + 4 g + (i & 3), i.e. exactly registers 0..3 of the two tiles in lane group g --
// a layer's accumulators, split in place, ARE the next layer's B operands.  Layer 2's K = 208 is 6.5 blocks: the 7th
// block's upper half is zero (weights and activations).
//
// act_eval's per-tau quantile output (QUANT = true) runs the output layer on the matrix pipe as well.

#ifndef SP_COSJOB
#define SP_COSJOB 1      // the next environment's cos embedding inside stage 5 / the tail (CosJob)
#endif
#ifndef SP_ABL
#define SP_ABL 0      // measurement builds only (scripts/act_split_ablation.sh): 1 no operand split, 2 no weight LDS reads, 4 no ReLU / Hadamard, 8 no cos, 16 only the hi.hi products, 64 s_memtime phase timing (printf)
#endif

namespace sp {

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int KB2 = 7;                                  // layer-2 K blocks (208 -> 224)
// weight image, in 16-byte units (8 halves = one lane's A operand): [.. tile ..][piece: hi, lo][64 lanes]
constexpr int W1_U4 = 0;                                // [13 mt][2 kb]
constexpr int W2_U4 = W1_U4 + T1 * 2 * 2 * 64;          // [4 mt][7 kb]
constexpr int W3_U4 = W2_U4 + 4 * KB2 * 2 * 64;         // [4 mt][2 kb]
constexpr int END_U4 = W3_U4 + 4 * 2 * 2 * 64;
// float part (indices in floats from the start of the image)
constexpr int OFF_W4 = END_U4 * 4;                      // [4 t2][64 l][4 r] f32 output layer, as in the exact kernel
constexpr int OFF_B1 = OFF_W4 + 4 * 64 * 4;             // [208]
constexpr int OFF_B2 = OFF_B1 + F;                      // [64]
constexpr int OFF_B3 = OFF_B2 + H;                      // [64]
constexpr int OFF_B4 = OFF_B3 + H;                      // [16]
constexpr int OFF_BND = OFF_B4 + 16;                    // [208] B1_j
constexpr int OFF_CST = OFF_BND + F;                    // [16] c1 c2 c3 (accumulator unscale), a2 d2 a3 d3 (activation bounds)
constexpr int OFF_WS = OFF_CST + 16;                    // [6 i4][176 sf][4] sensor encoder, inputs 4 + 4 i4 + c
constexpr int OFF_WVG = OFF_WS + 6 * 176 * 4;           // [32 f][2] velocity / goal encoders
constexpr int OFF_BE = OFF_WVG + 64;                    // [208] encoder biases
constexpr int OFF_W4H = OFF_BE + F;                     // [2 kb][piece][64 l] x 8 halves: output layer as an MFMA A operand (act_eval's per-tau quantiles)
constexpr int OFF_FB = OFF_W4H + 2 * 2 * 64 * 4;        // [waves][208] per-wave scaled features
constexpr int WAVES = 8;                                // one 512-thread workgroup per CU, two waves per SIMD
constexpr int LDS_FLOATS = OFF_FB + WAVES * F;
constexpr int LDS_ACT_FLOATS = OFF_W4H + WAVES * F;       // acting form (no quantile output): the feature buffers take the place of the output layer's MFMA operands
static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS image of the split-f16 act kernel must fit the CU's 160 KB");
static_assert(OFF_WS % 4 == 0 && OFF_WVG % 4 == 0 && OFF_BE % 4 == 0 && OFF_FB % 4 == 0 && OFF_CST % 4 == 0 && OFF_W4H % 4 == 0, "16-byte aligned blocks");
constexpr int PACK_BLOCKS = (OFF_FB + 255) / 256;       // one thread per 32-bit word of the image
constexpr int N_CONST = 16;

// power of two p with amax * p in [2^14, 2^15)  (amax > 0 finite); 1 for degenerate input
__device__ __forceinline__ float pow2_to_2p15(float amax) {
    if (!(amax > 1e-30f) || !(amax < 1e30f)) return 1.0f;
    const int e = (int)(__builtin_bit_cast(uint32_t, amax) >> 23);       // amax in [2^(e-127), 2^(e-126))
    return __builtin_bit_cast(float, (uint32_t)(268 - e) << 23);         // 2^(141 - e)
}

// consts[0..2] = 2^k_l (weight scales), [3..5] = 2^-k_l, [6] = R2, [7] = beta2, [8] = R3 R2, [9] = R3 beta2 + beta3,
// [10], [11] = 2^k4, 2^-k4 (output layer)
// (bounds inflated by 2^-10 relative against the rounding of the sums).  Runs after every optimizer step, in front of the act kernel.
// Round 3: CONST_BLOCKS workgroups instead of one of 1 024 threads (14.4 us: a single CU pulling 30 000 weights other CUs just wrote) --
// block b takes a 1 / 16 slice of every matrix for the maxima and rows 4 b .. 4 b + 3 of W2 and W3 for the row sums (one wavefront
// per row, fixed xor tree); its eight partial results go to consts[N_CONST + 8 b ..], and the block that finishes LAST (ticket at
// consts[N_CONST + 8 CONST_BLOCKS]; maxima, so the order of arrival cannot matter) combines them.  The buffer is N_CONST_BUF floats,
// zero-filled once.
constexpr int CONST_BLOCKS = 16;
constexpr int N_CONST_BUF = N_CONST + 8 * CONST_BLOCKS + 4;
__global__ __launch_bounds__(256) void iqn_split_consts_kernel(IqnWeights w, float *__restrict__ consts) {
    __shared__ float red[4][8];
    __shared__ int s_last;
    const int tid = threadIdx.x, b = blockIdx.x, lane = tid & 63, wv = tid >> 6;
    // v[0..2] = max |W1|, |W2|, |W3| over this block's slices;  v[3], v[4] = row sums of W2, W3 (row = 4 b + wave);  v[5], v[6] = |b2|, |b3|;
    // v[7] = max |W4| (the 32x32 kernel runs the output layer on the matrix pipe)
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    constexpr int S12 = F * N_COS / CONST_BLOCKS, S3 = H * H / CONST_BLOCKS;      // 832, 256
    static_assert(F * N_COS % CONST_BLOCKS == 0 && H * F == F * N_COS && H * H % CONST_BLOCKS == 0 && H == 4 * CONST_BLOCKS, "slices");
    for (int i = tid; i < S12; i += 256) { v[0] = fmaxf(v[0], fabsf(w.W1[b * S12 + i])); v[1] = fmaxf(v[1], fabsf(w.W2[b * S12 + i])); }
    if (tid < S3) v[2] = fabsf(w.W3[b * S3 + tid]);
    const int row = 4 * b + wv;
    for (int j = lane; j < F; j += 64) v[3] += fabsf(w.W2[row * F + j]);
    v[4] = fabsf(w.W3[row * H + lane]);
    if (b == 0 && tid < H) { v[5] = fabsf(w.b2[tid]); v[6] = fabsf(w.b3[tid]); }
    for (int i = b * 256 + tid; i < A_OUT * H; i += 256 * CONST_BLOCKS) v[7] = fmaxf(v[7], fabsf(w.W4[i]));
    // the row sums first (within the wavefront), then all eight as maxima through the same shuffle rounds and one pass through LDS
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { v[3] += __shfl_xor(v[3], off); v[4] += __shfl_xor(v[4], off); }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (q != 3 && q != 4) v[q] = fmaxf(v[q], __shfl_xor(v[q], off));
    if (lane == 0)
#pragma unroll
        for (int q = 0; q < 8; ++q) red[wv][q] = v[q];
    __syncthreads();
    float *part = consts + N_CONST;
    unsigned *ticket = reinterpret_cast<unsigned *>(consts + N_CONST + 8 * CONST_BLOCKS);
    if (tid < 8) {
        const float r = fmaxf(fmaxf(red[0][tid], red[1][tid]), fmaxf(red[2][tid], red[3][tid]));
        __hip_atomic_store(part + 8 * b + tid, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // agent scope: read by another block below
        __threadfence();
    }
    __syncthreads();
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        s_last = old == CONST_BLOCKS - 1;
        if (s_last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!s_last) return;
    if (tid < 8) {
        float r = 0.f;
        for (int k = 0; k < CONST_BLOCKS; ++k) r = fmaxf(r, __hip_atomic_load(part + 8 * k + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        red[0][tid] = r;
    }
    __syncthreads();
    if (tid == 0) {
        const float *r = red[0];
        const float s1 = pow2_to_2p15(r[0]), s2 = pow2_to_2p15(r[1]), s3 = pow2_to_2p15(r[2]), infl = 1.0009765625f;
        const float r2 = r[3], r3 = r[4], b2m = r[5], b3m = r[6];
        consts[0] = s1; consts[1] = s2; consts[2] = s3;
        consts[3] = 1.0f / s1; consts[4] = 1.0f / s2; consts[5] = 1.0f / s3;      // exact: powers of two
        consts[6] = r2 * infl; consts[7] = b2m * infl;
        consts[8] = r3 * r2 * infl * infl; consts[9] = (r3 * b2m * infl + b3m) * infl;
        const float s4 = pow2_to_2p15(r[7]);
        consts[10] = s4; consts[11] = 1.0f / s4;                                  // output layer scale (32x32 kernel)
        for (int i = 12; i < N_CONST; ++i) consts[i] = 0.f;
    }
}

// the f32 weight behind A-operand k slot (g, i8) of [layer][mt][kb], row `row`
__device__ __forceinline__ float split_weight(const IqnWeights &w, int layer, int mt, int kb, int g, int row, int i8) {
    if (layer == 1) return w.W1[(16 * mt + row) * N_COS + 32 * kb + 8 * g + i8];
    const int feat = 16 * (2 * kb + (i8 >> 2)) + 4 * g + (i8 & 3);
    if (layer == 2) return feat < F ? w.W2[(16 * mt + row) * F + feat] : 0.f;
    return w.W3[(16 * mt + row) * H + feat];
}

__device__ __forceinline__ uint32_t half_bits(_Float16 h) { return (uint32_t)__builtin_bit_cast(uint16_t, h); }

// 32-bit word i of the image
__device__ __forceinline__ uint32_t pack_word(const IqnWeights &w, const float *__restrict__ consts, int i) {
    if (i < OFF_W4) {
        const int u4 = i >> 2, pair = i & 3, lane = u4 & 63, piece = (u4 >> 6) & 1, g = lane >> 4, row = lane & 15;
        int q = u4 >> 7, layer, mt, kb;
        if (q < T1 * 2) { layer = 1; mt = q >> 1; kb = q & 1; }
        else if (q < T1 * 2 + 4 * KB2) { q -= T1 * 2; layer = 2; mt = q / KB2; kb = q % KB2; }
        else { q -= T1 * 2 + 4 * KB2; layer = 3; mt = q >> 1; kb = q & 1; }
        const float sc = consts[layer - 1];
        uint32_t out = 0;
        for (int j = 0; j < 2; ++j) {
            const float x = split_weight(w, layer, mt, kb, g, row, 2 * pair + j) * sc;
            const _Float16 hi = (_Float16)x;
            const _Float16 v = piece == 0 ? hi : (_Float16)(x - (float)hi);
            out |= half_bits(v) << (16 * j);
        }
        return out;
    }
    if (i >= OFF_W4H) {              // output layer, A-operand order of layer 3's K blocks, scaled by 2^k4 and split like the others
        const int k = i - OFF_W4H, u4 = k >> 2, pair = k & 3, lane = u4 & 63, piece = (u4 >> 6) & 1, kb = u4 >> 7, g = lane >> 4, row = lane & 15;
        uint32_t out = 0;
        for (int j = 0; j < 2; ++j) {
            const int i8 = 2 * pair + j, feat = 16 * (2 * kb + (i8 >> 2)) + 4 * g + (i8 & 3);
            const float x = (row < A_OUT ? w.W4[row * H + feat] : 0.f) * consts[10];
            const _Float16 hi = (_Float16)x;
            const _Float16 v16 = piece == 0 ? hi : (_Float16)(x - (float)hi);
            out |= half_bits(v16) << (16 * j);
        }
        return out;
    }
    float v;
    if (i < OFF_B1) {                // W4p[t2][l][r] = W4[l & 15][16 t2 + 4 (l >> 4) + r] (rows >= 9 are zero)
        const int k = i - OFF_W4, r = k & 3, l = (k >> 2) & 63, t2 = k >> 8;
        v = (l & 15) < A_OUT ? w.W4[(l & 15) * H + 16 * t2 + 4 * (l >> 4) + r] : 0.f;
    } else if (i < OFF_B2) v = w.b1[i - OFF_B1] * consts[0];    // pre-scaled like W1: it is the layer-1 accumulator's initial value
    else if (i < OFF_B3) v = w.b2[i - OFF_B2];
    else if (i < OFF_B4) v = w.b3[i - OFF_B3];
    else if (i < OFF_BND) v = (i - OFF_B4) < A_OUT ? w.b4[i - OFF_B4] : 0.f;
    else if (i < OFF_CST) {          // B1_j = sum_k |W1_jk| + |b1_j|, inflated against the rounding of the sum
        const int j = i - OFF_BND;
        float s = fabsf(w.b1[j]);
        for (int k = 0; k < N_COS; ++k) s += fabsf(w.W1[j * N_COS + k]);
        v = s * 1.0009765625f;
    } else if (i < OFF_WS) {         // CST[j] = consts[3 + j]: c1 c2 c3 a2 d2 a3 d3 2^k4 2^-k4
        const int j = i - OFF_CST;
        v = 3 + j < N_CONST ? consts[3 + j] : 0.f;
    }
    else if (i < OFF_WVG) {          // WS[i4][sf][c] = se_w[sf][4 i4 + c] (22 inputs, zero padded to 24)
        const int k = i - OFF_WS, c = k & 3, sf = (k >> 2) % 176, i4 = (k >> 2) / 176, inp = 4 * i4 + c;
        v = inp < 22 ? w.se_w[sf * 22 + inp] : 0.f;
    } else if (i < OFF_BE) {         // WVG[f][c]
        const int k = i - OFF_WVG, c = k & 1, f = k >> 1;
        v = f < 16 ? w.ve_w[f * 2 + c] : w.ge_w[(f - 16) * 2 + c];
    } else {
        const int f = i - OFF_BE;
        v = f < 16 ? w.ve_b[f] : (f < 32 ? w.ge_b[f - 16] : w.se_b[f - 32]);
    }
    return __builtin_bit_cast(uint32_t, v);
}

__global__ __launch_bounds__(256) void iqn_split_pack_kernel(IqnWeights w, const float *__restrict__ consts, uint32_t *__restrict__ packed) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < OFF_FB) packed[i] = pack_word(w, consts, i);
}

// weight image (when stale; consts from iqn_split_consts_kernel earlier in the stream) + the call's random numbers
__global__ __launch_bounds__(256) void iqn_split_prep_kernel(IqnWeights w, const float *__restrict__ consts, uint32_t *__restrict__ packed,
                                                             const uint64_t *__restrict__ rng_state, float *__restrict__ draws, int n,
                                                             const float *__restrict__ cvar_row, float cvar, int pack_blocks) {
    if ((int)blockIdx.x < pack_blocks) {
        const int i = blockIdx.x * blockDim.x + threadIdx.x;
        if (i < OFF_FB) packed[i] = pack_word(w, consts, i);
        return;
    }
    draw_block(rng_state, draws, n, cvar_row, cvar, pack_blocks);
}

__device__ __forceinline__ f32x4 mf(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// (x, y) -> hi pair, lo pair: v_cvt_pk_f16_f32, 2 x v_fma_mix_f32 (x - hi, exact; written as asm because the SLP vectoriser
// otherwise turns the pair into 2 x v_cvt_f32_f16 + v_pk_fma_f32), v_cvt_pk_f16_f32
__device__ __forceinline__ void split2(float x, float y, f16x2 &h, f16x2 &l) {
    const f32x2 v = {x, y};
    h = __builtin_convertvector(v, f16x2);
    const uint32_t hb = __builtin_bit_cast(uint32_t, h);
    f32x2 r;
    asm("v_fma_mix_f32 %0, 1.0, %1, -%2 op_sel_hi:[0,0,1]" : "=v"(r[0]) : "v"(x), "v"(hb));
    asm("v_fma_mix_f32 %0, 1.0, %1, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r[1]) : "v"(y), "v"(hb));
    l = __builtin_convertvector(r, f16x2);
}

__device__ __forceinline__ f16x8 cat4(f16x2 a, f16x2 b, f16x2 c, f16x2 d) {
    const f16x4 ab = __builtin_shufflevector(a, b, 0, 1, 2, 3), cd = __builtin_shufflevector(c, d, 0, 1, 2, 3);
    return __builtin_shufflevector(ab, cd, 0, 1, 2, 3, 4, 5, 6, 7);
}

// two C tiles (registers of one lane) -> the hi / lo B operands of the K block they form
__device__ __forceinline__ void split_tiles(f32x4 t0, f32x4 t1, f16x8 &bh, f16x8 &bl) {
    f16x2 h0, h1, h2, h3, l0, l1, l2, l3;
    split2(t0[0], t0[1], h0, l0); split2(t0[2], t0[3], h1, l1);
    split2(t1[0], t1[1], h2, l2); split2(t1[2], t1[3], h3, l3);
    bh = cat4(h0, h1, h2, h3); bl = cat4(l0, l1, l2, l3);
}

// max(x, 0) as ONE instruction.  Written as a float compare / select (or fmaxf, or med3), a ReLU whose input is a raw MFMA
// result gets a second v_max_f32 x, x, x in front of it (sNaN canonicalisation of an operand the compiler cannot prove
// canonical).  On the bit patterns it is an integer max: negative floats are negative ints, non-negative floats order
// like ints.  (Not inline asm: the MFMA -> VALU read hazard is software-managed and the compiler only counts wait states
// for instructions it knows.)
__device__ __forceinline__ float relu1(float x) {
    const int b = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, b > 0 ? b : 0);
}
__device__ __forceinline__ f32x4 relu4s(f32x4 v) { return (f32x4){relu1(v.x), relu1(v.y), relu1(v.z), relu1(v.w)}; }

__device__ __forceinline__ f32x4 fma4(f32x4 a, float c, f32x4 b) {
    f32x4 r;
    r.x = fmaf(a.x, c, b.x); r.y = fmaf(a.y, c, b.y); r.z = fmaf(a.z, c, b.z); r.w = fmaf(a.w, c, b.w);
    return r;
}

// sums over the 16 lanes of a row for 16 values at once, step by step over all values: the DPP steps of one value depend on each
// other (and a DPP read of a register just written needs wait states), the 16 values do not
__device__ __forceinline__ void row_sum16_x16(float (&v)[16]) {
#define SP_DPP_STEP(ctrl)                                                                                                        \
    _Pragma("unroll") for (int i = 0; i < 16; ++i)                                                                              \
        v[i] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[i]), ctrl, 0xF, 0xF, true));
    SP_DPP_STEP(0xB1)      // quad xor 1
    SP_DPP_STEP(0x4E)      // quad xor 2
    SP_DPP_STEP(0x141)     // row_half_mirror
    SP_DPP_STEP(0x140)     // row_mirror
#undef SP_DPP_STEP
}
// sum over the four 16-lane rows of a wave, result in every lane: two v_permlane*_swap (VALU, no LDS round trip)
__device__ __forceinline__ float sum_rows4(float x) {
    float a = x, b = x;
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));     // a = rows [0 0 2 2], b = rows [1 1 3 3]
    float c = a + b, d = c;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(c), "+v"(d));     // c = halves [lo lo], d = halves [hi hi]
    return c + d;
}

// max over the wave of a non-negative value (uniform result)
__device__ __forceinline__ float wave_max_nonneg(float v) {
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true)));
    const int iv = __builtin_bit_cast(int, v);
    const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)), b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
    const float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)), d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
    return fmaxf(fmaxf(a, b), fmaxf(c, d));
}

constexpr int NT = 2;   // one environment = 32 tau rows = 2 column tiles per wave iteration

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(<N - 1>) -- the pipeline slots below are unrolled by
// construction (a 48- or 72-slot `#pragma unroll` body exceeds the unroller's size limit and falls back to a real loop with
// dynamically indexed -- scratch -- register arrays)
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// LDS addressing.  The image is 154 KB and a ds_read carries a 16-bit byte offset, so every read is written as
// (opaque base register) + (compile-time constant < 64 KB): four bases cover the image.  Left to itself the compiler
// materialises one address VGPR per read outside the environment loop (~70 registers), which caps the occupancy.
struct LdsBase {
    int w_lo;     // lane                       : 16-byte units [0, 4096)
    int w_hi;     // lane + 4096                : 16-byte units [4096, 8192)  (rest of W2, W3, W4)
    int fl;       // (OFF_B1 >> 2) + g          : biases, bounds (16-byte units, indexed by lane group)
    int fb;       // this wave's feature buffer + g (16-byte units)
};
__device__ __forceinline__ u32x4 ld_w(const u32x4 *__restrict__ lds4, const LdsBase &lb, int c) {     // c: unit index without the lane
    if (SP_ABL & 2) return (u32x4){(uint32_t)lb.w_lo, (uint32_t)lb.w_hi, (uint32_t)lb.fl, (uint32_t)lb.fb};
    return c < 4096 - 64 ? lds4[lb.w_lo + c] : lds4[lb.w_hi + (c - 4096)];
}

// layer-1 MFMAs of layer-2 K block b (feature tiles 2b, 2b + 1; only 2b for the last block): 3 products x 2 cos K blocks
// ---- the fused layer 1 + layer 2 pipeline; layer-2 K block B = feature tiles 2B, 2B + 1 (the last block has one)
constexpr int ntiles_of(int B) { return (2 * B + 1 < T1) ? 2 : 1; }

// The split of one register pair in three schedulable pieces (see stage()).
__device__ __forceinline__ f16x2 cvt_pair(float x, float y) {
    if (SP_ABL & 1) return __builtin_bit_cast(f16x2, x);
    const f32x2 v = {x, y};
    return __builtin_convertvector(v, f16x2);
}
__device__ __forceinline__ void residual_pair(float x, float y, f16x2 h, float &rx, float &ry) {   // x - hi, exact
    const uint32_t hb = __builtin_bit_cast(uint32_t, h);
    if (SP_ABL & 1) { rx = y; ry = x; return; }
    asm("v_fma_mix_f32 %0, 1.0, %1, -%2 op_sel_hi:[0,0,1]" : "=v"(rx) : "v"(x), "v"(hb));
    asm("v_fma_mix_f32 %0, 1.0, %1, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(ry) : "v"(y), "v"(hb));
}

// The NEXT environment's cos embedding (layer-1 B operands: 32 values per lane -> 16 register pairs, each split into a hi and a lo pair)
// as 80 pieces of 1-3 plain VALU instructions that stage 5 and the tail of the CURRENT environment issue between their matrix instructions
// (round 3).  Those two have spare slots -- stage 5 has no layer-1 work, the tail's last 24 slots have no epilogue -- and the cos operands of
// the current environment are dead after stage 4, so the results need no second set of registers.  Beside the matrix pipe a plain VALU
// instruction costs ~1.6 cycles and a v_cos_f32 ~5 (profiles/r03_valu_cost_probe.txt); as a phase of their own, in front of the encoders,
// the same 176 instructions took ~830 of an environment's ~13 600 clock ticks.
struct CosJob {
    float tau[NT];          // the next environment's taus of this lane
    float hk0;
    float x0, x1, r0, r1;
    f16x2 hc;
    f16x2 H[16], L[16];     // unit u = (kb NT + nt) 4 + p: cos(tau[nt] (hk0 + 16 kb + p)), cos(tau[nt] (hk0 + 16 kb + p + 0.5))
    template <int P>
    __device__ __forceinline__ void piece() {
        if constexpr (P >= 0 && P < 80) {
            constexpr int u = P / 5, part = P % 5, kb = u / (4 * NT), nt = (u / 4) % NT, p4 = u % 4;
            if constexpr (part == 0) {
                x0 = tau[nt] * (hk0 + (16.0f * kb + 0.5f * (2 * p4)));
                x1 = tau[nt] * (hk0 + (16.0f * kb + 0.5f * (2 * p4 + 1)));
            } else if constexpr (part == 1) {
                if (!(SP_ABL & 8)) x0 = __builtin_amdgcn_cosf(x0);
            } else if constexpr (part == 2) {
                if (!(SP_ABL & 8)) x1 = __builtin_amdgcn_cosf(x1);
            } else if constexpr (part == 3) {
                hc = cvt_pair(x0, x1);
                residual_pair(x0, x1, hc, r0, r1);
                H[u] = hc;
            } else {
                L[u] = cvt_pair(r0, r1);
            }
        }
    }
    __device__ __forceinline__ void finish(f16x8 (&cbh)[2][NT], f16x8 (&cbl)[2][NT]) const {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int u = (kb * NT + nt) * 4;
                cbh[kb][nt] = cat4(H[u], H[u + 1], H[u + 2], H[u + 3]);
                cbl[kb][nt] = cat4(L[u], L[u + 1], L[u + 2], L[u + 3]);
            }
    }
};

// the two halves of residual_pair as separate instructions (the epilogue is issued in pieces of two instructions, see stage())
__device__ __forceinline__ float residual_lo32(float x, f16x2 h) {      // x - h.x, exact
    const uint32_t hb = __builtin_bit_cast(uint32_t, h);
    float r;
    if (SP_ABL & 1) return x;
    asm("v_fma_mix_f32 %0, 1.0, %1, -%2 op_sel_hi:[0,0,1]" : "=v"(r) : "v"(x), "v"(hb));
    return r;
}
__device__ __forceinline__ float residual_hi32(float y, f16x2 h) {      // y - h.y, exact
    const uint32_t hb = __builtin_bit_cast(uint32_t, h);
    float r;
    if (SP_ABL & 1) return y;
    asm("v_fma_mix_f32 %0, 1.0, %1, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(y), "v"(hb));
    return r;
}

// One pipeline stage of the fused layers 1 + 2.  Stage B issues, as ONE hand-interleaved instruction stream,
//   * the 24 layer-2 MFMAs of K block B            (inputs: bh / bl, the split activations of block B),
//   * the layer-1 MFMAs of block B + 2             (into accW),
//   * the VALU epilogue of block B + 1             (accR -> bhN / blN: ReLU, Hadamard, split),
// which are mutually independent.  The epilogue is cut
// into sub-steps of ~3 VALU instructions and one sub-step follows every second MFMA, pinned with sched_barrier(0): a wave's
// own VALU / LDS work has to sit BETWEEN its MFMAs -- the matrix pipe hides ~2.5 other instructions per 16-cycle MFMA when
// they are interleaved at that grain and almost none of a VALU burst that follows an MFMA burst
// (profiles/r02_mfma_valu_overlap_probe.txt; sched_group_barrier did not move hipcc's clustered schedule for this kernel).
// Stages -2 and -1 fill the pipeline (no layer-2 work yet).
template <int B>
__device__ __forceinline__ void stage(const u32x4 *__restrict__ lds4, const f32x4 *__restrict__ ldsv, const LdsBase &lb,
                                      const f16x8 (&cbh)[2][NT], const f16x8 (&cbl)[2][NT], const f16x8 (&bh)[NT], const f16x8 (&bl)[NT],
                                      f32x4 (&acc2)[4][NT], f32x4 (&accW)[2][NT], const f32x4 (&accR)[2][NT], f16x8 (&bhN)[NT], f16x8 (&blN)[NT],
                                      CosJob &cj) {
    constexpr int NTI_W = (B + 2 < KB2) ? ntiles_of(B + 2) : 0;                  // layer-1 tiles written (block B + 2)
    constexpr int NTI_R = (B + 1 >= 0 && B + 1 < KB2) ? ntiles_of(B + 1) : 0;    // layer-1 tiles read by the epilogue (block B + 1)
    constexpr bool HAS_L2 = B >= 0;
    constexpr int N_L2 = HAS_L2 ? 3 * 4 * NT : 0, PER_KB = 3 * NTI_W * NT, NM = N_L2 + 2 * PER_KB;
    // The epilogue of a register pair is 8 plain VALU instructions, issued as FOUR pieces of two (round 3; before: pieces of 4 / 3 / 1 after
    // every second MFMA).  Beside the matrix pipe the cost of K interleaved instructions is convex in K (+0.3 / +0.9 / +2.0 / +3.9 ns for
    // K = 1 .. 4, profiles/r03_valu_cost_probe.txt), so the same instructions are cheaper spread two by two over more slots.
    constexpr int N_UNIT = NTI_R * NT * 2, N_SUB = 4 * N_UNIT;

    // Operands are loaded by the stage itself (layer-2 weights, features, bias up front; layer-1 weights 12 slots ahead of
    // their use).  Having the predecessor stage prefetch them (measured) changes nothing: the LDS latency is already covered
    // by the partner wave, and the extra live registers (+40) are better spent elsewhere.
    f16x8 a2h[4], a2l[4];
    if (HAS_L2) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int c = W2_U4 + ((mt * KB2 + (HAS_L2 ? B : 0)) * 2) * 64;
            a2h[mt] = __builtin_bit_cast(f16x8, ld_w(lds4, lb, c));
            a2l[mt] = __builtin_bit_cast(f16x8, ld_w(lds4, lb, c + 64));
        }
    }
    f32x4 fv[2], bias[2];
#pragma unroll
    for (int ti = 0; ti < NTI_R; ++ti) fv[ti] = ldsv[lb.fb + 4 * (2 * (B + 1) + ti)];          // S 2^-k1 features[16t + 4g + r]
#pragma unroll
    for (int ti = 0; ti < NTI_W; ++ti) bias[ti] = ldsv[lb.fl + 4 * (2 * (B + 2) + ti)];        // 2^k1 b1: the accumulators' initial value
    f16x8 a1h[2][2], a1l[2][2];
    f16x2 hP[NT][4], lP[NT][4];
    const f16x2 zero2 = {(_Float16)0.f, (_Float16)0.f};
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) { hP[nt][q] = zero2; lP[nt][q] = zero2; }
    float x0 = 0.f, x1 = 0.f, r0 = 0.f, r1 = 0.f;
    f16x2 hcur = zero2;
    __builtin_amdgcn_sched_barrier(0);

    static_for<NM>([&](auto M_) {
        constexpr int m = decltype(M_)::value;
        // ---- LDS reads of the layer-1 weights, issued ~12 MFMAs ahead of their first use
        if (NTI_W > 0) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
                if (m == (N_L2 + kb * PER_KB - 12 > 0 ? N_L2 + kb * PER_KB - 12 : 0)) {
#pragma unroll
                    for (int ti = 0; ti < NTI_W; ++ti) {
                        const int c = W1_U4 + (((2 * (B + 2) + ti) * 2 + kb) * 2) * 64;
                        a1h[kb][ti] = __builtin_bit_cast(f16x8, ld_w(lds4, lb, c));
                        a1l[kb][ti] = __builtin_bit_cast(f16x8, ld_w(lds4, lb, c + 64));
                    }
                }
        }
        // ---- the MFMA of this slot
        if constexpr (m < N_L2) {
            constexpr int p = m / (4 * NT), mt = (m % (4 * NT)) / NT, nt = m % NT;
            if (!((SP_ABL & 16) && p < 2)) acc2[mt][nt] = mf(p == 0 ? a2l[mt] : a2h[mt], p == 1 ? bl[nt] : bh[nt], acc2[mt][nt]);
        } else if constexpr (NTI_W > 0) {
            constexpr int q = m - N_L2, kb = q / PER_KB, r = q % PER_KB, p = r / (NTI_W * NT), ti = (r % (NTI_W * NT)) / NT, nt = r % NT;
            if (!((SP_ABL & 16) && p < 2))
                accW[ti][nt] = mf(p == 0 ? a1l[kb][ti] : a1h[kb][ti], p == 1 ? cbl[kb][nt] : cbh[kb][nt],
                                  (kb == 0 && p == ((SP_ABL & 16) ? 2 : 0)) ? bias[ti] : accW[ti][nt]);
        }
        // ---- epilogue sub-steps of this slot: sub-step s goes after MFMA (s + 1) NM / (N_SUB + 1)
#pragma unroll
        for (int sub = 0; sub < N_SUB; ++sub) {
            if ((sub + 1) * NM / (N_SUB + 1) == m) {
                const int u = sub / 4, ti = u / (NT * 2), nt = (u / 2) % NT, pr = u % 2;
                if (sub % 4 == 0) {                    // ReLU: 2 v_max_i32
                    x0 = (SP_ABL & 4) ? accR[ti][nt][2 * pr] : relu1(accR[ti][nt][2 * pr]);
                    x1 = (SP_ABL & 4) ? accR[ti][nt][2 * pr + 1] : relu1(accR[ti][nt][2 * pr + 1]);
                } else if (sub % 4 == 1) {             // Hadamard: 2 v_mul_f32
                    if (!(SP_ABL & 4)) { x0 *= fv[ti][2 * pr]; x1 *= fv[ti][2 * pr + 1]; }
                } else if (sub % 4 == 2) {             // hi pair, first residual: v_cvt_pk_f16_f32, v_fma_mix_f32
                    hcur = cvt_pair(x0, x1);
                    r0 = residual_lo32(x0, hcur);
                    hP[nt][2 * ti + pr] = hcur;
                } else {                               // second residual, lo pair: v_fma_mix_f32, v_cvt_pk_f16_f32
                    r1 = residual_hi32(x1, hcur);
                    lP[nt][2 * ti + pr] = cvt_pair(r0, r1);
                }
            }
        }
        if constexpr (B == 5 && SP_COSJOB) cj.template piece<m>();      // pieces 0..23 of the next environment's cos embedding
        __builtin_amdgcn_sched_barrier(0);
    });
    if (N_UNIT > 0) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            bhN[nt] = cat4(hP[nt][0], hP[nt][1], hP[nt][2], hP[nt][3]);
            blN[nt] = cat4(lP[nt][0], lP[nt][1], lP[nt][2], lP[nt][3]);
        }
    }
}

// ---- launch-shared taus (mn_iqn_set_tau_mode(ctx, 1); round 4) ---------------------------------------------------------------------
// When every environment of a launch sees the SAME 32 quantile fractions, layer 1 -- relu(W1 cos(pi k tau) + b1), model.py:141-157,
// 176-178 -- does not depend on the environment: it is a [32 tau x 208 feature] constant of the launch, computed once by the preparation
// launch (iqn_shared_prep_kernel, exact float32 FMA chains on accurate cosines) and parked in the LDS region the layer-1 weights would
// occupy, as [13 tiles][2 tau tiles][64 lanes] float4 = the C-tile registers the per-env kernel's layer-1 accumulators hold.  Per
// environment what remains is the Hadamard product with the (scaled) encoder features and the operand split: six plain VALU instructions
// per register pair in three pieces of two, no ReLU, no v_cos, no layer-1 matrix instructions -- 216 MFMAs per environment instead of 372.
constexpr int H1_FLOATS = T1 * NT * 64 * 4;      // 6 656 = 32 taus x 208 features
static_assert(H1_FLOATS / 4 <= W2_U4, "the layer-1 constant fits into the W1 region of the image");

// Stage B of the shared-tau pipeline: the 24 layer-2 MFMAs of K block B (B >= 0) || Hadamard + split of block B + 1 (-> bhN / blN).
template <int B>
__device__ __forceinline__ void stage_sh(const u32x4 *__restrict__ lds4, const f32x4 *__restrict__ ldsv, const LdsBase &lb,
                                         const f16x8 (&bh)[NT], const f16x8 (&bl)[NT], f32x4 (&acc2)[4][NT], f16x8 (&bhN)[NT], f16x8 (&blN)[NT]) {
    constexpr int NTI_R = ntiles_of(B + 1);
    constexpr bool HAS_L2 = B >= 0;
    constexpr int N_L2 = HAS_L2 ? 3 * 4 * NT : 0;
    constexpr int N_UNIT = NTI_R * NT * 2, N_SUB = 3 * N_UNIT;
    constexpr int NM = HAS_L2 ? N_L2 : N_SUB;          // stage -1 has no matrix instructions: one slot per piece
    f16x8 a2h[4], a2l[4];
    if (HAS_L2) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int c = W2_U4 + ((mt * KB2 + (HAS_L2 ? B : 0)) * 2) * 64;
            a2h[mt] = __builtin_bit_cast(f16x8, ld_w(lds4, lb, c));
            a2l[mt] = __builtin_bit_cast(f16x8, ld_w(lds4, lb, c + 64));
        }
    }
    f32x4 fv[2], h1v[2][NT];
#pragma unroll
    for (int ti = 0; ti < NTI_R; ++ti) {
        fv[ti] = ldsv[lb.fb + 4 * (2 * (B + 1) + ti)];                                           // S1 features[16t + 4g + r]
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) h1v[ti][nt] = ldsv[lb.w_lo + ((2 * (B + 1) + ti) * NT + nt) * 64];     // relu(layer 1)[16t + 4g + r][tau 16nt + col]
    }
    f16x2 hP[NT][4], lP[NT][4];
    const f16x2 zero2 = {(_Float16)0.f, (_Float16)0.f};
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) { hP[nt][q] = zero2; lP[nt][q] = zero2; }
    float x0 = 0.f, x1 = 0.f, r0 = 0.f, r1 = 0.f;
    f16x2 hcur = zero2;
    __builtin_amdgcn_sched_barrier(0);
    static_for<NM>([&](auto M_) {
        constexpr int m = decltype(M_)::value;
        if constexpr (m < N_L2) {
            constexpr int p = m / (4 * NT), mt = (m % (4 * NT)) / NT, nt = m % NT;
            acc2[mt][nt] = mf(p == 0 ? a2l[mt] : a2h[mt], p == 1 ? bl[nt] : bh[nt], acc2[mt][nt]);
        }
#pragma unroll
        for (int sub = 0; sub < N_SUB; ++sub) {
            if ((HAS_L2 ? (sub + 1) * NM / (N_SUB + 1) : sub) == m) {
                const int u = sub / 3, ti = u / (NT * 2), nt = (u / 2) % NT, pr = u % 2;
                if (sub % 3 == 0) {                    // Hadamard: 2 v_mul_f32
                    x0 = h1v[ti][nt][2 * pr] * fv[ti][2 * pr];
                    x1 = h1v[ti][nt][2 * pr + 1] * fv[ti][2 * pr + 1];
                } else if (sub % 3 == 1) {             // hi pair, first residual
                    hcur = cvt_pair(x0, x1);
                    r0 = residual_lo32(x0, hcur);
                    hP[nt][2 * ti + pr] = hcur;
                } else {                               // second residual, lo pair
                    r1 = residual_hi32(x1, hcur);
                    lP[nt][2 * ti + pr] = cvt_pair(r0, r1);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    });
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        bhN[nt] = cat4(hP[nt][0], hP[nt][1], hP[nt][2], hP[nt][3]);
        blN[nt] = cat4(lP[nt][0], lP[nt][1], lP[nt][2], lP[nt][3]);
    }
}

// The launch's layer-1 constant + its random numbers (+ the weight image when stale), one launch in front of the shared-tau act kernel:
//   blocks [0, pack_blocks)                       weight image (as iqn_split_prep_kernel)
//   blocks [pack_blocks, pack_blocks + H1_BLOCKS) h1[((t NT + nt) 64 + lane) 4 + r] = relu(b1[j] + sum_k W1[j][k] cos(tau pi k)), j = 16 t + 4 (lane >> 4) + r,
//                                                 tau = taus[16 nt + (lane & 15)]: the float32 product tau * (float)(pi k) and an accurate cosine, as
//                                                 model.py:149-155 forms them; k ascending float32 FMA chain
//   the rest                                      draws[0 .. 32) = the launch's taus = U[0,1) cvar, draws[32 .. 32 + n) = exploration uniforms
// taus_in != nullptr: injected taus (mn_iqn_act), no draws.  Every H1 block re-derives the 32 taus itself (counter-based: same values).
constexpr int H1_BLOCKS = H1_FLOATS / 256;      // 26
__global__ __launch_bounds__(256) void iqn_shared_prep_kernel(IqnWeights w, const float *__restrict__ consts, uint32_t *__restrict__ packed,
                                                              const uint64_t *__restrict__ rng_state, float *__restrict__ draws, int n,
                                                              const float *__restrict__ taus_in, float cvar, int pack_blocks, float *__restrict__ h1) {
    __shared__ float cs[K_TAUS][N_COS + 1];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < pack_blocks) {
        const int i = blockIdx.x * blockDim.x + tid;
        if (i < OFF_FB) packed[i] = pack_word(w, consts, i);
        return;
    }
    uint32_t k0 = 0, k1 = 0;
    if (!taus_in) {
        const uint64_t base = mix64(rng_state[0] + 0x9E3779B97F4A7C15ull * (rng_state[1] + 1));
        k0 = (uint32_t)base; k1 = (uint32_t)(base >> 32);
    }
    const int hb = (int)blockIdx.x - pack_blocks;
    if (hb < H1_BLOCKS) {
        for (int i = tid; i < K_TAUS * N_COS; i += 256) {
            const int ti = i / N_COS, k = i % N_COS;
            const float tau = taus_in ? taus_in[ti] : u01((uint32_t)ti, k0, k1) * cvar;
            cs[ti][k] = cosf(tau * (float)(3.14159265358979323846 * k));
        }
        __syncthreads();
        const int o = hb * 256 + tid, r = o & 3, lane = (o >> 2) & 63, nt = (o >> 8) % NT, t = o / (256 * NT);
        const int j = 16 * t + 4 * (lane >> 4) + r, ti = 16 * nt + (lane & 15);
        float a = w.b1[j];
        for (int k = 0; k < N_COS; ++k) a = fmaf(w.W1[j * N_COS + k], cs[ti][k], a);
        a = fmaxf(a, 0.f);
        h1[o] = a;
        // this block's maximum behind the constant (the tiled kernel's preparation scales T = W2 h1 by it)
        __shared__ float bmax[4];
        float m = a;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        if ((tid & 63) == 0) bmax[tid >> 6] = m;
        __syncthreads();
        if (tid == 0) h1[H1_FLOATS + hb] = fmaxf(fmaxf(bmax[0], bmax[1]), fmaxf(bmax[2], bmax[3]));
        return;
    }
    if (taus_in) return;
    const long total = (long)n + K_TAUS;
    const long stride = (long)((int)gridDim.x - pack_blocks - H1_BLOCKS) * 256;
    for (long idx = (long)(hb - H1_BLOCKS) * 256 + tid; idx < total; idx += stride) {
        float u = u01((uint32_t)idx, k0, k1);
        if (idx < K_TAUS) u *= cvar;
        draws[idx] = u;
    }
}

// ---- observation encoders (model.py:170-173) in schedulable pieces.  Lane l computes sensor features l, l + 64, l + 128 (22
// inputs each) and velocity / goal feature l (2 inputs; lanes < 32), then its share of the activation bound.  25 sub-steps of
// 1 LDS read + 2-4 VALU.
struct EncState {
    float fval[4];
    float bnd;
    f32x2 a2;
};
template <int IDX>
__device__ __forceinline__ void enc_substep(const float *__restrict__ lds, const f32x4 *__restrict__ ldsv, int enc_w, int enc_f, int lane,
                                            const float (&ov)[28], EncState &st, int off_wvg = OFF_WVG) {
    if constexpr (IDX < 24) {
        constexpr int j = IDX / 8, k = IDX % 8;
        if constexpr (k == 0) {
            if (j == 0) st.bnd = 0.f;
            st.a2 = (f32x2){lds[enc_f + (OFF_BE - OFF_BND) + 32 + 64 * j], 0.f};
        } else if constexpr (k < 7) {                  // two v_pk_fma_f32 per 4 inputs
            constexpr int i4 = k - 1;
            const f32x4 wv = ldsv[enc_w + i4 * 176 + 64 * j];
            st.a2 += (f32x2){wv[0], wv[1]} * (f32x2){ov[4 + 4 * i4], ov[5 + 4 * i4]};
            st.a2 += (f32x2){wv[2], wv[3]} * (f32x2){ov[6 + 4 * i4], ov[7 + 4 * i4]};
        } else {
            const bool valid = lane + 64 * j < 176;    // lanes past the 176 sensor features computed on in-range garbage
            const float a = valid ? st.a2[0] + st.a2[1] : 0.f;
            st.fval[j] = a;
            st.bnd = fmaxf(st.bnd, fabsf(a) * (valid ? lds[enc_f + 32 + 64 * j] : 0.f));
        }
    } else {
        const f32x2 wv = reinterpret_cast<const f32x2 *>(lds + off_wvg)[lane & 31];
        const float i0 = lane < 16 ? ov[0] : ov[2], i1 = lane < 16 ? ov[1] : ov[3];
        const float a = lane < 32 ? lds[enc_f + (OFF_BE - OFF_BND)] + wv[0] * i0 + wv[1] * i1 : 0.f;
        st.fval[3] = a;
        st.bnd = fmaxf(st.bnd, fabsf(a) * lds[enc_f]);
    }
}
constexpr int N_ENC_SUB = 25;

// the per-environment scales from the lanes' bounds (see the header): one power of two per hidden layer, S_l |h_l| < 2^15
struct EnvScale {
    float S1, S2, S3;      // layer-1 / -2 / -3 activations are carried as S_l h_l
    float r21, r32;        // S2 / S1, S3 / S2 (exact powers of two): folded into the accumulator unscale of layers 2, 3
    float invS3;
};
__device__ __forceinline__ int bound_exponent(float M) {       // e with M in [2^(e-127), 2^(e-126)), M clamped to a sane range
    M = fminf(fmaxf(M, 1e-30f), 1e30f);
    return (int)(__builtin_bit_cast(uint32_t, M) >> 23);
}
__device__ __forceinline__ EnvScale env_scale(float bnd, float a2, float d2, float a3, float d3) {
    const float m1 = wave_max_nonneg(bnd);
    const int e1 = bound_exponent(m1), e2 = bound_exponent(fmaf(a2, m1, d2)), e3 = bound_exponent(fmaf(a3, m1, d3));
    EnvScale sc;
    sc.S1 = __builtin_bit_cast(float, (uint32_t)(268 - e1) << 23);           // 2^(141 - e): S M < 2^15
    sc.S2 = __builtin_bit_cast(float, (uint32_t)(268 - e2) << 23);
    sc.S3 = __builtin_bit_cast(float, (uint32_t)(268 - e3) << 23);
    sc.r21 = __builtin_bit_cast(float, (uint32_t)(127 + e1 - e2) << 23);
    sc.r32 = __builtin_bit_cast(float, (uint32_t)(127 + e2 - e3) << 23);
    sc.invS3 = __builtin_bit_cast(float, (uint32_t)(e3 - 14) << 23);         // 2^(e - 141)
    return sc;
}
// S 2^-k1 feature -> this wave's LDS buffer (the Hadamard multiplier of the layer-1 epilogue)
__device__ __forceinline__ void store_features(float *__restrict__ lds, int fb_f, int lane, const EncState &st, float Sc) {
#pragma unroll
    for (int j = 0; j < 3; ++j)
        if (lane + 64 * j < 176) lds[fb_f + 32 + 64 * j] = st.fval[j] * Sc;
    if (lane < 32) lds[fb_f] = st.fval[3] * Sc;
}

// The end of an environment's pipeline as one hand-interleaved stream of 72 MFMAs:
//   slots  0..11  layer-2 MFMAs of the last K block for output tiles 0, 1
//   slots 12..23  the same for tiles 2, 3            || layer-2 epilogue of tiles 0, 1 (unscale + bias, ReLU, split)
//   slots 24..47  layer-3 MFMAs of K block 0         || layer-2 epilogue of tiles 2, 3
//   slots 48..71  layer-3 MFMAs of K block 1 (output tiles 0, 1 first)
// S2 h2 = relu(acc2 c2 + S b2) with c2 = 2^-k2 S2 / S1 and S = S2 passed by the caller; the layer-3 accumulators are left in acc3.
template <bool COS = true>
__device__ __forceinline__ void tail(const u32x4 *__restrict__ lds4, const f32x4 *__restrict__ ldsv, const LdsBase &lb, float c2, float S,
                                     const f16x8 (&bh)[NT], const f16x8 (&bl)[NT], f32x4 (&acc2)[4][NT], f32x4 (&acc3)[4][NT], CosJob &cj) {
    f16x8 a2h[4], a2l[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int c = W2_U4 + ((mt * KB2 + (KB2 - 1)) * 2) * 64;
        a2h[mt] = __builtin_bit_cast(f16x8, ld_w(lds4, lb, c));
        a2l[mt] = __builtin_bit_cast(f16x8, ld_w(lds4, lb, c + 64));
    }
    f32x4 sb[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) sb[t] = ldsv[lb.fl + ((OFF_B2 - OFF_B1) >> 2) + 4 * t] * S;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc3[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f16x8 a3h[2][4], a3l[2][4];
    f16x2 hP[2][NT][4], lP[2][NT][4];
    f16x8 b3h[2][NT], b3l[2][NT];
    float t0 = 0.f, t1 = 0.f, r0 = 0.f, r1 = 0.f;
    f16x2 hcur = {(_Float16)0.f, (_Float16)0.f};
    __builtin_amdgcn_sched_barrier(0);
    static_for<72>([&](auto M_) {
        constexpr int m = decltype(M_)::value;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
            if (m == 12 + 24 * kb) {            // layer-3 weights, 12 slots ahead
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const int c = W3_U4 + ((mt * 2 + kb) * 2) * 64;
                    a3h[kb][mt] = __builtin_bit_cast(f16x8, ld_w(lds4, lb, c));
                    a3l[kb][mt] = __builtin_bit_cast(f16x8, ld_w(lds4, lb, c + 64));
                }
            }
        if constexpr (m < 24) {                 // layer 2, last K block: tiles {0, 1} then {2, 3}
            constexpr int half = m / 12, q = m % 12, p = q / 4, mt = 2 * half + (q % 4) / 2, nt = q % 2;
            if (!((SP_ABL & 16) && p < 2)) acc2[mt][nt] = mf(p == 0 ? a2l[mt] : a2h[mt], p == 1 ? bl[nt] : bh[nt], acc2[mt][nt]);
        } else if constexpr (m < 48) {          // layer 3, K block 0
            constexpr int q = m - 24, p = q / 8, mt = (q % 8) / 2, nt = q % 2;
            if (!((SP_ABL & 16) && p < 2)) acc3[mt][nt] = mf(p == 0 ? a3l[0][mt] : a3h[0][mt], p == 1 ? b3l[0][nt] : b3h[0][nt], acc3[mt][nt]);
        } else {                                // layer 3, K block 1: tiles {0, 1} then {2, 3}
            constexpr int q = m - 48, half = q / 12, r = q % 12, p = r / 4, mt = 2 * half + (r % 4) / 2, nt = r % 2;
            if (!((SP_ABL & 16) && p < 2)) acc3[mt][nt] = mf(p == 0 ? a3l[1][mt] : a3h[1][mt], p == 1 ? b3l[1][nt] : b3h[1][nt], acc3[mt][nt]);
        }
        // epilogue sub-steps (48 = 2 halves x 8 register pairs x 3): half 0 two per slot in slots 12..23, half 1 one per slot in 24..47
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) {
            const int sub = (m >= 12 && m < 24) ? 2 * (m - 12) + rep : ((m >= 24 && m < 48 && rep == 0) ? m : -1);
            if (sub >= 0) {
                const int half = sub / 24, u = (sub % 24) / 3, ti = u / (NT * 2), nt = (u / 2) % NT, pr = u % 2, t = 2 * half + ti;
                if (sub % 3 == 0) {
                    t0 = relu1(fmaf(acc2[t][nt][2 * pr], c2, sb[t][2 * pr]));
                    t1 = relu1(fmaf(acc2[t][nt][2 * pr + 1], c2, sb[t][2 * pr + 1]));
                } else if (sub % 3 == 1) {
                    hcur = cvt_pair(t0, t1);
                    residual_pair(t0, t1, hcur, r0, r1);
                    hP[half][nt][2 * ti + pr] = hcur;
                } else {
                    lP[half][nt][2 * ti + pr] = cvt_pair(r0, r1);
                }
                if (sub == 23 || sub == 47) {
                    const int hdone = sub / 24;
#pragma unroll
                    for (int n2 = 0; n2 < NT; ++n2) {
                        b3h[hdone][n2] = cat4(hP[hdone][n2][0], hP[hdone][n2][1], hP[hdone][n2][2], hP[hdone][n2][3]);
                        b3l[hdone][n2] = cat4(lP[hdone][n2][0], lP[hdone][n2][1], lP[hdone][n2][2], lP[hdone][n2][3]);
                    }
                }
            }
        }
        if constexpr (SP_COSJOB && COS) {      // pieces 24..79 of the next environment's cos embedding: two per slot where the tail has no epilogue work of its own
            if constexpr (m < 12) { cj.template piece<24 + 2 * m>(); cj.template piece<25 + 2 * m>(); }
            else if constexpr (m >= 48 && m < 56) { cj.template piece<48 + 2 * (m - 48)>(); cj.template piece<49 + 2 * (m - 48)>(); }
            else if constexpr (m >= 56) cj.template piece<64 + (m - 56)>();
        }
        __builtin_amdgcn_sched_barrier(0);
    });
}

// QUANT = false: acting / training (tau mean before the linear output layer, f32 VALU mat-vec).
// QUANT = true : IQNAgent.act_eval (agent.py:217-236): the output layer runs per tau on the matrix pipe (12 more MFMAs on a padded
//                16-row tile), the [n][32][9] quantile values are written out and Q is their mean.
// SHARED = true : launch-shared taus (see stage_sh): `taus` is unused, `h1` is the launch's layer-1 constant (iqn_shared_prep_kernel).
// NW = wavefronts per workgroup (one workgroup per CU): 8 = two per SIMD; the shared-tau form needs ~155 registers and also runs three per SIMD.
// LATE = true : rows whose observation is still being written when the launch starts (the episode resets of the vector step, running on
//                another stream under this kernel: mn_reset_done_async) are taken LAST by their wavefront, each after its "row is final"
//                word (`late_flag[e] == tick`, written by the reset wave behind its write-through row) has arrived; `late_mask` = the
//                step's done flags.  Same arithmetic, same results; a wavefront handles at most 64 rows in this form (launch_act checks).
struct LateRows {
    const uint8_t *mask;       // [n] != 0: the row is rewritten by the reset running beside this launch
    const uint32_t *flag;      // [n] == tick once it has been
    uint32_t tick;
    uint32_t *status;          // += 1 per wait that ran out (then the row is taken as it is: the caller falls back to resets in front / raises)
    uint32_t *status_host;     // host-mapped copy of that count, which the host reads without synchronising (mn_iqn_late_timeouts_peek)
    uint64_t bound_ticks;      // the bound of a wait in ticks of the 100 MHz counter (mn_iqn_set_late_bound_ms; default 0.5 s)
};
constexpr uint64_t LATE_BOUND_TICKS = 50000000ull;      // 0.5 s of the 100 MHz counter

template <bool QUANT, bool SHARED = false, int NW = WAVES, bool LATE = false>
__global__ __launch_bounds__(64 * NW) void iqn_qvals_split_kernel(const float *__restrict__ obs, const float *__restrict__ taus,
                                                                 const uint32_t *__restrict__ packed, float *__restrict__ qvals,
                                                                 const float *__restrict__ explore_u, float eps,
                                                                 int32_t *__restrict__ actions, int n, uint64_t *__restrict__ rng_state,
                                                                 float *__restrict__ quantiles, const float *__restrict__ h1 = nullptr,
                                                                 const LateRows late = {}) {
    static_assert(!LATE || (!QUANT && !SHARED), "late rows: the acting form with per-environment taus");
#ifndef SP_LATE_PRIO
#define SP_LATE_PRIO 0
#endif
    if constexpr (LATE) __builtin_amdgcn_s_setprio(SP_LATE_PRIO);      // the reset wavefronts that share these SIMDs take the issue slots this kernel leaves
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    if (rng_state && blockIdx.x == 0 && tid == 0) rng_state[1] += 1;   // the draws of this call were made by the prep kernel
    // the acting form has no use for the output layer's MFMA operands (act_eval's quantiles): its feature buffers sit there, and the 4 KB
    // that frees are what lets a reset workgroup (3.1 KB of LDS) share the CU with this one
    constexpr int IMG = QUANT ? OFF_FB : OFF_W4H;
    {
        const u32x4 *src = reinterpret_cast<const u32x4 *>(packed);
        u32x4 *dst = reinterpret_cast<u32x4 *>(lds);
        if constexpr (SHARED) {      // the layer-1 constant takes the place of the layer-1 weights
            const u32x4 *hsrc = reinterpret_cast<const u32x4 *>(h1);
            for (int i = tid; i < H1_FLOATS / 4; i += blockDim.x) dst[i] = hsrc[i];
            for (int i = W2_U4 + tid; i < IMG / 4; i += blockDim.x) dst[i] = src[i];
        } else {
            for (int i = tid; i < IMG / 4; i += blockDim.x) dst[i] = src[i];
        }
    }
    __syncthreads();

    const int lane = tid & 63, g = lane >> 4, col = lane & 15;
    const int wave = tid >> 6, waves_per_block = blockDim.x >> 6;
    const f32x4 *ldsv = reinterpret_cast<const f32x4 *>(lds);
    const u32x4 *lds4 = reinterpret_cast<const u32x4 *>(lds);
    LdsBase lb;
    // per-wave feature buffer: behind the image; shared-tau form: in the rest of the W1 region, behind the layer-1 constant
    constexpr int FB0 = SHARED ? H1_FLOATS : IMG;
    static_assert(!SHARED || H1_FLOATS + NW * F <= W2_U4 * 4, "feature buffers of the shared-tau kernel fit into the W1 region");
    static_assert(SHARED || NW <= WAVES, "feature buffers behind the image: WAVES of them");
    lb.w_lo = lane; lb.w_hi = lane + 4096; lb.fl = (OFF_B1 >> 2) + g; lb.fb = ((FB0 + wave * F) >> 2) + g;
    int enc_w = (OFF_WS >> 2) + lane;       // sensor encoder weights (16-byte units)
    int enc_f = OFF_BND + lane;             // bounds / encoder biases (floats)
    int fb_f = FB0 + wave * F + lane;       // this wave's feature buffer (floats)
    asm volatile("" : "+v"(lb.w_lo), "+v"(lb.w_hi), "+v"(lb.fl), "+v"(lb.fb), "+v"(enc_w), "+v"(enc_f), "+v"(fb_f));
    const float c1 = lds[OFF_CST + 0], c2 = lds[OFF_CST + 1], c3 = lds[OFF_CST + 2];
    const float a2 = lds[OFF_CST + 3], d2 = lds[OFF_CST + 4], a3 = lds[OFF_CST + 5], d3 = lds[OFF_CST + 6];

    // cos(tau * pi * k), k = 32 kb + 8 g + i: v_cos_f32 takes its argument in revolutions (tau * k / 2 <= 32) and reduces it itself
    const float hk0 = 4.0f * (float)g;     // k / 2 = hk0 + (16 kb + i / 2)

    // Software-pipelining the loop ACROSS environments (next environment's taus / observation row loaded and its encoders run in
    // the pipeline's issue gaps) was built and measured: 359 us against 326 us -- the kernel is
    // bound by the SIMD's aggregate instruction issue (~1 instruction per 5 cycles over both waves, the same rate as
    // profiles/r02_mfma_valu_overlap_probe.txt at K = 3), so moving instructions around buys nothing and the extra live
    // registers cost spills.  Requesting ONLY the next environment's taus and observation row one iteration ahead (2 VGPRs,
    // 28 SGPRs) changes nothing either (338 vs 337 us, alternating runs on one GPU): that latency is covered by the partner wave.
    [[maybe_unused]] int sp_iter = 0;
    // layer-1 B operands: the cos embedding (model.py:155), unscaled, split.  Computed here for a wave's FIRST environment only; for every
    // later one by the CosJob pieces inside stage 5 / the tail of the environment before it (same expressions, same bits).
    f16x8 cbh[2][NT], cbl[2][NT];
    CosJob cj;
    cj.hk0 = hk0;
    const int e_first = blockIdx.x * waves_per_block + wave, e_stride = gridDim.x * waves_per_block;
    // LATE: the wave's rows as two bit sets (bit k = row e_first + k e_stride): final observations first, late ones last
    [[maybe_unused]] unsigned long long rows_now = 0, rows_late = 0;
    [[maybe_unused]] bool is_late = false;
    int e0 = e_first;
    if constexpr (LATE) {
        const int e_l = e_first + lane * e_stride;
        const bool v = e_l < n, l = v && late.mask[v ? e_l : 0] != 0;
        const unsigned long long mv = __ballot(v), ml = __ballot(l);
        rows_now = mv & ~ml; rows_late = ml;
        if (rows_now) { const int k = __builtin_ctzll(rows_now); rows_now &= rows_now - 1; e0 = e_first + k * e_stride; }
        else if (rows_late) { const int k = __builtin_ctzll(rows_late); rows_late &= rows_late - 1; e0 = e_first + k * e_stride; is_late = true; }
        else e0 = n;
        e0 = __builtin_amdgcn_readfirstlane(e0);
    }
    if (!SHARED && e0 < n) {
        float tau[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) tau[nt] = taus[(size_t)e0 * K_TAUS + 16 * nt + col];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f16x2 h[4], l[4];
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    if (SP_ABL & 8) { h[p] = __builtin_bit_cast(f16x2, tau[nt]); l[p] = h[p]; }
                    else
                    split2(__builtin_amdgcn_cosf(tau[nt] * (hk0 + (16.0f * kb + 0.5f * (2 * p)))),
                           __builtin_amdgcn_cosf(tau[nt] * (hk0 + (16.0f * kb + 0.5f * (2 * p + 1)))), h[p], l[p]);
                cbh[kb][nt] = cat4(h[0], h[1], h[2], h[3]);
                cbl[kb][nt] = cat4(l[0], l[1], l[2], l[3]);
            }
    }
    int e_follow = n;      // LATE: the row after `e` (n: none)
    [[maybe_unused]] bool follow_late = false;
    for (int e = e0; e < n; e = LATE ? e_follow : e + e_stride) {
        [[maybe_unused]] unsigned long long tk[16];
#define SP_TICK(i) do { if (SP_ABL & 64) { __builtin_amdgcn_sched_barrier(0); tk[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
        SP_TICK(0);
        const float u_explore = (explore_u && eps > 0.f) ? explore_u[__builtin_amdgcn_readfirstlane(e)] : 2.0f;     // used ~10 us later
        [[maybe_unused]] const bool late_row = is_late;
        if constexpr (LATE) {
            e_follow = n; follow_late = false;
            if (rows_now) { const int k = __builtin_ctzll(rows_now); rows_now &= rows_now - 1; e_follow = e_first + k * e_stride; }
            else if (rows_late) { const int k = __builtin_ctzll(rows_late); rows_late &= rows_late - 1; e_follow = e_first + k * e_stride; follow_late = true; }
            e_follow = __builtin_amdgcn_readfirstlane(e_follow);
            is_late = follow_late;
        }
        if constexpr (!SHARED) {   // the next environment's taus (the last iteration re-reads its own: straight-line code); consumed from stage 5 on
            const int e_nx = LATE ? (e_follow < n ? e_follow : e) : (e + e_stride < n ? e + e_stride : e);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) cj.tau[nt] = taus[(size_t)e_nx * K_TAUS + 16 * nt + col];
        }
        SP_TICK(1);
        // observation encoders, per-environment scale S, S 2^-k1 features -> this wave's LDS buffer
        EnvScale sc;
        {
            const float *orow = obs + (size_t)__builtin_amdgcn_readfirstlane(e) * OBS;
            float ov[28];
            if (LATE && late_row) {      // (wave-uniform) wait for the reset wave's "row is final" word, then read the row past the caches
                const uint32_t *fp = late.flag + __builtin_amdgcn_readfirstlane(e);
                const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
                while (__hip_atomic_load(fp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != late.tick) {
                    __builtin_amdgcn_s_sleep(16);
                    if (__builtin_amdgcn_s_memrealtime() - t0 > late.bound_ticks) {
                        if (lane == 0) __hip_atomic_store(late.status_host, atomicAdd(late.status, 1u) + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        break;
                    }
                }
                const uint32_t x = __hip_atomic_load(reinterpret_cast<const uint32_t *>(orow) + (lane < OBS ? lane : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int i = 0; i < 28; ++i) ov[i] = i < OBS ? __builtin_bit_cast(float, __builtin_amdgcn_readlane((int)x, i)) : 0.f;
            } else {
#pragma unroll
            for (int i = 0; i < 28; ++i) ov[i] = i < OBS ? orow[i] : 0.f;
            }
            EncState st;
            static_for<N_ENC_SUB>([&](auto I_) { enc_substep<decltype(I_)::value>(lds, ldsv, enc_w, enc_f, lane, ov, st); });
            sc = env_scale(st.bnd, a2, d2, a3, d3);
            store_features(lds, fb_f, lane, st, SHARED ? sc.S1 : sc.S1 * c1);      // (the shared layer-1 constant carries no 2^k1)
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }

        SP_TICK(2);
        // ---- layers 1 + 2 fused over the 7 K blocks of layer 2, software-pipelined as in the exact kernel: the layer-1
        // MFMAs of block b + 1 are issued before the VALU epilogue of block b
        f32x4 acc2[4][NT];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc2[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // Stage b = layer-2 MFMAs of block b + layer-1 MFMAs of block b + 2 + VALU epilogue of block b + 1, see stage()
        f16x8 bhA[NT], blA[NT], bhB[NT], blB[NT];
        f32x4 acc3[4][NT];
        if constexpr (SHARED) {
            stage_sh<-1>(lds4, ldsv, lb, bhB, blB, acc2, bhA, blA);      // Hadamard + split of block 0
            SP_TICK(3); SP_TICK(4);
            stage_sh<0>(lds4, ldsv, lb, bhA, blA, acc2, bhB, blB);
            SP_TICK(5);
            stage_sh<1>(lds4, ldsv, lb, bhB, blB, acc2, bhA, blA);
            SP_TICK(6);
            stage_sh<2>(lds4, ldsv, lb, bhA, blA, acc2, bhB, blB);
            SP_TICK(7);
            stage_sh<3>(lds4, ldsv, lb, bhB, blB, acc2, bhA, blA);
            SP_TICK(8);
            stage_sh<4>(lds4, ldsv, lb, bhA, blA, acc2, bhB, blB);
            SP_TICK(9);
            stage_sh<5>(lds4, ldsv, lb, bhB, blB, acc2, bhA, blA);
            SP_TICK(10);
            tail<false>(lds4, ldsv, lb, c2 * sc.r21, sc.S2, bhA, blA, acc2, acc3, cj);
        } else {
        f32x4 accA[2][NT], accB[2][NT];
        stage<-2>(lds4, ldsv, lb, cbh, cbl, bhB, blB, acc2, accA, accB, bhB, blB, cj);      // layer-1 block 0
        SP_TICK(3);
        stage<-1>(lds4, ldsv, lb, cbh, cbl, bhB, blB, acc2, accB, accA, bhA, blA, cj);      // layer-1 block 1, epilogue of block 0
        SP_TICK(4);
        stage<0>(lds4, ldsv, lb, cbh, cbl, bhA, blA, acc2, accA, accB, bhB, blB, cj);
        SP_TICK(5);
        stage<1>(lds4, ldsv, lb, cbh, cbl, bhB, blB, acc2, accB, accA, bhA, blA, cj);
        SP_TICK(6);
        stage<2>(lds4, ldsv, lb, cbh, cbl, bhA, blA, acc2, accA, accB, bhB, blB, cj);
        SP_TICK(7);
        stage<3>(lds4, ldsv, lb, cbh, cbl, bhB, blB, acc2, accB, accA, bhA, blA, cj);
        SP_TICK(8);
        stage<4>(lds4, ldsv, lb, cbh, cbl, bhA, blA, acc2, accA, accB, bhB, blB, cj);
        SP_TICK(9);
        stage<5>(lds4, ldsv, lb, cbh, cbl, bhB, blB, acc2, accB, accA, bhA, blA, cj);
        SP_TICK(10);
        tail(lds4, ldsv, lb, c2 * sc.r21, sc.S2, bhA, blA, acc2, acc3, cj);
        if (SP_COSJOB) cj.finish(cbh, cbl);      // (register renaming: the operands of the next environment)
        }
        SP_TICK(11);
        const float c3e = c3 * sc.r32;      // layer-3 accumulators carry S2 2^k3: to S3
        float qv;
        if constexpr (!QUANT) {
            // ---- layer 3 epilogue, tau mean, f32 output layer (as in the exact kernel; the sums carry the factor S3) ---------
            float hs[16];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const f32x4 sb = ldsv[lb.fl + ((OFF_B3 - OFF_B1) >> 2) + 4 * mt] * sc.S3;
                const f32x4 h0 = relu4s(fma4(acc3[mt][0], c3e, sb)), h1 = relu4s(fma4(acc3[mt][1], c3e, sb));
#pragma unroll
                for (int r = 0; r < 4; ++r) hs[4 * mt + r] = h0[r] + h1[r];
            }
            row_sum16_x16(hs);                          // sum over the 32 taus of h3[16 mt + 4 g + r], in every lane of row group g
            float part = 0.f;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const f32x4 a = ldsv[lb.w_hi + ((OFF_W4 >> 2) - 4096) + mt * 64];
#pragma unroll
                for (int r = 0; r < 4; ++r) part = fmaf(a[r], hs[4 * mt + r], part);
            }
            part = sum_rows4(part);                     // the four row groups' shares of action `col`
            qv = part * (sc.invS3 * (1.0f / K_TAUS)) + lds[OFF_B4 + col];     // Q(s, action = col), valid for col < 9
        } else {
            // ---- quantile values Z(tau, a) = W4 h3(tau) + b4 (model.py:185): layer 3 epilogue + split, then the output layer as 12 MFMAs
            // on a padded 16-row tile; lane (g, col) ends up with actions 4 g + r of tau 16 nt + col (scaled by S 2^k4)
            f16x8 b4h[2][NT], b4l[2][NT];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const f32x4 sb0 = ldsv[lb.fl + ((OFF_B3 - OFF_B1) >> 2) + 4 * (2 * kb)] * sc.S3, sb1 = ldsv[lb.fl + ((OFF_B3 - OFF_B1) >> 2) + 4 * (2 * kb + 1)] * sc.S3;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    split_tiles(relu4s(fma4(acc3[2 * kb][nt], c3e, sb0)), relu4s(fma4(acc3[2 * kb + 1][nt], c3e, sb1)), b4h[kb][nt], b4l[kb][nt]);
            }
            f32x4 acc4[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc4[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const u32x4 *w4 = reinterpret_cast<const u32x4 *>(lds + OFF_W4H);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const f16x8 ah = __builtin_bit_cast(f16x8, w4[(kb * 2) * 64 + lane]), al = __builtin_bit_cast(f16x8, w4[(kb * 2 + 1) * 64 + lane]);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc4[nt] = mf(al, b4h[kb][nt], acc4[nt]);
                    acc4[nt] = mf(ah, b4l[kb][nt], acc4[nt]);
                    acc4[nt] = mf(ah, b4h[kb][nt], acc4[nt]);
                }
            }
            const float unscale = sc.invS3 * lds[OFF_CST + 8];    // 1 / (S3 2^k4)
            const f32x4 b4 = ldsv[(OFF_B4 >> 2) + g];
            float mine = 0.f;      // lane `a` (< 9) ends up with Q(s, a) = mean over the 32 taus
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int a_idx = 4 * g + r;
                float sum = 0.f;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float z = fmaf(acc4[nt][r], unscale, b4[r]);
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
            // lane a holds action a: nine v_readlane (no LDS round trip); first maximum wins, like np.argmax
            float best = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, qv), 0));
            int arg = 0;
#define SP_ARG(a) { const float v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, qv), a)); if (v > best) { best = v; arg = a; } }
            SP_ARG(1) SP_ARG(2) SP_ARG(3) SP_ARG(4) SP_ARG(5) SP_ARG(6) SP_ARG(7) SP_ARG(8)
#undef SP_ARG
            if (lane == 0) {
                int act = arg;
                if (explore_u && eps > 0.f) {
                    const float u = u_explore;               // greedy iff u > eps (agent.py:200); requested at the top of the iteration
                    if (!(u > eps)) { act = (int)(u / eps * (float)A_OUT); act = act > A_OUT - 1 ? A_OUT - 1 : act; }
                }
                actions[e] = act;
            }
        }
        SP_TICK(12);
        if ((SP_ABL & 64) && blockIdx.x == 3 && wave == 1 && lane == 0 && ++sp_iter == 6)
            printf("phase cycles (block 3, wave 1, 6th env): cos %llu  encoder+scale %llu  stage-2 %llu  stage-1 %llu  stages0..5 %llu %llu %llu %llu %llu %llu  tail %llu  output %llu  | env total %llu\n",
                   tk[1] - tk[0], tk[2] - tk[1], tk[3] - tk[2], tk[4] - tk[3], tk[5] - tk[4], tk[6] - tk[5], tk[7] - tk[6], tk[8] - tk[7], tk[9] - tk[8], tk[10] - tk[9],
                   tk[11] - tk[10], tk[12] - tk[11], tk[12] - tk[0]);
#undef SP_TICK
    }
}

}  // namespace sp
