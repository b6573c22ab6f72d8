// iqn_act_split32.h -- the split-f16 act kernel of iqn_act_split.h on v_mfma_f32_32x32x16_f16 tiles.
// Included by iqn_act.hip after iqn_act_split.h, inside its anonymous namespace (reuses namespace sp's helpers).
//
// Why a second tile shape: the split-f16 kernel is bound by instruction issue, and the same matrix work costs half as many MFMA
// instructions in 32x32x16 tiles (2 x the FLOPs per instruction at the same FLOP rate).  Probe (scripts/probes/mfma_valu_overlap.hip,
// profiles/r02_mfma_valu_overlap_probe.txt, two waves per SIMD): one 32x32x16 MFMA + 6 other instructions 16.0 ns against
// 2 x (16x16x32 + 3) = 19.0 ns.  Arithmetic, range scaling and error analysis are those of iqn_act_split.h; only the layout differs.
//
// Layout.  A operand = weights [32 output features x 16 k]: lane (h = l >> 5, row = l & 31) holds k slots 8 h + i.  B operand =
// activations [16 k x 32 taus]: all 32 taus of an environment are ONE column tile, lane (h, c = l & 31) holds k slots 8 h + i of
// tau c.  C tile [32 features x 32 taus]: lane (h, c) holds rows 8 j + 4 h + r in register 4 j + r (j, r = 0..3).  K step
// (t, s2) of the NEXT layer (s2 = 0, 1) consumes registers 8 s2 .. 8 s2 + 7 of C tile t: k slot (h, i) := feature
// 32 t + 8 (2 s2 + (i >> 2)) + 4 h + (i & 3) -- a tile's accumulators, split in place, are the B operands of two K steps.
// 208 features = 6.5 tiles: the 7th tile's upper 16 rows are padding (zero weights; +7.7 % layer-1 matrix work), and layer 2 simply
// has 13 K steps (no padding there).  The 9 x 64 output layer runs on the matrix pipe too (one padded 32-row tile, 12 MFMAs):
// with 32 taus per tile the tau-sum of 32 hidden features would cost 160 cross-lane adds, the tau-sum of 5 registers of
// quantile values costs 25.  198 MFMAs per environment (84 + 78 + 24 + 12) of 32 cycles each.
// Per-lane vectors (bias, scaled features) are stored PERMUTED, [tile][h][4 j + r], so that a lane reads its 16 values as four
// ds_read_b128.

namespace sp32 {

using sp::f16x2;
using sp::f16x8;
using sp::f32x2;
using sp::u32x4;
using sp::static_for;
using sp::cat4;
using sp::cvt_pair;
using sp::residual_pair;
using sp::relu1;
using sp::split2;
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int NTILE = 7;                                // layer-1 feature tiles of 32 (208 -> 224)
constexpr int KS2 = 13;                                 // layer-2 K steps of 16 features
constexpr int FP = NTILE * 32;                          // padded feature count of the permuted per-lane vectors
// weight image, in 16-byte units (8 halves = one lane's A operand): [.. tile ..][piece: hi, lo][64 lanes]
constexpr int W1_U4 = 0;                                // [7 t][4 s]
constexpr int W2_U4 = W1_U4 + NTILE * 4 * 2 * 64;       // [2 mt][13 ks]
constexpr int W3_U4 = W2_U4 + 2 * KS2 * 2 * 64;         // [2 mt][4 ks]
constexpr int W4_U4 = W3_U4 + 2 * 4 * 2 * 64;           // [4 ks] (rows >= 9 zero)
constexpr int END_U4 = W4_U4 + 4 * 2 * 64;
// float part (indices in floats from the start of the image); BND .. BE in the order / sizes of namespace sp (shared encoder code)
constexpr int OFF_B1 = END_U4 * 4;                      // [7 t][2 h][16] 2^k1 b1, permuted
constexpr int OFF_B2 = OFF_B1 + FP;                     // [2 mt][2 h][16] b2, permuted
constexpr int OFF_B3 = OFF_B2 + H;                      // [2 mt][2 h][16] b3, permuted
constexpr int OFF_B4 = OFF_B3 + H;                      // [16]
constexpr int OFF_BND = OFF_B4 + 16;                    // [208] B1_j
constexpr int OFF_CST = OFF_BND + F;                    // [16] c1 c2 c3 a2 d2 a3 d3 2^k4 2^-k4
constexpr int OFF_WS = OFF_CST + 16;                    // [6 i4][176 sf][4] sensor encoder
constexpr int OFF_WVG = OFF_WS + 6 * 176 * 4;           // [32 f][2] velocity / goal encoders
constexpr int OFF_BE = OFF_WVG + 64;                    // [208] encoder biases
constexpr int OFF_FB = OFF_BE + F;                      // [8 waves][7 t][2 h][16] per-wave scaled features, permuted
constexpr int LDS_FLOATS = OFF_FB + sp::WAVES * FP;
static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS image of the 32x32 split-f16 act kernel must fit the CU's 160 KB");
static_assert(OFF_WS % 4 == 0 && OFF_WVG % 4 == 0 && OFF_BE % 4 == 0 && OFF_FB % 4 == 0 && OFF_CST % 4 == 0 && OFF_B2 % 4 == 0, "16-byte aligned blocks");
static_assert(OFF_BE - OFF_BND == sp::OFF_BE - sp::OFF_BND, "encoder block laid out as in namespace sp");
constexpr int PACK_BLOCKS = (OFF_FB + 255) / 256;       // one thread per 32-bit word of the image

// position of feature f (or hidden unit f) inside its permuted 32-block: [h][4 j + r] for f = 8 j + 4 h + r
__device__ __forceinline__ int perm32(int f) { return (f & ~31) + ((f >> 2) & 1) * 16 + ((f >> 3) & 3) * 4 + (f & 3); }
// inverse: the feature stored at position q of a permuted block
__device__ __forceinline__ int unperm32(int q) { return (q & ~31) + 8 * ((q >> 2) & 3) + 4 * ((q >> 4) & 1) + (q & 3); }

// the f32 weight behind A-operand k slot (h, i8) of [layer][mt][ks], row `row`
__device__ __forceinline__ float split_weight(const IqnWeights &w, int layer, int mt, int ks, int h, int row, int i8) {
    if (layer == 1) {                  // mt = feature tile t, ks = cos K step s
        const int f = 32 * mt + row;
        return f < F ? w.W1[f * N_COS + 16 * ks + 8 * h + i8] : 0.f;
    }
    const int feat = 32 * (ks >> 1) + 8 * (2 * (ks & 1) + (i8 >> 2)) + 4 * h + (i8 & 3);
    if (layer == 2) return feat < F ? w.W2[(32 * mt + row) * F + feat] : 0.f;
    if (layer == 3) return w.W3[(32 * mt + row) * H + feat];
    return row < A_OUT ? w.W4[row * H + feat] : 0.f;
}

// 32-bit word i of the image
__device__ __forceinline__ uint32_t pack_word(const IqnWeights &w, const float *__restrict__ consts, int i) {
    if (i < OFF_B1) {
        const int u4 = i >> 2, pair = i & 3, lane = u4 & 63, piece = (u4 >> 6) & 1, h = lane >> 5, row = lane & 31;
        int q = u4 >> 7, layer, mt, ks;
        if (q < NTILE * 4) { layer = 1; mt = q >> 2; ks = q & 3; }
        else if (q < NTILE * 4 + 2 * KS2) { q -= NTILE * 4; layer = 2; mt = q / KS2; ks = q % KS2; }
        else if (q < NTILE * 4 + 2 * KS2 + 8) { q -= NTILE * 4 + 2 * KS2; layer = 3; mt = q >> 2; ks = q & 3; }
        else { q -= NTILE * 4 + 2 * KS2 + 8; layer = 4; mt = 0; ks = q; }
        const float sc = layer < 4 ? consts[layer - 1] : consts[10];
        uint32_t out = 0;
        for (int j = 0; j < 2; ++j) {
            const float x = split_weight(w, layer, mt, ks, h, row, 2 * pair + j) * sc;
            const _Float16 hi = (_Float16)x;
            const _Float16 v = piece == 0 ? hi : (_Float16)(x - (float)hi);
            out |= sp::half_bits(v) << (16 * j);
        }
        return out;
    }
    float v;
    if (i < OFF_B2) {                // 2^k1 b1, permuted; it is the layer-1 accumulators' initial value
        const int f = unperm32(i - OFF_B1);
        v = f < F ? w.b1[f] * consts[0] : 0.f;
    } else if (i < OFF_B3) v = w.b2[unperm32(i - OFF_B2)];
    else if (i < OFF_B4) v = w.b3[unperm32(i - OFF_B3)];
    else if (i < OFF_BND) v = (i - OFF_B4) < A_OUT ? w.b4[i - OFF_B4] : 0.f;
    else if (i < OFF_CST) {          // B1_j = sum_k |W1_jk| + |b1_j|, inflated against the rounding of the sum
        const int j = i - OFF_BND;
        float s = fabsf(w.b1[j]);
        for (int k = 0; k < N_COS; ++k) s += fabsf(w.W1[j * N_COS + k]);
        v = s * 1.0009765625f;
    } else if (i < OFF_WS) {         // CST[j] = consts[3 + j]: c1 c2 c3 a2 d2 a3 d3 2^k4 2^-k4
        const int j = i - OFF_CST;
        v = 3 + j < sp::N_CONST ? consts[3 + j] : 0.f;
    } else if (i < OFF_WVG) {        // WS[i4][sf][c] = se_w[sf][4 i4 + c] (22 inputs, zero padded to 24)
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

__global__ __launch_bounds__(256) void iqn_split32_pack_kernel(IqnWeights w, const float *__restrict__ consts, uint32_t *__restrict__ packed) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < OFF_FB) packed[i] = pack_word(w, consts, i);
}

// weight image (when stale; consts from sp::iqn_split_consts_kernel earlier in the stream) + the call's random numbers
__global__ __launch_bounds__(256) void iqn_split32_prep_kernel(IqnWeights w, const float *__restrict__ consts, uint32_t *__restrict__ packed,
                                                               const uint64_t *__restrict__ rng_state, float *__restrict__ draws, int n,
                                                               const float *__restrict__ cvar_row, float cvar, int pack_blocks) {
    if ((int)blockIdx.x < pack_blocks) {
        const int i = blockIdx.x * blockDim.x + threadIdx.x;
        if (i < OFF_FB) packed[i] = pack_word(w, consts, i);
        return;
    }
    draw_block(rng_state, draws, n, cvar_row, cvar, pack_blocks);
}

__device__ __forceinline__ f32x16 mf(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

// LDS addressing as in namespace sp: opaque base registers + compile-time offsets < 64 KB
struct LdsBase {
    int w0, w1, w2;   // lane + 0 / 4096 / 8192 : 16-byte units of the weight image
    int fl;           // (OFF_B1 >> 2) + 4 h    : permuted bias vectors (16-byte units)
    int fb;           // this wave's permuted feature buffer + 4 h (16-byte units)
};
__device__ __forceinline__ u32x4 ld_w(const u32x4 *__restrict__ lds4, const LdsBase &lb, int c) {     // c: unit index without the lane
    return c < 4096 - 64 ? lds4[lb.w0 + c] : (c < 8192 - 64 ? lds4[lb.w1 + (c - 4096)] : lds4[lb.w2 + (c - 8192)]);
}
// a lane's 16 values of a permuted vector (tile `t` of the vector that starts `off_units` after lb's base)
__device__ __forceinline__ f32x16 ld_vec16(const f32x4 *__restrict__ ldsv, int base, int t) {
    const f32x4 a = ldsv[base + 8 * t], b = ldsv[base + 8 * t + 1], c = ldsv[base + 8 * t + 2], d = ldsv[base + 8 * t + 3];
    return (f32x16){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3], c[0], c[1], c[2], c[3], d[0], d[1], d[2], d[3]};
}

// One pipeline stage of the fused layers 1 + 2 (cf. sp::stage).  Stage T issues, as one hand-interleaved stream of 24 MFMA slots,
//   * the 12 layer-2 MFMAs of the two K steps fed by feature tile T     (inputs: bh / bl),
//   * the 12 layer-1 MFMAs of feature tile T + 2                       (into accW, which starts from the bias),
//   * the VALU epilogue of tile T + 1                                   (accR -> bhN / blN: ReLU, Hadamard, split), one sub-step per slot.
// Stages -2 and -1 fill the pipeline; tile 6 (the half-empty one) is fed to layer 2 by tail().
template <int T>
__device__ __forceinline__ void stage(const u32x4 *__restrict__ lds4, const f32x4 *__restrict__ ldsv, const LdsBase &lb,
                                      const f16x8 (&cbh)[4], const f16x8 (&cbl)[4], const f16x8 (&bh)[2], const f16x8 (&bl)[2],
                                      f32x16 (&acc2)[2], f32x16 &accW, const f32x16 &accR, f16x8 (&bhN)[2], f16x8 (&blN)[2]) {
    constexpr bool HAS_L2 = T >= 0, HAS_L1 = T + 2 < NTILE, HAS_EP = T + 1 >= 0 && T + 1 < NTILE;
    constexpr int N_L2 = HAS_L2 ? 12 : 0, N_L1 = HAS_L1 ? 12 : 0, NM = N_L2 + N_L1;
    constexpr int N_PAIR = HAS_EP ? ((T + 1 == NTILE - 1) ? 4 : 8) : 0, N_SUB = 3 * N_PAIR;      // the last tile's upper half is padding

    f16x8 a2h[2][2], a2l[2][2];        // [K step of the tile][mt]
    if (HAS_L2) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int c = W2_U4 + ((mt * KS2 + 2 * (HAS_L2 ? T : 0) + ks) * 2) * 64;
                a2h[ks][mt] = __builtin_bit_cast(f16x8, ld_w(lds4, lb, c));
                a2l[ks][mt] = __builtin_bit_cast(f16x8, ld_w(lds4, lb, c + 64));
            }
    }
    f32x16 fv, bias;
    if (HAS_EP) fv = ld_vec16(ldsv, lb.fb, T + 1);                       // S 2^-k1 features of tile T + 1, this lane's 16 rows
    if (HAS_L1) bias = ld_vec16(ldsv, lb.fl, T + 2);                     // 2^k1 b1 of tile T + 2: the accumulator's initial value
    f16x8 a1h[4], a1l[4];
    f16x2 hP[8], lP[8];
    const f16x2 zero2 = {(_Float16)0.f, (_Float16)0.f};
#pragma unroll
    for (int q = 0; q < 8; ++q) { hP[q] = zero2; lP[q] = zero2; }
    float x0 = 0.f, x1 = 0.f, r0 = 0.f, r1 = 0.f;
    f16x2 hcur = zero2;
    __builtin_amdgcn_sched_barrier(0);

    static_for<NM>([&](auto M_) {
        constexpr int m = decltype(M_)::value;
        // LDS reads of the layer-1 weights: K steps 0, 1 at the first slot, 2, 3 eight slots later
        constexpr int LOAD2 = N_L2 >= 12 ? 8 : 0;      // K steps 2, 3 are first used in slot N_L2 + 6
        if (HAS_L1 && (m == 0 || m == LOAD2)) {
#pragma unroll
            for (int s = 0; s < (LOAD2 == 0 ? 4 : 2); ++s) {
                const int ss = (m == 0 ? 0 : 2) + s;
                const int c = W1_U4 + (((HAS_L1 ? T + 2 : 0) * 4 + ss) * 2) * 64;
                a1h[ss] = __builtin_bit_cast(f16x8, ld_w(lds4, lb, c));
                a1l[ss] = __builtin_bit_cast(f16x8, ld_w(lds4, lb, c + 64));
            }
        }
        // the MFMA of this slot
        if constexpr (m < N_L2) {
            constexpr int ks = m / 6, r = m % 6, p = r / 2, mt = r % 2;
            acc2[mt] = mf(p == 0 ? a2l[ks][mt] : a2h[ks][mt], p == 1 ? bl[ks] : bh[ks], acc2[mt]);
        } else {
            constexpr int q = m - N_L2, s = q / 3, p = q % 3;
            accW = mf(p == 0 ? a1l[s] : a1h[s], p == 1 ? cbl[s] : cbh[s], q == 0 ? bias : accW);
        }
        // epilogue sub-steps of this slot: sub-step u goes after MFMA (u + 1) NM / (N_SUB + 1)
#pragma unroll
        for (int sub = 0; sub < N_SUB; ++sub) {
            if ((sub + 1) * NM / (N_SUB + 1) == m) {
                const int q = sub / 3;                 // register pair 2q, 2q + 1 of the tile
                if (sub % 3 == 0) {                    // ReLU + Hadamard: 2 v_max_i32, v_pk_mul_f32
                    x0 = relu1(accR[2 * q]) * fv[2 * q];
                    x1 = relu1(accR[2 * q + 1]) * fv[2 * q + 1];
                } else if (sub % 3 == 1) {             // hi pair, residuals: v_cvt_pk_f16_f32, 2 v_fma_mix_f32
                    hcur = cvt_pair(x0, x1);
                    residual_pair(x0, x1, hcur, r0, r1);
                    hP[q] = hcur;
                } else {                               // lo pair: v_cvt_pk_f16_f32
                    lP[q] = cvt_pair(r0, r1);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    });
    if (HAS_EP) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bhN[ks] = cat4(hP[4 * ks], hP[4 * ks + 1], hP[4 * ks + 2], hP[4 * ks + 3]);
            blN[ks] = cat4(lP[4 * ks], lP[4 * ks + 1], lP[4 * ks + 2], lP[4 * ks + 3]);
        }
    }
}

// unscale + bias, ReLU and split of one half (registers 8 s2 .. 8 s2 + 7) of a C tile -> the B operands of one K step, cut into
// 12 sub-steps (4 register pairs x 3) that a caller places between MFMAs
struct HalfSplit {
    f16x2 hP[4], lP[4];
    float t0, t1, r0, r1;
};
template <int SUB, int S2>
__device__ __forceinline__ void half_substep(const f32x16 &acc, float cx, const f32x16 &sb, HalfSplit &hs) {
    constexpr int q = SUB / 3, reg = 8 * S2 + 2 * q;
    if constexpr (SUB % 3 == 0) {
        hs.t0 = relu1(fmaf(acc[reg], cx, sb[reg]));
        hs.t1 = relu1(fmaf(acc[reg + 1], cx, sb[reg + 1]));
    } else if constexpr (SUB % 3 == 1) {
        const f16x2 h = cvt_pair(hs.t0, hs.t1);
        residual_pair(hs.t0, hs.t1, h, hs.r0, hs.r1);
        hs.hP[q] = h;
    } else {
        hs.lP[q] = cvt_pair(hs.r0, hs.r1);
    }
}
__device__ __forceinline__ void half_finish(const HalfSplit &hs, f16x8 &bh, f16x8 &bl) {
    bh = cat4(hs.hP[0], hs.hP[1], hs.hP[2], hs.hP[3]);
    bl = cat4(hs.lP[0], hs.lP[1], hs.lP[2], hs.lP[3]);
}

// The end of an environment's pipeline: the layer-2 K step of the half tile 6, the layer-2 epilogue, layer 3, its epilogue and the
// output layer, MFMAs and VALU sub-steps interleaved wherever two independent pieces exist.  Leaves the quantile-value tile
// (rows = actions, padded to 32; columns = taus; scaled by S 2^k4) in acc4.
__device__ __forceinline__ void tail(const u32x4 *__restrict__ lds4, const f32x4 *__restrict__ ldsv, const LdsBase &lb, float c2, float c3, float S2, float S3,
                                     const f16x8 (&bh)[2], const f16x8 (&bl)[2], f32x16 (&acc2)[2], f32x16 &acc4) {
    f16x8 a2h[2], a2l[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int c = W2_U4 + ((mt * KS2 + (KS2 - 1)) * 2) * 64;
        a2h[mt] = __builtin_bit_cast(f16x8, ld_w(lds4, lb, c));
        a2l[mt] = __builtin_bit_cast(f16x8, ld_w(lds4, lb, c + 64));
    }
    f32x16 sb2[2], sb3[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        sb2[mt] = ld_vec16(ldsv, lb.fl + ((OFF_B2 - OFF_B1) >> 2), mt) * S2;
        sb3[mt] = ld_vec16(ldsv, lb.fl + ((OFF_B3 - OFF_B1) >> 2), mt) * S3;
    }
    f16x8 a3h[4][2], a3l[4][2], a4h[4], a4l[4];
    f16x8 b3h[4], b3l[4], b4h[4], b4l[4];
    f32x16 acc3[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc3[mt][i] = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc4[i] = 0.f;
    HalfSplit hs;
    __builtin_amdgcn_sched_barrier(0);

    // ---- A: layer 2, last K step (6 MFMAs); layer-3 weights of K steps 0, 1 requested
    static_for<6>([&](auto M_) {
        constexpr int m = decltype(M_)::value, p = m / 2, mt = m % 2;
        if (m == 0) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int mt3 = 0; mt3 < 2; ++mt3) {
                    const int c = W3_U4 + ((mt3 * 4 + ks) * 2) * 64;
                    a3h[ks][mt3] = __builtin_bit_cast(f16x8, ld_w(lds4, lb, c));
                    a3l[ks][mt3] = __builtin_bit_cast(f16x8, ld_w(lds4, lb, c + 64));
                }
        }
        acc2[mt] = mf(p == 0 ? a2l[mt] : a2h[mt], p == 1 ? bl[0] : bh[0], acc2[mt]);
        __builtin_amdgcn_sched_barrier(0);
    });
    // ---- B: layer-2 epilogue of tile 0, lower half (feeds layer-3 K step 0) -- nothing to hide it under
    static_for<12>([&](auto U_) { half_substep<decltype(U_)::value, 0>(acc2[0], c2, sb2[0], hs); });
    half_finish(hs, b3h[0], b3l[0]);
    __builtin_amdgcn_sched_barrier(0);
    // ---- C: layer 3 (24 MFMAs, K step major) || the remaining three halves of the layer-2 epilogue, two sub-steps per slot
    static_for<24>([&](auto M_) {
        constexpr int m = decltype(M_)::value, ks = m / 6, r = m % 6, p = r / 2, mt3 = r % 2;
        if (m == 0) {
#pragma unroll
            for (int k2 = 2; k2 < 4; ++k2)
#pragma unroll
                for (int m3 = 0; m3 < 2; ++m3) {
                    const int c = W3_U4 + ((m3 * 4 + k2) * 2) * 64;
                    a3h[k2][m3] = __builtin_bit_cast(f16x8, ld_w(lds4, lb, c));
                    a3l[k2][m3] = __builtin_bit_cast(f16x8, ld_w(lds4, lb, c + 64));
                }
        }
        if (m == 12) {
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
                const int c = W4_U4 + (k4 * 2) * 64;
                a4h[k4] = __builtin_bit_cast(f16x8, ld_w(lds4, lb, c));
                a4l[k4] = __builtin_bit_cast(f16x8, ld_w(lds4, lb, c + 64));
            }
        }
        acc3[mt3] = mf(p == 0 ? a3l[ks][mt3] : a3h[ks][mt3], p == 1 ? b3l[ks] : b3h[ks], acc3[mt3]);
        if constexpr (m < 18) {              // K step ks + 1's operands are produced during K step ks
            constexpr int nk = ks + 1, u = 2 * (m % 6);
            half_substep<u, nk & 1>(acc2[nk >> 1], c2, sb2[nk >> 1], hs);
            half_substep<u + 1, nk & 1>(acc2[nk >> 1], c2, sb2[nk >> 1], hs);
            if constexpr (m % 6 == 5) half_finish(hs, b3h[nk], b3l[nk]);
        }
        __builtin_amdgcn_sched_barrier(0);
    });
    // ---- D: layer-3 epilogue of tile 0, lower half (feeds output-layer K step 0)
    static_for<12>([&](auto U_) { half_substep<decltype(U_)::value, 0>(acc3[0], c3, sb3[0], hs); });
    half_finish(hs, b4h[0], b4l[0]);
    __builtin_amdgcn_sched_barrier(0);
    // ---- E: output layer (12 MFMAs) || the remaining three halves of the layer-3 epilogue, four sub-steps per slot
    static_for<12>([&](auto M_) {
        constexpr int m = decltype(M_)::value, ks = m / 3, p = m % 3;
        acc4 = mf(p == 0 ? a4l[ks] : a4h[ks], p == 1 ? b4l[ks] : b4h[ks], acc4);
        if constexpr (m < 9) {
            constexpr int nk = ks + 1, u = 4 * (m % 3);
            half_substep<u, nk & 1>(acc3[nk >> 1], c3, sb3[nk >> 1], hs);
            half_substep<u + 1, nk & 1>(acc3[nk >> 1], c3, sb3[nk >> 1], hs);
            half_substep<u + 2, nk & 1>(acc3[nk >> 1], c3, sb3[nk >> 1], hs);
            half_substep<u + 3, nk & 1>(acc3[nk >> 1], c3, sb3[nk >> 1], hs);
            if constexpr (m % 3 == 2) half_finish(hs, b4h[nk], b4l[nk]);
        }
        __builtin_amdgcn_sched_barrier(0);
    });
}

// sum over the 32 lanes of a half wave (lanes sharing l >> 5)
__device__ __forceinline__ float half_sum32(float v) {
    v = row_sum16(v);
    return v + __shfl_xor(v, 16);
}

__global__ __launch_bounds__(64 * sp::WAVES) void iqn_qvals_split32_kernel(const float *__restrict__ obs, const float *__restrict__ taus,
                                                                         const uint32_t *__restrict__ packed, float *__restrict__ qvals,
                                                                         const float *__restrict__ explore_u, float eps,
                                                                         int32_t *__restrict__ actions, int n, uint64_t *__restrict__ rng_state) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    if (rng_state && blockIdx.x == 0 && tid == 0) rng_state[1] += 1;   // the draws of this call were made by the prep kernel
    {
        const u32x4 *src = reinterpret_cast<const u32x4 *>(packed);
        u32x4 *dst = reinterpret_cast<u32x4 *>(lds);
        for (int i = tid; i < OFF_FB / 4; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();

    const int lane = tid & 63, h = lane >> 5, col = lane & 31;
    const int wave = tid >> 6, waves_per_block = blockDim.x >> 6;
    const f32x4 *ldsv = reinterpret_cast<const f32x4 *>(lds);
    const u32x4 *lds4 = reinterpret_cast<const u32x4 *>(lds);
    LdsBase lb;
    lb.w0 = lane; lb.w1 = lane + 4096; lb.w2 = lane + 8192;
    lb.fl = (OFF_B1 >> 2) + 4 * h; lb.fb = ((OFF_FB + wave * FP) >> 2) + 4 * h;
    int enc_w = (OFF_WS >> 2) + lane;       // sensor encoder weights (16-byte units)
    int enc_f = OFF_BND + lane;             // bounds / encoder biases (floats)
    int fb_w = OFF_FB + wave * FP;          // this wave's permuted feature buffer (floats)
    asm volatile("" : "+v"(lb.w0), "+v"(lb.w1), "+v"(lb.w2), "+v"(lb.fl), "+v"(lb.fb), "+v"(enc_w), "+v"(enc_f));
    const float c1 = lds[OFF_CST + 0], c2 = lds[OFF_CST + 1], c3 = lds[OFF_CST + 2];
    const float a2 = lds[OFF_CST + 3], d2 = lds[OFF_CST + 4], a3 = lds[OFF_CST + 5], d3 = lds[OFF_CST + 6], c4 = lds[OFF_CST + 8];
    // cos(tau * pi * k), k = 16 s + 8 h + i: v_cos_f32 takes its argument in revolutions (tau * k / 2 <= 32) and reduces it itself
    const float hk0 = 4.0f * (float)h;      // k / 2 = hk0 + (8 s + i / 2)

    for (int e = blockIdx.x * waves_per_block + wave; e < n; e += gridDim.x * waves_per_block) {
        const float tau = taus[(size_t)e * K_TAUS + col];
        // layer-1 B operands: the cos embedding (model.py:155), unscaled, split
        f16x8 cbh[4], cbl[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f16x2 hh[4], ll[4];
#pragma unroll
            for (int p = 0; p < 4; ++p)
                split2(__builtin_amdgcn_cosf(tau * (hk0 + (8.0f * s + 0.5f * (2 * p)))),
                       __builtin_amdgcn_cosf(tau * (hk0 + (8.0f * s + 0.5f * (2 * p + 1)))), hh[p], ll[p]);
            cbh[s] = cat4(hh[0], hh[1], hh[2], hh[3]);
            cbl[s] = cat4(ll[0], ll[1], ll[2], ll[3]);
        }
        // observation encoders, per-environment scale S, S 2^-k1 features -> this wave's (permuted) LDS buffer
        sp::EnvScale sc;
        {
            const float *orow = obs + (size_t)__builtin_amdgcn_readfirstlane(e) * OBS;
            float ov[28];
#pragma unroll
            for (int i = 0; i < 28; ++i) ov[i] = i < OBS ? orow[i] : 0.f;
            sp::EncState st;
            static_for<sp::N_ENC_SUB>([&](auto I_) { sp::enc_substep<decltype(I_)::value>(lds, ldsv, enc_w, enc_f, lane, ov, st, OFF_WVG); });
            sc = sp::env_scale(st.bnd, a2, d2, a3, d3);
            const float Sc = sc.S1 * c1;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int sf = lane + 64 * j;
                if (sf < 176) lds[fb_w + perm32(32 + sf)] = st.fval[j] * Sc;
            }
            if (lane < 32) lds[fb_w + perm32(lane)] = st.fval[3] * Sc;
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }

        f32x16 acc2[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc2[mt][i] = 0.f;
        f32x16 accA, accB;
        f16x8 bhA[2], blA[2], bhB[2], blB[2];
        stage<-2>(lds4, ldsv, lb, cbh, cbl, bhB, blB, acc2, accA, accA, bhB, blB);      // layer-1 tile 0 (no epilogue yet: accR unused)
        stage<-1>(lds4, ldsv, lb, cbh, cbl, bhB, blB, acc2, accB, accA, bhA, blA);      // layer-1 tile 1, epilogue of tile 0
        stage<0>(lds4, ldsv, lb, cbh, cbl, bhA, blA, acc2, accA, accB, bhB, blB);
        stage<1>(lds4, ldsv, lb, cbh, cbl, bhB, blB, acc2, accB, accA, bhA, blA);
        stage<2>(lds4, ldsv, lb, cbh, cbl, bhA, blA, acc2, accA, accB, bhB, blB);
        stage<3>(lds4, ldsv, lb, cbh, cbl, bhB, blB, acc2, accB, accA, bhA, blA);
        stage<4>(lds4, ldsv, lb, cbh, cbl, bhA, blA, acc2, accA, accB, bhB, blB);
        stage<5>(lds4, ldsv, lb, cbh, cbl, bhB, blB, acc2, accB, accA, bhA, blA);
        f32x16 acc4;
        tail(lds4, ldsv, lb, c2 * sc.r21, c3 * sc.r32, sc.S2, sc.S3, bhA, blA, acc2, acc4);

        // ---- tau mean of the quantile values: lane (h, c) holds actions 4 h + r of tau c in registers r = 0..3 and (h = 0) action 8
        // in register 4; Q(s, a) = mean over the 32 taus, unscaled by 2^-k4 / S (model.py:185,190)
        const float s0 = half_sum32(acc4[0]), s1 = half_sum32(acc4[1]), s2 = half_sum32(acc4[2]), s3 = half_sum32(acc4[3]);
        const float s8 = half_sum32(acc4[4]);
        const int r4 = lane & 3;
        const float mine = r4 == 0 ? s0 : (r4 == 1 ? s1 : (r4 == 2 ? s2 : s3));     // action 4 h + (lane & 3)
        const float got = __shfl(mine, lane < 4 ? lane : 32 + (lane & 3));          // lanes 4..7 take half 1's value
        const float qraw = lane == 8 ? s8 : got;
        const float qv = qraw * (sc.invS3 * c4 * (1.0f / K_TAUS)) + lds[OFF_B4 + (lane & 15)];     // Q(s, action = lane), valid for lane < 9
        if (qvals && lane < A_OUT) qvals[(size_t)e * A_OUT + lane] = qv;
        // ---- IQNAgent.act epilogue (agent.py:199-203): argmax, epsilon-greedy ------------------------
        if (actions) {
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

}  // namespace sp32
