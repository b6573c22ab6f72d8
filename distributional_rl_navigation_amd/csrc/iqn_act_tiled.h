// iqn_act_tiled.h -- launch-shared taus, second form: the MFMA columns are ENVIRONMENTS (round 4).
// Included by iqn_act.hip inside its anonymous namespace, after iqn_act_split.h (uses namespace sp's helpers and image offsets).
//
// With one set of 32 taus per launch (mn_iqn_set_tau_mode), layer 1 is a constant h1[tau][j] of the launch and layer 2 of environment e
// at quantile tau is
//     h2pre[e][tau][i] = sum_j W2[i][j] h1[tau][j] f[e][j] + b2[i] = sum_j T[tau][i][j] f[e][j] + b2[i],      T[tau][i][j] := W2[i][j] h1[tau][j].
// T depends on the launch only, f (the encoder features) on the environment only.  iqn_act_split.h's shared-tau kernel keeps the taus in
// the MFMA columns and forms the Hadamard product h1 . f and its hi / lo split for every (environment, tau, feature): 312 of its 729
// vector instructions per environment.  Here the product moves into the WEIGHTS: the preparation launch builds T once ([32][64][208], as
// hi / lo f16 pairs in A-operand order, 1.8 MB), a wavefront takes 32 environments as two 16-column tiles, splits their features ONCE
// (the B operand of all 32 taus), and loops over the taus with T[tau] streamed through a double-buffered LDS tile:
//     per tau and 16 environments   84 MFMAs layer 2 (4 output tiles x 7 K blocks x 3 products) + 24 MFMAs layer 3,
//                                   the layer-2 epilogue + split (64 vector instructions), the layer-3 epilogue + tau sum (48)
// i.e. per environment the same 216 matrix instructions, but ~260 vector instructions instead of 729, and no cross-lane tau sum at
// all (a lane's column IS one environment: the tau mean is a running sum in registers).  The encoders run as exact-f32 MFMAs
// (v_mfma_f32_16x16x4_f32, block-diagonal 208 x 32 matrix): their C tiles are the B operand's k slots, as everywhere in this file family.
// A workgroup (8 waves = 256 environments) streams all of T once, so the form only pays for large batches: the dispatcher uses it from
// TILED_MIN_ENVS environments up and the wave-per-environment kernel below that (mn_iqn_set_tau_mode(ctx, 3) forces it: tests).
// Where the time goes (scripts/act_tiled_ablation.sh, profiles/r04_act_tiled_ablation.txt, 65 536 rows incl. 12 us of preparation launches):
// 192 us as built; 162 without the layer-2 / -3 epilogues; 179 without the per-tau barrier and T prefetch; 172 without the layer-3 MFMAs;
// 127 without the layer-2 MFMAs; 110 without any MFMA; 68 without MFMAs, epilogues and barrier -- i.e. the matrix instructions add only
// ~80 us on top of ~110 us of everything else (LDS reads of T: 14.7 MB per CU and launch, epilogues, barriers, encoders).  Same arithmetic classes as the other split-f16 kernels
// (three f16 products per float32 product, power-of-two scaling from guaranteed bounds); one more float32 rounding in T = W2 h1.

#ifndef TILED_ABL
#define TILED_ABL 0      // measurement builds only (scripts/act_tiled_ablation.sh; results are wrong by construction): 1 no layer-2 / -3 epilogues,
#endif                   // 2 no per-tau barrier and no T prefetch, 4 no layer-3 MFMAs, 8 no layer-2 MFMAs, 16 no LDS reads of T

namespace sp {

// A workgroup takes 256 environments through all 32 taus in ~170 us whatever the batch: the form wins once every CU has a workgroup
// (65 536 rows on a 256-CU chip: 186 vs 198 us per launch incl. the preparation launches) and loses below (the wavefront-per-row kernel
// scales down with the batch: 16 384 rows in ~55 us).
constexpr int TILED_MIN_ENVS = 65536;
constexpr int T_U4_PER_TAU = 4 * KB2 * 2 * 64;               // 3 584 16-byte units = 57 344 B: [mt][kb][piece][lane]
constexpr int T_WORDS = K_TAUS * T_U4_PER_TAU * 4;           // 458 752 32-bit words
// auxiliary float block behind T (indices in floats): the encoders as one block-diagonal matrix in dense A-fragment order, their biases
constexpr int TA_WE = 0;                                     // [13 mt][2 kt][64 lanes][4 r]: We[16 mt + row][16 kt + 4 g + r], 26 inputs zero-padded to 32
constexpr int TA_BE = TA_WE + T1 * 2 * 256;                  // [208]
constexpr int TA_CST = TA_BE + F;                            // [0] 1 / scaleT
constexpr int TA_FLOATS = TA_CST + 16;
constexpr int T_PREP_BLOCKS = (T_WORDS + 255) / 256, TA_PREP_BLOCKS = (TA_FLOATS + 255) / 256;
// LDS of the act kernel (16-byte units / floats)
constexpr int TL_T0 = 0, TL_T1 = T_U4_PER_TAU, TL_W3 = 2 * T_U4_PER_TAU;      // two T buffers, then W3 hi / lo [4 mt][2 kb][piece][lane]
constexpr int TL_W3_U4 = 4 * 2 * 2 * 64;
constexpr int TL_FL = (TL_W3 + TL_W3_U4) * 4;                // float part: b2 [64] | b3 [64] | B1 bound [208] | W4 [9][64] | b4 [16] | consts [16]
constexpr int TL_B2 = TL_FL, TL_B3 = TL_B2 + H, TL_BND = TL_B3 + H, TL_W4 = TL_BND + F, TL_B4 = TL_W4 + A_OUT * H, TL_CST = TL_B4 + 16;
constexpr int TL_FLOATS = TL_CST + 16;
static_assert(TL_FLOATS * 4 <= 160 * 1024, "LDS of the tiled act kernel");
static_assert(TL_B2 % 4 == 0 && TL_B3 % 4 == 0 && TL_BND % 4 == 0 && TL_W4 % 4 == 0 && TL_B4 % 4 == 0, "16-byte aligned blocks");

__device__ __forceinline__ float h1_at(const float *__restrict__ h1, int tau, int j) {      // iqn_shared_prep_kernel's layout
    return h1[(((j >> 4) * NT + (tau >> 4)) * 64 + ((j >> 2) & 3) * 16 + (tau & 15)) * 4 + (j & 3)];
}

// T (hi / lo, A-operand order, scaled by scaleT = 2^k2 p with max(h1) p in [0.5, 1)) + the auxiliary block, from the weights and the
// launch's layer-1 constant h1 (+ its 26 per-block maxima behind it).  One thread per 32-bit word.
__global__ __launch_bounds__(256) void iqn_tiled_prep_kernel(IqnWeights w, const float *__restrict__ consts, const float *__restrict__ h1,
                                                             uint32_t *__restrict__ timg, float *__restrict__ taux) {
    float hmax = 0.f;
    for (int b = 0; b < H1_BLOCKS; ++b) hmax = fmaxf(hmax, h1[H1_FLOATS + b]);
    // p = 2^-(e+1) for hmax in [2^e, 2^(e+1)): hmax p in [0.5, 1); degenerate (all of layer 1 dead): 1
    float p = 1.0f;
    if (hmax > 1e-30f && hmax < 1e30f) p = __builtin_bit_cast(float, (uint32_t)(253 - (int)(__builtin_bit_cast(uint32_t, hmax) >> 23)) << 23);
    const float scale_t = consts[1] * p;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if ((int)blockIdx.x >= T_PREP_BLOCKS) {
        const int k = i - T_PREP_BLOCKS * 256;
        if (k >= TA_FLOATS) return;
        float v = 0.f;
        if (k < TA_BE) {
            const int r = k & 3, lane = (k >> 2) & 63, q = k >> 8, kt = q & 1, mt = q >> 1;
            const int f = 16 * mt + (lane & 15), in = 16 * kt + 4 * (lane >> 4) + r;
            if (f < 16) v = in < 2 ? w.ve_w[f * 2 + in] : 0.f;
            else if (f < 32) v = (in >= 2 && in < 4) ? w.ge_w[(f - 16) * 2 + (in - 2)] : 0.f;
            else v = (in >= 4 && in < OBS) ? w.se_w[(f - 32) * 22 + (in - 4)] : 0.f;
        } else if (k < TA_CST) {
            const int f = k - TA_BE;
            v = f < 16 ? w.ve_b[f] : (f < 32 ? w.ge_b[f - 16] : w.se_b[f - 32]);
        } else if (k == TA_CST) v = 1.0f / scale_t;      // exact: a power of two
        taux[k] = v;
        return;
    }
    if (i >= T_WORDS) return;
    const int tau = i / (T_U4_PER_TAU * 4), k = i % (T_U4_PER_TAU * 4);
    const int u4 = k >> 2, pair = k & 3, lane = u4 & 63, piece = (u4 >> 6) & 1, q = u4 >> 7, kb = q % KB2, mt = q / KB2;
    const int g = lane >> 4, row = 16 * mt + (lane & 15);
    uint32_t out = 0;
    for (int jj = 0; jj < 2; ++jj) {
        const int i8 = 2 * pair + jj, j = 16 * (2 * kb + (i8 >> 2)) + 4 * g + (i8 & 3);
        const float x = j < F ? (w.W2[row * F + j] * h1_at(h1, tau, j)) * scale_t : 0.f;
        const _Float16 hi = (_Float16)x;
        const _Float16 v = piece == 0 ? hi : (_Float16)(x - (float)hi);
        out |= half_bits(v) << (16 * jj);
    }
    timg[i] = out;
}

// per-environment scales as env_scale(), but one environment per LANE COLUMN (every lane of a column ends up with its env's values)
// The B operand here is the feature vector itself (the layer-1 factor sits in T), so ITS scale Sf comes from max |f_j| -- |Sf f_j| < 2^15 --
// while the layer-2 / -3 activation bounds still follow from m1 = max_j B1_j |f_j| >= |h1_j f_j| as in env_scale().
struct ColScale {
    float Sf, S2, S3, c2e, c3e, invS3;      // c2e = 2^-kT S2 / Sf, c3e = 2^-k3 S3 / S2
};
__device__ __forceinline__ ColScale col_scale(float bnd, float fmx, float a2, float d2, float a3, float d3, float inv_scale_t, float c3) {
    bnd = fmaxf(bnd, __shfl_xor(bnd, 16)); fmx = fmaxf(fmx, __shfl_xor(fmx, 16));
    bnd = fmaxf(bnd, __shfl_xor(bnd, 32)); fmx = fmaxf(fmx, __shfl_xor(fmx, 32));
    const int ef = bound_exponent(fmx), e2 = bound_exponent(fmaf(a2, bnd, d2)), e3 = bound_exponent(fmaf(a3, bnd, d3));
    ColScale sc;
    sc.Sf = __builtin_bit_cast(float, (uint32_t)(268 - ef) << 23);
    sc.S2 = __builtin_bit_cast(float, (uint32_t)(268 - e2) << 23);
    sc.S3 = __builtin_bit_cast(float, (uint32_t)(268 - e3) << 23);
    sc.c2e = inv_scale_t * __builtin_bit_cast(float, (uint32_t)(127 + ef - e2) << 23);
    sc.c3e = c3 * __builtin_bit_cast(float, (uint32_t)(127 + e2 - e3) << 23);
    sc.invS3 = __builtin_bit_cast(float, (uint32_t)(e3 - 14) << 23);
    return sc;
}

constexpr int TC = 2;      // 16-environment column tiles per wavefront

__global__ __launch_bounds__(512) void iqn_qvals_tiled_kernel(const float *__restrict__ obs, const uint32_t *__restrict__ packed,
                                                              const uint32_t *__restrict__ timg, const float *__restrict__ taux,
                                                              float *__restrict__ qvals, const float *__restrict__ explore_u, float eps,
                                                              int32_t *__restrict__ actions, int n, uint64_t *__restrict__ rng_state) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    u32x4 *lds4 = reinterpret_cast<u32x4 *>(lds);
    const f32x4 *ldsv = reinterpret_cast<const f32x4 *>(lds);
    typedef __attribute__((address_space(3))) u32x4 lds_u4;
    typedef __attribute__((address_space(1))) u32x4 glb_u4;
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, col = lane & 15, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    if (rng_state && blockIdx.x == 0 && tid == 0) rng_state[1] += 1;      // the draws of this call were made by the prep kernel
    const u32x4 *t4 = reinterpret_cast<const u32x4 *>(timg);
    {   // T[0], W3, and the float part gathered from the weight image
        for (int i = tid; i < T_U4_PER_TAU; i += 512) lds4[TL_T0 + i] = t4[i];
        const u32x4 *img4 = reinterpret_cast<const u32x4 *>(packed);
        for (int i = tid; i < TL_W3_U4; i += 512) lds4[TL_W3 + i] = img4[W3_U4 + i];
        const float *imgf = reinterpret_cast<const float *>(packed);
        for (int i = tid; i < H; i += 512) { lds[TL_B2 + i] = imgf[OFF_B2 + i]; lds[TL_B3 + i] = imgf[OFF_B3 + i]; }
        for (int i = tid; i < F; i += 512) lds[TL_BND + i] = imgf[OFF_BND + i];
        for (int i = tid; i < A_OUT * H; i += 512) {      // W4[a][feat] from the image's [t2][l][r] order: l = 16 g + a, feat = 16 t2 + 4 g + r
            const int a = i / H, feat = i % H;
            lds[TL_W4 + i] = imgf[OFF_W4 + ((feat >> 4) * 64 + ((feat >> 2) & 3) * 16 + a) * 4 + (feat & 3)];
        }
        if (tid < 16) { lds[TL_B4 + tid] = imgf[OFF_B4 + tid]; lds[TL_CST + tid] = imgf[OFF_CST + tid]; }
    }
    __syncthreads();
    const float c3 = lds[TL_CST + 2], a2 = lds[TL_CST + 3], d2 = lds[TL_CST + 4], a3 = lds[TL_CST + 5], d3 = lds[TL_CST + 6];
    const float inv_scale_t = taux[TA_CST];
    const int e0 = (blockIdx.x * 8 + wave) * (16 * TC);

    // ---- encoders (exact f32 MFMA), per-environment scales, the feature split: the B operand of all 32 taus ----------------------------
    f16x8 fh[TC][KB2], fl[TC][KB2];
    ColScale sc[TC];
    const f32x4 *we4 = reinterpret_cast<const f32x4 *>(taux + TA_WE) + lane;
    const f32x4 *be4 = reinterpret_cast<const f32x4 *>(taux + TA_BE) + g;
#pragma unroll
    for (int c = 0; c < TC; ++c) {
        const int e = e0 + 16 * c + col;
        const bool live = e < n;
        const float *row = obs + (size_t)(live ? e : 0) * OBS;
        f32x4 x0[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = 16 * kt + 4 * g + r;
                x0[kt][r] = (live && k < OBS) ? row[k < OBS ? k : 0] : 0.f;
            }
        f32x4 f[T1 + 1];
        float bnd = 0.f, fmx = 0.f;
#pragma unroll
        for (int mt = 0; mt < T1; ++mt) {
            f32x4 acc = be4[4 * mt];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                const f32x4 a = we4[(mt * 2 + kt) * 64];
#pragma unroll
                for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], x0[kt][r], acc, 0, 0, 0);
            }
            f[mt] = acc;
            const f32x4 b1 = ldsv[(TL_BND >> 2) + 4 * mt + g];
#pragma unroll
            for (int r = 0; r < 4; ++r) { bnd = fmaxf(bnd, fabsf(acc[r]) * b1[r]); fmx = fmaxf(fmx, fabsf(acc[r])); }
        }
        f[T1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        sc[c] = col_scale(bnd, fmx, a2, d2, a3, d3, inv_scale_t, c3);
#pragma unroll
        for (int kb = 0; kb < KB2; ++kb) split_tiles(f[2 * kb] * sc[c].Sf, f[2 * kb + 1] * sc[c].Sf, fh[c][kb], fl[c][kb]);
    }

    // ---- the tau loop -------------------------------------------------------------------------------------------------------------------
    f32x4 hs[TC][4];
#pragma unroll
    for (int c = 0; c < TC; ++c)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) hs[c][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int PF = T_U4_PER_TAU / 512;      // 7 16-byte units of the next tau's tile per thread
    static_assert(T_U4_PER_TAU % 512 == 0 && PF == KB2, "prefetch shape: one unit per thread and K block");
    int lbase = lane;      // opaque LDS index base (keeps hipcc from materialising one address register per read)
    asm volatile("" : "+v"(lbase));
    // (not unrolled: unrolled by two -- hipcc's choice, the buffer toggle becomes a constant -- the kernel needs 15 more registers than there are: 60 bytes of scratch
    // per lane, 12 MB of spill traffic per launch; round 5: 187 -> 181 us)
#pragma unroll 1
    for (int tau = 0; tau < K_TAUS; ++tau) {
        const int cur = ((tau & 1) ? TL_T1 : TL_T0) + lbase;
        const bool more = tau + 1 < K_TAUS;
        const u32x4 *tnext = t4 + (size_t)(more ? tau + 1 : tau) * T_U4_PER_TAU + tid;
        // where this WAVE's 1 KB pieces of the next tile land: LDS-DMA writes wave-uniform base + lane x 16 bytes (round 5: was global -> 8 registers -> ds_write)
        lds_u4 *nxt_wave = (lds_u4 *)(lds) + ((tau & 1) ? TL_T0 : TL_T1) + 64 * wave_u;
        // layer 2: acc2[mt][c] = T[tau] (Sf f), three f16 products per float32 product.  One software-pipelined stream of 28 steps
        // (K block, output tile): the A operands of step s + 2 are requested from LDS before the six MFMAs of step s are issued, so a
        // wave hides its own LDS latency (the per-tau barrier keeps the waves of a workgroup in step: a partner wave is in the same
        // phase, not in another one).  The next tau's tile travels global -> the other LDS buffer by LDS-DMA (global_load_lds_dwordx4: no staging registers, no
        // ds_write) in the same stream, one 1 KB piece per wave and K block (the buffer was last read one iteration ago, before the barrier that ended it; the
        // barrier that ends THIS iteration waits for the pieces to land: hipcc puts the vmcnt(0) in front of it).
        f32x4 acc2[4][TC];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int c = 0; c < TC; ++c) acc2[mt][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#ifndef TILED_AHEAD
#define TILED_AHEAD 2
#endif
        constexpr int NS = 4 * KB2, AHEAD = TILED_AHEAD;
        u32x4 ah[AHEAD + 1], al[AHEAD + 1];
#pragma unroll
        for (int s0 = 0; s0 < AHEAD; ++s0) {
            ah[s0] = lds4[cur + (((s0 & 3) * KB2 + (s0 >> 2)) * 2) * 64];
            al[s0] = lds4[cur + (((s0 & 3) * KB2 + (s0 >> 2)) * 2 + 1) * 64];
        }
        static_for<NS>([&](auto S_) {
            constexpr int s = decltype(S_)::value, kb = s >> 2, mt = s & 3, slot = s % (AHEAD + 1);
            if constexpr (s + AHEAD < NS) {
                constexpr int s2 = s + AHEAD, slot2 = s2 % (AHEAD + 1);
                ah[slot2] = lds4[cur + (((s2 & 3) * KB2 + (s2 >> 2)) * 2) * 64];
                al[slot2] = lds4[cur + (((s2 & 3) * KB2 + (s2 >> 2)) * 2 + 1) * 64];
            }
            if constexpr (mt == 0 && !(TILED_ABL & 2)) {      // next tau, piece kb of this wave
                if (more) __builtin_amdgcn_global_load_lds((const glb_u4 *)(tnext + kb * 512), nxt_wave + kb * 512, 16, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);      // (left alone hipcc sinks the global load to its use and waits for it there)
            const f16x8 a_h = __builtin_bit_cast(f16x8, ah[slot]), a_l = __builtin_bit_cast(f16x8, al[slot]);
            if constexpr (!(TILED_ABL & 8)) {
#pragma unroll
            for (int c = 0; c < TC; ++c) acc2[mt][c] = mf(a_l, fh[c][kb], acc2[mt][c]);
#pragma unroll
            for (int c = 0; c < TC; ++c) acc2[mt][c] = mf(a_h, fl[c][kb], acc2[mt][c]);
#pragma unroll
            for (int c = 0; c < TC; ++c) acc2[mt][c] = mf(a_h, fh[c][kb], acc2[mt][c]);
            } else {
#pragma unroll
                for (int c = 0; c < TC; ++c) acc2[mt][c] += __builtin_bit_cast(f32x4, ah[slot]) + __builtin_bit_cast(f32x4, al[slot]);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        // layer-2 epilogue: S2 h2 = relu(acc2 c2e + S2 b2), split in place: the B operands of layer 3
        f16x8 b3h[2][TC], b3l[2][TC];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const f32x4 bb0 = ldsv[(TL_B2 >> 2) + 4 * (2 * kb) + g], bb1 = ldsv[(TL_B2 >> 2) + 4 * (2 * kb + 1) + g];
#pragma unroll
            for (int c = 0; c < TC; ++c) {
                if (TILED_ABL & 1) { b3h[kb][c] = __builtin_bit_cast(f16x8, acc2[2 * kb][c]); b3l[kb][c] = __builtin_bit_cast(f16x8, acc2[2 * kb + 1][c]); continue; }
                split_tiles(relu4s(fma4(acc2[2 * kb][c], sc[c].c2e, bb0 * sc[c].S2)), relu4s(fma4(acc2[2 * kb + 1][c], sc[c].c2e, bb1 * sc[c].S2)),
                            b3h[kb][c], b3l[kb][c]);
            }
        }
        // layer 3
        f32x4 acc3[4][TC];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int c = 0; c < TC; ++c) acc3[mt][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const f16x8 ah = __builtin_bit_cast(f16x8, lds4[TL_W3 + lbase + ((mt * 2 + kb) * 2) * 64]);
                const f16x8 al = __builtin_bit_cast(f16x8, lds4[TL_W3 + lbase + ((mt * 2 + kb) * 2 + 1) * 64]);
                if (TILED_ABL & 4) {
#pragma unroll
                    for (int c = 0; c < TC; ++c) acc3[mt][c] += __builtin_bit_cast(f32x4, ah) + __builtin_bit_cast(f32x4, b3h[kb][c]) + __builtin_bit_cast(f32x4, b3l[kb][c]);
                    continue;
                }
#pragma unroll
                for (int c = 0; c < TC; ++c) acc3[mt][c] = mf(al, b3h[kb][c], acc3[mt][c]);
#pragma unroll
                for (int c = 0; c < TC; ++c) acc3[mt][c] = mf(ah, b3l[kb][c], acc3[mt][c]);
#pragma unroll
                for (int c = 0; c < TC; ++c) acc3[mt][c] = mf(ah, b3h[kb][c], acc3[mt][c]);
            }
        // layer-3 epilogue + the tau sum (a column is one environment: a running sum, no cross-lane work)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const f32x4 bb = ldsv[(TL_B3 >> 2) + 4 * mt + g];
#pragma unroll
            for (int c = 0; c < TC; ++c) hs[c][mt] += (TILED_ABL & 1) ? acc3[mt][c] : relu4s(fma4(acc3[mt][c], sc[c].c3e, bb * sc[c].S3));
        }
        if (!(TILED_ABL & 2)) {
        __syncthreads();      // every wave has read T[tau], and its pieces of T[tau + 1] have landed
        }
    }

    // ---- output layer on the tau mean (linear: W4 mean(h3) + b4), argmax, epsilon-greedy --------------------------------------------------
#pragma unroll
    for (int c = 0; c < TC; ++c) {
        const int e = e0 + 16 * c + col;
        float q[A_OUT];
#pragma unroll
        for (int a = 0; a < A_OUT; ++a) {
            float part = 0.f;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const f32x4 w4 = ldsv[(TL_W4 >> 2) + a * (H / 4) + 4 * mt + g];      // W4[a][16 mt + 4 g + r]
#pragma unroll
                for (int r = 0; r < 4; ++r) part = fmaf(w4[r], hs[c][mt][r], part);
            }
            part += __shfl_xor(part, 16);
            part += __shfl_xor(part, 32);
            q[a] = part * (sc[c].invS3 * (1.0f / K_TAUS)) + lds[TL_B4 + a];
        }
        if (e < n && g == 0) {
            if (qvals)
#pragma unroll
                for (int a = 0; a < A_OUT; ++a) qvals[(size_t)e * A_OUT + a] = q[a];
            if (actions) {
                float best = q[0];
                int arg = 0;
#pragma unroll
                for (int a = 1; a < A_OUT; ++a)
                    if (q[a] > best) { best = q[a]; arg = a; }
                if (explore_u && eps > 0.f) {
                    const float u = explore_u[e];               // greedy iff u > eps (agent.py:200)
                    if (!(u > eps)) { arg = (int)(u / eps * (float)A_OUT); arg = arg > A_OUT - 1 ? A_OUT - 1 : arg; }
                }
                actions[e] = arg;
            }
        }
    }
}

}  // namespace sp
