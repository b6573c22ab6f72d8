// mn_reset.hip -- world generation / episode reset kernel for gfx950 (MI355X): one wavefront per finished environment
// (mn_reset_env, mn_reset_body.h), env indices pulled from the done-queue the step kernel filled -- plus the small
// kernels around the RNG streams (seeding, mask -> queue, peek).
#include "mn_reset_body.h"

#ifndef MN_RESET_MAX_BLOCKS
#define MN_RESET_MAX_BLOCKS 8192u   // waves launched for a queue-driven reset (each loops over queue entries)
#endif

namespace {

template <typename M, bool PARITY>
__global__ __launch_bounds__(MN_WAVE) void mn_reset_kernel(MnArrays A, MnDev P, const uint32_t *__restrict__ count_dev,
                                                           uint32_t count_host, const int32_t *__restrict__ list, int mode,
                                                           float *__restrict__ obs_out) {
    __shared__ MtLds S;
    __shared__ WorldLds W;
    const uint32_t count = count_dev ? *count_dev : count_host;
    for (uint32_t qi = blockIdx.x; qi < count; qi += gridDim.x)
        mn_reset_env<M, PARITY>(A, P, S, W, list ? list[qi] : (int)qi, mode, obs_out);
}

// init_genrand (numpy legacy seeding): key[0] = seed, key[i] = 1812433253*(key[i-1]^(key[i-1]>>30)) + i
__global__ void mn_seed_kernel(MnArrays A, const uint32_t *__restrict__ seeds) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= A.n) return;
    uint32_t v = seeds[e];
    uint32_t *k = A.mt + (size_t)e * MT_N;
    k[0] = v;
    for (int i = 1; i < MT_N; ++i) {
        v = 1812433253u * (v ^ (v >> 30)) + (uint32_t)i;
        k[i] = v;
    }
    A.mt_pos[e] = MT_N;  // first draw regenerates
}

__global__ void mn_mask_to_queue_kernel(MnArrays A, const uint8_t *__restrict__ mask, uint32_t *count, int32_t *list) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const bool on = e < A.n && mask[e] != 0;
    const unsigned long long m = __ballot(on);
    if (!m) return;
    const int lane = threadIdx.x & (MN_WAVE - 1);
    unsigned base = 0;
    if (lane == first_lane(m)) base = atomicAdd(count, (unsigned)__popcll(m));
    base = __shfl(base, first_lane(m));
    if (on) list[base + __popcll(m & ((1ull << lane) - 1ull))] = e;
}

// next double of each env's stream without consuming it (lane per env; test hook)
__global__ void mn_peek_kernel(MnArrays A, int first, int count, double *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const int e = first + i;
    const uint32_t *k = A.mt + (size_t)e * MT_N;
    const int pos = A.mt_pos[e];
    uint32_t w[2];
    for (int q = 0; q < 2; ++q) {
        const int p = pos + q;
        if (p < MT_N) w[q] = mt_temper(k[p]);
        else {
            // element p-624 of the NEXT block; needs next-block elements only for index >= 227
            const int i2 = p - MT_N;  // 0 or 1
            w[q] = mt_temper(mt_twist(k[i2], k[i2 + 1], k[i2 + MT_M]));
        }
    }
    out[i] = ((w[0] >> 5) * 67108864.0 + (w[1] >> 6)) / 9007199254740992.0;
}

}  // namespace

void mn_launch_reset(const MnArrays &A, const MnDev &P, int precision, const uint32_t *count_dev, uint32_t count_host,
                     const int32_t *list_dev, int mode, float *obs, hipStream_t s) {
    // enough waves to fill the chip several times over; each wave loops over queue entries
    uint32_t cap = count_dev ? (uint32_t)A.n : count_host;
    uint32_t blocks = cap < MN_RESET_MAX_BLOCKS ? cap : MN_RESET_MAX_BLOCKS;
    if (blocks == 0) return;
    if (precision == MN_PRECISION_F64)
        hipLaunchKernelGGL((mn_reset_kernel<double, true>), dim3(blocks), dim3(MN_WAVE), 0, s, A, P, count_dev, count_host, list_dev, mode, obs);
    else
        hipLaunchKernelGGL((mn_reset_kernel<float, false>), dim3(blocks), dim3(MN_WAVE), 0, s, A, P, count_dev, count_host, list_dev, mode, obs);
}

void mn_launch_seed(const MnArrays &A, const uint32_t *seeds_dev, hipStream_t s) {
    hipLaunchKernelGGL(mn_seed_kernel, dim3((A.n + 255) / 256), dim3(256), 0, s, A, seeds_dev);
}

void mn_launch_mask_to_queue(const MnArrays &A, const uint8_t *mask, uint32_t *count, int32_t *list, hipStream_t s) {
    hipLaunchKernelGGL(mn_mask_to_queue_kernel, dim3((A.n + 255) / 256), dim3(256), 0, s, A, mask, count, list);
}

void mn_launch_peek(const MnArrays &A, int first, int count, double *out_dev, hipStream_t s) {
    hipLaunchKernelGGL(mn_peek_kernel, dim3((count + 255) / 256), dim3(256), 0, s, A, first, count, out_dev);
}
