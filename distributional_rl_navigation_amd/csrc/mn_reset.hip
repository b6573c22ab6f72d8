// mn_reset.hip -- world generation / episode reset kernel for gfx950 (MI355X): one wavefront per finished environment
// (mn_reset_env, mn_reset_body.h), env indices pulled from the done-queue the step kernel filled -- plus the small
// kernels around the RNG streams (seeding, mask -> queue, peek).
#include "mn_reset_body.h"

#ifndef MN_RESET_MAX_BLOCKS
#define MN_RESET_MAX_BLOCKS 8192u   // waves launched for a queue-driven reset (each loops over queue entries)
#endif

namespace {

// `seen` (may be NULL): {device word, host-mapped word}.  Every queue-driven reset launch updates a decaying peak of the number of episodes it starts,
// peak <- max(count, peak - peak / 8), in the device word and copies it to the host-mapped one, which the host reads WITHOUT synchronising: its estimate
// of how many episodes end per vector step (mn_reset_done_async decides with it where the next reset runs).  Launches of one handle are sequential.
struct MnSeen { uint32_t *peak_dev; uint32_t *peak_host; };
__device__ __forceinline__ void mn_note_count(const MnSeen seen, uint32_t count) {
    if (seen.peak_dev && blockIdx.x == 0 && threadIdx.x == 0) {
        uint32_t p = *seen.peak_dev;
        p -= p >> 3;
        p = p > count ? p : count;
        *seen.peak_dev = p;
        __hip_atomic_store(seen.peak_host, p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// The handle's done-queue of one step as a reset wavefront reads it (mn_internal.h: MN_QSHARDS lists, each with its own counter).  open(): lane s < 32 reads
// shard s's count, `incl` = inclusive prefix sum over the shards (kept in one register), returns the total.  at(qi): the qi-th finished env overall.
struct MnDoneQueue {
    uint32_t incl;
    __device__ __forceinline__ uint32_t open(const uint32_t *__restrict__ counts) {
        const int lane = threadIdx.x & (MN_WAVE - 1);
        uint32_t v = lane < MN_QSHARDS ? counts[lane * MN_QSTRIDE] : 0u;
#pragma unroll
        for (int off = 1; off < MN_QSHARDS; off <<= 1) {
            const uint32_t t = __shfl_up(v, off);
            if (lane >= off) v += t;
        }
        incl = v;
        return (uint32_t)__builtin_amdgcn_readlane((int)v, MN_QSHARDS - 1);
    }
    __device__ __forceinline__ int at(const int32_t *__restrict__ lists, int qcap, uint32_t qi) const {
        const int lane = threadIdx.x & (MN_WAVE - 1);
        const int sh = __popcll(__ballot(lane < MN_QSHARDS && incl <= qi));      // shards that end at or before entry qi
        const uint32_t before = sh ? (uint32_t)__builtin_amdgcn_readlane((int)incl, __builtin_amdgcn_readfirstlane(sh - 1)) : 0u;
        return lists[(size_t)sh * qcap + (qi - before)];
    }
};

template <typename M, bool PARITY>
__global__ __launch_bounds__(MN_WAVE) void mn_reset_kernel(MnArrays A, MnDev P, const uint32_t *__restrict__ count_dev,
                                                           uint32_t count_host, const int32_t *__restrict__ list, int mode,
                                                           float *__restrict__ obs_out, const MnSeen seen, int qcap) {
    __shared__ MtLds S;
    __shared__ WorldLds W;
    MnDoneQueue Q;
    const uint32_t count = qcap ? Q.open(count_dev) : (count_dev ? *count_dev : count_host);      // qcap != 0: the sharded done-queue
    mn_note_count(seen, count);
    for (uint32_t qi = blockIdx.x; qi < count; qi += gridDim.x)
        mn_reset_env<M, PARITY>(A, P, S, W, qcap ? Q.at(list, qcap, qi) : (list ? list[qi] : (int)qi), mode, obs_out);
}

// The same body for mn_reset_done_async: it runs on the handle's side stream UNDER the act kernel of the next vector step, whose 512-thread
// workgroups hold every CU with two wavefronts of 208 registers per SIMD and all but 5.8 KB of the LDS -- so this one is compiled for the 96 registers
// that are left on every SIMD (five waves per SIMD as the occupancy target) and keeps the MT19937 block in device memory (MtInPlace: 0.6 KB of LDS per
// wave instead of 3.1), so that FOUR reset wavefronts fit beside an act workgroup, one per SIMD (round 5: one per CU); every finished env announces its
// first observation (`ready`).
template <typename M, bool PARITY>
__global__ __launch_bounds__(MN_WAVE, 5) void mn_reset_under_act_kernel(MnArrays A, MnDev P, const uint32_t *__restrict__ count_dev,
                                                                        const int32_t *__restrict__ list, float *__restrict__ obs_out,
                                                                        uint32_t *__restrict__ ready, uint32_t tick, const MnSeen seen) {
    __shared__ MtCarryLds S;
    __shared__ WorldLds W;
    MnDoneQueue Q;
    const uint32_t count = Q.open(count_dev);
    mn_note_count(seen, count);
    for (uint32_t qi = blockIdx.x; qi < count; qi += gridDim.x)
        mn_reset_env<M, PARITY, MtInPlace>(A, P, S, W, Q.at(list, A.qcap, qi), 0, obs_out, ready, tick);
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
                     const int32_t *list_dev, int mode, float *obs, hipStream_t s, uint32_t *peak_dev, uint32_t *peak_host, bool sharded) {
    const int qcap = sharded ? A.qcap : 0;
    const MnSeen seen = {peak_dev, peak_host};
    // enough waves to fill the chip several times over; each wave loops over queue entries
    uint32_t cap = count_dev ? (uint32_t)A.n : count_host;
    uint32_t blocks = cap < MN_RESET_MAX_BLOCKS ? cap : MN_RESET_MAX_BLOCKS;
    if (blocks == 0) return;
    if (precision == MN_PRECISION_F64)
        hipLaunchKernelGGL((mn_reset_kernel<double, true>), dim3(blocks), dim3(MN_WAVE), 0, s, A, P, count_dev, count_host, list_dev, mode, obs, seen, qcap);
    else
        hipLaunchKernelGGL((mn_reset_kernel<float, false>), dim3(blocks), dim3(MN_WAVE), 0, s, A, P, count_dev, count_host, list_dev, mode, obs, seen, qcap);
}

void mn_launch_reset_under_act(const MnArrays &A, const MnDev &P, int precision, const uint32_t *count_dev, const int32_t *list_dev, float *obs,
                               uint32_t *ready, uint32_t tick, uint32_t *peak_dev, uint32_t *peak_host, hipStream_t s) {
    const MnSeen seen = {peak_dev, peak_host};
    uint32_t blocks = (uint32_t)A.n < MN_RESET_MAX_BLOCKS ? (uint32_t)A.n : MN_RESET_MAX_BLOCKS;
    if (blocks == 0) return;
    if (precision == MN_PRECISION_F64)
        hipLaunchKernelGGL((mn_reset_under_act_kernel<double, true>), dim3(blocks), dim3(MN_WAVE), 0, s, A, P, count_dev, list_dev, obs, ready, tick, seen);
    else
        hipLaunchKernelGGL((mn_reset_under_act_kernel<float, false>), dim3(blocks), dim3(MN_WAVE), 0, s, A, P, count_dev, list_dev, obs, ready, tick, seen);
}

// one wavefront that does nothing for `us` microseconds (mn_debug_side_delay_us)
__global__ void mn_sleep_kernel(uint32_t us) {
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();      // 100 MHz
    while (__builtin_amdgcn_s_memrealtime() - t0 < 100ull * us) __builtin_amdgcn_s_sleep(64);
}
void mn_launch_sleep(uint32_t us, hipStream_t s) { hipLaunchKernelGGL(mn_sleep_kernel, dim3(1), dim3(MN_WAVE), 0, s, us); }

void mn_launch_seed(const MnArrays &A, const uint32_t *seeds_dev, hipStream_t s) {
    hipLaunchKernelGGL(mn_seed_kernel, dim3((A.n + 255) / 256), dim3(256), 0, s, A, seeds_dev);
}

void mn_launch_mask_to_queue(const MnArrays &A, const uint8_t *mask, uint32_t *count, int32_t *list, hipStream_t s) {
    hipLaunchKernelGGL(mn_mask_to_queue_kernel, dim3((A.n + 255) / 256), dim3(256), 0, s, A, mask, count, list);
}

void mn_launch_peek(const MnArrays &A, int first, int count, double *out_dev, hipStream_t s) {
    hipLaunchKernelGGL(mn_peek_kernel, dim3((count + 255) / 256), dim3(256), 0, s, A, first, count, out_dev);
}
