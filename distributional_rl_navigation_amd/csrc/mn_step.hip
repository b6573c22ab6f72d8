// mn_step.hip -- fused marinenav_env step kernel for gfx950 (MI355X): one MarineNavEnv.step (marinenav_env.py:199-262)
// for every environment of the batch in one launch, optionally with the replay append of agent.py:124 fused in
// (mn_step_append).  The step itself -- mapping, arithmetic, numerics -- is MnLane::step in mn_step_body.h; this file is
// the kernel around it (loads, stores, the done-queue hand-off to the reset kernel) and its launcher.
#include "mn_step_body.h"

namespace {

template <typename M, bool PARITY, int L, bool APPEND>
__global__ __launch_bounds__(MN_STEP_BLOCK, 2) void mn_step_kernel(MnArrays A, MnDev P, const int32_t *__restrict__ actions,
                                                      float *__restrict__ obs_out, float *__restrict__ reward_out,
                                                      uint8_t *__restrict__ done_out, uint8_t *__restrict__ info_out,
                                                      int parity, MnRing R) {
    using Lane = MnLane<M, PARITY, L>;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int e = tid / L;        // environment (< npad: the grid covers exactly npad * L lanes)
    const int q = tid % L;        // lane within the env's group

    if (tid < MN_QSHARDS) A.queue_count[(parity ^ 1) * MN_QWORDS + tid * MN_QSTRIDE] = 0u;  // counters the NEXT step will fill

    Lane ln;
    ln.load(A, e, q);
    const int action = ln.active ? actions[e] : 0;
    // Fused replay append (mn_step_append): this lane's share of the obs_t row -- the same float2 columns it will
    // write of obs_t+1 -- is loaded here with everything else, so the transition leaves in this launch instead of a
    // separate append kernel that re-reads both observation tiles.
    float2 prev_head[2], prev_beam[Lane::BPL];
    if constexpr (APPEND) {
        const float2 *prow = reinterpret_cast<const float2 *>(R.prev_obs + (size_t)(ln.active ? e : 0) * MN_OBS_DIM);
        prev_head[0] = prow[0]; prev_head[1] = prow[1];
#pragma unroll
        for (int j = 0; j < Lane::BPL; ++j) {
            const int b = q + L * j;
            prev_beam[j] = prow[2 + (b < MN_NUM_BEAMS ? b : MN_NUM_BEAMS - 1)];
        }
    }

    const MnStepOut o = ln.template step<APPEND>(A, P, action, obs_out + (size_t)e * MN_OBS_DIM,
                                                 (PARITY && A.obs64) ? A.obs64 + (size_t)e * MN_OBS_DIM : nullptr, R, prev_head, prev_beam);
    ln.store(A);
    if (ln.active && q == 0) {
        reward_out[e] = (float)o.reward;
        done_out[e] = (uint8_t)o.done;
        info_out[e] = (uint8_t)o.info;
    }
    // done-queue: one atomic per wave, on the counter of the wave's shard (its 64 / L envs lie in one aligned group of 64: mn_internal.h)
    {
        const bool mine = ln.active && o.done && q == 0;
        const unsigned long long m = __ballot(mine);
        if (m) {
            const int lane = threadIdx.x & (MN_WAVE - 1);
            const int leader = __ffsll((long long)m) - 1;
            const int sh = __shfl((e >> 6) & (MN_QSHARDS - 1), leader);
            unsigned base = 0;
            if (lane == leader) base = atomicAdd(&A.queue_count[parity * MN_QWORDS + sh * MN_QSTRIDE], (unsigned)__popcll(m));
            base = __shfl(base, leader);
            if (mine) A.queue[(size_t)sh * A.qcap + base + __popcll(m & ((1ull << lane) - 1ull))] = e;
        }
    }
}

template <typename M, bool PARITY, bool APPEND>
void launch_l(int lanes, const MnArrays &A, const MnDev &P, const int32_t *actions, float *obs, float *reward,
              uint8_t *done, uint8_t *info, int parity, const MnRing &R, hipStream_t s) {
    const dim3 block(MN_STEP_BLOCK);
#define MN_LAUNCH(LL)                                                                                              \
    hipLaunchKernelGGL((mn_step_kernel<M, PARITY, LL, APPEND>), dim3((unsigned)((size_t)A.npad * LL / MN_STEP_BLOCK)), block, 0, s, A, P, \
                       actions, obs, reward, done, info, parity, R)
    // Default (lanes == 0), measured on MI355X: up to ~128 K envs the launch is latency-bound and two lanes per env
    // win (19.6 vs 20.5 us at 65 536); beyond that several rounds of waves hide latency by themselves and the
    // mapping with the least total work wins (1 M envs: 121 us at L = 1 -> 44 % of the HBM roofline, 165 us at L = 2).
    // Round 3: the kernel that also appends the transition (the training loop's) wants four -- with the obstacle rotation shared by the
    // lane group more lanes no longer repeat it, and the transition's loads / stores spread over the group: 65 536 envs, float64, in the
    // loop 27.9 -> 25.2 us per launch (mixed 22.1 -> 21.0; 16 384 / 32 768 / 131 072 envs likewise); the plain step kernel stays at two
    // (24.6 vs 25.9 us).
    if (lanes == 0) lanes = A.n <= 131072 ? (APPEND ? 4 : 2) : 1;
    switch (lanes) {
        case 1: MN_LAUNCH(1); break;
        case 4: MN_LAUNCH(4); break;
        case 8: MN_LAUNCH(8); break;
        default: MN_LAUNCH(2); break;
    }
#undef MN_LAUNCH
}

}  // namespace

void mn_launch_step(const MnArrays &A, const MnDev &P, int precision, int lanes, const int32_t *actions, float *obs,
                    float *reward, uint8_t *done, uint8_t *info, int parity, const MnRing *ring, hipStream_t s) {
    static const MnRing none = {};
    if (precision == MN_PRECISION_F64) {
        if (ring) launch_l<double, true, true>(lanes, A, P, actions, obs, reward, done, info, parity, *ring, s);
        else launch_l<double, true, false>(lanes, A, P, actions, obs, reward, done, info, parity, none, s);
    } else {
        if (ring) launch_l<float, false, true>(lanes, A, P, actions, obs, reward, done, info, parity, *ring, s);
        else launch_l<float, false, false>(lanes, A, P, actions, obs, reward, done, info, parity, none, s);
    }
}
