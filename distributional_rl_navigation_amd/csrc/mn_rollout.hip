// mn_rollout.hip -- T consecutive vector steps of the random-policy workload in ONE launch (gfx950).
//
// BASELINE configs[1] (4 096 envs, random policy, step kernel only) is launch-latency-bound when every vector step is
// its own launch: 14 us per 4 096-env step launch + the reset launch + the action generator, for ~3 us of arithmetic.
// Here a wavefront keeps its environments' pose, counters and world tables in registers (MnLane, mn_step_body.h) and
// runs T steps back to back; the per-step HBM traffic is the outputs only.  Actions are drawn inside the kernel from a
// counter-based generator keyed by (seed, step index, global env index) -- mn_random_actions produces the same draws
// for callers that want to replay a rollout with single launches -- or read from a caller-supplied [T][n] tensor.
// An env that finishes is reset on the spot by its own wavefront (mn_reset_env, mn_reset_body.h: the wave-cooperative
// world generation of the reset kernel, same code, same bits), so the sequence of T x (mn_step, mn_reset_done) and one
// mn_rollout are bit-identical in every output, every counter and every RNG stream.
#ifdef MN_ABLATION
// [0..5] phases of MnLane::step, [6] trace writes + done ballot, [7] in-kernel resets, [8] steps counted, [9] resets counted
__device__ unsigned long long g_rollout_phase[16];
#define MN_PHASE_VAR g_rollout_phase
#endif
#include "mn_planners.h"
#include "mn_reset_body.h"
#include "mn_step_body.h"

// Waves per SIMD the register allocator has to leave room for.  Round 3: with the obstacle tables shared by an env's lane group the
// 8- and 4-lane kernels need 258-290 registers, i.e. a handful of spills buy a second wave per SIMD: 65 536 envs 2.87 -> 4.07 G env
// steps/s (4 096 envs, one wave per two SIMDs, unchanged).  The 2-lane kernel (170 registers over) stays at one.
#ifndef MN_ROLLOUT_MIN_WAVES
#define MN_ROLLOUT_MIN_WAVES(L) ((L) == 4 || (L) == 8 ? 2 : 1)
#endif

namespace {

__device__ __host__ __forceinline__ uint64_t mix64r(uint64_t x) {   // splitmix64 finaliser
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// action of global env `env` at step `step`: uniform over the 9 actions (agent.py:203 `random.choice(np.arange(9))`)
__device__ __forceinline__ int draw_action(uint64_t seed, uint64_t step, uint64_t env) {
    const uint64_t k = mix64r(seed + 0x9E3779B97F4A7C15ull * (step + 1));
    const uint64_t x = mix64r(k ^ (0xD1B54A32D192ED03ull * (env + 1)));
    return (int)__umul64hi(x, (uint64_t)MN_NUM_ACTIONS);
}

__global__ __launch_bounds__(256) void mn_random_actions_kernel(uint64_t seed, uint64_t step, uint64_t env0, int n,
                                                                int32_t *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = draw_action(seed, step, env0 + (uint64_t)i);
}

struct MnTrace {
    float *obs;        // [T][n][26] observation returned by each step (terminal observation for a finished env), or NULL
    float *reward;     // [T][n]
    uint8_t *done;     // [T][n]
    uint8_t *info;     // [T][n]
    int32_t *action;   // [T][n] the action taken
};

template <typename M, bool PARITY, int L>
__global__ __launch_bounds__(MN_WAVE, MN_ROLLOUT_MIN_WAVES(L)) void mn_rollout_kernel(MnArrays A, MnDev P, int n_steps, const int32_t *__restrict__ actions_in,
                                                             uint64_t seed, uint64_t step0, uint64_t env0,
                                                             float *__restrict__ obs_out, MnTrace T) {
    static_assert(MN_STEP_BLOCK == MN_WAVE, "one wavefront per workgroup: the in-kernel reset is wave-cooperative");
    __shared__ MtLds S;
    __shared__ WorldLds W;
    using Lane = MnLane<M, PARITY, L>;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int e = tid / L, q = tid % L;
    const size_t n = (size_t)A.n;
    if (tid < 2 * MN_QSHARDS) A.queue_count[tid * MN_QSTRIDE] = 0u;   // (both parities' shard counters) nothing is left for a later mn_reset_done
    const MnRing none = {};

    Lane ln;
    ln.load(A, e, q);
    bool stepped = false;      // this lane's registers are newer than the handle's arrays
    for (int t = 0; t < n_steps; ++t) {
        int action = 0;
        if (ln.active) action = actions_in ? actions_in[(size_t)t * n + e] : draw_action(seed, step0 + (uint64_t)t, env0 + (uint64_t)e);
        // the row goes to the trace; the LAST step's row also to obs_out, which ends up holding what T x (mn_step,
        // mn_reset_done) would leave there (finished envs: overwritten below with the new episode's first observation)
        const bool last = t == n_steps - 1;
        float *trow = T.obs ? T.obs + ((size_t)t * n + e) * MN_OBS_DIM : nullptr;
        float *orow = obs_out + (size_t)e * MN_OBS_DIM;
        const MnStepOut o = ln.template step<false>(A, P, action, (last || !trow) ? orow : trow,
                                                    (PARITY && A.obs64) ? A.obs64 + (size_t)e * MN_OBS_DIM : nullptr, none, nullptr, nullptr,
                                                    (last && trow) ? trow : nullptr);
        stepped = true;
#ifdef MN_ABLATION
        unsigned long long rt_ = __builtin_amdgcn_s_memtime();
#endif
        if (ln.active && q == 0) {
            const size_t k = (size_t)t * n + e;
            if (T.reward) T.reward[k] = (float)o.reward;
            if (T.done) T.done[k] = (uint8_t)o.done;
            if (T.info) T.info[k] = (uint8_t)o.info;
            if (T.action) T.action[k] = action;
        }
        // in-kernel reset hand-off: the wave resets its finished envs one after the other, all 64 lanes on each
        unsigned long long m = __ballot(ln.active && o.done && q == 0);
#ifdef MN_ABLATION
        if (tid == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); g_rollout_phase[6] += t_ - rt_; g_rollout_phase[8] += 1; rt_ = t_; }
#endif
        if (m) {
            ln.store(A);                       // pose + counters (total_timesteps drives the curriculum lookup)
            while (m) {
                const int src = __ffsll((long long)m) - 1;
                m &= m - 1;
                const int env = __builtin_amdgcn_readlane(e, src);
                mn_reset_env<M, PARITY>(A, P, S, W, env, 0, obs_out);
            }
            __syncthreads();                   // the reset's global writes are visible to this wave's reload
            ln.load(A, e, q);
            stepped = false;
#ifdef MN_ABLATION
            if (tid == 0) { g_rollout_phase[7] += __builtin_amdgcn_s_memtime() - rt_; g_rollout_phase[9] += 1; }
#endif
        }
    }
    // (an env that was reset by the last step keeps the reset's float64 velocity in the arrays, exactly as after
    // mn_reset_done: storing the register copy would round it through the mixed kernel's float32 velocity)
    if (stepped) ln.store(A);
}

// ---- episodes under an observation-reading policy, one launch (mn_rollout_policy; round 4) ------------------------------------------
// The reference's classical-baseline sweeps (run_experiments.py:100-190,213-282: APF / BA on 500 worlds, one episode each) call a python
// policy on every observation; the batched path so far paid a launch trio per policy step.  Here the policy is a device function
// (mn_planners.h) and runs INSIDE the rollout: after a step the lane group parks the observation row in LDS, lane 0 of the group evaluates
// the policy on it -- on the float32 row, exactly what the launch-per-step path feeds planners.py -- and hands the action to the group
// for the next step.  EPISODE semantics: an env that finishes is NOT reset; it idles for the rest of the launch (its traces read reward 0,
// done 1, the terminal info code, action -1), its terminal pose and counters are stored.  Per step the same MnLane::step as everywhere else:
// bit-identical to a loop of (policy launch, mn_step) on the same worlds.
template <typename M, bool PARITY, int L>
__global__ __launch_bounds__(MN_WAVE, 1) void mn_rollout_policy_kernel(MnArrays A, MnDev P, int n_steps, int policy, float *__restrict__ obs_io, MnTrace T) {
    __shared__ float rows[MN_WAVE / L][28];
    using Lane = MnLane<M, PARITY, L>;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int e = tid / L, q = tid % L, slot = (threadIdx.x & (MN_WAVE - 1)) / L;
    const size_t n = (size_t)A.n;
    if (tid < 2 * MN_QSHARDS) A.queue_count[tid * MN_QSTRIDE] = 0u;
    const MnRing none = {};
    MnPlanTabs tabs;
#pragma unroll
    for (int k = 0; k < 3; ++k) { tabs.a[k] = P.a[k]; tabs.w[k] = P.w[k]; }
    Lane ln;
    ln.load(A, e, q);
    // the observation the episode continues from (mn_reset / mn_load_worlds left it in obs_io)
    for (int k = q; k < MN_OBS_DIM; k += L) rows[slot][k] = ln.active ? obs_io[(size_t)e * MN_OBS_DIM + k] : 0.f;
    bool alive = ln.active;
    int last_info = 0;
    for (int t = 0; t < n_steps; ++t) {
        __syncthreads();      // (one wavefront per workgroup) the row of the previous step is complete
        int action = 0;
        if (q == 0 && alive) action = mn_policy_act(policy, rows[slot], tabs);
        action = __shfl(action, (int)(threadIdx.x & (MN_WAVE - 1)) - q);      // from the group's lane 0
        __syncthreads();      // every lane has its action before the step overwrites the row
        float *trow = T.obs ? T.obs + ((size_t)t * n + (ln.active ? e : 0)) * MN_OBS_DIM : nullptr;
        const MnStepOut o = ln.template step<false>(A, P, action, rows[slot], (PARITY && A.obs64 && alive) ? A.obs64 + (size_t)e * MN_OBS_DIM : nullptr, none,
                                                    nullptr, nullptr, (alive && trow) ? trow : nullptr);
        if (ln.active && q == 0) {
            const size_t k = (size_t)t * n + e;
            if (T.reward) T.reward[k] = alive ? (float)o.reward : 0.f;
            if (T.done) T.done[k] = alive ? (uint8_t)o.done : (uint8_t)1;
            if (T.info) T.info[k] = alive ? (uint8_t)o.info : (uint8_t)last_info;
            if (T.action) T.action[k] = alive ? action : -1;
        }
        if (alive) {
            if (o.done) {      // terminal pose, counters and observation of this env are final
                ln.store(A);
                __builtin_amdgcn_wave_barrier();
                for (int k = q; k < MN_OBS_DIM; k += L) obs_io[(size_t)e * MN_OBS_DIM + k] = rows[slot][k];
                last_info = o.info;
                alive = false;
            }
        }
        // (an env that has finished keeps stepping from its terminal pose -- the lane group's cross-lane work is wave-uniform -- but
        // nothing of it is stored or traced -- incl. the float64 copies of mn_enable_obs64, which keep the TERMINAL observation and reward --;
        // its row in LDS no longer feeds a policy call)
        if (!__any(alive)) {      // the whole wave is done: fill the remaining trace entries and leave
            for (int t2 = t + 1; t2 < n_steps; ++t2)
                if (ln.active && q == 0) {
                    const size_t k = (size_t)t2 * n + e;
                    if (T.reward) T.reward[k] = 0.f;
                    if (T.done) T.done[k] = 1;
                    if (T.info) T.info[k] = (uint8_t)last_info;
                    if (T.action) T.action[k] = -1;
                }
            return;
        }
    }
    if (alive) {      // still running after n_steps: the state the next call continues from
        ln.store(A);
        __syncthreads();
        for (int k = q; k < MN_OBS_DIM; k += L) obs_io[(size_t)e * MN_OBS_DIM + k] = rows[slot][k];
    }
}

// one policy step for a whole vector of observation rows (the launch-per-step path of experiments.py)
__global__ __launch_bounds__(256) void mn_planner_act_kernel(const float *__restrict__ obs, int n, int policy, MnPlanTabs tabs, int32_t *__restrict__ actions) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float o[MN_OBS_DIM];
    const float2 *row = reinterpret_cast<const float2 *>(obs + (size_t)i * MN_OBS_DIM);
#pragma unroll
    for (int k = 0; k < MN_OBS_DIM / 2; ++k) { const float2 v = row[k]; o[2 * k] = v.x; o[2 * k + 1] = v.y; }
    actions[i] = mn_policy_act(policy, o, tabs);
}

template <typename M, bool PARITY>
void launch_rollout(int lanes, const MnArrays &A, const MnDev &P, int n_steps, const int32_t *actions_in, uint64_t seed,
                    uint64_t step0, uint64_t env0, float *obs_out, const MnTrace &T, hipStream_t s) {
#define MN_LAUNCH(LL)                                                                                                  \
    hipLaunchKernelGGL((mn_rollout_kernel<M, PARITY, LL>), dim3((unsigned)((size_t)A.npad * LL / MN_WAVE)), dim3(MN_WAVE), 0, s, \
                       A, P, n_steps, actions_in, seed, step0, env0, obs_out, T)
    // Default: a rollout launch is latency-bound per wave (T dependent steps), so small batches want many lanes per env
    // -- 16 lanes up to 4 096 envs, 8 up to 16 K -- and large ones less total work.  There is no
    // 1-lane variant: with 64 envs' tables resident per wave next to the reset code it needs more than the 512 registers
    // a lane can have (the compiler spills to scratch), and batches that large are better served by mn_step launches.
    // Round 3: 16 lanes per env while that still leaves at most one wave per SIMD (4 096 envs = 1 024 waves): four envs per wave, so half
    // the in-kernel resets a wave has to sit through, one obstacle and one beam per lane (4 096 envs: 794 -> 897 M env steps/s; at 8 192
    // envs 8 lanes are faster again, 1 542 vs 1 482 M).
    if (lanes == 0) lanes = A.n <= 4096 ? 16 : (A.n <= 16384 ? 8 : (A.n <= 65536 ? 4 : 2));
    switch (lanes) {
        case 2: MN_LAUNCH(2); break;
        case 4: MN_LAUNCH(4); break;
        case 16: MN_LAUNCH(16); break;
        default: MN_LAUNCH(8); break;
    }
#undef MN_LAUNCH
}

}  // namespace

void mn_launch_rollout(const MnArrays &A, const MnDev &P, int precision, int lanes, int n_steps, const int32_t *actions_in,
                       uint64_t seed, uint64_t step0, uint64_t env0, float *obs_out, float *obs_trace, float *reward_trace,
                       uint8_t *done_trace, uint8_t *info_trace, int32_t *action_trace, hipStream_t s) {
    const MnTrace T = {obs_trace, reward_trace, done_trace, info_trace, action_trace};
    if (precision == MN_PRECISION_F64) launch_rollout<double, true>(lanes, A, P, n_steps, actions_in, seed, step0, env0, obs_out, T, s);
    else launch_rollout<float, false>(lanes, A, P, n_steps, actions_in, seed, step0, env0, obs_out, T, s);
}

#ifdef MN_ABLATION
extern "C" int mn_debug_rollout_phases(unsigned long long *out_host, int reset) {
    if (hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_rollout_phase), sizeof(unsigned long long) * 16) != hipSuccess) return MN_ERR_HIP;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_rollout_phase), z, sizeof(z)) != hipSuccess) return MN_ERR_HIP;
    }
    return MN_OK;
}
#endif

void mn_launch_rollout_policy(const MnArrays &A, const MnDev &P, int precision, int n_steps, int policy, float *obs_io, float *obs_trace,
                              float *reward_trace, uint8_t *done_trace, uint8_t *info_trace, int32_t *action_trace, hipStream_t s) {
    const MnTrace T = {obs_trace, reward_trace, done_trace, info_trace, action_trace};
    constexpr int LL = 8;      // eight lanes per env: the sweeps this serves are a few hundred to a few thousand envs, latency-bound per wave
    const dim3 grid((unsigned)((size_t)A.npad * LL / MN_WAVE));
    if (precision == MN_PRECISION_F64) hipLaunchKernelGGL((mn_rollout_policy_kernel<double, true, LL>), grid, dim3(MN_WAVE), 0, s, A, P, n_steps, policy, obs_io, T);
    else hipLaunchKernelGGL((mn_rollout_policy_kernel<float, false, LL>), grid, dim3(MN_WAVE), 0, s, A, P, n_steps, policy, obs_io, T);
}

void mn_launch_planner_act(const float *obs, int n, int policy, const double *a, const double *w, int32_t *actions, hipStream_t s) {
    MnPlanTabs tabs;
    for (int k = 0; k < 3; ++k) { tabs.a[k] = a[k]; tabs.w[k] = w[k]; }
    hipLaunchKernelGGL(mn_planner_act_kernel, dim3((n + 255) / 256), dim3(256), 0, s, obs, n, policy, tabs, actions);
}

void mn_launch_random_actions(uint64_t seed, uint64_t step, uint64_t env0, int n, int32_t *out, hipStream_t s) {
    hipLaunchKernelGGL(mn_random_actions_kernel, dim3((n + 255) / 256), dim3(256), 0, s, seed, step, env0, n, out);
}
