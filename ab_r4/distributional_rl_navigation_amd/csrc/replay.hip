// replay.hip -- one-launch append of a whole vector step to the device replay ring (gfx950).
//
// Counterpart of ReplayBuffer.add (thirdparty/IQN/replay_buffer.py:26-34) for n transitions at once:
// the reference appends python tuples to a deque(maxlen); here the ring is five HBM tensors in the
// layout sample() hands to the learner (float32 states / next_states [cap][26], int64 actions,
// float32 rewards and dones [cap][1]) and row i of the batch goes to slot (ptr + i) mod cap.
// Pure data movement (~220 B per transition): one coalesced pass instead of ~12 indexed-copy launches.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "marinenav_hip.h"

namespace {

constexpr int ROW4 = MN_OBS_DIM * 4 / 8;  // 26 floats = 13 float2

__global__ __launch_bounds__(256) void replay_append_kernel(const float2 *__restrict__ obs, const int32_t *__restrict__ actions,
                                                            const float *__restrict__ reward, const float2 *__restrict__ next_obs,
                                                            const uint8_t *__restrict__ done, float2 *__restrict__ r_states,
                                                            float2 *__restrict__ r_next, int64_t *__restrict__ r_actions,
                                                            float *__restrict__ r_rewards, float *__restrict__ r_dones,
                                                            int64_t first, int64_t n, int64_t ptr, int64_t cap) {
    // element index over n * 13 float2 of an observation row block
    const int64_t total = n * ROW4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / ROW4, c = i - row * ROW4;
        int64_t slot = ptr + row;
        slot = slot >= cap ? slot - cap : slot;
        const int64_t src = (first + row) * ROW4 + c;
        r_states[slot * ROW4 + c] = obs[src];
        r_next[slot * ROW4 + c] = next_obs[src];
        if (c == 0) {
            r_actions[slot] = (int64_t)actions[first + row];
            r_rewards[slot] = reward[first + row];
            r_dones[slot] = done[first + row] ? 1.0f : 0.0f;
        }
    }
}

}  // namespace

extern "C" int mn_replay_append(const float *obs_dev, const int32_t *actions_dev, const float *reward_dev,
                                const float *next_obs_dev, const uint8_t *done_dev, float *ring_states, float *ring_next_states,
                                int64_t *ring_actions, float *ring_rewards, float *ring_dones, int64_t n, int64_t ptr,
                                int64_t capacity, void *stream) {
    if (!obs_dev || !actions_dev || !reward_dev || !next_obs_dev || !done_dev || !ring_states || !ring_next_states ||
        !ring_actions || !ring_rewards || !ring_dones)
        return MN_ERR_INVALID;
    if (n <= 0 || capacity <= 0 || ptr < 0 || ptr >= capacity) return MN_ERR_INVALID;
    int64_t first = 0;
    if (n > capacity) { first = n - capacity; n = capacity; }   // only the newest `capacity` rows survive (deque maxlen)
    const int64_t total = n * ROW4;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(replay_append_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float2 *>(obs_dev), actions_dev, reward_dev,
                       reinterpret_cast<const float2 *>(next_obs_dev), done_dev, reinterpret_cast<float2 *>(ring_states),
                       reinterpret_cast<float2 *>(ring_next_states), ring_actions, ring_rewards, ring_dones, first, n, ptr, capacity);
    return hipGetLastError() == hipSuccess ? MN_OK : MN_ERR_HIP;
}
