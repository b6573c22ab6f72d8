// Internal device-side structures shared by the gfx950 kernels and the C-ABI host code.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "marinenav_hip.h"

#define MN_WAVE 64
#define MN_PAD 256
#define MN_FIX_SCALE 16777216.0            // 2^24
#define MN_FIX_INV (1.0 / 16777216.0)
#define MN_TAB_ROWS (3 * MN_MAX_CORES + 3 * MN_MAX_OBS)

// Device-resident state.  Everything is struct-of-arrays with the env index fastest, padded to a
// multiple of 64 envs (one wavefront tile), so lane i of a wave touches element base+i of every
// array.  npad is a multiple of 256 so that any lanes-per-env setting fills whole workgroups.
struct MnArrays {
    int32_t n, npad;
    // robot pose (robot.py:40-44); kept in float64 in both precisions
    double *x, *y, *theta, *speed, *vx, *vy;
    // per-env episode constants
    double *start_x, *start_y, *goal_x, *goal_y, *init_theta, *init_speed;
    int32_t *ep_t;    // marinenav_env.py:70 episode_timesteps
    int64_t *tot_t;   // marinenav_env.py:71 total_timesteps
    int32_t *counts;  // placed cores | placed obstacles << 8
    // world tables, [k][npad], generation order (order matters for the sonar `break` quirk)
    double *cx, *cy, *cg;    // cg = +Gamma if clockwise else -Gamma
    double *ox, *oy, *orad;
    // compact copy of the tables read by the mixed-precision step kernel (half the bytes):
    // positions as int32 fixed point (2^-24 m: 6e-8 m resolution everywhere on the 50 m map, where a
    // float32 would have 3.8e-6), signed Gamma and radius as float32.  Written by the reset kernel.
    int32_t *qcx, *qcy, *qox, *qoy;
    float *qcg, *qor;
    // numpy RandomState streams: [n][624] key words + position in the block
    uint32_t *mt;
    int32_t *mt_pos;
    // float64 copy of the observations (parity precision only), [npad][26]
    double *obs64;
    double *rew64;  // float64 copy of the last reward (parity precision only), [npad]
    // per-sub-step positions of the last step, [npad][traj_n][2] (parity precision only, allocated by
    // mn_enable_trajectory): what marinenav_env.py:211-212 appends to robot.trajectory
    double *traj;
    int32_t traj_n;
    // done-queue filled by the step kernel, drained by the reset kernel.  SHARDED: env e pushes into shard (e >> 6) % MN_QSHARDS (wave-uniform for every lanes-per-env
    // mapping), every shard has its own counter on its own 128-byte line -- one counter for all (round 1-5) made a vector step with ~2 000 episode ends 11 us longer:
    // that many returning atomics on ONE address are served one after the other (profiles/r06_done_queue.txt)
    uint32_t *queue_count;  // [2][MN_QSHARDS * MN_QSTRIDE], alternating per step; shard s at word s * MN_QSTRIDE
    int32_t *queue;         // [MN_QSHARDS][qcap]
    int32_t qcap;           // entries per shard = ceil(npad / (64 * MN_QSHARDS)) * 64: every env of the shard can be in it
};
#define MN_QSHARDS 32
#define MN_QSTRIDE 32
#define MN_QWORDS (MN_QSHARDS * MN_QSTRIDE)

// Host-derived constants (computed once in double with the host libm so that the generated world
// tables are bit-identical to the numpy reference).
struct MnDev {
    double width, height, core_r, v_rel_max, p, v_lo, v_span, or_lo, or_span, clear_r, goal_dis;
    double timestep_penalty, collision_penalty, goal_reward;
    double min_start_goal_dis, init_theta, init_speed;
    double dt, robot_r, max_speed, k_drag, a[3], w[3];
    double sonar_range;
    double beam_rel[MN_NUM_BEAMS], beam_cos[MN_NUM_BEAMS], beam_sin[MN_NUM_BEAMS];
    double rot_c[3], rot_s[3];  // cos / sin of w[i]*dt: per-sub-step heading rotation
    double fan_sin, fan_cos;    // sin / cos of half the sonar opening angle (work-list wedge test)
    int32_t fan_filter;         // 1 if the fan is a convex wedge (half angle < 90 deg)
    double two_pi;          // 2*pi as python computes it
    double two_pi_r;        // (2*pi)*r              -> Gamma = two_pi_r * v_edge
    double two_pi_vrel;     // (2*pi)*v_rel_max      -> check_core same-direction boundary
    double inv_two_pi_vrel; // its reciprocal: squared-distance pre-test only (the exact rule divides, like the reference)
    double two_pi_r_r;      // ((2*pi)*r)*r          -> compute_speed inside the core
    double binom_q;         // exp(1*log(1-0.5))     -> binomial(1,.5) == (U > binom_q)
    double sg_lo_x, sg_span_x, sg_lo_y, sg_span_y;  // start/goal uniform: 2 + (w-2-2)*U
    double c_span_x, c_span_y;                      // core centre: 0 + w*U
    double o_lo, o_span_x, o_span_y;                // obstacle centre: 5 + (w-5-5)*U
    double timestep_scale;
    int32_t num_cores, num_obs, reset_start_and_goal, random_reset_state, set_boundary, max_episode_steps, N;
    int32_t n_stages;
    int32_t debug_skip;  // read ONLY by -DMN_ABLATION builds (mn_set_debug_skip): 1 = sub-steps, 2 = sonar scan, 4 = sincos, 8 = obstacle rotation, 16 = beam stores
    int64_t sched_t[MN_MAX_STAGES];
    int32_t sched_nc[MN_MAX_STAGES], sched_no[MN_MAX_STAGES];
    double sched_md[MN_MAX_STAGES];
};

// Replay ring the step kernel appends the transition (obs_t, a_t, r_t, obs_t+1, done_t) to (mn_step_append): the
// layout ReplayBuffer.sample hands to the learner (thirdparty/IQN/replay_buffer.py:49-57).  Env e goes to slot
// (ptr + e - first) mod cap for e >= first = max(0, n - cap) (deque(maxlen) keeps only the newest cap rows).
struct MnRing {
    const float *prev_obs;   // [n][26] observations the actions were chosen from (obs_t)
    float *states, *next_states;   // [cap][26]
    int64_t *actions;        // [cap]
    float *rewards, *dones;  // [cap]
    int64_t ptr, cap;
};

// kernels (defined in mn_step.hip / mn_reset.hip)
void mn_launch_step(const MnArrays &A, const MnDev &P, int precision, int lanes, const int32_t *actions, float *obs,
                    float *reward, uint8_t *done, uint8_t *info, int parity, const MnRing *ring, hipStream_t s);
void mn_launch_rollout(const MnArrays &A, const MnDev &P, int precision, int lanes, int n_steps, const int32_t *actions_in,
                       uint64_t seed, uint64_t step0, uint64_t env0, float *obs_out, float *obs_trace, float *reward_trace,
                       uint8_t *done_trace, uint8_t *info_trace, int32_t *action_trace, hipStream_t s);
void mn_launch_random_actions(uint64_t seed, uint64_t step, uint64_t env0, int n, int32_t *out, hipStream_t s);
// episodes under a device-side policy (MN_POLICY_APF / MN_POLICY_BA, mn_planners.h), and one policy step for a vector of observations
void mn_launch_rollout_policy(const MnArrays &A, const MnDev &P, int precision, int n_steps, int policy, float *obs_io, float *obs_trace,
                              float *reward_trace, uint8_t *done_trace, uint8_t *info_trace, int32_t *action_trace, hipStream_t s);
void mn_launch_planner_act(const float *obs, int n, int policy, const double *a, const double *w, int32_t *actions, hipStream_t s);
// mode 0: full reset (RNG); mode 1: pose-only (keeps the loaded world, no RNG)
// (mn_launch_reset: `sharded` = count_dev / list_dev are the handle's done-queue of one step -- MN_QSHARDS counters and lists; otherwise one counter, one list)
void mn_launch_reset_under_act(const MnArrays &A, const MnDev &P, int precision, const uint32_t *count_dev, const int32_t *list_dev, float *obs,
                               uint32_t *ready, uint32_t tick, uint32_t *peak_dev, uint32_t *peak_host, hipStream_t s);
void mn_launch_reset(const MnArrays &A, const MnDev &P, int precision, const uint32_t *count_dev, uint32_t count_host,
                     const int32_t *list_dev, int mode, float *obs, hipStream_t s, uint32_t *peak_dev = nullptr, uint32_t *peak_host = nullptr, bool sharded = false);
void mn_launch_seed(const MnArrays &A, const uint32_t *seeds_dev, hipStream_t s);
void mn_launch_mask_to_queue(const MnArrays &A, const uint8_t *mask, uint32_t *count, int32_t *list, hipStream_t s);
void mn_launch_peek(const MnArrays &A, int first, int count, double *out_dev, hipStream_t s);
void mn_launch_sleep(uint32_t us, hipStream_t s);
