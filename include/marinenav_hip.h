/*
 * marinenav_hip.h -- C-ABI of libmarinenav_hip.so, the MI355X (gfx950) batched marinenav_env.
 *
 * The reference (RobustFieldAutonomyLab/Distributional_RL_Navigation) is pure Python and has no
 * FFI layer; its boundary for this path is the Python class MarineNavEnv
 * (marinenav_env/envs/marinenav_env.py:25-627).  Each entry point below names the reference
 * method(s) it replaces for a batch of `n_envs` independent environments.  A reference
 * maintainer binds these with ctypes; the package's own binding is
 * distributional_rl_navigation_amd/_capi.py.
 *
 * Conventions
 *  - Every function returns 0 on success or a negative mn_status; mn_last_error() gives text.
 *    No C++ exceptions cross the boundary.
 *  - `*_dev` pointers are DEVICE pointers owned by the caller (e.g. torch.Tensor.data_ptr());
 *    `*_host` pointers are host memory.  The library never frees caller memory and never
 *    allocates in mn_step / mn_reset_done / mn_reset.
 *  - Environment state (robot pose, world tables, MT19937 streams) lives in device memory
 *    owned by the handle (allocated in mn_create, released in mn_destroy).
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream).  Kernels are launched
 *    on it with no implicit synchronisation; host<->device accessors (get / set / load) are
 *    synchronous with respect to that handle's previous work on the NULL stream only, so call
 *    them after synchronising your stream.
 *  - One handle per GPU per process; a handle is not re-entrant across threads.
 *    A handle is bound to the HIP device that was current at mn_create(); entry points called while another
 *    device is current return MN_ERR_INVALID (mn_destroy switches to the owning device itself).
 */
#ifndef MARINENAV_HIP_H
#define MARINENAV_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MN_MAX_CORES 8      /* curriculum maximum, train_IQN_model.py:86-90 */
#define MN_MAX_OBS 10       /* curriculum maximum, train_IQN_model.py:86-90 */
#define MN_NUM_BEAMS 11     /* robot.py:9 */
#define MN_OBS_DIM 26       /* 2 + 2 + 2*11, marinenav_env.py:80-81 */
#define MN_NUM_ACTIONS 9    /* robot.py:55-56 */
#define MN_MAX_STAGES 8     /* curriculum stages held on device */

typedef enum mn_status {
    MN_OK = 0,
    MN_ERR_INVALID = -1,   /* bad argument */
    MN_ERR_HIP = -2,       /* a HIP runtime call failed (text in mn_last_error) */
    MN_ERR_NO_DEVICE = -3, /* no gfx950 device visible */
    MN_ERR_ALLOC = -4,
    MN_ERR_PEER = -5       /* mn_xchg_export / mn_xchg_import: the runtime refused to export this rank's mailbox or to map a peer's (no IPC between the processes --
                              HSA_ENABLE_IPC_MODE_LEGACY=0 missing? -- or no peer access between the devices); text in mn_xchg_last_error */
} mn_status;

/* info codes written by mn_step; strings at marinenav_env.py:243,246,250,254,257 */
enum { MN_INFO_NORMAL = 0, MN_INFO_OUT_OF_BOUNDARY = 1, MN_INFO_TOO_LONG = 2, MN_INFO_COLLISION = 3, MN_INFO_REACH_GOAL = 4 };

/* arithmetic of the step/observation kernels */
enum {
    MN_PRECISION_F64 = 0,   /* everything float64: whole-episode parity with the reference */
    MN_PRECISION_MIXED = 1  /* f64 pose integration + f64->f32 relative geometry, f32 field/sonar:
                               single-step outputs within 1e-5 of the reference */
};

/* Scalar attributes of MarineNavEnv.__init__ (marinenav_env.py:40-73), Robot.__init__
 * (robot.py:25-50) and Sonar.__init__ (robot.py:5-12).  Uniform over the batch. */
typedef struct mn_params {
    double width, height;            /* :40-41 */
    double core_r;                   /* :42 self.r */
    double v_rel_max, p;             /* :43-44 */
    double v_range[2];               /* :45 */
    double obs_r_range[2];           /* :46 */
    double clear_r;                  /* :47 */
    double goal_dis;                 /* :54 */
    double timestep_penalty;         /* :55 */
    double collision_penalty;        /* :59 */
    double goal_reward;              /* :60 */
    double discount;                 /* :61 (host-side only) */
    double min_start_goal_dis;       /* :64 */
    double init_theta, init_speed;   /* :51-52, used when random_reset_state == 0 */
    double dt;                       /* robot.py:28 */
    double robot_r;                  /* robot.py:33 */
    double max_speed;                /* robot.py:34 */
    double a[3], w[3];               /* robot.py:35-36 */
    double sonar_range, sonar_angle; /* robot.py:7-8 */
    int32_t num_cores, num_obs;      /* :62-63, requested world size when no schedule is set */
    int32_t reset_start_and_goal;    /* :48 */
    int32_t random_reset_state;      /* :50 */
    int32_t set_boundary;            /* :73 */
    int32_t max_episode_steps;       /* 1000, :244 */
    int32_t N;                       /* robot.py:29 sub-steps per action */
    int32_t num_beams;               /* robot.py:9, must equal MN_NUM_BEAMS */
    int32_t precision;               /* MN_PRECISION_* */
    int32_t step_lanes;              /* lanes per env in the step kernel: 0 = auto (up to 128 K envs 2, or 4 in mn_step_append; else 1), or 1, 2, 4, 8.
                                        Results do not depend on it (fixed summation tree): a performance knob only */
    int32_t rollout_lanes;           /* the same for mn_rollout: 0 = auto (16 up to 4 096 envs, 8 up to 16 K, 4 up to 64 K, else 2), or 2, 4, 8, 16 */
} mn_params;

typedef struct mn_handle mn_handle;

/* Build flags of the loaded library.  The shipped libmarinenav_hip.so returns 0: its kernels have no switch that
 * removes work.  libmarinenav_hip_ablation.so (make ablation; profiling scripts only) returns MN_BUILD_ABLATION. */
#define MN_BUILD_ABLATION 1
int32_t mn_build_info(void);

/* Fills *p with the reference defaults (marinenav_env.py:40-73, robot.py:5-50). */
int mn_default_params(mn_params *p);

/* MarineNavEnv.__init__ for n_envs environments (marinenav_env.py:27-73).  RNG streams are
 * seeded with seed 0..n_envs-1 until mn_seed is called. */
int mn_create(int32_t n_envs, const mn_params *p, mn_handle **out);
int mn_destroy(mn_handle *h);
const char *mn_last_error(const mn_handle *h); /* h may be NULL for create-time errors */
int32_t mn_num_envs(const mn_handle *h);

/* Attribute writes such as `env.num_cores = 4`, `env.set_boundary = True`, `env.robot.N = 5`
 * (train_IQN_model.py:131-140, run_experiments.py:197-207). */
int mn_set_params(mn_handle *h, const mn_params *p);
int mn_get_params(const mn_handle *h, mn_params *p);

/* MarineNavEnv.seed (marinenav_env.py:75-78): env i <- np.random.RandomState(seeds_host[i])
 * (MT19937, legacy init_genrand seeding). */
int mn_seed(mn_handle *h, const uint32_t *seeds_host, void *stream);

/* Curriculum `schedule` constructor argument (marinenav_env.py:27,89-98;
 * train_IQN_model.py:86-90).  n_stages == 0 clears it.  The stage is looked up with
 * floor(total_timesteps[i] * timestep_scale): scale 1 reproduces the reference for one env,
 * scale n_envs makes the curriculum advance with aggregate experience. */
int mn_set_schedule(mn_handle *h, int32_t n_stages, const int64_t *timesteps, const int32_t *num_cores,
                    const int32_t *num_obstacles, const double *min_start_goal_dis, double timestep_scale);

/* env.start / env.goal attribute writes (train_IQN_model.py:133-134); env_idx < 0 = all envs. */
int mn_set_start_goal(mn_handle *h, int32_t env_idx, const double start[2], const double goal[2]);

/* MarineNavEnv.reset (marinenav_env.py:86-186) for the envs with mask_dev[i] != 0 (NULL = all):
 * curriculum lookup, start/goal, vortex cores, obstacles by bounded rejection sampling from the
 * env's own MT19937 stream (bit-exact draw order), robot pose, first observation.
 * Writes obs rows [i][26] (float32) of the reset envs into obs_dev (other rows untouched). */
int mn_reset(mn_handle *h, const uint8_t *mask_dev, float *obs_dev, void *stream);

/* MarineNavEnv.step (marinenav_env.py:199-262) for all envs: N sub-steps of
 * get_velocity (:422-465) + Robot.update_state (robot.py:102-123), get_observation (:273-326,
 * robot.py:125-198), reward and termination ladder (:220-257), counters (:259-260).
 * actions_dev[n] int32 in [0,9); obs_dev [n][26] f32 (terminal observation for finished envs);
 * reward_dev [n] f32; done_dev [n] u8; info_dev [n] u8 (MN_INFO_*).  No auto-reset. */
int mn_step(mn_handle *h, const int32_t *actions_dev, float *obs_dev, float *reward_dev, uint8_t *done_dev,
            uint8_t *info_dev, void *stream);

/* mn_step plus ReplayBuffer.add (thirdparty/IQN/replay_buffer.py:26-34, as called at agent.py:124) for every env, in
 * the SAME launch: the transition (prev_obs_dev[i] = the observation actions_dev[i] was chosen from, action, reward,
 * obs_dev[i] = terminal observation for finished envs, done) of env i goes to ring slot (ptr + i - first) mod capacity,
 * first = max(0, n_envs - capacity) (deque(maxlen): only the newest `capacity` rows survive; envs below `first` are
 * not stored).  Ring layout as mn_replay_append.  prev_obs_dev and obs_dev must be different buffers.  The caller
 * advances ptr by min(n_envs, capacity). */
int mn_step_append(mn_handle *h, const int32_t *actions_dev, const float *prev_obs_dev, float *obs_dev, float *reward_dev,
                   uint8_t *done_dev, uint8_t *info_dev, float *ring_states, float *ring_next_states,
                   int64_t *ring_actions, float *ring_rewards, float *ring_dones, int64_t ptr, int64_t capacity,
                   void *stream);

/* T = n_steps consecutive vector steps in ONE launch -- the loop `for t: a = policy(); env.step(a); if done: env.reset()`
 * (agent.py:113-170 without the learner) for every env, for policies that do not look at the observation: the uniform
 * random policy of BASELINE configs[1] (actions_dev == NULL: env i takes, at step t, the action
 * mn_random_actions(action_seed, first_step_index + t, first_env_index + i) -- a counter-based draw, uniform over the 9
 * actions) or a pre-drawn action tensor actions_dev [n_steps][n_envs] i32.  Environment state stays in registers
 * between steps and an env that finishes is reset at once by its own wavefront (same world generation, same RNG stream
 * positions as mn_reset_done), so the result is BIT-IDENTICAL to n_steps x (mn_step, mn_reset_done) with the same
 * actions.  Outputs (device pointers; every trace may be NULL):
 *   obs_dev          [n][26] f32       : what obs_dev holds after the last (mn_step, mn_reset_done) pair -- the observation
 *                                        each env continues from (first observation of the new episode where it just finished)
 *   obs_trace_dev    [n_steps][n][26]  : the observation each step returned (terminal observation for a finished env)
 *   reward_trace_dev [n_steps][n] f32, done_trace_dev / info_trace_dev [n_steps][n] u8, action_trace_dev [n_steps][n] i32
 * first_env_index = this handle's offset in a sharded run (rank * n_envs), so shards draw the actions of their slice.
 * Afterwards nothing is pending for mn_reset_done and mn_last_done_count reports 0.  (Envs a preceding mn_step flagged
 * done are NOT reset by this call: finish the mn_step / mn_reset_done pair before switching to mn_rollout.) */
int mn_rollout(mn_handle *h, int32_t n_steps, const int32_t *actions_dev, uint64_t action_seed, uint64_t first_step_index,
               uint64_t first_env_index, float *obs_dev, float *obs_trace_dev, float *reward_trace_dev,
               uint8_t *done_trace_dev, uint8_t *info_trace_dev, int32_t *action_trace_dev, void *stream);
/* EPISODES under an observation-reading policy in ONE launch (the classical baselines of run_experiments.py:100-190,
 * 213-282).  policy = MN_POLICY_APF (APF_agent.act, APF.py:17-78) or MN_POLICY_BA (BA_agent.act, BA.py:14-155), evaluated on the device on
 * the float32 observation row each step produces (tables a / w = the handle's robot parameters).  Every env runs its CURRENT episode --
 * starting from the observation in obs_dev -- for up to n_steps steps; an env that finishes is NOT reset: it idles, its traces read
 * reward 0 / done 1 / its terminal info code / action -1 from then on, obs_dev keeps its terminal observation.  Step for step
 * bit-identical to a loop of (mn_planner_act, mn_step).  Traces as mn_rollout ([n_steps][n] ...; any may be NULL). */
#define MN_POLICY_APF 1
#define MN_POLICY_BA 2
int mn_rollout_policy(mn_handle *h, int32_t n_steps, int32_t policy, float *obs_dev, float *obs_trace_dev, float *reward_trace_dev,
                      uint8_t *done_trace_dev, uint8_t *info_trace_dev, int32_t *action_trace_dev, void *stream);
/* One policy step of the same device functions for n observation rows [n][26] f32 -> actions [n] i32; a[3], w[3]: HOST arrays, the
 * robot's acceleration / angular-velocity tables (robot.py:35-36). */
int mn_planner_act(const float *obs_dev, int32_t n, int32_t policy, const double *a, const double *w, int32_t *actions_dev, void *stream);
/* The action draws of mn_rollout for one step: actions_dev[i] = action of env (first_env_index + i) at step step_index. */
int mn_random_actions(uint64_t action_seed, uint64_t step_index, uint64_t first_env_index, int32_t n, int32_t *actions_dev,
                      void *stream);

/* The caller-side `if done: state = train_env.reset()` (thirdparty/IQN/agent.py:152-170), batched:
 * resets exactly the envs the LAST mn_step flagged done and overwrites their rows of obs_dev
 * (which may or may not be the buffer given to mn_step). */
int mn_reset_done(mn_handle *h, float *obs_dev, void *stream);
/* The same reset OFF the caller's critical path: launched on a stream the handle owns (behind an event recorded on `stream`), so it runs
 * under whatever the caller enqueues on `stream` next.  In the training loop that is the act kernel of the next vector step
 * (agent.py:113-124: `state = next_state` / `env.reset()` then `act(state)`), which depends on the reset only through the rows of the
 * finished envs: tell it which they are and how to recognise a finished row -- mn_iqn_set_late_rows(ctx, <the step's done_dev>, *ready_out,
 * *tick_out, n) -- and it takes those rows last, each after `(*ready_out)[e] == *tick_out` (written by the reset wavefront behind its
 * write-through row).  Results are those of mn_reset_done, bit for bit.  `stream` is joined again (everything enqueued later waits for the
 * reset launch to end) by mn_reset_join or by the next mn_step / mn_step_append / mn_reset_done[_async] on it; every other entry point of
 * the handle waits for it on the host.  Between this call and the join the caller must not read obs_dev rows of finished envs except
 * through such an act launch.
 * Beside the act kernel's workgroups a CU has room for FOUR reset wavefronts (one per SIMD: 96 registers each; the kernel that runs there reads the env's
 * MT19937 row in place instead of copying it into LDS), which run several times slower there: that hides the resets of a few thousand episodes per vector
 * step (alone, such a launch is a latency chain that leaves most of the chip idle; measured cross-over 6 000 - 7 800 steadily arriving per 65 536-env step, depending on the box),
 * not a burst of tens of thousands.  The call therefore launches under the act kernel only while the decaying peak of the episodes started per reset
 * launch -- peak <- max(count, 7/8 peak), kept by the launches themselves and read by the host from a mapped word without synchronising -- is at most
 * `under_act_max` (mn_set_reset_under_act_max: default MN_RESET_UNDER_ACT_MAX_DEFAULT; 0x7fffffff always, -1 never); otherwise it is mn_reset_done on
 * `stream` and *ready_out is NULL (no late rows).  mn_set_reset_under_act_max also reports that peak as of the last launch seen (-1: none yet).
 * mn_debug_side_delay_us (test hook): a kernel that sleeps `us` microseconds in front of every such launch on the handle's stream, i.e. a reset
 * launch that does not run beside the act kernel -- what the callers' fallback (late-row timeouts -> resets in front) is tested with. */
#define MN_RESET_UNDER_ACT_MAX_DEFAULT 5000
int mn_reset_done_async(mn_handle *h, float *obs_dev, void *stream, const uint32_t **ready_out, uint32_t *tick_out);
int mn_reset_join(mn_handle *h, void *stream);
int mn_set_reset_under_act_max(mn_handle *h, int32_t under_act_max, int64_t *last_seen);
int mn_debug_side_delay_us(mn_handle *h, int32_t us);

/* MarineNavEnv.reset_with_eval_config (marinenav_env.py:467-555), world + pose fields, for `count`
 * consecutive envs starting at first_env.  Host arrays, row-major:
 *   n_cores[count], cores_xy[count][MN_MAX_CORES][2], clockwise[count][MN_MAX_CORES],
 *   gamma[count][MN_MAX_CORES], n_obs[count], obs_xy[count][MN_MAX_OBS][2], obs_r[count][MN_MAX_OBS],
 *   start[count][2], goal[count][2], init_theta[count], init_speed[count].
 * Does not touch the RNG streams.  Resets episode_timesteps, places the robot at `start` and
 * writes the first observation rows into obs_dev if it is not NULL. */
int mn_load_worlds(mn_handle *h, int32_t first_env, int32_t count, const int32_t *n_cores, const double *cores_xy,
                   const int32_t *clockwise, const double *gamma, const int32_t *n_obs, const double *obs_xy,
                   const double *obs_r, const double *start, const double *goal, const double *init_theta,
                   const double *init_speed, float *obs_dev, void *stream);

/* MarineNavEnv.episode_data (marinenav_env.py:557-622), world + pose fields; same layouts. */
int mn_get_worlds(mn_handle *h, int32_t first_env, int32_t count, int32_t *n_cores, double *cores_xy,
                  int32_t *clockwise, double *gamma, int32_t *n_obs, double *obs_xy, double *obs_r, double *start,
                  double *goal, double *init_theta, double *init_speed);

/* Robot pose and counters (robot.py:40-44, marinenav_env.py:70-71).  state[count][6] =
 * x, y, theta, speed, velocity_x, velocity_y (float64).  NULL pointers are skipped. */
int mn_get_state(mn_handle *h, int32_t first_env, int32_t count, double *state, int32_t *episode_timesteps,
                 int64_t *total_timesteps);
int mn_set_state(mn_handle *h, int32_t first_env, int32_t count, const double *state,
                 const int32_t *episode_timesteps, const int64_t *total_timesteps);

/* Float64 copy of the last observation each env produced (the reference returns float64,
 * marinenav_env.py:326).  Only kept when precision == MN_PRECISION_F64; out[count][26]. */
/* Float64 copies of the last observation rows / rewards (what the reference's float64 step returns before the agent's .float()),
 * for parity checks and the n = 1 gym-shaped facade.  OFF by default since round 4 -- the training loop does not read them and they
 * were 14 MB of the step kernel's 39 MB of writes per 65 536-env launch: mn_enable_obs64(h, 1) (MN_PRECISION_F64 handles only) makes
 * every later mn_step / mn_step_append / mn_reset* / mn_rollout write them; (h, 0) stops again.  mn_get_obs64 / mn_get_reward64 return
 * MN_ERR_INVALID while disabled.  Enabling does not change any other output (bit-identity tests). */
int mn_enable_obs64(mn_handle *h, int32_t on);
int mn_get_obs64(mn_handle *h, int32_t first_env, int32_t count, double *out);
/* Float64 copy of the last reward (marinenav_env.py:220-255 computes it in float64); same rule. */
int mn_get_reward64(mn_handle *h, int32_t first_env, int32_t count, double *out);

/* robot.trajectory (marinenav_env.py:211-212: one [x, y] per kinematic SUB-step): after mn_enable_trajectory(h, max_N)
 * every mn_step also records the N sub-step positions of each env; mn_get_trajectory copies out[count][n_substeps][2]
 * of the LAST step.  MN_PRECISION_F64 handles only (the facade / evaluation path); off by default. */
int mn_enable_trajectory(mn_handle *h, int32_t max_substeps);
int mn_get_trajectory(mn_handle *h, int32_t first_env, int32_t count, int32_t n_substeps, double *out);

/* Next double each env's RandomState would return, without consuming it (test hook pinning the
 * RNG stream position; cf. np.random.RandomState.random_sample). */
int mn_peek_next_double(mn_handle *h, int32_t first_env, int32_t count, double *out);

/* Number of envs the last mn_step flagged done (synchronises the stream). */
int mn_last_done_count(mn_handle *h, void *stream, int32_t *out);

/* Timing hook for benchmarks: records hipEvents on `stream` around the step kernel of the next max_launches mn_step / mn_step_append /
 * mn_rollout calls and around the reset kernel of the next max_launches mn_reset_done calls.  mn_profile_end returns the mean duration of
 * the recorded step launches and closes the window; mn_profile_reset_end (call it first) that of the recorded mn_reset_done launches. */
int mn_profile_begin(mn_handle *h, int32_t max_launches);
int mn_profile_reset_end(mn_handle *h, void *stream, double *mean_ms, int32_t *launches);
int mn_profile_end(mn_handle *h, void *stream, double *mean_ms, int32_t *launches);

/* ---- IQN inference ---------------------------------------------------------------------------
 * Context of one acting agent (the counterpart of holding an `IQNAgent`, thirdparty/IQN/agent.py:10-84): owns the
 * permuted copy of the network weights the act kernel stages into LDS, and the profiling events.  The copy is CACHED:
 * it is rebuilt by the first act call after mn_iqn_create and by the first act call after mn_iqn_weights_changed, which
 * the caller invokes whenever the weights behind the `weights` pointers were written (an optimizer step, a checkpoint
 * load, soft_update into this network).  Calls that share a context must be stream-ordered; different contexts are
 * independent (two agents may act concurrently on two streams of one device).  A context is bound to the HIP device
 * that was current at mn_iqn_create; act calls made while another device is current return MN_ERR_INVALID. */
typedef struct mn_iqn_ctx mn_iqn_ctx;
int mn_iqn_create(mn_iqn_ctx **out);
int mn_iqn_destroy(mn_iqn_ctx *c);
int mn_iqn_weights_changed(mn_iqn_ctx *c);
/* Which acting kernel serves the context's calls (with or without quantile output).
 *   2 (default): the split-f16 kernel -- every float32 operand is split into two f16 pieces (hi = RNE16(x), lo = RNE16(x - hi))
 *      and a product is accumulated as lo.hi + hi.lo + hi.hi on v_mfma_f32_16x16x32_f16 with power-of-two range scaling chosen
 *      from a guaranteed bound, so the result has the error class of float32 arithmetic (measured against a float64
 *      evaluation it is as close as the exact kernel and as eager PyTorch float32) at ~1/3 of the time;
 *   0: the exact-f32 v_mfma_f32_16x16x4_f32 kernel.
 * Same network in both; they differ by float32 rounding only.  Anything else is MN_ERR_INVALID (1 and 3 were the 32x32 re-layouts of the two kernels,
 * measured 2.6 % / 4 % slower on MI355X in rounds 2 / 4 and removed in round 6). */
int mn_iqn_set_variant(mn_iqn_ctx *c, int32_t variant);
/* How an act launch's quantile fractions are drawn (round 4).
 *   0 (default): every observation row gets its own 32 taus -- what a batch of independent calls of the reference's batch-1
 *      IQNAgent.act (agent.py:186-205 -> model.py:149-153) would draw.  mn_iqn_act takes taus [n][32]; mn_iqn_act_rng writes
 *      draws[0 .. 32 n) = taus, draws[32 n .. 33 n) = exploration uniforms.
 *   1: ONE set of 32 taus (x the launch's cvar) for all n rows of the launch.  Every row still sees 32 i.i.d. U(0,1) cvar fractions per
 *      call (the reference's act is batch-1, so independence ACROSS environments is not a reference property), but layer 1 of the
 *      network -- relu(W1 cos(pi k tau) + b1), model.py:141-157,176-178 -- becomes a [32 x 208] constant of the launch that the
 *      preparation launch computes once (exact float32): 216 instead of 372 matrix instructions per row, no per-row cosines.
 *      mn_iqn_act then reads taus [32]; mn_iqn_act_rng writes draws[0 .. 32) = the taus, draws[32 .. 32 + n) = exploration uniforms.
 *      Only with variant 2, without per-row cvar (cvar_row_dev == NULL) and without a selected image slot: else MN_ERR_INVALID.
 *      Two kernel forms serve it: up to 65 535 rows (and for quantile output) one wavefront per row with the taus in the MFMA columns;
 *      from 65 536 rows (every CU of the chip gets a 256-row workgroup) the ENVIRONMENTS are the columns -- T[tau] = W2 diag(h1[tau]) is built once per launch, a wavefront splits the
 *      features of 32 rows once and streams T through LDS: ~260 instead of ~730 vector instructions per row.
 *   2: as 1, but always the wavefront-per-row form; 3: as 1, but always the environment-tiled form (A / B measurements, tests).
 * A row's result for GIVEN taus is the same function in both modes up to float32 rounding (tests). */
int mn_iqn_set_tau_mode(mn_iqn_ctx *c, int32_t mode);
/* Late rows of the NEXT act launch of this context (mn_reset_done_async): mask_dev [n] u8 != 0 marks the rows whose observation another
 * stream is still writing, flags_dev [n] / tick say when a row is final (flags_dev[e] == tick; the row itself written through at agent
 * scope before the word).  The launch takes those rows last and reads them past the caches; results equal those of a launch behind the
 * reset.  Returns MN_OK if the next launch of n rows will honour it (launch it next on this context), 1 if the context's current form
 * cannot (the exact-f32 variant, launch-shared taus, quantile capture, more than 64 rows per wavefront): the caller joins the reset
 * (mn_reset_join) instead.  NULL, NULL clears.  mn_iqn_late_timeouts: waits that ran out (0.5 s bound; mn_iqn_set_late_bound_ms, in (0, 60 000] ms,
 * for launches armed afterwards) since the context was made -- anything but 0 means an action was computed on an unfinished row, i.e. the reset
 * launch did not run beside the act kernel on this box: the caller goes back to mn_reset_done (synchronises `stream`).  mn_iqn_late_timeouts_peek:
 * the same count as far as the launches executed so far have reported it through a host-mapped word -- no synchronisation, for a look every few
 * vector steps. */
int mn_iqn_set_late_rows(mn_iqn_ctx *c, const uint8_t *mask_dev, const uint32_t *flags_dev, uint32_t tick, int32_t n);
int mn_iqn_late_timeouts(mn_iqn_ctx *c, void *stream, uint32_t *out);
int mn_iqn_late_timeouts_peek(mn_iqn_ctx *c, uint32_t *out);
int mn_iqn_set_late_bound_ms(mn_iqn_ctx *c, double ms);
/* Measurement aid (bench.py: `gpu_clock_probe`): runs a pure stream of the act kernel's matrix instruction (v_mfma_f32_16x16x32_f16, two
 * waves per SIMD on every CU) for about target_ms milliseconds on `stream` and returns out[0] = elapsed ms (HIP events), out[1] = the clock
 * in GHz the matrix pipe sustained (16 cycles per instruction), out[2] = the clock by the waves' own counters (s_memtime ticks per
 * s_memrealtime tick x 100 MHz), out[3] = sustained f16 TFLOP/s of the chip, out[4] = CUs.  Synchronises the stream.  The boxes of a pool
 * differ in the clock they hold under matrix load; this tells a slow box from a slow kernel. */
int mn_probe_mfma_clock(double target_ms, double *out, void *stream);
/* Grid of the act kernel.  0 (default): at most one PERSISTENT workgroup per CU, each looping over its share of the observations --
 * the weight image is staged into LDS once per CU, the fastest form when nothing else runs.  max_workgroups > 0: up to that many
 * workgroups (more than CUs = several rounds of shorter workgroups, e.g. 2048 for 65 536 observations = 4 per wavefront, ~3 % more
 * time in isolation): CUs are released every few tens of microseconds, so kernels of OTHER streams (the learner's gradient steps,
 * the env kernels of another batch half) are dispatched in between instead of waiting for the whole act launch.  Results do not
 * depend on it. */
int mn_iqn_set_grid(mn_iqn_ctx *c, int32_t max_workgroups);
/* Rebuilds the cached weight image of the selected acting kernel now, on `stream`, if it is stale (what the first act call after
 * mn_iqn_weights_changed would do).  For callers that issue act calls of ONE context on several streams: refresh on one stream,
 * make the others wait for it, and no act call has to write the image. */
int mn_iqn_refresh(mn_iqn_ctx *c, const float *const *weights, void *stream);

/* Fused IQNAgent.act (thirdparty/IQN/agent.py:186-205) for n observations: ObsEncoder.forward with
 * K = 32 quantile samples + mean over them (model.py:141-191), then argmax and the epsilon-greedy choice.
 *   obs_dev   [n][26] f32 : observations (row-major, as mn_step writes them)
 *   taus_dev  [n][32] f32 : quantile fractions, already multiplied by cvar (model.py:149-153)
 *   weights   [14]        : HOST array of DEVICE pointers, nn.Linear layout [out][in], in state-dict order:
 *                           velocity_encoder.weight [16][2], .bias [16], goal_encoder.weight [16][2], .bias [16],
 *                           sensor_encoder.weight [176][22], .bias [176], cos_embedding.weight [208][64], .bias [208],
 *                           hidden_layer.weight [64][208], .bias [64], hidden_layer_2.weight [64][64], .bias [64],
 *                           output_layer.weight [9][64], .bias [9].  Read only when the cached image is stale.
 *   qvals_dev [n][9] f32  : mean over taus of the quantile values (may be NULL)
 *   explore_u_dev [n] f32 : uniform [0,1) draws for exploration (may be NULL = greedy); env i takes the
 *                           greedy action iff u_i > eps (agent.py:200), else action floor(u_i / eps * 9)
 *   actions_dev [n] i32   : chosen actions (may be NULL if only Q-values are wanted)
 *   quantiles_dev [n][32][9] f32 : IQNAgent.act_eval's `quantiles` (agent.py:217-236; model.py:185 before the mean), or
 *                           NULL.  When given, the output layer runs per tau and Q is the mean of these values.
 * Exact float32 MFMA (v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32).  num_taus must be 32. */
int mn_iqn_act(mn_iqn_ctx *c, const float *obs_dev, const float *taus_dev, const float *const *weights, float *qvals_dev,
               const float *explore_u_dev, float eps, int32_t *actions_dev, float *quantiles_dev, int32_t n,
               int32_t num_taus, void *stream);

/* Same act kernel, with the random numbers of the call drawn by the library (no separate generator kernels):
 * draws_dev [33 n] f32 (caller-owned scratch) receives tau[e][j] = U[0,1) * cvar (cvar_row_dev[e] if given, else the
 * scalar `cvar`; model.py:149-153) in its first 32 n entries -- act_eval's `taus` -- and the exploration uniforms of
 * agent.py:199 in the last n; the act kernel then consumes them.  Counter-based generator keyed by
 * rng_state_dev = u64[2] {seed, call counter} on the device; the counter is advanced by the call. */
int mn_iqn_act_rng(mn_iqn_ctx *c, const float *obs_dev, const float *const *weights, uint64_t *rng_state_dev,
                   float *draws_dev, const float *cvar_row_dev, float cvar, float eps, int32_t *actions_dev,
                   float *qvals_dev, float *quantiles_dev, int32_t n, int32_t num_taus, void *stream);

/* ---- replay ring ------------------------------------------------------------------------------
 * ReplayBuffer.add (thirdparty/IQN/replay_buffer.py:26-34) for n transitions in one launch: batch row i
 * goes to ring slot (ptr + i) mod capacity (FIFO eviction like deque(maxlen); if n > capacity only the
 * newest `capacity` rows are written, starting at ptr).  Batch: obs / next_obs [n][26] f32, actions [n]
 * i32, reward [n] f32, done [n] u8.  Ring (device, the layout ReplayBuffer.sample returns): states /
 * next_states [cap][26] f32, actions [cap] i64, rewards / dones [cap] f32.  The caller advances ptr. */
int mn_replay_append(const float *obs_dev, const int32_t *actions_dev, const float *reward_dev, const float *next_obs_dev,
                     const uint8_t *done_dev, float *ring_states, float *ring_next_states, int64_t *ring_actions,
                     float *ring_rewards, float *ring_dones, int64_t n, int64_t ptr, int64_t capacity, void *stream);

/* ---- IQN gradient step ------------------------------------------------------------------------
 * One optimizer step of IQNAgent.train (thirdparty/IQN/agent.py:269-304) on `batch` transitions gathered from the
 * replay ring (layout above) at rows idx_dev[batch] (i64): target-network forward on next_states, local-network
 * forward on states, quantile-Huber TD loss (kappa = 1, 8 x 8 tau pairs; agent.py:279-295, 401-407), backward.
 * All pointers are device pointers, float32 unless noted, 16-byte aligned.
 *   taus_*_dev [batch][8]     : the uniform(0,1) tau draws of model.py:149 for the target / local forward
 *   params_local/_target      : FLAT parameter vectors, 35 785 floats in ObsEncoder.named_parameters() order
 *                               (model.py:120-136): velocity_encoder.{weight[16][2],bias[16]}, goal_encoder.{[16][2],[16]},
 *                               sensor_encoder.{[176][22],[176]}, cos_embedding.{[208][64],[208]},
 *                               hidden_layer.{[64][208],[64]}, hidden_layer_2.{[64][64],[64]}, output_layer.{[9][64],[9]}
 *   workspace                 : mn_iqn_train_workspace_floats(batch) floats (per-workgroup partial gradients, norm partials, the
 *                               TD-target hand-off granules with their epoch word, the Adam ticket).  Call
 *                               mn_iqn_train_workspace_init(workspace, batch, stream) ONCE before the first step and then pass the same
 *                               buffer and batch, untouched, to every call of one learner: the workspace carries state between calls
 *                               (hand-off epoch, tickets, the staged next batch).  A workspace that was never initialised is refused
 *                               on the device: the step returns a NaN loss and leaves gradient, moments and parameters untouched
 *   grad_out [35 785]         : d loss / d params_local (un-clipped); loss_out [1]: the loss
 * mn_iqn_train_grad computes loss and gradient (2 kernels, deterministic: no float atomics).  The forward / backward launch has
 * two workgroup roles -- batch / 2 TARGET workgroups (lower block indices) run the target network and hand their 16 TD targets
 * each to the LOCAL workgroup of the same two batch elements inside the launch -- so it wants batch <= 256 (one workgroup per CU);
 * larger batches, and mn_iqn_train_set_mode(1), let every local workgroup compute its own targets.  The caller may average
 * grad_out over ranks (RCCL all-reduce(SUM); pass grad_scale = 1 / world_size and grad_rewritten = 1) before mn_iqn_train_adam,
 * which applies clip_grad_norm_(max_norm) (agent.py:299) to grad_scale * grad and one torch.optim.Adam update (agent.py:300;
 * exp_avg / exp_avg_sq [35 785], step_dev: i32 step counter on the device, incremented by the call; grad is overwritten with the
 * clipped gradient).  grad_rewritten = 0 promises that grad_out is exactly what the preceding mn_iqn_train_grad* call on the same
 * workspace left there (the norm then comes from partial sums that call stored); grad_rewritten = 2 that mn_iqn_train_exchange formed
 * them for grad_scale * grad; with grad_rewritten = 1 or (grad_rewritten = 0 and grad_scale != 1) the
 * norm is recomputed from grad.
 * batch must be even and <= 1024, num_taus must be 8.  Exact float32 (v_mfma_f32_16x16x4_f32). */
/* The whole gradient step of a single learner -- IQNAgent.train (agent.py:269-304) incl. clip_grad_norm_ and optimizer.step() -- as TWO
 * launches: forward / backward, then a launch in which every block reduces the partial gradients of its own 256 parameters, exchanges the norm
 * partials with the other blocks as self-tagged granules and applies clip + Adam (round 4) -- or, with MN_TRAIN_ONE_LAUNCH in `flags`, as ONE launch (the FUSED
 * step; batch a multiple of 16 and every workgroup a CU of its own: batch <= 256 on an MI355X; else two launches): the target workgroups of the forward /
 * backward launch are also its reduction + Adam blocks.  XCD-grouped: the partial-gradient rows of the workgroups that share an XCD (block index mod 8: the
 * dispatcher deals workgroups out round-robin) are summed inside that XCD's L2 and only the eight group rows cross to the other XCDs, as self-tagged granules the
 * reduction blocks poll.  All forms are bit-identical.  rng_state_dev != NULL: the batch is drawn in the
 * launch (arguments as mn_iqn_train_grad_sampled; idx_dev / taus_*_dev ignored); NULL: the given batch (as mn_iqn_train_grad).  params_local is
 * updated in place, grad_out receives the clipped gradient.  Bit-identical to mn_iqn_train_grad* + mn_iqn_train_adam(grad_scale = 1): those stay
 * for callers that put something between the two (the shared learner's all-reduce / exchange). */
int mn_iqn_train_step(const float *ring_states, const float *ring_next_states, const int64_t *ring_actions, const float *ring_rewards,
                      const float *ring_dones, int64_t ring_size, uint64_t *rng_state_dev, const int64_t *idx_dev, const float *taus_target_dev,
                      const float *taus_local_dev, int64_t *idx_out, float *taus_out, float *params_local, const float *params_target,
                      float *workspace, float *grad_out, float *loss_out, float *exp_avg, float *exp_avg_sq, int32_t *step_dev, int32_t batch,
                      int32_t num_taus, float gamma, int32_t flags, double lr, double beta1, double beta2, double eps, double max_norm, void *stream);
int64_t mn_iqn_train_workspace_floats(int32_t batch);
/* Diagnostic: float index inside the workspace of a u32 counter -- local workgroups of one-launch steps that did not run on the XCD of the first
 * workgroup of their group (block index % 8) since mn_iqn_train_workspace_init (their partial-gradient rows go through memory instead of staying in the
 * XCD's L2: correct, slower; the dispatcher deals workgroups out to the XCDs round-robin, so 0 is expected). */
int64_t mn_iqn_train_workspace_misplaced_word(int32_t batch);
/* Float index inside the workspace of the u32 STATUS word: reduction + Adam blocks of which a bounded wait ran out since mn_iqn_train_workspace_init -- a hand-off
 * inside a fused launch, or (shared learner) a peer's granules.  Such a block leaves moments and parameters untouched and writes NaN into its piece of the
 * gradient.  0 in a healthy run; the caller reads the word where it synchronises anyway (evaluation points, the end of a run) and treats non-zero as an error. */
int64_t mn_iqn_train_workspace_status_word(int32_t batch);
int mn_iqn_train_workspace_init(float *workspace, int32_t batch, void *stream);
/* The fused forms wait, inside a launch, for other workgroups of the same launch, which is only safe while those are resident together: the launch plan is made
 * from the device's CU count and the kernels' occupancy (round 5), and a device that cannot hold them takes three launches (four for a shared learner:
 * reduction, gather, Adam) -- bit-identical.  mn_iqn_train_plan: launches one gradient step takes on the current device for this batch and these flags
 * (exchange != 0: mn_iqn_train_step_xchg); < 0: -error.  mn_iqn_train_set_cu_limit: plan as if the device had at most n_cu CUs (0: what it reports) -- tests, and
 * ranks that share one GPU (each plans for its share). */
int mn_iqn_train_plan(int32_t batch, int32_t flags, int32_t exchange);
int mn_iqn_train_set_cu_limit(int32_t n_cu);

/* ---- One-shot gradient exchange of a shared learner (BASELINE configs[4]; SURVEY 8e: one flat 143 KB bucket per gradient step, latency-
 * bound).  Alternative to an RCCL all-reduce between mn_iqn_train_grad* and mn_iqn_train_adam: every rank owns a MAILBOX in device memory;
 * once a workspace is attached, the gradient step's reduction kernel also publishes the reduced gradient there as self-tagged 8-byte
 * granules {step tag, value} (system-scope stores); mn_iqn_train_exchange launches ONE kernel that reads all ranks' mailboxes -- the peers'
 * through IPC-mapped pointers, i.e. directly over xGMI -- polling each granule until it carries the current step's tag, and leaves
 *     grad = sum over ranks, in rank order (bit-identical on every rank; equal to an all-reduce(SUM) for two ranks)
 * plus the norm partials of grad_scale * grad, so that the step continues with mn_iqn_train_adam(..., grad_scale, grad_rewritten = 2).
 * Four launches per step instead of five (no collective launch, no separate norm pass), no host synchronisation, graph-capturable -- the form a device too
 * small for the fused launches falls back to; mn_iqn_train_step_xchg (below) is the exchange INSIDE the gradient step's own launch(es).
 *   mn_xchg_create(rank, world <= 8)    this rank's context + mailbox on the current device
 *   mn_xchg_export(x, handle[64])       hipIpcMemHandle_t of the mailbox, to be sent to every peer (e.g. torch.distributed.all_gather_object)
 *   mn_xchg_import(x, peer, handle[64]) maps a peer's mailbox (once per peer)
 *   mn_xchg_attach(x, workspace, batch) the learner that steps on `workspace` publishes into x's mailbox from now on (x = NULL detaches)
 *   mn_iqn_train_exchange(x, grad, workspace, batch, grad_scale, stream)   after mn_iqn_train_grad* on the same stream
 *   mn_xchg_status(x, &timeouts)        gathers that did not get a peer's granules within the bound (0 in a healthy run; the poll is bounded so that a
 *                                       missing peer can never hang the device; the step that timed out updates nothing and raises the workspace's
 *                                       status word too).  mn_xchg_set_timeout_ms(x, ms): the bound; default 30 s with peers (one of them may be
 *                                       evaluating or writing a checkpoint meanwhile), 2 s alone
 *   mn_xchg_memory_kind(x)              how the mailbox was allocated: 2 = uncached device memory (hipDeviceMallocUncached), 1 = fine-grained, 0 = plain
 *                                       hipMalloc (coarse-grained: cross-device visibility inside a running kernel is then not promised by the memory model)
 * All ranks must call the step functions the same number of times (the tag is the workspace's step count).  RCCL stays the default
 * transport of iqn/fused_train.py; this path is opt-in (IQNAgent.exchange = "mailbox"). */
/* ---- DQN baseline, acting: the greedy policy of the reference's sb3 `ObsEncoderPolicy` for n observation rows in ONE
 * launch -- encoders (no activation) -> hidden_layer -> hidden_layer_2 -> output_layer (thirdparty/stable_baselines3/common/
 * torch_layers.py:96-135) -> q_net.0 -> q_net.2 -> q_net.4 (dqn/policies.py:48-58, net_arch [64, 64]) -> argmax (:69-73), exact float32 MFMA.
 *   weights[18]   device pointers, nn.Linear layout: {velocity, goal, sensor}_encoder, hidden_layer, hidden_layer_2, output_layer,
 *                 q_net.0, q_net.2, q_net.4, each as (weight, bias)
 *   image_dev     caller-owned scratch of mn_dqn_image_floats() floats: the permuted weight image; repack != 0 rebuilds it from
 *                 `weights` in front of the launch (first call, and after the weights changed)
 *   qvals_dev     [n][9] Q-values or NULL; actions_dev [n] greedy actions (first maximum) or NULL */
int64_t mn_dqn_image_floats(void);
int mn_dqn_act(const float *obs_dev, const float *const *weights, float *image_dev, int32_t repack, float *qvals_dev, int32_t *actions_dev,
               int32_t n, void *stream);

typedef struct mn_xchg mn_xchg;
int mn_xchg_create(int32_t rank, int32_t world, mn_xchg **out);
int mn_xchg_export(mn_xchg *x, void *handle_out);
int mn_xchg_import(mn_xchg *x, int32_t peer_rank, const void *handle);
int mn_xchg_attach(mn_xchg *x, float *workspace, int32_t batch, void *stream);
int mn_iqn_train_exchange(mn_xchg *x, float *grad, float *workspace, int32_t batch, float grad_scale, void *stream);
int mn_xchg_memory_kind(mn_xchg *x);
int mn_xchg_set_timeout_ms(mn_xchg *x, int64_t ms);
/* mn_iqn_train_step for a SHARED learner: the exchange happens INSIDE the reduction + clip + Adam role (every Adam block publishes its 64
 * reduced columns into this rank's mailbox, gathers the same columns of every rank in rank order and continues with grad_scale x the sum).  As many
 * launches per gradient step as a single learner's -- ONE with MN_TRAIN_ONE_LAUNCH (round 5), two without; bit-identical to the four-launch sequence
 * above, which is also what a device too small for the fused launches takes.  Arguments as mn_iqn_train_step. */
int mn_iqn_train_step_xchg(mn_xchg *x, const float *ring_states, const float *ring_next_states, const int64_t *ring_actions,
                           const float *ring_rewards, const float *ring_dones, int64_t ring_size, uint64_t *rng_state_dev, const int64_t *idx_dev,
                           const float *taus_target_dev, const float *taus_local_dev, int64_t *idx_out, float *taus_out, float *params_local,
                           const float *params_target, float *workspace, float *grad_out, float *loss_out, float *exp_avg, float *exp_avg_sq,
                           int32_t *step_dev, int32_t batch, int32_t num_taus, float gamma, int32_t flags, double lr, double beta1, double beta2,
                           double eps, double max_norm, float grad_scale, void *stream);
int mn_xchg_status(mn_xchg *x, int32_t *timeouts);
/* Text of the last failure of an mn_xchg_* call on this exchange ("" if none; valid until the next call on it). */
const char *mn_xchg_last_error(const mn_xchg *x);
int mn_xchg_destroy(mn_xchg *x);
/* ReplayBuffer.sample (replay_buffer.py:42-47, random.sample: `batch` DISTINCT uniform rows of [0, ring_size)) -> idx_out
 * [batch] i64, plus n_taus_total uniform [0,1) floats -> taus_out (the step's tau draws, model.py:149; may be 0).  Slot k reads
 * row perm(k) of a keyed pseudo-random permutation of [0, ring_size) (4-round Feistel network + cycle walking): distinct by
 * construction, O(1) per slot.  rng_state_dev: u64[2] = {seed, call counter} on the device; the counter is advanced by the call.
 * batch <= 1024, ring_size >= batch. */
int mn_iqn_sample(int64_t ring_size, int32_t batch, uint64_t *rng_state_dev, int64_t *idx_out, float *taus_out,
                  int32_t n_taus_total, void *stream);
int mn_iqn_train_grad(const float *ring_states, const float *ring_next_states, const int64_t *ring_actions,
                      const float *ring_rewards, const float *ring_dones, const int64_t *idx_dev,
                      const float *taus_target_dev, const float *taus_local_dev, const float *params_local,
                      const float *params_target, float *workspace, float *grad_out, float *loss_out, int32_t batch,
                      int32_t num_taus, float gamma, void *stream);
/* The same gradient step with the batch drawn inside the launch: every workgroup of the forward / backward kernel evaluates
 * mn_iqn_sample's permutation for its own two slots (and its own taus) from {seed, call counter}, so there is no sampling launch.
 * Bit-identical to mn_iqn_sample followed by mn_iqn_train_grad from the same state; the call counter is advanced once (by the
 * reduction kernel).  idx_out [batch] i64 and taus_out [2][batch][8] receive the batch (NULL: not written).  Needs
 * batch <= ring_size < 2^31. */
/* flags: MN_TRAIN_STAGE_NEXT -- the reduction kernel of this step also draws the NEXT step's batch (call counter + 1) from the ring
 * as it is now and stages it (rows, transitions, taus) in the workspace; MN_TRAIN_USE_STAGED -- start from the batch a previous call
 * staged instead of drawing and gathering it (one memory round trip at the head of the launch instead of three dependent ones).
 * The staged batch is used only if it was drawn for this call counter and this ring_size (checked on the device; otherwise the
 * launch draws and gathers as usual), and its transitions are those the ring held when it was staged: pass MN_TRAIN_USE_STAGED only
 * if the ring was not written since the staging call (then the step is bit-identical to the unstaged one). */
#define MN_TRAIN_USE_STAGED 1
#define MN_TRAIN_STAGE_NEXT 2
#define MN_TRAIN_ONE_LAUNCH 4      /* mn_iqn_train_step[_xchg]: the fused step -- reduction + clip + Adam inside the forward / backward launch */
#define MN_TRAIN_TEST_MISPLACE(k) ((k) << 4)   /* test hook, k = 1..3, with MN_TRAIN_ONE_LAUNCH: treat some workgroups as if they had landed on another XCD */
int mn_iqn_train_grad_sampled(const float *ring_states, const float *ring_next_states, const int64_t *ring_actions,
                              const float *ring_rewards, const float *ring_dones, int64_t ring_size, uint64_t *rng_state_dev,
                              int64_t *idx_out, float *taus_out, const float *params_local, const float *params_target,
                              float *workspace, float *grad_out, float *loss_out, int32_t batch, int32_t num_taus, float gamma,
                              int32_t flags, void *stream);
int mn_iqn_train_adam(float *params, float *grad, float *exp_avg, float *exp_avg_sq, int32_t *step_dev, float *workspace,
                      int32_t batch, double lr, double beta1, double beta2, double eps, double max_norm, float grad_scale,
                      int32_t grad_rewritten, void *stream);
/* Process-wide switch of the forward / backward launch: 0 (default) = two workgroup roles with the in-launch TD-target hand-off,
 * 1 = every local workgroup runs the target forward itself (no inter-workgroup communication; what batches > 256 always use).
 * Same arithmetic either way: results are bit-identical. */
int mn_iqn_train_set_mode(int32_t mode);

/* Benchmark hook: HIP events on the launch stream around the next act launches of this context (weight / random-number
 * preparation launch included). */
int mn_iqn_profile_begin(mn_iqn_ctx *c, int32_t max_launches);
int mn_iqn_profile_end(mn_iqn_ctx *c, void *stream, double *mean_ms, int32_t *launches);

#ifdef __cplusplus
}
#endif
#endif /* MARINENAV_HIP_H */
