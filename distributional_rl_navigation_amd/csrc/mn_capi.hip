// mn_capi.hip -- C-ABI host side of libmarinenav_hip.so (see include/marinenav_hip.h).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "mn_internal.h"

struct mn_handle {
    MnArrays A;
    MnDev P;
    mn_params params;
    std::vector<void *> allocs;
    std::string err;
    int step_parity = 0;  // queue counter the NEXT step will fill
    int last_parity = 0;  // queue counter the LAST step filled
    uint32_t *seeds_dev = nullptr;
    uint32_t *mask_count = nullptr;
    int32_t *list_scratch = nullptr;
    double *peek_scratch = nullptr;
    double *obs64_buf = nullptr, *rew64_buf = nullptr;      // mn_enable_obs64: A.obs64 / A.rew64 point here while enabled
    int device = -1;      // HIP device the handle's memory lives on (the caller's current device at mn_create)
    // profiling
    std::vector<hipEvent_t> ev;
    int prof_max = 0, prof_n = 0;
    std::vector<hipEvent_t> ev_reset;      // ... and around mn_reset_done (mn_profile_reset_end)
    int prof_reset_n = 0;
    // mn_reset_done_async: the reset launch on the handle's own stream, under the caller's next act kernel
    hipStream_t side = nullptr;
    hipEvent_t ev_stepped = nullptr, ev_reset_end = nullptr;
    uint32_t *ready = nullptr;             // [n] "first observation is final" words (value = tick of the reset that wrote it)
    uint32_t tick = 0;
    bool reset_pending = false;            // a reset launched by mn_reset_done_async has not been joined by the caller's stream yet
    // decaying peak of the episodes the queue-driven reset launches start (mn_reset.hip: mn_note_count): device word + host-mapped copy (read without synchronising)
    volatile uint32_t *seen_host = nullptr;
    uint32_t *seen_dev = nullptr, *peak_dev = nullptr;
    int32_t under_act_max = MN_RESET_UNDER_ACT_MAX_DEFAULT;      // mn_reset_done_async: above this peak the reset runs in front of the act kernel
    bool async_ready = false;              // side stream, events, `ready`, the peak words: all there
    int32_t side_delay_us = 0;             // mn_debug_side_delay_us (tests): a sleeping kernel in front of every reset launch on the side stream
};

static thread_local std::string g_create_err;

#define MN_HIP(h, call)                                                                                     \
    do {                                                                                                    \
        hipError_t _e = (call);                                                                             \
        if (_e != hipSuccess) {                                                                             \
            char _b[512];                                                                                   \
            snprintf(_b, sizeof(_b), "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
            if (h) (h)->err = _b; else g_create_err = _b;                                                   \
            return MN_ERR_HIP;                                                                              \
        }                                                                                                   \
    } while (0)

static int fail(mn_handle *h, int code, const char *msg) {
    if (h) h->err = msg; else g_create_err = msg;
    return code;
}

// Every entry point that touches device memory runs on the device the handle was created on: a caller that has
// switched devices since (one process driving several GPUs) gets an error instead of a fault on a foreign pointer.
static int on_device(mn_handle *h) {
    if (!h) return MN_ERR_INVALID;
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) return fail(h, MN_ERR_HIP, "hipGetDevice failed");
    if (cur != h->device) {
        char b[160];
        snprintf(b, sizeof(b), "handle lives on HIP device %d but the calling thread's current device is %d", h->device, cur);
        return fail(h, MN_ERR_INVALID, b);
    }
    // an asynchronous reset (mn_reset_done_async) that the caller's stream has not joined: whatever this entry point is about to do with the
    // handle's state must come after it.  The per-step entry points join on their stream (join_reset) before they get here; the others wait on the host.
    if (h->reset_pending) {
        if (hipEventSynchronize(h->ev_reset_end) != hipSuccess) return fail(h, MN_ERR_HIP, "waiting for the asynchronous reset failed");
        h->reset_pending = false;
    }
    return MN_OK;
}
// stream-side join: everything enqueued on `s` from here on runs after the asynchronous reset
static void join_reset(mn_handle *h, hipStream_t s) {
    if (h && h->reset_pending) {
        if (hipStreamWaitEvent(s, h->ev_reset_end, 0) == hipSuccess) h->reset_pending = false;
    }
}
#define MN_ON_DEVICE(h) do { int _rc = on_device(h); if (_rc) return _rc; } while (0)

extern "C" int mn_default_params(mn_params *p) {
    if (!p) return MN_ERR_INVALID;
    memset(p, 0, sizeof(*p));
    p->width = 50; p->height = 50; p->core_r = 0.5; p->v_rel_max = 1.0; p->p = 0.8;
    p->v_range[0] = 5; p->v_range[1] = 10; p->obs_r_range[0] = 1; p->obs_r_range[1] = 3;
    p->clear_r = 10.0; p->goal_dis = 2.0; p->timestep_penalty = -1.0; p->collision_penalty = -50.0;
    p->goal_reward = 100.0; p->discount = 0.99; p->min_start_goal_dis = 25.0;
    p->init_theta = M_PI / 4; p->init_speed = 0.0;
    p->dt = 0.1; p->robot_r = 0.8; p->max_speed = 2.0;
    p->a[0] = -0.4; p->a[1] = 0.0; p->a[2] = 0.4;
    p->w[0] = -M_PI / 6; p->w[1] = 0.0; p->w[2] = M_PI / 6;
    p->sonar_range = 10.0; p->sonar_angle = 2 * M_PI / 3;
    p->num_cores = 8; p->num_obs = 5; p->reset_start_and_goal = 1; p->random_reset_state = 1;
    p->set_boundary = 0; p->max_episode_steps = 1000; p->N = 10; p->num_beams = MN_NUM_BEAMS;
    p->precision = MN_PRECISION_MIXED;
    return MN_OK;
}

// Host-side constants, evaluated in the reference's (python) expression order.
static int derive(mn_handle *h, const mn_params &p) {
    const int32_t keep_skip = h->P.debug_skip;   // only ever non-zero in -DMN_ABLATION builds (mn_set_debug_skip)
    if (p.num_beams != MN_NUM_BEAMS) return fail(h, MN_ERR_INVALID, "num_beams must be 11");
    if (p.num_cores < 0 || p.num_cores > MN_MAX_CORES) return fail(h, MN_ERR_INVALID, "num_cores out of [0, 8]");
    if (p.num_obs < 0 || p.num_obs > MN_MAX_OBS) return fail(h, MN_ERR_INVALID, "num_obs out of [0, 10]");
    if (p.N < 1 || p.N > 1000) return fail(h, MN_ERR_INVALID, "robot N out of range");
    if (p.precision != MN_PRECISION_F64 && p.precision != MN_PRECISION_MIXED) return fail(h, MN_ERR_INVALID, "bad precision");
    if (p.step_lanes != 0 && p.step_lanes != 1 && p.step_lanes != 2 && p.step_lanes != 4 && p.step_lanes != 8) return fail(h, MN_ERR_INVALID, "step_lanes must be 0 (default), 1, 2, 4 or 8");
    if (p.rollout_lanes != 0 && p.rollout_lanes != 2 && p.rollout_lanes != 4 && p.rollout_lanes != 8 && p.rollout_lanes != 16) return fail(h, MN_ERR_INVALID, "rollout_lanes must be 0 (default), 2, 4, 8 or 16");
    MnDev &d = h->P;
    const int32_t keep_n = d.n_stages;
    d.width = p.width; d.height = p.height; d.core_r = p.core_r; d.v_rel_max = p.v_rel_max; d.p = p.p;
    d.v_lo = p.v_range[0]; d.v_span = p.v_range[1] - p.v_range[0];
    d.or_lo = p.obs_r_range[0]; d.or_span = p.obs_r_range[1] - p.obs_r_range[0];
    d.clear_r = p.clear_r; d.goal_dis = p.goal_dis;
    d.timestep_penalty = p.timestep_penalty; d.collision_penalty = p.collision_penalty; d.goal_reward = p.goal_reward;
    d.min_start_goal_dis = p.min_start_goal_dis; d.init_theta = p.init_theta; d.init_speed = p.init_speed;
    d.dt = p.dt; d.robot_r = p.robot_r; d.max_speed = p.max_speed;
    double amax = p.a[0];
    for (int i = 0; i < 3; ++i) {
        d.a[i] = p.a[i]; d.w[i] = p.w[i]; if (p.a[i] > amax) amax = p.a[i];
        d.rot_c[i] = cos(p.w[i] * p.dt); d.rot_s[i] = sin(p.w[i] * p.dt);
    }
    d.k_drag = amax / p.max_speed;  // robot.py:52
    d.sonar_range = p.sonar_range;
    const double phi = p.sonar_angle / (p.num_beams - 1);  // robot.py:14-21
    const double a0 = -p.sonar_angle / 2;
    for (int i = 0; i < MN_NUM_BEAMS; ++i) {
        d.beam_rel[i] = a0 + i * phi;
        d.beam_cos[i] = cos(d.beam_rel[i]);
        d.beam_sin[i] = sin(d.beam_rel[i]);
    }
    d.fan_sin = sin(p.sonar_angle / 2); d.fan_cos = cos(p.sonar_angle / 2);
    d.fan_filter = (p.sonar_angle / 2 < 0.49 * M_PI) ? 1 : 0;
    d.two_pi = 2 * M_PI;
    d.two_pi_r = 2 * M_PI * p.core_r;
    d.two_pi_vrel = 2 * M_PI * p.v_rel_max;
    d.inv_two_pi_vrel = 1.0 / d.two_pi_vrel;
    d.two_pi_r_r = 2 * M_PI * p.core_r * p.core_r;
    d.binom_q = exp(1.0 * log(1.0 - 0.5));
    d.sg_lo_x = 2.0; d.sg_span_x = (p.width - 2.0) - 2.0; d.sg_lo_y = 2.0; d.sg_span_y = (p.height - 2.0) - 2.0;
    d.c_span_x = p.width - 0.0; d.c_span_y = p.height - 0.0;
    d.o_lo = 5.0; d.o_span_x = (p.width - 5.0) - 5.0; d.o_span_y = (p.height - 5.0) - 5.0;
    d.num_cores = p.num_cores; d.num_obs = p.num_obs; d.reset_start_and_goal = p.reset_start_and_goal;
    d.random_reset_state = p.random_reset_state; d.set_boundary = p.set_boundary;
    d.max_episode_steps = p.max_episode_steps; d.N = p.N;
    d.n_stages = keep_n;
    d.debug_skip = keep_skip;
    h->params = p;
    return MN_OK;
}

template <typename T>
static int dev_alloc(mn_handle *h, T **out, size_t count, bool zero = true) {
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, count * sizeof(T));
    if (e != hipSuccess) { h->err = std::string("hipMalloc failed: ") + hipGetErrorString(e); return MN_ERR_ALLOC; }
    if (zero) { e = hipMemset(p, 0, count * sizeof(T)); if (e != hipSuccess) { h->err = hipGetErrorString(e); return MN_ERR_HIP; } }
    h->allocs.push_back(p);
    *out = (T *)p;
    return MN_OK;
}

extern "C" int mn_destroy(mn_handle *h) {
    if (!h) return MN_ERR_INVALID;
    int cur = -1;
    const bool moved = hipGetDevice(&cur) == hipSuccess && h->device >= 0 && cur != h->device;
    if (moved) (void)hipSetDevice(h->device);   // free on the owning device, then restore the caller's
    for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->ev_reset) (void)hipEventDestroy(e);
    if (h->side) { (void)hipStreamSynchronize(h->side); (void)hipStreamDestroy(h->side); }
    if (h->ev_stepped) (void)hipEventDestroy(h->ev_stepped);
    if (h->ev_reset_end) (void)hipEventDestroy(h->ev_reset_end);
    if (h->seen_host) (void)hipHostFree((void *)h->seen_host);
    for (void *p : h->allocs) (void)hipFree(p);
    if (moved) (void)hipSetDevice(cur);
    delete h;
    return MN_OK;
}

extern "C" int mn_create(int32_t n_envs, const mn_params *p, mn_handle **out) {
    if (!out || !p || n_envs <= 0) return fail(nullptr, MN_ERR_INVALID, "mn_create: bad arguments");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(nullptr, MN_ERR_NO_DEVICE, "no HIP device visible");
    mn_handle *h = new mn_handle();
    if (hipGetDevice(&h->device) != hipSuccess) { delete h; return fail(nullptr, MN_ERR_HIP, "hipGetDevice failed"); }
    memset(&h->A, 0, sizeof(h->A));
    memset(&h->P, 0, sizeof(h->P));
    h->P.timestep_scale = 1.0;
    h->params.precision = p->precision;
    int rc = derive(h, *p);
    if (rc) { g_create_err = h->err; delete h; return rc; }
    MnArrays &A = h->A;
    A.n = n_envs;
    A.npad = (n_envs + MN_PAD - 1) / MN_PAD * MN_PAD;
    const size_t np = (size_t)A.npad;
#define ALLOC(field, count)                                    \
    if ((rc = dev_alloc(h, &A.field, (count))) != MN_OK) {      \
        g_create_err = h->err;                                 \
        mn_destroy(h);                                         \
        return rc;                                             \
    }
    ALLOC(x, np) ALLOC(y, np) ALLOC(theta, np) ALLOC(speed, np) ALLOC(vx, np) ALLOC(vy, np)
    ALLOC(start_x, np) ALLOC(start_y, np) ALLOC(goal_x, np) ALLOC(goal_y, np) ALLOC(init_theta, np) ALLOC(init_speed, np)
    ALLOC(ep_t, np) ALLOC(tot_t, np) ALLOC(counts, np)
    ALLOC(cx, np * MN_MAX_CORES) ALLOC(cy, np * MN_MAX_CORES) ALLOC(cg, np * MN_MAX_CORES)
    ALLOC(ox, np * MN_MAX_OBS) ALLOC(oy, np * MN_MAX_OBS) ALLOC(orad, np * MN_MAX_OBS)
    ALLOC(qcx, np * MN_MAX_CORES) ALLOC(qcy, np * MN_MAX_CORES) ALLOC(qcg, np * MN_MAX_CORES)
    ALLOC(qox, np * MN_MAX_OBS) ALLOC(qoy, np * MN_MAX_OBS) ALLOC(qor, np * MN_MAX_OBS)
    ALLOC(mt, np * 624) ALLOC(mt_pos, np)
    A.qcap = (int32_t)(((np >> 6) + MN_QSHARDS - 1) / MN_QSHARDS * 64);
    ALLOC(queue_count, 2 * MN_QWORDS) ALLOC(queue, (size_t)MN_QSHARDS * A.qcap)
    // (obs64 / rew64 -- float64 copies of the last observation rows / rewards -- are allocated and written only after mn_enable_obs64)
#undef ALLOC
    if ((rc = dev_alloc(h, &h->seeds_dev, np)) || (rc = dev_alloc(h, &h->mask_count, 1)) ||
        (rc = dev_alloc(h, &h->list_scratch, np)) || (rc = dev_alloc(h, &h->peek_scratch, np))) {
        g_create_err = h->err; mn_destroy(h); return rc;
    }
    // default start / goal (marinenav_env.py:49,53) and default seeds 0..n-1
    {
        std::vector<double> v(np);
        for (size_t i = 0; i < np; ++i) v[i] = 5.0;
        (void)hipMemcpy(A.start_x, v.data(), np * 8, hipMemcpyHostToDevice);
        (void)hipMemcpy(A.start_y, v.data(), np * 8, hipMemcpyHostToDevice);
        for (size_t i = 0; i < np; ++i) v[i] = 45.0;
        (void)hipMemcpy(A.goal_x, v.data(), np * 8, hipMemcpyHostToDevice);
        (void)hipMemcpy(A.goal_y, v.data(), np * 8, hipMemcpyHostToDevice);
        std::vector<uint32_t> s(np);
        for (size_t i = 0; i < np; ++i) s[i] = (uint32_t)i;
        rc = mn_seed(h, s.data(), nullptr);
        if (rc) { g_create_err = h->err; mn_destroy(h); return rc; }
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) { g_create_err = hipGetErrorString(e); mn_destroy(h); return MN_ERR_HIP; }
    }
    *out = h;
    return MN_OK;
}

extern "C" const char *mn_last_error(const mn_handle *h) { return h ? h->err.c_str() : g_create_err.c_str(); }
extern "C" int32_t mn_num_envs(const mn_handle *h) { return h ? h->A.n : 0; }

extern "C" int mn_set_params(mn_handle *h, const mn_params *p) {
    if (!h || !p) return MN_ERR_INVALID;
    if (p->precision != h->params.precision) return fail(h, MN_ERR_INVALID, "precision is fixed at mn_create");
    return derive(h, *p);
}

extern "C" int mn_get_params(const mn_handle *h, mn_params *p) {
    if (!h || !p) return MN_ERR_INVALID;
    *p = h->params;
    return MN_OK;
}

extern "C" int mn_seed(mn_handle *h, const uint32_t *seeds_host, void *stream) {
    if (!h || !seeds_host) return MN_ERR_INVALID;
    MN_ON_DEVICE(h);
    hipStream_t s = (hipStream_t)stream;
    MN_HIP(h, hipMemcpyAsync(h->seeds_dev, seeds_host, (size_t)h->A.n * 4, hipMemcpyHostToDevice, s));
    mn_launch_seed(h->A, h->seeds_dev, s);
    MN_HIP(h, hipGetLastError());
    MN_HIP(h, hipStreamSynchronize(s));  // seeds_host may be freed by the caller
    return MN_OK;
}

extern "C" int mn_set_schedule(mn_handle *h, int32_t n, const int64_t *ts, const int32_t *nc, const int32_t *no,
                               const double *md, double timestep_scale) {
    if (!h || n < 0 || n > MN_MAX_STAGES) return fail(h, MN_ERR_INVALID, "schedule: 0..8 stages");
    if (n > 0 && (!ts || !nc || !no || !md)) return MN_ERR_INVALID;
    for (int i = 0; i < n; ++i) {
        if (nc[i] < 0 || nc[i] > MN_MAX_CORES || no[i] < 0 || no[i] > MN_MAX_OBS) return fail(h, MN_ERR_INVALID, "schedule world size exceeds capacity (8 cores, 10 obstacles)");
        h->P.sched_t[i] = ts[i]; h->P.sched_nc[i] = nc[i]; h->P.sched_no[i] = no[i]; h->P.sched_md[i] = md[i];
    }
    h->P.n_stages = n;
    h->P.timestep_scale = timestep_scale > 0 ? timestep_scale : 1.0;
    return MN_OK;
}

static int range_ok(mn_handle *h, int first, int count) {
    if (!h || first < 0 || count < 0 || first + count > h->A.n) return fail(h, MN_ERR_INVALID, "env range out of bounds");
    return on_device(h);
}

extern "C" int mn_set_start_goal(mn_handle *h, int32_t env_idx, const double start[2], const double goal[2]) {
    if (!h || !start || !goal || env_idx >= h->A.n) return MN_ERR_INVALID;
    MN_ON_DEVICE(h);
    const int first = env_idx < 0 ? 0 : env_idx, count = env_idx < 0 ? h->A.n : 1;
    std::vector<double> v(count);
    double *dst[4] = {h->A.start_x, h->A.start_y, h->A.goal_x, h->A.goal_y};
    const double val[4] = {start[0], start[1], goal[0], goal[1]};
    for (int k = 0; k < 4; ++k) {
        for (int i = 0; i < count; ++i) v[i] = val[k];
        MN_HIP(h, hipMemcpy(dst[k] + first, v.data(), (size_t)count * 8, hipMemcpyHostToDevice));
    }
    return MN_OK;
}

extern "C" int mn_reset(mn_handle *h, const uint8_t *mask_dev, float *obs_dev, void *stream) {
    if (!h || !obs_dev) return MN_ERR_INVALID;
    MN_ON_DEVICE(h);
    hipStream_t s = (hipStream_t)stream;
    if (!mask_dev) {
        mn_launch_reset(h->A, h->P, h->params.precision, nullptr, (uint32_t)h->A.n, nullptr, 0, obs_dev, s);
    } else {
        MN_HIP(h, hipMemsetAsync(h->mask_count, 0, 4, s));
        mn_launch_mask_to_queue(h->A, mask_dev, h->mask_count, h->list_scratch, s);
        mn_launch_reset(h->A, h->P, h->params.precision, h->mask_count, 0, h->list_scratch, 0, obs_dev, s);
    }
    MN_HIP(h, hipGetLastError());
    return MN_OK;
}

static int step_common(mn_handle *h, const int32_t *actions_dev, float *obs_dev, float *reward_dev, uint8_t *done_dev,
                       uint8_t *info_dev, const MnRing *ring, void *stream) {
    if (!h || !actions_dev || !obs_dev || !reward_dev || !done_dev || !info_dev) return MN_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    join_reset(h, s);
    MN_ON_DEVICE(h);
    const int parity = h->step_parity;
    const bool prof = h->prof_n < h->prof_max;
    if (prof) (void)hipEventRecord(h->ev[2 * h->prof_n], s);
    mn_launch_step(h->A, h->P, h->params.precision, h->params.step_lanes, actions_dev, obs_dev, reward_dev, done_dev, info_dev, parity, ring, s);
    if (prof) { (void)hipEventRecord(h->ev[2 * h->prof_n + 1], s); h->prof_n++; }
    MN_HIP(h, hipGetLastError());
    h->last_parity = parity;
    h->step_parity = parity ^ 1;
    return MN_OK;
}

extern "C" int mn_step(mn_handle *h, const int32_t *actions_dev, float *obs_dev, float *reward_dev, uint8_t *done_dev,
                       uint8_t *info_dev, void *stream) {
    return step_common(h, actions_dev, obs_dev, reward_dev, done_dev, info_dev, nullptr, stream);
}

extern "C" int mn_step_append(mn_handle *h, const int32_t *actions_dev, const float *prev_obs_dev, float *obs_dev,
                              float *reward_dev, uint8_t *done_dev, uint8_t *info_dev, float *ring_states,
                              float *ring_next_states, int64_t *ring_actions, float *ring_rewards, float *ring_dones,
                              int64_t ptr, int64_t capacity, void *stream) {
    if (!h || !prev_obs_dev || !ring_states || !ring_next_states || !ring_actions || !ring_rewards || !ring_dones)
        return MN_ERR_INVALID;
    if (prev_obs_dev == obs_dev) return fail(h, MN_ERR_INVALID, "mn_step_append: obs_t and obs_t+1 must be different buffers");
    if (capacity <= 0 || ptr < 0 || ptr >= capacity) return fail(h, MN_ERR_INVALID, "mn_step_append: ptr out of [0, capacity)");
    const MnRing R = {prev_obs_dev, ring_states, ring_next_states, ring_actions, ring_rewards, ring_dones, ptr, capacity};
    return step_common(h, actions_dev, obs_dev, reward_dev, done_dev, info_dev, &R, stream);
}

extern "C" int mn_rollout(mn_handle *h, int32_t n_steps, const int32_t *actions_dev, uint64_t action_seed,
                          uint64_t first_step_index, uint64_t first_env_index, float *obs_dev, float *obs_trace_dev,
                          float *reward_trace_dev, uint8_t *done_trace_dev, uint8_t *info_trace_dev,
                          int32_t *action_trace_dev, void *stream) {
    if (!h || !obs_dev || n_steps < 1) return MN_ERR_INVALID;
    MN_ON_DEVICE(h);
    hipStream_t s = (hipStream_t)stream;
    const bool prof = h->prof_n < h->prof_max;
    if (prof) (void)hipEventRecord(h->ev[2 * h->prof_n], s);
    mn_launch_rollout(h->A, h->P, h->params.precision, h->params.rollout_lanes, n_steps, actions_dev, action_seed, first_step_index,
                      first_env_index, obs_dev, obs_trace_dev, reward_trace_dev, done_trace_dev, info_trace_dev,
                      action_trace_dev, s);
    if (prof) { (void)hipEventRecord(h->ev[2 * h->prof_n + 1], s); h->prof_n++; }
    MN_HIP(h, hipGetLastError());
    // the kernel zeroed both done-queue counters: a following mn_reset_done has nothing to do, the next mn_step starts clean
    h->step_parity = 0;
    h->last_parity = 0;
    return MN_OK;
}

extern "C" int mn_rollout_policy(mn_handle *h, int32_t n_steps, int32_t policy, float *obs_dev, float *obs_trace_dev, float *reward_trace_dev,
                                 uint8_t *done_trace_dev, uint8_t *info_trace_dev, int32_t *action_trace_dev, void *stream) {
    if (!h || !obs_dev || n_steps < 1 || (policy != MN_POLICY_APF && policy != MN_POLICY_BA)) return MN_ERR_INVALID;
    MN_ON_DEVICE(h);
    hipStream_t s = (hipStream_t)stream;
    const bool prof = h->prof_n < h->prof_max;
    if (prof) (void)hipEventRecord(h->ev[2 * h->prof_n], s);
    mn_launch_rollout_policy(h->A, h->P, h->params.precision, n_steps, policy, obs_dev, obs_trace_dev, reward_trace_dev, done_trace_dev,
                             info_trace_dev, action_trace_dev, s);
    if (prof) { (void)hipEventRecord(h->ev[2 * h->prof_n + 1], s); h->prof_n++; }
    MN_HIP(h, hipGetLastError());
    h->step_parity = 0;
    h->last_parity = 0;
    return MN_OK;
}

extern "C" int mn_planner_act(const float *obs_dev, int32_t n, int32_t policy, const double *a, const double *w, int32_t *actions_dev, void *stream) {
    if (!obs_dev || !actions_dev || !a || !w || n <= 0 || (policy != MN_POLICY_APF && policy != MN_POLICY_BA)) return MN_ERR_INVALID;
    mn_launch_planner_act(obs_dev, n, policy, a, w, actions_dev, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? MN_OK : MN_ERR_HIP;
}

extern "C" int mn_random_actions(uint64_t action_seed, uint64_t step_index, uint64_t first_env_index, int32_t n,
                                 int32_t *actions_dev, void *stream) {
    if (!actions_dev || n <= 0) return MN_ERR_INVALID;
    mn_launch_random_actions(action_seed, step_index, first_env_index, n, actions_dev, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? MN_OK : MN_ERR_HIP;
}

extern "C" int32_t mn_build_info(void) {
#ifdef MN_ABLATION
    return MN_BUILD_ABLATION;
#else
    return 0;
#endif
}

#ifdef MN_ABLATION
// Developer builds only (libmarinenav_hip_ablation.so): removes parts of the step kernel to attribute its time.
extern "C" int mn_set_debug_skip(mn_handle *h, int32_t mask) {
    if (!h) return MN_ERR_INVALID;
    h->P.debug_skip = mask;
    return MN_OK;
}
#endif

// mn_reset_done on the handle's own stream, so that it runs UNDER whatever the caller enqueues next on `stream` -- in the training loop the act
// kernel of the next vector step, which is told which rows are still being written (mn_iqn_set_late_rows: the step's done flags, `*ready_out`,
// `*tick_out`) and takes them last.  The caller's stream is joined again by mn_reset_join, or by the next per-step entry point of this handle.
// Beside the act kernel's workgroups a CU has room for ONE reset wavefront (LDS), which runs ~4 x slower there: that hides a few hundred resets
// (a latency chain that leaves the chip idle when it runs alone) but not thousands.  So the launch goes under the act kernel only while the last
// reset launch the host has seen started at most `under_act_max` episodes (mn_set_reset_under_act_max; the count arrives through a host-mapped word,
// no synchronisation); otherwise this IS mn_reset_done and *ready_out is NULL.
extern "C" int mn_reset_done_async(mn_handle *h, float *obs_dev, void *stream, const uint32_t **ready_out, uint32_t *tick_out) {
    if (!h || !obs_dev || !ready_out || !tick_out) return MN_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    join_reset(h, s);
    MN_ON_DEVICE(h);
    if (!h->async_ready) {
        // a stream of ANOTHER priority gets a hardware queue of its own: created with the default priority it can end up sharing the queue of the caller's
        // stream (HIP deals streams out over a few queues; with an RCCL communicator in the process it did), and then the reset launch simply runs in front of
        // the act kernel again, behind two cross-stream events (measured: 0.380 instead of 0.360 ms per vector step)
        // (every piece is created if it is not there yet -- a call that failed half way is completed by the next one -- and `async_ready` is set last)
        int prio_lo = 0, prio_hi = 0;
        MN_HIP(h, hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
        if (!h->side) MN_HIP(h, hipStreamCreateWithPriority(&h->side, hipStreamNonBlocking, prio_hi));
        if (!h->ev_stepped) MN_HIP(h, hipEventCreateWithFlags(&h->ev_stepped, hipEventDisableTiming));
        if (!h->ev_reset_end) MN_HIP(h, hipEventCreateWithFlags(&h->ev_reset_end, hipEventDisableTiming));
        if (!h->ready) { int rc = dev_alloc(h, &h->ready, (size_t)h->A.npad); if (rc) return rc; }
        if (!h->seen_host) {
            void *hp = nullptr;
            MN_HIP(h, hipHostMalloc(&hp, sizeof(uint32_t), hipHostMallocMapped));
            *(volatile uint32_t *)hp = 0xffffffffu;      // nothing seen yet: the first launches run in front
            h->seen_host = (volatile uint32_t *)hp;
        }
        if (!h->seen_dev) MN_HIP(h, hipHostGetDevicePointer((void **)&h->seen_dev, (void *)h->seen_host, 0));
        if (!h->peak_dev) { int rc = dev_alloc(h, &h->peak_dev, 1); if (rc) return rc; }
        h->async_ready = true;
    }
    *ready_out = nullptr;
    *tick_out = 0;
    // the estimate: the launches' own decaying peak of their episode counts, as of the last launch whose word has arrived (the host may run several
    // launches ahead of the device) -- a burst of episode ends keeps the resets in front for ~20 vector steps
    const uint32_t seen = *h->seen_host;
    if (h->under_act_max < 0 || (h->under_act_max != 0x7fffffff && (seen == 0xffffffffu || seen > (uint32_t)h->under_act_max))) {
        const bool prof = h->prof_reset_n < h->prof_max;
        if (prof) (void)hipEventRecord(h->ev_reset[2 * h->prof_reset_n], s);
        mn_launch_reset(h->A, h->P, h->params.precision, h->A.queue_count + h->last_parity * MN_QWORDS, 0, h->A.queue, 0, obs_dev, s, h->peak_dev, h->seen_dev, true);
        if (prof) { (void)hipEventRecord(h->ev_reset[2 * h->prof_reset_n + 1], s); h->prof_reset_n++; }
        MN_HIP(h, hipGetLastError());
        return MN_OK;
    }
    MN_HIP(h, hipEventRecord(h->ev_stepped, s));
    MN_HIP(h, hipStreamWaitEvent(h->side, h->ev_stepped, 0));
    h->tick += 1;
    const bool prof = h->prof_reset_n < h->prof_max;
    if (prof) (void)hipEventRecord(h->ev_reset[2 * h->prof_reset_n], h->side);
    if (h->side_delay_us > 0) mn_launch_sleep((uint32_t)h->side_delay_us, h->side);
    mn_launch_reset_under_act(h->A, h->P, h->params.precision, h->A.queue_count + h->last_parity * MN_QWORDS, h->A.queue, obs_dev, h->ready, h->tick, h->peak_dev, h->seen_dev, h->side);
    if (prof) { (void)hipEventRecord(h->ev_reset[2 * h->prof_reset_n + 1], h->side); h->prof_reset_n++; }
    MN_HIP(h, hipGetLastError());
    MN_HIP(h, hipEventRecord(h->ev_reset_end, h->side));
    h->reset_pending = true;
    *ready_out = h->ready;
    *tick_out = h->tick;
    return MN_OK;
}

// under_act_max: mn_reset_done_async launches under the next act kernel while the decaying peak of the episodes started per reset launch (as of the last
// launch seen) is at most this (default 1200; 0x7fffffff: always, -1: never).  *last_seen: that peak (-1: none seen yet).
extern "C" int mn_set_reset_under_act_max(mn_handle *h, int32_t under_act_max, int64_t *last_seen) {
    if (!h) return MN_ERR_INVALID;
    h->under_act_max = under_act_max;
    if (last_seen) *last_seen = (h->seen_host && *h->seen_host != 0xffffffffu) ? (int64_t)*h->seen_host : -1;
    return MN_OK;
}

// Test hook: every reset launch mn_reset_done_async puts on the handle's own stream is preceded there by a kernel that sleeps `us` microseconds -- a reset
// launch that does NOT run beside the act kernel (shared hardware queue, no room on the CUs), without needing such a box.  0 switches it off.
extern "C" int mn_debug_side_delay_us(mn_handle *h, int32_t us) {
    if (!h || us < 0 || us > 10000000) return MN_ERR_INVALID;
    h->side_delay_us = us;
    return MN_OK;
}

extern "C" int mn_reset_join(mn_handle *h, void *stream) {
    if (!h) return MN_ERR_INVALID;
    join_reset(h, (hipStream_t)stream);
    return h->reset_pending ? fail(h, MN_ERR_HIP, "hipStreamWaitEvent failed") : MN_OK;
}

extern "C" int mn_reset_done(mn_handle *h, float *obs_dev, void *stream) {
    if (!h || !obs_dev) return MN_ERR_INVALID;
    join_reset(h, (hipStream_t)stream);
    MN_ON_DEVICE(h);
    const bool prof = h->prof_reset_n < h->prof_max;
    if (prof) (void)hipEventRecord(h->ev_reset[2 * h->prof_reset_n], (hipStream_t)stream);
    mn_launch_reset(h->A, h->P, h->params.precision, h->A.queue_count + h->last_parity * MN_QWORDS, 0, h->A.queue, 0, obs_dev, (hipStream_t)stream, h->peak_dev, h->seen_dev, true);
    if (prof) { (void)hipEventRecord(h->ev_reset[2 * h->prof_reset_n + 1], (hipStream_t)stream); h->prof_reset_n++; }
    MN_HIP(h, hipGetLastError());
    return MN_OK;
}

extern "C" int mn_last_done_count(mn_handle *h, void *stream, int32_t *out) {
    if (!h || !out) return MN_ERR_INVALID;
    MN_ON_DEVICE(h);
    MN_HIP(h, hipStreamSynchronize((hipStream_t)stream));
    uint32_t w[MN_QWORDS], v = 0;      // the shard counters of the last step's done-queue
    MN_HIP(h, hipMemcpy(w, h->A.queue_count + h->last_parity * MN_QWORDS, sizeof(w), hipMemcpyDeviceToHost));
    for (int sh = 0; sh < MN_QSHARDS; ++sh) v += w[sh * MN_QSTRIDE];
    *out = (int32_t)v;
    return MN_OK;
}

// [count][K] host rows <-> [K][npad] device table
static int table_to_dev(mn_handle *h, double *dev, int first, int count, int K, const std::vector<double> &host_kc) {
    MN_HIP(h, hipMemcpy2D(dev + first, (size_t)h->A.npad * 8, host_kc.data(), (size_t)count * 8, (size_t)count * 8, K, hipMemcpyHostToDevice));
    return MN_OK;
}
static int table_from_dev(mn_handle *h, const double *dev, int first, int count, int K, std::vector<double> &host_kc) {
    host_kc.resize((size_t)K * count);
    MN_HIP(h, hipMemcpy2D(host_kc.data(), (size_t)count * 8, dev + first, (size_t)h->A.npad * 8, (size_t)count * 8, K, hipMemcpyDeviceToHost));
    return MN_OK;
}

extern "C" int mn_load_worlds(mn_handle *h, int32_t first, int32_t count, const int32_t *n_cores, const double *cores_xy,
                              const int32_t *clockwise, const double *gamma, const int32_t *n_obs, const double *obs_xy,
                              const double *obs_r, const double *start, const double *goal, const double *init_theta,
                              const double *init_speed, float *obs_dev, void *stream) {
    int rc = range_ok(h, first, count);
    if (rc) return rc;
    if (count == 0) return MN_OK;
    if (!n_cores || !cores_xy || !clockwise || !gamma || !n_obs || !obs_xy || !obs_r || !start || !goal || !init_theta || !init_speed)
        return MN_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    MN_HIP(h, hipStreamSynchronize(s));
    const int C = MN_MAX_CORES, O = MN_MAX_OBS;
    std::vector<double> cx((size_t)C * count), cy((size_t)C * count), cg((size_t)C * count);
    std::vector<double> ox((size_t)O * count), oy((size_t)O * count), orr((size_t)O * count);
    std::vector<double> sx(count), sy(count), gx(count), gy(count);
    std::vector<int32_t> cnt(count), zeros(count, 0), list(count);
    for (int i = 0; i < count; ++i) {
        if (n_cores[i] < 0 || n_cores[i] > C || n_obs[i] < 0 || n_obs[i] > O) return fail(h, MN_ERR_INVALID, "world exceeds capacity (8 cores, 10 obstacles)");
        for (int k = 0; k < C; ++k) {
            const bool v = k < n_cores[i];
            cx[(size_t)k * count + i] = v ? cores_xy[((size_t)i * C + k) * 2] : 0.0;
            cy[(size_t)k * count + i] = v ? cores_xy[((size_t)i * C + k) * 2 + 1] : 0.0;
            cg[(size_t)k * count + i] = v ? (clockwise[(size_t)i * C + k] ? gamma[(size_t)i * C + k] : -gamma[(size_t)i * C + k]) : 0.0;
        }
        for (int k = 0; k < O; ++k) {
            const bool v = k < n_obs[i];
            ox[(size_t)k * count + i] = v ? obs_xy[((size_t)i * O + k) * 2] : 0.0;
            oy[(size_t)k * count + i] = v ? obs_xy[((size_t)i * O + k) * 2 + 1] : 0.0;
            orr[(size_t)k * count + i] = v ? obs_r[(size_t)i * O + k] : 0.0;
        }
        sx[i] = start[2 * i]; sy[i] = start[2 * i + 1]; gx[i] = goal[2 * i]; gy[i] = goal[2 * i + 1];
        cnt[i] = n_cores[i] | (n_obs[i] << 8);
        list[i] = first + i;
    }
    if ((rc = table_to_dev(h, h->A.cx, first, count, C, cx)) || (rc = table_to_dev(h, h->A.cy, first, count, C, cy)) ||
        (rc = table_to_dev(h, h->A.cg, first, count, C, cg)) || (rc = table_to_dev(h, h->A.ox, first, count, O, ox)) ||
        (rc = table_to_dev(h, h->A.oy, first, count, O, oy)) || (rc = table_to_dev(h, h->A.orad, first, count, O, orr)))
        return rc;
    const size_t b8 = (size_t)count * 8;
    MN_HIP(h, hipMemcpy(h->A.start_x + first, sx.data(), b8, hipMemcpyHostToDevice));
    MN_HIP(h, hipMemcpy(h->A.start_y + first, sy.data(), b8, hipMemcpyHostToDevice));
    MN_HIP(h, hipMemcpy(h->A.goal_x + first, gx.data(), b8, hipMemcpyHostToDevice));
    MN_HIP(h, hipMemcpy(h->A.goal_y + first, gy.data(), b8, hipMemcpyHostToDevice));
    MN_HIP(h, hipMemcpy(h->A.init_theta + first, init_theta, b8, hipMemcpyHostToDevice));
    MN_HIP(h, hipMemcpy(h->A.init_speed + first, init_speed, b8, hipMemcpyHostToDevice));
    MN_HIP(h, hipMemcpy(h->A.counts + first, cnt.data(), (size_t)count * 4, hipMemcpyHostToDevice));
    MN_HIP(h, hipMemcpy(h->list_scratch, list.data(), (size_t)count * 4, hipMemcpyHostToDevice));
    mn_launch_reset(h->A, h->P, h->params.precision, nullptr, (uint32_t)count, h->list_scratch, 1, obs_dev, s);
    MN_HIP(h, hipGetLastError());
    MN_HIP(h, hipStreamSynchronize(s));
    return MN_OK;
}

extern "C" int mn_get_worlds(mn_handle *h, int32_t first, int32_t count, int32_t *n_cores, double *cores_xy, int32_t *clockwise,
                             double *gamma, int32_t *n_obs, double *obs_xy, double *obs_r, double *start, double *goal,
                             double *init_theta, double *init_speed) {
    int rc = range_ok(h, first, count);
    if (rc) return rc;
    if (count == 0) return MN_OK;
    MN_HIP(h, hipDeviceSynchronize());
    const int C = MN_MAX_CORES, O = MN_MAX_OBS;
    std::vector<double> cx, cy, cg, ox, oy, orr, sx(count), sy(count), gx(count), gy(count);
    std::vector<int32_t> cnt(count);
    if ((rc = table_from_dev(h, h->A.cx, first, count, C, cx)) || (rc = table_from_dev(h, h->A.cy, first, count, C, cy)) ||
        (rc = table_from_dev(h, h->A.cg, first, count, C, cg)) || (rc = table_from_dev(h, h->A.ox, first, count, O, ox)) ||
        (rc = table_from_dev(h, h->A.oy, first, count, O, oy)) || (rc = table_from_dev(h, h->A.orad, first, count, O, orr)))
        return rc;
    const size_t b8 = (size_t)count * 8;
    MN_HIP(h, hipMemcpy(sx.data(), h->A.start_x + first, b8, hipMemcpyDeviceToHost));
    MN_HIP(h, hipMemcpy(sy.data(), h->A.start_y + first, b8, hipMemcpyDeviceToHost));
    MN_HIP(h, hipMemcpy(gx.data(), h->A.goal_x + first, b8, hipMemcpyDeviceToHost));
    MN_HIP(h, hipMemcpy(gy.data(), h->A.goal_y + first, b8, hipMemcpyDeviceToHost));
    if (init_theta) MN_HIP(h, hipMemcpy(init_theta, h->A.init_theta + first, b8, hipMemcpyDeviceToHost));
    if (init_speed) MN_HIP(h, hipMemcpy(init_speed, h->A.init_speed + first, b8, hipMemcpyDeviceToHost));
    MN_HIP(h, hipMemcpy(cnt.data(), h->A.counts + first, (size_t)count * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < count; ++i) {
        const int nc = cnt[i] & 0xff, no = (cnt[i] >> 8) & 0xff;
        if (n_cores) n_cores[i] = nc;
        if (n_obs) n_obs[i] = no;
        for (int k = 0; k < C; ++k) {
            const double g = cg[(size_t)k * count + i];
            if (cores_xy) { cores_xy[((size_t)i * C + k) * 2] = cx[(size_t)k * count + i]; cores_xy[((size_t)i * C + k) * 2 + 1] = cy[(size_t)k * count + i]; }
            if (clockwise) clockwise[(size_t)i * C + k] = g > 0.0 ? 1 : 0;
            if (gamma) gamma[(size_t)i * C + k] = fabs(g);
        }
        for (int k = 0; k < O; ++k) {
            if (obs_xy) { obs_xy[((size_t)i * O + k) * 2] = ox[(size_t)k * count + i]; obs_xy[((size_t)i * O + k) * 2 + 1] = oy[(size_t)k * count + i]; }
            if (obs_r) obs_r[(size_t)i * O + k] = orr[(size_t)k * count + i];
        }
        if (start) { start[2 * i] = sx[i]; start[2 * i + 1] = sy[i]; }
        if (goal) { goal[2 * i] = gx[i]; goal[2 * i + 1] = gy[i]; }
    }
    return MN_OK;
}

extern "C" int mn_get_state(mn_handle *h, int32_t first, int32_t count, double *state, int32_t *ep_t, int64_t *tot_t) {
    int rc = range_ok(h, first, count);
    if (rc) return rc;
    if (count == 0) return MN_OK;
    MN_HIP(h, hipDeviceSynchronize());
    if (state) {
        std::vector<double> v(count);
        const double *src[6] = {h->A.x, h->A.y, h->A.theta, h->A.speed, h->A.vx, h->A.vy};
        for (int k = 0; k < 6; ++k) {
            MN_HIP(h, hipMemcpy(v.data(), src[k] + first, (size_t)count * 8, hipMemcpyDeviceToHost));
            for (int i = 0; i < count; ++i) state[(size_t)i * 6 + k] = v[i];
        }
    }
    if (ep_t) MN_HIP(h, hipMemcpy(ep_t, h->A.ep_t + first, (size_t)count * 4, hipMemcpyDeviceToHost));
    if (tot_t) MN_HIP(h, hipMemcpy(tot_t, h->A.tot_t + first, (size_t)count * 8, hipMemcpyDeviceToHost));
    return MN_OK;
}

extern "C" int mn_set_state(mn_handle *h, int32_t first, int32_t count, const double *state, const int32_t *ep_t,
                            const int64_t *tot_t) {
    int rc = range_ok(h, first, count);
    if (rc) return rc;
    if (count == 0) return MN_OK;
    MN_HIP(h, hipDeviceSynchronize());
    if (state) {
        std::vector<double> v(count);
        double *dst[6] = {h->A.x, h->A.y, h->A.theta, h->A.speed, h->A.vx, h->A.vy};
        for (int k = 0; k < 6; ++k) {
            for (int i = 0; i < count; ++i) v[i] = state[(size_t)i * 6 + k];
            MN_HIP(h, hipMemcpy(dst[k] + first, v.data(), (size_t)count * 8, hipMemcpyHostToDevice));
        }
    }
    if (ep_t) MN_HIP(h, hipMemcpy(h->A.ep_t + first, ep_t, (size_t)count * 4, hipMemcpyHostToDevice));
    if (tot_t) MN_HIP(h, hipMemcpy(h->A.tot_t + first, tot_t, (size_t)count * 8, hipMemcpyHostToDevice));
    return MN_OK;
}

extern "C" int mn_get_obs64(mn_handle *h, int32_t first, int32_t count, double *out) {
    int rc = range_ok(h, first, count);
    if (rc) return rc;
    if (count == 0) return MN_OK;
    if (!out) return MN_ERR_INVALID;
    if (!h->A.obs64) return fail(h, MN_ERR_INVALID, "float64 observations are kept only with MN_PRECISION_F64 after mn_enable_obs64(h, 1)");
    MN_HIP(h, hipDeviceSynchronize());
    MN_HIP(h, hipMemcpy(out, h->A.obs64 + (size_t)first * MN_OBS_DIM, (size_t)count * MN_OBS_DIM * 8, hipMemcpyDeviceToHost));
    return MN_OK;
}

extern "C" int mn_get_reward64(mn_handle *h, int32_t first, int32_t count, double *out) {
    int rc = range_ok(h, first, count);
    if (rc) return rc;
    if (count == 0) return MN_OK;
    if (!out) return MN_ERR_INVALID;
    if (!h->A.rew64) return fail(h, MN_ERR_INVALID, "float64 rewards are kept only with MN_PRECISION_F64 after mn_enable_obs64(h, 1)");
    MN_HIP(h, hipDeviceSynchronize());
    MN_HIP(h, hipMemcpy(out, h->A.rew64 + first, (size_t)count * 8, hipMemcpyDeviceToHost));
    return MN_OK;
}

extern "C" int mn_enable_obs64(mn_handle *h, int32_t on) {
    if (!h || on < 0 || on > 1) return MN_ERR_INVALID;
    MN_ON_DEVICE(h);
    if (h->params.precision != MN_PRECISION_F64) return fail(h, MN_ERR_INVALID, "float64 observation copies exist only with MN_PRECISION_F64");
    MN_HIP(h, hipDeviceSynchronize());
    if (!on) { h->A.obs64 = nullptr; h->A.rew64 = nullptr; return MN_OK; }      // the buffers stay owned by the handle
    if (!h->obs64_buf) {
        int rc = dev_alloc(h, &h->obs64_buf, (size_t)h->A.npad * MN_OBS_DIM);
        if (rc == MN_OK) rc = dev_alloc(h, &h->rew64_buf, (size_t)h->A.npad);
        if (rc) return rc;
    }
    h->A.obs64 = h->obs64_buf; h->A.rew64 = h->rew64_buf;
    return MN_OK;
}

extern "C" int mn_enable_trajectory(mn_handle *h, int32_t max_substeps) {
    if (!h || max_substeps < 1 || max_substeps > 1000) return MN_ERR_INVALID;
    MN_ON_DEVICE(h);
    if (h->params.precision != MN_PRECISION_F64) return fail(h, MN_ERR_INVALID, "sub-step trajectories are recorded only with MN_PRECISION_F64");
    if (h->A.traj && h->A.traj_n >= max_substeps) return MN_OK;
    MN_HIP(h, hipDeviceSynchronize());
    int rc = dev_alloc(h, &h->A.traj, (size_t)h->A.npad * max_substeps * 2);   // an earlier, smaller buffer stays owned by the handle
    if (rc) return rc;
    h->A.traj_n = max_substeps;
    return MN_OK;
}

extern "C" int mn_get_trajectory(mn_handle *h, int32_t first, int32_t count, int32_t n_substeps, double *out) {
    int rc = range_ok(h, first, count);
    if (rc) return rc;
    if (count == 0) return MN_OK;
    if (!out || n_substeps < 1) return MN_ERR_INVALID;
    if (!h->A.traj || n_substeps > h->A.traj_n) return fail(h, MN_ERR_INVALID, "call mn_enable_trajectory(h, >= n_substeps) before stepping");
    MN_HIP(h, hipDeviceSynchronize());
    MN_HIP(h, hipMemcpy2D(out, (size_t)n_substeps * 16, h->A.traj + (size_t)first * h->A.traj_n * 2, (size_t)h->A.traj_n * 16,
                          (size_t)n_substeps * 16, count, hipMemcpyDeviceToHost));
    return MN_OK;
}

extern "C" int mn_peek_next_double(mn_handle *h, int32_t first, int32_t count, double *out) {
    int rc = range_ok(h, first, count);
    if (rc) return rc;
    if (count == 0) return MN_OK;
    if (!out) return MN_ERR_INVALID;
    MN_HIP(h, hipDeviceSynchronize());
    mn_launch_peek(h->A, first, count, h->peek_scratch, nullptr);
    MN_HIP(h, hipGetLastError());
    MN_HIP(h, hipMemcpy(out, h->peek_scratch, (size_t)count * 8, hipMemcpyDeviceToHost));
    return MN_OK;
}

extern "C" int mn_profile_begin(mn_handle *h, int32_t max_launches) {
    if (!h || max_launches < 0) return MN_ERR_INVALID;
    while ((int)h->ev.size() < 2 * max_launches) {
        hipEvent_t e;
        MN_HIP(h, hipEventCreate(&e));
        h->ev.push_back(e);
    }
    while ((int)h->ev_reset.size() < 2 * max_launches) {
        hipEvent_t e;
        MN_HIP(h, hipEventCreate(&e));
        h->ev_reset.push_back(e);
    }
    h->prof_max = max_launches;
    h->prof_n = 0;
    h->prof_reset_n = 0;
    return MN_OK;
}

// The mn_reset_done launches of the same window (call BEFORE mn_profile_end, which closes the window).
extern "C" int mn_profile_reset_end(mn_handle *h, void *stream, double *mean_ms, int32_t *launches) {
    if (!h) return MN_ERR_INVALID;
    MN_HIP(h, hipStreamSynchronize((hipStream_t)stream));
    if (h->side) MN_HIP(h, hipStreamSynchronize(h->side));      // (launches of mn_reset_done_async are timed on the handle's own stream)
    double sum = 0.0;
    for (int i = 0; i < h->prof_reset_n; ++i) {
        float ms = 0.f;
        MN_HIP(h, hipEventElapsedTime(&ms, h->ev_reset[2 * i], h->ev_reset[2 * i + 1]));
        sum += ms;
    }
    if (mean_ms) *mean_ms = h->prof_reset_n ? sum / h->prof_reset_n : 0.0;
    if (launches) *launches = h->prof_reset_n;
    h->prof_reset_n = 0;
    return MN_OK;
}

extern "C" int mn_profile_end(mn_handle *h, void *stream, double *mean_ms, int32_t *launches) {
    if (!h) return MN_ERR_INVALID;
    MN_HIP(h, hipStreamSynchronize((hipStream_t)stream));
    double sum = 0.0;
    for (int i = 0; i < h->prof_n; ++i) {
        float ms = 0.f;
        MN_HIP(h, hipEventElapsedTime(&ms, h->ev[2 * i], h->ev[2 * i + 1]));
        sum += ms;
    }
    if (mean_ms) *mean_ms = h->prof_n ? sum / h->prof_n : 0.0;
    if (launches) *launches = h->prof_n;
    h->prof_max = 0;
    h->prof_n = 0;
    h->prof_reset_n = 0;
    return MN_OK;
}
