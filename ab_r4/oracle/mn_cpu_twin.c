/*
 * mn_cpu_twin.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * The environment entry points of include/marinenav_hip.h (mn_create ... mn_last_done_count) with IDENTICAL
 * signatures, implemented on the host over the scalar float64 oracle (marinenav_oracle.c, a literal restatement of
 * marinenav_env/envs/marinenav_env.py + utils/robot.py).  Every `*_dev` pointer is a HOST pointer here and `stream` is
 * ignored.  Built as oracle/libmarinenav_cpu.so so that ONE ctypes binding (the SIGNATURES table of the package's
 * _capi.py, or the stub of INTEGRATION.md) can be pointed at either library: tests/test_cpu_twin.py drives both through
 * the same code.  Only tests load it; the product package never does (SURVEY.md section 8b "CPU twin").
 *
 * Semantics follow the header: float32 observations / rewards out (rounded from the oracle's float64), float64 copies
 * behind mn_get_obs64 / mn_get_reward64 (kept in both precisions: the twin always computes in float64), done-queue
 * behaviour of mn_reset_done, per-env MT19937 streams seeded 0..n-1 by default.
 * Not provided: the IQN / replay entry points (GPU kernels without a scalar restatement here), sub-step trajectories.
 */
#include "marinenav_oracle.c"

#include <stdio.h>

#include "../include/marinenav_hip.h"

struct mn_handle {
    int32_t n;
    orc_env **env;
    mn_params params;
    double timestep_scale;
    double *obs64;      /* [n][26] */
    double *rew64;      /* [n] */
    uint8_t *last_done; /* [n] flags of the last mn_step */
    int32_t last_done_count;
    char err[256];
};

static char g_create_err[256];

static int fail(mn_handle *h, int code, const char *msg) {
    snprintf(h ? h->err : g_create_err, 256, "%s", msg);
    return code;
}

int mn_default_params(mn_params *p) {
    if (!p) return MN_ERR_INVALID;
    memset(p, 0, sizeof(*p));
    orc_env *e = orc_create(0);
    p->width = e->width; p->height = e->height; p->core_r = e->r; p->v_rel_max = e->v_rel_max; p->p = e->p;
    p->v_range[0] = e->v_range[0]; p->v_range[1] = e->v_range[1];
    p->obs_r_range[0] = e->obs_r_range[0]; p->obs_r_range[1] = e->obs_r_range[1];
    p->clear_r = e->clear_r; p->goal_dis = e->goal_dis; p->timestep_penalty = e->timestep_penalty;
    p->collision_penalty = e->collision_penalty; p->goal_reward = e->goal_reward; p->discount = e->discount;
    p->min_start_goal_dis = e->min_start_goal_dis; p->init_theta = e->init_theta; p->init_speed = e->init_speed;
    p->dt = e->dt; p->robot_r = e->robot_r; p->max_speed = e->max_speed;
    for (int i = 0; i < 3; i++) { p->a[i] = e->a[i]; p->w[i] = e->w[i]; }
    p->sonar_range = e->sonar_range; p->sonar_angle = e->sonar_angle;
    p->num_cores = e->num_cores; p->num_obs = e->num_obs; p->reset_start_and_goal = e->reset_start_and_goal;
    p->random_reset_state = e->random_reset_state; p->set_boundary = e->set_boundary;
    p->max_episode_steps = 1000; p->N = e->N; p->num_beams = e->num_beams;
    p->precision = MN_PRECISION_MIXED;
    orc_destroy(e);
    return MN_OK;
}

static int check_params(mn_handle *h, const mn_params *p) {
    if (p->num_beams != MN_NUM_BEAMS) return fail(h, MN_ERR_INVALID, "num_beams must be 11");
    if (p->num_cores < 0 || p->num_cores > MN_MAX_CORES) return fail(h, MN_ERR_INVALID, "num_cores out of [0, 8]");
    if (p->num_obs < 0 || p->num_obs > MN_MAX_OBS) return fail(h, MN_ERR_INVALID, "num_obs out of [0, 10]");
    if (p->N < 1 || p->N > 1000) return fail(h, MN_ERR_INVALID, "robot N out of range");
    if (p->precision != MN_PRECISION_F64 && p->precision != MN_PRECISION_MIXED) return fail(h, MN_ERR_INVALID, "bad precision");
    if (p->max_episode_steps != 1000) return fail(h, MN_ERR_INVALID, "the CPU twin keeps the reference's 1000-step episode limit");
    return MN_OK;
}

static void apply_params(orc_env *e, const mn_params *p) {
    e->width = p->width; e->height = p->height; e->r = p->core_r; e->v_rel_max = p->v_rel_max; e->p = p->p;
    e->v_range[0] = p->v_range[0]; e->v_range[1] = p->v_range[1];
    e->obs_r_range[0] = p->obs_r_range[0]; e->obs_r_range[1] = p->obs_r_range[1];
    e->clear_r = p->clear_r; e->goal_dis = p->goal_dis; e->timestep_penalty = p->timestep_penalty;
    e->collision_penalty = p->collision_penalty; e->goal_reward = p->goal_reward; e->discount = p->discount;
    e->min_start_goal_dis = p->min_start_goal_dis; e->init_theta = p->init_theta; e->init_speed = p->init_speed;
    e->dt = p->dt; e->robot_r = p->robot_r; e->max_speed = p->max_speed;
    double amax = p->a[0];
    for (int i = 0; i < 3; i++) { e->a[i] = p->a[i]; e->w[i] = p->w[i]; if (p->a[i] > amax) amax = p->a[i]; }
    e->k = amax / p->max_speed;
    e->sonar_range = p->sonar_range; e->sonar_angle = p->sonar_angle; e->num_beams = p->num_beams;
    sonar_setup(e);
    e->num_cores = p->num_cores; e->num_obs = p->num_obs; e->reset_start_and_goal = p->reset_start_and_goal;
    e->random_reset_state = p->random_reset_state; e->set_boundary = p->set_boundary; e->N = p->N;
}

int mn_destroy(mn_handle *h) {
    if (!h) return MN_ERR_INVALID;
    if (h->env) for (int i = 0; i < h->n; i++) if (h->env[i]) orc_destroy(h->env[i]);
    free(h->env); free(h->obs64); free(h->rew64); free(h->last_done);
    free(h);
    return MN_OK;
}

int mn_create(int32_t n_envs, const mn_params *p, mn_handle **out) {
    if (!out || !p || n_envs <= 0) return fail(NULL, MN_ERR_INVALID, "mn_create: bad arguments");
    mn_handle *h = (mn_handle *)calloc(1, sizeof(mn_handle));
    int rc = check_params(h, p);
    if (rc) { snprintf(g_create_err, 256, "%s", h->err); free(h); return rc; }
    h->n = n_envs; h->params = *p; h->timestep_scale = 1.0;
    h->env = (orc_env **)calloc((size_t)n_envs, sizeof(orc_env *));
    h->obs64 = (double *)calloc((size_t)n_envs * MN_OBS_DIM, sizeof(double));
    h->rew64 = (double *)calloc((size_t)n_envs, sizeof(double));
    h->last_done = (uint8_t *)calloc((size_t)n_envs, 1);
    for (int i = 0; i < n_envs; i++) { h->env[i] = orc_create((uint32_t)i); apply_params(h->env[i], p); }
    *out = h;
    return MN_OK;
}

const char *mn_last_error(const mn_handle *h) { return h ? h->err : g_create_err; }
int32_t mn_num_envs(const mn_handle *h) { return h ? h->n : 0; }
int32_t mn_build_info(void) { return 0; }

int mn_set_params(mn_handle *h, const mn_params *p) {
    if (!h || !p) return MN_ERR_INVALID;
    if (p->precision != h->params.precision) return fail(h, MN_ERR_INVALID, "precision is fixed at mn_create");
    int rc = check_params(h, p);
    if (rc) return rc;
    h->params = *p;
    for (int i = 0; i < h->n; i++) apply_params(h->env[i], p);
    return MN_OK;
}

int mn_get_params(const mn_handle *h, mn_params *p) {
    if (!h || !p) return MN_ERR_INVALID;
    *p = h->params;
    return MN_OK;
}

int mn_seed(mn_handle *h, const uint32_t *seeds_host, void *stream) {
    (void)stream;
    if (!h || !seeds_host) return MN_ERR_INVALID;
    for (int i = 0; i < h->n; i++) orc_seed(h->env[i], seeds_host[i]);
    return MN_OK;
}

int mn_set_schedule(mn_handle *h, int32_t n, const int64_t *ts, const int32_t *nc, const int32_t *no, const double *md,
                    double timestep_scale) {
    if (!h || n < 0 || n > MN_MAX_STAGES) return fail(h, MN_ERR_INVALID, "schedule: 0..8 stages");
    if (n > 0 && (!ts || !nc || !no || !md)) return MN_ERR_INVALID;
    for (int i = 0; i < n; i++)
        if (nc[i] < 0 || nc[i] > MN_MAX_CORES || no[i] < 0 || no[i] > MN_MAX_OBS)
            return fail(h, MN_ERR_INVALID, "schedule world size exceeds capacity (8 cores, 10 obstacles)");
    for (int i = 0; i < h->n; i++) orc_set_schedule(h->env[i], n, ts, nc, no, md);
    h->timestep_scale = timestep_scale > 0 ? timestep_scale : 1.0;
    return MN_OK;
}

int mn_set_start_goal(mn_handle *h, int32_t env_idx, const double start[2], const double goal[2]) {
    if (!h || !start || !goal || env_idx >= h->n) return MN_ERR_INVALID;
    const int first = env_idx < 0 ? 0 : env_idx, count = env_idx < 0 ? h->n : 1;
    for (int i = first; i < first + count; i++) orc_set_start_goal(h->env[i], start[0], start[1], goal[0], goal[1]);
    return MN_OK;
}

static void store_obs(mn_handle *h, int i, const double *o, float *obs_out) {
    memcpy(h->obs64 + (size_t)i * MN_OBS_DIM, o, MN_OBS_DIM * sizeof(double));
    if (obs_out) for (int k = 0; k < MN_OBS_DIM; k++) obs_out[(size_t)i * MN_OBS_DIM + k] = (float)o[k];
}

/* MarineNavEnv.reset with the curriculum looked up at floor(total_timesteps * timestep_scale) (see mn_set_schedule) */
static void reset_one(mn_handle *h, int i, float *obs_out) {
    orc_env *e = h->env[i];
    double o[MN_OBS_DIM];
    const int64_t keep = e->total_timesteps;
    e->total_timesteps = (int64_t)((double)keep * h->timestep_scale);
    orc_reset(e, o);
    e->total_timesteps = keep;
    store_obs(h, i, o, obs_out);
}

int mn_reset(mn_handle *h, const uint8_t *mask_dev, float *obs_dev, void *stream) {
    (void)stream;
    if (!h || !obs_dev) return MN_ERR_INVALID;
    for (int i = 0; i < h->n; i++) if (!mask_dev || mask_dev[i]) reset_one(h, i, obs_dev);
    return MN_OK;
}

int mn_step(mn_handle *h, const int32_t *actions_dev, float *obs_dev, float *reward_dev, uint8_t *done_dev,
            uint8_t *info_dev, void *stream) {
    (void)stream;
    if (!h || !actions_dev || !obs_dev || !reward_dev || !done_dev || !info_dev) return MN_ERR_INVALID;
    h->last_done_count = 0;
    for (int i = 0; i < h->n; i++) {
        double o[MN_OBS_DIM], r;
        int info, a = actions_dev[i];
        a = a < 0 ? 0 : (a > 8 ? 8 : a);
        const int done = orc_step(h->env[i], a, o, &r, &info);
        store_obs(h, i, o, obs_dev);
        h->rew64[i] = r;
        reward_dev[i] = (float)r; done_dev[i] = (uint8_t)done; info_dev[i] = (uint8_t)info;
        h->last_done[i] = (uint8_t)done;
        h->last_done_count += done;
    }
    return MN_OK;
}

int mn_step_append(mn_handle *h, const int32_t *actions_dev, const float *prev_obs_dev, float *obs_dev, float *reward_dev,
                   uint8_t *done_dev, uint8_t *info_dev, float *ring_states, float *ring_next_states,
                   int64_t *ring_actions, float *ring_rewards, float *ring_dones, int64_t ptr, int64_t capacity,
                   void *stream) {
    if (!h || !prev_obs_dev || !ring_states || !ring_next_states || !ring_actions || !ring_rewards || !ring_dones)
        return MN_ERR_INVALID;
    if (prev_obs_dev == obs_dev) return fail(h, MN_ERR_INVALID, "mn_step_append: obs_t and obs_t+1 must be different buffers");
    if (capacity <= 0 || ptr < 0 || ptr >= capacity) return fail(h, MN_ERR_INVALID, "mn_step_append: ptr out of [0, capacity)");
    int rc = mn_step(h, actions_dev, obs_dev, reward_dev, done_dev, info_dev, stream);
    if (rc) return rc;
    const int64_t first = h->n > capacity ? h->n - capacity : 0;
    for (int64_t e = first; e < h->n; e++) {
        int64_t slot = ptr + (e - first);
        slot = slot >= capacity ? slot - capacity : slot;
        memcpy(ring_states + slot * MN_OBS_DIM, prev_obs_dev + e * MN_OBS_DIM, MN_OBS_DIM * sizeof(float));
        memcpy(ring_next_states + slot * MN_OBS_DIM, obs_dev + e * MN_OBS_DIM, MN_OBS_DIM * sizeof(float));
        ring_actions[slot] = actions_dev[e];
        ring_rewards[slot] = reward_dev[e];
        ring_dones[slot] = done_dev[e] ? 1.0f : 0.0f;
    }
    return MN_OK;
}

int mn_reset_done(mn_handle *h, float *obs_dev, void *stream) {
    (void)stream;
    if (!h || !obs_dev) return MN_ERR_INVALID;
    for (int i = 0; i < h->n; i++) if (h->last_done[i]) reset_one(h, i, obs_dev);
    return MN_OK;
}

int mn_last_done_count(mn_handle *h, void *stream, int32_t *out) {
    (void)stream;
    if (!h || !out) return MN_ERR_INVALID;
    *out = h->last_done_count;
    return MN_OK;
}

static int range_ok(mn_handle *h, int first, int count) {
    if (!h || first < 0 || count < 0 || first + count > h->n) return fail(h, MN_ERR_INVALID, "env range out of bounds");
    return MN_OK;
}

int mn_load_worlds(mn_handle *h, int32_t first, int32_t count, const int32_t *n_cores, const double *cores_xy,
                   const int32_t *clockwise, const double *gamma, const int32_t *n_obs, const double *obs_xy,
                   const double *obs_r, const double *start, const double *goal, const double *init_theta,
                   const double *init_speed, float *obs_dev, void *stream) {
    (void)stream;
    int rc = range_ok(h, first, count);
    if (rc) return rc;
    if (count == 0) return MN_OK;
    if (!n_cores || !cores_xy || !clockwise || !gamma || !n_obs || !obs_xy || !obs_r || !start || !goal || !init_theta || !init_speed)
        return MN_ERR_INVALID;
    for (int i = 0; i < count; i++)
        if (n_cores[i] < 0 || n_cores[i] > MN_MAX_CORES || n_obs[i] < 0 || n_obs[i] > MN_MAX_OBS)
            return fail(h, MN_ERR_INVALID, "world exceeds capacity (8 cores, 10 obstacles)");
    for (int i = 0; i < count; i++) {
        double o[MN_OBS_DIM];
        int cw[MN_MAX_CORES];
        for (int k = 0; k < MN_MAX_CORES; k++) cw[k] = clockwise[(size_t)i * MN_MAX_CORES + k];
        orc_load_world(h->env[first + i], n_cores[i], cores_xy + (size_t)i * MN_MAX_CORES * 2, cw, gamma + (size_t)i * MN_MAX_CORES,
                       n_obs[i], obs_xy + (size_t)i * MN_MAX_OBS * 2, obs_r + (size_t)i * MN_MAX_OBS, start + 2 * i, goal + 2 * i,
                       init_theta[i], init_speed[i], o);
        store_obs(h, first + i, o, obs_dev);
    }
    return MN_OK;
}

int mn_get_worlds(mn_handle *h, int32_t first, int32_t count, int32_t *n_cores, double *cores_xy, int32_t *clockwise,
                  double *gamma, int32_t *n_obs, double *obs_xy, double *obs_r, double *start, double *goal,
                  double *init_theta, double *init_speed) {
    int rc = range_ok(h, first, count);
    if (rc) return rc;
    for (int i = 0; i < count; i++) {
        const orc_env *e = h->env[first + i];
        if (n_cores) n_cores[i] = e->n_cores;
        if (n_obs) n_obs[i] = e->n_obstacles;
        for (int k = 0; k < MN_MAX_CORES; k++) {
            const int v = k < e->n_cores;
            if (cores_xy) { cores_xy[((size_t)i * MN_MAX_CORES + k) * 2] = v ? e->cores[k].x : 0.0; cores_xy[((size_t)i * MN_MAX_CORES + k) * 2 + 1] = v ? e->cores[k].y : 0.0; }
            if (clockwise) clockwise[(size_t)i * MN_MAX_CORES + k] = v ? e->cores[k].clockwise : 0;
            if (gamma) gamma[(size_t)i * MN_MAX_CORES + k] = v ? e->cores[k].Gamma : 0.0;
        }
        for (int k = 0; k < MN_MAX_OBS; k++) {
            const int v = k < e->n_obstacles;
            if (obs_xy) { obs_xy[((size_t)i * MN_MAX_OBS + k) * 2] = v ? e->obstacles[k].x : 0.0; obs_xy[((size_t)i * MN_MAX_OBS + k) * 2 + 1] = v ? e->obstacles[k].y : 0.0; }
            if (obs_r) obs_r[(size_t)i * MN_MAX_OBS + k] = v ? e->obstacles[k].r : 0.0;
        }
        if (start) { start[2 * i] = e->start[0]; start[2 * i + 1] = e->start[1]; }
        if (goal) { goal[2 * i] = e->goal[0]; goal[2 * i + 1] = e->goal[1]; }
        if (init_theta) init_theta[i] = e->r_init_theta;
        if (init_speed) init_speed[i] = e->r_init_speed;
    }
    return MN_OK;
}

int mn_get_state(mn_handle *h, int32_t first, int32_t count, double *state, int32_t *ep_t, int64_t *tot_t) {
    int rc = range_ok(h, first, count);
    if (rc) return rc;
    for (int i = 0; i < count; i++) {
        double s6[6];
        int64_t c2[2];
        orc_get_state(h->env[first + i], s6, c2);
        if (state) memcpy(state + (size_t)i * 6, s6, sizeof(s6));
        if (ep_t) ep_t[i] = (int32_t)c2[0];
        if (tot_t) tot_t[i] = c2[1];
    }
    return MN_OK;
}

int mn_set_state(mn_handle *h, int32_t first, int32_t count, const double *state, const int32_t *ep_t, const int64_t *tot_t) {
    int rc = range_ok(h, first, count);
    if (rc) return rc;
    for (int i = 0; i < count; i++) {
        orc_env *e = h->env[first + i];
        if (state) orc_set_state(e, state + (size_t)i * 6, ep_t ? ep_t[i] : e->episode_timesteps);
        else if (ep_t) e->episode_timesteps = ep_t[i];
        if (tot_t) e->total_timesteps = tot_t[i];
    }
    return MN_OK;
}

/* the twin always computes in float64 and always keeps the copies: the switch only has to exist */
int mn_enable_obs64(mn_handle *h, int32_t on) {
    if (!h || on < 0 || on > 1) return MN_ERR_INVALID;
    return MN_OK;
}

int mn_get_obs64(mn_handle *h, int32_t first, int32_t count, double *out) {
    int rc = range_ok(h, first, count);
    if (rc) return rc;
    if (count == 0) return MN_OK;
    if (!out) return MN_ERR_INVALID;
    memcpy(out, h->obs64 + (size_t)first * MN_OBS_DIM, (size_t)count * MN_OBS_DIM * sizeof(double));
    return MN_OK;
}

int mn_get_reward64(mn_handle *h, int32_t first, int32_t count, double *out) {
    int rc = range_ok(h, first, count);
    if (rc) return rc;
    if (count == 0) return MN_OK;
    if (!out) return MN_ERR_INVALID;
    memcpy(out, h->rew64 + first, (size_t)count * sizeof(double));
    return MN_OK;
}

int mn_peek_next_double(mn_handle *h, int32_t first, int32_t count, double *out) {
    int rc = range_ok(h, first, count);
    if (rc) return rc;
    if (count == 0) return MN_OK;
    if (!out) return MN_ERR_INVALID;
    for (int i = 0; i < count; i++) out[i] = orc_peek_next_double(h->env[first + i]);
    return MN_OK;
}

int mn_profile_begin(mn_handle *h, int32_t max_launches) { (void)max_launches; return h ? MN_OK : MN_ERR_INVALID; }
int mn_profile_end(mn_handle *h, void *stream, double *mean_ms, int32_t *launches) {
    (void)stream;
    if (!h) return MN_ERR_INVALID;
    if (mean_ms) *mean_ms = 0.0;
    if (launches) *launches = 0;
    return MN_OK;
}
