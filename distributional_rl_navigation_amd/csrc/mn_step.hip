// mn_step.hip -- fused marinenav_env step kernel for gfx950 (MI355X).
//
// Replaces, for a batch of environments, MarineNavEnv.step (marinenav_env.py:199-262):
//   N x [ get_velocity (:422-465) -> Robot.update_state (robot.py:102-123) ],
//   get_observation (:273-326) with Robot.sonar_reflection (robot.py:125-198),
//   reward + termination ladder (:220-257), counters (:259-260).
//
// Mapping: one lane per environment, one 64-lane wavefront (= one workgroup) per tile of 64
// consecutive envs.  All state is SoA with the env index fastest, so every global load/store
// instruction moves one contiguous run per wave.  The vortex/obstacle tables of the tile are
// staged in LDS as [row][lane] (lane-private columns, conflict-free ds_read_b64), which gives
// dynamically indexable per-env tables without spending ~108 VGPRs on them; the same LDS is
// reused at the end to transpose the 64x26 observation tile so it leaves as 16-byte coalesced
// row-major stores.
#include "mn_device.h"

namespace {

template <typename M, bool PARITY>
__global__ __launch_bounds__(MN_WAVE) void mn_step_kernel(MnArrays A, MnDev P, const int32_t *__restrict__ actions,
                                                          float *__restrict__ obs_out, float *__restrict__ reward_out,
                                                          uint8_t *__restrict__ done_out, uint8_t *__restrict__ info_out,
                                                          int parity) {
    __shared__ double tab[MN_TAB_ROWS][MN_WAVE];  // 54 x 64 x 8 B = 27 KiB per wave
    const int lane = threadIdx.x;
    const int e = blockIdx.x * MN_WAVE + lane;
    const bool active = e < A.n;
    const int np = A.npad;

    if (blockIdx.x == 0 && lane == 0) A.queue_count[parity ^ 1] = 0u;  // counter of the NEXT step

    // ---- load state (coalesced, one element per lane) -----------------------------------------
    double x = A.x[e], y = A.y[e], theta = A.theta[e], speed = A.speed[e];
    const double gx = A.goal_x[e], gy = A.goal_y[e];
    const int cnt = A.counts[e];
    const int nc = cnt & 0xff, no = (cnt >> 8) & 0xff;
    int ep_t = A.ep_t[e];
    int action = active ? actions[e] : 0;
    action = action < 0 ? 0 : (action > 8 ? 8 : action);

    // ---- stage world tables in LDS (lane-private columns: no barrier needed) ------------------
#pragma unroll
    for (int k = 0; k < MN_MAX_CORES; ++k) {
        const bool v = k < nc;
        tab[k][lane] = v ? A.cx[k * np + e] : 0.0;
        tab[MN_MAX_CORES + k][lane] = v ? A.cy[k * np + e] : 0.0;
        tab[2 * MN_MAX_CORES + k][lane] = v ? A.cg[k * np + e] : 0.0;
    }
    constexpr int OB = 3 * MN_MAX_CORES;
#pragma unroll
    for (int k = 0; k < MN_MAX_OBS; ++k) {
        const bool v = k < no;
        tab[OB + k][lane] = v ? A.ox[k * np + e] : 0.0;
        tab[OB + MN_MAX_OBS + k][lane] = v ? A.oy[k * np + e] : 0.0;
        tab[OB + 2 * MN_MAX_OBS + k][lane] = v ? A.orad[k * np + e] : 0.0;
    }

    // marinenav_env.py:205 dis_before
    const double dbx = gx - x, dby = gy - y;
    const double dis_before = sqrt(dbx * dbx + dby * dby);

    // robot.py:55-56: actions[i] = (a[i // 3], w[i % 3])
    const int ai = action / 3, wi = action - 3 * ai;
    const double acc = ai == 0 ? P.a[0] : (ai == 1 ? P.a[1] : P.a[2]);
    const double wv = wi == 0 ? P.w[0] : (wi == 1 ? P.w[1] : P.w[2]);
    const double dt = P.dt;
    const double two_pi = P.two_pi;

    const M r2 = (M)(P.core_r * P.core_r);
    const M inv_two_pi_r2 = (M)(1.0 / P.two_pi_r_r);
    const M inv_two_pi = (M)(1.0 / P.two_pi);

    // ---- N kinematic sub-steps (marinenav_env.py:208-212) -------------------------------------
    M velx = 0, vely = 0;
    for (int s = 0; s < P.N; ++s) {
        // current at the pre-move position: superposition over ALL cores
        M cvx = 0, cvy = 0;
        for (int k = 0; k < nc; ++k) {
            const M dx = (M)(tab[k][lane] - x);
            const M dy = (M)(tab[MN_MAX_CORES + k][lane] - y);
            mn_core_velocity<M>(dx, dy, (M)tab[2 * MN_MAX_CORES + k][lane], r2, inv_two_pi_r2, inv_two_pi, cvx, cvy);
        }
        // robot.py:98-107: velocity = speed*(cos,sin) + current ; position += velocity*dt
        M sn, cs;
        MnMath<M>::sincos_((M)theta, &sn, &cs);
        velx = (M)speed * cs + cvx;
        vely = (M)speed * sn + cvy;
        x += (double)velx * dt;
        y += (double)vely * dt;
        // robot.py:113-114: drag + clip
        speed += (acc - P.k_drag * speed) * dt;
        speed = speed < 0.0 ? 0.0 : (speed > P.max_speed ? P.max_speed : speed);
        // robot.py:117-123: heading + wrap to [0, 2pi)
        theta += wv * dt;
        while (theta < 0.0) theta += two_pi;
        while (theta >= two_pi) theta -= two_pi;
    }

    // marinenav_env.py:214 dis_after
    const double dax = gx - x, day = gy - y;
    const double dis_after = sqrt(dax * dax + day * day);

    // ---- observation (marinenav_env.py:273-326) ------------------------------------------------
    // Final-heading rotation in float64 (one sincos per step): the goal vector (|g| up to 70 m) and
    // the sonar geometry cancel too much for float32 sin/cos.
    double sn, cs;
    sincos(theta, &sn, &cs);
    M ob[MN_OBS_DIM];
    {
        const M c = (M)cs, s_ = (M)sn;
        ob[0] = c * velx + s_ * vely;  // R(theta)^T * velocity: lagged velocity, final heading (App. A K5)
        ob[1] = -s_ * velx + c * vely;
    }
    ob[2] = (M)(cs * dax + sn * day);  // R(theta)^T (goal - p)
    ob[3] = (M)(-sn * dax + cs * day);
    const M range = (M)P.sonar_range;
    // obstacle centres in the robot frame, once per obstacle
    double mrx[MN_MAX_OBS], mry[MN_MAX_OBS], orr2[MN_MAX_OBS];
#pragma unroll
    for (int k = 0; k < MN_MAX_OBS; ++k) {
        const double mx = tab[OB + k][lane] - x, my = tab[OB + MN_MAX_OBS + k][lane] - y;
        const double r = tab[OB + 2 * MN_MAX_OBS + k][lane];
        mrx[k] = cs * mx + sn * my;
        mry[k] = -sn * mx + cs * my;
        orr2[k] = r * r;
    }
    const double half_pi = 0.5 * 3.141592653589793, three_half_pi = 3 * 3.141592653589793 / 2;
#pragma unroll
    for (int b = 0; b < MN_NUM_BEAMS; ++b) {
        const double angle = theta + P.beam_rel[b];  // robot.py:134, not wrapped
        const bool up = fabs(angle - half_pi) < 1e-03;
        const bool down = fabs(angle - three_half_pi) < 1e-03;
        // beam direction in the robot frame: the constant (cos rel, sin rel); a snapped beam points
        // along world (0,+-1), i.e. R^T (0,+-1) = +-(sin theta, cos theta)
        double bx = P.beam_cos[b], by = P.beam_sin[b];
        if (up || down) {
            const double sg = up ? 1.0 : -1.0;
            bx = sg * sn; by = sg * cs;
        }
        bool hit = false, stopped = false;
        M dist = M(0);
#pragma unroll
        for (int k = 0; k < MN_MAX_OBS; ++k) {
            if (k < no) {
                double tc, h2;
                mn_beam_geom(mrx[k], mry[k], orr2[k], bx, by, tc, h2);
                mn_beam_update<M>((M)tc, (M)h2, range, hit, dist, stopped);
            }
        }
        ob[4 + 2 * b] = hit ? dist * (M)bx : M(0);  // misses are (0,0): marinenav_env.py:315-316
        ob[5 + 2 * b] = hit ? dist * (M)by : M(0);
    }

    // ---- reward + termination ladder (marinenav_env.py:220-257) -------------------------------
    double reward = P.timestep_penalty;
    reward += dis_before - dis_after;
    // check_collision (:329-336): nearest-CENTRE obstacle only
    bool collide = false;
    {
        double best = 1e300, best_r = 0.0;
        for (int k = 0; k < no; ++k) {
            const double mx = tab[OB + k][lane] - x, my = tab[OB + MN_MAX_OBS + k][lane] - y;
            const double d2 = mx * mx + my * my;
            if (d2 < best) { best = d2; best_r = tab[OB + 2 * MN_MAX_OBS + k][lane]; }
        }
        collide = no > 0 && sqrt(best) <= best_r + P.robot_r;
    }
    const bool reach = dis_after <= P.goal_dis;  // check_reach_goal (:338-342)
    const bool out = (x < 0.0 || x > P.width) || (y < 0.0 || y > P.height);
    int done, info;
    if (P.set_boundary && out) { done = 1; info = MN_INFO_OUT_OF_BOUNDARY; }
    else if (ep_t >= P.max_episode_steps) { done = 1; info = MN_INFO_TOO_LONG; }
    else if (collide) { reward += P.collision_penalty; done = 1; info = MN_INFO_COLLISION; }
    else if (reach) { reward += P.goal_reward; done = 1; info = MN_INFO_REACH_GOAL; }
    else { done = 0; info = MN_INFO_NORMAL; }

    // ---- write back ----------------------------------------------------------------------------
    if (active) {
        A.x[e] = x; A.y[e] = y; A.theta[e] = theta; A.speed[e] = speed;
        A.vx[e] = (double)velx; A.vy[e] = (double)vely;
        A.ep_t[e] = ep_t + 1;
        A.tot_t[e] += 1;
        reward_out[e] = (float)reward;
        done_out[e] = (uint8_t)done;
        info_out[e] = (uint8_t)info;
        if (PARITY) {
            A.rew64[e] = reward;
#pragma unroll
            for (int j = 0; j < MN_OBS_DIM; ++j) A.obs64[(size_t)e * MN_OBS_DIM + j] = (double)ob[j];
        }
    }
    // done-queue: one atomic per wave
    {
        const unsigned long long m = __ballot(active && done);
        if (m) {
            unsigned base = 0;
            if (lane == 0) base = atomicAdd(&A.queue_count[parity], (unsigned)__popcll(m));
            base = __shfl(base, 0);
            if (active && done) A.queue[base + __popcll(m & ((1ull << lane) - 1ull))] = e;
        }
    }
    // observation tile: transpose through LDS -> row-major [env][26] float32, 16-byte stores
    __syncthreads();  // every lane is done with its table column
    float *ot = reinterpret_cast<float *>(&tab[0][0]);
#pragma unroll
    for (int j = 0; j < MN_OBS_DIM; ++j) ot[lane * MN_OBS_DIM + j] = (float)ob[j];
    __syncthreads();
    const int tile_first = blockIdx.x * MN_WAVE;
    const int valid = A.n - tile_first;
    float *gdst = obs_out + (size_t)tile_first * MN_OBS_DIM;
    if (valid >= MN_WAVE) {
        const float4 *src4 = reinterpret_cast<const float4 *>(ot);
        float4 *dst4 = reinterpret_cast<float4 *>(gdst);
        constexpr int NV = MN_WAVE * MN_OBS_DIM / 4;  // 416
        for (int q = lane; q < NV; q += MN_WAVE) dst4[q] = src4[q];
    } else {
        const int nf = valid * MN_OBS_DIM;
        for (int q = lane; q < nf; q += MN_WAVE) gdst[q] = ot[q];
    }
}

}  // namespace

void mn_launch_step(const MnArrays &A, const MnDev &P, int precision, const int32_t *actions, float *obs, float *reward,
                    uint8_t *done, uint8_t *info, int parity, hipStream_t s) {
    const dim3 grid(A.npad / MN_WAVE), block(MN_WAVE);
    if (precision == MN_PRECISION_F64)
        hipLaunchKernelGGL((mn_step_kernel<double, true>), grid, block, 0, s, A, P, actions, obs, reward, done, info, parity);
    else
        hipLaunchKernelGGL((mn_step_kernel<float, false>), grid, block, 0, s, A, P, actions, obs, reward, done, info, parity);
}
