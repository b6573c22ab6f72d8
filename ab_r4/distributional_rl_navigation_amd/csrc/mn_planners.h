// mn_planners.h -- the classical baselines of the reference's comparison table as device functions (gfx950).
//
// APF_agent.act (APF.py:17-78: artificial potential field) and BA_agent.act (BA.py:14-155: bug algorithm with a least-squares wall)
// map ONE 26-dim observation -- [velocity (2), goal (2), 11 sonar points (22), misses are exactly (0, 0)] in the robot frame -- to one
// of the 9 actions a_idx * 3 + w_idx.  Both are stateless.  Here: float64 like the reference's numpy arithmetic, one observation per
// call, no loops over python objects; used by mn_planner_act (one launch per policy step for a whole vector of observations) and by the
// episode rollout kernel (mn_rollout.hip), where the policy runs on the observation the step just produced, in the same launch.
// The observation arrives as the float32 row the step kernels write (what the batched planners of planners.py are fed as well).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "marinenav_hip.h"      // MN_POLICY_APF, MN_POLICY_BA

struct MnPlanTabs {
    double a[3], w[3];      // the robot's linear-acceleration / angular-velocity tables (robot.py:35-38)
};

// BA.py:157-162 / APF.py:55-59.  The arguments are differences of two atan2 values or one +- a margin, i.e. inside (-3 pi, 3 pi): two
// conditional shifts are the reference's while loops, and values already in range pass through unchanged
__device__ __forceinline__ double mn_wrap_pi(double a) {
    const double PI = 3.141592653589793;
#pragma unroll
    for (int i = 0; i < 2; ++i) a = a < -PI ? a + 2.0 * PI : (a >= PI ? a - 2.0 * PI : a);
    return a;
}

// index of the first minimum of |t[k] - x| over the three table entries (np.argmin)
__device__ __forceinline__ int mn_nearest3(const double (&t)[3], double x) {
    const double d0 = fabs(t[0] - x), d1 = fabs(t[1] - x), d2 = fabs(t[2] - x);
    int k = 0;
    double best = d0;
    if (d1 < best) { best = d1; k = 1; }
    if (d2 < best) { k = 2; }
    return k;
}
__device__ __forceinline__ int mn_argmax3(const double (&t)[3]) { int k = 0; if (t[1] > t[k]) k = 1; if (t[2] > t[k]) k = 2; return k; }
__device__ __forceinline__ int mn_argmin3(const double (&t)[3]) { int k = 0; if (t[1] < t[k]) k = 1; if (t[2] < t[k]) k = 2; return k; }

// APF_agent.act (APF.py:17-78)
__device__ __forceinline__ int mn_apf_act(const float *__restrict__ o, const MnPlanTabs &T) {
    const double k_att = 50.0, k_rep = 500.0, mass = 500.0, d0 = 10.0, min_vel = 1.0;      // APF.py:7-12 (n = 2)
    const double vx = o[0], vy = o[1], gx = o[2], gy = o[3];
    const double d_goal = sqrt(gx * gx + gy * gy);
    double rx = 0.0, ry = 0.0;
    for (int i = 0; i < 11; ++i) {
        const double px = o[4 + 2 * i], py = o[5 + 2 * i];
        if (px == 0.0 && py == 0.0) continue;
        const double d = sqrt(px * px + py * py);
        const double inv = 1.0 / d - 1.0 / d0;
        const double mag1 = k_rep * inv * (d_goal * d_goal) / (d * d);                      // away from the obstacle (APF.py:37-39)
        const double mag2 = 1.0 * k_rep * (inv * inv) * d_goal;                             // towards the goal (APF.py:42-44): (n / 2) k_rep inv^2 d_goal^(n-1)
        rx += mag1 * (-px / d) + mag2 * (-gx / d_goal);
        ry += mag1 * (-py / d) + mag2 * (-gy / d_goal);
    }
    const double fx = k_att * gx + rx, fy = k_att * gy + ry;
    const double speed = sqrt(vx * vx + vy * vy);
    const bool moving = speed > 1e-03;
    const double v_angle = moving ? atan2(vy, vx) : 0.0;
    const int w_idx = mn_nearest3(T.w, mn_wrap_pi(atan2(fy, fx) - v_angle));
    const double dx = moving ? vx / speed : 1.0, dy = moving ? vy / speed : 0.0;
    const double a_proj = (fx / mass) * dx + (fy / mass) * dy;
    double a_eff[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) a_eff[k] = (speed < min_vel && T.a[k] <= 0.0) ? -INFINITY : T.a[k];      // APF.py:71-74: mandate acceleration when slow
    return mn_nearest3(a_eff, a_proj) * 3 + w_idx;
}

// BA_agent.act (BA.py:14-155)
__device__ __forceinline__ int mn_ba_act(const float *__restrict__ o, const MnPlanTabs &T) {
    const double PI = 3.141592653589793;
    const double follow_dist = 5.0, detect_angle = 2.0 * PI / 3.0, margin = 10.0 * PI / 180.0, min_vel = 1.0;      // BA.py:7-12
    const double vx = o[0], vy = o[1], gx = o[2], gy = o[3];
    int cnt = 0;
    double max_ang = -INFINITY, min_ang = INFINITY;
    double p0x = 0.0, p1x = 0.0;                       // x of the first two returns in beam order (the regression rows are [x, 1])
    double sxx = 0.0, sx = 0.0, sxy = 0.0, sy = 0.0;   // normal equations of y = k x + c over the returns
    for (int i = 0; i < 11; ++i) {
        const double px = o[4 + 2 * i], py = o[5 + 2 * i];
        if (px == 0.0 && py == 0.0) continue;
        const double ang = atan2(py, px);
        max_ang = fmax(max_ang, ang); min_ang = fmin(min_ang, ang);
        if (cnt == 0) p0x = px;
        if (cnt == 1) p1x = px;
        sxx += px * px; sx += px; sxy += px * py; sy += py;
        ++cnt;
    }
    const double speed = sqrt(vx * vx + vy * vy);
    const double g_angle = atan2(gy, gx);
    bool use_goal = cnt == 0;
    if (!use_goal) {      // obstacle span (BA.py:52-65)
        double hi = mn_wrap_pi(max_ang + margin), lo = mn_wrap_pi(min_ang - margin);
        if (hi >= 0.5 * detect_angle) hi = PI;
        if (lo <= -0.5 * detect_angle) lo = -PI;
        use_goal = (g_angle < lo) || (g_angle > hi);
    }
    if (use_goal) {       // move_to_goal (BA.py:73-84)
        const bool still = speed < 1e-03;
        const double v_angle = atan2(still ? 0.0 : vy, still ? 1.0 : vx);
        return mn_argmax3(T.a) * 3 + mn_nearest3(T.w, mn_wrap_pi(g_angle - v_angle));
    }
    // wall_follow (BA.py:86-155): wall tangent `dir` and distance `d` by the number of returns
    double dirx, diry, d;
    if (cnt == 1) {             // d = |(x, 1)|, dir = R(90 deg) (x, 1)   (sic: the reference rotates the regression row [x, 1])
        d = sqrt(p0x * p0x + 1.0);
        dirx = -1.0; diry = p0x;
    } else if (cnt == 2) {      // rows [x0, 1], [x1, 1]: dir = row1 - row0 = (x1 - x0, 0), v_1 = (x0, 1)
        dirx = p1x - p0x; diry = 0.0;
        d = fabs(p0x * diry - 1.0 * dirx) / sqrt(dirx * dirx + diry * diry);
    } else {                    // least squares; the wall is "vertical" when A^T A is (nearly) singular
        const double sn = (double)cnt;
        const double tr = sxx + sn, det = sxx * sn - sx * sx;
        const double disc = sqrt(fmax(tr * tr / 4.0 - det, 0.0));
        const double s0 = tr / 2.0 + disc, s1 = tr / 2.0 - disc;      // singular values of the PSD 2 x 2 matrix A^T A
        if (s1 < 1e-03 * s0) {
            dirx = 0.0; diry = 1.0;
            d = fabs(sx / sn);
        } else {
            const double k_ = (sn * sxy - sx * sy) / det, c_ = (-sx * sxy + sxx * sy) / det;
            dirx = 1.0; diry = k_;
            d = fabs(1.0 * k_ - (k_ + c_) * 1.0) / sqrt(1.0 + k_ * k_);      // v_1 = (1, k + c)
        }
    }
    if (dirx * vx + diry * vy < 0.0) { dirx = -dirx; diry = -diry; }
    const double diff = mn_wrap_pi(atan2(diry, dirx) - atan2(vy, vx));
    const int w_idx = d < follow_dist ? (diff > 0.0 ? mn_argmax3(T.w) : mn_argmin3(T.w)) : mn_nearest3(T.w, diff);
    int a_idx;
    if (speed < min_vel) {      // mandate acceleration: the smallest positive entry
        double a_pos[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) a_pos[k] = T.a[k] > 0.0 ? T.a[k] : INFINITY;
        a_idx = mn_argmin3(a_pos);
    } else {
        const double zero = 0.0;
        a_idx = mn_nearest3(T.a, zero);
    }
    return a_idx * 3 + w_idx;
}

__device__ __forceinline__ int mn_policy_act(int policy, const float *__restrict__ o, const MnPlanTabs &T) {
    return policy == MN_POLICY_APF ? mn_apf_act(o, T) : mn_ba_act(o, T);
}
