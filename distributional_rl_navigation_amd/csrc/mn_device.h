// Device helpers shared by the step and reset kernels (gfx950).
#pragma once
#include "mn_internal.h"

// ---- arithmetic flavours -------------------------------------------------------------------
// M = double : IEEE sqrt/div, OCML sincos.  M = float : 1-ulp hardware v_sqrt_f32 / v_rcp_f32.
template <typename M>
struct MnMath;

template <>
struct MnMath<double> {
    static __device__ __forceinline__ double sqrt_(double v) { return sqrt(v); }
    static __device__ __forceinline__ double rcp(double v) { return 1.0 / v; }
    static __device__ __forceinline__ void sincos_(double a, double *s, double *c) { sincos(a, s, c); }
};

template <>
struct MnMath<float> {
    static __device__ __forceinline__ float sqrt_(float v) { return __builtin_amdgcn_sqrtf(v); }
    static __device__ __forceinline__ float rcp(float v) { return __builtin_amdgcn_rcpf(v); }
    static __device__ __forceinline__ void sincos_(float a, float *s, float *c) { sincosf(a, s, c); }
};

// One beam against one obstacle (robot.py:147-198 restated in ray-parametric form, SURVEY App. A
// S2).  m = obstacle centre - robot position, (dx,dy) = unit beam direction (snapped to exactly
// vertical by the caller when within 1e-3 rad of +-pi/2, robot.py:150-162).
//   h2 = r^2 - (m x d)^2 < 0          -> no real solution          (robot.py:156,175 `continue`)
//   nearer root t = t_c -/+ h          (robot.py:184 picks the root with the smaller |t|)
//   |t| > range or t < 0               -> `continue`               (robot.py:185,188)
//   already hit and t >= best          -> `break`: later obstacles are never examined (:192-195)
template <typename M>
__device__ __forceinline__ void mn_beam_obstacle(M mx, M my, M r, M dx, M dy, M range, bool &hit, M &dist, bool &stopped) {
    M tc = dx * mx + dy * my;
    M perp = mx * dy - my * dx;
    M h2 = r * r - perp * perp;
    M h = MnMath<M>::sqrt_(h2 > M(0) ? h2 : M(0));
    M t = tc > M(0) ? tc - h : tc + h;
    bool cand = (!stopped) && (h2 >= M(0)) && (t >= M(0)) && (t <= range);
    bool brk = cand && hit && (t >= dist);
    bool acc = cand && !brk;
    stopped = stopped || brk;
    dist = acc ? t : dist;
    hit = hit || acc;
}

// Rankine vortex contribution of one core at relative position (dx,dy) = core - point
// (marinenav_env.py:433-453,461-465).  tangent*speed = (-dy,dx)/d * Gamma/(2 pi d) outside the
// core and (-dy,dx)/d * Gamma d/(2 pi r^2) inside; signed Gamma carries the spin direction.
template <typename M>
__device__ __forceinline__ void mn_core_velocity(M dx, M dy, M gs, M r2, M inv_two_pi_r2, M inv_two_pi, M &vx, M &vy) {
    M d2 = dx * dx + dy * dy;
    M f = d2 <= r2 ? inv_two_pi_r2 : inv_two_pi * MnMath<M>::rcp(d2);
    f *= gs;
    vx -= dy * f;
    vy += dx * f;
}
