// Device helpers shared by the step, reset and rollout kernels (gfx950).
//
// Floating-point contract: every translation unit of the library is compiled with -ffp-contract=off and fused
// multiply-adds are written out (fma / fmaf / __builtin_elementwise_fma) where the arithmetic wants them.  World
// generation (mn_reset_body.h) uses none, so it rounds exactly like numpy's float64 arithmetic; the step arithmetic
// (mn_step_body.h, the helpers below) is the same instruction-level expression in every kernel it is inlined into --
// single step, step + replay append, multi-step rollout, any lanes-per-env mapping -- so all of them produce
// bit-identical results (the compiler is never free to pick a different contraction in a different instantiation).
#pragma once
#include "mn_internal.h"

// ---- arithmetic flavours -------------------------------------------------------------------
// M = double : IEEE sqrt/div, OCML sincos.  M = float : 1-ulp hardware v_sqrt_f32 / v_rcp_f32.
template <typename M>
struct MnMath;

template <>
struct MnMath<double> {
    static __device__ __forceinline__ double sqrt_(double v) { return sqrt(v); }
    // 1 / v for the current field (v = squared distance to a vortex core): v_rcp_f64 + two Newton steps, i.e. within an ulp of the IEEE
    // quotient in 5 instructions instead of the ~25 of a float64 division (40 of them per env and step).  v = 0 -> NaN here, which the
    // `f < cap ? f : cap` of mn_core_velocity turns into the cap exactly as it does the division's +inf.
    static __device__ __forceinline__ double rcp(double v) {
        double r = __builtin_amdgcn_rcp(v);
        r = fma(fma(-v, r, 1.0), r, r);
        return fma(fma(-v, r, 1.0), r, r);
    }
    static __device__ __forceinline__ double fma_(double a, double b, double c) { return fma(a, b, c); }
};

template <>
struct MnMath<float> {
    static __device__ __forceinline__ float sqrt_(float v) { return __builtin_amdgcn_sqrtf(v); }
    static __device__ __forceinline__ float rcp(float v) { return __builtin_amdgcn_rcpf(v); }
    static __device__ __forceinline__ float fma_(float a, float b, float c) { return fmaf(a, b, c); }
};

// One beam against one obstacle (robot.py:147-198 restated in ray-parametric form, SURVEY App. A
// S2), in the ROBOT frame: (mrx,mry) = R(theta)^T (obstacle centre - robot position), (bx,by) = unit
// beam direction in the robot frame -- the constant (cos rel, sin rel), or +-(sin theta, cos theta)
// when the beam is snapped to exactly vertical (robot.py:150-162).
// The geometry (t_c, perpendicular offset, h^2) is always float64: a grazing hit amplifies an error
// in the perpendicular offset by r/h, which float32 cannot hold to 1e-5; the 5 f64 FMAs per pair are
// cheaper than any float32 compensation.  Everything after h^2 runs in M.
__device__ __forceinline__ void mn_beam_geom(double mrx, double mry, double r2, double bx, double by, double &tc, double &h2) {
    tc = fma(bx, mrx, by * mry);
    const double perp = fma(mrx, by, -(mry * bx));
    h2 = fma(-perp, perp, r2);
}

// Per-beam scan state: `dist` = accepted range (valid when hit), `limit` = what the next candidate
// must beat: +inf before the first hit, the accepted range afterwards, -inf once the reference's
// `break` has fired (nothing can be accepted any more).
//   h2 < 0                             -> no real solution          (robot.py:156,175 `continue`):
//                                         sqrt gives NaN, every comparison below is false
//   nearer root t = t_c -/+ h          (robot.py:184 picks the root with the smaller |t|)
//   |t| > range or t < 0               -> `continue`               (robot.py:185,188)
//   already hit and t >= best          -> `break`: later obstacles are never examined (:192-195)
template <typename M>
struct MnBeam {
    M dist, limit;
    double atc, ah2;   // float64 geometry (t_c, h^2) of the ACCEPTED candidate: lets the caller re-derive the accepted
                       // range in float64 after a float32 scan has made the discrete choices (dead code when unused)
    __device__ __forceinline__ void init() { dist = M(0); limit = (M)INFINITY; atc = 0.0; ah2 = 0.0; }
    __device__ __forceinline__ bool hit() const { return limit != (M)INFINITY; }
    __device__ __forceinline__ void update(double tc64, double h264, M range) {
        const M tc = (M)tc64, h2 = (M)h264;
        const M h = MnMath<M>::sqrt_(h2);
        const M t = tc > M(0) ? tc - h : tc + h;
        const bool in_range = (t >= M(0)) && (t <= range);
        const bool acc = in_range && (t < limit);
        dist = acc ? t : dist;
        atc = acc ? tc64 : atc;
        ah2 = acc ? h264 : ah2;
        limit = in_range ? (acc ? t : -(M)INFINITY) : limit;
    }
    // Accepted range re-derived in float64 from the accepted candidate's float64 geometry: sqrt(h^2) by one Newton step
    // on the float32 root (relative error ~1e-14), same root choice as update() (robot.py:184).
    __device__ __forceinline__ double dist64() const {
        const float h0f = __builtin_amdgcn_sqrtf((float)ah2);
        double h = (double)h0f;
        if (h0f > 0.f) h = fma(fma(-h, h, ah2), 0.5 * (double)__builtin_amdgcn_rcpf(h0f), h);
        return (M)atc > M(0) ? atc - h : atc + h;
    }
};

// Rankine vortex contribution of one core at relative position (dx,dy) = core - point
// (marinenav_env.py:433-453,461-465).  tangent*speed = (-dy,dx)/d * Gamma/(2 pi d) outside the
// core and (-dy,dx)/d * Gamma d/(2 pi r^2) inside; signed Gamma carries the spin direction.
// The two branches of compute_speed (:461-465) meet at d = r, and Gamma/(2 pi d^2) <= Gamma/(2 pi r^2)
// exactly when d >= r, so the profile is min(1/(2 pi r^2), 1/(2 pi d^2)) -- one v_min instead of a
// compare + select, and d = 0 stays finite.  Returns the contribution (ux, uy) as two rounded products: the caller
// adds the cores up in a FIXED balanced tree (mn_step_body.h), which is what makes the result independent of how
// many lanes share an environment.
// IEEE_DIV: the quotient as an IEEE division instead of MnMath::rcp (float64: v_rcp_f64 + two Newton steps, within an ulp of it) -- the
// reset kernel's one evaluation per episode (first observation), which is not on the hot path.
template <typename M, bool IEEE_DIV = false>
__device__ __forceinline__ void mn_core_velocity(M dx, M dy, M gs, M inv_two_pi_r2, M inv_two_pi, M &ux, M &uy) {
    const M d2 = MnMath<M>::fma_(dx, dx, dy * dy);
    M f = IEEE_DIV ? inv_two_pi / d2 : inv_two_pi * MnMath<M>::rcp(d2);
    f = f < inv_two_pi_r2 ? f : inv_two_pi_r2;
    f *= gs;
    ux = -dy * f;
    uy = dx * f;
}
