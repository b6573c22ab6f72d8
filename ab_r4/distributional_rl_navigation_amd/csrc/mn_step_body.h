// mn_step_body.h -- one environment step as a device object (gfx950), shared by the single-step kernel (mn_step.hip)
// and the multi-step rollout kernel (mn_rollout.hip).  Compiled with -ffp-contract=off; the fused multiply-adds are written
// out (see mn_device.h), so every kernel this is inlined into computes bit-identical results.
//
// Replaces, for a batch of environments, MarineNavEnv.step (marinenav_env.py:199-262):
//   N x [ get_velocity (:422-465) -> Robot.update_state (robot.py:102-123) ],
//   get_observation (:273-326) with Robot.sonar_reflection (robot.py:125-198),
//   reward + termination ladder (:220-257), counters (:259-260).
//
// Mapping: L lanes per environment, 64/L environments per wavefront.
//   * 65 536 envs are only 1024 wavefronts at one lane per env -- ONE wave per SIMD, nothing to hide the dependent chain
//     of a step behind.  With L lanes per env the independent parts of a step are spread over the lane group -- the 8
//     vortex cores (8/L per lane, summed with DPP quad_perm / row_half_mirror adds, no LDS) and the 11 sonar beams
//     (ceil(11/L) per lane) -- while the short sequential part (float64 pose integration) is replicated in every lane
//     of the group; all lanes of a group hold bit-identical poses because the DPP adds are commutative pairs.
//   * Everything a lane needs lives in registers (its cores, all 10 obstacles); world tables are SoA [row][env], so a
//     wave's load of row k touches 64/L consecutive envs (the L lanes of a group read the same address).
//   * Heading: ONE float64 sincos per step; the sub-steps advance (cos, sin) by the constant rotation of w*dt.
//   * Observations leave as float2 stores into a row-major [env][26] tile.
// MnLane keeps an env's pose, counters and tables in registers between load() and store(), so the rollout kernel can run
// many steps without touching HBM for anything but its outputs.

// Developer ablation (profiling only): compiled in ONLY with -DMN_ABLATION, into libmarinenav_hip_ablation.so
// (make ablation).  The shipped library has no switch that removes work from the kernel.
#pragma once
#include "mn_device.h"

#ifdef MN_ABLATION
#define MN_SKIP(bit) ((P.debug_skip & (bit)) != 0)
#else
#define MN_SKIP(bit) false
#endif
// Phase stamps of ONE wavefront (workgroup 0), ablation build only and only in a translation unit that defines MN_PHASE_VAR
// (mn_rollout.hip: scripts/rollout_phase_timing.py): elapsed s_memtime ticks per phase, accumulated over the steps of a launch.
#if defined(MN_ABLATION) && defined(MN_PHASE_VAR)
#define MN_TICK_BEGIN() unsigned long long mn_tick_prev = __builtin_amdgcn_s_memtime()
#define MN_TICK(k) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
        if (blockIdx.x == 0 && threadIdx.x == 0) MN_PHASE_VAR[k] += t_ - mn_tick_prev; mn_tick_prev = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define MN_TICK_BEGIN() do { } while (0)
#define MN_TICK(k) do { } while (0)
#endif

// Put inside a wave-uniform `if`: keeps hipcc from if-converting it (it would evaluate the guarded float64 square root on every
// step and select afterwards, which is exactly what the guard is there to avoid).
#define MN_REAL_BRANCH() asm volatile("" ::: "memory")

#ifndef MN_STEP_BLOCK
#define MN_STEP_BLOCK 64    // threads per workgroup (npad is a multiple of 256, so 64 / 128 / 256 all tile it); measured: same at 65 536 envs, 64 is 4 % faster at 1 M
#endif

// sum over the L lanes of a group; every lane ends with the bit-identical total
template <int L>
__device__ __forceinline__ float group_sum(float v) {
    if (L >= 2) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    if (L >= 4) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    if (L >= 8) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    // 16 lanes per env: lanes 8-15 of a group hold the SAME eight cores as lanes 0-7 (MnLane::load), so each half forms the identical sum
    // by the identical tree and no fourth stage exists -- the value is bit-identical to L <= 8 and the sub-step chain is no longer
    return v;
}

template <int CTRL>
__device__ __forceinline__ double dpp_get(double v) {      // the partner lane's value
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned int)lo);
}
template <int L, int CTRL>
__device__ __forceinline__ double dpp_add(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, true);
    return v + __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned int)lo);
}

template <int L>
__device__ __forceinline__ double group_sum(double v) {
    if (L >= 2) v = dpp_add<L, 0xB1>(v);
    if (L >= 4) v = dpp_add<L, 0x4E>(v);
    if (L >= 8) v = dpp_add<L, 0x141>(v);
    return v;      // (L = 16: see the float overload)
}

// Nearest obstacle centre of the env over its lane group: minimum of (d2, k) in lexicographic order -- the FIRST obstacle in
// generation order among equally near ones, as the sequential `d2 < best` scan of check_collision (marinenav_env.py:329-336) --
// with that obstacle's radius.  Every lane of the group ends up with the group's result.
template <int CTRL>
__device__ __forceinline__ void nearest_step(double &d2, double &r2, int &k) {
    const double od2 = dpp_get<CTRL>(d2), or2 = dpp_get<CTRL>(r2);
    const int ok = __builtin_amdgcn_update_dpp(0, k, CTRL, 0xF, 0xF, true);
    const bool take = (od2 < d2) || (od2 == d2 && ok < k);
    d2 = take ? od2 : d2; r2 = take ? or2 : r2; k = take ? ok : k;
}
template <int L>
__device__ __forceinline__ void group_nearest(double &d2, double &r2, int &k) {
    if (L >= 2) nearest_step<0xB1>(d2, r2, k);
    if (L >= 4) nearest_step<0x4E>(d2, r2, k);
    if (L >= 8) nearest_step<0x141>(d2, r2, k);
    if (L >= 16) nearest_step<0x140>(d2, r2, k);
}

// What one step hands back to its caller (all lanes of an env's group hold the same values).
struct MnStepOut {
    double reward;
    int done, info;
};

template <typename M, bool PARITY, int L>
struct MnLane {
    static constexpr int CPL = L <= MN_MAX_CORES ? MN_MAX_CORES / L : 1;   // vortex cores per lane (L = 16: lanes 8-15 of a group repeat lanes 0-7's)
    static constexpr int BPL = (MN_NUM_BEAMS + L - 1) / L;  // sonar beams per lane
    static constexpr int OPL = (MN_MAX_OBS + L - 1) / L;    // obstacles per lane: lane q holds obstacles q, q + L, q + 2 L, ...
    static_assert(MN_MAX_CORES % L == 0 || L == 16, "L must divide 8, or be 16");
    static_assert(MN_STEP_BLOCK == 64 && L <= 16, "a lane group lives inside one wavefront (and one DPP row); the work-list hand-off below relies on it");

    int e, q;              // environment, lane within the env's group
    bool active;
    // pose, counters, episode constants
    double x, y, theta, speed, gx, gy;
    M velx, vely;          // velocity of the last sub-step (what the observation reports, App. A K5)
    double dis_c;          // distance to the goal after the last step of this lane object (= dis_before of the next one: same pose, same
    bool dis_ok;           // goal, same expression), valid until load(): the rollout loop saves one float64 sqrt per step
    int ep_t, nc, no;
    long long tot_t;
    // this lane's vortex cores (generation order is irrelevant for a sum) and this lane's SHARE of the obstacles (round 3: the
    // group rotates them into the robot frame together and hands the sonar work-list over in LDS, in generation order), padded:
    // missing cores are far away with zero circulation, missing obstacles far away with r = 0
    double ccx[CPL], ccy[CPL];
    M cgs[CPL];
    double obx[OPL], oby[OPL], obr[OPL];

    // Every load is UNCONDITIONAL (rows beyond the placed count hold zeros), so all ~40-70 loads of a lane are in flight
    // together and the kernel pays one memory latency, not a counts -> tables dependent chain.
    __device__ __forceinline__ void load(const MnArrays &A, int env, int lane_in_group) {
        e = env; q = lane_in_group;
        active = e < A.n;
        dis_c = 0.0; dis_ok = false;
        const int np = A.npad;
        x = A.x[e]; y = A.y[e]; theta = A.theta[e]; speed = A.speed[e];
        velx = (M)A.vx[e]; vely = (M)A.vy[e];
        gx = A.goal_x[e]; gy = A.goal_y[e];
        const int cnt = A.counts[e];
        ep_t = A.ep_t[e];
        tot_t = A.tot_t[e];
        double ccg[CPL];
        if (PARITY) {   // float64 master tables
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                const int k = (q & (MN_MAX_CORES - 1)) * CPL + j;
                ccx[j] = A.cx[k * np + e]; ccy[j] = A.cy[k * np + e]; ccg[j] = A.cg[k * np + e];
            }
#pragma unroll
            for (int j = 0; j < OPL; ++j) {
                const int k = min(q + L * j, MN_MAX_OBS - 1);
                obx[j] = A.ox[k * np + e]; oby[j] = A.oy[k * np + e]; obr[j] = A.orad[k * np + e];
            }
        } else {        // compact tables: int32 fixed-point positions (2^-24 m), float32 Gamma / radius
            int qx[CPL], qy[CPL], px[OPL], py[OPL];
            float qg[CPL], pr[OPL];
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                const int k = (q & (MN_MAX_CORES - 1)) * CPL + j;
                qx[j] = A.qcx[k * np + e]; qy[j] = A.qcy[k * np + e]; qg[j] = A.qcg[k * np + e];
            }
#pragma unroll
            for (int j = 0; j < OPL; ++j) {
                const int k = min(q + L * j, MN_MAX_OBS - 1);
                px[j] = A.qox[k * np + e]; py[j] = A.qoy[k * np + e]; pr[j] = A.qor[k * np + e];
            }
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                ccx[j] = (double)qx[j] * MN_FIX_INV; ccy[j] = (double)qy[j] * MN_FIX_INV; ccg[j] = (double)qg[j];
            }
#pragma unroll
            for (int j = 0; j < OPL; ++j) {
                obx[j] = (double)px[j] * MN_FIX_INV; oby[j] = (double)py[j] * MN_FIX_INV; obr[j] = (double)pr[j];
            }
        }
        nc = cnt & 0xff; no = (cnt >> 8) & 0xff;
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            const bool v = (q & (MN_MAX_CORES - 1)) * CPL + j < nc;
            ccx[j] = v ? ccx[j] : 1.0e6;   // padding: far away, zero circulation
            ccy[j] = v ? ccy[j] : 1.0e6;
            cgs[j] = v ? (M)ccg[j] : M(0);
        }
#pragma unroll
        for (int j = 0; j < OPL; ++j) {
            const bool v = q + L * j < no; // padding: far away, r = 0 -> never relevant, never the nearest
            obx[j] = v ? obx[j] : 1.0e6;
            oby[j] = v ? oby[j] : 1.0e6;
            obr[j] = v ? obr[j] : 0.0;
        }
    }

    // pose + counters back to the handle's arrays (lane 0 of the group)
    __device__ __forceinline__ void store(const MnArrays &A) const {
        if (active && q == 0) {
            A.x[e] = x; A.y[e] = y; A.theta[e] = theta; A.speed[e] = speed;
            A.vx[e] = (double)velx; A.vy[e] = (double)vely;
            A.ep_t[e] = ep_t;
            A.tot_t[e] = tot_t;
        }
    }

    // One MarineNavEnv.step.  Writes the observation row (terminal observation for a finished env) to obs_row
    // ([26] floats of THIS env; float64 copy to obs_row64 in parity precision) and, with APPEND, the transition to the
    // replay ring R (prev_head / prev_beam = this lane's share of obs_t, loaded by the caller).
    // obs_row_b: optional second destination of the same row (the rollout kernel's trace), or nullptr.
    template <bool APPEND>
    __device__ __forceinline__ MnStepOut step(const MnArrays &A, const MnDev &P, int action_raw, float *__restrict__ obs_row,
                                              double *__restrict__ obs_row64, const MnRing &R, const float2 *prev_head,
                                              const float2 *prev_beam, float *__restrict__ obs_row_b = nullptr) {
        MN_TICK_BEGIN();
        int action = action_raw < 0 ? 0 : (action_raw > 8 ? 8 : action_raw);

        // marinenav_env.py:205 dis_before
        const double dbx = gx - x, dby = gy - y;
        double dis_before = dis_c;
        if (__any(!dis_ok)) {      // (bitwise the cached value where that one is valid)
            MN_REAL_BRANCH();
            dis_before = sqrt(fma(dbx, dbx, dby * dby));
        }

        // robot.py:55-56: actions[i] = (a[i // 3], w[i % 3])
        const int ai = action / 3, wi = action - 3 * ai;
        const double acc = ai == 0 ? P.a[0] : (ai == 1 ? P.a[1] : P.a[2]);
        const double wdt = (wi == 0 ? P.w[0] : (wi == 1 ? P.w[1] : P.w[2])) * P.dt;
        const double rot_c = wi == 0 ? P.rot_c[0] : (wi == 1 ? P.rot_c[1] : P.rot_c[2]);   // cos(w*dt)
        const double rot_s = wi == 0 ? P.rot_s[0] : (wi == 1 ? P.rot_s[1] : P.rot_s[2]);   // sin(w*dt)
        const double dt = P.dt, two_pi = P.two_pi;

        const double inv_two_pi_r2 = 1.0 / P.two_pi_r_r;
        const double inv_two_pi = 1.0 / P.two_pi;

        // heading: wrap once on entry (state may have been set from outside), one sincos per step
        while (theta < 0.0) theta += two_pi;
        while (theta >= two_pi) theta -= two_pi;
        double sn = 0.0, cs = 1.0;
        if (!MN_SKIP(4)) sincos(theta, &sn, &cs);

        MN_TICK(0);      // action decode, distance, sincos
        // ---- N kinematic sub-steps (marinenav_env.py:208-212) -------------------------------------
        // The current is the superposition over ALL cores.  Every core's contribution is formed as two
        // rounded products and the eight of them are added in ONE fixed balanced tree -- ((c0+c1)+(c2+c3))+((c4+c5)+(c6+c7))
        // in generation order: the in-lane part over this lane's contiguous block of cores, the rest by the DPP stages --
        // so the sum, and with it every output of the step, is bit-identical for every lanes-per-env mapping L.
        const int nsub = MN_SKIP(1) ? 0 : P.N;
        if constexpr (!PARITY) {
            // Mixed precision: the core positions RELATIVE to the robot are formed once in float64, then
            // tracked in float32 (d -= v*dt): the rounding of a relative position is relative to the
            // DISTANCE to that core, which is what the 1/d field is sensitive to (a far core's 4e-6 m costs
            // 1e-8 m/s; a core 0.5 m away is tracked to 3e-8 m).  The absolute pose still integrates in
            // float64 below.  Cores are processed two at a time on float2 (v_pk_mul / v_pk_fma); a lane with a
            // single core (L = 8) runs the same operations on scalars.
            typedef float f2 __attribute__((ext_vector_type(2)));
            constexpr int NP = CPL >= 2 ? CPL / 2 : 1;
            const float dtf = (float)dt, i2p = (float)inv_two_pi, i2pr = (float)inv_two_pi_r2;
            f2 rdx[NP], rdy[NP], gsv[NP];
#pragma unroll
            for (int p_ = 0; p_ < NP; ++p_) {
                if constexpr (CPL >= 2) {
                    rdx[p_] = (f2){(float)(ccx[2 * p_] - x), (float)(ccx[2 * p_ + 1] - x)};
                    rdy[p_] = (f2){(float)(ccy[2 * p_] - y), (float)(ccy[2 * p_ + 1] - y)};
                    gsv[p_] = (f2){(float)cgs[2 * p_], (float)cgs[2 * p_ + 1]};
                } else {      // one real core in .x; .y is a zero-circulation dummy far away (contributes exactly +0)
                    rdx[p_] = (f2){(float)(ccx[0] - x), 1.0e6f};
                    rdy[p_] = (f2){(float)(ccy[0] - y), 1.0e6f};
                    gsv[p_] = (f2){(float)cgs[0], 0.f};
                }
            }
            for (int s = 0; s < nsub; ++s) {
                float hx[NP], hy[NP];      // per pair: u_2p + u_2p+1
#pragma unroll
                for (int p_ = 0; p_ < NP; ++p_) {
                    const f2 d2 = __builtin_elementwise_fma(rdx[p_], rdx[p_], rdy[p_] * rdy[p_]);
                    f2 f = (f2){__builtin_amdgcn_rcpf(d2.x), __builtin_amdgcn_rcpf(d2.y)} * i2p;
                    f.x = f.x < i2pr ? f.x : i2pr;
                    f.y = f.y < i2pr ? f.y : i2pr;
                    f *= gsv[p_];
                    const f2 ux = -rdy[p_] * f, uy = rdx[p_] * f;
                    hx[p_] = CPL >= 2 ? ux.x + ux.y : ux.x;      // (a dummy's +0 would not change the sum; skipped anyway)
                    hy[p_] = CPL >= 2 ? uy.x + uy.y : uy.x;
                }
                float sx_, sy_;
                if constexpr (NP == 4) { sx_ = (hx[0] + hx[1]) + (hx[2] + hx[3]); sy_ = (hy[0] + hy[1]) + (hy[2] + hy[3]); }
                else if constexpr (NP == 2) { sx_ = hx[0] + hx[1]; sy_ = hy[0] + hy[1]; }
                else { sx_ = hx[0]; sy_ = hy[0]; }
                const float cvx = group_sum<L>(sx_), cvy = group_sum<L>(sy_);
                velx = (float)(speed * cs) + cvx;
                vely = (float)(speed * sn) + cvy;
                x = fma((double)velx, dt, x);
                y = fma((double)vely, dt, y);
                const f2 nvx = (f2){-velx, -velx}, nvy = (f2){-vely, -vely}, dt2 = (f2){dtf, dtf};
#pragma unroll
                for (int p_ = 0; p_ < NP; ++p_) {
                    rdx[p_] = __builtin_elementwise_fma(nvx, dt2, rdx[p_]);
                    rdy[p_] = __builtin_elementwise_fma(nvy, dt2, rdy[p_]);
                }
                speed = fma(fma(-P.k_drag, speed, acc), dt, speed);      // robot.py:113
                speed = fmin(fmax(speed, 0.0), P.max_speed);           // robot.py:114 clip
                const double c2 = fma(cs, rot_c, -(sn * rot_s));
                sn = fma(sn, rot_c, cs * rot_s);
                cs = c2;
            }
            // heading (robot.py:117-123): N increments of w*dt, each wrapped into [0, 2pi).  The wrapped sum is
            // formed once here (it differs from the reference's step-by-step sum by rounding only, ~1e-15 rad)
            theta = fma((double)nsub, wdt, theta);
            while (theta < 0.0) theta += two_pi;
            while (theta >= two_pi) theta -= two_pi;
        } else {
            for (int s = 0; s < nsub; ++s) {
                // current at the pre-move position, float64 throughout
                double ux[CPL], uy[CPL];
#pragma unroll
                for (int j = 0; j < CPL; ++j)
                    mn_core_velocity<double>(ccx[j] - x, ccy[j] - y, cgs[j], inv_two_pi_r2, inv_two_pi, ux[j], uy[j]);
                double sx_, sy_;
                if constexpr (CPL == 8) {
                    sx_ = ((ux[0] + ux[1]) + (ux[2] + ux[3])) + ((ux[4] + ux[5]) + (ux[6] + ux[7]));
                    sy_ = ((uy[0] + uy[1]) + (uy[2] + uy[3])) + ((uy[4] + uy[5]) + (uy[6] + uy[7]));
                } else if constexpr (CPL == 4) { sx_ = (ux[0] + ux[1]) + (ux[2] + ux[3]); sy_ = (uy[0] + uy[1]) + (uy[2] + uy[3]); }
                else if constexpr (CPL == 2) { sx_ = ux[0] + ux[1]; sy_ = uy[0] + uy[1]; }
                else { sx_ = ux[0]; sy_ = uy[0]; }
                const double cvx = group_sum<L>(sx_), cvy = group_sum<L>(sy_);
                // robot.py:98-107: velocity = speed*(cos,sin) + current ; position += velocity*dt
                velx = speed * cs + cvx;
                vely = speed * sn + cvy;
                x = fma(velx, dt, x);
                y = fma(vely, dt, y);
                // marinenav_env.py:211-212: robot.trajectory gets one point per sub-step
                if (A.traj && q == 0 && active && s < A.traj_n) {
                    A.traj[((size_t)e * A.traj_n + s) * 2] = x;
                    A.traj[((size_t)e * A.traj_n + s) * 2 + 1] = y;
                }
                // robot.py:113-114: drag + clip
                speed = fma(fma(-P.k_drag, speed, acc), dt, speed);
                speed = speed < 0.0 ? 0.0 : (speed > P.max_speed ? P.max_speed : speed);
                // robot.py:117-123: heading + wrap to [0, 2pi); (cos, sin) advance by the constant rotation
                theta += wdt;
                theta = theta < 0.0 ? theta + two_pi : theta;
                theta = theta >= two_pi ? theta - two_pi : theta;
                const double c2 = fma(cs, rot_c, -(sn * rot_s));
                sn = fma(sn, rot_c, cs * rot_s);
                cs = c2;
            }
        }

        MN_TICK(1);      // N sub-steps (current field + integration)
        // marinenav_env.py:214 dis_after
        const double dax = gx - x, day = gy - y;
        const double dis_after = sqrt(fma(dax, dax, day * day));
        dis_c = dis_after; dis_ok = true;

        // ---- observation (marinenav_env.py:273-326) ------------------------------------------------
        // Obstacle centres in the robot frame, m_r = R(theta)^T (c - p) (|m_r| = |c - p|), and at the same
        // time the sonar work-list: only obstacles that can intersect the fan at all -- within range + r of
        // the robot and inside the +-60 degree wedge widened by r -- are appended, IN GENERATION ORDER, to a
        // lane-private LDS column.  A dropped obstacle can never produce a candidate, so it can neither be
        // hit nor trigger the reference's `break`; the scan over the list is therefore equivalent to the
        // scan over all obstacles (robot.py:147-198).  Typically 0-3 of the 10 obstacles survive, and the
        // beam loop runs to the longest list in the wavefront instead of 10.
        // Round 3: the lanes of an env's group SHARE this work -- lane q rotates obstacles q, q + L, ... (10 / L of them instead of
        // all 10: the float64 rotation was 29 % of a rollout step, profiles/r03_rollout_phase_timing.txt) and appends the relevant ones
        // to ONE list per env; an obstacle's slot is the number of relevant obstacles before it in generation order, counted from the
        // wavefront's ballots (lane q of iteration j owns obstacle q + L j, so a group's L ballot bits of iteration j are L
        // consecutive obstacles).  Values are computed by the same expressions whichever lane owns an obstacle, the list order is the
        // generation order: results do not depend on L, bit for bit (tests).
        constexpr int NGRP = MN_STEP_BLOCK / L;
        __shared__ double lst_x[MN_MAX_OBS][NGRP], lst_y[MN_MAX_OBS][NGRP];
        __shared__ M lst_r[MN_MAX_OBS][NGRP];     // radius: float32 is exact for the compact tables, float64 in parity mode
        const int tl = threadIdx.x / L;
        const int gshift = (threadIdx.x & 63) & ~(L - 1);           // first lane of this env's group in the wavefront
        int nrel = 0;
        double best = 1e300, best_r = 0.0;    // check_collision (:329-336): nearest-CENTRE obstacle only
        int best_k = 1 << 20;
        const double reach0 = P.sonar_range + 0.05;
        if (!MN_SKIP(8))
#pragma unroll
        for (int j = 0; j < OPL; ++j) {
            const int k = q + L * j;
            const double mx = obx[j] - x, my = oby[j] - y;
            const double rx_ = fma(cs, mx, sn * my), ry_ = fma(cs, my, -(sn * mx));
            const double d2 = fma(mx, mx, my * my);
            const bool in = k < no;
            const bool nearer = in && (d2 < best);      // within a lane k grows with j: first of equals wins
            best = nearer ? d2 : best;
            best_r = nearer ? obr[j] : best_r;
            best_k = nearer ? k : best_k;
            const double r = obr[j];
            const double reach = reach0 + r;
            const bool rel = in && (d2 <= reach * reach) &&
                             (!P.fan_filter || (fma(P.fan_sin, rx_, -(P.fan_cos * fabs(ry_))) >= -(r + 0.05)));
            const unsigned gbits = (unsigned)(__ballot(rel) >> gshift) & ((1u << L) - 1u);     // obstacles L j .. L j + L - 1 of this env
            const int pos = nrel + __popc(gbits & ((1u << q) - 1u));
            if (rel) { lst_x[pos][tl] = rx_; lst_y[pos][tl] = ry_; lst_r[pos][tl] = (M)r; }
            nrel += __popc(gbits);
        }
        group_nearest<L>(best, best_r, best_k);
        if (L > 1) __syncthreads();      // (one wavefront per workgroup) the list entries other lanes wrote are visible
        const M range = (M)P.sonar_range;
        const double half_pi = 0.5 * 3.141592653589793, three_half_pi = 3 * 3.141592653589793 / 2;
        M bxo[BPL], byo[BPL];   // this lane's beams, hit point in the robot frame
        double bdx[BPL], bdy[BPL];
        MnBeam<M> beam[BPL];
#pragma unroll
        for (int j = 0; j < BPL; ++j) {
            const int b = q + L * j;
            const int bb = b < MN_NUM_BEAMS ? b : MN_NUM_BEAMS - 1;
            const double angle = theta + P.beam_rel[bb];  // robot.py:134, not wrapped
            const bool up = fabs(angle - half_pi) < 1e-03;
            const bool down = fabs(angle - three_half_pi) < 1e-03;
            // beam direction in the robot frame: the constant (cos rel, sin rel); a snapped beam points
            // along world (0,+-1), i.e. R^T (0,+-1) = +-(sin theta, cos theta)
            bdx[j] = P.beam_cos[bb]; bdy[j] = P.beam_sin[bb];
            if (up || down) {
                const double sg = up ? 1.0 : -1.0;
                bdx[j] = sg * sn; bdy[j] = sg * cs;
            }
            beam[j].init();
        }
        MN_TICK(2);      // obstacle rotation, work-list, beam directions
        if (!MN_SKIP(2))
        for (int s_ = 0; __any(s_ < nrel); ++s_) {
            const bool v = s_ < nrel;
            const double ox_ = lst_x[s_][tl], oy_ = lst_y[s_][tl];
            const double rr = (double)lst_r[s_][tl];
            const double r2o = v ? rr * rr : -1.0;    // exhausted list: h^2 < 0 -> NaN -> never a candidate
#pragma unroll
            for (int j = 0; j < BPL; ++j) {
                double tc, h2;
                mn_beam_geom(ox_, oy_, r2o, bdx[j], bdy[j], tc, h2);
                beam[j].update(tc, h2, range);
            }
        }
#pragma unroll
        for (int j = 0; j < BPL; ++j) {
            const bool hit = beam[j].hit();
            // Mixed precision: the float32 scan above made the discrete choices (which obstacle, hit / miss, `break`); the
            // accepted range itself is re-derived in float64 from that candidate's float64 geometry, so the returned point
            // is as accurate as the pose it was cast from (north-star: 1e-5 absolute on float32 outputs).
            const double td = PARITY ? (double)beam[j].dist : beam[j].dist64();
            bxo[j] = hit ? (M)(td * bdx[j]) : M(0);  // misses are (0,0): marinenav_env.py:315-316
            byo[j] = hit ? (M)(td * bdy[j]) : M(0);
        }

        MN_TICK(3);      // sonar scan over the work-list + range re-derivation
        // ---- reward + termination ladder (marinenav_env.py:220-257) -------------------------------
        double reward = P.timestep_penalty;
        reward += dis_before - dis_after;
        // check_collision: sqrt(d2) <= r + robot_r.  sqrt is monotone and rounds within an ulp, so the comparison of the squares
        // decides it whenever d2 is not within 2^-48 (relative) of the squared threshold; only then is the square root taken
        // (wave-uniform branch, practically never) -- same truth value in every case, one float64 sqrt less per step.
        const double thr = best_r + P.robot_r, thr2 = thr * thr;
        bool near_enough = best <= thr2;
        if (__any(fabs(best - thr2) <= thr2 * 0x1p-48)) {
            MN_REAL_BRANCH();
            near_enough = sqrt(best) <= thr;
        }
        const bool collide = no > 0 && near_enough;
        const bool reach = dis_after <= P.goal_dis;  // check_reach_goal (:338-342)
        const bool out = (x < 0.0 || x > P.width) || (y < 0.0 || y > P.height);
        // the ladder as selects, lowest priority first (a chain of divergent branches costs more than its five selects)
        int info = reach ? MN_INFO_REACH_GOAL : MN_INFO_NORMAL;
        info = collide ? MN_INFO_COLLISION : info;
        info = ep_t >= P.max_episode_steps ? MN_INFO_TOO_LONG : info;
        info = (P.set_boundary && out) ? MN_INFO_OUT_OF_BOUNDARY : info;
        const int done = info != MN_INFO_NORMAL;
        reward = info == MN_INFO_COLLISION ? reward + P.collision_penalty : (info == MN_INFO_REACH_GOAL ? reward + P.goal_reward : reward);
        ep_t += 1;       // marinenav_env.py:259-260
        tot_t += 1;

        MN_TICK(4);      // reward, termination ladder
        // ---- observation row (+ replay transition) ---------------------------------------------------
        if (active) {
            // replay slot of this env's transition (FIFO ring; only the newest `cap` rows of a launch survive)
            long long slot = -1;
            if constexpr (APPEND) {
                const long long first = (long long)A.n > R.cap ? (long long)A.n - R.cap : 0;
                if (e >= first) { slot = R.ptr + (e - first); slot = slot >= R.cap ? slot - R.cap : slot; }
            }
            float2 *rs = APPEND && slot >= 0 ? reinterpret_cast<float2 *>(R.states + slot * MN_OBS_DIM) : nullptr;
            float2 *rn = APPEND && slot >= 0 ? reinterpret_cast<float2 *>(R.next_states + slot * MN_OBS_DIM) : nullptr;
            if (q == 0) {
                // R(theta)^T * velocity: lagged velocity, final heading (App. A K5); R(theta)^T (goal - p)
                const M c = (M)cs, s_ = (M)sn;
                const M o0 = MnMath<M>::fma_(c, velx, s_ * vely), o1 = MnMath<M>::fma_(c, vely, -(s_ * velx));
                const M o2 = (M)fma(cs, dax, sn * day), o3 = (M)fma(cs, day, -(sn * dax));
                *reinterpret_cast<float2 *>(obs_row) = make_float2((float)o0, (float)o1);
                *reinterpret_cast<float2 *>(obs_row + 2) = make_float2((float)o2, (float)o3);
                if (obs_row_b) {
                    *reinterpret_cast<float2 *>(obs_row_b) = make_float2((float)o0, (float)o1);
                    *reinterpret_cast<float2 *>(obs_row_b + 2) = make_float2((float)o2, (float)o3);
                }
                if constexpr (APPEND) {
                    if (rs) {
                        rs[0] = prev_head[0]; rs[1] = prev_head[1];
                        rn[0] = make_float2((float)o0, (float)o1); rn[1] = make_float2((float)o2, (float)o3);
                        R.actions[slot] = (int64_t)action_raw;       // as chosen (replay_buffer.py:50 stores the agent's action)
                        R.rewards[slot] = (float)reward;
                        R.dones[slot] = done ? 1.0f : 0.0f;
                    }
                }
                if (PARITY && obs_row64) {      // float64 copies for parity checks: only after mn_enable_obs64 (wave-uniform)
                    A.rew64[e] = reward;
                    obs_row64[0] = (double)o0; obs_row64[1] = (double)o1; obs_row64[2] = (double)o2; obs_row64[3] = (double)o3;
                }
            }
            if (!MN_SKIP(16))
#pragma unroll
            for (int j = 0; j < BPL; ++j) {
                const int b = q + L * j;
                if (b < MN_NUM_BEAMS) {
                    *reinterpret_cast<float2 *>(obs_row + 4 + 2 * b) = make_float2((float)bxo[j], (float)byo[j]);
                    if (obs_row_b) *reinterpret_cast<float2 *>(obs_row_b + 4 + 2 * b) = make_float2((float)bxo[j], (float)byo[j]);
                    if constexpr (APPEND) {
                        if (rs) { rs[2 + b] = prev_beam[j]; rn[2 + b] = make_float2((float)bxo[j], (float)byo[j]); }
                    }
                    if (PARITY && obs_row64) { obs_row64[4 + 2 * b] = (double)bxo[j]; obs_row64[5 + 2 * b] = (double)byo[j]; }
                }
            }
        }
        MN_TICK(5);      // observation row stores
        MnStepOut o;
        o.reward = reward; o.done = done; o.info = info;
        return o;
    }
};
