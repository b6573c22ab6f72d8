// iqn_train_host.h -- host side of csrc/iqn_train.hip (included at its end: the kernels live in that translation unit's anonymous namespace): workspace layout accessors, the
// launch plan (one / two / three / four launches per gradient step from the device's CU count), the launchers, the C-ABI entry points of include/marinenav_hip.h for the
// learner (mn_iqn_train_*, mn_iqn_sample) and the mailbox exchange object of a shared learner (mn_xchg_*).
#pragma once

#ifdef MN_TRAIN_PHASES
extern "C" int mn_iqn_train_debug_phases(unsigned long long *out_host) {   // [2][32]: target workgroup 0, first local workgroup
    return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_phase), sizeof(unsigned long long) * 64) == hipSuccess ? MN_OK : MN_ERR_HIP;
}
extern "C" int mn_iqn_train_debug_wgt(unsigned long long *out_host) {   // [1024][2]: start, end of every forward / backward workgroup
    return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_wgt), sizeof(unsigned long long) * 4096) == hipSuccess ? MN_OK : MN_ERR_HIP;
}
extern "C" int mn_iqn_train_debug_phases3(unsigned long long *out_host) {   // [first, middle, last reduction + Adam block][8]
    return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_phase3), sizeof(unsigned long long) * 24) == hipSuccess ? MN_OK : MN_ERR_HIP;
}
extern "C" int mn_iqn_train_debug_phases2(unsigned long long *out_host) {   // [reduce, adam][block 0, middle block][8]
    return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_phase2), sizeof(unsigned long long) * 32) == hipSuccess ? MN_OK : MN_ERR_HIP;
}
#endif

extern "C" int mn_iqn_train_set_mode(int32_t mode) {
    if (mode != MODE_TWO_ROLES && mode != MODE_LOCAL_ONLY) return MN_ERR_INVALID;
    g_train_mode = mode;
    return MN_OK;
}

extern "C" int mn_iqn_sample(int64_t ring_size, int32_t batch, uint64_t *rng_state_dev, int64_t *idx_out, float *taus_out,
                             int32_t n_taus_total, void *stream) {
    if (!rng_state_dev || !idx_out || (n_taus_total > 0 && !taus_out) || n_taus_total < 0) return MN_ERR_INVALID;
    if (batch <= 0 || batch > MAX_BATCH || ring_size < batch || ring_size > 0x7fffffff) return MN_ERR_INVALID;
    hipLaunchKernelGGL(iqn_sample_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, ring_size, batch, rng_state_dev, idx_out,
                       taus_out, n_taus_total);
    return hipGetLastError() == hipSuccess ? MN_OK : MN_ERR_HIP;
}

// float index, inside the workspace, of the u32 count of local workgroups of XCD-grouped one-launch steps that did not run on XCD (block index % 8)
// as the first workgroup of its group (block index % 8) (0 on every launch observed: the dispatcher deals workgroups out round-robin; such a workgroup's row
// takes the slow way through memory)
extern "C" int64_t mn_iqn_train_workspace_misplaced_word(int32_t batch) {
    if (batch <= 0 || batch > MAX_BATCH || batch % BE) return -1;
    return ws_epoch(batch / BE) + 9;
}

// float index of the u32 status word: reduction + Adam blocks of which a bounded wait ran out since mn_iqn_train_workspace_init (0 in a healthy run; such a block
// skipped its update -- the caller must treat a non-zero count as an error)
extern "C" int64_t mn_iqn_train_workspace_status_word(int32_t batch) {
    if (batch <= 0 || batch > MAX_BATCH || batch % BE) return -1;
    return ws_epoch(batch / BE) + 12;
}

extern "C" int64_t mn_iqn_train_workspace_floats(int32_t batch) {
    if (batch <= 0 || batch % BE) return -1;
    return ws_total(batch / BE);
}

extern "C" int mn_iqn_train_workspace_init(float *workspace, int32_t batch, void *stream) {
    if (!workspace || batch <= 0 || batch % BE) return MN_ERR_INVALID;
    const int n_part = batch / BE;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(workspace, 0, (size_t)ws_total(n_part) * sizeof(float), s) != hipSuccess ||
        hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(workspace + ws_epoch(n_part) + 8), (int)WS_MAGIC, 1, s) != hipSuccess)
        return MN_ERR_HIP;
    return MN_OK;
}

struct mn_xchg {
    int rank = 0, world = 1, device = -1;
    gu64 *own = nullptr;                       // [2][P_PAD] granules, this rank's reduced gradient of the last two steps
    const gu64 *peer[XCHG_MAX_RANKS] = {};     // own + IPC-mapped peers, by rank
    bool opened[XCHG_MAX_RANKS] = {};
    unsigned *status = nullptr;                // device word: number of granule groups that timed out
    int kind = 0;                              // how `own` was allocated: 2 = uncached, 1 = fine-grained, 0 = plain hipMalloc (see mn_xchg_create)
    uint64_t bound = 200000000ull;             // ticks of the 100 MHz counter a gather waits for a peer's granules (mn_xchg_set_timeout_ms)
    XchgArgs *dev_args = nullptr;              // the record above in device memory, as the fused kernels read it (xchg_sync)
    char err[384] = "";                        // mn_xchg_last_error
};

struct AdamArgs {      // non-null: the step also performs clip + Adam (mn_iqn_train_step*)
    float *params, *exp_avg, *exp_avg_sq;
    int32_t *step_dev;
    double lr, beta1, beta2, eps, max_norm;
    const mn_xchg *x;      // shared learner: the one-shot exchange inside the same launch
    float grad_scale;
};

// ---- what this device can hold at once ------------------------------------------------------------------------------------------------------
// The fused forms wait, inside a launch, for other workgroups of the SAME launch: the local workgroups of the one-launch step for the rows of their XCD group, the
// reduction + Adam blocks for each other's norm partials.  That is only safe while those workgroups are resident together, which depends on the device (a
// partitioned MI300-class part in CPX mode shows 32 CUs, not 256) -- so the launch plan is made from the device's CU count and the kernels' occupancy, not
// from the constant 256 (round 5), and falls back to the forms without in-launch waits between workgroups of one role: three launches (four with
// the shared learner's exchange).  mn_iqn_train_set_cu_limit: plan as if the device had fewer CUs (tests; two ranks sharing one GPU plan for half of it each).
struct DevInfo { bool known; int n_cu, ra_per_cu, rax_per_cu; };
static int g_cu_limit = 0;
static int dev_info(DevInfo *out) {
    static std::mutex mu;
    static DevInfo info[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return MN_ERR_HIP;
    std::lock_guard<std::mutex> lock(mu);
    DevInfo &d = info[dev];
    if (!d.known) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return MN_ERR_HIP;
        // the forward / backward kernel's dynamic LDS (97 KB) is above the default limit: raised once per device
        const void *kernels[] = {reinterpret_cast<const void *>(iqn_train_fwdbwd<false, false>), reinterpret_cast<const void *>(iqn_train_fwdbwd<false, true>),
                                 reinterpret_cast<const void *>(iqn_train_fwdbwd<true, true>)};
        for (const void *kf : kernels)
            if (hipFuncSetAttribute(kf, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) return MN_ERR_HIP;
        int a = 0, b = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, reinterpret_cast<const void *>(iqn_grad_reduce_adam), RA_BT, 0) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, reinterpret_cast<const void *>(iqn_grad_reduce_adam_xchg), RA_BT, 0) != hipSuccess)
            return MN_ERR_HIP;
        d.n_cu = prop.multiProcessorCount;
        d.ra_per_cu = std::max(1, a);
        d.rax_per_cu = std::max(1, b);
        d.known = true;
    }
    *out = d;
    if (g_cu_limit > 0) out->n_cu = std::min(out->n_cu, g_cu_limit);
    return MN_OK;
}

struct LaunchPlan {
    int mode;        // MODE_TWO_ROLES / MODE_LOCAL_ONLY
    int n_fwd;       // forward / backward workgroups
    int launches;    // per gradient step -- 1: the fused step (reduction + Adam ride in the forward / backward launch);
                     // 2: + iqn_grad_reduce_adam[_xchg]; 3: + iqn_grad_reduce + iqn_adam; 4: + iqn_grad_reduce, iqn_grad_gather, iqn_adam (shared learner on a device
                     // too small for the fused launches)
    int n_extra;     // fused: workgroups behind the forward / backward ones that only run reduction + Adam blocks
};
static LaunchPlan plan_launch(const DevInfo &d, int batch, int flags, bool adam, bool xchg) {
    LaunchPlan p = {};
    const int n_part = batch / BE;
    // two-role launches need every local workgroup's target workgroup dispatched no later than itself (target workgroups have the lower block indices) and
    // a CU for every workgroup; beyond that a local workgroup computes its own targets
    p.mode = (g_train_mode == MODE_TWO_ROLES && 2 * n_part <= d.n_cu) ? MODE_TWO_ROLES : MODE_LOCAL_ONLY;
    p.n_fwd = p.mode == MODE_TWO_ROLES ? 2 * n_part : n_part;
    if (!adam) { p.launches = 2; return p; }      // (mn_iqn_train_grad*: forward / backward + iqn_grad_reduce; Adam is the caller's next call)
    if ((flags & MN_TRAIN_ONE_LAUNCH) && p.mode == MODE_TWO_ROLES && n_part % 8 == 0) {
        // The fused step: the target workgroups are the reduction + Adam blocks (two virtual blocks each; small batches add blocks that do nothing else).  Its
        // workgroups wait for each other -- local ones for their XCD group's rows, reduction + Adam blocks for each other's norm partials
        // -- so ALL of them must be resident together: one CU each (97 KB of LDS, 226 registers).
        p.n_extra = std::max(0, (N_ADAM + 1) / 2 - n_part);
        if (p.n_fwd + p.n_extra <= d.n_cu) { p.launches = 1; return p; }
    }
    p.n_extra = 0;
    if (N_ADAM <= d.n_cu * (xchg ? d.rax_per_cu : d.ra_per_cu)) { p.launches = 2; return p; }      // all 140 blocks of the fused reduction + Adam launch resident
    p.launches = xchg ? 4 : 3;
    return p;
}

static int xchg_sync(mn_xchg *x) {      // (create / import / set_timeout: never on the step path)
    XchgArgs xa = {};
    for (int r = 0; r < x->world; ++r) xa.peers.mb[r] = x->peer[r];
    xa.own = x->own;
    xa.world = x->world;
    xa.status = x->status;
    xa.bound = x->bound;
    return hipMemcpy(x->dev_args, &xa, sizeof(xa), hipMemcpyHostToDevice) == hipSuccess ? MN_OK : MN_ERR_HIP;
}

static int launch_adam(float *params, float *grad, float *exp_avg, float *exp_avg_sq, int32_t *step_dev, float *workspace, int n_part, double lr, double beta1,
                       double beta2, double eps, double max_norm, float grad_scale, hipStream_t s) {
    unsigned *ticket = reinterpret_cast<unsigned *>(workspace + ws_epoch(n_part) + 2);
    hipLaunchKernelGGL(iqn_adam, dim3(N_ADAM), dim3(256), 0, s, params, grad, exp_avg, exp_avg_sq, (const float *)(workspace + ws_sq(n_part)), step_dev, ticket, lr, beta1,
                       beta2, eps, max_norm, grad_scale);
    return hipGetLastError() == hipSuccess ? MN_OK : MN_ERR_HIP;
}

static int launch_gather(const mn_xchg *x, float *grad, float *workspace, int n_part, float grad_scale, hipStream_t s) {
    XchgPeers peers;
    for (int r = 0; r < XCHG_MAX_RANKS; ++r) peers.mb[r] = nullptr;
    for (int r = 0; r < x->world; ++r) {
        peers.mb[r] = x->peer[r];
        if (!peers.mb[r]) return MN_ERR_INVALID;      // a peer's mailbox was never imported
    }
    hipLaunchKernelGGL(iqn_grad_gather, dim3(N_RED), dim3(RED_COLS), 0, s, peers, x->world, (const float *)workspace, n_part, grad, workspace + ws_sq(n_part), grad_scale,
                       x->status, x->bound);
    return hipGetLastError() == hipSuccess ? MN_OK : MN_ERR_HIP;
}

static int launch_grad(const float *ring_states, const float *ring_next_states, const int64_t *ring_actions,
                       const float *ring_rewards, const float *ring_dones, const int64_t *idx_dev, const float *taus_target_dev,
                       const float *taus_local_dev, const float *params_local, const float *params_target, float *workspace,
                       float *grad_out, float *loss_out, int32_t batch, int32_t num_taus, float gamma, uint64_t *rng_state_dev,
                       int64_t ring_size, int64_t *idx_out, float *taus_out, int32_t flags, void *stream, const AdamArgs *adam = nullptr) {
    if (!ring_states || !ring_next_states || !ring_actions || !ring_rewards || !ring_dones || !params_local || !params_target ||
        !workspace || !grad_out || !loss_out)
        return MN_ERR_INVALID;
    if (rng_state_dev ? (ring_size < batch || ring_size > 0x7fffffff) : (!idx_dev || !taus_target_dev || !taus_local_dev))
        return MN_ERR_INVALID;
    if (batch <= 0 || batch > MAX_BATCH || batch % BE || num_taus != NQ) return MN_ERR_INVALID;
    DevInfo dev;
    if (int rc = dev_info(&dev)) return rc;
    const int n_part = batch / BE;
    const mn_xchg *x = adam ? adam->x : nullptr;
    if (x)
        for (int r = 0; r < x->world; ++r)
            if (!x->peer[r]) return MN_ERR_INVALID;      // a peer's mailbox was never imported
    const LaunchPlan plan = plan_launch(dev, batch, flags, adam != nullptr, x != nullptr);
    const BatchArgs ba = {ring_states, ring_next_states, ring_rewards, ring_dones, ring_actions, idx_dev, taus_target_dev,
                          taus_local_dev, (const uint64_t *)rng_state_dev, ring_size, idx_out, taus_out};
    hipStream_t s = (hipStream_t)stream;
    const int use_staged = rng_state_dev && (flags & MN_TRAIN_USE_STAGED) ? 1 : 0;
    const int prefetch_next = rng_state_dev && (flags & MN_TRAIN_STAGE_NEXT) ? 1 : 0;
    if (plan.launches == 1) {      // the fused step
        const StepTail tail = {N_ADAM, plan.n_extra, (flags >> 4) & 3, prefetch_next, grad_out, loss_out, adam->params, adam->exp_avg, adam->exp_avg_sq, adam->step_dev,
                               rng_state_dev, adam->lr, adam->beta1, adam->beta2, adam->eps, adam->max_norm, x ? x->dev_args : nullptr, adam->grad_scale};
        const TrainArgs ta = {ba, params_local, params_target, workspace, batch, gamma, plan.mode, use_staged, tail};
        const dim3 grid(plan.n_fwd + plan.n_extra);
        if (x) hipLaunchKernelGGL((iqn_train_fwdbwd<true, true>), grid, dim3(THREADS), LDS_BYTES, s, ta);
        else hipLaunchKernelGGL((iqn_train_fwdbwd<false, true>), grid, dim3(THREADS), LDS_BYTES, s, ta);
        return hipGetLastError() == hipSuccess ? MN_OK : MN_ERR_HIP;
    }
    const StepTail no_tail = {};
    hipLaunchKernelGGL((iqn_train_fwdbwd<false, false>), dim3(plan.n_fwd), dim3(THREADS), LDS_BYTES, s,
                       TrainArgs{ba, params_local, params_target, workspace, batch, gamma, plan.mode, use_staged, no_tail});
    if (plan.launches == 2 && x)
        hipLaunchKernelGGL(iqn_grad_reduce_adam_xchg, dim3(N_ADAM), dim3(RA_BT), 0, s, workspace, n_part, grad_out, loss_out, rng_state_dev, ba, prefetch_next,
                           adam->params, adam->exp_avg, adam->exp_avg_sq, adam->step_dev, adam->lr, adam->beta1, adam->beta2, adam->eps, adam->max_norm,
                           (const XchgArgs *)x->dev_args, adam->grad_scale);
    else if (plan.launches == 2 && adam)
        hipLaunchKernelGGL(iqn_grad_reduce_adam, dim3(N_ADAM), dim3(RA_BT), 0, s, workspace, n_part, grad_out, loss_out, rng_state_dev, ba, prefetch_next,
                           adam->params, adam->exp_avg, adam->exp_avg_sq, adam->step_dev, adam->lr, adam->beta1, adam->beta2, adam->eps, adam->max_norm);
    else      // the stand-alone reduction (mn_iqn_train_grad*; or the first of the launches a small device takes instead of the fused one -- a shared learner's publishes into its mailbox)
        hipLaunchKernelGGL(iqn_grad_reduce, dim3(N_RED), dim3(RED_COLS * RED_SEG), 0, s, workspace, n_part, grad_out, loss_out,
                           rng_state_dev, ba, prefetch_next, (uint64_t)(uintptr_t)(adam && x ? x->own : nullptr));
    if (hipGetLastError() != hipSuccess) return MN_ERR_HIP;
    if (plan.launches == 4)
        if (int rc = launch_gather(x, grad_out, workspace, n_part, adam->grad_scale, s)) return rc;
    if (plan.launches >= 3)      // (the norm partials are the reduction's, or the gather's for grad_scale x the sum)
        return launch_adam(adam->params, grad_out, adam->exp_avg, adam->exp_avg_sq, adam->step_dev, workspace, n_part, adam->lr, adam->beta1, adam->beta2, adam->eps,
                           adam->max_norm, x ? adam->grad_scale : 1.0f, s);
    return MN_OK;
}

// Test / multi-tenant hook: plan launches as if the device had at most `n_cu` CUs (0 = what the device reports).  Two ranks that share ONE GPU (tests) set half of it each.
extern "C" int mn_iqn_train_set_cu_limit(int32_t n_cu) {
    if (n_cu < 0) return MN_ERR_INVALID;
    g_cu_limit = n_cu;
    return MN_OK;
}

// Launches one gradient step takes on the current device: 1 (reduction + Adam ride in the forward / backward launch), 2, 3 (a device on which the 140 blocks of
// the fused reduction + Adam launch are not resident together), 4 (the same for a shared learner's exchange); < 0: error.  `exchange` != 0: a shared learner's step.
extern "C" int mn_iqn_train_plan(int32_t batch, int32_t flags, int32_t exchange) {
    if (batch <= 0 || batch > MAX_BATCH || batch % BE) return MN_ERR_INVALID;      // (error codes are negative)
    DevInfo dev;
    if (int rc = dev_info(&dev)) return rc;
    return plan_launch(dev, batch, flags, true, exchange != 0).launches;
}

// One gradient step of a single learner: forward / backward, reduction, clip + Adam -- as one launch (MN_TRAIN_ONE_LAUNCH), two (iqn_grad_reduce_adam) or, on a device that
// cannot hold the fused launches' workgroups together, three.  Either the batch is drawn in the launch (rng_state_dev != NULL: the arguments of
// mn_iqn_train_grad_sampled) or given (idx_dev / taus_*_dev: those of mn_iqn_train_grad).  params_local is updated in place; every form is bit-identical to
// mn_iqn_train_grad* followed by mn_iqn_train_adam(grad_scale = 1).
extern "C" int mn_iqn_train_step(const float *ring_states, const float *ring_next_states, const int64_t *ring_actions, const float *ring_rewards,
                                 const float *ring_dones, int64_t ring_size, uint64_t *rng_state_dev, const int64_t *idx_dev,
                                 const float *taus_target_dev, const float *taus_local_dev, int64_t *idx_out, float *taus_out, float *params_local,
                                 const float *params_target, float *workspace, float *grad_out, float *loss_out, float *exp_avg, float *exp_avg_sq,
                                 int32_t *step_dev, int32_t batch, int32_t num_taus, float gamma, int32_t flags, double lr, double beta1, double beta2,
                                 double eps, double max_norm, void *stream) {
    if (!params_local || !exp_avg || !exp_avg_sq || !step_dev) return MN_ERR_INVALID;
    const AdamArgs adam = {params_local, exp_avg, exp_avg_sq, step_dev, lr, beta1, beta2, eps, max_norm, nullptr, 1.0f};
    return launch_grad(ring_states, ring_next_states, ring_actions, ring_rewards, ring_dones, rng_state_dev ? nullptr : idx_dev,
                       rng_state_dev ? nullptr : taus_target_dev, rng_state_dev ? nullptr : taus_local_dev, params_local, params_target, workspace,
                       grad_out, loss_out, batch, num_taus, gamma, rng_state_dev, ring_size, idx_out, taus_out, flags, stream, &adam);
}

extern "C" int mn_iqn_train_grad(const float *ring_states, const float *ring_next_states, const int64_t *ring_actions,
                                 const float *ring_rewards, const float *ring_dones, const int64_t *idx_dev,
                                 const float *taus_target_dev, const float *taus_local_dev, const float *params_local,
                                 const float *params_target, float *workspace, float *grad_out, float *loss_out,
                                 int32_t batch, int32_t num_taus, float gamma, void *stream) {
    return launch_grad(ring_states, ring_next_states, ring_actions, ring_rewards, ring_dones, idx_dev, taus_target_dev, taus_local_dev,
                       params_local, params_target, workspace, grad_out, loss_out, batch, num_taus, gamma, nullptr, 0, nullptr, nullptr,
                       0, stream);
}

extern "C" int mn_iqn_train_grad_sampled(const float *ring_states, const float *ring_next_states, const int64_t *ring_actions,
                                         const float *ring_rewards, const float *ring_dones, int64_t ring_size,
                                         uint64_t *rng_state_dev, int64_t *idx_out, float *taus_out, const float *params_local,
                                         const float *params_target, float *workspace, float *grad_out, float *loss_out,
                                         int32_t batch, int32_t num_taus, float gamma, int32_t flags, void *stream) {
    if (!rng_state_dev) return MN_ERR_INVALID;
    return launch_grad(ring_states, ring_next_states, ring_actions, ring_rewards, ring_dones, nullptr, nullptr, nullptr, params_local,
                       params_target, workspace, grad_out, loss_out, batch, num_taus, gamma, rng_state_dev, ring_size, idx_out, taus_out,
                       flags, stream);
}

extern "C" int mn_iqn_train_adam(float *params, float *grad, float *exp_avg, float *exp_avg_sq, int32_t *step_dev,
                                 float *workspace, int32_t batch, double lr, double beta1, double beta2, double eps,
                                 double max_norm, float grad_scale, int32_t grad_rewritten, void *stream) {
    if (!params || !grad || !exp_avg || !exp_avg_sq || !step_dev || !workspace || batch <= 0 || batch % BE) return MN_ERR_INVALID;
    if (!(grad_scale > 0.f)) return MN_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    if ((grad_rewritten || grad_scale != 1.0f) && grad_rewritten != 2)      // the reduction kernel's partial sums of squares no longer describe grad (2: mn_iqn_train_exchange already wrote them for grad_scale * grad)
        hipLaunchKernelGGL(iqn_grad_sumsq, dim3(N_RED), dim3(RED_COLS), 0, s, grad, workspace + ws_sq(batch / BE), grad_scale);
    return launch_adam(params, grad, exp_avg, exp_avg_sq, step_dev, workspace, batch / BE, lr, beta1, beta2, eps, max_norm, grad_scale, s);
}

// ---- one-shot gradient exchange of a shared learner over IPC-mapped mailboxes (see iqn_grad_gather) ---------------------------------
// The mailbox is polled by OTHER devices while this device's kernel is still writing it.  Granules are written and read with system-scope accesses (sc0 sc1: past
// every cache of the issuing device), but what the memory's own caching attributes allow matters too: ordinary hipMalloc memory is coarse-grained -- coherent
// with other agents at kernel boundaries only, as far as the HIP memory model promises -- so the mailbox is allocated UNCACHED (hipDeviceMallocUncached; what
// RCCL uses for the buffers its kernels poll across GPUs on gfx942 / gfx950), else fine-grained, and only if the runtime refuses both as plain device memory
// (mn_xchg_memory_kind tells; round 4 used that: correct between two processes on ONE GPU, unproven across xGMI).  Both kinds are exportable over IPC.
extern "C" int mn_xchg_create(int32_t rank, int32_t world, mn_xchg **out) {
    if (!out || world < 1 || world > XCHG_MAX_RANKS || rank < 0 || rank >= world) return MN_ERR_INVALID;
    mn_xchg *x = new mn_xchg();
    x->rank = rank; x->world = world;
    x->bound = world > 1 ? 3000000000ull : 200000000ull;      // 30 s with peers (one of them may be evaluating or writing a checkpoint), 2 s alone
    const size_t bytes = 2 * (size_t)P_PAD * sizeof(uint64_t);
    void *p = nullptr;
    if (hipGetDevice(&x->device) != hipSuccess) { delete x; return MN_ERR_HIP; }
    if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached) == hipSuccess) x->kind = 2;
    else if ((void)hipGetLastError(), hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) == hipSuccess) x->kind = 1;
    else if ((void)hipGetLastError(), hipMalloc(&p, bytes) == hipSuccess) x->kind = 0;
    else p = nullptr;
    if (!p || hipMemset(p, 0, bytes) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&x->status), sizeof(unsigned)) != hipSuccess || hipMemset(x->status, 0, sizeof(unsigned)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&x->dev_args), sizeof(XchgArgs)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        (void)hipFree(p); (void)hipFree(x->status); (void)hipFree(x->dev_args); delete x;
        return MN_ERR_ALLOC;
    }
    x->own = (gu64 *)p;
    x->peer[rank] = x->own;
    if (int rc = xchg_sync(x)) { (void)hipFree(p); (void)hipFree(x->status); (void)hipFree(x->dev_args); delete x; return rc; }
    *out = x;
    return MN_OK;
}

extern "C" int mn_xchg_memory_kind(mn_xchg *x) { return x ? x->kind : -1; }
extern "C" const char *mn_xchg_last_error(const mn_xchg *x) { return x ? x->err : "mn_xchg: NULL handle"; }

// How long a gather waits for a peer's granules before it gives up (status word raised, the block's update skipped).  Default: 30 s when there are peers, 2 s alone.
extern "C" int mn_xchg_set_timeout_ms(mn_xchg *x, int64_t ms) {
    if (!x || ms <= 0 || ms > 3600000) return MN_ERR_INVALID;
    x->bound = (uint64_t)ms * 100000ull;      // s_memrealtime: 100 MHz
    return xchg_sync(x);
}

extern "C" int mn_xchg_export(mn_xchg *x, void *handle_out) {
    if (!x || !handle_out) return MN_ERR_INVALID;
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "mn_xchg_export / _import exchange 64-byte handles");
    hipIpcMemHandle_t h;
    if (const hipError_t e = hipIpcGetMemHandle(&h, (void *)x->own); e != hipSuccess) {
        (void)hipGetLastError();
        snprintf(x->err, sizeof(x->err), "hipIpcGetMemHandle of rank %d's mailbox failed: %s (memory kind %d; is HSA_ENABLE_IPC_MODE_LEGACY=0 set?)", x->rank, hipGetErrorString(e), x->kind);
        return MN_ERR_PEER;
    }
    memcpy(handle_out, &h, sizeof(h));
    return MN_OK;
}

extern "C" int mn_xchg_import(mn_xchg *x, int32_t peer_rank, const void *handle) {
    if (!x || !handle || peer_rank < 0 || peer_rank >= x->world || peer_rank == x->rank || x->opened[peer_rank]) return MN_ERR_INVALID;
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof(h));
    void *p = nullptr;
    if (const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess); e != hipSuccess || !p) {
        (void)hipGetLastError();
        snprintf(x->err, sizeof(x->err), "hipIpcOpenMemHandle of rank %d's mailbox on device %d failed: %s -- the exchange needs IPC between the ranks' processes and peer "
                 "access between their devices; use the RCCL exchange (--exchange collective) on this node", peer_rank, x->device, hipGetErrorString(e));
        return MN_ERR_PEER;
    }
    x->peer[peer_rank] = (const gu64 *)p;
    x->opened[peer_rank] = true;
    return xchg_sync(x);
}

extern "C" int mn_xchg_attach(mn_xchg *x, float *workspace, int32_t batch, void *stream) {
    if (!workspace || batch <= 0 || batch % BE) return MN_ERR_INVALID;
    const uint64_t v = x ? (uint64_t)(uintptr_t)x->own : 0ull;      // x == NULL detaches
    if (hipMemcpyAsync(workspace + ws_epoch(batch / BE) + 10, &v, sizeof(v), hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess ||
        hipStreamSynchronize((hipStream_t)stream) != hipSuccess)
        return MN_ERR_HIP;
    return MN_OK;
}

extern "C" int mn_iqn_train_exchange(mn_xchg *x, float *grad, float *workspace, int32_t batch, float grad_scale, void *stream) {
    if (!x || !grad || !workspace || batch <= 0 || batch % BE || !(grad_scale > 0.f)) return MN_ERR_INVALID;
    return launch_gather(x, grad, workspace, batch / BE, grad_scale, (hipStream_t)stream);
}

extern "C" int mn_xchg_status(mn_xchg *x, int32_t *timeouts) {
    if (!x || !timeouts) return MN_ERR_INVALID;
    unsigned v = 0;
    if (hipMemcpy(&v, x->status, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return MN_ERR_HIP;
    *timeouts = (int32_t)v;
    return MN_OK;
}

extern "C" int mn_xchg_destroy(mn_xchg *x) {
    if (!x) return MN_ERR_INVALID;
    for (int r = 0; r < x->world; ++r)
        if (x->opened[r]) (void)hipIpcCloseMemHandle((void *)x->peer[r]);
    (void)hipFree((void *)x->own);
    (void)hipFree(x->status);
    (void)hipFree(x->dev_args);
    delete x;
    return MN_OK;
}

// mn_iqn_train_step for a SHARED learner: the one-shot gradient exchange happens inside the reduction + Adam role -- every Adam block
// publishes its 64 reduced columns into this rank's mailbox, gathers the same columns of every rank (rank order), and goes on as in
// mn_iqn_train_step with grad_scale * sum.  One launch per step with MN_TRAIN_ONE_LAUNCH (round 5: the exchange rides in the third role of the forward /
// backward launch), two without, no collective; on a device too small for the fused launches: reduction (publishes), gather, Adam.  All bit-identical to
// mn_iqn_train_grad* + mn_iqn_train_exchange + mn_iqn_train_adam(grad_rewritten = 2).  (The workspace need not be attached with mn_xchg_attach.)
extern "C" int mn_iqn_train_step_xchg(mn_xchg *x, const float *ring_states, const float *ring_next_states, const int64_t *ring_actions,
                                      const float *ring_rewards, const float *ring_dones, int64_t ring_size, uint64_t *rng_state_dev, const int64_t *idx_dev,
                                      const float *taus_target_dev, const float *taus_local_dev, int64_t *idx_out, float *taus_out, float *params_local,
                                      const float *params_target, float *workspace, float *grad_out, float *loss_out, float *exp_avg, float *exp_avg_sq,
                                      int32_t *step_dev, int32_t batch, int32_t num_taus, float gamma, int32_t flags, double lr, double beta1, double beta2,
                                      double eps, double max_norm, float grad_scale, void *stream) {
    if (!x || !params_local || !exp_avg || !exp_avg_sq || !step_dev || !(grad_scale > 0.f)) return MN_ERR_INVALID;
    const AdamArgs adam = {params_local, exp_avg, exp_avg_sq, step_dev, lr, beta1, beta2, eps, max_norm, x, grad_scale};
    return launch_grad(ring_states, ring_next_states, ring_actions, ring_rewards, ring_dones, rng_state_dev ? nullptr : idx_dev,
                       rng_state_dev ? nullptr : taus_target_dev, rng_state_dev ? nullptr : taus_local_dev, params_local, params_target, workspace,
                       grad_out, loss_out, batch, num_taus, gamma, rng_state_dev, ring_size, idx_out, taus_out, flags, stream, &adam);
}
