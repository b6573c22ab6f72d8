#!/usr/bin/env python3
"""bench.py -- env steps/sec of the batched marinenav_env + IQN training loop on MI355X.

One "step" = one vector step of the hot path on one GPU: IQN act (K = 32 quantile samples) for
65 536 envs -> HIP step kernel -> replay append -> HIP reset of finished envs -> (every 4th vector
step) one IQN grad step (batch 256, 8 quantiles, replay 100 000; fused HIP step csrc/iqn_train.hip, or PyTorch
autograd + Adam with --torch-train).  This is BASELINE.json configs[2],
the configuration its metric is quoted on.  With --gpus N every rank runs the same per-GPU workload
on its own env shard (weak scaling; no data-path collective; --shared-learner adds the RCCL gradient
all-reduce of configs[4]).

Prints ONE JSON line on rank 0.  `roofline` is for the DOMINANT kernel of the timed loop, the fused IQN act
kernel (MFMA-bound: 2.003 MFLOP/env-step x envs per launch / mean launch duration vs the dense f32 MFMA peak);
`roofline_env_step` is the HIP step kernel (HBM-bound by north-star: 406 B/env-step (SURVEY 8d, 8 cores + 10
obstacles) x envs per launch / mean launch duration).  With --no-learner the step kernel is the dominant kernel
and `roofline` is its entry.  Both durations are measured with HIP events on the launch stream inside the timed
region.  `cpu_baseline` is the scalar C oracle (oracle/, a port) timed on one host core of this box on a bounded
sample; `cpu_baseline_all_cores` is the same oracle with one env per host thread (up to 64 threads; on the
GPU box 256 threads gave 5.2 M env steps/s, i.e. the container's CPU share is ~9 cores' worth).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_PER_ENV_STEP = {(8, 10): 406, (8, 5): 346, (4, 6): 310}   # 190 + 12 * (n_cores + n_obs)
HBM_PEAK_GBS = 8000.0
# IQN act, K = 32 taus: 2 * 32 * (64*208 + 208*64 + 64*64 + 64*9) ALGORITHMIC FLOP per env-step ("~2.0 MFLOP";
# the kernel takes the tau-mean before the linear output layer, so it executes 960 MFMAs = 1.966 MFLOP + a 9x64 mat-vec)
ACT_FLOP_PER_ENV_STEP = 2 * 32 * (64 * 208 + 208 * 64 + 64 * 64 + 64 * 9)
F32_MFMA_PEAK_TFLOPS = 157.3   # dense v_mfma_f32_*_f32 peak, MI355X_MICROARCH.md
F16_MFMA_PEAK_TFLOPS = 2500.0  # dense f16 / bf16 MFMA peak, MI355X_MICROARCH.md (micro-benchmark ceiling 2382; profiles/r02_f16_split_probe.txt: 2057 sustained by one instruction stream)
# launch-shared taus (mn_iqn_set_tau_mode 1): layer 1 is a constant of the launch; what remains per env-step is the Hadamard product, layers 2-3, output
ACT_SHARED_FLOP_PER_ENV_STEP = 2 * 32 * (208 * 64 + 64 * 64 + 64 * 9)
ACT_SHARED_MFMA_FLOP_PER_ENV_STEP = 216 * 16384
ACT_SPLIT_MFMA_FLOP_PER_ENV_STEP = 372 * 16384   # split-f16 act kernel: 372 v_mfma_f32_16x16x32_f16 per environment (3 per f32 product, layer-2 K padded 208 -> 224)
# HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), measured at
# 65 536 envs, 8 cores / 10 obstacles: profiles/r01_full_loop_kernel_stats.txt.  Not measured live.
# 2 * FETCH_SIZE + WRITE_SIZE (calibration: profiles/r01_pmc_calibration.txt).  step = plain mn_step (r01), step_append = the
# fused step + replay append kernel of the training loop, rollout = mn_rollout at 4 096 envs x 100 steps per launch
# (profiles/r02_full_loop_kernel_stats.txt, profiles/r02_configs1_rollout.txt)
# r04 (profiles/r04_full_loop_kernel_stats.txt): the float64 step + append kernel without the float64 observation copies (mn_enable_obs64 off in the loop):
# 2 x 20.31 + 25.36 = 66.0 MB per 65 536-env launch (r03, copies always written: 79.8 MB); the reset kernel 2 x 0.77 + 1.64 = 3.2 MB per launch (~2 000 episodes end per step)
PMC_TRAFFIC_BYTES = {"step": 30.0e6, "step_append": 51.5e6, "step_append_f64": 66.0e6, "act": 24.8e6, "act_split": 24.2e6, "rollout_4096x100": 69.2e6, "reset_f64": 3.2e6}
RESET_KERNEL_US_FROM_PROFILE = 24.7      # mn_reset_kernel<double, true>, mean of the 120 in-loop launches of the same profile (28.9 incl. the initial all-env reset of 539 us)
# mn_step_append also moves the transition into the replay ring: + 104 B (obs_t row read) + 2 x 104 + 8 + 4 + 4 B written
APPEND_BYTES_PER_ENV_STEP = 104 + 2 * 104 + 8 + 4 + 4


def default_precision(learner):
    """Env-kernel arithmetic when --precision is not given: the strict float64 kernels whenever an IQN is in the loop (every
    float32 output within 1e-5 of the reference with no outliers; measured free next to the act kernel), the mixed-precision
    kernels (SURVEY 8d's float32-SoA design point) for the kernel-only configs."""
    return "f64" if learner else "mixed"


def cpu_baseline(n_steps, world):
    """Scalar float64 oracle (oracle/marinenav_oracle.c), one host thread, resets included."""
    import numpy as np
    from oracle.oracle import OracleEnv
    env = OracleEnv(0)
    env.set_world_size(*world)
    env.reset()
    actions = np.random.RandomState(1000).randint(9, size=n_steps).astype(np.int32)
    t0 = time.perf_counter()
    env.rollout(actions)
    dt = time.perf_counter() - t0
    return n_steps / dt, dt


def cpu_baseline_all_cores(n_steps_each, world, threads):
    """The same oracle, one env per host thread (ctypes releases the GIL for the whole C rollout)."""
    import threading
    import numpy as np
    from oracle.oracle import OracleEnv
    envs = []
    for i in range(threads):
        e = OracleEnv(i)
        e.set_world_size(*world)
        e.reset()
        envs.append(e)
    acts = [np.random.RandomState(1000 + i).randint(9, size=n_steps_each).astype(np.int32) for i in range(threads)]
    ths = [threading.Thread(target=envs[i].rollout, args=(acts[i],)) for i in range(threads)]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    dt = time.perf_counter() - t0
    return threads * n_steps_each / dt, dt


def gpu_clock_probe(device, target_ms=50.0):
    """The clock this GPU holds under f16 matrix load (C-ABI mn_probe_mfma_clock: a pure stream of the act kernel's matrix instruction on
    every CU for ~50 ms).  The same act binary runs 10-12 % slower on some boxes of the pool; with this in the line a reader can tell a
    slow box from a slow kernel (launch time x clock = the kernel's cycles, which do not depend on the box)."""
    import ctypes as C
    import torch
    from distributional_rl_navigation_amd import _capi
    out = (C.c_double * 5)()
    with torch.cuda.device(device):
        rc = _capi.lib().mn_probe_mfma_clock(C.c_double(target_ms), out, C.c_void_p(torch.cuda.current_stream(device).cuda_stream))
    if rc:
        return {"error": rc}
    return {"probe_ms": out[0], "mfma_clock_ghz": out[1], "wave_counter_clock_ghz": out[2], "f16_mfma_tflops_sustained": out[3], "n_cu": int(out[4]),
            "how": "pure v_mfma_f32_16x16x32_f16 stream, 2 waves per SIMD on every CU; clock = 16 cycles x instructions per SIMD / elapsed"}


def rocm_smi_power():
    """Power cap / average power / clocks as rocm-smi reports them, if it is there and answers (never fails the run)."""
    import subprocess
    try:
        r = subprocess.run(["rocm-smi", "--showpower", "--showmaxpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=20)
        j = json.loads(r.stdout[r.stdout.index("{"):])
        card = j.get("card0", next(iter(j.values())))
        keep = {k: v for k, v in card.items() if any(t in k.lower() for t in ("power", "sclk", "mclk"))}
        return keep or None
    except Exception as e:
        return {"unavailable": repr(e)[:120]}


def _timed(device, fn, steps, warmup, state, before_timed=None):
    """W untimed + K timed calls of fn(state) -> state, bracketed by device synchronisation.  Returns (seconds, state)."""
    import torch
    for _ in range(warmup):
        state = fn(state)
    torch.cuda.synchronize(device)
    if before_timed is not None:
        before_timed()
    t0 = time.perf_counter()
    for _ in range(steps):
        state = fn(state)
    torch.cuda.synchronize(device)
    return time.perf_counter() - t0, state


def also_legs(args, env, agent, obs, device, total_timesteps, dist_up, one_batch_leg=False):
    """The other configurations the README / DESIGN quote, timed by the same run so that they are driver-timed numbers too:
      act_exact_f32      the main loop with the exact-f32 MFMA act kernel (variant 0) instead of the split-f16 one
      train_cadence      the cadence that trains (train_iqn's default): 16 gradient steps per vector step, eps 0.05
      config1            BASELINE configs[1]: 4 096 envs, random policy, step kernel only -- one launch pair per vector step, and
                         mn_rollout with T = 100 steps per launch
      shared_learner_ws1 learner alone (batch drawn in the launch), without and with a single-rank RCCL group (all-reduce executed)
    Same synthetic worlds, same agent (its replay ring is full by now)."""
    import torch
    import torch.distributed as dist
    from distributional_rl_navigation_amd.iqn.fused_act import act_context
    from distributional_rl_navigation_amd.marinenav_env.vec_env import VecMarineNavEnv
    n = env.n_envs
    out = {}
    ctx = act_context(agent.qnetwork_local)

    def loop_step(o, eps=None, cvar=args.cvar):
        e = agent.linear_eps(total_timesteps) if eps is None else eps
        return agent.vec_step(env, o, e, cvar, per_iter=n)[0]

    if not one_batch_leg and n % 2 == 0:      # the main loop as two half batches on two streams (what --halves 2 runs)
        from distributional_rl_navigation_amd.iqn.overlap import SplitBatchLoop
        hv = [VecMarineNavEnv(n // 2, seed=0, first_index=env.first_index + h * (n // 2), device=device, precision=env.precision) for h in range(2)]
        for e_ in hv:
            e_.set_attrs(num_cores=args.cores, num_obs=args.obstacles, min_start_goal_dis={4: 30.0, 6: 35.0, 8: 40.0}.get(args.cores, 25.0))
        sp = SplitBatchLoop(agent, hv, act_grid=args.act_grid)
        sp.reset()
        steps = max(20, args.steps // 2)

        def sp_step(_):
            sp.step(agent.linear_eps(total_timesteps) if args.eps is None else args.eps, args.cvar, per_iter=n)
        for _ in range(10):
            sp_step(None)
        sp.join(); torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(steps):
            sp_step(None)
        sp.join(); torch.cuda.synchronize(device)
        dt = time.perf_counter() - t0
        out["two_halves_two_streams"] = {"value": n * steps / dt, "unit": "env steps/s", "ms_per_step": 1e3 * dt / steps, "steps": steps,
                                         "sub_batches": 2, "act_grid": args.act_grid,
                                         "note": "same workload; each half's env kernels run under the other half's act kernel (iqn/overlap.py)"}
        sp.close()
        torch.cuda.synchronize(device)
        for e_ in hv:
            e_.close()
    if one_batch_leg:      # the main loop as ONE batch on one stream (what --halves 1 runs)
        steps = max(20, args.steps // 2)
        dt, obs = _timed(device, loop_step, steps, 10, obs)
        out["one_batch_one_stream"] = {"value": n * steps / dt, "unit": "env steps/s", "ms_per_step": 1e3 * dt / steps, "steps": steps}
    # (a) exact-f32 act kernel
    steps, warm = max(20, args.steps // 2), 10
    ctx.set_variant(0)
    dt, obs = _timed(device, loop_step, steps, warm, obs, lambda: ctx.profile_begin(min(steps, 50)))
    act_ms, _ = ctx.profile_end()
    ctx.set_variant(args.act_variant)
    alg_tf = ACT_FLOP_PER_ENV_STEP * n / (act_ms * 1e-3) / 1e12 if act_ms > 0 else None
    out["act_exact_f32"] = {"value": n * steps / dt, "unit": "env steps/s", "ms_per_step": 1e3 * dt / steps, "steps": steps,
                            "act_kernel": "iqn_qvals_kernel<false> (v_mfma_f32_16x16x4_f32)", "act_launch_ms": act_ms,
                            "roofline_frac_f32_mfma": alg_tf / F32_MFMA_PEAK_TFLOPS if alg_tf else None}
    # (a') launch-shared taus: one set of 32 quantile fractions per act launch instead of per env (opt-in, IQNAgent.shared_taus)
    def shared_leg(fn, steps, warm, form=True):
        nonlocal obs
        agent.shared_taus = form
        try:
            dt, obs = _timed(device, fn, steps, warm, obs, lambda: ctx.profile_begin(min(steps, 50)))
            ms, _ = ctx.profile_end()
        finally:
            agent.shared_taus = False
        return dt, ms
    steps = max(20, args.steps // 2)
    dt, act_ms = shared_leg(loop_step, steps, 10)
    rate = n / (act_ms * 1e-3) / 1e12 if act_ms > 0 else None
    dt_w, act_ms_wave = shared_leg(loop_step, 30, 5, form="wave")
    tiled = n >= 65536
    out["act_shared_taus"] = {"value": n * steps / dt, "unit": "env steps/s", "ms_per_step": 1e3 * dt / steps, "steps": steps,
                              "act_kernel": ("iqn_qvals_tiled_kernel (launch-shared taus, the MFMA columns are environments: T = W2 diag(h1) built once per launch, features split "
                                             "once per env; 216 v_mfma_f32_16x16x32_f16 + ~260 vector instructions per env)" if tiled else
                                             "iqn_qvals_split_kernel<false, SHARED=true> (layer 1 = a [32 x 208] constant of the launch; 216 v_mfma_f32_16x16x32_f16 per env)"),
                              "launch_ms_wavefront_per_env_form": act_ms_wave, "value_wavefront_per_env_form": n * 30 / dt_w,
                              "launch_ms": act_ms, "tau_draw": "32 taus ~ U[0,1) x cvar per LAUNCH, shared by its envs (default: per env)",
                              "frac_algorithmic_remaining_flops": ACT_SHARED_FLOP_PER_ENV_STEP * rate / F16_MFMA_PEAK_TFLOPS if rate else None,
                              "frac_algorithmic_full_network_flops": ACT_FLOP_PER_ENV_STEP * rate / F16_MFMA_PEAK_TFLOPS if rate else None,
                              "frac_issued_mfma": ACT_SHARED_MFMA_FLOP_PER_ENV_STEP * rate / F16_MFMA_PEAK_TFLOPS if rate else None,
                              "remaining_flop_per_env_step": ACT_SHARED_FLOP_PER_ENV_STEP, "full_network_flop_per_env_step": ACT_FLOP_PER_ENV_STEP}
    # (b) the cadence that trains
    ue, gs = agent.UPDATE_EVERY, agent.grad_steps_per_update
    agent.UPDATE_EVERY, agent.grad_steps_per_update = 1, 16
    g0 = agent.grad_steps
    steps = max(20, args.steps // 2)
    dt, obs = _timed(device, lambda o: agent.vec_step(env, o, 0.05, args.cvar, train_every=1, per_iter=n)[0], steps, 10, obs)
    # _timed's warm-up steps train too: count the timed ones only
    out["train_cadence"] = {"value": n * steps / dt, "unit": "env steps/s", "ms_per_step": 1e3 * dt / steps, "steps": steps,
                            "grad_steps_per_vector_step": 16, "eps": 0.05,
                            "grad_steps_per_sec": 16 * steps / dt, "grad_steps_counted": agent.grad_steps - g0 - 16 * 10,
                            "launches_per_grad_step": (1 if agent._fused._one_launch_flags(agent.BATCH_SIZE) else 2) if getattr(agent, "_fused", None) is not None else None,
                            "xcd_misplaced_workgroups": agent._fused.xcd_misplaced() if getattr(agent, "_fused", None) is not None else None}
    dt, act_ms = shared_leg(lambda o: agent.vec_step(env, o, 0.05, args.cvar, train_every=1, per_iter=n)[0], steps, 10)
    out["train_cadence_shared_taus"] = {"value": n * steps / dt, "unit": "env steps/s", "ms_per_step": 1e3 * dt / steps, "steps": steps,
                                        "grad_steps_per_vector_step": 16, "eps": 0.05, "grad_steps_per_sec": 16 * steps / dt, "act_launch_ms": act_ms}
    agent.UPDATE_EVERY, agent.grad_steps_per_update = ue, gs
    # (c) configs[1]
    n1 = 4096

    def config1(cores, obstacles, min_dis, precision, single_steps=1000):
        """4 096 envs, random policy, step kernel only: (single launch pairs, mn_rollout T = 100) for one world size and arithmetic."""
        e1 = VecMarineNavEnv(n1, seed=0, device=device, precision=precision)
        e1.set_attrs(num_cores=cores, num_obs=obstacles, min_start_goal_dis=min_dis)
        e1.reset()
        gen = torch.Generator(device=device); gen.manual_seed(0)
        bytes_step = BYTES_PER_ENV_STEP.get((cores, obstacles), 190 + 12 * (cores + obstacles))

        def pair(_):
            e1.step(torch.randint(0, 9, (n1,), device=device, dtype=torch.int32, generator=gen))
            return e1.reset_done()
        dt, _ = _timed(device, pair, single_steps, 100, None, lambda: e1.profile_begin(50))
        k_ms, _ = e1.profile_end()
        single = {"value": n1 * single_steps / dt, "unit": "env steps/s", "ms_per_step": 1e3 * dt / single_steps, "steps": single_steps,
                  "step_kernel_ms": k_ms, "algorithmic_bytes_per_env_step": bytes_step,
                  "hbm_frac": bytes_step * n1 / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if k_ms > 0 else None,
                  "hbm_frac_wall": bytes_step * n1 * single_steps / dt / 1e9 / HBM_PEAK_GBS}
        T, ctr = 100, [0]

        def roll_launch(_):
            e1.rollout(T, action_seed=0, first_step=ctr[0], trace=("obs", "reward", "done"))
            ctr[0] += T
        dt, _ = _timed(device, roll_launch, 10, 2, None, lambda: e1.profile_begin(10))
        k_ms, _ = e1.profile_end()
        rollout = {"value": n1 * T * 10 / dt, "unit": "env steps/s", "ms_per_step": 1e3 * dt / (T * 10), "steps": T * 10, "steps_per_launch": T,
                   "launch_ms": k_ms, "algorithmic_bytes_per_env_step": bytes_step,
                   "hbm_frac": bytes_step * n1 * T / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if k_ms > 0 else None}
        e1.close()
        return single, rollout
    single, rollout = config1(args.cores, args.obstacles, {4: 30.0, 6: 35.0, 8: 40.0}.get(args.cores, 25.0), "mixed")
    # stage 2 (8 cores / 10 obstacles, the headline world), the reference's default world (8 / 5, marinenav_env.py:62-63) and
    # curriculum stage 0 (4 / 6, train_IQN_model.py:85-88); in the float64 kernels (north-star tolerance with zero outliers) and in mixed precision
    worlds = {}
    for (nc, no, md) in ((8, 10, 40.0), (8, 5, 25.0), (4, 6, 30.0)):
        for prec in ("f64", "mixed"):
            if (nc, no, prec) == (args.cores, args.obstacles, "mixed"):
                s_, r_ = single, rollout
            else:
                s_, r_ = config1(nc, no, md, prec, single_steps=500)
            worlds.setdefault(f"{nc}_cores_{no}_obstacles", {})[prec] = {"single_launch_pair": s_, "mn_rollout": r_}
    out["config1"] = {"envs": n1, "precision": "mixed", "single_launch_pair": single, "mn_rollout": rollout, "worlds": worlds}
    # (d) learner alone, without / with a single-rank RCCL group
    def learner_rate(reps=400):
        for _ in range(10):
            agent.train_from_memory()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(reps):
            agent.train_from_memory()
        torch.cuda.synchronize(device)
        return reps / (time.perf_counter() - t0)
    sl = {}
    was = agent.distributed
    agent.distributed = False
    sl["no_group"] = learner_rate()
    made = False
    try:
        if not dist_up:
            import socket
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=device)
            made = True
        agent.distributed = True
        sl["nccl_ws1_allreduce"] = learner_rate()
        sl["allreduce_overhead_us_per_step"] = 1e6 * (1.0 / sl["nccl_ws1_allreduce"] - 1.0 / sl["no_group"])
        # the same gradient steps as ONE captured hipGraph per 16-step training event, RCCL all-reduce captured too (--graph-train):
        # does capture hide any of the collective's cost?
        def graphed_rate(events=25, G=16):
            was_g = agent.use_fused_graph
            agent.use_fused_graph = True
            try:
                for _ in range(3):
                    agent.train_steps_from_memory(G)
                torch.cuda.synchronize(device)
                t0 = time.perf_counter()
                for _ in range(events):
                    agent.train_steps_from_memory(G)
                torch.cuda.synchronize(device)
                return events * G / (time.perf_counter() - t0)
            finally:
                agent.use_fused_graph = was_g
        sl["nccl_ws1_allreduce_graphed_16_step_events"] = graphed_rate()
        agent.distributed = False
        sl["no_group_graphed_16_step_events"] = graphed_rate()
        sl["xcd_misplaced_workgroups"] = agent._fused.xcd_misplaced()
        agent.distributed = True
        sl["allreduce_overhead_us_per_step_graphed"] = 1e6 * (1.0 / sl["nccl_ws1_allreduce_graphed_16_step_events"] - 1.0 / sl["no_group_graphed_16_step_events"])
        # the one-shot exchange over IPC-mapped mailboxes instead of the RCCL all-reduce (iqn/mailbox.py; one rank: its own mailbox only)
        agent.exchange = "mailbox"
        try:
            agent.exchange_fused_adam, agent.two_launch_step = False, False
            sl["mailbox_ws1_four_launches"] = learner_rate()      # forward / backward, reduction (publishes), gather, Adam
            agent.exchange_fused_adam = True
            sl["mailbox_ws1_three_launches"] = learner_rate()     # ... gather + clip + Adam in one launch
            agent.two_launch_step = True
            sl["mailbox_ws1_exchange"] = learner_rate()           # the exchange inside the reduction + Adam launch: two launches, like a single learner
            sl["mailbox_overhead_us_per_step"] = 1e6 * (1.0 / sl["mailbox_ws1_exchange"] - 1.0 / sl["no_group"])
            sl["mailbox_ws1_exchange_graphed_16_step_events"] = graphed_rate()
            sl["mailbox_timeouts"] = agent._fused._mailbox.timeouts()
        finally:
            agent.exchange = "collective"
    except Exception as e:      # the line must still come out if RCCL cannot initialise on this box
        sl["nccl_ws1_error"] = repr(e)
    finally:
        agent.distributed = was
        if made:
            torch.cuda.synchronize(device)
            dist.destroy_process_group()
    sl["unit"] = "grad-steps/s (batch %d, drawn in the launch)" % agent.BATCH_SIZE
    out["shared_learner_ws1"] = sl
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--envs", type=int, default=65536, help="envs per GPU")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--replay", type=int, default=100_000)
    ap.add_argument("--cores", type=int, default=8)
    ap.add_argument("--obstacles", type=int, default=10)
    ap.add_argument("--shared-learner", action="store_true", help="one IQN, RCCL grad all-reduce (configs[4])")
    ap.add_argument("--cvar", type=float, default=1.0)
    ap.add_argument("--no-learner", action="store_true", help="random policy, step kernel only (configs[1])")
    ap.add_argument("--cpu-steps", type=int, default=8_000_000, help="oracle sample for cpu_baseline (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=-1,
                    help="threads for the multi-thread oracle baseline (-1 = host CPUs this process may use, capped at 64; 0 = skip)")
    ap.add_argument("--act-chunk", type=int, default=8192)
    ap.add_argument("--no-train-graph", action="store_true", help="eager grad step instead of the captured hipGraph")
    ap.add_argument("--no-learner-only", action="store_true", help="skip the learner-alone measurement after the timed loop (cleaner profiles)")
    ap.add_argument("--torch-train", action="store_true", help="grad step through PyTorch autograd + Adam instead of the fused HIP step")
    ap.add_argument("--torch-act", action="store_true", help="act through eager PyTorch instead of the fused HIP kernel")
    ap.add_argument("--robot-n", type=int, default=10, help="sub-steps per action (robot.N; 10 = reference; ablation only)")
    ap.add_argument("--lanes", type=int, default=0, help="lanes per env in the step kernel (0 = library default)")
    ap.add_argument("--update-every", type=int, default=4, help="vector steps between training events (UPDATE_EVERY; SURVEY 8d C3: 4)")
    ap.add_argument("--grad-steps", type=int, default=1, help="gradient steps per training event (1 = the reference's cadence)")
    ap.add_argument("--eps", type=float, default=None, help="fixed exploration rate (default: the reference's linear schedule at the start of training, ~1.0)")
    ap.add_argument("--rollout", type=int, default=0, metavar="T",
                    help="with --no-learner: T vector steps per launch through mn_rollout (in-kernel random actions and resets); "
                         "--steps must be a multiple of T.  0 = one mn_step + mn_reset_done launch pair per vector step")
    ap.add_argument("--rollout-trace", default="obs,reward,done", help="per-step outputs mn_rollout writes ([T][n] traces), comma separated")
    ap.add_argument("--precision", default=None, choices=["mixed", "f64"],
                    help="env kernels: f64 (everything float64, 1e-9; default when an IQN is in the loop) or mixed (float32 field / sonar "
                         "decisions; default with --no-learner)")
    ap.add_argument("--no-also", action="store_true", help="skip the extra driver-timed legs (`also`: exact-f32 act kernel, training "
                                                            "cadence, configs[1], single-rank RCCL learner) after the main timed region")
    ap.add_argument("--act-variant", type=int, default=2, choices=(0, 1, 2, 3), help="acting kernel: 2 = split-f16 MFMA at float32 accuracy (default), 0 = exact-f32 v_mfma_f32_16x16x4_f32, 1 = its v_mfma_f32_32x32x2_f32 re-layout, 3 = split-f16 on 32x32x16 tiles")
    ap.add_argument("--separate-append", action="store_true", help="mn_step + mn_replay_append as two launches instead of the fused mn_step_append")
    ap.add_argument("--halves", type=int, default=1,
                    help="sub-batches of the envs, each with its own act -> step -> reset chain on its own HIP stream (iqn/overlap.py: the env "
                         "kernels of one half run under the act kernel of the other; timed by the default run as also.two_halves_two_streams: "
                         "+5-7 %% env steps/s).  1 (default) = one batch on one stream: one act launch = one vector step, so the launch duration "
                         "behind `roofline` is that of a kernel that has the GPU to itself (with two streams the two halves' act kernels "
                         "overlap each other and a per-launch duration no longer measures the kernel)")
    ap.add_argument("--graph-train", action="store_true", help="the gradient steps of a training event as one captured hipGraph (IQNAgent.use_fused_graph)")
    ap.add_argument("--shared-taus", action="store_true", help="one set of 32 taus per act LAUNCH instead of per env (IQNAgent.shared_taus; opt-in, "
                                                               "timed by the default run as also.act_shared_taus)")
    ap.add_argument("--no-clock-probe", action="store_true", help="skip the two ~50 ms MFMA clock probes around the timed region (cleaner rocprofv3 tables)")
    ap.add_argument("--act-grid", type=int, default=1024, help="with --halves > 1: workgroups of an act launch (mn_iqn_set_grid)")
    args = ap.parse_args()
    if args.precision is None:
        args.precision = default_precision(learner=not args.no_learner)

    import torch
    import torch.distributed as dist
    from distributional_rl_navigation_amd.iqn.agent import IQNAgent
    from distributional_rl_navigation_amd.marinenav_env.vec_env import VecMarineNavEnv

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as plain `python bench.py --gpus N`: re-launch as N ranks (one per GPU) under torch.distributed.run
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.execvp(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                                   "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus} (WORLD_SIZE={world})")
    device = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(device)
    # under torch.distributed.run: RCCL, even for 1 rank; a plain `python bench.py --shared-learner` (N = 1) forms a
    # single-rank RCCL group itself so that configs[4]'s gradient all-reduce executes
    use_dist = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ or args.shared_learner
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if "MASTER_ADDR" in os.environ and "RANK" in os.environ:
            dist.init_process_group("nccl", device_id=device)
        else:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=device)

    n = args.envs
    min_dis = {4: 30.0, 6: 35.0, 8: 40.0}.get(args.cores, 25.0)
    H = args.halves if (not args.no_learner and not args.torch_act and not args.separate_append and n % max(1, args.halves) == 0) else 1
    envs = [VecMarineNavEnv(n // H, seed=0, first_index=rank * n + h * (n // H), device=device, precision=args.precision, step_lanes=args.lanes if args.lanes in (1, 2, 4, 8) else 0,
                            rollout_lanes=args.lanes if args.lanes != 1 else 0) for h in range(H)]
    for e_ in envs:
        e_.set_attrs(num_cores=args.cores, num_obs=args.obstacles, min_start_goal_dis=min_dis, N=args.robot_n)
    env = envs[0]
    obs = env.reset() if H == 1 else None
    agent = None
    if not args.no_learner:
        agent = IQNAgent(26, 9, BATCH_SIZE=args.batch, BUFFER_SIZE=args.replay, device=device,
                         seed=100 if args.shared_learner else 100 + rank, learning_starts=0,
                         distributed=args.shared_learner and use_dist, act_chunk=args.act_chunk,
                         UPDATE_EVERY=args.update_every, rank=rank if args.shared_learner else 0)
        agent.grad_steps_per_update = args.grad_steps
        agent.use_fused_graph = args.graph_train
        agent.shared_taus = args.shared_taus
    if agent is not None and args.torch_act:
        agent.use_fused_act = False
    if agent is not None and args.no_train_graph:
        agent.use_train_graph = False
    if agent is not None:
        agent.use_fused_train = not args.torch_train
    total_timesteps = 3_000_000 * n * world      # eps stays on the reference's initial 10 % ramp
    gen = torch.Generator(device=device)
    gen.manual_seed(rank)

    class _NoAppend:      # --separate-append: hide step_append so that vec_step takes the two-launch path
        def __init__(self, e):
            self._e = e
        def __getattr__(self, k):
            if k == "step_append":
                raise AttributeError(k)
            return getattr(self._e, k)
    loop_env = _NoAppend(env) if args.separate_append else env
    eps_seen = []
    split = None
    if H > 1:
        from distributional_rl_navigation_amd.iqn.overlap import SplitBatchLoop
        split = SplitBatchLoop(agent, envs, act_grid=args.act_grid)
        split.reset()

    def one_step(o):
        if agent is None:
            a = torch.randint(0, 9, (n,), device=device, dtype=torch.int32, generator=gen)
            env.step(a)
            return env.reset_done()
        eps = agent.linear_eps(total_timesteps) if args.eps is None else args.eps
        eps_seen.append(eps)
        if split is not None:
            return split.step(eps, args.cvar, per_iter=n * world)[0]
        return agent.vec_step(loop_env, o, eps, args.cvar, per_iter=n * world)[0]

    def fence():
        if split is not None:
            split.join()
        torch.cuda.synchronize(device)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(device)

    if args.rollout > 0 and agent is not None:
        raise SystemExit("--rollout is the random-policy workload (BASELINE configs[1]): use it with --no-learner")
    roll = args.rollout if args.rollout > 0 else 0
    if roll and (args.steps % roll or args.warmup % roll):
        raise SystemExit(f"--rollout {roll}: --steps and --warmup must be multiples of it")
    trace = tuple(k for k in args.rollout_trace.split(",") if k)
    step_ctr = [0]

    def run_steps(k, o):
        if not roll:
            for _ in range(k):
                o = one_step(o)
            return o
        for _ in range(k // roll):      # k vector steps as k / T launches of T steps each
            env.rollout(roll, action_seed=rank, first_step=step_ctr[0], trace=trace)
            step_ctr[0] += roll
        return env.obs

    fused = agent is not None and agent.use_fused_act
    if fused:      # before the warm-up: the first timed launch must not contain the weight-image pack of a freshly selected kernel
        from distributional_rl_navigation_amd.iqn.fused_act import act_context
        act_context(agent.qnetwork_local).set_variant(args.act_variant)
    obs = run_steps(args.warmup, obs)
    g0 = agent.grad_steps if agent else 0
    clock_before = gpu_clock_probe(device) if (rank == 0 and not args.no_clock_probe) else None      # ~50 ms of matrix load, outside the timed region
    fence()
    # HIP-event pairs are recorded around the first n_prof act / step launches of the timed region; not around all of
    # them, because the four event records per vector step cost ~18 us of stream time (measured: 1.079 ms/step with
    # 200 instrumented steps, 1.067 with 50, 1.061 with 1) -- `launches_timed` in the roofline objects says how many
    n_prof = min(args.steps // max(1, roll), int(os.environ.get("MN_BENCH_NPROF", "50")))
    env.profile_begin(n_prof)
    import ctypes as C
    from distributional_rl_navigation_amd import _capi
    if fused:
        act_context(agent.qnetwork_local).profile_begin(n_prof)
    t0 = time.perf_counter()
    obs = run_steps(args.steps, obs)
    fence()
    elapsed = time.perf_counter() - t0
    step_kernel_ms, launches = env.profile_end()
    act_ms, act_launches = 0.0, 0
    if fused:
        act_ms, act_launches = act_context(agent.qnetwork_local).profile_end()
    grad_steps = (agent.grad_steps - g0) if agent else 0
    clock_after = gpu_clock_probe(device) if (rank == 0 and not args.no_clock_probe) else None
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    all_reduce_ms = None
    if use_dist and args.shared_learner:      # the shared learner's collective alone: 35 785-float bucket, back to back (max over ranks)
        bucket = torch.zeros(35785, dtype=torch.float32, device=device)
        for _ in range(20):
            dist.all_reduce(bucket)
        torch.cuda.synchronize(device)
        t1 = time.perf_counter()
        for _ in range(200):
            dist.all_reduce(bucket)
        torch.cuda.synchronize(device)
        tt = torch.tensor([(time.perf_counter() - t1) / 200 * 1e3], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        all_reduce_ms = float(tt.item())
    also = {}
    if world == 1 and not args.no_also and agent is not None and fused and not roll and not args.torch_train and not args.torch_act:
        if split is not None:      # the extra legs run the one-batch loop on their own handle
            split.close()
            torch.cuda.synchronize(device)
            for e_ in envs:
                e_.close()
            env = VecMarineNavEnv(n, seed=0, first_index=rank * n, device=device, precision=args.precision)
            env.set_attrs(num_cores=args.cores, num_obs=args.obstacles, min_start_goal_dis=min_dis, N=args.robot_n)
            obs = env.reset()
            envs = [env]
        also = also_legs(args, env, agent, obs, device, total_timesteps, use_dist, one_batch_leg=split is not None)

    # learner alone (outside the timed region): back-to-back IQN grad steps (sample + train), batch 256, 8 quantiles.
    # Eager = what the loop uses (there the ~130 launches hide behind the act kernel); back to back the eager step is
    # CPU-launch-bound, which is where the captured hipGraph (IQNAgent.use_train_graph) pays.
    learner_only = {}
    if agent is not None and len(agent.memory) > agent.BATCH_SIZE and not args.no_learner_only:
        was_fused = agent.use_fused_train
        for mode in ("fused_hip", "eager", "hipgraph"):       # the torch modes last: they advance torch's own Adam state
            if mode == "hipgraph" and agent.distributed:
                continue
            agent.use_fused_train = (mode == "fused_hip")
            agent.use_train_graph = (mode == "hipgraph")
            reps = 200 if mode == "fused_hip" else 50
            for _ in range(5):
                agent.train_from_memory()
            torch.cuda.synchronize(device)
            t1 = time.perf_counter()
            for _ in range(reps):
                agent.train_from_memory()
            torch.cuda.synchronize(device)
            learner_only[mode] = reps / (time.perf_counter() - t1)
        agent.use_fused_train = was_fused

    if rank == 0:
        env_steps = n * world * args.steps
        bytes_step = BYTES_PER_ENV_STEP.get((args.cores, args.obstacles), 190 + 12 * (args.cores + args.obstacles))
        fused_append = agent is not None and not args.separate_append and device.type == "cuda"
        # the training loop's env kernel is mn_step_append: the step's 406 B + the transition it writes to the ring
        bytes_per = bytes_step + (APPEND_BYTES_PER_ENV_STEP if fused_append else 0)
        # one launch processes n env-steps (single step) or n * T env-steps (mn_rollout)
        per_launch = (n // H) * max(1, roll)
        achieved = bytes_per * per_launch / (step_kernel_ms * 1e-3) / 1e9 if step_kernel_ms > 0 else 0.0
        out = {
            "metric": "env steps/sec (whole node) at 65 536 envs; IQN grad-steps/sec",
            "value": env_steps / elapsed,
            "unit": "env steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "dtype_detail": ("IQN act: f32 results from split-f16 MFMA (hi/lo f16 pieces, 3 products per f32 product; error vs float64 equal to the "
                             "exact-f32 kernel's, tests/test_act_split_gpu.py), IQN train: f32 (exact-f32 MFMA); env kernels: " if args.act_variant == 2 else
                             "IQN act / train: f32 (exact-f32 MFMA); env kernels: ") + ("f64 pose integration + f64 sonar geometry, f32 field and sonar decisions"
                                                                                      if args.precision == "mixed" else "float64 throughout (strict: every float32 output within 1e-5 of the reference, zero outliers; "
                                                                                      "tests/test_env_gpu.py::test_loop_default_precision_is_strict_1e5_with_zero_outliers)"),
            "data": "synthetic (seeded random worlds, random-init IQN)",
            "act_kernel_variant": args.act_variant,
            "act_tau_mode": "shared per launch (opt-in)" if args.shared_taus else "per env (default)",
            "config": {
                "workload": ((f"step kernel only, random policy, {roll} vector steps per launch (mn_rollout: in-kernel actions + resets, traces: {','.join(trace) or 'none'})"
                              if roll else "step kernel only, random policy, one mn_step + mn_reset_done launch pair per vector step") if agent is None else
                             f"{n} envs/GPU + IQN training (act K=32, 8 quantiles, replay {args.replay}, batch {args.batch}, "
                             f"{args.grad_steps} grad step(s) every {args.update_every} vector steps)"
                             + (f"; the envs of a GPU stepped as {H} sub-batches of {n // H} on {H} HIP streams" if H > 1 else "")),
                "envs_per_gpu": n, "n_cores": args.cores, "n_obstacles": args.obstacles,
                "learner": "none" if agent is None else ("shared, RCCL grad all-reduce" if args.shared_learner else "independent per GPU"),
                "cvar": args.cvar,
                "process_group": dist.get_backend() if use_dist else None,
                # exploration rate of the timed steps: the reference's schedule at the start of a run (the policy is
                # ~uniformly random; the act kernel evaluates every Q-value regardless of eps), or --eps
                "eps": None if agent is None else (sum(eps_seen[-args.steps:]) / max(1, len(eps_seen[-args.steps:]))),
                "update_every_vector_steps": None if agent is None else args.update_every,
                "grad_steps_per_event": None if agent is None else args.grad_steps,
                "sub_batches": H, "act_grid": args.act_grid if H > 1 else 0,
                "replay_append": "none" if agent is None else ("separate launch" if args.separate_append else "fused into the step kernel (mn_step_append)"),
                "ablation": bool(_capi.lib().mn_build_info() & 1),     # from the loaded library: False = full kernels
            },
            "grad_steps_per_sec": grad_steps * (1 if args.shared_learner else world) / elapsed,
            "all_reduce_ms": all_reduce_ms,      # shared learner only: one RCCL all-reduce of the 143 KB gradient bucket, back to back
            "learner_only_grad_steps_per_sec_per_gpu": learner_only,   # sample + train back to back, outside the timed region
            # further configurations timed by THIS run, after the main timed region (N = 1 only): see also_legs()
            "also": also,
            # which clock this box sustains under f16 matrix load, probed right before and right after the timed region, + rocm-smi's view
            "gpu_clock_probe": {"before_timed_region": clock_before, "after_timed_region": clock_after, "rocm_smi": rocm_smi_power()},
            "roofline_env_step": {
                "precision": args.precision,
                "kernel": (lambda t_: f"mn_rollout_kernel<{t_},L>" if roll else (f"mn_step_kernel<{t_},L,APPEND=true> (step + replay append)" if fused_append else f"mn_step_kernel<{t_},L>"))
                          ("float,false" if args.precision == "mixed" else "double,true"),
                "env_steps_per_launch": per_launch,
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": None,      # not counted live
                "traffic_from_profile": {"bytes_per_launch": (PMC_TRAFFIC_BYTES["rollout_4096x100"] if (roll, n) == (100, 4096) else None) if roll else
                                         (PMC_TRAFFIC_BYTES["step_append_f64"] if (fused_append and (n // H, args.cores, args.obstacles, args.precision) == (65536, 8, 10, "f64")) else
                                          (PMC_TRAFFIC_BYTES["step_append" if fused_append else "step"] if (n // H, args.cores, args.obstacles, args.precision) == (65536, 8, 10, "mixed") else None)),
                                         "source": "rocprofv3 PMC passes (2 x FETCH_SIZE + WRITE_SIZE per launch): float64 kernels profiles/r04_full_loop_kernel_stats.txt, mixed-precision kernels profiles/r02_full_loop_kernel_stats_split_act.txt, mn_rollout profiles/r03_rollout_phase_timing.txt"},
                # the other env kernel of a vector step: the reset of the envs that finished (one wavefront per env, outside the figures above)
                "reset_kernel": None if (roll or agent is None or args.precision != "f64" or n // H != 65536) else
                                {"kernel": "mn_reset_kernel<double, true>", "launch_us_from_profile": RESET_KERNEL_US_FROM_PROFILE,
                                 "traffic_bytes_per_launch_from_profile": PMC_TRAFFIC_BYTES["reset_f64"], "episodes_ending_per_vector_step": "~2 000 of 65 536",
                                 "source": "profiles/r04_full_loop_kernel_stats.txt"},
                "algorithmic_bytes_per_env_step": bytes_per,
                "algorithmic_bytes_step_only": bytes_step,
                "frac_step_bytes_only": (bytes_step * per_launch / (step_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if step_kernel_ms > 0 else None,
                "launch_ms": step_kernel_ms,
                "launches_timed": launches,
                "kernel_only_env_steps_per_sec": per_launch / (step_kernel_ms * 1e-3) if step_kernel_ms > 0 else None,
            },
        }
        if fused and act_ms > 0:
            n_act = n // H      # envs per act launch
            alg_tf = ACT_FLOP_PER_ENV_STEP * n_act / (act_ms * 1e-3) / 1e12
            if args.act_variant in (2, 3):
                # split-f16 kernel: the matrix pipe executes three f16 MFMAs per float32 product; `achieved` counts the FLOPs it
                # ISSUES (incl. the 3x and the K padding) against the f16 dense peak, i.e. the fraction of the pipe that is busy
                issued = ACT_SPLIT_MFMA_FLOP_PER_ENV_STEP if args.act_variant == 2 else 198 * 32768
                tf = issued * n_act / (act_ms * 1e-3) / 1e12
                kern = ("iqn_qvals_split_kernel (3 x v_mfma_f32_16x16x32_f16 per f32 product, f32-class accuracy)" if args.act_variant == 2 else
                        "iqn_qvals_split32_kernel (3 x v_mfma_f32_32x32x16_f16 per f32 product, f32-class accuracy)")
                peak = F16_MFMA_PEAK_TFLOPS
                if args.shared_taus and args.act_variant == 2:
                    tf = ACT_SHARED_MFMA_FLOP_PER_ENV_STEP * n_act / (act_ms * 1e-3) / 1e12
                    kern = ("iqn_qvals_tiled_kernel (launch-shared taus, environments in the MFMA columns: T = W2 diag(h1) built once per launch, 216 MFMAs per env)" if n_act >= 65536 else
                            "iqn_qvals_split_kernel<false, SHARED=true> (launch-shared taus: layer 1 is a constant of the launch, 216 MFMAs per env)")
            else:
                tf = alg_tf
                kern = "iqn_qvals32_kernel (v_mfma_f32_32x32x2_f32)" if args.act_variant == 1 else "iqn_qvals_kernel<false> (v_mfma_f32_16x16x4_f32)"
                peak = F32_MFMA_PEAK_TFLOPS
            out["roofline"] = {   # dominant kernel of this workload (~85 % of GPU time): the fused IQN act kernel
                "kernel": kern, "bound": "mfma", "achieved": tf, "peak": peak,
                "unit": "TFLOP/s", "frac": tf / peak,
                # `frac` counts the matrix FLOPs the kernel ISSUES (split-f16: 3 products per float32 product + K padding);
                # `frac_algorithmic` is SURVEY 8d's figure: the network's 2 002 944 FLOP per env-step over the same peak
                "frac_algorithmic": alg_tf / peak,
                "traffic": None,      # HBM bytes are not counted live; the rocprofv3 PMC figure of the same kernel is next to it
                "traffic_from_profile": {"bytes_per_launch": PMC_TRAFFIC_BYTES["act_split" if args.act_variant in (2, 3) else "act"] if n_act == 65536 else None,
                                         "source": "rocprofv3 PMC passes (2 x FETCH_SIZE + WRITE_SIZE per launch: observations + taus in, actions out), "
                                                   + ("profiles/r03_full_loop_kernel_stats.txt" if args.act_variant in (2, 3) else "profiles/r01_full_loop_kernel_stats.txt")},
                "issued_mfma_flop_per_env_step": ACT_SPLIT_MFMA_FLOP_PER_ENV_STEP if args.act_variant == 2 else (198 * 32768 if args.act_variant == 3 else ACT_FLOP_PER_ENV_STEP),
                "algorithmic_flop_per_env_step": ACT_FLOP_PER_ENV_STEP, "algorithmic_tflops": alg_tf,
                "algorithmic_tflops_over_f32_mfma_peak": alg_tf / F32_MFMA_PEAK_TFLOPS,
                "launch_ms": act_ms, "launches_timed": act_launches, "env_steps_per_launch": n_act,
                # box-independent form of launch_ms: the launch's duration in cycles of the clock the probe measured (mean of before / after)
                "act_effective_clock_ghz": (lambda cs: sum(cs) / len(cs) if cs else None)([c["mfma_clock_ghz"] for c in (clock_before, clock_after) if c and "mfma_clock_ghz" in c]),
            }
            ghz = out["roofline"]["act_effective_clock_ghz"]
            out["roofline"]["launch_kilocycles_at_that_clock"] = act_ms * 1e-3 * ghz * 1e9 / 1e3 if ghz else None
        else:
            out["roofline"] = out["roofline_env_step"]
        if args.cpu_steps > 0 and world == 1:      # reported baseline: rank 0 at N = 1 only
            v, dt = cpu_baseline(args.cpu_steps, (args.cores, args.obstacles, min_dis))
            out["cpu_baseline"] = {
                "value": v, "unit": "env steps/s", "cores": 1, "kind": "port",
                "sample": f"{args.cpu_steps} steps of one env (oracle/marinenav_oracle.c, float64 scalar), "
                          f"{args.cores} cores / {args.obstacles} obstacles, random actions, resets included, {dt:.1f} s",
                "host_cpus": os.cpu_count(),
                # the reference's own Python step() cannot travel to the GPU box; its rate was measured where the reference can be imported
                "reference_python": {"value": 293, "unit": "env steps/s", "cores": 1, "where": "build container, 8 vCPU Xeon 2.1 GHz",
                                     "what": "the reference's MarineNavEnv.step (marinenav_env.py:199), 8 cores / 10 obstacles, 1 process, random actions, resets included "
                                             "(default world 8 / 5: 275; stage 0, 4 / 6: 604; 8 processes: ~1 800 aggregate)", "source": "BASELINE.md section 2"},
            }
            nth = min(64, len(os.sched_getaffinity(0))) if args.cpu_threads < 0 else args.cpu_threads
            if nth > 1:
                each = max(100_000, args.cpu_steps // 32)
                v, dt = cpu_baseline_all_cores(each, (args.cores, args.obstacles, min_dis), nth)
                out["cpu_baseline_all_cores"] = {
                    "value": v, "unit": "env steps/s", "cores": nth, "kind": "port",
                    "sample": f"{nth} host threads x {each} steps, one oracle env each, {dt:.1f} s",
                }
        result_line = json.dumps(out)
    else:
        result_line = None
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    for e_ in envs:
        e_.close()
    return result_line


def _main_with_clean_stdout():
    """The contract is ONE JSON line on stdout.  RCCL prints a version banner through C stdio when a process group is created
    (it lands after the JSON line when stdout is a file), so everything the run itself writes to file descriptor 1 -- Python
    or C level -- is sent to stderr, and the result line is written to the real stdout at the very end."""
    import ctypes
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        line = main()
    finally:
        sys.stdout.flush()
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        os.dup2(real_stdout, 1)
        os.close(real_stdout)
    if line is not None:
        print(line, flush=True)


if __name__ == "__main__":
    _main_with_clean_stdout()
