#!/usr/bin/env python3
"""bench.py -- env steps/sec of the batched marinenav_env + IQN training loop on MI355X.

One "step" = one vector step of the hot path on one GPU: IQN act (K = 32 quantile samples) for 65 536 envs -> HIP step kernel that also
appends the transition to the replay ring -> HIP reset of the finished envs -> (every 4th vector step) one IQN gradient step (batch 256,
8 quantiles, replay 100 000; csrc/iqn_train.hip).  This is BASELINE.json configs[2], the configuration its metric is quoted on.  With
--gpus N every rank runs the same per-GPU workload on its own env shard (weak scaling, no data-path collective: configs[3]);
--shared-learner makes the IQN one data-parallel learner (configs[4]): --exchange collective = one RCCL all-reduce of the flat 143 KB
gradient per gradient step, --exchange mailbox = the exchange inside the gradient step's own launch over IPC-mapped mailboxes.

Prints ONE JSON line on rank 0, kept under 6 KB (the driver's record keeps 8 KB of tail): numbers only, 4 significant digits; what a key
means is written HERE, not in the line.

  value / ms_per_step   the MEDIAN of `windows.n` timed windows of --steps vector steps each (every window bracketed by a barrier + device
                        synchronisation on both sides, max over ranks); windows.min / .max = the slowest / fastest window's env steps/s
  roofline              the dominant kernel of the loop, the fused IQN act kernel (`iqn_qvals_split_kernel`: the float32 network on the f16 matrix
                        pipe, every f32 product as three f16 MFMA products).  launch_ms: HIP events on the launch stream around the first
                        `launches_timed` act launches of the first window (incl. the 5.6 us draw launch in front of each).  frac = the matrix
                        FLOPs the kernel ISSUES (372 x 16 384 per env) / launch_ms over the 2.5 PFLOP/s dense f16 peak; frac_algorithmic = SURVEY
                        8(d)'s 2 002 944 FLOP per env-step over the same peak.  traffic: null (not counted live); traffic_mb_profiled: the rocprofv3
                        PMC figure of the same kernel (2 x FETCH_SIZE + WRITE_SIZE per launch, profiles/), clock_ghz / kilocycles: the launch in
                        cycles of the clock `clock` measured (a box-independent figure)
  roofline_env_step     `mn_step_kernel<double, ..., APPEND>` (HBM-bound by the north-star): (406 + 328) B per env-step x envs / launch_ms over
                        8 TB/s; frac_step_only counts SURVEY 8(d)'s 406 B alone.  reset_kernel: `mn_reset_kernel` of the same vector steps, live events:
                        launch_ms, resets per launch (counted over 16 extra steps after the timed region), bytes per reset = 4 B x MT19937 words an
                        episode start consumes + the tables / pose / first observation it writes (RESET_BYTES), frac of 8 TB/s; on_critical_path false:
                        the launch ran on the env handle's own stream UNDER the next vector step's act kernel (config.resets; its launch_ms is then the
                        time beside that kernel's workgroups, not time a vector step waits for); under_act_share: the share of the run's reset launches
                        the library put there (it keeps them in front while many episodes end per vector step, e.g. right after the initial reset)
  cpu_baseline          the scalar C oracle (oracle/, kind "port") on one host core, bounded sample; cpu_all_cores: one oracle env per host thread;
                        cpu_reference_python: the reference's own MarineNavEnv.step measured where it can be imported
  clock                 mn_probe_mfma_clock (a pure f16 MFMA stream on every CU for ~50 ms) right before / after the timed region, GHz; power: rocm-smi
  also.*                other configurations timed by the same run (N = 1): config1 = BASELINE configs[1] (4 096 envs, random policy, kernels only) per
                        world size (cores_obstacles) and arithmetic: [launch-pair M env steps/s, mn_rollout T=100 M env steps/s, step kernel us];
                        reset_in_front = the loop with the episode resets in front of the act kernel instead of under it (config.resets);
                        late_curriculum = the same loop where the end of a training run lives (eps 0.05, ~2 300 episode ends per vector step: resets_per_launch;
                        under_act_share of its reset launches; .reset_in_front = that regime with the resets in front of the act kernel);
                        act_exact_f32 = the loop with the exact-f32 MFMA act kernel; shared_learner_ws1 = learner alone, grad-steps/s: one launch per
                        step, the RCCL all-reduce at world size 1 eager / inside captured 16-step graphs, the mailbox exchange inside the one-launch
                        step; act_shared_taus (opt-in: 32 taus per launch instead of per env; tiled = environments in the MFMA columns, wave = wavefront
                        per env); train_cadence = what train_iqn runs: 16 gradient steps per vector step, eps 0.05
  learner_only          back-to-back gradient steps, grad-steps/s: fused_hip (one launch per step), eager PyTorch, hipGraph of PyTorch
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_PER_ENV_STEP = {(8, 10): 406, (8, 5): 346, (4, 6): 310}   # 190 + 12 * (n_cores + n_obs), SURVEY 8(d)
APPEND_BYTES_PER_ENV_STEP = 104 + 2 * 104 + 8 + 4 + 4           # mn_step_append: obs_t row read + the transition written to the ring
HBM_PEAK_GBS = 8000.0
ACT_FLOP_PER_ENV_STEP = 2 * 32 * (64 * 208 + 208 * 64 + 64 * 64 + 64 * 9)      # 2 002 944 ("~2.0 MFLOP")
ACT_SHARED_FLOP_PER_ENV_STEP = 2 * 32 * (208 * 64 + 64 * 64 + 64 * 9)          # launch-shared taus: layer 1 is a constant of the launch
ACT_SPLIT_MFMA_FLOP = 372 * 16384            # issued by the split-f16 kernel per env: 372 v_mfma_f32_16x16x32_f16
ACT_SHARED_MFMA_FLOP = 216 * 16384
F32_MFMA_PEAK_TFLOPS = 157.3                 # dense v_mfma_f32_*_f32 peak, MI355X_MICROARCH.md
F16_MFMA_PEAK_TFLOPS = 2500.0                # dense f16 / bf16 MFMA peak, MI355X_MICROARCH.md
# HBM bytes per launch from rocprofv3 PMC passes (2 x FETCH_SIZE + WRITE_SIZE, gfx950 correction; 65 536 envs, 8 cores / 10 obstacles).  Not measured live.
PMC_TRAFFIC_MB = {"step_append_f64": 63.7, "act_split": 24.7, "reset_under_act_f64": 6.6,
                  "source": "profiles/r06_full_loop_kernel_stats.txt (scripts/assemble_profiles.py refuses a constant that disagrees with the round's profile)"}
# mn_reset_kernel, algorithmic bytes per episode start: 4 B x the MT19937 words a reset consumes on average (oracle, 10 000 resets per world size: 294 / 179 / 138
# words at (8, 10, 40 m) / (8, 5, 25 m) / (4, 6, 30 m)) + what it writes: cores nc x 24 B + obstacles no x 24 B + the fixed-point copy (nc + no) x 12 B + pose / start /
# goal / initial state 112 B + counters 8 B + first observation 104 B
RESET_BYTES = {(nc, no): 4 * w + 24 * (nc + no) + 12 * (nc + no) + 112 + 8 + 104 for (nc, no, w) in ((8, 10, 294), (8, 5, 179), (4, 6, 138))}


def sig(x, n=4):
    """Round every float in a nested structure to n significant digits (the line must stay short)."""
    if isinstance(x, float):
        return float(f"{x:.{n}g}") if x == x and abs(x) != float("inf") else None
    if isinstance(x, dict):
        return {k: sig(v, n) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [sig(v, n) for v in x]
    return x


def default_precision(learner):
    """Env-kernel arithmetic when --precision is not given: the strict float64 kernels whenever an IQN is in the loop (every float32 output within 1e-5 of the
    reference with no outliers; measured free next to the act kernel), mixed precision (SURVEY 8d's float32-SoA design point) for the kernel-only configs."""
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
    """The clock this GPU holds under f16 matrix load (C-ABI mn_probe_mfma_clock).  The same act binary runs 10-12 % slower on some boxes of the pool; with this
    in the line a reader can tell a slow box from a slow kernel (launch time x clock = the kernel's cycles, which do not depend on the box)."""
    import ctypes as C
    import torch
    from distributional_rl_navigation_amd import _capi
    out = (C.c_double * 5)()
    with torch.cuda.device(device):
        rc = _capi.lib().mn_probe_mfma_clock(C.c_double(target_ms), out, C.c_void_p(torch.cuda.current_stream(device).cuda_stream))
    return None if rc else out[1]


def rocm_smi_power():
    """Average power / power cap in W as rocm-smi reports them, if it is there and answers (never fails the run)."""
    import subprocess
    try:
        r = subprocess.run(["rocm-smi", "--showpower", "--showmaxpower", "--json"], capture_output=True, text=True, timeout=20)
        j = json.loads(r.stdout[r.stdout.index("{"):])
        card = j.get("card0", next(iter(j.values())))
        vals = [float(v) for k, v in sorted(card.items()) if "power" in k.lower() and str(v).replace(".", "", 1).isdigit()]
        return vals or None
    except Exception:
        return None


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


def also_legs(args, env, agent, obs, device, total_timesteps, dist_up):
    """The other configurations README / DESIGN quote, timed by the same run so that they are driver-timed numbers too (see the module docstring).  Same
    synthetic worlds, same agent (its replay ring is full by now).  Order = order in the line: what README leads with comes LAST."""
    import torch
    import torch.distributed as dist
    from distributional_rl_navigation_amd.iqn.fused_act import act_context
    from distributional_rl_navigation_amd.marinenav_env.vec_env import VecMarineNavEnv
    n = env.n_envs
    out = {}
    ctx = act_context(agent.qnetwork_local)
    M = 1e-6

    def loop_step(o, eps=None, cvar=args.cvar):
        e = agent.linear_eps(total_timesteps) if eps is None else eps
        return agent.vec_step(env, o, e, cvar, per_iter=n)[0]

    # (c) configs[1]: 4 096 envs, random policy, kernels only -- per world size and arithmetic
    n1 = 4096

    def config1(cores, obstacles, min_dis, precision, single_steps=500):
        e1 = VecMarineNavEnv(n1, seed=0, device=device, precision=precision)
        e1.set_attrs(num_cores=cores, num_obs=obstacles, min_start_goal_dis=min_dis)
        e1.reset()
        gen = torch.Generator(device=device); gen.manual_seed(0)

        def pair(_):
            e1.step(torch.randint(0, 9, (n1,), device=device, dtype=torch.int32, generator=gen))
            return e1.reset_done()
        dt, _ = _timed(device, pair, single_steps, 100, None, lambda: e1.profile_begin(50))
        k_ms, _ = e1.profile_end()
        T, ctr = 100, [0]

        def roll_launch(_):
            e1.rollout(T, action_seed=0, first_step=ctr[0], trace=("obs", "reward", "done"))
            ctr[0] += T
        dt_r, _ = _timed(device, roll_launch, 10, 2, None)
        e1.close()
        return [n1 * single_steps / dt * M, n1 * T * 10 / dt_r * M, 1e3 * k_ms]
    worlds = {}
    for (nc, no, md) in ((8, 10, 40.0), (8, 5, 25.0), (4, 6, 30.0)):
        worlds[f"{nc}_{no}"] = {prec: config1(nc, no, md, prec) for prec in ("f64", "mixed")}
    out["config1"] = dict(worlds, envs=n1, row="[launch pairs M/s, mn_rollout T=100 M/s, step kernel us]")
    # (a0) the loop with the episode resets IN FRONT of the act kernel (the default runs them under it: IQNAgent.reset_under_act)
    steps, warm = max(20, args.steps // 2), 10
    if agent.reset_under_act:
        agent.reset_under_act = False
        dt, obs = _timed(device, loop_step, steps, warm, obs)
        agent.reset_under_act = True
        out["reset_in_front"] = {"value": n * steps / dt, "ms_per_step": 1e3 * dt / steps}
    # (a) exact-f32 act kernel
    ctx.set_variant(0)
    dt, obs = _timed(device, loop_step, steps, warm, obs, lambda: ctx.profile_begin(min(steps, 50)))
    act_ms, _ = ctx.profile_end()
    ctx.set_variant(args.act_variant)
    alg_tf = ACT_FLOP_PER_ENV_STEP * n / (act_ms * 1e-3) / 1e12 if act_ms > 0 else None
    out["act_exact_f32"] = {"value": n * steps / dt, "ms_per_step": 1e3 * dt / steps, "act_launch_ms": act_ms,
                            "frac_f32_mfma": alg_tf / F32_MFMA_PEAK_TFLOPS if alg_tf else None}
    # (d) learner alone: one launch per step; RCCL all-reduce at world size 1; the mailbox exchange inside the step's launch
    def learner_rate(reps=400):
        for _ in range(10):
            agent.train_from_memory()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(reps):
            agent.train_from_memory()
        torch.cuda.synchronize(device)
        return reps / (time.perf_counter() - t0)

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
    sl = {}
    was = agent.distributed
    agent.distributed = False
    sl["no_group"] = learner_rate()
    sl["no_group_graphed"] = graphed_rate()
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
        sl["nccl_ws1"] = learner_rate()
        sl["nccl_ws1_graphed"] = graphed_rate()
        sl["allreduce_us"] = 1e6 * (1.0 / sl["nccl_ws1"] - 1.0 / sl["no_group"])
        sl["allreduce_us_graphed"] = 1e6 * (1.0 / sl["nccl_ws1_graphed"] - 1.0 / sl["no_group_graphed"])
        agent.exchange = "mailbox"
        try:
            sl["mailbox_ws1"] = learner_rate()      # the exchange inside the ONE launch of the step
            sl["mailbox_us"] = 1e6 * (1.0 / sl["mailbox_ws1"] - 1.0 / sl["no_group"])
            sl["mailbox_launches"] = agent._fused.launches_per_step()
            agent.one_launch_step = False
            sl["mailbox_ws1_two_launches"] = learner_rate()
            agent.one_launch_step = None
            sl["mailbox_memory"] = agent._fused._mailbox.memory_kind()
        finally:
            agent.exchange = "collective"
            agent.one_launch_step = None
        sl["timeouts"] = agent._fused.timeouts()
        sl["xcd_misplaced"] = agent._fused.xcd_misplaced()
    except Exception as e:      # the line must still come out if RCCL cannot initialise on this box
        sl["error"] = repr(e)[:160]
    finally:
        agent.distributed = was
        if made:
            torch.cuda.synchronize(device)
            dist.destroy_process_group()
    out["shared_learner_ws1"] = sl
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
    out["act_shared_taus"] = {"value": n * steps / dt, "ms_per_step": 1e3 * dt / steps, "form": "tiled" if n >= 65536 else "wave",
                              "launch_ms": act_ms, "launch_ms_wave_form": act_ms_wave,
                              "frac_algorithmic_remaining": ACT_SHARED_FLOP_PER_ENV_STEP * rate / F16_MFMA_PEAK_TFLOPS if rate else None,
                              "frac_algorithmic_full_network": ACT_FLOP_PER_ENV_STEP * rate / F16_MFMA_PEAK_TFLOPS if rate else None,
                              "frac_issued": ACT_SHARED_MFMA_FLOP * rate / F16_MFMA_PEAK_TFLOPS if rate else None}
    # (a'') the main loop in the regime the END of a training run is in: eps 0.05 and ~2 300 episode ends per vector step (scripts/soak.py: the curriculum's steady
    # state; the headline's eps ~ 1 start has ~300).  Forced here with a short episode limit and episode ages spread uniformly (no burst): 65 536 / 40 time-outs
    # per step on top of the collisions / goals of the untrained policy.  Same kernels, same cadence as `value`.
    if agent.reset_under_act:
        import numpy as np
        L_late = 40
        env.join_reset()
        env.params.max_episode_steps = L_late
        env.set_attrs()
        env.set_state(episode_timesteps=np.random.RandomState(0).randint(0, L_late, size=n))
        late_step = lambda o: agent.vec_step(env, o, 0.05, args.cvar, per_iter=n)[0]
        steps = max(100, args.steps)      # (37 ms: the driver's --steps 20 would be 7 ms of timed work)
        rl0 = list(env.reset_launches)
        dt, obs = _timed(device, late_step, steps, 40, obs)
        rl1 = [b - a for a, b in zip(rl0, env.reset_launches)]
        agent.reset_under_act = False
        dt_f, obs = _timed(device, late_step, steps, 10, obs)
        agent.reset_under_act = True
        env.join_reset()
        cnt = []
        for _ in range(8):
            a = agent.act_batch(obs, 0.05, args.cvar)
            env.step(a); cnt.append(env.last_done_count()); obs = env.reset_done()
        env.params.max_episode_steps = 1000
        env.set_attrs()
        out["late_curriculum"] = {"value": n * steps / dt, "ms_per_step": 1e3 * dt / steps, "eps": 0.05, "resets_per_launch": sum(cnt) / len(cnt),
                                  "under_act_share": rl1[1] / max(1, rl1[0] + rl1[1]), "reset_in_front": n * steps / dt_f}
    # (b) the cadence that trains
    ue, gs = agent.UPDATE_EVERY, agent.grad_steps_per_update
    agent.UPDATE_EVERY, agent.grad_steps_per_update = 1, 16
    steps = max(20, args.steps // 2)
    train_step = lambda o: agent.vec_step(env, o, 0.05, args.cvar, train_every=1, per_iter=n)[0]
    dt, act_ms = shared_leg(train_step, steps, 10)
    out["train_cadence_shared_taus"] = {"value": n * steps / dt, "ms_per_step": 1e3 * dt / steps, "grad_steps_per_sec": 16 * steps / dt, "act_launch_ms": act_ms}
    dt, obs = _timed(device, train_step, steps, 10, obs)
    out["train_cadence"] = {"value": n * steps / dt, "ms_per_step": 1e3 * dt / steps, "grad_steps_per_sec": 16 * steps / dt,
                            "grad_steps_per_vector_step": 16, "launches_per_grad_step": agent._fused.launches_per_step()}
    agent.UPDATE_EVERY, agent.grad_steps_per_update = ue, gs
    agent.check_learner()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--windows", type=int, default=5, help="timed windows of --steps vector steps each; `value` is the median window")
    ap.add_argument("--envs", type=int, default=65536, help="envs per GPU")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--replay", type=int, default=100_000)
    ap.add_argument("--cores", type=int, default=8)
    ap.add_argument("--obstacles", type=int, default=10)
    ap.add_argument("--shared-learner", action="store_true", help="one IQN over all ranks (configs[4])")
    ap.add_argument("--exchange", default="collective", choices=["collective", "mailbox"],
                    help="with --shared-learner: collective = RCCL all-reduce of the flat gradient per gradient step; mailbox = the exchange inside the gradient "
                         "step's own launch over IPC-mapped mailboxes (iqn/mailbox.py)")
    ap.add_argument("--ranks-per-gpu", type=int, default=int(os.environ.get("MN_BENCH_RANKS_PER_GPU", "1")),
                    help="tests: this many ranks share one GPU (rank r on device r // R; the group is gloo because RCCL refuses two ranks per device, "
                         "and every rank plans its fused launches for 1 / R of the CUs)")
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
    ap.add_argument("--no-also", action="store_true", help="skip the extra driver-timed legs (`also`) after the main timed region")
    ap.add_argument("--act-variant", type=int, default=2, choices=(0, 2), help="acting kernel: 2 = split-f16 MFMA at float32 accuracy (default), 0 = exact-f32 v_mfma_f32_16x16x4_f32")
    ap.add_argument("--separate-append", action="store_true", help="mn_step + mn_replay_append as two launches instead of the fused mn_step_append")
    ap.add_argument("--reset-in-front", action="store_true", help="episode resets in front of the act kernel (mn_reset_done) instead of under it "
                                                                  "(mn_reset_done_async + late rows; IQNAgent.reset_under_act, the default)")
    ap.add_argument("--reset-under-act-max", type=int, default=None, help="mn_set_reset_under_act_max: resets go under the act kernel while the launches' decaying peak of "
                                                                          "episode ends per vector step is at most this (default: the library's, 5000)")
    ap.add_argument("--graph-train", action="store_true", help="the gradient steps of a training event as one captured hipGraph (IQNAgent.use_fused_graph)")
    ap.add_argument("--shared-taus", action="store_true", help="one set of 32 taus per act LAUNCH instead of per env (IQNAgent.shared_taus; opt-in, "
                                                               "timed by the default run as also.act_shared_taus)")
    ap.add_argument("--no-clock-probe", action="store_true", help="skip the two ~50 ms MFMA clock probes around the timed region (cleaner rocprofv3 tables)")
    args = ap.parse_args()
    if args.precision is None:
        args.precision = default_precision(learner=not args.no_learner)

    import torch
    import torch.distributed as dist
    from distributional_rl_navigation_amd import _capi
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
    rpg = max(1, args.ranks_per_gpu)
    device = torch.device(f"cuda:{local_rank // rpg}")
    torch.cuda.set_device(device)
    if rpg > 1:      # ranks sharing a GPU: each plans its fused launches (which wait for their own workgroups) for its share of the CUs
        _capi.lib().mn_iqn_train_set_cu_limit(max(8, torch.cuda.get_device_properties(device).multi_processor_count // rpg - 8))
    # under torch.distributed.run: RCCL, even for 1 rank; a plain `python bench.py --shared-learner` (N = 1) forms a
    # single-rank RCCL group itself so that configs[4]'s gradient all-reduce executes
    use_dist = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ or args.shared_learner
    backend = "gloo" if rpg > 1 else "nccl"
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        kw = {} if backend == "gloo" else {"device_id": device}
        if "MASTER_ADDR" in os.environ and "RANK" in os.environ:
            dist.init_process_group(backend, **kw)
        else:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            dist.init_process_group(backend, init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, **kw)

    n = args.envs
    min_dis = {4: 30.0, 6: 35.0, 8: 40.0}.get(args.cores, 25.0)
    env = VecMarineNavEnv(n, seed=0, first_index=rank * n, device=device, precision=args.precision, step_lanes=args.lanes if args.lanes in (1, 2, 4, 8) else 0,
                          rollout_lanes=args.lanes if args.lanes != 1 else 0)
    env.set_attrs(num_cores=args.cores, num_obs=args.obstacles, min_start_goal_dis=min_dis, N=args.robot_n)
    obs = env.reset()
    agent = None
    if not args.no_learner:
        agent = IQNAgent(26, 9, BATCH_SIZE=args.batch, BUFFER_SIZE=args.replay, device=device,
                         seed=100 if args.shared_learner else 100 + rank, learning_starts=0,
                         distributed=args.shared_learner and use_dist, act_chunk=args.act_chunk,
                         UPDATE_EVERY=args.update_every, rank=rank if args.shared_learner else 0)
        agent.grad_steps_per_update = args.grad_steps
        # (two ranks on one GPU exchange over gloo, i.e. through the host: not capturable -- eager events there)
        agent.use_fused_graph = args.graph_train and not (args.shared_learner and args.exchange == "collective" and args.ranks_per_gpu > 1)
        agent.shared_taus = args.shared_taus
        agent.reset_under_act = not args.reset_in_front
        agent.exchange = args.exchange
        agent.use_fused_train = not args.torch_train
        if args.torch_act:
            agent.use_fused_act = False
        if args.no_train_graph:
            agent.use_train_graph = False
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

    def one_step(o):
        if agent is None:
            a = torch.randint(0, 9, (n,), device=device, dtype=torch.int32, generator=gen)
            env.step(a)
            return env.reset_done()
        eps = agent.linear_eps(total_timesteps) if args.eps is None else args.eps
        eps_seen.append(eps)
        return agent.vec_step(loop_env, o, eps, args.cvar, per_iter=n * world)[0]

    def fence():
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
    resets_note = None
    if agent is not None and agent.reset_under_act and fused and not roll:
        # the under-act resets need the reset launch to run BESIDE the act kernel (own hardware queue, room on the CUs): make sure of it before the timed
        # region, the way IQNAgent.learn_vec does -- a dozen more untimed steps with every reset forced under the act kernel (iqn/agent.py: UnderActGuard); if
        # a late row's wait ran out the loop goes back to resets in front and the line says so
        from distributional_rl_navigation_amd.iqn.agent import UnderActGuard
        if args.reset_under_act_max is not None:
            env.set_reset_under_act_max(args.reset_under_act_max)
        guard = UnderActGuard(agent, env, preflight=12, poll_every=0)
        for i in range(12):
            obs = run_steps(1, obs)
            guard.after_step(i)
        guard.close()
        if guard.fallback is not None:
            resets_note = "in_front_of_act (the reset launch did not run beside the act kernel on this box: late rows timed out)"
    g0 = agent.grad_steps if agent else 0
    clock_before = gpu_clock_probe(device) if (rank == 0 and not args.no_clock_probe) else None      # ~50 ms of matrix load, outside the timed region
    # HIP-event pairs are recorded around the first n_prof act / step / reset launches of the FIRST window; not around all of them, because the event records
    # of a vector step cost ~18 us of stream time (measured: 1.079 ms/step with 200 instrumented steps, 1.067 with 50, 1.061 with 1)
    n_prof = min(args.steps // max(1, roll), int(os.environ.get("MN_BENCH_NPROF", "50")))
    env.profile_begin(n_prof)
    if fused:
        act_context(agent.qnetwork_local).profile_begin(n_prof)
    window_s = []
    rl_before = list(env.reset_launches)
    for w in range(max(1, args.windows)):
        fence()
        t0 = time.perf_counter()
        obs = run_steps(args.steps, obs)
        fence()
        el = time.perf_counter() - t0
        if use_dist:      # the slowest rank's clock
            t = torch.tensor([el], dtype=torch.float64, device="cpu" if backend == "gloo" else device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        window_s.append(el)
    elapsed = statistics.median(window_s)
    reset_ms, reset_launches = env.profile_reset_end()
    rl_timed = [b - a for a, b in zip(rl_before, env.reset_launches)]      # reset launches of the timed windows: [in front of, under] the act kernel
    step_kernel_ms, launches = env.profile_end()
    act_ms, act_launches = 0.0, 0
    if fused:
        act_ms, act_launches = act_context(agent.qnetwork_local).profile_end()
    grad_steps = (agent.grad_steps - g0) if agent else 0
    clock_after = gpu_clock_probe(device) if (rank == 0 and not args.no_clock_probe) else None
    # episodes that end per vector step (the reset kernel's work): counted over 16 more steps (mn_last_done_count synchronises: outside the timed region)
    resets_per_step = None
    if not roll:
        cnt = []
        for _ in range(16):
            if agent is None:
                env.step(torch.randint(0, 9, (n,), device=device, dtype=torch.int32, generator=gen))
                cnt.append(env.last_done_count())
                obs = env.reset_done()
            else:
                a = agent.act_batch(obs, agent.linear_eps(total_timesteps) if args.eps is None else args.eps, args.cvar)
                env.step(a)
                cnt.append(env.last_done_count())
                obs = env.reset_done()
        resets_per_step = sum(cnt) / len(cnt)

    all_reduce_ms = None
    if use_dist and args.shared_learner:      # the shared learner's collective alone: 35 785-float bucket, back to back (max over ranks)
        bucket = torch.zeros(35785, dtype=torch.float32, device="cpu" if backend == "gloo" else device)
        for _ in range(20):
            dist.all_reduce(bucket)
        torch.cuda.synchronize(device)
        t1 = time.perf_counter()
        for _ in range(200):
            dist.all_reduce(bucket)
        torch.cuda.synchronize(device)
        tt = torch.tensor([(time.perf_counter() - t1) / 200 * 1e3], dtype=torch.float64, device=bucket.device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        all_reduce_ms = float(tt.item())
    timeouts = None
    if agent is not None and getattr(agent, "_fused", None) is not None:
        timeouts = agent._fused.timeouts()      # bounded waits that ran out (0: every step updated; a shared learner's ranks agree)
    late_to = None
    if agent is not None and agent.reset_under_act and agent.use_fused_act:
        from distributional_rl_navigation_amd.iqn.fused_act import late_timeouts
        env.join_reset()
        late_to = late_timeouts(agent.qnetwork_local)      # act rows taken before their reset had finished (0: the reset launch ran beside the act kernel)
    also = {}
    if world == 1 and not args.no_also and agent is not None and fused and not roll and not args.torch_train and not args.torch_act:
        also = also_legs(args, env, agent, obs, device, total_timesteps, use_dist)

    # learner alone (outside the timed region): back-to-back IQN grad steps (sample + train), batch 256, 8 quantiles
    learner_only = {}
    if agent is not None and len(agent.memory) > agent.BATCH_SIZE and not args.no_learner_only:
        was_fused = agent.use_fused_train
        for mode in ("fused_hip", "eager", "hipgraph"):       # the torch modes last: they advance torch's own Adam state
            if mode == "hipgraph" and agent.distributed:
                continue
            agent.use_fused_train = (mode == "fused_hip")
            agent.use_train_graph = (mode == "hipgraph")
            reps = 400 if mode == "fused_hip" else 50
            for _ in range(5):
                agent.train_from_memory()
            torch.cuda.synchronize(device)
            t1 = time.perf_counter()
            for _ in range(reps):
                agent.train_from_memory()
            torch.cuda.synchronize(device)
            learner_only[mode] = reps / (time.perf_counter() - t1)
        agent.use_fused_train = was_fused

    result_line = None
    if rank == 0:
        bytes_step = BYTES_PER_ENV_STEP.get((args.cores, args.obstacles), 190 + 12 * (args.cores + args.obstacles))
        fused_append = agent is not None and not args.separate_append
        bytes_per = bytes_step + (APPEND_BYTES_PER_ENV_STEP if fused_append else 0)
        per_launch = n * max(1, roll)      # one launch processes n env-steps (single step) or n * T env-steps (mn_rollout)
        achieved = bytes_per * per_launch / (step_kernel_ms * 1e-3) / 1e9 if step_kernel_ms > 0 else 0.0
        headline = (n, args.cores, args.obstacles) == (65536, 8, 10)
        rates = sorted(n * world * args.steps / s for s in window_s)
        kern = "mn_rollout_kernel" if roll else ("mn_step_kernel<%s,APPEND>" if fused_append else "mn_step_kernel<%s>") % ("double" if args.precision == "f64" else "float")
        env_roof = {
            "kernel": kern, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "frac_step_only": (bytes_step * per_launch / (step_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if step_kernel_ms > 0 else None,
            "bytes_per_env_step": bytes_per, "traffic": None,
            "traffic_mb_profiled": PMC_TRAFFIC_MB.get(("step_append_" if fused_append else "step_") + args.precision) if (headline and not roll) else None,
            "launch_ms": step_kernel_ms, "launches_timed": launches, "env_steps_per_launch": per_launch,
        }
        rb = RESET_BYTES.get((args.cores, args.obstacles))
        if reset_launches and reset_ms > 0 and resets_per_step:
            gbs = (rb * resets_per_step / (reset_ms * 1e-3) / 1e9) if rb else None
            under = agent is not None and agent.reset_under_act
            rl = rl_timed
            env_roof["reset_kernel"] = {"kernel": "mn_reset_under_act_kernel" if under else "mn_reset_kernel", "on_critical_path": not under,
                                 "under_act_share": (rl[1] / max(1, rl[0] + rl[1])) if under else 0.0, "launch_ms": reset_ms, "launches_timed": reset_launches, "resets_per_launch": resets_per_step,
                                 "bytes_per_reset": rb, "achieved": gbs, "frac": gbs / HBM_PEAK_GBS if gbs else None,
                                 "traffic_mb_profiled": PMC_TRAFFIC_MB["reset_under_act_f64"] if (headline and args.precision == "f64" and under) else None}
        out = {
            "metric": "env steps/sec (whole node) at 65 536 envs; IQN grad-steps/sec",
            "value": n * world * args.steps / elapsed,
            "unit": "env steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "windows": {"n": len(window_s), "min": rates[0], "median": n * world * args.steps / elapsed, "max": rates[-1]},
            "config": {
                "workload": ((f"step kernel only, random policy, mn_rollout {roll} steps per launch" if roll else "step kernel only, random policy, launch pair per step")
                             if agent is None else f"{n} envs/GPU + IQN training (act K=32, 8 quantiles, replay {args.replay}, batch {args.batch}, "
                             f"{args.grad_steps} grad step(s) every {args.update_every} vector steps)"),
                "envs_per_gpu": n, "n_cores": args.cores, "n_obstacles": args.obstacles, "precision": args.precision,
                "learner": "none" if agent is None else ("shared" if args.shared_learner else "independent"),
                "exchange": (args.exchange if args.shared_learner else None), "cvar": args.cvar,
                "process_group": dist.get_backend() if use_dist else None, "ranks_per_gpu": rpg,
                "eps": None if agent is None else (sum(eps_seen[-args.steps:]) / max(1, len(eps_seen[-args.steps:]))),
                "act_variant": args.act_variant, "taus": "shared" if args.shared_taus else "per_env",
                "resets": None if agent is None else (resets_note or ("under_next_act" if agent.reset_under_act else "in_front_of_act")),
                "late_row_timeouts": late_to,
                "launches_per_grad_step": agent._fused.launches_per_step() if (agent is not None and getattr(agent, "_fused", None) is not None) else None,
                "ablation": bool(_capi.lib().mn_build_info() & 1),
            },
            "all_reduce_ms": all_reduce_ms,
            "timeouts": timeouts,
            "roofline_env_step": env_roof,
        }
        if fused and act_ms > 0:
            alg_tf = ACT_FLOP_PER_ENV_STEP * n / (act_ms * 1e-3) / 1e12
            if args.act_variant == 2:
                issued = ACT_SHARED_MFMA_FLOP if args.shared_taus else ACT_SPLIT_MFMA_FLOP
                tf, peak = issued * n / (act_ms * 1e-3) / 1e12, F16_MFMA_PEAK_TFLOPS
                kern = "iqn_qvals_tiled_kernel" if (args.shared_taus and n >= 65536) else "iqn_qvals_split_kernel"
            else:
                tf, peak = alg_tf, F32_MFMA_PEAK_TFLOPS
                kern = "iqn_qvals_kernel"
            clocks = [c for c in (clock_before, clock_after) if c]
            ghz = sum(clocks) / len(clocks) if clocks else None
            out["roofline"] = {
                "kernel": kern, "bound": "mfma", "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak, "frac_algorithmic": alg_tf / peak,
                "traffic": None, "traffic_mb_profiled": PMC_TRAFFIC_MB.get("act_split") if (args.act_variant == 2 and n == 65536 and not args.shared_taus) else None,
                "launch_ms": act_ms, "launches_timed": act_launches, "env_steps_per_launch": n,
                "clock_ghz": ghz, "kilocycles": act_ms * 1e-3 * ghz * 1e6 if ghz else None,
            }
        else:
            out["roofline"] = env_roof
        if args.cpu_steps > 0 and world == 1:      # reported baseline: rank 0 at N = 1 only
            v, dt = cpu_baseline(args.cpu_steps, (args.cores, args.obstacles, min_dis))
            out["cpu_baseline"] = {"value": v, "unit": "env steps/s", "cores": 1, "kind": "port",
                                   "sample": f"{args.cpu_steps} steps, 1 env, C oracle, {dt:.1f} s"}
            nth = min(64, len(os.sched_getaffinity(0))) if args.cpu_threads < 0 else args.cpu_threads
            if nth > 1:
                each = max(100_000, args.cpu_steps // 32)
                v, dt = cpu_baseline_all_cores(each, (args.cores, args.obstacles, min_dis), nth)
                out["cpu_all_cores"] = {"value": v, "cores": nth, "host_cpus": os.cpu_count()}
            out["cpu_reference_python"] = 293
        out["clock"] = {"before": clock_before, "after": clock_after, "power": rocm_smi_power()}
        # what README leads with goes LAST: the driver's record keeps the tail of the line
        out["also"] = also
        out["learner_only"] = learner_only
        out["grad_steps_per_sec"] = grad_steps * (1 if args.shared_learner else world) / sum(window_s)
        result_line = json.dumps(sig(out), separators=(",", ":"))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    env.close()
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
