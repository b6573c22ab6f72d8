"""Training driver: counterpart of the reference's train_IQN_model.py (:15-179) for the batched path.

Same JSON config (`config/config_IQN.json` schema: agent, seed (list -> grid), total_timesteps,
eval_freq, save_dir), same per-trial outputs (`trial_config.json`, `training_schedule.json`,
`eval_config.json`, `greedy_evaluations.npz`, `adaptive_evaluations.npz`, `network_params.pth`,
`constructor_params.json`).  One process per GPU: launch with torch.distributed.run for several
GPUs (independent learners per rank by default, `--shared-learner` for one IQN with an RCCL
gradient all-reduce).

    python -m distributional_rl_navigation_amd.train_iqn -C config_IQN.json [--n-envs 4096]

Cadence.  The reference does one batch-32 gradient step per 4 env steps (replay ratio 8 sampled per generated
transition).  With 65 536 envs a vector step IS 65 536 env steps, so that ratio is out of reach (16 384 gradient
steps per vector step); the batched loop instead spends the reference's LEARNER budget (750 000 x 32 samples =
93 750 gradient steps of batch 256) at G gradient steps per vector step (16 per 65 536 envs, one per 4 096: replay ratio
0.06 either way) and rescales every run-fraction cadence (exploration ramp, curriculum, evaluations) -- `plan_cadence`.

Envs per GPU.  Default 4 096 (round 6): the same env steps and learner budget cut into 93 760 vector steps of ONE
gradient step each, ring 100 000 = 24 vector steps of history.  Of the cells swept (profiles/r06_learning_curve.txt: envs
2 048 ... 65 536 x ring 100 k ... 16 M, 12-24 seeds each, 252 runs) it is the one whose LAST evaluation sits on the reference's
own final evaluation without best-checkpoint selection -- 26.1 +- 1.4 of 30 successes, mean return 68.0 +- 7.1 over 24
runs (reference: 26, 69.25) -- at 9.6 s per run on one MI355X; --n-envs 65536 (the bench's configuration: 5 860 vector
steps of 16 gradient steps) runs 6.4 s and ends 1 success / 6 return points lower (25.0 +- 1.8, 62.3 +- 11.0 over 24 runs).
"""
import argparse
import itertools
import json
import os
from datetime import datetime

import numpy as np


def trial_params(params):
    """train_IQN_model.py:52-65: list-valued keys expand to a Cartesian grid."""
    if isinstance(params, (str, int, float)):
        return [params]
    if isinstance(params, list):
        return params
    if isinstance(params, dict):
        keys, vals = zip(*params.items())
        return [dict(zip(keys, mix)) for mix in itertools.product(*[trial_params(v) for v in vals])]
    raise TypeError("Parameter type is incorrect.")


TRAINING_SCHEDULE = dict(timesteps=[0, 1000000, 2000000], num_cores=[4, 6, 8], num_obstacles=[6, 8, 10],
                         min_start_goal_dis=[30.0, 35.0, 40.0])   # train_IQN_model.py:86-90


def create_eval_configs(device, seed=348):
    """train_IQN_model.py:123-148: 30 evaluation worlds from ONE RNG stream (seed 348), fixed start/goal,
    10 x (4 cores, 6 obstacles), 10 x (6, 8), 10 x (8, 10).  Bit-identical to the reference's."""
    from .marinenav_env.env import MarineNavEnv
    env = MarineNavEnv(seed=seed, device=device)
    env.obs_r_range = [1, 3]
    env.reset_start_and_goal = False
    env.start = np.array([5.0, 5.0])
    env.goal = np.array([45.0, 45.0])
    cfg, count = {}, 0
    for nc, no in ((4, 6), (6, 8), (8, 10)):
        for _ in range(10):
            env.num_cores, env.num_obs = nc, no
            env.reset()
            cfg[f"env_{count}"] = env.episode_data()
            count += 1
    env.close()
    return cfg


def plan_cadence(total_timesteps, eval_freq, n_envs_total, batch, ref_batch=32, ref_update_every=4,
                 ref_target_interval=10_000, grad_steps_per_vector_step=None, total_grad_steps=None, n_evals=None):
    """Translate the reference's env-step cadences (config_IQN.json + agent.py defaults: 3 M timesteps, one batch-32
    gradient step every 4 env steps, target copy every 10 000 learning steps, evaluation every 10 000) into the
    batched loop's units.

    The reference consumes total_timesteps / 4 gradient steps x 32 samples.  A vector step is n_envs_total env steps
    at once, so the batched loop cannot keep the reference's replay ratio (that would be n_envs_total / 4 gradient
    steps per vector step); it keeps the reference's LEARNER budget instead -- the same number of sampled transitions,
    total_grad_steps = total_timesteps / 4 * 32 / batch -- and spreads it over as many vector steps as the chosen
    gradient-steps-per-vector-step G needs.  Everything that the reference expresses as a fraction of the run
    (exploration ramp, curriculum stages, evaluation points) keeps its fraction.
    Returns a dict; `replay_ratio` = sampled transitions per generated env step (reference: 8)."""
    ref_grad_steps = total_timesteps // ref_update_every
    if total_grad_steps is None:
        total_grad_steps = max(1, int(round(ref_grad_steps * ref_batch / batch)))
    if grad_steps_per_vector_step is None:
        # keep the learner at roughly half of the GPU time: one fused grad step ~ 57 us, one vector step ~ 16 ns / env
        grad_steps_per_vector_step = int(min(32, max(1, round(n_envs_total / 65536 * 16))))
    G = int(grad_steps_per_vector_step)
    vector_steps = int(np.ceil(total_grad_steps / G))
    if n_evals is None:
        n_evals = int(min(30, max(1, total_timesteps // max(1, eval_freq))))
    plan = dict(
        vector_steps=vector_steps, grad_steps_per_vector_step=G, total_grad_steps=vector_steps * G,
        reference_grad_steps=ref_grad_steps, reference_samples=ref_grad_steps * ref_batch, samples=vector_steps * G * batch,
        env_steps=vector_steps * n_envs_total,
        # target copy every 10 000 learning steps = 2 500 gradient steps x 32 samples -> same number of samples
        target_sync_grad_steps=max(50, int(round(ref_target_interval / ref_update_every * ref_batch / batch))),
        eval_every_vector_steps=max(1, vector_steps // n_evals), n_evals=n_evals,
        # env.total_timesteps counts vector steps per env; the curriculum (and eps) see reference-scaled time
        timestep_scale=total_timesteps / vector_steps,
        replay_ratio=G * batch / n_envs_total, reference_replay_ratio=ref_batch / ref_update_every)
    return plan


def run_trial(device, params, n_envs, rank=0, world=1, shared=False, batch=256, replay=100_000, verbose=True,
              grad_steps=None, torch_train=False, total_grad_steps=None, n_evals=None, cvar=1.0, precision="f64",
              exchange="collective", shared_taus=False, target_sync_mult=1.0, final_eps=0.05, eval_adaptive=True, n_step=1):
    """train_IQN_model.py:74-121 on the vector env.  `params` is one trial of the reference's config grid
    (seed, total_timesteps, eval_freq, save_dir); see `plan_cadence` for how its env-step cadences map to vector steps.
    `precision`: the env kernels' arithmetic.  "f64" (default: every float32 output within 1e-5 of the reference, no
    outliers; measured free while an IQN acts in the loop) or "mixed" (float32 field / sonar decisions).
    `exchange` (shared learner): "collective" = RCCL all-reduce of the flat gradient, "mailbox" = the exchange inside the gradient step's launch (iqn/mailbox.py).
    `shared_taus`: acting draws its 32 quantile fractions once per act launch instead of once per env (opt-in: a different random variable from the
    reference's per-call draw, model.py:149; A/B on learning in profiles/; the learner's taus are untouched).
    `target_sync_mult`, `final_eps`: study knobs (scripts/learning_curve.py) -- the target network is copied every target_sync_mult x the planned number of gradient
    steps; the exploration floor (agent.py: 0.05); `n_step`: the agent's n-step returns (agent.py:12-29; the reference's scripts use 1; > 1 takes the step + append launch pair
    instead of the fused mn_step_append).  `eval_adaptive` = False skips the adaptive-CVaR evaluation at the evaluation points (the reference runs both)."""
    import torch
    from .iqn.agent import IQNAgent
    from .marinenav_env.vec_env import VecMarineNavEnv

    exp_dir = os.path.join(params["save_dir"], "training_" + params["training_time"], "seed_" + str(params["seed"]))
    if world > 1 and not shared:
        exp_dir = os.path.join(exp_dir, f"rank_{rank}")
    writer = rank == 0 or not shared
    total = n_envs * world
    plan = plan_cadence(params["total_timesteps"], params["eval_freq"], total, batch,
                        grad_steps_per_vector_step=grad_steps, total_grad_steps=total_grad_steps, n_evals=n_evals)
    if plan["total_grad_steps"] * batch < 0.1 * plan["reference_samples"]:
        raise ValueError(f"planned learner budget ({plan['total_grad_steps']} grad steps x {batch}) is more than 10x below the "
                         f"reference's ({plan['reference_grad_steps']} x 32): raise --total-grad-steps")
    if writer:
        os.makedirs(exp_dir, exist_ok=True)
        with open(os.path.join(exp_dir, "trial_config.json"), "w+") as f:
            json.dump(dict(params, batched=dict(plan, n_envs=n_envs, world=world, batch=batch, replay=replay)), f)
        with open(os.path.join(exp_dir, "training_schedule.json"), "w+") as f:
            json.dump(TRAINING_SCHEDULE, f)
        if verbose:
            print(f"[train_iqn] {plan['vector_steps']} vector steps x {total} envs = {plan['env_steps']:.3g} env steps; "
                  f"{plan['total_grad_steps']} grad steps of batch {batch} ({plan['grad_steps_per_vector_step']} per vector step; "
                  f"reference: {plan['reference_grad_steps']} of 32); replay ratio {plan['replay_ratio']:.3g} sampled / generated "
                  f"transition (reference {plan['reference_replay_ratio']:.0f}); target copy every {plan['target_sync_grad_steps']} "
                  f"grad steps; evaluation every {plan['eval_every_vector_steps']} vector steps")

    train_env = VecMarineNavEnv(n_envs, seed=params["seed"], first_index=rank * n_envs, schedule=TRAINING_SCHEDULE,
                                timestep_scale=plan["timestep_scale"], device=device, precision=precision)
    eval_config = create_eval_configs(device)
    if writer:
        with open(os.path.join(exp_dir, "eval_config.json"), "w+") as f:
            json.dump(eval_config, f)
    eval_env = VecMarineNavEnv(len(eval_config), device=device, precision=precision) if writer else None

    agent = IQNAgent(26, 9, n_step=n_step, BATCH_SIZE=batch, BUFFER_SIZE=replay, device=device,
                     seed=params["seed"] + 100 + (0 if shared else rank), distributed=shared and world > 1,
                     UPDATE_EVERY=1, learning_starts=0, rank=rank if shared else 0, final_eps=final_eps)
    agent.grad_steps_per_update = plan["grad_steps_per_vector_step"]
    agent.target_sync_grad_steps = max(1, int(round(plan["target_sync_grad_steps"] * target_sync_mult)))
    agent.exchange = exchange
    agent.shared_taus = bool(shared_taus)
    if torch_train:
        agent.use_fused_train = False          # PyTorch autograd + Adam instead of csrc/iqn_train.hip
    agent.learn_vec(total_vector_steps=plan["vector_steps"], train_env=train_env, eval_env=eval_env, eval_config=eval_config,
                    eval_freq=plan["eval_every_vector_steps"], eval_log_path=exp_dir if writer else None,
                    total_timesteps=plan["vector_steps"] * total, world_size=world, cvar=cvar, verbose=False,
                    report_timestep_scale=params["total_timesteps"] / (plan["vector_steps"] * total), eval_adaptive=eval_adaptive)
    if writer:
        agent.qnetwork_local.save(exp_dir)
    train_env.close()
    if eval_env is not None:
        eval_env.close()
    torch.cuda.synchronize()
    return exp_dir


def _worker_device(requested, i, n_gpu):
    """Device of pool worker i: `-D cuda:K` pins every worker to GPU K; `-D cuda` or no -D spreads them, worker i on GPU i modulo the
    visible GPUs; anything else (`-D cpu`) is passed through."""
    if requested is None or requested == "cuda":
        return f"cuda:{i % max(1, n_gpu)}"
    return requested


def _trial_worker(device, params, n_envs, kw, sharers=1):
    """One trial in a pool worker.  `sharers`: how many workers run on this worker's GPU at a time.  The one-launch gradient step needs every one of its
    workgroups resident at once and plans for the whole device; two such launches of two processes on one GPU would starve each other (bounded waits run
    out, the step is skipped): with company, the worker plans for its share of the CUs (mn_iqn_train_set_cu_limit -> the two- / three-launch forms, same
    results); whether the episode resets still find room beside the act kernel is checked by the loop's own first steps (iqn/agent.py: UnderActGuard)."""
    import torch
    dev = torch.device(device)
    if dev.type == "cuda" and dev.index is not None:      # (an un-indexed or non-CUDA device has no current-device to set)
        torch.cuda.set_device(dev)
    if sharers > 1 and dev.type == "cuda":
        from . import _capi
        cus = torch.cuda.get_device_properties(dev).multi_processor_count
        _capi.lib().mn_iqn_train_set_cu_limit(max(8, cus // sharers - 8))
    return run_trial(device, params, n_envs, verbose=False, **kw)


def main(argv=None):
    ap = argparse.ArgumentParser(description="Train IQN model (batched MI355X path)")
    ap.add_argument("-C", "--config-file", dest="config_file", type=open, required=True)
    ap.add_argument("-D", "--device", dest="device", type=str, default=None)
    ap.add_argument("-P", "--num-procs", dest="num_procs", type=int, default=1,
                    help="train_IQN_model.py:24-30: run the trials of the config grid (seeds) in this many worker processes at a time; "
                         "worker i uses GPU i modulo the visible GPUs (several seeds on one MI355X share it).  Not combinable with torch.distributed.run")
    ap.add_argument("--n-envs", type=int, default=4096, help="environments per GPU (default 4096: see the module docstring; 65536 = the bench's configuration)")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--replay", type=int, default=100_000)
    ap.add_argument("--shared-learner", action="store_true")
    ap.add_argument("--grad-steps", type=int, default=None,
                    help="gradient steps per vector step (default: 16 per 65 536 envs, i.e. learner ~ half of the GPU time)")
    ap.add_argument("--total-grad-steps", type=int, default=None,
                    help="learner budget (default: the reference's sample count, total_timesteps / 4 * 32 / batch)")
    ap.add_argument("--n-evals", type=int, default=None, help="evaluation points over the run (default: min(30, total_timesteps / eval_freq))")
    ap.add_argument("--cvar", type=float, default=1.0, help="CVaR level of the acting policy while training (configs[4]: 0.5)")
    ap.add_argument("--torch-train", action="store_true", help="gradient step through PyTorch instead of the fused HIP kernels")
    ap.add_argument("--precision", default="f64", choices=["f64", "mixed"],
                    help="env kernels: f64 (default; strict 1e-5 parity, free next to the IQN act kernel) or mixed")
    ap.add_argument("--exchange", default="collective", choices=["collective", "mailbox"],
                    help="with --shared-learner: how the ranks' gradients meet -- one RCCL all-reduce per gradient step, or the exchange inside the step's own launch")
    ap.add_argument("--shared-taus", action="store_true", help="acting: one set of 32 taus per act launch instead of per env (opt-in, ~11 %% shorter runs)")
    args = ap.parse_args(argv)
    params = json.load(args.config_file)
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # the reference's -D takes "cpu" / "cuda" (train_IQN_model.py:33-40); here: "cuda" = this rank's GPU, "cuda:K" = that GPU, anything else is refused
    # by the package itself (no CPU path: VecMarineNavEnv raises)
    device = f"cuda:{local}" if args.device in (None, "cuda") else args.device
    if torch.device(device).type == "cuda":
        torch.cuda.set_device(torch.device(device))
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device(device))
    stamp = datetime.now().strftime("%Y-%m-%d-%H-%M-%S")
    trials = trial_params(params)
    for p in trials:
        p["training_time"] = stamp
    kw = dict(batch=args.batch, replay=args.replay, grad_steps=args.grad_steps, torch_train=args.torch_train,
              total_grad_steps=args.total_grad_steps, n_evals=args.n_evals, cvar=args.cvar, precision=args.precision,
              exchange=args.exchange, shared_taus=args.shared_taus)
    if args.num_procs > 1:
        # train_IQN_model.py:173-179: a Pool of workers, one trial each.  `spawn`: every worker gets its own HIP context
        if world > 1:
            raise SystemExit("--num-procs runs independent trials; do not combine it with torch.distributed.run")
        import multiprocessing as mp
        n_gpu = max(1, torch.cuda.device_count())
        with mp.get_context("spawn").Pool(processes=args.num_procs) as pool:
            devs = [_worker_device(args.device, i, n_gpu) for i in range(len(trials))]
            # workers that can be on one GPU at a time: the pool runs `num_procs` trials at once, worker i on devs[i]
            sharers = max(1, max(sum(1 for d in devs[:args.num_procs] if d == x) for x in set(devs[:args.num_procs]))) if trials else 1
            jobs = [pool.apply_async(_trial_worker, (devs[i], p, args.n_envs, kw, sharers)) for i, p in enumerate(trials)]
            pool.close()
            for j in jobs:
                j.get()
            pool.join()
        return
    for p in trials:
        run_trial(device, p, args.n_envs, rank, world, args.shared_learner, verbose=True, **kw)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
