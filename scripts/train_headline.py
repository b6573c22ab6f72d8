"""Does BASELINE configs[2] -- 65 536 envs, replay 100 000, batch 256 -- learn, and at which cadence?

Trains a fresh IQN on the headline configuration with a given cadence (G gradient steps every U vector steps) for a
wall-clock budget, evaluating the greedy policy on the reference's 30 evaluation worlds along the way.  Everything is
the product path: HIP env (mn_step_append / mn_reset_done), fused act kernel, fused HIP gradient step.

    python scripts/train_headline.py --update-every 4 --grad-steps 1  --seconds 120     # SURVEY 8d C3's cadence (bench default)
    python scripts/train_headline.py --update-every 1 --grad-steps 16 --seconds 60      # recommended (train_iqn.plan_cadence)

Reference (pretrained_models/IQN/seed_3, 3 M env steps / 750 k grad steps of batch 32): 26/30 successes, mean return 69.25.
"""
import argparse, contextlib, io, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from distributional_rl_navigation_amd.iqn.agent import IQNAgent
from distributional_rl_navigation_amd.marinenav_env.vec_env import VecMarineNavEnv
from distributional_rl_navigation_amd.train_iqn import TRAINING_SCHEDULE, plan_cadence

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=65536)
ap.add_argument("--replay", type=int, default=100_000)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--update-every", type=int, default=1)
ap.add_argument("--grad-steps", type=int, default=16)
ap.add_argument("--seconds", type=float, default=60.0, help="wall-clock budget of the training loop")
ap.add_argument("--vector-steps", type=int, default=0, help="length of the run the eps ramp / curriculum are spread over (0 = plan_cadence)")
ap.add_argument("--evals", type=int, default=10)
ap.add_argument("--seed", type=int, default=100)
ap.add_argument("--cvar", type=float, default=1.0)
ap.add_argument("--shared-taus", action="store_true", help="acting: one set of 32 taus per launch instead of per env (IQNAgent.shared_taus); the learner is unchanged")
args = ap.parse_args()

with open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "eval_config_seed3.json")) as f:
    cfg = json.load(f)
n = args.envs
plan = plan_cadence(3_000_000, 10_000, n, args.batch, grad_steps_per_vector_step=max(1, args.grad_steps // args.update_every) if args.grad_steps >= args.update_every else None)
V = args.vector_steps or (plan["vector_steps"] if args.grad_steps >= args.update_every else int(np.ceil(plan["total_grad_steps"] * args.update_every / args.grad_steps)))
env = VecMarineNavEnv(n, seed=0, schedule=TRAINING_SCHEDULE, timestep_scale=3_000_000 / V, device="cuda:0", precision="f64")   # whole curriculum over the run; strict env kernels (the loop default)
eval_env = VecMarineNavEnv(30, device="cuda:0", precision="f64")
agent = IQNAgent(26, 9, BATCH_SIZE=args.batch, BUFFER_SIZE=args.replay, device="cuda:0", seed=args.seed, learning_starts=0,
                 UPDATE_EVERY=args.update_every)
agent.grad_steps_per_update = args.grad_steps
agent.shared_taus = args.shared_taus
agent.target_sync_grad_steps = plan["target_sync_grad_steps"]
print(f"# seed {args.seed}, act taus {'shared per launch' if args.shared_taus else 'per env'}; {n} envs, replay {args.replay}, batch {args.batch}: {args.grad_steps} grad step(s) every {args.update_every} vector step(s); "
      f"eps ramp / curriculum over {V} vector steps ({V * n:.3g} env steps); target copy every {agent.target_sync_grad_steps} grad steps; "
      f"replay ratio {args.grad_steps * args.batch / (args.update_every * n):.4f} sampled / generated transition", flush=True)


def evaluate():
    with contextlib.redirect_stdout(io.StringIO()):
        r = agent.evaluation_vec(eval_env, cfg, greedy=True)
    return sum(r["successes"]), float(np.mean(r["rewards"]))


agent.reset_under_act = True      # episode resets under the next step's act kernel while few episodes end per vector step (the library decides)
obs = env.reset()
t_train, it, next_eval = 0.0, 0, 0.0
rows = []
while True:
    if t_train >= next_eval or t_train >= args.seconds or it >= V:
        torch.cuda.synchronize()
        s, ret = evaluate()
        rows.append((it, agent.current_timestep, agent.grad_steps, t_train, s, ret))
        print(f"[vector step {it:6d} | env steps {agent.current_timestep:12d} | grad steps {agent.grad_steps:7d} | train time {t_train:6.1f} s"
              f" | eps {agent.linear_eps(V * n):.3f}] greedy eval: success {s:2d}/30  mean return {ret:7.2f}", flush=True)
        next_eval += args.seconds / args.evals
        if t_train >= args.seconds or it >= V:
            break
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        eps = agent.linear_eps(V * n)
        obs, *_ = agent.vec_step(env, obs, eps, args.cvar, per_iter=n)
        it += 1
        if it >= V:
            break
    torch.cuda.synchronize(); t_train += time.perf_counter() - t0
best = max(rows, key=lambda r: (r[4], r[5]))
print(f"# {it} vector steps in {t_train:.1f} s = {it * n / t_train / 1e6:.1f} M env steps/s, {agent.grad_steps / t_train:.0f} grad steps/s; "
      f"best evaluation {best[4]}/30, mean return {best[5]:.2f} after {best[3]:.1f} s / {best[2]} grad steps / {best[1]:.3g} env steps")
