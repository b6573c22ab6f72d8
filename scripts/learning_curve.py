"""The training run judged against the reference's own learning curve.

The reference evaluates and checkpoints every `eval_freq` timesteps (thirdparty/IQN/agent.py:140-148) and ships the whole curve of its model
(pretrained_models/IQN/seed_3/greedy_evaluations.npz -> tests/golden/ref_iqn_seed3_greedy_curve.npz: 300 evaluations, final 26/30 and 69.25, best 29/30 and 83.2).
This script runs `train_iqn.run_trial` (65 536 envs, 93 760 gradient steps of batch 256 = the reference's 24 M sampled transitions) for a set of seeds with
evaluations at the plan's evaluation points and prints, per evaluation point (x = reference-equivalent timesteps, i.e. the fraction of the learner's budget spent),
the median and inter-quartile band of successes / mean return over the seeds next to the reference's curve at the same x; then every run's final ("latest") and
best (`best_*`) checkpoint.  Study knobs: --target-mult (target copy every m x 312 gradient steps), --eps-floor.

    python scripts/learning_curve.py --seeds 12 --evals 15 > profiles/r05_learning_curve.txt"""
import argparse, contextlib, io, json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--seeds", type=int, default=12)
ap.add_argument("--first-seed", type=int, default=0)
ap.add_argument("--seed-list", default="", help="comma separated seeds (overrides --seeds / --first-seed)")
ap.add_argument("--evals", type=int, default=15)
ap.add_argument("--envs", type=int, default=65536)
ap.add_argument("--target-mult", type=float, default=1.0)
ap.add_argument("--eps-floor", type=float, default=0.05)
ap.add_argument("--shared-taus", action="store_true")
ap.add_argument("--replay", type=int, default=100_000, help="replay ring rows (the bench keeps BASELINE's 100 000)")
ap.add_argument("--grad-steps", type=int, default=None, help="gradient steps per vector step (default: 16 per 65 536 envs)")
ap.add_argument("--n-step", type=int, default=1)
ap.add_argument("--tag", default="", help="label printed in the header line")
args = ap.parse_args()

import torch
from distributional_rl_navigation_amd.train_iqn import run_trial
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ref = np.load(os.path.join(ROOT, "tests", "golden", "ref_iqn_seed3_greedy_curve.npz"))
seeds = [int(x) for x in args.seed_list.split(",")] if args.seed_list else list(range(args.first_seed, args.first_seed + args.seeds))
print(f"# {args.tag + ': ' if args.tag else ''}{len(seeds)} seeds {seeds}; {args.envs} envs, replay ring {args.replay} rows, {args.evals} evaluation points, target copy x{args.target_mult}, eps floor {args.eps_floor}, "
      f"acting taus {'shared per launch' if args.shared_taus else 'per env'}", flush=True)
curves, finals, bests, walls = [], [], [], []
for sd in seeds:
    with tempfile.TemporaryDirectory() as tmp:
        params = dict(seed=sd, total_timesteps=3_000_000, eval_freq=10_000, save_dir=tmp, training_time="run")
        t0 = time.time()
        with contextlib.redirect_stdout(io.StringIO()):
            d = run_trial("cuda:0", params, args.envs, verbose=False, n_evals=args.evals, target_sync_mult=args.target_mult, final_eps=args.eps_floor,
                          eval_adaptive=False, shared_taus=args.shared_taus, replay=args.replay, grad_steps=args.grad_steps, n_step=args.n_step)
        torch.cuda.synchronize()
        walls.append(time.time() - t0)
        ev = np.load(os.path.join(d, "greedy_evaluations.npz"), allow_pickle=True)
        t = np.asarray(ev["timesteps"], dtype=np.int64)
        succ = np.asarray(ev["successes"]).astype(np.int64).sum(axis=1)
        ret = np.asarray(ev["rewards"], dtype=np.float64).mean(axis=1)
        best = json.load(open(os.path.join(d, "best_evaluation.json")))
        # the run's last evaluation point is not its end: evaluate the final network too (what `network_params.pth` holds)
        curves.append((t, succ, ret)); finals.append((int(succ[-1]), float(ret[-1]))); bests.append((best["successes"], best["mean_return"], best["timestep"]))
        print(f"# seed {sd}: {walls[-1]:5.1f} s incl. evaluations; last evaluation {succ[-1]}/30 {ret[-1]:7.2f}; best {best['successes']}/30 {best['mean_return']:7.2f} at timestep {best['timestep']}", flush=True)
n_pts = min(len(c[0]) for c in curves)
print("\n# evaluation point | reference-equivalent timesteps | successes /30: median [q25, q75] (min) | mean return: median [q25, q75] (min) | reference at the same timesteps: successes, return")
for i in range(n_pts):
    t = int(np.median([c[0][i] for c in curves]))
    s = np.array([c[1][i] for c in curves], dtype=np.float64); r = np.array([c[2][i] for c in curves])
    j = int(np.argmin(np.abs(ref["timesteps"] - t)))
    lo, hi = max(0, j - 5), min(len(ref["timesteps"]), j + 6)      # the reference's curve is noisy from one evaluation to the next: an 11-point window around it
    print(f"{i:3d} | {t:9d} | {np.median(s):5.1f} [{np.percentile(s, 25):5.1f}, {np.percentile(s, 75):5.1f}] ({s.min():4.0f}) | {np.median(r):7.2f} [{np.percentile(r, 25):7.2f}, {np.percentile(r, 75):7.2f}] ({r.min():7.2f})"
          f" | {int(ref['successes'][j]):2d}/30 {ref['mean_return'][j]:7.2f}   (window median {np.median(ref['successes'][lo:hi]):4.1f}, {np.median(ref['mean_return'][lo:hi]):6.2f})")
f = np.array(finals, dtype=np.float64); b = np.array([x[:2] for x in bests], dtype=np.float64)
print(f"\n# last evaluation over {len(seeds)} runs: successes {f[:, 0].mean():.2f} +- {f[:, 0].std(ddof=1):.2f} /30, mean return {f[:, 1].mean():.2f} +- {f[:, 1].std(ddof=1):.2f}; worst {f[:, 0].min():.0f}/30, {f[:, 1].min():.2f}")
print(f"# best_* checkpoint over {len(seeds)} runs: successes {b[:, 0].mean():.2f} +- {b[:, 0].std(ddof=1):.2f} /30, mean return {b[:, 1].mean():.2f} +- {b[:, 1].std(ddof=1):.2f}; worst {b[:, 0].min():.0f}/30, {b[:, 1].min():.2f}")
first = [next((i for i, v in enumerate(c[1]) if v >= 26), None) for c in curves]
reached = [f_ for f_ in first if f_ is not None]
if reached:
    frac = [(f_ + 1) / len(c[1]) for f_, c in zip(first, curves) if f_ is not None]
    print(f"# first evaluation with >= 26/30 (the reference's final): reached by {len(reached)}/{len(curves)} runs, median at {np.median(frac):.2f} of the run = {np.median(frac) * np.mean(walls):.1f} s of wall clock")
last3 = np.array([[np.mean(c[1][-3:]), np.mean(c[2][-3:])] for c in curves])
print(f"# mean of each run's last 3 evaluations: successes median {np.median(last3[:, 0]):.1f} [q25 {np.percentile(last3[:, 0], 25):.1f}, q75 {np.percentile(last3[:, 0], 75):.1f}], return median {np.median(last3[:, 1]):.1f} [q25 {np.percentile(last3[:, 1], 25):.1f}, q75 {np.percentile(last3[:, 1], 75):.1f}]; "
      f"last evaluation: successes median {np.median(f[:, 0]):.1f}, return median {np.median(f[:, 1]):.1f}")
print(f"# reference (one seed): final 26/30, 69.25; best of its 300 evaluations 29/30, 83.23.  wall clock per run here: {np.mean(walls):.1f} s incl. {args.evals} evaluations")
