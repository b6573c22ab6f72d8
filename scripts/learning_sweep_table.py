"""One table from the per-configuration outputs of scripts/learning_curve.py (scripts/learning_sweep.sh, learning_sweep2.sh): usage: python scripts/learning_sweep_table.py <dir> [<dir> ...]
Per configuration (runs with the same envs / ring / n-step are pooled over seed sets): last evaluation of the run -- mean +- sd, median, worst -- mean of the last 3 evaluations (median over
runs), best_* checkpoint (a max over 16 noisy evaluations: selection, quoted only next to the reference's own best), seconds per run, share of runs that reach 26/30 at some evaluation."""
import glob, os, re, sys
import numpy as np
rows = {}
for d in sys.argv[1:]:
    for f in sorted(glob.glob(os.path.join(d, "lc_*.txt"))):
        t = open(f).read()
        h = re.search(r"(\d+) envs, replay ring (\d+) rows", t)
        if not h:
            continue
        nstep = re.search(r"--n-step (\d+)", t)
        key = (int(h.group(1)), int(h.group(2)), int(nstep.group(1)) if nstep else 1)
        r = rows.setdefault(key, dict(last=[], best=[], wall=[], seeds=[], last3=[], reach=[]))
        for m in re.finditer(r"# seed (\d+):\s+([\d.]+) s incl\. evaluations; last evaluation (\d+)/30\s+(-?[\d.]+); best (\d+)/30\s+(-?[\d.]+)", t):
            r["seeds"].append(int(m.group(1))); r["wall"].append(float(m.group(2))); r["last"].append((int(m.group(3)), float(m.group(4)))); r["best"].append((int(m.group(5)), float(m.group(6))))
        m = re.search(r"reached by (\d+)/(\d+) runs", t)
        if m:
            r["reach"].append((int(m.group(1)), int(m.group(2))))
print("# envs per GPU | replay ring rows | n-step | runs | LAST evaluation: successes /30 mean +- sd (median, worst) | mean return mean +- sd (median, worst) | best_* checkpoint successes, return (selection) | s per run | runs reaching 26/30 at some evaluation")
for key in sorted(rows, key=lambda k: (-k[0], k[1], k[2])):
    r = rows[key]
    if not r["last"]:
        continue
    L = np.array(r["last"], dtype=float); B = np.array(r["best"], dtype=float)
    reach = f"{sum(a for a, _ in r['reach'])}/{sum(b for _, b in r['reach'])}" if r["reach"] else "-"
    print(f"{key[0]:6d} | {key[1]:9d} | {key[2]} | {len(L):2d} | {L[:, 0].mean():5.2f} +- {L[:, 0].std(ddof=1):4.2f} ({np.median(L[:, 0]):4.1f}, {L[:, 0].min():2.0f}) | {L[:, 1].mean():6.2f} +- {L[:, 1].std(ddof=1):5.2f} ({np.median(L[:, 1]):6.2f}, {L[:, 1].min():7.2f}) | "
          f"{B[:, 0].mean():5.2f}, {B[:, 1].mean():6.2f} | {np.mean(r['wall']):4.1f} | {reach}")
