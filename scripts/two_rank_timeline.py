"""What two ranks that share ONE GPU do to each other, from rocprofv3 kernel traces of both processes (one *_kernel_trace.csv per rank; timestamps of one clock).
usage: python scripts/two_rank_timeline.py <dir with the traces> [label]
Per rank and kernel class (act / step / reset / gradient-step kernels): launches, mean duration, and the share of its duration during which a kernel of the OTHER rank was running (by class);
then, for the gradient-step kernel that carries the exchange, its duration split by what the peer ran meanwhile."""
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]; label = sys.argv[2] if len(sys.argv) > 2 else d
def cls(n):
    if "iqn_qvals" in n: return "act"
    if "mn_step_kernel" in n: return "step"
    if "mn_reset" in n: return "reset"
    if "iqn_grad_reduce_adam" in n or "iqn_grad_gather" in n: return "train_xchg"
    if "iqn_train_fwdbwd" in n: return "train_fwdbwd"
    if "iqn_grad_reduce" in n or "iqn_adam" in n or "iqn_grad_sumsq" in n: return "train_rest"
    if "iqn_split_prep" in n or "iqn_split_consts" in n: return "draw"
    return None
ranks = []
for f in sorted(glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)):
    rows = []
    for r in csv.DictReader(open(f)):
        c = cls(r["Kernel_Name"])
        if c:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), c))
    if len(rows) > 200:
        ranks.append(sorted(rows))
print(f"# {label}: {len(ranks)} processes with kernels")
if len(ranks) < 2:
    sys.exit(0)
ranks = ranks[:2]
# the steady part: drop each rank's first 40 % (start-up, warm-up, preflight)
t0 = max(r[int(0.4 * len(r))][0] for r in ranks); t1 = min(r[-1][1] for r in ranks)
for me in (0, 1):
    other = [x for x in ranks[1 - me] if x[1] > t0 and x[0] < t1]
    stat = defaultdict(lambda: [0, 0, defaultdict(int)])
    j0 = 0
    for s, e, c in ranks[me]:
        if s < t0 or e > t1:
            continue
        st = stat[c]; st[0] += 1; st[1] += e - s
        while j0 < len(other) and other[j0][1] <= s:
            j0 += 1
        j = j0
        while j < len(other) and other[j][0] < e:
            st[2][other[j][2]] += min(e, other[j][1]) - max(s, other[j][0])
            j += 1
    print(f"rank {me}: steady window {1e-6 * (t1 - t0):.1f} ms")
    for c, (n, tot, ov) in sorted(stat.items(), key=lambda kv: -kv[1][1]):
        o = ", ".join(f"{k} {100.0 * v / tot:.0f} %" for k, v in sorted(ov.items(), key=lambda kv: -kv[1]) if v > 0.01 * tot)
        print(f"  {c:13s} {n:5d} launches  mean {1e-3 * tot / n:9.1f} us  total {1e-6 * tot:8.2f} ms   peer running meanwhile: {o or '-'}")
