"""Condense a rocprofv3 --kernel-trace --stats CSV (…_kernel_stats.csv) into a short table."""
import csv, re, sys

def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"^(Cijk_\w+?_MT\d+x\d+x\d+)", name)
    if m:
        return m.group(1) + " (hipBLASLt f32 GEMM)"
    name = re.sub(r"void ", "", name)
    name = re.sub(r"at::native::", "", name)
    if len(name) > 110:
        name = name[:107] + "..."
    return name

rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
print(f"# total kernel time {tot/1e6:.3f} ms over {sum(int(r['Calls']) for r in rows)} launches, {len(rows)} distinct kernels")
print(f"{'calls':>6} {'total_ms':>9} {'avg_us':>9} {'min_us':>8} {'max_us':>8} {'pct':>6}  kernel")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:n]:
    print(f"{int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:9.3f} {float(r['AverageNs'])/1e3:9.2f} {float(r['MinNs'])/1e3:8.2f} "
          f"{float(r['MaxNs'])/1e3:8.2f} {float(r['Percentage']):6.2f}  {short(r['Name'])}")
