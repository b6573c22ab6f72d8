# Pure execution time (end - start timestamps of rocprofv3 --kernel-trace) of the loop kernels of bench.py, mean / median over 200 vector steps.
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
rm -rf /tmp/pr; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr -- python $R/bench.py --steps 200 --warmup 20 --cpu-steps 0 --no-learner-only --no-also --no-clock-probe > /tmp/pr.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/pr/**/*kernel_trace.csv', recursive=True)[0]
d = {}
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    for key in ('mn_reset_kernel', 'mn_step_kernel', 'iqn_qvals_split_kernel', 'iqn_train_fwdbwd', 'iqn_split_prep'):
        if key in k:
            d.setdefault(key, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in d.items():
    v = sorted(v)[2:-2] if len(v) > 10 else v
    print(f"{k:28s} n={len(v):4d} mean {sum(v)/len(v):8.2f} us  median {v[len(v)//2]:8.2f}")
PY
