"""Aggregate rocprofv3 --pmc counter_collection.csv per kernel (mean over launches, skipping warmup)."""
import csv, collections, sys
pat = sys.argv[2] if len(sys.argv) > 2 else None
d = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name']
    short = 'rollout' if 'mn_rollout_kernel' in k else 'step' if 'mn_step_kernel' in k else ('reset' if 'mn_reset_kernel' in k else ('reset_ua' if 'mn_reset_under_act_kernel' in k else ('act' if 'iqn_qvals' in k else ('train' if 'iqn_train_fwdbwd' in k else ('reduce' if 'iqn_grad_reduce' in k else ('adam' if 'iqn_adam' in k else None))))))
    if short and (pat is None or short == pat):
        d[short][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in d.items():
    for c, vals in sorted(v.items()):
        vals = vals[5:] if len(vals) > 10 else vals
        print(f"{k:8s} {c:32s} n={len(vals):4d} mean={sum(vals)/len(vals):16.1f}")
