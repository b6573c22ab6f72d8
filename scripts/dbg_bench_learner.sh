#!/bin/bash
# which leg of bench.py slows the one-launch learner down?
cd $GRAFT_REPO_ROOT
python - <<'PY'
import json, subprocess, sys
for extra in (["--no-also"], []):
    for one in ("1", "0"):
        import os
        env = dict(os.environ, MN_ONE_LAUNCH=one)
        r = subprocess.run([sys.executable, "bench.py", "--no-clock-probe", "--cpu-steps", "0", "--steps", "100"] + extra, capture_output=True, text=True, env=env)
        try:
            d = json.loads(r.stdout.strip().splitlines()[-1])
            a = d.get("also") or {}
            tc = a.get("train_cadence", {})
            print(extra, "one_launch", one, "learner_only", d["learner_only_grad_steps_per_sec_per_gpu"].get("fused_hip"), "train_cadence", tc.get("ms_per_step"), tc.get("xcd_misplaced_workgroups"), flush=True)
        except Exception as e:
            print("failed", extra, one, r.stderr[-500:])
PY
