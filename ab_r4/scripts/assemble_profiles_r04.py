"""Copy the summaries scripts/measure_round4.sh left under gpurun_out/m5 into profiles/r04_* with headers."""
import json, os, re, shutil
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
M, P = R + '/gpurun_out/m5/', R + '/profiles/'
body = lambda f: open(M + f).read()
pm = body('pmc_summary.txt')
def mean(txt, k, c):
    m = re.search(r'^%s\s+%s\s+n=\s*\d+ mean=\s*([\d.]+)' % (k, c), txt, re.M)
    return float(m.group(1)) / 1e3 if m else float('nan')
hdr = ("# r04: bench.py --steps 100 --warmup 20 --cpu-steps 0 --no-learner-only --no-also --no-clock-probe under rocprofv3 --kernel-trace --stats (MI355X, 1 GPU; scripts/measure_round4.sh):\n"
       "# BASELINE configs[2] -- 65 536 envs + IQN training, 1 gradient step every 4 vector steps, float64 env kernels, per-env taus (the default), one batch / one stream.\n"
       "# Kernel durations are rocprofv3's (start to start: they include the launch boundary).  The gradient step is ONE launch since round 4 (iqn_train_fwdbwd with the reduction + clip + Adam blocks as its third workgroup role, XCD-grouped).\n")
out = hdr + body('prof_loop_summary.txt').rstrip() + '\n\n'
out += ("# PMC passes (separate runs, one counter each: rocprofv3 --kernel-trace --pmc <counter>; the same command with --steps 24 --warmup 8 --update-every 1 --grad-steps 4),\n"
        "# mean per launch; FETCH_SIZE / WRITE_SIZE in KB of 1000 B; HBM bytes = 2 x FETCH_SIZE (gfx950 correction, profiles/r01_pmc_calibration.txt) + WRITE_SIZE\n")
out += pm.rstrip() + '\n'
out += ('# derived: step kernel (float64, with replay append, float64 observation copies OFF = mn_enable_obs64 default): 2 x %.2f + %.2f = %.1f MB per 65 536-env launch (r03, copies\n'
        '# always written: 79.8 MB; algorithmic 734 B x 65 536 = 48.1 MB, step only 406 B = 26.6 MB); reset kernel: 2 x %.2f + %.2f = %.1f MB per launch (~2 000 episodes end per\n'
        '# vector step); act kernel (per-env taus): %.2f M MFMA-busy cycles (372 x 16 x 65 536 = 390.07 M), 2 x %.2f + %.2f = %.1f MB.\n') % (
    mean(pm, 'step', 'FETCH_SIZE'), mean(pm, 'step', 'WRITE_SIZE'), 2 * mean(pm, 'step', 'FETCH_SIZE') + mean(pm, 'step', 'WRITE_SIZE'),
    mean(pm, 'reset', 'FETCH_SIZE'), mean(pm, 'reset', 'WRITE_SIZE'), 2 * mean(pm, 'reset', 'FETCH_SIZE') + mean(pm, 'reset', 'WRITE_SIZE'),
    mean(pm, 'act', 'SQ_VALU_MFMA_BUSY_CYCLES') / 1e3, mean(pm, 'act', 'FETCH_SIZE'), mean(pm, 'act', 'WRITE_SIZE'), 2 * mean(pm, 'act', 'FETCH_SIZE') + mean(pm, 'act', 'WRITE_SIZE'))
open(P + 'r04_full_loop_kernel_stats.txt', 'w').write(out)
ps = body('pmc_shared_summary.txt')
out = ("# r04: the same loop with launch-shared taus (bench.py --shared-taus ...; opt-in): the act kernel is iqn_qvals_tiled_kernel at 65 536 envs (csrc/iqn_act_tiled.h) behind two\n"
       "# preparation launches (iqn_shared_prep_kernel: the call's draws + the layer-1 constant; iqn_tiled_prep_kernel: T = W2 diag(h1) as hi / lo f16 pairs)\n")
out += body('prof_shared_summary.txt').rstrip() + '\n\n# PMC passes, act kernel only (mean per launch)\n' + ps.rstrip() + '\n'
out += ('# derived: %.2f M MFMA-busy cycles (216 x 16 x 65 536 = 226.5 M + the encoders\' 104 exact-f32 MFMAs per 16 envs); %.1f M vector-ALU instructions (matrix instructions included) per launch = %.0f per env\n'
        '# (per-env-tau kernel: 93.0 M = 1 419); WRITE_SIZE %.1f MB: the kernel spills 24 registers per lane (96 B x 131 072\n'
        '# threads) in its prologue (encoders + feature split of two 16-env column tiles); FETCH 2 x %.1f MB.\n') % (
    mean(ps, 'act', 'SQ_VALU_MFMA_BUSY_CYCLES') / 1e3, mean(ps, 'act', 'SQ_INSTS_VALU') / 1e3, mean(ps, 'act', 'SQ_INSTS_VALU') * 1e3 / 65536,
    mean(ps, 'act', 'WRITE_SIZE'), mean(ps, 'act', 'FETCH_SIZE'))
open(P + 'r04_shared_taus_loop_kernel_stats.txt', 'w').write(out)
out = ("# r04: the cadence that trains -- bench.py --update-every 1 --grad-steps 16 --eps 0.05 (16 gradient steps per vector step) under rocprofv3 --kernel-trace --stats.\n"
       "# Kernel durations include the launch boundary.  A gradient step = ONE launch of iqn_train_fwdbwd (reduction + clip + Adam inside, XCD-grouped; r03: iqn_train_fwdbwd, iqn_grad_reduce, iqn_adam).\n")
out += body('prof_g16_summary.txt').rstrip() + '\n'
open(P + 'r04_train_cadence_kernel_stats.txt', 'w').write(out)
shutil.copy(M + 'bench_default.json', P + 'r04_bench_default.json')
shutil.copy(M + 'bench_shared_taus.json', P + 'r04_bench_shared_taus.json')
open(P + 'r04_experiment_sweep.txt', 'w').write("# r04: scripts/experiment_sweep.py (MI355X): the reference's full comparison, run_experiments.py:213-282 -- IQN x 5 through the fused act kernel (act_eval form not needed),\n"
    "# DQN through csrc/dqn_act.hip, APF / BA as one mn_rollout_policy launch each (r01, all eager / launch-per-step: 2.5 s)\n" + '\n'.join(l for l in body('experiment_sweep.txt').split('\n') if 'amdgpu.ids' not in l))
open(P + 'r04_learner_bench.txt', 'w').write("# r04: scripts/learner_bench.py 3000 (learner alone, batch 256 drawn in the launch, replay 100 000), by launches per gradient step; see r04_train_step_launches.txt\n"
    + '\n'.join(l for l in body('learner_bench.txt').split('\n') if 'amdgpu.ids' not in l))
j = json.load(open(M + 'bench_default.json'))
print(j['value'] / 1e6, j['ms_per_step'], j['roofline']['launch_ms'], j['roofline_env_step']['launch_ms'])
