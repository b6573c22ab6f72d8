"""Copy the summaries scripts/measure_round.sh left under gpurun_out/<measure dir> into profiles/<round tag>_* with headers.
usage: python scripts/assemble_profiles.py [round tag, default r06] [measure dir, default m6]
Fails if a PMC constant in bench.py (PMC_TRAFFIC_MB) disagrees by more than 5 % with the profile it cites."""
import json, os, re, shutil, sys
RT = sys.argv[1] if len(sys.argv) > 1 else 'r06'
MD = sys.argv[2] if len(sys.argv) > 2 else 'm6'
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
M, P = R + '/gpurun_out/' + MD + '/', R + '/profiles/' + RT + '_'
clean = lambda t: '\n'.join(l for l in t.split('\n') if 'amdgpu.ids' not in l and not l.startswith('[W') and 'RCCL version' not in l and 'HIP version' not in l
                            and 'ROCm version' not in l and 'Hostname ' not in l and 'Librccl path' not in l)
body = lambda f: clean(open(M + f).read())
pm = body('pmc_summary.txt')
def mean(txt, k, c):
    m = re.search(r'^%s\s+%s\s+n=\s*\d+ mean=\s*([\d.]+)' % (k, c), txt, re.M)
    return float(m.group(1)) / 1e3 if m else float('nan')
hbm = lambda txt, k: 2 * mean(txt, k, 'FETCH_SIZE') + mean(txt, k, 'WRITE_SIZE')

out = ("# %s: bench.py --steps 100 --warmup 20 --cpu-steps 0 --no-learner-only --no-also --no-clock-probe under rocprofv3 --kernel-trace --stats (MI355X, 1 GPU; scripts/measure_round.sh):\n"
       "# BASELINE configs[2] -- 65 536 envs + IQN training, 1 gradient step every 4 vector steps, float64 env kernels, per-env taus (the default), one batch / one stream.\n"
       "# Kernel durations are rocprofv3's (start to start: they include the launch boundary).  The gradient step is ONE launch: iqn_train_fwdbwd<XCHG = false, FUSED = true>\n"
       "# (target / local / reduction + clip + Adam workgroup roles, XCD-grouped).  mn_reset_kernel's average contains the initial all-env resets (max column); in the loop: bench.py's live row.\n") % RT
out += ("# Episode resets: mn_reset_under_act_kernel on the env handle's own stream, beside the act kernel's workgroups (iqn_qvals_split_kernel<.., LATE = true> takes the finished\n"
        "# envs' rows last; four reset wavefronts per CU, MT19937 rows read in place); mn_reset_kernel / the LATE = false act kernel are the initial reset and the very first vector\n"
        "# steps (nothing seen yet by the host).\n")
out += body('prof_loop_summary.txt').rstrip() + '\n\n'
if os.path.exists(M + 'prof_loop_reset_in_front_summary.txt'):
    out += "# the same command with --reset-in-front (mn_reset_done on the caller's stream, every vector step waits for it):\n" + body('prof_loop_reset_in_front_summary.txt').rstrip() + '\n\n'
out += ("# PMC passes (separate runs, one counter each: rocprofv3 --kernel-trace --pmc <counter>; the same command with --steps 24 --warmup 8 --update-every 1 --grad-steps 4),\n"
        "# mean per launch; FETCH_SIZE / WRITE_SIZE in KB of 1000 B; HBM bytes = 2 x FETCH_SIZE (gfx950 correction, profiles/r01_pmc_calibration.txt) + WRITE_SIZE\n")
out += pm.rstrip() + '\n'
out += ('# derived, per launch: step kernel (float64, with replay append): 2 x %.2f + %.2f = %.1f MB.  Accounting (VERDICT r5 item 7): SURVEY 8(d)\'s 406 B per env-step prices a FLOAT32 SoA;\n'
        '# this kernel keeps pose and world tables in float64 (north-star tolerance, DESIGN 2) and appends the transition: read pose 48 + goal 16 + counters / action 20 + tables (8 + 10) x 24 = 432 + obs_t row 104\n'
        '# = 620 B, written pose 48 + counters 12 + obs 104 + reward / done / info 6 + ring row 224 = 394 B: 1 014 B per env-step = 66.5 MB per launch.  Measured / that = %.2f: the kernel moves the bytes\n'
        '# of its layout, no partial-line or write-allocate excess (stores are byte-masked, nothing is fetched for a write).  Against the float32 figures: 734 B x 65 536 = 48.1 MB (x %.2f), step only 406 B = 26.6 MB;\n'
        '# reset kernels (~220 episode ends per vector step at this cadence, 2 048 B algorithmic each = 0.45 MB): mn_reset_under_act_kernel 2 x %.2f + %.2f = %.1f MB per launch -- ~25 KB per reset:\n'
        '# the under-act form reads its candidates\' words from the env\'s MT19937 row in place (8-byte agent-scope loads, four per 32-byte piece of the row), regenerates the row in place with 4-byte\n'
        '# write-through stores, writes the observation row through, and spills ~20 registers per lane to scratch (it is compiled for the 96 registers a SIMD has left beside two act wavefronts);\n'
        '# all of it off the critical path, under a kernel that moves 25 MB in 300 us.  mn_reset_kernel (in front: the first launches of a run) 2 x %.2f + %.2f = %.1f MB;\n'
        '# act kernel (per-env taus): %.2f M MFMA-busy cycles (372 x 16 x 65 536 = 390.07 M), %.1f M vector-ALU instructions = %.0f per env, %.1f MB;\n'
        '# gradient step (one launch): %.1f MB (18.3 MB of partial-gradient rows are written once; read back through the XCDs\' L2s).\n') % (
    mean(pm, 'step', 'FETCH_SIZE'), mean(pm, 'step', 'WRITE_SIZE'), hbm(pm, 'step'), hbm(pm, 'step') / 66.5, hbm(pm, 'step') / 48.1,
    mean(pm, 'reset_ua', 'FETCH_SIZE'), mean(pm, 'reset_ua', 'WRITE_SIZE'), hbm(pm, 'reset_ua'), mean(pm, 'reset', 'FETCH_SIZE'), mean(pm, 'reset', 'WRITE_SIZE'), hbm(pm, 'reset'),
    mean(pm, 'act', 'SQ_VALU_MFMA_BUSY_CYCLES') / 1e3, mean(pm, 'act', 'SQ_INSTS_VALU') / 1e3, mean(pm, 'act', 'SQ_INSTS_VALU') * 1e3 / 65536, hbm(pm, 'act'), hbm(pm, 'train'))
open(P + 'full_loop_kernel_stats.txt', 'w').write(out)

ps = body('pmc_shared_summary.txt')
out = ("# %s: the same loop with launch-shared taus (bench.py --shared-taus ...; opt-in): the act kernel is iqn_qvals_tiled_kernel at 65 536 envs (csrc/iqn_act_tiled.h) behind two\n"
       "# preparation launches (iqn_shared_prep_kernel: the call's draws + the layer-1 constant; iqn_tiled_prep_kernel: T = W2 diag(h1) as hi / lo f16 pairs)\n") % RT
out += body('prof_shared_summary.txt').rstrip() + '\n\n# PMC passes, act kernel only (mean per launch)\n' + ps.rstrip() + '\n'
out += ('# derived: %.2f M MFMA-busy cycles (216 x 16 x 65 536 = 226.5 M + the encoders\' exact-f32 MFMAs); %.1f M vector-ALU instructions (matrix instructions included) per launch = %.0f per env\n'
        '# (per-env-tau kernel: 1 419); WRITE_SIZE %.2f MB (r04: 12.9 MB -- 24 spilled registers per lane; round 5: T streamed by LDS-DMA, tau loop not unrolled, 254 VGPRs, no scratch);\n'
        '# FETCH 2 x %.1f MB.\n') % (
    mean(ps, 'act', 'SQ_VALU_MFMA_BUSY_CYCLES') / 1e3, mean(ps, 'act', 'SQ_INSTS_VALU') / 1e3, mean(ps, 'act', 'SQ_INSTS_VALU') * 1e3 / 65536,
    mean(ps, 'act', 'WRITE_SIZE'), mean(ps, 'act', 'FETCH_SIZE'))
open(P + 'shared_taus_loop_kernel_stats.txt', 'w').write(out)

out = ("# %s: the cadence that trains -- bench.py --update-every 1 --grad-steps 16 --eps 0.05 (16 gradient steps per vector step) under rocprofv3 --kernel-trace --stats.\n"
       "# Kernel durations include the launch boundary.  A gradient step = ONE launch of iqn_train_fwdbwd<false, true> (reduction + clip + Adam inside, XCD-grouped).\n") % RT
out += body('prof_g16_summary.txt').rstrip() + '\n'
open(P + 'train_cadence_kernel_stats.txt', 'w').write(out)

shutil.copy(M + 'bench_default.json', P + 'bench_default.json')
shutil.copy(M + 'bench_shared_taus.json', P + 'bench_shared_taus.json')
open(P + 'experiment_sweep.txt', 'w').write(("# %s: scripts/experiment_sweep.py (MI355X): the reference's full comparison, run_experiments.py:213-282 -- IQN x 5 through the fused act kernel,\n"
    "# DQN through csrc/dqn_act.hip, APF / BA as one mn_rollout_policy launch each\n" % RT) + body('experiment_sweep.txt'))

out = ("# %s: the gradient step by launch form (learner alone, batch 256 drawn in the launch, replay 100 000; MI355X, one box, one session).\n"
       "# (1) scripts/learner_bench.py 3000: wall clock over 3 000 back-to-back steps.  'mode 1' = every local workgroup computes its own target forward (the fallback of the in-launch TD hand-off).\n"
       "#     1 launch = iqn_train_fwdbwd<., FUSED> with the reduction + clip + Adam workgroups as its tail role; 2 = iqn_train_fwdbwd + iqn_grad_reduce_adam; 3 = + iqn_grad_reduce, iqn_adam (the RCCL form);\n"
       ) % RT
out += body('learner_bench.txt').rstrip() + '\n\n'
out += "# (2) rocprofv3 --kernel-trace --stats of scripts/learner_prof.py <form> (600 steps each; durations start to start)\n"
for form, name in ((1, 'one launch per step'), (2, 'two launches per step'), (3, 'three launches per step')):
    rows = [l for l in body('prof_learner_%d_summary.txt' % form).split('\n') if 'iqn_' in l]
    out += '# -- %s\n' % name + '\n'.join(rows) + '\n'
out += ("#\n# (The persistent multi-step launch of round 5 -- 33.4 us per step against 32.4 for single fused launches, profiles/r05_train_step_launches.txt -- was removed in round 6.)\n")
open(P + 'train_step_launches.txt', 'w').write(out)

out = ("# %s: scripts/scale.sh on ONE MI355X (no multi-GPU node was available to the build).  c3 = configs[3] (independent learners), c4 = configs[4] (shared learner, CVaR 0.5, RCCL all-reduce),\n"
       "# c4t = c4 at the cadence that trains (16 gradient steps per vector step, graphed events), c4m / c4mt = the same two with the in-launch mailbox exchange.\n"
       "# (1) bash scripts/scale.sh 1 -- N = 1, one rank, RCCL group of one\n") % RT
out += body('scale_n1.txt').rstrip() + '\n'
out += ("# (2) RANKS_PER_GPU=2 bash scripts/scale.sh 2 -- bench.py's world > 1 branch under torch.distributed.run with TWO ranks sharing the GPU (gloo for the host-side group: RCCL refuses two\n"
        "#     ranks per device; the mailbox legs map each other's mailbox over real hipIpc handles).  Each rank steps 65 536 envs, so the GPU does twice the work per vector step: the N = 2\n"
        "#     rows show that the path runs and what it costs on one device, not scaling.  all_reduce_ms here is the gloo all-reduce through the host.\n")
out += '\n'.join(l for l in body('scale_two_ranks_one_gpu.txt').rstrip().split('\n') if not l.startswith('c4m4')) + '\n'
tr = R + '/gpurun_out/two_rank_summary.txt'      # bash scripts/two_rank_trace.sh gpurun_out/two_rank | tee gpurun_out/two_rank_summary.txt
if os.path.exists(tr):
    out += '# (3) bash scripts/two_rank_trace.sh -- what the N = 2 rows of (2) are made of (VERDICT r5 item 5: "say what the two-rank mailbox number means"): rocprofv3 --kernel-trace of BOTH ranks\n#     (c3, c4, c4m; bench.py --steps 60 --windows 2), scripts/two_rank_timeline.py: per rank and kernel class the mean duration over the steady part of the run and what the PEER was\n#     running meanwhile.  train_xchg = the launch that carries the exchange (iqn_grad_reduce_adam[_xchg]: with two ranks on one GPU each plans for half of the CUs, so a step is two launches).\n'
    out += '\n'.join(l for l in open(tr).read().split('\n') if l.startswith('#') or l.startswith('rank') or l.startswith('  ')).rstrip() + '\n'
    out += "# Reading.  c3 / c4: the two ranks' kernels overlap freely (an act launch of 324-440 us shares the chip with the peer's act / step / reset launches); a gradient step's launches are 31 us.\n# c4m: the rank that reaches a gradient step FIRST (rank 1 here) sits in its exchange launch for 2 152 us on average -- 140 workgroups of 512 threads polling the peer's mailbox -- and during\n# 75 % of that time the peer is running ACT kernels that now take 594 us instead of 324-344: an act workgroup needs 154 of a CU's 160 KB of LDS, the 140 CUs that hold a polling workgroup\n# (33 KB of LDS) cannot take one, so the peer's 256 act workgroups queue up on the 116 CUs that are left.  The peer therefore needs ~2 ms instead of ~1 ms to arrive at its own gradient step,\n# and the waiting rank waits that long: 1.07 ms per vector step instead of 0.70.  This is the price of two ranks SHARING a GPU, not of the protocol: the polling workgroups hold CUs the\n# peer needs, and no back-off in the poll (s_sleep is already there) gives a CU's LDS back.  On a node with one rank per GPU the polling workgroups occupy the waiting rank's OWN GPU,\n# whose stream has nothing else to run until the step is complete -- what a rank pays there is (a) the arrival skew of the ranks, as with any collective, and (b) the protocol: +2.5 ... +2.9 us\n# per step at one rank (bench.py also.shared_learner_ws1.mailbox_us), plus, over xGMI, the gather of (world - 1) x 286 KB of 8-byte granules per rank and step: each peer's piece crosses\n# its own link, ~5 us at ~50 GB/s per direction -- an ESTIMATE: xGMI between two GPUs has never executed in this build.  RCCL's all-reduce of the same bucket: +17 us eager, +7 us inside captured\n# graphs at world size 1.\n"
open(P + 'scale.txt', 'w').write(out)
if os.path.exists(M + 'reset_under_act_ab.txt'):
    t = open(P + 'reset_under_act.txt').read() if os.path.exists(P + 'reset_under_act.txt') else ''
    lines = t.split('\n')
    i0 = next((i for i, l in enumerate(lines) if l.startswith('max_episode_steps')), None)
    i1 = max((i for i, l in enumerate(lines) if l.startswith('max_episode_steps')), default=None)
    if i0 is not None:
        lines[i0:i1 + 1] = [l for l in body('reset_under_act_ab.txt').split('\n') if l.startswith('max_episode_steps')]
        open(P + 'reset_under_act.txt', 'w').write('\n'.join(lines))
j = json.load(open(M + 'bench_default.json'))
print(j['value'] / 1e6, j['ms_per_step'], j['roofline']['launch_ms'], j['roofline_env_step']['launch_ms'])
# the constants bench.py puts into its line as `traffic_mb_profiled` must be THIS round's measurements
sys.path.insert(0, R)
import bench
want = {"step_append_f64": hbm(pm, 'step'), "act_split": hbm(pm, 'act'), "reset_under_act_f64": hbm(pm, 'reset_ua')}
bad = {k: (bench.PMC_TRAFFIC_MB.get(k), round(v, 1)) for k, v in want.items() if v == v and not (abs(bench.PMC_TRAFFIC_MB.get(k, -1) - v) <= 0.05 * v)}
if bad or RT not in bench.PMC_TRAFFIC_MB["source"]:
    raise SystemExit(f"bench.py PMC_TRAFFIC_MB disagrees with profiles/{RT}_full_loop_kernel_stats.txt (constant, measured): {bad}; source says {bench.PMC_TRAFFIC_MB['source']!r}")
