"""Copy the summaries scripts/measure_round3.sh left under gpurun_out/m3 into profiles/r03_* (headers kept, bodies replaced)."""
import os, re, shutil
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
M, P = R + '/gpurun_out/m3/', R + '/profiles/'
body = lambda f: open(M + f).read()
old = open(P + 'r03_full_loop_kernel_stats.txt').read().split('\n')
hdr = old[:3]
pm_i = [i for i, l in enumerate(old) if l.startswith('# PMC passes')][0]
pmc_hdr = old[pm_i:pm_i + 3]
pm = body('pmc_summary.txt')
def mean(k, c):
    return float(re.search(r'^%s\s+%s\s+n=\s*\d+ mean=\s*([\d.]+)' % (k, c), pm, re.M).group(1)) / 1e3
out = '\n'.join(hdr) + '\n' + body('prof_loop_summary.txt').rstrip() + '\n\n' + '\n'.join(pmc_hdr) + '\n' + pm.rstrip() + '\n'
out += ('# derived (KB = 1000 B): iqn_train_fwdbwd WRITE_SIZE %.1f MB per launch for the 18.3 MB of partial gradients it stores (128 x 35 788 floats, non-temporal\n'
        '# 16-byte stores), reads 2 x %.2f = %.1f MB; iqn_grad_reduce reads 2 x %.2f = %.1f MB (the partials + the staged batch\'s ring rows); act kernel: %.2f M MFMA-busy\n'
        '# cycles (372 x 16 x 65 536 = 390.07 M), 2 x %.2f + %.2f = %.1f MB of HBM traffic for 15.5 MB of algorithmic input / output; step kernel (float64,\n'
        '# with replay append): 2 x %.2f + %.2f = %.1f MB.\n') % (
    mean('train', 'WRITE_SIZE'), mean('train', 'FETCH_SIZE'), 2 * mean('train', 'FETCH_SIZE'), mean('reduce', 'FETCH_SIZE'), 2 * mean('reduce', 'FETCH_SIZE'),
    mean('act', 'SQ_VALU_MFMA_BUSY_CYCLES') / 1e3, mean('act', 'FETCH_SIZE'), mean('act', 'WRITE_SIZE'), 2 * mean('act', 'FETCH_SIZE') + mean('act', 'WRITE_SIZE'),
    mean('step', 'FETCH_SIZE'), mean('step', 'WRITE_SIZE'), 2 * mean('step', 'FETCH_SIZE') + mean('step', 'WRITE_SIZE'))
open(P + 'r03_full_loop_kernel_stats.txt', 'w').write(out)
print(out[-520:])
for src, dst in (('prof_g16_summary.txt', 'r03_train_cadence_kernel_stats.txt'), ('prof_h2_summary.txt', 'r03_two_halves_kernel_stats.txt')):
    o = open(P + dst).read().split('\n')
    h = [l for l in o if l.startswith('# r03') or l.startswith('# Kernel durations')]
    open(P + dst, 'w').write('\n'.join(h) + '\n' + body(src).rstrip() + '\n')
shutil.copy(M + 'bench_default.json', P + 'r03_bench_default.json')
shutil.copy(M + 'bench_halves2.json', P + 'r03_bench_halves2.json')
