"""Writes profiles/<tag>_summary.md and profiles/<tag>_train_summary.md from the artefacts tools/final_profiles.sh produced and that were copied
into profiles/ (bench line, traffic JSON) / left in gpurun_out/ (training trace text).  usage: python tools/mk_summary.py r03"""
import json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else 'r03'
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda *a: os.path.join(root, *a)
d = json.loads(open(P('profiles', f'{tag}_bench_line.json')).read().strip().splitlines()[-1])
t = json.load(open(P('profiles', f'{tag}_traffic.json')))['kernels']
rows = d['roofline']['all_kernels'][:14]
tab = ['| kernel | shape | launches / step | µs / launch (brackets) | share | bound | MFMA frac | HBM frac | HBM MB / launch (PMC) vs algorithmic |', '|---|---|---|---|---|---|---|---|---|']
for r in rows:
    tr = t.get(f"{r['kernel']} | {r['shape']}")
    tb = (f"{tr['hbm_bytes_per_launch']/1e6:.0f} vs {r['algorithmic_mb_per_launch']:.0f} ({tr['hbm_bytes_per_launch']/1e6/max(r['algorithmic_mb_per_launch'],1e-9):.2f}×)" if tr else '—')
    tab.append(f"| `{r['kernel']}` | {r['shape']} | {r['launches_per_step']} | {r['avg_launch_us']:.0f} | {100*r['share_of_step']:.1f} % | {r['bound']} | {r['mfma_frac']:.2f} | {r['hbm_frac']:.2f} | {tb} |")
old = open(P('profiles', f'{tag}_summary.md')).read()
how = old[old.index('## How the numbers are produced'):old.index("## The step's largest rows")]
tail = old[old.index('## What changed in the second half of the round'):]
oc, y, trn, cb, rf = d['other_configs'], d['y_shape'], d['train'], d['cpu_baseline'], d['roofline']
traffic = f"{rf['traffic']/1e6:.0f} MB per launch (PMC, committed `{tag}_traffic.json`)" if rf.get('traffic') else 'n/a'
txt = f"""# Round 3 — profile summary (N shape: dim 64, 16 f × 64 × 64, B = 64 per GPU, bf16 operands + bf16 activation storage)

Bench line of the profile box (`profiles/{tag}_bench_line.json`, `python bench.py`): **{d['value']:.2f} denoised frames/s, {d['ms_per_step']:.3f} ms per step** (boxes of
the pool differ by ± 3 %: 20.2–20.9 ms were seen for the final code; the first half of the round ended at 21.9–22.6); Y shape {y['ms_per_step']:.2f} ms =
{y['frames_per_s']:.1f} frames/s ({y['tflops']:.0f} TFLOP/s = {y['tflops']/y['n_shape_tflops']:.2f} of the N shape's {y['n_shape_tflops']:.0f}); training {trn['ms_per_step']:.2f} ms per step = {trn['samples_per_s']:.1f} samples/s
(gradients bit-reproducible); CPU restatement {cb['value']:.4f} frames/s on {cb['cores']} cores ({cb['train']['value']:.2f} training samples/s);
`other_configs`: configs[3] (dim 128, 32 f × 128², DDIM) {oc['configs3_f16']['ms_per_step']:.1f} ms per step at B = 1 with fp16 operands ({oc['configs3_bf16']['ms_per_step']:.1f} with bf16 operands + bf16 storage),
configs[4] (cond 768 + guidance, B = 32) {oc['configs4_bf16attn']['ms_per_step']:.1f} ms per step ({oc['configs4_fp8attn']['ms_per_step']:.1f} with fp8 attention cores).

{how}## The step's largest rows (bench brackets; the trace's durations are in `{tag}_step_table.md` and agree within 1–4 %)

""" + '\n'.join(tab) + f"""

Bracketed sum {rf['bracketed_ms_per_step']:.2f} ms against {d['ms_per_step']:.2f} ms per graph-replayed step.  The JSON line's `roofline` = the first row:
`{rf['kernel']}`, bound {rf['bound']}, frac {rf['frac']:.2f}, traffic {traffic}.

""" + tail
open(P('profiles', f'{tag}_summary.md'), 'w').write(txt)
# training summary: the text block of gpurun_out/<tag>_train.txt behind the fixed header of the committed file
tp = P('gpurun_out', f'{tag}_train.txt')
if os.path.exists(tp):
    tr = open(tp).read()
    body = '\n'.join(tr[tr.index('# '):].split('\n')[:110])
    oldt = open(P('profiles', f'{tag}_train_summary.md')).read()
    head = oldt[:oldt.index('```')]
    head = head.replace(head[head.index('**'):head.index('** on one MI355X')], f"**{trn['ms_per_step']:.2f} ms per step = {trn['samples_per_s']:.1f} samples/s")
    open(P('profiles', f'{tag}_train_summary.md'), 'w').write(head + '```\n' + body + '\n```\n')
print('written')
