"""Per-(kernel, shape) table of a rocprofv3 run of bench.py: joins the launches of the trace with the launch sequence of ONE denoising
step that `bench.py --launch-seq seq.json` wrote (library hook vdx_set_launch_hook: kernel, template arguments + shape, algorithmic
FLOPs / bytes per launch).  One symbol serves several levels of the network, so per-symbol averages of `--stats` mix launches whose
durations differ 4x; this table does not.

    python tools/shape_table.py <rocprof dir with *kernel_trace.csv> <seq.json> [--md] [--skip-steps 3] [--take N] [--traffic out.json <pmc dir f> <pmc dir w>] [--pmc N]

Alignment: the trace is walked in start-time order with a cursor into the step's sequence; a dispatch whose name contains the
expected kernel name is assigned to that entry (the memset entry is optional: graph replays may not show it as a kernel), anything
else (weight packing, randn, the training leg ...) resets the cursor.  Only COMPLETE steps are kept, the first --skip-steps of them
(eager warm-up and capture) are dropped, so the averages are over graph replays."""
import collections
import csv
import json
import pathlib
import sys

MFMA_PEAK = 2500e12
HBM_PEAK = 8e12


def load_rows(root, pattern, name_col='Kernel_Name'):
    files = sorted(pathlib.Path(root).rglob(pattern))
    if not files:
        sys.exit(f'no {pattern} under {root}')
    rows = []
    for f in files:
        rows += list(csv.DictReader(open(f)))
    return rows


def matches(kernel, name):
    """`kernel` = __global__ function name without template arguments; `name` = what rocprofv3 prints (namespace, template arguments, maybe the
    parameter list)."""
    if kernel == 'fillBufferAligned':
        return 'fillBuffer' in name
    base = name.split('(')[0].replace('void ', '').strip()
    base = base.split('<')[0]
    return base == kernel or base.endswith('::' + kernel)


def align(names, seq):
    """names: dispatch kernel names in time order.  Returns a list of steps, each a list of (dispatch index, sequence index)."""
    steps, cur, j = [], [], 0
    n = len(seq)
    i = 0
    while i < len(names):
        nm = names[i]
        while j < n and seq[j]['kernel'] == 'fillBufferAligned' and 'fillBuffer' not in nm:
            j += 1                                    # optional entry not present as a kernel
        if j < n and matches(seq[j]['kernel'], nm):
            cur.append((i, j)); j += 1; i += 1
            while j < n and seq[j]['kernel'] == 'fillBufferAligned' and (i >= len(names) or 'fillBuffer' not in names[i]):
                j += 1
            if j == n:
                steps.append(cur); cur, j = [], 0
            continue
        if cur:                                       # mismatch inside a step: drop the partial step, retry this dispatch at the start
            cur, j = [], 0
            continue
        i += 1
    return steps


def main():
    args = sys.argv[1:]
    md = '--md' in args
    skip = 3
    if '--skip-steps' in args:
        skip = int(args[args.index('--skip-steps') + 1])
    traffic = None
    if '--traffic' in args:
        k = args.index('--traffic')
        traffic = args[k + 1:k + 4]
    root, seqf = args[0], args[1]
    meta = json.load(open(seqf))
    seq = meta['sequence']
    keys = [f"{e['kernel']} | {e['shape']}" for e in seq]
    if traffic:
        out, fdir, wdir = traffic
        acc = collections.defaultdict(lambda: {'f': [], 'w': []})
        for tag, d, counter in (('f', fdir, 'FETCH_SIZE'), ('w', wdir, 'WRITE_SIZE')):
            rows = [r for r in load_rows(d, '*counter_collection.csv') if r['Counter_Name'] == counter]
            rows.sort(key=lambda r: int(r['Dispatch_Id']))
            steps = align([r['Kernel_Name'] for r in rows], seq)
            for st in steps[1:]:
                for i, j in st:
                    acc[keys[j]][tag].append(float(rows[i]['Counter_Value']))
        kernels = {}
        for k, d in acc.items():
            f = 2.0 * 1024.0 * sum(d['f']) / max(len(d['f']), 1)          # KB -> B, doubled (gfx950: MI355X_MICROARCH.md)
            w = 1024.0 * sum(d['w']) / max(len(d['w']), 1)
            kernels[k] = {'launches_sampled': len(d['f']), 'fetch_bytes_per_launch': f, 'write_bytes_per_launch': w, 'hbm_bytes_per_launch': f + w}
        json.dump({'note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of `python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline '
                           '--no-train --no-y-shape`; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 64 B per 128-B request; calibrated for '
                           '16-B-per-lane loads only); per (kernel | shape) averages, dispatches joined with the launch sequence by tools/shape_table.py',
                   'batch': meta['batch'], 'mode': meta['mode'], 'dim': meta['dim'], 'act': meta['act'], 'kernels': kernels}, open(out, 'w'), indent=1)
        print('wrote', out, len(kernels), 'keys')
        return
    if '--pmc' in args:
        # per-(kernel | shape) counter table of tools/pmc_passes.sh runs of tools/bench_short.py: every dispatch of every counter_collection.csv
        # under <root> is joined with the launch sequence; MFMA busy, LDS activity, bank-conflict share, split of wave life (MI355X_MICROARCH.md
        # counter units: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* in quad-cycles per wave, SQ_VALU_MFMA_BUSY_CYCLES per SIMD, GRBM_GUI_ACTIVE per XCD)
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in sorted(pathlib.Path(root).rglob('*counter_collection.csv')):
            rows = list(csv.DictReader(open(f)))
            by_counter = collections.defaultdict(list)
            for r in rows:
                by_counter[r['Counter_Name']].append(r)
            for cname, rs in by_counter.items():
                rs.sort(key=lambda r: int(r['Dispatch_Id']))
                for st in align([r['Kernel_Name'] for r in rs], seq)[1:]:
                    for i, j in st:
                        acc[keys[j]][cname].append(float(rs[i]['Counter_Value']))
        out = []
        for k, d in acc.items():
            if 'GRBM_GUI_ACTIVE' not in d or 'SQ_WAVE_CYCLES' not in d:
                continue
            avg = lambda c: sum(d[c]) / len(d[c]) if d.get(c) else 0.0
            cyc = avg('GRBM_GUI_ACTIVE') / 8
            wc = avg('SQ_WAVE_CYCLES')
            out.append((cyc * len(d['GRBM_GUI_ACTIVE']), k, len(d['GRBM_GUI_ACTIVE']), cyc, avg('SQ_VALU_MFMA_BUSY_CYCLES') / 1024 / cyc,
                        avg('SQ_LDS_IDX_ACTIVE') / 256 / cyc, avg('SQ_LDS_BANK_CONFLICT') / max(avg('SQ_LDS_IDX_ACTIVE'), 1.0),
                        avg('SQ_WAIT_ANY') / wc, avg('SQ_WAIT_INST_ANY') / wc, avg('SQ_ACTIVE_INST_ANY') / wc))
        out.sort(reverse=True)
        print('| kernel | template arguments, shape | launches sampled | kernel cycles (GUI_ACTIVE / 8) | MFMA busy | LDS array active | bank-conflict share of LDS cycles | '
              'wave life parked (`SQ_WAIT_ANY`) | issue-stalled (`SQ_WAIT_INST_ANY`) | issuing |')
        print('|---|---|---|---|---|---|---|---|---|---|')
        for r in out[:int(args[args.index('--pmc') + 1]) if args[args.index('--pmc') + 1].isdigit() else 30]:
            kn, sh = r[1].split(' | ', 1)
            print(f'| `{kn}` | {sh} | {r[2]} | {r[3] / 1e3:.0f} k | {100 * r[4]:.0f} % | {100 * r[5]:.0f} % | {100 * r[6]:.0f} % | {100 * r[7]:.0f} % | {100 * r[8]:.0f} % | {100 * r[9]:.0f} % |')
        return
    rows = load_rows(root, '*kernel_trace.csv')
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    steps = align([r['Kernel_Name'] for r in rows], seq)
    if len(steps) <= skip:
        sys.exit(f'only {len(steps)} complete steps found in the trace')
    steps = steps[skip:]
    if '--take' in args:                              # (the eager steps of the roofline leg, bracketed by event markers, come last: leave them out)
        steps = steps[:int(args[args.index('--take') + 1])]
    agg = collections.OrderedDict()
    span = 0.0
    for st in steps:
        span += (int(rows[st[-1][0]]['End_Timestamp']) - int(rows[st[0][0]]['Start_Timestamp'])) / 1e3
        for i, j in st:
            d = agg.setdefault(keys[j], {'n': 0, 'us': 0.0, 'flops': seq[j]['flops'], 'bytes': seq[j]['bytes'], 'sym': rows[i]['Kernel_Name'], 'scratch': rows[i].get('Scratch_Size', rows[i].get('Private_Segment_Size', '')),
                                         'vgpr': rows[i].get('VGPR_Count', ''), 'lds': rows[i].get('LDS_Block_Size', '')})
            d['n'] += 1
            d['us'] += (int(rows[i]['End_Timestamp']) - int(rows[i]['Start_Timestamp'])) / 1e3
    ns = len(steps)
    tot = sum(d['us'] for d in agg.values()) / ns
    print(f'# {ns} graph-replay steps joined with {seqf}: {len(seq)} launches / step, kernel time {tot / 1e3:.3f} ms / step, span {span / ns / 1e3:.3f} ms / step')
    order = sorted(agg.items(), key=lambda kv: -kv[1]['us'])
    if md:
        print('| kernel | template arguments, shape | launches / step | us / launch | ms / step | % | GFLOP | TFLOP/s | MFMA frac | alg. MB | TB/s | HBM frac | scratch B |')
        print('|---|---|---|---|---|---|---|---|---|---|---|---|---|')
    for key, d in order:
        per = d['n'] // ns
        us = d['us'] / d['n']
        tf = d['flops'] / (us * 1e-6) if us > 0 else 0.0
        bw = d['bytes'] / (us * 1e-6) if us > 0 else 0.0
        k, sh = key.split(' | ', 1)
        if md:
            print(f"| `{k}` | {sh} | {per} | {us:.1f} | {us * per / 1e3:.3f} | {100 * us * per / tot:.1f} | {d['flops'] / 1e9:.1f} | {tf / 1e12:.0f} | {tf / MFMA_PEAK:.2f} | "
                  f"{d['bytes'] / 1e6:.0f} | {bw / 1e12:.2f} | {bw / HBM_PEAK:.2f} | {d['scratch']} |")
        else:
            print(f"{k:28s} {sh[:70]:70s} n {per:2d} {us:8.1f} us {us * per / 1e3:7.3f} ms {100 * us * per / tot:5.1f}% {tf / 1e12:6.0f} TF ({tf / MFMA_PEAK:.2f}) "
                  f"{bw / 1e12:5.2f} TB/s ({bw / HBM_PEAK:.2f}) scratch {d['scratch']}")


if __name__ == '__main__':
    main()
