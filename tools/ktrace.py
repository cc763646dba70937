"""Per-launch view of a rocprofv3 --kernel-trace run: groups the launches of the LAST `steps`-th of the trace by
(kernel, grid, workgroup) and prints count, average duration and total per group, plus the idle time between kernels.
    python tools/ktrace.py gpurun_out/prof_train_x [n_parts=6] [top=60]
(n_parts: the trace is cut into that many equal launch-count parts and the last one is shown: use warm-up + timed steps.)"""
import csv
import pathlib
import sys
from collections import defaultdict

root = pathlib.Path(sys.argv[1])
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 6
top = int(sys.argv[3]) if len(sys.argv) > 3 else 60
files = sorted(root.rglob('*kernel_trace.csv'))
if not files:
    sys.exit(f'no kernel_trace.csv under {root}')
rows = list(csv.DictReader(open(files[-1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = len(rows) // parts
rows = rows[len(rows) - n:]
groups = defaultdict(lambda: [0, 0.0])
busy = 0.0
for r in rows:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    name = r['Kernel_Name'].replace('vdx::', '').split('(')[0].replace('void ', '')[:58]
    grid = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z'])
    wg = int(r['Workgroup_Size_X']) * int(r['Workgroup_Size_Y']) * int(r['Workgroup_Size_Z'])
    g = groups[(name, grid // max(wg, 1), wg)]
    g[0] += 1
    g[1] += d
    busy += d
span = (int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])) / 1e3
print(f'# {files[-1].name}: last 1/{parts} of the trace = {len(rows)} launches, span {span / 1e3:.2f} ms, kernel time {busy / 1e3:.2f} ms')
for (name, wgs, wg), (cnt, tot) in sorted(groups.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f'{name:58s} wgs {wgs:6d} x {wg:4d} n {cnt:4d} avg {tot / cnt:8.1f} us tot {tot / 1e3:7.2f} ms')
# overlap between queues (two-stream backward): time with >= 2 kernels in flight
ev = []
for r in rows:
    ev.append((int(r['Start_Timestamp']), 1)); ev.append((int(r['End_Timestamp']), -1))
ev.sort()
depth = 0; last = ev[0][0]; t_by_depth = defaultdict(float)
for t, d in ev:
    t_by_depth[depth] += (t - last) / 1e6; last = t; depth += d
print('# time by number of kernels in flight (ms):', {k: round(v, 2) for k, v in sorted(t_by_depth.items())})
print('# queues:', sorted({r['Queue_Id'] for r in rows}))
# the longest idle gaps (nothing in flight) and the kernels either side
ivs = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('vdx::', '').split('(')[0].replace('void ', '')[:40]) for r in rows))
gaps = []
cur_end, cur_name = ivs[0][1], ivs[0][2]
for st, en, nm in ivs[1:]:
    if st > cur_end:
        gaps.append(((st - cur_end) / 1e3, cur_name, nm))
    if en > cur_end:
        cur_end, cur_name = en, nm
gaps.sort(reverse=True)
print(f'# {len(gaps)} idle gaps, total {sum(g[0] for g in gaps) / 1e3:.2f} ms; histogram (us):',
      {b: sum(1 for g in gaps if lo <= g[0] < hi) for b, lo, hi in (('<2', 0, 2), ('2-5', 2, 5), ('5-10', 5, 10), ('10-20', 10, 20), ('>20', 20, 1e9))})
for g in gaps[:12]:
    print(f'  {g[0]:7.1f} us  after {g[1]:40s} before {g[2]}')
