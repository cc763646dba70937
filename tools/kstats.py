"""Summarise a rocprofv3 --kernel-trace --stats run: prints the per-kernel table (calls, avg us, total ms, %) from the
*kernel_stats.csv found under the given directory.  Usage: python tools/kstats.py gpurun_out/prof [top_n]"""
import csv
import pathlib
import sys

root = pathlib.Path(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
files = sorted(root.rglob('*kernel_stats.csv'))
if not files:
    sys.exit(f'no kernel_stats.csv under {root}')
rows = list(csv.DictReader(open(files[-1])))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
total = sum(float(r['TotalDurationNs']) for r in rows)
print(f'# {files[-1].name}: total kernel time {total / 1e6:.1f} ms')
for r in rows[:top]:
    name = r['Name'].replace('vdx::', '').split('(')[0][:60]
    print(f"{name:60s} {int(r['Calls']):6d} {float(r['AverageNs']) / 1e3:9.1f} us {float(r['TotalDurationNs']) / 1e6:9.1f} ms {100 * float(r['TotalDurationNs']) / total:6.2f}%")
