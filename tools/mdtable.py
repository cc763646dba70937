"""Markdown table of a rocprofv3 kernel_stats.csv (+ optional per-kernel HBM MB/launch from a tools/traffic.sh json).
    python tools/mdtable.py profiles/x_kernel_stats.csv [profiles/r02_traffic.json] [top=28]"""
import csv
import json
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
traffic = {}
top = 28
for a in sys.argv[2:]:
    if a.endswith('.json'):
        traffic = json.load(open(a)).get('kernels', {})
    else:
        top = int(a)
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
total = sum(float(r['TotalDurationNs']) for r in rows)
print(f'total kernel time {total / 1e6:.1f} ms\n')
print('| kernel | calls | avg us | total ms | % |' + (' HBM MB/launch (PMC) |' if traffic else ''))
print('|---|---|---|---|---|' + ('---|' if traffic else ''))
for r in rows[:top]:
    full = r['Name']
    name = full.replace('vdx::', '').replace('void ', '').split('(')[0][:70]
    line = f"| {name} | {int(r['Calls'])} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['TotalDurationNs']) / 1e6:.1f} | {100 * float(r['TotalDurationNs']) / total:.2f} |"
    if traffic:
        key = full.replace('void ', '').split('(')[0]
        t = traffic.get(key)
        line += f" {t['hbm_bytes_per_launch'] / 1e6:.1f} |" if t else '  |'
    print(line)
