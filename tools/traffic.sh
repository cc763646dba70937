#!/bin/bash
# HBM traffic per launch of every kernel of the default bench workload: two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE;
# never combined with tracing), FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950.  Writes profiles/<name>.json.
# usage: bash tools/traffic.sh <out-name> [bench args]
set -e
name=${1:-r01_traffic}; shift || true
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/traffic_$name
rm -rf "$out"; mkdir -p "$out/f" "$out/w"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/f" -- python3 "$root/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-train "$@" > "$out/f/log.txt" 2>&1
echo "fetch pass done" >> "$out/progress.txt"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/w" -- python3 "$root/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-train "$@" > "$out/w/log.txt" 2>&1
echo "write pass done" >> "$out/progress.txt"
python3 - "$out" "$root/gpurun_out/$name.json" "$@" <<'PY'
import collections, csv, json, pathlib, sys
root, dst = pathlib.Path(sys.argv[1]), sys.argv[2]
def collect(sub, counter):
    acc = collections.defaultdict(list)
    for f in (root / sub).rglob('*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter:
                acc[r['Kernel_Name'].split('(')[0].replace('void ', '').strip()].append(float(r['Counter_Value']))
    return acc
fe, wr = collect('f', 'FETCH_SIZE'), collect('w', 'WRITE_SIZE')
line = [l for l in open(root / 'f' / 'log.txt') if l.startswith('{"metric"')][-1]
cfg = json.loads(line)['config']
kernels = {}
for k in sorted(set(fe) | set(wr)):
    f = 2.0 * 1024.0 * sum(fe.get(k, [0])) / max(len(fe.get(k, [0])), 1)       # KB -> B, doubled (gfx950)
    w = 1024.0 * sum(wr.get(k, [0])) / max(len(wr.get(k, [0])), 1)
    kernels[k] = {'launches_sampled': len(fe.get(k, [])), 'fetch_bytes_per_launch': f, 'write_bytes_per_launch': w, 'hbm_bytes_per_launch': f + w}
args = sys.argv[3:]
json.dump({'note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of `python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline '
                   + ' '.join(args) + '`; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 64 B per 128-B request; calibrated for 16-B-per-lane '
                   'loads only); averages over all launches of a symbol',
           'batch': cfg['batch_per_gpu'], 'mode': cfg['mfma_operands'], 'dim': 64, 'act': 'bf16' if 'activations bf16' in cfg['storage'] else 'f32',
           'kernels': kernels}, open(dst, 'w'), indent=1)
print('wrote', dst, len(kernels), 'kernels')
PY
