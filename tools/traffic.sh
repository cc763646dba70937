#!/bin/bash
# HBM traffic per launch of every (kernel, shape) of the default bench workload: two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE;
# never combined with tracing), FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950; the dispatches are joined with the
# launch sequence of one step (bench.py --launch-seq) by tools/shape_table.py.  Writes gpurun_out/<name>.json (copy to profiles/).
# usage: bash tools/traffic.sh <out-name> <seq.json> [bench args]
set -e
name=${1:-r03_traffic}; shift || true
seq=$1; shift || true
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/traffic_$name
rm -rf "$out"; mkdir -p "$out/f" "$out/w"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/f" -- python3 "$root/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-train --no-y-shape --no-other-configs "$@" > "$out/f/log.txt" 2>&1
echo "fetch pass done" >> "$out/progress.txt"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/w" -- python3 "$root/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-train --no-y-shape --no-other-configs "$@" > "$out/w/log.txt" 2>&1
echo "write pass done" >> "$out/progress.txt"
python3 "$root/tools/shape_table.py" "$out" "$seq" --traffic "$root/gpurun_out/$name.json" "$out/f" "$out/w"
