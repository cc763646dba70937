#!/bin/bash
# Per-(kernel, shape) profile of the SAMPLING step alone (no training leg, no Y-shape leg in the process): rocprofv3 --kernel-trace --stats of
# bench.py with its roofline leg (which also writes the step's launch sequence), joined by tools/shape_table.py -> gpurun_out/prof_step_<tag>/
# usage: bash tools/prof_step.sh <tag> [extra bench.py args]
set -e
tag=${1:-x}; shift || true
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_step_$tag
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 "$root/bench.py" --no-cpu-baseline --no-train --no-y-shape --no-other-configs --steps 20 --warmup 3 \
    --launch-seq "$out/seq.json" "$@" > "$out/bench.json" 2> "$out/bench.log"
cp "$out"/*/*kernel_stats.csv "$out/kernel_stats.csv"
python3 "$root/tools/shape_table.py" "$out" "$out/seq.json" --skip-steps 2 --take 20 --md > "$out/shape_table.md"
python3 "$root/tools/shape_table.py" "$out" "$out/seq.json" --skip-steps 2 --take 20 | head -45
tail -c 600 "$out/bench.json"
