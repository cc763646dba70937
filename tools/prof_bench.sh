#!/bin/bash
# kernel-trace profile of the default bench (no CPU baseline, no roofline leg) -> gpurun_out/prof_<tag>; prints the table.
# usage: bash tools/prof_bench.sh <tag> [extra bench.py args]
set -e
tag=${1:-x}; shift || true
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 "$root/bench.py" --no-cpu-baseline --no-roofline --steps 20 --warmup 3 "$@" > "$out/bench.log" 2>&1
tail -1 "$out/bench.log"
python3 "$root/tools/kstats.py" "$out" 22
