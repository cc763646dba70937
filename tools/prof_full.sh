#!/bin/bash
# rocprofv3 --kernel-trace --stats of the default bench command (with its roofline leg, no CPU baseline) -> gpurun_out/prof_full_<tag>
set -e
tag=${1:-x}; shift || true
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_full_$tag
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 "$root/bench.py" --no-cpu-baseline "$@" > "$out/bench.json" 2> "$out/bench.log"
cp "$out"/*/*kernel_stats.csv "$out/kernel_stats.csv"
python3 "$root/tools/kstats.py" "$out" 40
