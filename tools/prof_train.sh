#!/bin/bash
# kernel-trace profile of tools/train_bench.py -> gpurun_out/prof_train_<tag>; prints the per-kernel table.
set -e
tag=${1:-x}; shift || true
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_train_$tag
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 "$root/tools/train_bench.py" --steps 4 "$@" > "$out/log.txt" 2>&1
tail -1 "$out/log.txt"
python3 "$root/tools/kstats.py" "$out" 40
