#!/bin/bash
# The measurement set of a round, on the GPU box: bench line, per-(kernel, shape) table of the sampling step, HBM traffic (two PMC passes),
# counter table (three PMC passes), training trace and its counter table.  Everything lands in gpurun_out/<tag>_*; copy what is judged
# into profiles/.   usage: bash tools/final_profiles.sh <tag> [part: a|b|all]
tag=${1:-r03}; part=${2:-all}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd "$root"
if [ "$part" = a ] || [ "$part" = all ]; then
  python3 bench.py > "$out/${tag}_bench.json" 2> "$out/${tag}_bench.log"; tail -2 "$out/${tag}_bench.log"
  bash tools/prof_step.sh "$tag" > "$out/${tag}_step.txt" 2>&1; head -3 "$out/${tag}_step.txt"
  bash tools/traffic.sh "${tag}_traffic" "$out/prof_step_$tag/seq.json" > "$out/${tag}_traffic.txt" 2>&1; tail -1 "$out/${tag}_traffic.txt"
fi
if [ "$part" = b ] || [ "$part" = all ]; then
  PMC_GROUPS="0 3 8" bash tools/pmc_passes.sh "$tag" tools/bench_short.py > "$out/${tag}_pmc_raw.txt" 2>&1
  # (part b may run on another box than part a: gpurun_out/ does not travel, the committed launch sequence does)
  seq="$out/prof_step_$tag/seq.json"; [ -f "$seq" ] || seq="$root/profiles/${tag}_launch_seq.json"
  python3 tools/shape_table.py "$out/pmc_$tag" "$seq" --pmc 32 > "$out/${tag}_pmc_step.md" 2>&1; head -5 "$out/${tag}_pmc_step.md"
  bash tools/prof_train.sh "$tag" > "$out/${tag}_train.txt" 2>&1
  python3 tools/ktrace.py "$out/prof_train_$tag" 6 70 >> "$out/${tag}_train.txt" 2>&1; head -4 "$out/${tag}_train.txt"
  PMC_GROUPS="0 3 8" bash tools/pmc_passes.sh "${tag}train" tools/train_bench.py > "$out/${tag}_pmc_train_raw.txt" 2>&1
  python3 tools/pmctable.py "$out/${tag}_pmc_train_raw.txt" 30 > "$out/${tag}_pmc_train.md" 2>&1; head -5 "$out/${tag}_pmc_train.md"
fi
