#!/bin/bash
# A/B timing of library variants INSIDE the denoising step (GPU box): for each variant the sampling leg of bench.py with its roofline
# leg (HIP events around every launch, per (kernel, shape)); prints ms/step and the rows whose kernel name matches <pattern>.
#   bash tools/ab_step.sh <pattern> <variant> [<variant> ...]        (variant "product" = the in-tree libvdx.so)
pat=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
for v in "$@"; do
  lib=$root/video_diffusion_nnx_amd/variants/libvdx_$v.so
  [ "$v" = product ] && lib=$root/video_diffusion_nnx_amd/libvdx.so
  VDX_LIB=$lib python3 "$root/bench.py" --no-cpu-baseline --no-train --no-y-shape --no-other-configs --steps 12 --warmup 3 > "$root/gpurun_out/ab_$v.json" 2> "$root/gpurun_out/ab_$v.log" || { echo "$v FAILED"; tail -3 "$root/gpurun_out/ab_$v.log"; continue; }
  python3 - "$root/gpurun_out/ab_$v.json" "$pat" "$v" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"== {sys.argv[3]}: {d['ms_per_step']:.3f} ms/step, {d['value']:.2f} frames/s; bracketed {d['roofline']['bracketed_ms_per_step']:.2f} ms")
for r in d['roofline']['all_kernels']:
    if any(p in r['kernel'] for p in sys.argv[2].split('|')) and r['ms_per_step'] > 0.15:
        print(f"   {r['kernel']:24s} {r['shape'][:60]:60s} n {r['launches_per_step']} {r['avg_launch_us']:8.1f} us  mfma {r['mfma_frac']:.3f} hbm {r['hbm_frac']:.3f}")
PY
done
