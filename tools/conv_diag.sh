#!/bin/bash
# Where does the generic conv kernel spend its time?  Rebuilds libvdx.so with one compile-time diagnostic at a time (results are
# WRONG, timing only; a runtime switch inside the MFMA loop perturbs the kernel by ~10 %) and runs tools/convprof.py.
# usage (on the GPU box): bash tools/conv_diag.sh > gpurun_out/conv_diag.txt     -- restores the normal build at the end
set -e
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$root/video_diffusion_nnx_amd/csrc"
for d in NONE NOMFMA NOFRAG NOWLOAD NOBAR "NOWLOAD -DVDX_DIAG_NOBAR" "NOMFMA -DVDX_DIAG_NOFRAG"; do
    echo "== VDX_DIAG_$d"
    touch conv_igemm.hip
    if [ "$d" = NONE ]; then make -s -j8 > /dev/null; else make -s -j8 EXTRA="-DVDX_DIAG_$d" > /dev/null; fi
    (cd "$root" && CONVPROF_B=${CONVPROF_B:-32} python3 tools/convprof.py 2>&1 | grep -E "^\\(128, 128, 32, 9|^\\(256, 256, 16, 9|^\\(512, 512, 8, 9|sum ms")
done
touch conv_igemm.hip; make -s -j8 > /dev/null
