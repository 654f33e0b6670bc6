#!/bin/bash
set -u
# Timing-only ablations of igemm8 on the u3 shape (ablation build, WRONG results): which part of the step costs what.
# usage (GPU box): tools/ablate_igemm8.sh   (needs gan-inpainting_amd/libganinpaint_abl.so: GI_OUT=../libganinpaint_abl.so GI_BUILD_DIR=build_abl build.sh -DGI_ABLATION)
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the tools run on the GPU box through gpurun)}
export GI_LIB_PATH=$R/gan-inpainting_amd/libganinpaint_abl.so
for d in 0 1 2 4 8 16 32 5 6 3 7 13 15 39; do
  export GI_IGEMM8_DBG=$d
  echo -n "DBG=$d  "; python3 $R/bench.py --kernel-only --kernel-iters 200 2>$R/gpurun_out/ablate8_err.log | cut -c1-200 || { echo "failed:"; tail -3 $R/gpurun_out/ablate8_err.log; }
done
