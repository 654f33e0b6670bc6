#!/bin/bash
set -u
# Timing-only ablations of igemm8's gather mode on the critic's conv2 shape (64 images, 128x128x64 -> 64x64x128; ablation build, WRONG
# results): 1 no LDS-DMA, 2 no MFMA, 4 no fragment reads, 16 no per-step barrier, 32 no epilogue, 5 = 1 + 4, 7 = 1 + 2 + 4, 39 = 1 + 2 + 4 + 32.
# usage (GPU box): tools/ablate_igemm8_gather.sh
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the tools run on the GPU box through gpurun)}
export GI_LIB_PATH=$R/gan-inpainting_amd/libganinpaint_abl.so
for d in 0 1 2 4 16 32 5 7 39 0; do
  export GI_IGEMM8_DBG0=$d
  echo -n "DBG=$d  "; python3 $R/tools/time_conv2.py 2>$R/gpurun_out/ablate8g_err.log || { echo "failed:"; tail -3 $R/gpurun_out/ablate8g_err.log; }
done
