#!/bin/bash
set -u
# Timing-only ablations of the dominant kernel (igemm6, generator u3 shape): builds the library with -DGI_ABLATION on the
# GPU box (the shipped .so is restored afterwards) and times each variant with bench.py --kernel-only.
# usage: tools/ablate_igemm6.sh   (through gpurun)
set -o pipefail
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the tools run on the GPU box through gpurun)}
cd $R
cp gan-inpainting_amd/libganinpaint.so /tmp/libganinpaint.so.keep
touch gan-inpainting_amd/csrc/igemm5.hip
bash gan-inpainting_amd/csrc/build.sh -DGI_ABLATION > /tmp/build_abl.log 2>&1 || { tail -20 /tmp/build_abl.log; exit 1; }
export GI_IGEMM6=1
for d in ${ABL:-0 1 2 4 8 16 3 5 6 7 9 23 32 39}; do
  echo -n "DBG=$d  "
  GI_IGEMM6_DBG=$d python3 bench.py --kernel-only --kernel-iters 200 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f us  %.0f TFLOP/s-equivalent' % (d['avg_ms']*1e3, d['tflops']))"
done
cp /tmp/libganinpaint.so.keep gan-inpainting_amd/libganinpaint.so
touch gan-inpainting_amd/csrc/igemm5.hip
