#!/bin/bash
set -u
# c1_wgrad_mfma_kernel grid cap (ablation build, GI_C1W_GRID) on a batch with a generator update: kernel durations per variant
set -o pipefail
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the tools run on the GPU box through gpurun)}
export GI_LIB_PATH=$R/gan-inpainting_amd/libganinpaint_abl.so
cd /tmp && export TMPDIR=/tmp
for cap in 512 1024 2048; do
  export GI_C1W_GRID=$cap
  OUT=$R/gpurun_out/c1w_$cap
  rm -rf $OUT; mkdir -p $OUT
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/tools/step_chain.py 3 gen > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
  CSV=$(find $OUT -name "*kernel_trace.csv" | head -1)
  python3 $R/tools/chain_table.py $CSV mask_apply_kernel > $OUT/chain.txt
  echo "cap=$cap: wgrad $(grep c1_wgrad_mfma $OUT/chain.txt | awk '{print $2}' | tr '\n' ' ') reduce $(grep c1_wgrad_reduce $OUT/chain.txt | awk '{print $2}' | tr '\n' ' ') col $(grep c1_col_kernel $OUT/chain.txt | awk '{print $2}' | tr '\n' ' ')"
done
