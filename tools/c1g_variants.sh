#!/bin/bash
# c1_gather_mfma_kernel experiment variants (ablation build): GI_C1G_VAR bits 1 = no loads, 2 = no stores, 4 = stores through an LDS
# transpose (full 128-byte lines per instruction); GI_C1G_BPC = workgroups per CU. Prints the kernel's durations in a critic-only batch.
set -o pipefail
R=$GRAFT_REPO_ROOT
export GI_LIB_PATH=$R/gan-inpainting_amd/libganinpaint_abl.so
cd /tmp && export TMPDIR=/tmp
for cfg in "0 8" "1 8" "2 8" "3 8" "4 8" "0 16" "4 16" "0 4" "4 4"; do
  set -- $cfg
  export GI_C1G_VAR=$1 GI_C1G_BPC=$2
  OUT=$R/gpurun_out/c1g_$1_$2
  rm -rf $OUT; mkdir -p $OUT
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/tools/step_chain.py 3 critic > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
  CSV=$(find $OUT -name "*kernel_trace.csv" | head -1)
  python3 $R/tools/chain_table.py $CSV mask_apply_kernel > $OUT/chain.txt
  echo "VAR=$1 BPC=$2: $(grep c1_gather_mfma $OUT/chain.txt | awk '{print $2}' | tr '\n' ' ')"
done
