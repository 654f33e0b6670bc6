#!/bin/bash
set -u
# Hardware counters of the dominant kernel (bench.py --kernel-only: generator u3), separate rocprofv3 --pmc passes, for the default
# kernel choice (igemm8) and for igemm6 (GI_IGEMM8=0) -> gpurun_out/<tag>_pmc_kernel.txt   usage (GPU box): tools/pmc_kernel.sh <tag>
set -o pipefail
TAG=${1:-k}
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the tools run on the GPU box through gpurun)}
cd /tmp && export TMPDIR=/tmp
for variant in igemm8 igemm6; do
  OUT=$R/gpurun_out/pmck_${TAG}_$variant
  rm -rf $OUT; mkdir -p $OUT
  if [ $variant = igemm6 ]; then export GI_IGEMM8=0; else unset GI_IGEMM8; fi
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" \
             "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
             "SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout -k 10 120 rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- python3 $R/bench.py --kernel-only --kernel-iters 10 > $OUT/g$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/g$i.log; }
  done
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/bench.py --kernel-only --kernel-iters 50 > $OUT/kt.log 2>&1
  echo "== $variant" >> $R/gpurun_out/${TAG}_pmc_kernel.txt
  python3 $R/tools/pmc_table.py $OUT | grep -E "^kernel|igemm" >> $R/gpurun_out/${TAG}_pmc_kernel.txt
  grep -h "igemm" $(find $OUT/kt -name "*kernel_stats.csv") | head -3 >> $R/gpurun_out/${TAG}_pmc_kernel.txt
done
unset GI_IGEMM8
cat $R/gpurun_out/${TAG}_pmc_kernel.txt
