#!/bin/bash
set -u
# Hardware counters of the dominant kernel (bench.py --kernel-only), one rocprofv3 --pmc pass per group.
# usage (on the GPU box): bash tools/pmc_dominant.sh <outdir-under-gpurun_out>
set -o pipefail
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the tools run on the GPU box through gpurun)}
OUT=$R/gpurun_out/${1:-pmc_dom}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_INT32 SQ_INSTS_BRANCH" \
           "SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- python3 $R/bench.py --kernel-only --kernel-iters 5 > $OUT/g$i.log 2>&1 || { echo "group $i failed"; tail -3 $OUT/g$i.log; }
done
echo done
