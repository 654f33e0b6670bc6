#!/bin/bash
set -u
# Hardware counters per dispatch for one WGAN batch (tools/step_chain.py): separate --pmc passes (FETCH_SIZE / WRITE_SIZE / SQ groups),
# merged into gpurun_out/<tag>_pmc.txt: one line per kernel of the last repetition with its counters.
# usage (GPU box): tools/pmc_chain.sh <tag> [gen|critic]
set -o pipefail
TAG=${1:-p}; KIND=${2:-critic}
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the tools run on the GPU box through gpurun)}
OUT=$R/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- python3 $R/tools/step_chain.py 2 $KIND > $OUT/g$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/g$i.log; }
done
python3 $R/tools/pmc_table.py $OUT > $R/gpurun_out/${TAG}_pmc.txt
tail -5 $R/gpurun_out/${TAG}_pmc.txt
