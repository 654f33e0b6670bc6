#!/bin/bash
set -u
# One batch with a generator update on one stream under rocprofv3 --kernel-trace -> gpurun_out/<tag>_chain.txt (every kernel in order)
# usage (on the GPU box, through gpurun): tools/quick_chain.sh <tag> [gen|critic]
set -o pipefail
TAG=${1:-q}; KIND=${2:-gen}
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the tools run on the GPU box through gpurun)}
OUT=$R/gpurun_out/qc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/tools/step_chain.py 4 $KIND > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
CSV=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 $R/tools/chain_table.py $CSV mask_apply_kernel > $R/gpurun_out/${TAG}_chain.txt
tail -3 $OUT/run.log; tail -1 $R/gpurun_out/${TAG}_chain.txt
