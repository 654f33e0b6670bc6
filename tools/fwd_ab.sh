#!/bin/bash
set -u
# Kernel chain of one generator forward (256x256, bs=32, fp16, train mode) under rocprofv3, for each setting of an
# environment switch. usage: tools/fwd_ab.sh <tag> [VAR=a VAR=b ...]   (run on the GPU box through gpurun)
set -o pipefail
TAG=${1:-ab}; shift
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the tools run on the GPU box through gpurun)}
cd /tmp && export TMPDIR=/tmp
for setting in "${@:-X=0}"; do
  OUT=$R/gpurun_out/fwd_${TAG}_${setting//=/_}
  rm -rf $OUT; mkdir -p $OUT
  export "$setting"
  rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/tools/gfwd_only.py 20 > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
  CSV=$(find $OUT -name "*kernel_trace.csv" | head -1)
  python3 $R/tools/chain_table.py $CSV c1_gather > $OUT/chain.txt
  tail -1 $OUT/chain.txt; grep "G forward" $OUT/run.log
  unset "${setting%%=*}"
done
