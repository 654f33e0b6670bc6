#!/bin/bash
set -u
# Kernel chains of one critic-only batch and one batch with a generator update (single stream, tools/step_chain.py) under
# rocprofv3, for each setting of an environment switch: the sum of the kernel durations is a steadier A/B number than a
# wall-clock benchmark line. usage: tools/step_ab.sh <tag> [VAR=a VAR=b ...]   (run on the GPU box through gpurun)
set -o pipefail
TAG=${1:-ab}; shift
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the tools run on the GPU box through gpurun)}
cd /tmp && export TMPDIR=/tmp
export GI_WGRAD_STREAM=0     # one stream: durations add up
i=0
for setting in "${@:-X=0}"; do
  i=$((i + 1))
  export "$setting"
  for mode in critic gen; do
    OUT=$R/gpurun_out/step_${TAG}_${i}_${mode}
    rm -rf $OUT; mkdir -p $OUT
    rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/tools/step_chain.py 6 $mode > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
    CSV=$(find $OUT -name "*kernel_trace.csv" | head -1)
    python3 $R/tools/chain_table.py $CSV mask_apply > $OUT/chain.txt
    echo "$setting $mode: $(tail -1 $OUT/chain.txt)"
    rm -f $CSV
  done
  unset "${setting%%=*}"
done
