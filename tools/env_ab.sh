#!/bin/bash
# Same-box A/B of one library option through its environment variable: headline line and the single-stream batch times, off / on, twice.
# usage: tools/env_ab.sh <tag> <OPTION> <value A> <value B>     (run on the GPU box through gpurun)
set -u -o pipefail
TAG=$1; OPT=$2; A=$3; B=$4
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
OUT=$R/gpurun_out/envab_$TAG
mkdir -p $OUT
for round in 1 2; do
  for v in $A $B; do
    echo "== $OPT=$v round $round"
    env $OPT=$v python3 $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>$OUT/bench_err.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('headline', d['value'], d['ms_per_step'], d.get('generator_fwd_ms'), d['roofline']['frac'])"
    env $OPT=$v python3 $R/tools/step_chain.py 6 gen 2>>$OUT/bench_err.log
    env $OPT=$v python3 $R/tools/step_chain.py 6 critic 2>>$OUT/bench_err.log
  done
done 2>&1 | tee $OUT/result.txt
