#!/bin/bash
# Same-box A/B of two builds of the library: forward kernel chain under rocprofv3 and the headline line, base first then new, twice.
# usage: tools/r4_ab.sh <tag> <base.so> <new.so>     (run on the GPU box through gpurun)
set -u -o pipefail
TAG=$1; BASE=$2; NEW=$3
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
OUT=$R/gpurun_out/ab_$TAG
mkdir -p $OUT
$R/tools/fwd_ab.sh ${TAG} GI_LIB_PATH=$BASE GI_LIB_PATH=$NEW 2>&1 | tee $OUT/fwd.txt
for round in 1 2; do
  for lib in $BASE $NEW; do
    echo "== bench $(basename $lib) round $round"
    GI_LIB_PATH=$lib python3 $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>$OUT/bench_err.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('generator_fwd_ms'), d['roofline']['frac'], d.get('roofline_short_k',{}).get('frac'))"
  done
done 2>&1 | tee $OUT/bench.txt
