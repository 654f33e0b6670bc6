#!/bin/bash
set -u
# Collect the rocprofv3 evidence for profiles/ (run on the GPU box through gpurun).
# usage: tools/profile_round.sh <tag>
set -o pipefail
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the tools run on the GPU box through gpurun)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kernel_only -- python3 $R/bench.py --kernel-only --kernel-iters 50 > $OUT/kernel_only.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -- python3 $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline > $OUT/bench.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --kernel-only --kernel-iters 10 > $OUT/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --kernel-only --kernel-iters 10 > $OUT/pmc_write.log 2>&1 || exit 1
find $OUT -name "*.csv" | head -30
