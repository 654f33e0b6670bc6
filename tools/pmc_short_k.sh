#!/bin/bash
set -u
# HBM traffic and rocprof duration of the short-K kernel (critic conv2, bench.py --kernel-only --short-k): one --pmc pass per counter
# + one --kernel-trace --stats pass -> gpurun_out/pmc_short_k/; summarised into profiles/short_k_kernel_traffic.json by
# tools/summarize_short_k.py. usage (GPU box): bash tools/pmc_short_k.sh
set -o pipefail
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the tools run on the GPU box through gpurun)}
OUT=$R/gpurun_out/pmc_short_k
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py --kernel-only --short-k --kernel-iters 10 > $OUT/fetch.log 2>&1 || tail -3 $OUT/fetch.log
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py --kernel-only --short-k --kernel-iters 10 > $OUT/write.log 2>&1 || tail -3 $OUT/write.log
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --kernel-only --short-k --kernel-iters 50 > $OUT/stats.log 2>&1 || tail -3 $OUT/stats.log
find $OUT -name "*.csv" | wc -l
