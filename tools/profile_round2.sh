#!/bin/bash
set -u
# rocprofv3 evidence for profiles/ (run on the GPU box through gpurun; every rocprofv3 run has the program directly after `--`).
# usage: tools/profile_round2.sh <tag>     -> gpurun_out/prof_<tag>/..., summarised by tools/summarize_round2.py <tag>
set -o pipefail
TAG=${1:-r02}
PART=${2:-all}      # all | 1 (A - E: dominant kernel, benchmark process, counters, chains) | 2 (F - I: secondary workloads, headline line, two ranks, layer table)
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the tools run on the GPU box through gpurun)}
OUT=$R/gpurun_out/prof_$TAG
if [ "$PART" != "2" ]; then rm -rf $OUT; fi
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; echo "== $name"; timeout -k 10 600 "$@" > $OUT/$name.log 2>&1 || { echo "$name failed"; tail -5 $OUT/$name.log; }; }
if [ "$PART" != "2" ]; then
# A. the dominant kernel alone; B. the whole benchmark process; C. its HBM traffic (one --pmc pass per counter)
run kernel_only rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kernel_only -- python3 $R/bench.py --kernel-only --kernel-iters 50
run bench rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -- python3 $R/bench.py --steps 20 --warmup 5 --preheat 50 --no-cpu-baseline
run pmc_fetch rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --kernel-only --kernel-iters 10
run pmc_write rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --kernel-only --kernel-iters 10
# D. counters of the dominant kernel
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_INT32 SQ_INSTS_BRANCH" \
           "SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  run pmc_g$i rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_g$i -- python3 $R/bench.py --kernel-only --kernel-iters 5
done
# E. one generator forward; one critic-only batch; one batch with a generator update (kernel chains)
run gfwd rocprofv3 --kernel-trace --output-format csv -d $OUT/gfwd -- python3 $R/tools/gfwd_only.py 20
export GI_WGRAD_STREAM=0   # the chains on ONE stream (the weight gradients in line): durations add up; the benchmark itself uses two
run step_critic rocprofv3 --kernel-trace --output-format csv -d $OUT/step_critic -- python3 $R/tools/step_chain.py 4 critic
run step_gen rocprofv3 --kernel-trace --output-format csv -d $OUT/step_gen -- python3 $R/tools/step_chain.py 4 gen
unset GI_WGRAD_STREAM
fi
if [ "$PART" != "1" ]; then
# F. secondary workloads: the plain line (no profiler) and the kernel statistics
for wl in wgan_gp_128 dual_d_256 config5_512 vgg_512; do
  run line_$wl python3 $R/bench.py --workload $wl --steps 20 --warmup 5 --preheat 50      # (with its reduced cpu_baseline: 1 + 2 + 1 batches)
  run stats_$wl rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$wl -- python3 $R/bench.py --workload $wl --steps 10 --warmup 5 --preheat 10 --no-cpu-baseline
done
# G. the headline line itself, unprofiled (with the CPU baseline); H. the multi-rank launch path rehearsed with two gloo ranks on
# this one card (RCCL needs one GPU per rank: the 8-GPU run is the driver's)
run line_headline python3 $R/bench.py --steps 20 --warmup 5
GI_DIST_BACKEND=gloo run line_two_ranks_gloo python3 $R/bench.py --gpus 2 --steps 10 --warmup 2 --preheat 20 --no-cpu-baseline
# I. per-layer / per-instantiation table of the GEMM-shaped launches (HIP events, no profiler)
run layer_table python3 $R/tools/layer_table.py $TAG 30
fi
find $OUT -name "*.csv" | wc -l
echo done
