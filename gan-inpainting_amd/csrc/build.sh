#!/bin/bash
# Build libganinpaint.so for gfx950 (cross-compiles without a GPU). Usage: ./build.sh [extra hipcc flags]
set -e
cd "$(dirname "$0")"
OUT=${GI_OUT:-../libganinpaint.so}      # GI_OUT / GI_BUILD_DIR: a second build beside the shipped one (tools: ablation A/B, -DGI_ABLATION)
BUILD=${GI_BUILD_DIR:-build}
SRCS="api.hip igemm.hip igemm3.hip igemm5.hip igemm7.hip igemm8.hip wgrad.hip wgrad2.hip c1.hip elementwise.hip ssim.hip evalmetrics.hip auxloss.hip vgg.hip resize.hip comm.hip net.hip"
OBJS=""
mkdir -p $BUILD
pids=""
for s in $SRCS; do
  o=$BUILD/${s%.hip}.o
  OBJS="$OBJS $o"
  if [ ! -f $o ] || [ $s -nt $o ] || [ common.h -nt $o ] || [ halo_args.h -nt $o ] || [ stat_acc.h -nt $o ] || [ ../../include/ganinpaint.h -nt $o ]; then
    ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable "$@" -c $s -o $o ) &
    pids="$pids $!"
  fi
done
for p in $pids; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $OBJS -ldl
echo "built $(readlink -f $OUT)"
