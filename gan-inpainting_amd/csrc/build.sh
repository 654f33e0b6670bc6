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
    # -Rpass-analysis=kernel-resource-usage: registers / scratch / occupancy per kernel into $BUILD/<file>.res (summary below)
    ( rc=0
      hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable -Rpass-analysis=kernel-resource-usage "$@" -c $s -o $o 2> $BUILD/${s%.hip}.log || rc=$?
      grep "kernel-resource-usage" $BUILD/${s%.hip}.log | sed -n 's/.*remark: *\(.*\) \[-Rpass.*/\1/p' > $BUILD/${s%.hip}.res || true
      grep -v "kernel-resource-usage" $BUILD/${s%.hip}.log | grep -v "^ *[0-9]* | \|^ *| *^" >&2 || true
      exit $rc ) &
    pids="$pids $!"
  fi
done
for p in $pids; do wait $p; done
# kernels with scratch (register spills or private arrays): the GEMM kernels must have none (tests/test_build_resources.py)
python3 - $BUILD <<'PY'
import glob, os, sys
b = sys.argv[1]
rows = []
for f in sorted(glob.glob(os.path.join(b, "*.res"))):
    name = None
    for line in open(f):
        line = line.strip()
        if line.startswith("Function Name:"):
            name = line.split(":", 1)[1].strip()
            rows.append([os.path.basename(f)[:-4], name, {}])
        elif name and ":" in line:
            k, v = line.rsplit(":", 1)
            rows[-1][2][k.strip()] = v.strip()
with open(os.path.join(b, "resources.txt"), "w") as out:
    for f, n, d in rows:
        out.write(f"{f}\t{n}\tvgpr={d.get('VGPRs')}\tagpr={d.get('AGPRs')}\tscratch={d.get('ScratchSize [bytes/lane]')}\tocc={d.get('Occupancy [waves/SIMD]')}\n")
bad = [(f, n, d.get('ScratchSize [bytes/lane]')) for f, n, d in rows if d.get('ScratchSize [bytes/lane]', '0') != '0']
for f, n, sc in bad:
    print(f"note: {f}: {n} uses {sc} bytes of scratch per lane", file=sys.stderr)
PY
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $OBJS -ldl
echo "built $(readlink -f $OUT)"
