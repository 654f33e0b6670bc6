#!/bin/bash
set -u
# Kernel chain of one WGAN batch WITH a generator update on one stream (256x256, bs=32, fp16) under rocprofv3.
# usage: tools/step_ab.sh <tag> [gen|critic]   (through gpurun) -> gpurun_out/step_<tag>/chain.txt + per-kernel-class summary
set -o pipefail
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the tools run on the GPU box through gpurun)}
OUT=$R/gpurun_out/step_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/tools/step_chain.py 4 ${2:-gen} > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
CSV=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 $R/tools/chain_table.py $CSV mask_apply > $OUT/chain.txt
tail -1 $OUT/chain.txt; grep "^batch " $OUT/run.log
python3 - $OUT/chain.txt <<'PY'
import re,sys,collections
agg=collections.defaultdict(lambda:[0,0.0])
for l in open(sys.argv[1]):
    m=re.match(r"\s*([\d.]+)\s+([\d.]+) us\s+wg=\s*\d+x\s*\d+\s+(.*)",l)
    if not m: continue
    k=m.group(3).split()[0].split("<")[0]
    k=re.sub(r"^_ZN12_GLOBAL__N_1\d+","",k)
    agg[k][0]+=1; agg[k][1]+=float(m.group(2))
tot=sum(v[1] for v in agg.values())
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1])[:25]:
    print(f"{v[1]:8.1f} us {v[1]/tot*100:5.1f}%  x{v[0]:3d}  {k}")
print(f"{tot:8.1f} us total")
PY
