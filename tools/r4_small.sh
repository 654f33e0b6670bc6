#!/bin/bash
set -u
# igemm7 variants on the generator's small-M layers (ablation build): ring depth x split-K tail prefetch. usage: tools/r4_small.sh
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
export GI_LIB_PATH=$R/gan-inpainting_amd/libganinpaint_abl.so
for round in 1 2; do
for ns in 4 6; do for pf in 1 2 4; do
  echo -n "NSTG=$ns PF=$pf : "; GI_IGEMM7_NSTG=$ns GI_IGEMM7_PF=$pf python3 $R/tools/time_small.py 50 2>&1 | tail -1
done; done
done
for ms in 4 16; do echo -n "MAXSPLIT=$ms NSTG=4 PF=4 : "; GI_IGEMM7_MAXSPLIT=$ms GI_IGEMM7_NSTG=4 GI_IGEMM7_PF=4 python3 $R/tools/time_small.py 50 2>&1 | tail -1; done
