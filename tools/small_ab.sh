#!/bin/bash
set -u
# The small-M layers (tools/time_small.py) on two builds of the library, alternating. usage: tools/small_ab.sh <a.so> <b.so>
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
for round in 1 2 3; do
for lib in $1 $2; do
  echo -n "$(basename $lib) : "; GI_LIB_PATH=$lib python3 $R/tools/time_small.py 50 2>&1 | tail -1
done
done
