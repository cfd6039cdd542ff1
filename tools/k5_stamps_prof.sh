#!/bin/bash
# stamps build + tools/micro/k5_stamps.py for the shapes given as arguments (EVC_ROWS_LDS_NT values; default: 14)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
# (the stamps build is a library of its own, selected by EVCONT_HIP_LIB: the product library stays untouched)
( cd $R && EVC_DEBUG_STAMPS=1 python3 evcont_amd/build.py > gpurun_out/build_stamps.log 2>&1 ) || exit 1
export EVCONT_HIP_LIB=$R/evcont_amd/libevcont_hip_stamps.so
for nt in ${@:-14}; do
  echo "== NT=$nt"
  EVC_ROWS_LDS_NT=$nt python3 $R/tools/micro/k5_stamps.py 2>&1 | tail -6
done
