#!/bin/bash
# eigen-kernel experiments on the GPU box: parity of the small solvers, then their timing (product library) and phase
# stamps (stamps library, built by `EVC_DEBUG_STAMPS=1 python evcont_amd/build.py` and selected by EVCONT_HIP_LIB)
R=${GRAFT_REPO_ROOT:-.}
cd $R
python -m pytest tests/test_gpu_eigensolvers.py tests/test_gpu_warm_start.py -x -q -m gpu 2>&1 | tail -4
for nt in "30 20" "10 5" "13 10" "28 30"; do
  echo "== n T = $nt"
  python tools/micro/loewdin_time.py $nt 2>&1 | grep " us"
done
if [ -f evcont_amd/libevcont_hip_stamps.so ]; then
  for nt in "30 20" "10 5"; do
    echo "== stamps n T = $nt"
    EVCONT_HIP_LIB=$R/evcont_amd/libevcont_hip_stamps.so python tools/micro/loewdin_time.py $nt 2>&1 | tail -8
  done
fi
