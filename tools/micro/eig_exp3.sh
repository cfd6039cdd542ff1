#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
cd $R
for nt in "30 20" "10 5"; do
  echo "== n T = $nt"; python tools/micro/loewdin_time.py $nt 2>&1 | grep " us"
  echo "== stamps n T = $nt"
  EVCONT_HIP_LIB=$R/evcont_amd/libevcont_hip_stamps.so python tools/micro/loewdin_time.py $nt 2>&1 | grep "eigh:\|kernel phases\| us$" | cut -c1-420
done
python -m pytest tests/test_gpu_eigensolvers.py tests/test_gpu_warm_start.py tests/test_gpu_api.py -x -q -m gpu 2>&1 | tail -3
