#!/bin/bash
# K8 variants on the Zundel100 shape (5050 x 82621, 32 geometries): tools/scan_k8_zundel100.sh  (on the GPU box)
R=$GRAFT_REPO_ROOT
run() {
  env "$@" python3 $R/bench.py --workload Zundel100 --streams 1 --steps 6 --warmup 2 --repeats 1 --no-md-regime --no-cpu-baseline 2>/dev/null |
    python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', 'k8 %.1f us  k5 %.1f us  value %.0f' % (1e3*d['kernels']['k8_cols_ms'], 1e3*d['kernels']['k5_rows_ms'], d['value']))"
}
run X=0
run EVC_COLS_RS_MAX=0
run EVC_COLS_RS_MAX=0 EVC_COLS_SHAPE2=214
run EVC_COLS_RS_MAX=0 EVC_COLS_SHAPE2=223
run EVC_COLS_RS_MAX=0 EVC_COLS_SHAPE2=233
run EVC_COLS_RS_MAX=0 EVC_COLS_SHAPE2=342
run EVC_COLS_RS_SHAPE2=1043
run EVC_COLS_RS_SHAPE2=1082
run EVC_COLS_RS_SHAPE2=1023
