#!/bin/bash
# K5 shapes on the Zundel100 shape (5050 x 82621, 32 geometries): tools/scan_k5_zundel100.sh  (on the GPU box)
R=$GRAFT_REPO_ROOT
run() {
  env "$@" python3 $R/bench.py --workload Zundel100 --streams 1 --steps 6 --warmup 2 --repeats 1 --no-md-regime --no-cpu-baseline 2>/dev/null |
    python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', 'k5 %.1f us  k8 %.1f us  value %.0f' % (1e3*d['kernels']['k5_rows_ms'], 1e3*d['kernels']['k8_cols_ms'], d['value']))"
}
run X=0
run EVC_ROWS_SHAPE2_NARROW=321
run EVC_ROWS_SHAPE2_NARROW=330
run EVC_ROWS_SHAPE2_NARROW=240
run EVC_ROWS_SHAPE2_NARROW=321 EVC_MFMA_TILES=2
run EVC_MFMA_TILES=5
run EVC_MFMA_TILES=6
