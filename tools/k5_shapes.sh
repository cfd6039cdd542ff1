#!/bin/bash
# K5 shapes of the LDS-staged kernel on one GPU box: rocprofv3 per-kernel averages.  tools/k5_shapes.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
one() {  # tag, env...
  tag=$1; shift
  rm -rf $R/gpurun_out/kst
  ( export "$@"; rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kst -- python3 $R/bench.py --no-cpu-baseline --no-md-regime --steps 20 --streams 1 --repeats 1 $BENCH_ARGS > $R/gpurun_out/kst.log 2>&1 )
  python3 - <<PY
import csv,glob,json
f=glob.glob("$R/gpurun_out/kst/*/*kernel_stats.csv")[0]
out=[]
for r in csv.DictReader(open(f)):
    n=r["Name"]
    for k in ("gemv_rows","rows_reduce","gemv_cols","subspace"):
        if k in n: out.append("%s %.1f" % (k, float(r["AverageNs"])/1e3))
v=[json.loads(l[l.index("{"):])["value"] for l in open("$R/gpurun_out/kst.log") if '"metric"' in l]
print("$tag:", "; ".join(out), "value", v)
PY
  rm -rf $R/gpurun_out/kst
}
one K8-LDS-8w EVC_COLS_LDS=1
one K8-LDS-4w EVC_COLS_LDS=1 EVC_COLS_LDS_NW=4
one K8-OLD EVC_COLS_LDS=0
one K8-LDS-8w EVC_COLS_LDS=1
one K8-LDS-4w EVC_COLS_LDS=1 EVC_COLS_LDS_NW=4
one K8-OLD EVC_COLS_LDS=0
