#!/bin/bash
# K5 / K8 kernel averages (rocprofv3), LDS-staged against the fragment-shaped kernels, one stream: tools/k58_ab.sh [bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
one() {  # tag, env...
  tag=$1; shift
  rm -rf $R/gpurun_out/kst
  ( export "$@"; rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kst -- python3 $R/bench.py --no-cpu-baseline --no-md-regime --steps 10 --streams 1 --repeats 1 $BENCH_ARGS > $R/gpurun_out/kst.log 2>&1 )
  python3 - <<PY
import csv,glob,json
f=glob.glob("$R/gpurun_out/kst/*/*kernel_stats.csv")[0]
out=[]
for r in csv.DictReader(open(f)):
    n=r["Name"]
    for k in ("gemv_rows","rows_reduce","gemv_cols","ip1_"):
        if k in n: out.append("%s %.1f" % (n[:44], float(r["AverageNs"])/1e3))
v=[json.loads(l[l.index("{"):])["value"] for l in open("$R/gpurun_out/kst.log") if '"metric"' in l]
print("$tag:", "; ".join(out), "value", v)
PY
  rm -rf $R/gpurun_out/kst
}
if [ -n "$SLABAB" ]; then one SLAB EVC_COLS_LDS_SLAB=1; one NOSLAB EVC_COLS_LDS_SLAB=0; one SLAB EVC_COLS_LDS_SLAB=1; one NOSLAB EVC_COLS_LDS_SLAB=0; exit 0; fi
if [ -n "$NTS" ]; then for nt in $NTS; do one NT=$nt EVC_ROWS_LDS_NT=$nt; done; exit 0; fi
one NEW EVC_ROWS_LDS=1 EVC_COLS_LDS=1 EVC_IP1_LDS=1
one OLD EVC_ROWS_LDS=0 EVC_COLS_LDS=0 EVC_IP1_LDS=0
one NEW EVC_ROWS_LDS=1 EVC_COLS_LDS=1 EVC_IP1_LDS=1
one OLD-IP1 EVC_ROWS_LDS=1 EVC_COLS_LDS=1 EVC_IP1_LDS=0
