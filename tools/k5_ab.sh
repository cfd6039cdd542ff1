#!/bin/bash
# A/B of K5 variants on one GPU box: rocprofv3 per-kernel averages, alternating order.  tools/k5_ab.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
one() {  # tag, env...
  tag=$1; shift
  rm -rf $R/gpurun_out/kst
  ( export "$@"; rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kst -- python3 $R/bench.py --no-cpu-baseline --no-md-regime --steps 20 --streams 1 --repeats 1 > $R/gpurun_out/kst.log 2>&1 )
  python3 - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/kst/*/*kernel_stats.csv")[0]
out=[]
for r in csv.DictReader(open(f)):
    n=r["Name"]
    for k in ("gemv_rows","rows_reduce","gemv_cols","pt_pipe_kernel<32, 0>","ip1_dh"):
        if k in n: out.append("%s %.1f" % (k, float(r["AverageNs"])/1e3))
print("$tag:", "; ".join(out))
PY
  rm -rf $R/gpurun_out/kst
}
one LDS EVC_ROWS_LDS=1
one OLD EVC_ROWS_LDS=0
one LDS EVC_ROWS_LDS=1
one OLD EVC_ROWS_LDS=0
if [ -n "$K5_EXTRA" ]; then
  ( cd $R && EVC_EXTRA_DEFS="$K5_EXTRA" python3 evcont_amd/build.py --force > gpurun_out/build_extra.log 2>&1 ) || exit 1
  one "LDS $K5_EXTRA" EVC_ROWS_LDS=1
  one "LDS $K5_EXTRA" EVC_ROWS_LDS=1
fi
