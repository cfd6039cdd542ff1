#!/bin/bash
# rocprofv3 per-kernel averages of one bench configuration (run on the GPU box): tools/kstats.sh [bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/kst
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kst -- python3 $R/bench.py --no-cpu-baseline --no-md-regime --steps 20 "$@" > $R/gpurun_out/kst.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/kst/*/*kernel_stats.csv")[0]
tot=0
for r in csv.DictReader(open(f)):
    if "evc::" in r["Name"]:
        print(r["Name"][:58].ljust(58), r["Calls"].rjust(4), round(float(r["AverageNs"])/1e3,1))
PY
grep -h '"metric"' $R/gpurun_out/kst.log | cut -c1-140
rm -rf $R/gpurun_out/kst
