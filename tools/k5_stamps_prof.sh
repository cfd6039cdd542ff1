#!/bin/bash
# stamps build + rocprofv3 around tools/micro/k5_stamps.py: the per-workgroup timeline and the kernel's duration as
# the profiler sees it, from the same process
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( cd $R && EVC_DEBUG_STAMPS=1 python3 evcont_amd/build.py --force > gpurun_out/build_stamps.log 2>&1 ) || exit 1
rm -rf $R/gpurun_out/kst
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kst -- python3 $R/tools/micro/k5_stamps.py > $R/gpurun_out/k5_stamps_prof.log 2>&1
tail -7 $R/gpurun_out/k5_stamps_prof.log
python3 - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/kst/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "gemv_rows" in r["Name"] or "rows_reduce" in r["Name"]: print(r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3)
f=glob.glob("$R/gpurun_out/kst/*/*kernel_trace.csv")[0]
rows=[r for r in csv.DictReader(open(f)) if "gemv_rows" in r["Kernel_Name"]]
r=rows[-1]
print({k: r[k] for k in r if k in ("Start_Timestamp","End_Timestamp","LDS_Block_Size","Scratch_Size","VGPR_Count","Accum_VGPR_Count","SGPR_Count","Workgroup_Size","Grid_Size")})
print("last launch duration us", (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
PY
rm -rf $R/gpurun_out/kst
