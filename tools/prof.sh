#!/bin/bash
# usage: tools_prof.sh <tag> [bench args...]  -- runs GPU tests (fast subset optional), then rocprofv3 kernel stats of bench.py
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/prof_$TAG.log 2>&1
grep metric $R/gpurun_out/prof_$TAG.log | cut -c1-330
python3 - <<PY
import csv,glob
f=glob.glob('$R/gpurun_out/prof_$TAG/*/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:24]:
    if 'at::native' in r['Name'] or 'rocclr' in r['Name'] or 'Cijk' in r['Name']: continue
    print(f"{r['Name'][:58]:58s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:8.1f} pct={r['Percentage']}")
PY
