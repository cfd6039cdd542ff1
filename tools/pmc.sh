#!/bin/bash
# usage: tools/pmc.sh <tag> "<counters>" [bench args...]   -- one rocprofv3 --pmc pass (kernel-trace only)
TAG=$1; CNT=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 ${PMC_TIMEOUT:-300} rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d $R/gpurun_out/pmc_$TAG -- python3 $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/pmc_$TAG.log 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob('$R/gpurun_out/pmc_$TAG/*/*counter_collection.csv')
if not f:
    print("no counter file"); print(open('$R/gpurun_out/pmc_$TAG.log').read()[-2000:]); raise SystemExit
rows=list(csv.DictReader(open(f[0])))
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k=r['Kernel_Name'][:48]
    if 'at::' in k or 'rocclr' in k or 'Cijk' in k: continue
    agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,d in agg.items():
    print(k, {c: round(sum(v)/len(v),1) for c,v in d.items()}, 'n=',len(next(iter(d.values()))))
PY
rm -rf $R/gpurun_out/pmc_$TAG
