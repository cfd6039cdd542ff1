#!/bin/bash
# usage: prof1.sh <tag>  (env passed through); prints evc kernel averages of one H2Ovtz profile
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$1 -- python3 $GRAFT_REPO_ROOT/bench.py --workload H2Ovtz --batch 4 --streams 1 --geoms 8 --steps 6 --warmup 2 --no-cpu-baseline --no-md-regime --repeats 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(ls gpurun_out/prof_$1/*/*kernel_stats.csv | head -1); grep "evc::" $f | sed "s/(.*)\"/\"/" | cut -d, -f1-4 | cut -c1-110 | head -${2:-4}; rm -rf gpurun_out/prof_$1
