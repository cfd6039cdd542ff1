#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel stats + separate PMC passes for the default bench
# configuration (32 geometries per pass), for 16 geometries per pass and for the MD regime.  Outputs land in
# gpurun_out/profiles_raw/ and are condensed into profiles/ by tools/condense_profiles.py in the build container.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/profiles_raw
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {  # tag, rocprof args..., -- bench args
  tag=$1; shift
  rocprofv3 "$@" > $O/$tag.log 2>&1
}
BENCH="python3 $R/bench.py --no-cpu-baseline --no-md-regime --steps 20 --warmup 4"
run stats_default --kernel-trace --stats --output-format csv -d $O/stats_default -- $BENCH
run stats_b32s1   --kernel-trace --stats --output-format csv -d $O/stats_b32s1 -- $BENCH --streams 1
run stats_b16s1   --kernel-trace --stats --output-format csv -d $O/stats_b16s1 -- $BENCH --batch 16 --streams 1
run stats_md      --kernel-trace --stats --output-format csv -d $O/stats_md -- $BENCH --batch 1 --streams 1 --steps 60
for b in 32 16; do
run pmc_fetch_b$b --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_b$b -- $BENCH --batch $b --streams 1 --steps 8
run pmc_write_b$b --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_b$b -- $BENCH --batch $b --streams 1 --steps 8
done
run pmc_fetch_md  --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_md -- $BENCH --batch 1 --streams 1 --steps 20
run pmc_write_md  --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_md -- $BENCH --batch 1 --streams 1 --steps 20
run pmc_sq_b32    --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/pmc_sq_b32 -- $BENCH --streams 1 --steps 8
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; grep -h "\"metric\"" $O/stats_*.log | cut -c1-160; du -sh $O
ls $O
