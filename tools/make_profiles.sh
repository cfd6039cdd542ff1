#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel stats + separate PMC passes for the default bench
# configuration (compressed layout sym8, 32 geometries per pass, 3 streams), for one stream, for the reference's
# pack2 layout and for the MD regime.  Outputs land in gpurun_out/profiles_raw/ and are condensed into profiles/
# by tools/condense_profiles.py in the build container.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/profiles_raw
rm -rf $O
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {  # tag, rocprof args..., -- bench args
  tag=$1; shift
  rocprofv3 "$@" > $O/$tag.log 2>&1
}
BENCH="python3 $R/bench.py --no-cpu-baseline --no-md-regime --steps 20 --warmup 4 --repeats 1"
run stats_default      --kernel-trace --stats --output-format csv -d $O/stats_default -- $BENCH
run stats_sym8_b32s1   --kernel-trace --stats --output-format csv -d $O/stats_sym8_b32s1 -- $BENCH --streams 1
run stats_pack2_b32s1  --kernel-trace --stats --output-format csv -d $O/stats_pack2_b32s1 -- $BENCH --layout pack2 --streams 1
run stats_sym8_md      --kernel-trace --stats --output-format csv -d $O/stats_sym8_md -- $BENCH --batch 1 --streams 1 --steps 60
run stats_pack2_md     --kernel-trace --stats --output-format csv -d $O/stats_pack2_md -- $BENCH --layout pack2 --batch 1 --streams 1 --steps 60
run stats_zundel100_b32s1 --kernel-trace --stats --output-format csv -d $O/stats_zundel100_b32s1 -- $BENCH --workload Zundel100 --streams 1 --steps 6 --warmup 2
run stats_zundel100_md --kernel-trace --stats --output-format csv -d $O/stats_zundel100_md -- $BENCH --workload Zundel100 --batch 1 --streams 1 --steps 20 --warmup 2
run stats_h10_md        --kernel-trace --stats --output-format csv -d $O/stats_h10_md -- $BENCH --workload H10 --batch 1 --streams 1 --steps 200
run stats_h2ovtz_b4s1  --kernel-trace --stats --output-format csv -d $O/stats_h2ovtz_b4s1 -- $BENCH --workload H2Ovtz --batch 4 --streams 1 --geoms 8 --steps 6 --warmup 2
run stats_h2ovtz_b32s1 --kernel-trace --stats --output-format csv -d $O/stats_h2ovtz_b32s1 -- $BENCH --workload H2Ovtz --batch 32 --streams 1 --geoms 32 --steps 4 --warmup 1
for lay in sym8 pack2; do
run pmc_fetch_${lay}_b32 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_${lay}_b32 -- $BENCH --layout $lay --streams 1 --steps 8
run pmc_write_${lay}_b32 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_${lay}_b32 -- $BENCH --layout $lay --streams 1 --steps 8
run pmc_fetch_${lay}_md  --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_${lay}_md -- $BENCH --layout $lay --batch 1 --streams 1 --steps 20
run pmc_write_${lay}_md  --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_${lay}_md -- $BENCH --layout $lay --batch 1 --streams 1 --steps 20
done
run pmc_sq_sym8_b32    --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/pmc_sq_sym8_b32 -- $BENCH --streams 1 --steps 8
run pmc_lds_sym8_b32   --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc_lds_sym8_b32 -- $BENCH --streams 1 --steps 8
H2O="--workload H2Ovtz --batch 4 --streams 1 --geoms 8 --steps 4 --warmup 1"
run pmc_sq_h2ovtz_b4   --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/pmc_sq_h2ovtz_b4 -- $BENCH $H2O
run pmc_lds_h2ovtz_b4  --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc_lds_h2ovtz_b4 -- $BENCH $H2O
grep -h "\"metric\"" $O/stats_*.log | cut -c1-160; du -sh $O
python3 $R/tools/condense_profiles.py ${ROUND:-r01} $R/gpurun_out/profiles_out > $R/gpurun_out/profiles_out.log 2>&1
rm -rf $O
ls $R/gpurun_out/profiles_out
