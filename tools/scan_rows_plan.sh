#!/bin/bash
# Scan the span plan of the batched rows kernel (run on the GPU box): prints K5/K8 times per setting.
mkdir -p gpurun_out
for t in 0 4; do for tg in 2048 3456 4096 5120 6144 8192 16384; do
EVC_MFMA_TILES=$t EVC_ROWS_TARGET_WGS=$tg EVC_ROWS_MIN_CPS=1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-md-regime --streams 1 --steps 40 > gpurun_out/scan.json 2>gpurun_out/err.log || exit 1
python - <<PY
import json
d=json.load(open("gpurun_out/scan.json"))
print("tiles",$t,"target",$tg, round(d["value"]), "k5", round(d["kernels"]["k5_rows_ms"]*1e3), "k8", round(d["kernels"]["k8_cols_ms"]*1e3), flush=True)
PY
done; done
