#!/bin/bash
# like scan_env.sh but with 32 geometries per step (BATCH, LAYOUT, STREAMS from the environment)
mkdir -p gpurun_out
var=$1; shift
for v in "$@"; do
env $var=$v timeout -k 10 200 python bench.py --no-cpu-baseline --no-md-regime --batch ${BATCH:-32} --layout ${LAYOUT:-pack2} --streams ${STREAMS:-1} --steps 30 > gpurun_out/scan.json 2>gpurun_out/err.log || { tail -5 gpurun_out/err.log; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/scan.json"))
print("$var=$v", "value", round(d["value"]), "ms/step", round(d["ms_per_step"],4), "k5", round(d["kernels"]["k5_rows_ms"]*1e3), "k8", round(d["kernels"]["k8_cols_ms"]*1e3), flush=True)
PY
done
