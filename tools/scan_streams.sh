#!/bin/bash
# bench.py at several stream counts / batch sizes (run on the GPU box)
mkdir -p gpurun_out
for b in ${BATCHES:-16}; do for s in "$@"; do
timeout -k 10 200 python bench.py --no-cpu-baseline --no-md-regime --batch $b --streams $s --steps ${STEPS:-80} > gpurun_out/scan.json 2>gpurun_out/err.log || { tail -5 gpurun_out/err.log; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/scan.json"))
print("batch",$b,"streams",$s, "value", round(d["value"]), "ms/step", round(d["ms_per_step"],4), "k5", round(d["kernels"]["k5_rows_ms"]*1e3), "k8", round(d["kernels"]["k8_cols_ms"]*1e3), flush=True)
PY
done; done
