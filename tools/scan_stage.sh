#!/bin/bash
# Usage: tools/scan_stage.sh VAR v1 v2 ...  -- default bench (3 streams + single-stream leg with every stage timed)
# per value of an env knob: headline value, single-stream value and the per-launch stage times in microseconds.
mkdir -p gpurun_out
var=$1; shift
for v in "$@"; do
env "$var=$v" timeout -k 10 300 python bench.py --no-cpu-baseline --steps 40 > gpurun_out/scan.json 2>gpurun_out/err.log || { tail -5 gpurun_out/err.log; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/scan.json"))
s=d.get("single_stream", {})
st=s.get("stages_ms_per_launch", {})
print("$var=$v", "value", round(d["value"]), "single", round(s.get("value", 0)), " ".join("%s %.1f" % (k.replace("_ms",""), x*1e3) for k, x in st.items()), flush=True)
PY
done
