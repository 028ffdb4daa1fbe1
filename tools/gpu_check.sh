#!/bin/bash
# One GPU session: the whole GPU test-suite, then the default bench with a summary of its blocks.
#   gpurun --timeout 2700 -- "bash tools/gpu_check.sh [TAG]"   ->  gpurun_out/TAG/{pytest.log, bench.json}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-check}
mkdir -p $OUT
cd $R
timeout 1800 python -m pytest tests -m gpu -x -q --durations=8 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -14 $OUT/pytest.log
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 400 $OUT/bench.err
python - $OUT/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"])
for k in ("batch512","small_batch","dropin_latency","warp_standalone","tv_off"): print(k, d.get(k))
e=d.get("e2e",{}); print("e2e", {k:e.get(k) for k in ("value","ms_per_step","build_pyramids","upsample_crop","error")})
c=d.get("config4",{}); print("config4", {k:c.get(k) for k in ("value","ms_per_frame","error")}, {k:v["ms_per_step"] for k,v in c.get("kernels",{}).items()}, c.get("cpu_baseline"))
print("cpu", d.get("cpu_baseline"))
PY
