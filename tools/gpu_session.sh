#!/bin/bash
# One GPU session: the whole GPU test-suite and smoke(), then the evidence of a round on the same box (PMC -> bench -> kernel traces):
#   gpurun --timeout 3000 -- "bash tools/gpu_session.sh TAG"   ->  gpurun_out/TAG/{pytest.log,smoke.log}, gpurun_out/prof_TAG/, gpurun_out/pmc_TAG/
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-r6f}; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q --durations=5 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -9 $OUT/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
bash tools/profile_round.sh ${1:-r6f} 16384 8192 fused > $OUT/profile_round.log 2>&1; tail -5 $OUT/profile_round.log
python - $R/gpurun_out/prof_${1:-r6f}/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "contract", d["contract"])
r=d["roofline"]; print("roofline", {k:r.get(k) for k in ("kernel","bound","achieved","frac","traffic","limited_by","pmc_build_id","traffic_note")})
print("levels", json.dumps(r.get("levels")))
print("pipeline", d.get("pipeline_roofline"))
print("patch issue", d["kernels"]["patch_optimize"].get("issue_roofline"))
for k in ("batch512","small_batch","dropin_latency","warp_standalone","tv_off","scaling_expectation"): print(k, json.dumps(d.get(k))[:1800])
c=d.get("config4",{}); print("config4", {k:c.get(k) for k in ("value","ms_per_frame","error","epe_bar_met","epe_verdict")}, {k:v["ms_per_step"] for k,v in c.get("kernels",{}).items()})
print("cpu", json.dumps(d.get("cpu_baseline"))[:600])
PY
