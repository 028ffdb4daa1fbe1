#!/bin/bash
# GPU session r2d: fixup-free divisions A/B, batch / pipeline concurrency sweep
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r2d
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "varref or flow_dropin or random_configurations or golden or upsample" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
run() { # label, env, args
  env $2 timeout 300 python bench.py --no-extras --cpu-seconds 0 --no-parity $3 > $OUT/$1.json 2>> $OUT/err.log
  python - "$1" $OUT/$1.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], "fps", d["value"], "ms/step", d["ms_per_step"], {k:v["ms_per_step"] for k,v in d["kernels"].items()})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
for rep in 1 2; do
run nofix_$rep "A=1" "--steps 20"
run fixup_$rep "OFDIS_LIB=$R/of_dis_amd/lib/ab_fixup/libofdis_hip.so" "--steps 20"
done
run b4096_p1 "A=1" "--steps 20 --pipeline 1"
run b4096_p3 "A=1" "--steps 20 --pipeline 3"
run b8192_p2 "A=1" "--steps 10 --batch 8192 --pipeline 2"
run b8192_p3 "A=1" "--steps 10 --batch 8192 --pipeline 3"
run b8192_p4 "A=1" "--steps 10 --batch 8192 --pipeline 4"
run b16384_p4 "A=1" "--steps 6 --batch 16384 --pipeline 4"
run b16384_p2 "A=1" "--steps 6 --batch 16384 --pipeline 2"
