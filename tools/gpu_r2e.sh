#!/bin/bash
# GPU session r2e: tests after the division rewrite, A/B (fixup, no-barrier timing probe), profile refresh r02_a
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r2e
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "varref or flow_dropin or random_configurations or golden or batch_matches" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
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
run new_$rep "A=1" "--steps 20"
run fixup_$rep "OFDIS_LIB=$R/of_dis_amd/lib/ab_fixup/libofdis_hip.so" "--steps 20"
done
run b64 "A=1" "--steps 200 --warmup 20 --batch 64"
run b64_nobar "OFDIS_LIB=$R/of_dis_amd/lib/ab_nobar/libofdis_hip.so" "--steps 200 --warmup 20 --batch 64"
run b1 "A=1" "--steps 200 --warmup 20 --batch 1"
run b1_nobar "OFDIS_LIB=$R/of_dis_amd/lib/ab_nobar/libofdis_hip.so" "--steps 200 --warmup 20 --batch 1"
bash tools/profile_round.sh r02a > $OUT/profile.log 2>&1; tail -25 $OUT/profile.log
