#!/bin/bash
# GPU session r2f: split (producer / solver) multi-wave TV kernel: parity + small-batch timing
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r2f
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "varref or flow_dropin or random_configurations or golden or batch_matches or launch_graph or dropin_context" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
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
for b in 1 64 128 256 512 1024; do
run split_b$b "A=1" "--steps 200 --warmup 20 --batch $b"
run nosplit_b$b "OFDIS_FUSED_NO_SPLIT=1" "--steps 200 --warmup 20 --batch $b"
done
run split_max_b512 "OFDIS_FUSED_MW_MAX=8192" "--steps 200 --warmup 20 --batch 512"
run split_max_b1024 "OFDIS_FUSED_MW_MAX=16384" "--steps 200 --warmup 20 --batch 1024"
