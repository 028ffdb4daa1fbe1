#!/bin/bash
# GPU session r2i: small-batch experiments (8-lane patch kernel, graph on/off)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r2i
mkdir -p $OUT
cd $R
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
for b in 1 64; do
run base_b$b "A=1" "--steps 300 --warmup 20 --batch $b"
run nogray8_b$b "OFDIS_NO_GRAY8=1" "--steps 300 --warmup 20 --batch $b"
run nograph_b$b "OFDIS_NO_GRAPH=1" "--steps 300 --warmup 20 --batch $b"
done
