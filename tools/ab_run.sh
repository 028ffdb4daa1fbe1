#!/bin/bash
# A/B of library builds made by tools/ab_build.py on the 64-pair and 1-pair workloads:  bash tools/ab_run.sh NAME [NAME...]
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/ab; mkdir -p $OUT
run() { env $2 timeout 300 python bench.py --no-extras --cpu-seconds 0 --no-parity $3 > $OUT/$1.json 2>> $OUT/err.log
  python - "$1" $OUT/$1.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], "fps", d["value"], "ms/step", d["ms_per_step"], {k:v["ms_per_step"] for k,v in d["kernels"].items()})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
for b in ${AB_BATCHES:-1 64}; do
run base_b$b "A=1" "--steps 300 --warmup 20 --batch $b"
for n in "$@"; do
run ${n}_b$b "OFDIS_LIB=$R/of_dis_amd/lib/ab_$n/libofdis_hip.so" "--steps 300 --warmup 20 --batch $b"
done; done
