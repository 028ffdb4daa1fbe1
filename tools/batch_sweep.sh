#!/bin/bash
# batch-size sweep of the headline workload (wave-quantisation of the fused TV kernel: 3 wavefronts per SIMD x 1024 SIMDs)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/batch_sweep; mkdir -p $OUT
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
for b in 3072 4096 6144 8192 9216 12288; do
for p in 1 2; do
run b${b}_p$p "A=1" "--steps 12 --batch $b --pipeline $p"
done; done
run b6144_p3 "A=1" "--steps 12 --batch 6144 --pipeline 3"
run b9216_p3 "A=1" "--steps 12 --batch 9216 --pipeline 3"
