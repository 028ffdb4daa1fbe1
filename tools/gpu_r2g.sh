#!/bin/bash
# GPU session r2g: warp fused into the derivatives kernel: parity + A/B; multi-wave budget sweep
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r2g
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
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
run fusedwarp_$rep "A=1" "--steps 20"
run sepwarp_$rep "OFDIS_NO_WARP_FUSION=1" "--steps 20"
done
run fusedwarp_b64 "A=1" "--steps 200 --warmup 20 --batch 64"
run sepwarp_b64 "OFDIS_NO_WARP_FUSION=1" "--steps 200 --warmup 20 --batch 64"
run fusedwarp_b1 "A=1" "--steps 200 --warmup 20 --batch 1"
for b in 512 1024 2048; do
for m in 2048 4096 8192 16384 32768; do
run mw${m}_b$b "OFDIS_FUSED_MW_MAX=$m" "--steps 100 --warmup 10 --batch $b"
done
done
