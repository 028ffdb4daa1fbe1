#!/bin/bash
# GPU session r2a: full GPU test-suite, default bench, small-batch A/B (multi-wave TV / launch graph on and off)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r2a
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 300 $OUT/bench.err
i=0
for v in "A=1" "OFDIS_FUSED_MW_MAX=0" "OFDIS_NO_GRAPH=1" "OFDIS_FUSED_MW_MAX=0 OFDIS_NO_GRAPH=1" "OFDIS_FUSED_MW_MAX=100000"; do
  for b in 64 512 1024; do
    i=$((i+1))
    env $v timeout 200 python bench.py --batch $b --steps 200 --warmup 20 --no-extras --cpu-seconds 0 --no-parity > $OUT/ab_$i.json 2>> $OUT/ab.err
    python - "$v" $b $OUT/ab_$i.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
    print(sys.argv[1], "batch", sys.argv[2], "fps", d["value"], "ms/step", d["ms_per_step"], {k:v["ms_per_step"] for k,v in d["kernels"].items()})
except Exception as e:
    print(sys.argv[1], sys.argv[2], "FAILED", e)
PY
  done
done
