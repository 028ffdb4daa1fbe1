#!/bin/bash
# round 6, first GPU session on the new ABI: the GPU test-suite, the small_batch block of bench.py, the sequence driver's rates
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-r6c}; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -14 $OUT/pytest.log
OFDIS_BENCH_BLOCKS=small_batch timeout 600 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 > $OUT/bench_small.json 2> $OUT/bench_small.err; echo "bench rc=$?"; tail -c 300 $OUT/bench_small.err
python - $OUT/bench_small.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], d["roofline"].get("traffic_note"))
print(json.dumps(d.get("small_batch"), indent=1))
print(json.dumps(d["kernels"]["patch_optimize"].get("issue_roofline"), indent=1))
PY
timeout 600 python tools/seq_probe.py 512 64 1,2,3 > $OUT/seq_probe.txt 2>&1; cat $OUT/seq_probe.txt
