#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/r3w; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_flow.py -x -q -k "varref or batch_matches or fallback or odd_geometries or dropin or strips or graph or uneven or kernel_selection" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
( for b in 1 64 256 512; do
  KB="--steps 50 --warmup 5 --no-extras --pipeline 1 --batch $b"
  echo -n "b$b xcu: "; timeout 300 python tools/kbench.py -- $KB
done ) 2>&1 | sed "s#$R/##g" | tee $OUT/variants.txt
timeout 600 python bench.py --no-parity --cpu-seconds 0 --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value', d['value']); print('small', d['small_batch']['ms_per_step'], 'dropin', d['dropin_latency']['ms_per_call'], 'b512', d['batch512']['ms_per_step'])"
