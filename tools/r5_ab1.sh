#!/bin/bash
# round 5, call 2: parity of the compact RGB weights + the derivative ring (subset of the GPU suite), then same-box A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/r5b; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -k "rgb or config4 or varref or fused or strips or sor_coupled or golden or random_config" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
bash tools/ab_bench.sh 2 main drsrc0 drnone 2>&1 | tee $OUT/ab.txt
echo "--- tp_pipe=2 (MODE 1 on every level)"
OFDIS_FUSED_TP_PIPE=2 bash tools/ab_bench.sh 1 main drnone 2>&1 | tee -a $OUT/ab.txt
echo "--- config4 only"
OFDIS_BENCH_BLOCKS=config4 timeout 600 python bench.py --batch 64 --steps 1 --warmup 0 --cpu-seconds 0 --no-parity 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config4']; print('config4', c['value'], c['ms_per_step'], {k:v['ms_per_step'] for k,v in c['kernels'].items()}, c.get('last_two_frames_bit_identical_to_a_batch_of_two'))" | tee -a $OUT/ab.txt
