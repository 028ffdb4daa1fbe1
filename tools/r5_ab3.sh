#!/bin/bash
# round 5, call 4: batch-size quantisation of the headline (the level-3 fused TV launch holds 768 workgroups per round)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/r5d; mkdir -p $OUT
for b in 16384 12288 24576 32768; do for pl in 2 4; do
timeout 300 python bench.py --no-extras --cpu-seconds 0 --no-parity --contract fused --batch $b --pipeline $pl 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernels']
print('batch $b pipeline $pl', round(d['value']), d['ms_per_step'], ' '.join(f\"{n}={k[n]['ms_per_step']}\" for n in k), 'tv levels', k.get('tv_fused',{}).get('ms_per_level'))" | tee -a $OUT/batch.txt
done; done
