#!/bin/bash
# round 5, call 5 (timing only, results of the variants are wrong on purpose): what the gray patch kernel's "fixed" part is made of
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/r5e; mkdir -p $OUT
for it in 12 0; do for v in main nostore l2planes; do
if [ "$v" = main ]; then L=of_dis_amd/lib/libofdis_hip.so; else L=of_dis_amd/lib/ab_$v/libofdis_hip.so; fi
OFDIS_BENCH_PARAMS=max_iter=$it,min_iter=$it OFDIS_LIB=$L timeout 300 python bench.py --no-extras --cpu-seconds 0 --no-parity --contract fused 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernels']
print('iters $it $v', round(d['value']), d['ms_per_step'], 'patch', k['patch_optimize']['ms_per_step'], k['patch_optimize'].get('ms_per_level'))" | tee -a $OUT/patch_parts.txt
done; done
