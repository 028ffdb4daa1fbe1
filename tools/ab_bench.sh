#!/bin/bash
# Developer helper: same-box A/B of library builds on the headline workload (fused contract unless CONTRACT is set).
#   tools/ab_bench.sh ROUNDS name1 name2 ...     (name = "main" or an ab_NAME directory under of_dis_amd/lib)
rounds=$1; shift
for r in $(seq $rounds); do
  for v in "$@"; do
    if [ "$v" = main ]; then L=of_dis_amd/lib/libofdis_hip.so; else L=of_dis_amd/lib/ab_$v/libofdis_hip.so; fi
    OFDIS_LIB=$L timeout 300 python bench.py --no-extras --cpu-seconds 0 --contract ${CONTRACT:-fused} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernels']
print('$v', round(d['value']), d['ms_per_step'], ' '.join(f\"{n}={k[n]['ms_per_step']}\" for n in k), 'tv levels', k.get('tv_fused',{}).get('ms_per_level'))"
  done
done
