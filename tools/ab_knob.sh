#!/bin/bash
# Same-box A/B of a kernel-selection knob on the headline workload:   bash tools/ab_knob.sh ROUNDS ENVVAR[=VALUE] [bench.py args...]
# (round r: bench.py without the variable, then with it; prints value, ms per step and the per-kernel times)
rounds=$1; var=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
name=${var%%=*}; val=${var#*=}; [ "$val" = "$var" ] && val=1
for r in $(seq $rounds); do
  for v in default "$name=$val"; do
    if [ "$v" = default ]; then unset $name; else export $name=$val; fi
    timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --cpu-seconds 0 --no-parity "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernels']
print('$v', d['contract'], round(d['value']), d['ms_per_step'], ' '.join(f\"{n}={k[n]['ms_per_step']}({k[n]['frac_of_hbm_peak']})\" for n in k), 'tv levels', k.get('tv_fused',{}).get('ms_per_level'), 'prep levels', k.get('derivatives',{}).get('ms_per_level'))"
  done
done
unset $name
