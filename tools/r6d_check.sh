#!/bin/bash
# round 6: GPU test-suite; same-box A/B of the densification inside the warp + derivatives kernel; sequence driver rates
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/${1:-r6d}; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -14 $OUT/pytest.log
for v in dens nodens dens2 nodens2; do
  if [[ $v == nodens* ]]; then export OFDIS_NO_PREP_DENSIFY=1; else unset OFDIS_NO_PREP_DENSIFY; fi
  timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --cpu-seconds 0 > $OUT/bench_$v.json 2> $OUT/bench_$v.err; echo "bench $v rc=$?"
  timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --cpu-seconds 0 --contract exact > $OUT/bench_exact_$v.json 2> $OUT/bench_exact_$v.err; echo "bench exact $v rc=$?"
done
unset OFDIS_NO_PREP_DENSIFY
python - $OUT <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+'/bench_*.json')):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,'ERR',e); continue
    print(f.split('/')[-1], d['contract'], d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['kernels'].items()}, d.get('parity_check'))
PY
timeout 900 python tools/seq_probe.py 512 64 1,2,3 > $OUT/seq_probe.txt 2>&1; cat $OUT/seq_probe.txt
