#!/bin/bash
# round 5: the warp + derivatives kernel is bound by its record writes: longer store runs (KD rows x 32 B, KW rows x 8 B), now
# that its occupancy is known not to matter (LDS: 27 / 37 / 44 KB per block)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/r5i; mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -x -q -k "prep or varref_levels or golden" > $OUT/pytest_main.log 2>&1; tail -1 $OUT/pytest_main.log
for v in kd4 kd4kw16 kd8; do OFDIS_LIB=$R/of_dis_amd/lib/ab_$v/libofdis_hip.so timeout 600 python -m pytest tests -m gpu -x -q -k "prep or varref_levels or golden" > $OUT/pytest_$v.log 2>&1; echo $v; tail -1 $OUT/pytest_$v.log; done
bash tools/ab_bench.sh 2 main kd4 kd4kw16 kd8 2>&1 | tee $OUT/ab.txt
