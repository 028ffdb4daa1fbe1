#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/r4h; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_flow.py -x -q -k "varref or batch_matches or fallback or odd_geometries or dropin or strips or graph or uneven or kernel_selection" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
( for rep in 1 2; do for b in 1 64 512; do
  KB="--steps 100 --warmup 10 --no-extras --pipeline 1 --batch $b"
  echo -n "b$b per-step fetch: "; timeout 300 python tools/kbench.py -- $KB
done; done ) 2>&1 | sed "s#$R/##g;s#OFDIS_LIB=[^ ]* ##" | tee $OUT/variants3.txt
