#!/bin/bash
# cross-CU fused TV variant: parity of the four mappings, then small-batch A/B, then the 3-waves-per-SIMD build of the throughput kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/r3r; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_flow.py -x -q -k "varref or batch_matches or fallback or odd_geometries or dropin or strips" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
KB="--steps 50 --warmup 5 --no-extras --pipeline 1"
( for b in 1 16 64 128 256 512; do
  for x in 0 1073741824; do
    echo -n "batch $b xcu_max $x: "; timeout 300 python tools/kbench.py OFDIS_FUSED_XCU_MAX=$x -- $KB --batch $b
  done
done
L=$R/of_dis_amd/lib/ab_w3/libofdis_hip.so
KB="--steps 8 --warmup 2 --no-extras --pipeline 2 --batch 16384"
for rep in 1 2; do
  echo -n "default: "; timeout 400 python tools/kbench.py -- $KB
  echo -n "w3: "; timeout 400 python tools/kbench.py OFDIS_LIB=$L -- $KB
done ) 2>&1 | sed "s#$R/##g" | tee $OUT/variants.txt
