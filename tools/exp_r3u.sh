#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/r3u; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_flow.py -x -q -k "varref or batch_matches or fallback or odd_geometries or dropin or strips" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
X=OFDIS_FUSED_XCU_MAX=1073741824
( for b in 1 64; do
  KB="--steps 50 --warmup 5 --no-extras --pipeline 1 --batch $b"
  echo -n "b$b split: "; timeout 300 python tools/kbench.py OFDIS_FUSED_XCU_MAX=0 -- $KB
  echo -n "b$b xcu pf3 lead3: "; timeout 300 python tools/kbench.py $X -- $KB
  for v in l2 l4 pf2l2 pf2l3 s0 s8; do
    echo -n "b$b xcu $v: "; timeout 300 python tools/kbench.py OFDIS_LIB=$R/of_dis_amd/lib/ab_$v/libofdis_hip.so $X -- $KB
  done
done ) 2>&1 | sed "s#$R/##g;s#OFDIS_LIB=[^ ]* ##;s#OFDIS_FUSED_XCU_MAX=[0-9]* ##" | tee $OUT/variants.txt
