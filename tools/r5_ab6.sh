#!/bin/bash
# round 5: how sensitive is the warp + derivatives kernel (level 3: two wavefronts per block, 18.4 KB LDS = 8 blocks per CU =
# four wavefronts per SIMD) to its occupancy?  LDS padding: +8 KB -> 5 blocks (2.5 per SIMD), +21 KB -> 4 blocks (2 per SIMD)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/r5h; mkdir -p $OUT
bash tools/ab_bench.sh 2 main pp8 pp21 2>&1 | tee $OUT/ab.txt
