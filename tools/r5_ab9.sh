#!/bin/bash
# round 5: prep with two 8-byte tap loads per pixel (main) against four 4-byte gathers (tap4); the one-wavefront-per-strip TV
# kernel (levels 4, 5: HBM-bound) at one instead of two wavefronts per SIMD (m0one)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/r5k; mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -x -q -k "prep or varref_levels or golden or image_warp or large_motion or outliers" > $OUT/pytest.log 2>&1; tail -1 $OUT/pytest.log
bash tools/ab_bench.sh 2 main tap4 m0one 2>&1 | tee $OUT/ab.txt
