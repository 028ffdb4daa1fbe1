#!/bin/bash
# round 5: densify_quad -- a quad's four pixels as two 16-byte stores (main) against four 8-byte stores (dens0); its occupancy
# capped by LDS padding (24 KB -> 6 blocks per CU, 40 KB -> 4) -- does the streaming gather like fewer wavefronts in flight?
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/r5j; mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -x -q -k "patchgrid or level_flows or golden or random_config or baseline_config" > $OUT/pytest.log 2>&1; tail -1 $OUT/pytest.log
bash tools/ab_bench.sh 2 main dens0 dp24 dp40 2>&1 | tee $OUT/ab.txt
