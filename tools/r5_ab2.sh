#!/bin/bash
# round 5, call 3: parity subset (prep C=2, patch prologue), then same-box A/B: main | patch0 (round-4 prologue) | patchA (coarse flow
# first only) | prepw2 (two-wavefront prep)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/r5c; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -k "prep or varref or golden or random or patchgrid or level_flows or baseline_config or constant or outliers" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
bash tools/ab_bench.sh 2 main patch0 patchA prepw2 2>&1 | tee $OUT/ab.txt
