#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/${1:-r3e}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "patchgrid_levels or varref_levels or trimmed or wave_sum" 2>&1 > $OUT/pytest_k.log; grep -E "^(FAILED|PASSED)|passed|failed" $OUT/pytest_k.log | sed 's/ - .*//' | awk '{print $2}' | sed 's/tests.test_gpu_kernels.py:://' | tr '\n' ' ' | cut -c1-6000; echo; grep -E "AssertionError" $OUT/pytest_k.log | sort | uniq -c | sort -rn | head -30 | cut -c1-260
