#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for i in 1 2 3; do timeout 900 python -m pytest tests/test_gpu_flow.py -x -q -k "uneven_load or several_threads or pipelined_sub" --durations=3 2>&1 | tail -6; done
