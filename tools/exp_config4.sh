#!/bin/bash
# parity of the generic patch kernel + BASELINE configs[3] timing:  gpurun -- "bash tools/exp_config4.sh [frames]"
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/exp_config4; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_flow.py tests/test_gpu_stereo.py -x -q -k "${TESTK:-patchgrid or config4 or rgb or stereo or cost_functions or random_flow or large_motion}" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
timeout 600 python tools/config4_probe.py ${1:-16} 2>&1 | grep -v "^TIME\|^$" | tail -4
timeout 600 python tools/config4_probe.py ${1:-16} 2>&1 | grep "TIME" | tail -14
echo "--- one patch per wavefront (OFDIS_RGB12_LPP=64)"
OFDIS_RGB12_LPP=64 timeout 600 python tools/config4_probe.py ${1:-16} 2>&1 | grep -v "^TIME\|^$" | tail -2
OFDIS_RGB12_LPP=64 timeout 600 python tools/config4_probe.py ${1:-16} 2>&1 | grep "TIME" | tail -3 | head -2
