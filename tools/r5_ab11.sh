#!/bin/bash
# round 5: one pad row between the strips of the sdiag record arrays (strip stride 7 * 2^18 B at level 3 -> 1792 * 1025 B): do the
# strips of concurrently running workgroups stop aliasing in the L2 / on the HBM channels?
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/r5p; mkdir -p $OUT
OFDIS_LIB=$R/of_dis_amd/lib/ab_pad1/libofdis_hip.so timeout 900 python -m pytest tests -m gpu -x -q -k "prep or varref or golden or fused or strips or level_flows or contract_production or xcu or random_config or batch" > $OUT/pytest_pad1.log 2>&1; tail -2 $OUT/pytest_pad1.log
bash tools/ab_bench.sh 2 main pad1 2>&1 | tee $OUT/ab.txt
export OFDIS_LIB=$R/of_dis_amd/lib/ab_pad1/libofdis_hip.so; bash tools/pmc_round.sh pad1 8192 fused 2>&1 | tail -3
