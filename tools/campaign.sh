#!/bin/bash
# Seed campaign: the seeded random parity tests at other seeds, from poisoned scratch (OFDIS_POISON_SCRATCH=1: every context's
# arena starts as NaN patterns), then the whole GPU suite from poisoned scratch.   gpurun --timeout 2400 -- "bash tools/campaign.sh 1000 3000 5000"
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
for o in "${@:-1000}"; do
  echo "== seed offset $o (poisoned scratch)"
  OFDIS_POISON_SCRATCH=1 OFDIS_TEST_SEED_OFFSET=$o timeout 900 python -m pytest tests/test_gpu_flow.py tests/test_gpu_kernels.py tests/test_gpu_contract.py -q -k random 2>&1 | tail -4
done
echo "== whole GPU suite, poisoned scratch"
OFDIS_POISON_SCRATCH=1 timeout 1500 python -m pytest tests -m gpu -q -k "not eight_ranks" 2>&1 | tail -4
