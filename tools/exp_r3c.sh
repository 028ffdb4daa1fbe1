#!/bin/bash
# round-3: the record-layout fused TV path (prep kernel, densify quads, record finish): GPU test-suite + kernel table
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/${1:-r3c}; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q ${PYTEST_ARGS} > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest.log
KB="--steps 10 --warmup 3 --no-extras"
( echo -n "p1: "; timeout 300 python tools/kbench.py A=1 -- $KB --pipeline 1
  echo -n "p2: "; timeout 300 python tools/kbench.py A=1 -- $KB --pipeline 2
  for x in $EXTRA_RUNS; do echo -n "$x p2: "; timeout 300 python tools/kbench.py $x -- $KB --pipeline 2; done
) 2>&1 | tee $OUT/variants.txt
