#!/bin/bash
# GPU session r2b: multi-wave TV with the light barrier (small batches), config4 batch sweep
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r2b
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests -m gpu -x -q -k "varref or launch_graph or dropin or batch_matches" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
i=0
for v in "A=1" "OFDIS_FUSED_MW_MAX=0"; do
  for b in 1 64 512 1024 2048; do
    i=$((i+1))
    env $v timeout 200 python bench.py --batch $b --steps 200 --warmup 20 --no-extras --cpu-seconds 0 --no-parity > $OUT/ab_$i.json 2>> $OUT/ab.err
    python - "$v" $b $OUT/ab_$i.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
    print(sys.argv[1], "batch", sys.argv[2], "fps", d["value"], "ms/step", d["ms_per_step"], {k:v["ms_per_step"] for k,v in d["kernels"].items()})
except Exception as e:
    print(sys.argv[1], sys.argv[2], "FAILED", e)
PY
  done
done
python - <<'PY'
import sys, time, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tools")
import torch, bench
from of_dis_amd import capi
from of_dis_amd.params import oppoint
dev = torch.device("cuda", 0)
W4, H4 = 1920, 1080
p4 = oppoint(4, W4, H4, noc=3, verbosity=0).copy(costfct=1, max_iter=50, min_iter=50)
xa, xb = bench.synth_frames_range(0, 64, W4, H4, 4242, dev, channels=3)
st = torch.cuda.Stream(device=dev); s = st.cuda_stream
for n in (1, 8, 32, 64):
    b4 = capi.Batch(p4, n)
    torch.cuda.synchronize()
    b4.build_pyramids_u8(xa.data_ptr(), xb.data_ptr(), W4, H4, s)
    dt = bench.timed_steps(torch, lambda: b4.run(s), 2, 1)
    k = bench.kernel_table(capi, torch, b4, p4, n, s, nrep=1)
    print("config4 batch", n, "ms/frame", round(dt / n * 1e3, 3), "fps", round(n / dt, 1), {a: v["ms_per_step"] for a, v in k.items()}, flush=True)
    b4.close()
PY
