"""The workload of a PMC counter pass, without torch: `--passes` un-pipelined passes of the headline configuration (1024x436,
operating point 2, TV on) over `--batch` pairs under `--contract`, through the C ABI only.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT -- python tools/pmc_pass.py --batch 8192 --contract fused

rocprofv3's counter collection does not survive torch's own kernels on this stack (bench.py under --pmc dies in its frame
generation); the kernels' counters do not depend on the image content at this operating point (fixed iteration counts), so
four synthetic pairs of tools/gen_synth.py are tiled over the batch.  tools/pmc_traffic.py turns the passes into
profiles/traffic.json; bench.py attaches it when contract / TV / batch match the run's sub-batch.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_synth  # noqa: E402
from of_dis_amd import capi  # noqa: E402
from of_dis_amd.params import oppoint  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8192)
ap.add_argument("--contract", choices=["exact", "fused"], default="fused")
ap.add_argument("--passes", type=int, default=4)
a = ap.parse_args()
W, H, T = 1024, 436, 4
L = capi.lib()
capi.check(L.ofdis_set_device(0))
capi.set_tuning(contract=1 if a.contract == "fused" else 0)
pairs = [gen_synth.make_pair(W, H, 1234 + k)[:2] for k in range(T)]
blk_a = np.ascontiguousarray(np.stack([p[0] for p in pairs]))
blk_b = np.ascontiguousarray(np.stack([p[1] for p in pairs]))
assert a.batch % T == 0
da, db = capi.Dev(nbytes=a.batch * W * H), capi.Dev(nbytes=a.batch * W * H)
for i in range(a.batch // T):
    capi.check(L.ofdis_memcpy_h2d(da.ptr + i * blk_a.nbytes, blk_a.ctypes.data, blk_a.nbytes))
    capi.check(L.ofdis_memcpy_h2d(db.ptr + i * blk_b.nbytes, blk_b.ctypes.data, blk_b.nbytes))
p = oppoint(2, W, H, noc=1, usetvref=True, verbosity=0)
b = capi.Batch(p, a.batch)
b.build_pyramids_u8(da.ptr, db.ptr, W, H)
capi.check(L.ofdis_sync(None))
for _ in range(a.passes):
    b.run()
capi.check(L.ofdis_sync(None))
print("pmc_pass done:", a.batch, "pairs,", a.passes, "passes,", a.contract, "contract, status", b.status())
b.close()
