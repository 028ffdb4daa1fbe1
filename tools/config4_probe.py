"""Developer probe: BASELINE config 4 (run_OF_RGB, 1920x1080, op-4 geometry, L1 cost, 50 iterations, TV on)
through the batched context with verbosity 2 (per-level TIME lines), one frame."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import gen_synth  # noqa: E402
from of_dis_amd import capi  # noqa: E402
from of_dis_amd.params import oppoint  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ia, ib, _ = gen_synth.make_pair(1920, 1080, 4242, 3)
p = oppoint(4, 1920, 1080, noc=3, verbosity=int(os.environ.get("OFDIS_PROBE_VERBOSITY", "0"))).copy(costfct=1, max_iter=50, min_iter=50)
b = capi.Batch(p, n)
da, db = capi.Dev(np.stack([ia] * n)), capi.Dev(np.stack([ib] * n))
b.build_pyramids_u8(da.ptr, db.ptr, 1920, 1080)
capi.check(capi.lib().ofdis_sync(None))
for rep in range(2):
    t0 = time.perf_counter()
    b.run()
    capi.check(capi.lib().ofdis_sync(None))
    print(f"run {rep}: {1e3 * (time.perf_counter() - t0):.1f} ms for {n} frame(s)")
b.timing(True)
b.run()
capi.check(capi.lib().ofdis_sync(None))
print("kernel ms:", " ".join(f"{name}={b.kernel_time(k)[0]:.2f}" for k, name in enumerate(capi.K_NAMES) if b.kernel_time(k)[1]))
