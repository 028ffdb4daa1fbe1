"""Developer probe: soak of the passes-in-flight regime.  D contexts of n pairs on D streams run passes for `seconds`, optionally
beside a second process that keeps the chip busy with 3000-pair passes; every `check` rounds every context's status is polled and
its flow compared bit for bit with the pass run alone.   python tools/inflight_soak.py [seconds=60] [D=4] [n=64] [load=1]"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from common import synth_case  # noqa: E402
from of_dis_amd import capi  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60
D = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n = int(sys.argv[3]) if len(sys.argv) > 3 else 64
load = int(sys.argv[4]) if len(sys.argv) > 4 else 1
L = capi.lib()
capi.check(L.ofdis_set_device(0))
capi.set_tuning(fused_xcu_max=1 << 30)
cs = [synth_case(1024, 436, 2700 + k, 1, 2, 1) for k in range(3)]
p = cs[0][0]
proc = None
if load:
    code = f"""
import sys, time
sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {ROOT!r} + "/tools"); sys.path.insert(0, {ROOT!r} + "/tests")
import numpy as np
from of_dis_amd import capi
from common import synth_case
c = synth_case(1024, 436, 2700, 1, 2, 1)
p = c[0]
b = capi.Batch(p, 3000)
for l in range(p.sc_l, p.sc_f + 1):
    for kind in range(4):
        plane = c[1][kind][l] if kind < 3 else c[2][0][l]
        b.set_input(l, kind, np.broadcast_to(plane, (3000,) + plane.shape))
print("ready", flush=True)
t0 = time.time()
while time.time() - t0 < {seconds + 20}:
    b.run()
    capi.check(capi.lib().ofdis_sync(None))
"""
    proc = subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, text=True)
    assert proc.stdout.readline().strip() == "ready"
streams = [capi.Stream() for _ in range(D)]
ctx = []
for k in range(D):
    b = capi.Batch(p, n)
    for l in range(p.sc_l, p.sc_f + 1):
        for kind in range(4):
            planes = [c[1][kind][l] if kind < 3 else c[2][0][l] for c in cs]
            b.set_input(l, kind, np.stack([planes[(s + k) % 3] for s in range(n)]))
    ctx.append(b)
capi.check(L.ofdis_sync(None))
alone = []
for k in range(D):
    ctx[k].run(streams[k].ptr)
    streams[k].sync()
    assert ctx[k].status() == 0
    alone.append(ctx[k].download_all())
t0 = time.time()
passes, checks, bad, failed = 0, 0, 0, 0
while time.time() - t0 < seconds:
    for r in range(200):
        for k in range(D):
            ctx[k].run(streams[k].ptr)
        passes += D
    for k in range(D):
        rc = L.ofdis_sync(streams[k].ptr)
        st = ctx[k].status()
        if rc != 0 or st != 0:
            failed += 1
            continue
        if not np.array_equal(ctx[k].download_all(), alone[k]):
            bad += 1
    checks += D
print(f"{passes} passes of {n} pairs with {D} in flight in {time.time() - t0:.1f} s ({'beside a busy second process' if load else 'alone'}): "
      f"{checks} context checks, {failed} reported a failed pass, {bad} wrong flows without a report")
if proc:
    proc.kill()
sys.exit(1 if bad else 0)
