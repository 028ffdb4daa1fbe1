"""Developer probe: what the sequence driver's device stage is made of (one 1024x436 chunk through one slot, phase by phase),
and what the link gives to asynchronous copies of pinned memory on one / two streams.   python tools/link_probe.py [chunk=64]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_synth  # noqa: E402
from of_dis_amd import capi  # noqa: E402
from of_dis_amd.params import oppoint  # noqa: E402

C_ = int(sys.argv[1]) if len(sys.argv) > 1 else 64
W, H = 1024, 436
L = capi.lib()
capi.check(L.ofdis_set_device(0))
p = oppoint(2, W, H, noc=1, usetvref=True, verbosity=0)
ia, ib, _ = gen_synth.make_pair(W, H, 77)
img = W * H
flo = 2 * W * H * 4


def t(fn, n=10):
    fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n


slots = []
for k in range(3):
    s = capi.Stream()
    ha, hb = capi.HostBuf((C_, H, W), np.uint8), capi.HostBuf((C_, H, W), np.uint8)
    ha.array[:] = ia
    hb.array[:] = ib
    hf = capi.HostBuf((C_, H, W, 2), np.float32)
    da, db, df = capi.Dev(nbytes=C_ * img), capi.Dev(nbytes=C_ * img), capi.Dev(nbytes=C_ * flo)
    b = capi.Batch(p, C_)
    slots.append((s, ha, hb, hf, da, db, df, b))


def up(k):
    s, ha, hb, hf, da, db, df, b = slots[k]
    capi.check(L.ofdis_memcpy_h2d_async(da.ptr, ha.ptr, C_ * img, s.ptr))
    capi.check(L.ofdis_memcpy_h2d_async(db.ptr, hb.ptr, C_ * img, s.ptr))


def pyr(k):
    s, ha, hb, hf, da, db, df, b = slots[k]
    b.build_pyramids_u8(da.ptr, db.ptr, W, H, s.ptr)


def run(k):
    s, ha, hb, hf, da, db, df, b = slots[k]
    b.run(s.ptr)


def ups(k):
    s, ha, hb, hf, da, db, df, b = slots[k]
    capi.check(L.ofdis_batch_upsample_frames(b.h, 0, C_, df.ptr, W, H, s.ptr))


def down(k):
    s, ha, hb, hf, da, db, df, b = slots[k]
    capi.check(L.ofdis_memcpy_d2h_async(hf.ptr, df.ptr, C_ * flo, s.ptr))


def sync(k):
    slots[k][0].sync()


for name, fn in (("h2d", up), ("pyramids", pyr), ("run", run), ("upsample", ups), ("d2h", down)):
    dt = t(lambda: (fn(0), sync(0)))
    extra = ""
    if name == "h2d":
        extra = f"  {2 * C_ * img / dt / 1e9:.1f} GB/s"
    if name == "d2h":
        extra = f"  {C_ * flo / dt / 1e9:.1f} GB/s"
    print(f"{name:10s} {dt * 1e3:8.3f} ms{extra}")
allp = lambda k: (up(k), pyr(k), run(k), ups(k), down(k))  # noqa: E731
dt = t(lambda: (allp(0), sync(0)))
print(f"one slot, everything, sync   {dt * 1e3:8.3f} ms = {C_ / dt:.0f} pairs/s")
# d2h on one stream while h2d runs on another
dt = t(lambda: (down(0), up(1), sync(0), sync(1)))
print(f"d2h || h2d on two streams    {dt * 1e3:8.3f} ms  (d2h {C_ * flo / dt / 1e9:.1f} GB/s + h2d {2 * C_ * img / dt / 1e9:.1f} GB/s)")
dt = t(lambda: (down(0), down(1), sync(0), sync(1)))
print(f"d2h || d2h on two streams    {dt * 1e3:8.3f} ms  ({2 * C_ * flo / dt / 1e9:.1f} GB/s)")
for D in (1, 2, 3):
    def loop(n):
        for i in range(n):
            k = i % D
            sync(k)
            allp(k)
        for k in range(D):
            sync(k)
    loop(2 * D)
    n = 24
    t0 = time.perf_counter()
    loop(n)
    dt = (time.perf_counter() - t0) / n
    print(f"slots in flight: {D}   {dt * 1e3:8.3f} ms per chunk = {C_ / dt:.0f} pairs/s")
# uploads on ONE stream, downloads on ONE stream, kernels on the slot's stream, events in between (run_seq_main.cpp)
s_in, s_out = capi.Stream(), capi.Stream()
ev = [(capi.Event(), capi.Event(), capi.Event()) for _ in range(3)]
for D in (1, 2, 3):
    def chunk3(k):
        s, ha, hb, hf, da, db, df, b = slots[k]
        e_up, e_done, e_down = ev[k]
        capi.check(L.ofdis_memcpy_h2d_async(da.ptr, ha.ptr, C_ * img, s_in.ptr))
        capi.check(L.ofdis_memcpy_h2d_async(db.ptr, hb.ptr, C_ * img, s_in.ptr))
        e_up.record(s_in)
        e_up.wait(s)
        pyr(k); run(k); ups(k)
        e_done.record(s)
        e_done.wait(s_out)
        capi.check(L.ofdis_memcpy_d2h_async(hf.ptr, df.ptr, C_ * flo, s_out.ptr))
        e_down.record(s_out)

    def loop3(n):
        for i in range(n):
            k = i % D
            ev[k][2].sync()
            chunk3(k)
        for k in range(D):
            ev[k][2].sync()
    loop3(2 * D)
    n = 24
    t0 = time.perf_counter()
    loop3(n)
    dt = (time.perf_counter() - t0) / n
    print(f"three streams, slots in flight: {D}   {dt * 1e3:8.3f} ms per chunk = {C_ / dt:.0f} pairs/s")
