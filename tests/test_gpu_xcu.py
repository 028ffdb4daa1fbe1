"""-m gpu: the cross-CU variant of the fused TV kernel (ofdis_fused_xcu.hip) must NEVER return a wrong flow with status 0.

Its workgroups hand du/dv rows to each other through global memory and wait -- bounded -- for workgroups with a lower
block index.  These tests force the wait to expire (ofdis_tuning.fused_xcu_spin = 1: the first re-read gives up), run the
variant on a CU-masked stream (8 compute units) and beside a second process that keeps the device busy, and check every
synchronising route: ofdis_sync, ofdis_batch_status, ofdis_batch_download, ofdis_flow, the run_OF_INT binary.
Round 6: the bound is a TIME (50 ms of the device's wall clock by default) -- a producer that really never hands over
(ofdis_tuning.fused_xcu_drop, a test hook) costs a drop-in call one bounded wait plus one repeated pass, < 100 ms -- and
several passes of several contexts in flight on several streams (how small shares keep the chip busy) stay exact.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from common import assert_bits_equal, synth_case

pytestmark = pytest.mark.gpu
ERR_DEVICE = -3


def _fill(b, cases, n):
    for slot in range(n):
        c = cases[slot % len(cases)]
        b.upload(slot, c[1][0], c[1][1], c[1][2], c[2][0])


def _fill_all(b, cs, n, shift=0):
    """Slot s of the context holds case (s + shift) % len(cs): one host-to-device copy per plane kind and level."""
    p = cs[0][0]
    for l in range(p.sc_l, p.sc_f + 1):
        for kind in range(4):
            planes = [c[1][kind][l] if kind < 3 else c[2][0][l] for c in cs]
            b.set_input(l, kind, np.stack([planes[(s + shift) % len(cs)] for s in range(n)]))


@pytest.fixture
def cases(orc):
    cs = [synth_case(1024, 436, 2700 + k, 1, 2, 1) for k in range(2)]
    refs = [orc.flow(c[0], c[1][0], c[1][1], c[1][2], c[2][0]) for c in cs]
    return cs, refs


def test_lost_handover_is_reported_by_every_synchronising_route(gpu, cases):
    """fused_xcu_spin = 1: every workgroup that has to wait gives up at once.  The pass must be reported as failed by
    ofdis_sync (on the stream of the pass), ofdis_batch_status and ofdis_batch_download; the failure belongs to THAT context
    (a second context run in between is unaffected); the next pass of the context no longer uses the variant and is exact."""
    cs, refs = cases
    p = cs[0][0]
    L = gpu.lib()
    old = gpu.set_tuning(fused_xcu_max=1 << 30, fused_xcu_spin=1)
    try:
        b = gpu.Batch(p, 6)
        _fill(b, cs, 6)
        gpu.check(L.ofdis_sync(None))
        b.run()
        rc_sync = L.ofdis_sync(None)
        assert rc_sync == ERR_DEVICE, "a lost hand-over must make ofdis_sync fail"
        assert b"hand-over" in L.ofdis_last_error()
        assert b.status() == ERR_DEVICE, "the failure stays with the context until its next pass"
        with pytest.raises(gpu.OfdisError):
            b.download(0)
        # another context, default spin limit: unaffected by the first one's failure
        gpu.set_tuning(fused_xcu_spin=0)
        other = gpu.Batch(p, 2)
        _fill(other, cs, 2)
        other.run()
        assert L.ofdis_sync(None) == 0, "ofdis_sync reports a failed pass once; the context itself keeps saying so"
        assert b.status() == ERR_DEVICE and other.status() == 0
        assert_bits_equal(other.download(1), refs[1], "an unrelated context")
        other.close()
        # the failed context again: the variant is off for it, the pass is exact, every route reports success
        b.run()
        assert L.ofdis_sync(None) == 0
        assert b.status() == 0
        out = b.download_all()
        for slot in range(6):
            assert_bits_equal(out[slot], refs[slot % 2], f"second pass, slot {slot}")
        b.close()
    finally:
        gpu.restore_tuning(old)


def test_dropin_repeats_the_pass_itself(gpu, cases):
    """ofdis_flow is synchronous: when the pass reports a lost hand-over it repeats it (without the variant) and returns the
    right flow with status 0."""
    cs, refs = cases
    old = gpu.set_tuning(fused_xcu_max=1 << 30, fused_xcu_spin=1)
    try:
        for rep in range(3):
            for k, c in enumerate(cs):
                assert_bits_equal(gpu.flow(c[0], c[1][0], c[1][1], c[1][2], c[2][0]), refs[k], f"ofdis_flow, call {rep}, pair {k}")
    finally:
        gpu.restore_tuning(old)


def test_a_real_stall_at_the_default_bound_costs_less_than_100_ms(gpu, cases):
    """The failure the bound exists for, not its shortcut: with ofdis_tuning.fused_xcu_drop the first fixed-point iteration
    never hands its rows over, so the iteration behind it waits out the DEFAULT bound (fused_xcu_spin = 0: 50 ms of the
    device's wall clock) before it reports the pass as failed.  ofdis_flow then repeats the pass on the other mapping: right
    bits, and the whole call -- bounded wait + repeated pass -- stays under 100 ms (a drop-in must not be
    able to stall for seconds; the bound used to be 2^22 re-reads = ~4 s).  A batch context: the failed pass is reported
    after the bound, not later."""
    import time
    cs, refs = cases
    c = cs[0]
    L = gpu.lib()

    def one_call():
        L.ofdis_flow_cache_clear()  # a fresh context: one that has seen a failure never launches the variant again
        t0 = time.perf_counter()
        out = gpu.flow(c[0], c[1][0], c[1][1], c[1][2], c[2][0])
        return time.perf_counter() - t0, out
    old = gpu.set_tuning(fused_xcu_max=1 << 30, fused_xcu_spin=0, fused_xcu_drop=0)
    try:
        one_call()
        t_normal, out = one_call()  # context creation + one pass
        assert_bits_equal(out, refs[0], "normal call")
        gpu.set_tuning(fused_xcu_drop=1)
        t_drop, out = one_call()
        assert_bits_equal(out, refs[0], "call whose first pass lost its hand-over")
        extra = t_drop - t_normal
        assert 0.03 < extra < 0.1, f"bounded wait + repeated pass took {extra * 1e3:.1f} ms (normal call {t_normal * 1e3:.1f} ms)"
        assert t_drop < 0.2, f"{t_drop * 1e3:.1f} ms end to end"
        # the same through a batch context: reported by the synchronising routes, about one bound after the launch
        b = gpu.Batch(c[0], 2)
        _fill(b, cs, 2)
        gpu.check(L.ofdis_sync(None))
        t0 = time.perf_counter()
        b.run()
        rc = L.ofdis_sync(None)
        dt = time.perf_counter() - t0
        assert rc == ERR_DEVICE and b.status() == ERR_DEVICE
        assert 0.03 < dt < 0.1, f"the failed pass took {dt * 1e3:.1f} ms to report itself"
        gpu.set_tuning(fused_xcu_drop=0)
        b.run()  # (the context has switched the variant off)
        assert L.ofdis_sync(None) == 0 and b.status() == 0
        assert_bits_equal(b.download(1), refs[1], "the pass after the reported failure")
        b.close()
    finally:
        gpu.restore_tuning(old)
        L.ofdis_flow_cache_clear()


@pytest.mark.parametrize("depth,nfr", [(4, 64), (8, 6), (3, 768)])
def test_several_passes_in_flight(gpu, cases, depth, nfr):
    """How a small share keeps the chip busy (bench.py small_batch.depth, run_OF_*_seq --depth): `depth` contexts, each on
    its own stream (ofdis_stream_create), their passes enqueued round-robin without waiting for each other -- up to `depth`
    launches of the cross-CU kernel spinning side by side.  A workgroup only ever waits for a LOWER block index of its own
    launch, and every launch's workgroups start in index order, so the lowest unfinished workgroup of every launch is always
    running: no deadlock by construction.  Checked: every pass of every context reports success and has the oracle's bits."""
    cs, refs = cases
    p = cs[0][0]
    L = gpu.lib()
    old = gpu.set_tuning(fused_xcu_max=1 << 30)
    try:
        streams = [gpu.Stream() for _ in range(depth)]
        ctx = []
        for k in range(depth):
            b = gpu.Batch(p, nfr)
            _fill_all(b, cs, nfr, shift=k)
            ctx.append(b)
        gpu.check(L.ofdis_sync(None))
        rounds = 30 if nfr <= 64 else 6
        for r in range(rounds):
            for k in range(depth):
                ctx[k].run(streams[k].ptr)
            if r % 5 == 4 or r == rounds - 1:
                for k in range(depth):
                    assert L.ofdis_sync(streams[k].ptr) == 0, f"round {r}, context {k}: {L.ofdis_last_error()}"
                    assert ctx[k].status() == 0
                    for slot in sorted({0, 1, nfr // 2, nfr - 1}):
                        assert_bits_equal(ctx[k].download(slot, streams[k].ptr), refs[(slot + k) % 2],
                                          f"{depth} passes in flight, round {r}, context {k}, slot {slot}")
        for b in ctx:
            b.close()
        for s in streams:
            s.close()
    finally:
        gpu.restore_tuning(old)


def test_cli_never_writes_a_wrong_flo(gpu, tmp_path):
    """run_OF_INT takes the device-pointer route (ofdis_batch_run -> ofdis_batch_upsample -> ofdis_sync -> copy).  With the
    wait forced to expire (OFDIS_FUSED_XCU_SPIN=1) it must notice, repeat the pass and write the same bytes as a normal run."""
    import gen_synth
    from of_dis_amd import build
    ia, ib, _ = gen_synth.make_pair(1024, 436, 99)
    fa, fb = str(tmp_path / "a.pgm"), str(tmp_path / "b.pgm")
    gen_synth.write_pgm(fa, ia)
    gen_synth.write_pgm(fb, ib)
    exe = os.path.join(os.path.dirname(build.lib_path()), "run_OF_INT")
    outs = []
    for name, env in (("normal", {}), ("forced", {"OFDIS_FUSED_XCU_SPIN": "1", "OFDIS_FUSED_XCU_MAX": "1073741824"})):
        out = str(tmp_path / f"{name}.flo")
        r = subprocess.run([exe, fa, fb, out, "2"], env=dict(os.environ, **env), capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, (name, r.stdout, r.stderr)
        if name == "forced":
            assert "hand-over" in r.stderr, "the binary must have noticed the failed pass"
        outs.append(open(out, "rb").read())
    assert outs[0] == outs[1], "the forced-failure run wrote another .flo than the normal run"


def _hip():
    for name in ("libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"):
        try:
            return C.CDLL(name)
        except OSError:
            continue
    pytest.skip("libamdhip64.so not loadable")


@pytest.mark.parametrize("nfr", [1, 768])
def test_on_a_cu_masked_stream(gpu, cases, nfr):
    """The variant's forward progress rests on workgroups starting in block-index order while far more workgroups are
    launched than fit on the chip.  A stream restricted to 8 compute units (hipExtStreamCreateWithCUMask) makes that as tight
    as it gets: up to 3072 workgroups per launch on 8 CUs.  Either the bits are right or the pass reports itself as failed."""
    cs, refs = cases
    p = cs[0][0]
    hip = _hip()
    stream = C.c_void_p()
    words = 8
    mask = (C.c_uint32 * words)(*([0xFF] + [0] * (words - 1)))
    hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
    if hip.hipExtStreamCreateWithCUMask(C.byref(stream), words, mask) != 0:
        pytest.skip("hipExtStreamCreateWithCUMask failed")
    L = gpu.lib()
    old = gpu.set_tuning(fused_xcu_max=1 << 30)
    try:
        b = gpu.Batch(p, nfr)
        _fill(b, cs, nfr)
        gpu.check(L.ofdis_sync(None))
        for rep in range(3):
            b.run(stream)
            rc = L.ofdis_sync(stream)
            assert rc in (0, ERR_DEVICE)
            assert (b.status() == 0) == (rc == 0)
            if rc == 0:
                out = b.download_all()
                for slot in sorted({0, nfr // 2, nfr - 1}):
                    assert_bits_equal(out[slot], refs[slot % 2], f"{nfr} pairs on 8 CUs, pass {rep}, slot {slot}")
            else:  # reported: the repeated pass (variant off) must be exact
                b.run(stream)
                assert L.ofdis_sync(stream) == 0
                assert_bits_equal(b.download(0), refs[0], "pass repeated after a reported failure")
        b.close()
    finally:
        gpu.restore_tuning(old)
        hip.hipStreamDestroy.argtypes = [C.c_void_p]
        hip.hipStreamDestroy(stream)


_LOAD = r"""
import sys, time
sys.path.insert(0, {root!r}); sys.path.insert(0, {root!r} + "/tools"); sys.path.insert(0, {root!r} + "/tests")
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
while time.time() - t0 < {seconds}:
    b.run()
    capi.check(capi.lib().ofdis_sync(None))
"""


def test_beside_a_second_process(gpu, cases):
    """A second PROCESS keeps every CU busy with 3000-pair passes of the throughput kernels while this one pushes single pairs
    through ofdis_flow and runs a 768-pair context (cross-CU variant): right bits, pass after pass, or a reported failure."""
    cs, refs = cases
    p = cs[0][0]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    load = subprocess.Popen([sys.executable, "-c", _LOAD.format(root=root, seconds=25)], stdout=subprocess.PIPE, text=True)
    try:
        assert load.stdout.readline().strip() == "ready"
        L = gpu.lib()
        b = gpu.Batch(p, 768)
        _fill(b, cs, 768)
        for rep in range(10):
            for k, c in enumerate(cs):
                assert_bits_equal(gpu.flow(c[0], c[1][0], c[1][1], c[1][2], c[2][0]), refs[k], f"ofdis_flow call {rep}, pair {k}")
            b.run()
            rc = L.ofdis_sync(None)
            assert rc in (0, ERR_DEVICE)
            if rc == 0:
                for slot in (0, 383, 767):
                    assert_bits_equal(b.download(slot), refs[slot % 2], f"768-pair context, pass {rep}, slot {slot}")
        b.close()
    finally:
        load.kill()
        load.wait()


def test_graph_replay_is_recaptured_after_a_lost_handover(gpu, cases):
    """ofdis_batch_set_graph(1): the captured launch graph contains the cross-CU kernel.  After a lost hand-over the context
    must not replay that graph ("run again" has to run WITHOUT the variant): the second pass is exact and reports success,
    and so is a third one (the re-captured graph)."""
    cs, refs = cases
    p = cs[0][0]
    L = gpu.lib()
    old = gpu.set_tuning(fused_xcu_max=1 << 30, fused_xcu_spin=1, graph=1)
    try:
        b = gpu.Batch(p, 6)
        _fill(b, cs, 6)
        gpu.check(L.ofdis_sync(None))
        b.set_graph(1)
        b.run()
        assert L.ofdis_sync(None) == ERR_DEVICE, "the captured pass must report the lost hand-over"
        for rep in range(2):
            b.run()
            assert L.ofdis_sync(None) == 0, f"pass {rep} after the failure replayed a graph that still uses the variant"
            assert b.status() == 0
            out = b.download_all()
            for slot in range(6):
                assert_bits_equal(out[slot], refs[slot % 2], f"graph mode, pass {rep} after the failure, slot {slot}")
        b.close()
    finally:
        gpu.restore_tuning(old)


def test_failure_nobody_polled_is_reported_late_once(gpu, cases):
    """A pass loses a hand-over, the caller synchronises through HIP directly (no ofdis_* poll) and starts the next pass: the
    earlier failure must not be swallowed -- the next synchronising route reports it once -- and the pass after that is clean."""
    cs, refs = cases
    p = cs[0][0]
    L = gpu.lib()
    hip = _hip()
    old = gpu.set_tuning(fused_xcu_max=1 << 30, fused_xcu_spin=1)
    try:
        b = gpu.Batch(p, 4)
        _fill(b, cs, 4)
        gpu.check(L.ofdis_sync(None))
        b.run()
        assert hip.hipDeviceSynchronize() == 0     # the caller's own synchronisation: nobody asked the library
        b.run()                                    # the variant is off now; the earlier pass's failure stays latched
        assert L.ofdis_sync(None) == ERR_DEVICE and b"EARLIER" in L.ofdis_last_error()
        assert b.status() == 0, "said once"
        out = b.download_all()                     # ... and this pass's results are the right ones
        for slot in range(4):
            assert_bits_equal(out[slot], refs[slot % 2], f"slot {slot}")
        b.run()
        assert L.ofdis_sync(None) == 0
        b.close()
    finally:
        gpu.restore_tuning(old)


def test_sequence_driver_repeats_a_failed_pass(gpu, tmp_path):
    """run_OF_INT_seq (resident contexts per share -- one per chunk in flight --, chunks of the list): with the wait forced to
    expire in every pass that uses the variant it must notice (ofdis_sync on the slot's stream / ofdis_batch_status of the
    slot's context), repeat the chunk's pass and write the bytes of a normal run -- also with two shares on the one device
    and with one or three chunks in flight."""
    import gen_synth
    from of_dis_amd import build
    n, w, h = 10, 320, 192
    lines = []
    for k in range(n):
        ia, ib, _ = gen_synth.make_pair(w, h, 900 + k)
        fa, fb = str(tmp_path / f"a{k}.pgm"), str(tmp_path / f"b{k}.pgm")
        gen_synth.write_pgm(fa, ia)
        gen_synth.write_pgm(fb, ib)
        lines.append((fa, fb))
    exe = os.path.join(os.path.dirname(build.lib_path()), "run_OF_INT_seq")
    args = "5 3 12 12 0.05 0.95 0 8 0.40 0 1 0 1 10 10 5 1 3 1.6 0".split()
    outs = {}
    for name, env, opts in (("normal", {}, ["--chunk", "4"]),
                            ("forced", {"OFDIS_FUSED_XCU_SPIN": "1", "OFDIS_FUSED_XCU_MAX": "1073741824"}, ["--chunk", "4"]),
                            ("forced2", {"OFDIS_FUSED_XCU_SPIN": "1", "OFDIS_FUSED_XCU_MAX": "1073741824"}, ["--devices", "0,0", "--chunk", "3"]),
                            ("forced3", {"OFDIS_FUSED_XCU_SPIN": "1", "OFDIS_FUSED_XCU_MAX": "1073741824"}, ["--chunk", "2", "--depth", "3"]),
                            ("forced1", {"OFDIS_FUSED_XCU_SPIN": "1", "OFDIS_FUSED_XCU_MAX": "1073741824"}, ["--chunk", "4", "--depth", "1"])):
        lst = tmp_path / f"{name}.txt"
        lst.write_text("".join(f"{fa} {fb} {tmp_path}/{name}{k}.flo\n" for k, (fa, fb) in enumerate(lines)))
        r = subprocess.run([exe, str(lst)] + opts + args, env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (name, r.stdout, r.stderr)
        if name != "normal":
            assert "hand-over" in r.stderr, "the driver must have noticed the failed pass"
        outs[name] = [open(tmp_path / f"{name}{k}.flo", "rb").read() for k in range(n)]
    for name in ("forced", "forced2", "forced3", "forced1"):
        assert outs[name] == outs["normal"], name
