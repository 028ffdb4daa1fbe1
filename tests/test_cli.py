"""The run_OF_INT / run_OF_RGB executables: the reference's command-line contract (README.md:48-88,
run_dense.cpp:185-431) -- three invocation variants, .flo layout, TIME lines -- and their output against the
oracle's restatement of the whole run_dense pipeline (pad, pyramid, OFClass, upsample, crop)."""
import os
import struct
import subprocess

import numpy as np
import pytest

import gen_synth
import oracle
from common import assert_bits_equal
from of_dis_amd.params import oppoint

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "of_dis_amd", "lib")
EXE = {1: os.path.join(ROOT, "of_dis_amd", "lib", "run_OF_INT"), 3: os.path.join(ROOT, "of_dis_amd", "lib", "run_OF_RGB")}


def read_flo(path):
    with open(path, "rb") as f:
        assert f.read(4) == b"PIEH"
        w, h = struct.unpack("<ii", f.read(8))
        d = np.frombuffer(f.read(), np.float32)
    assert d.size == 2 * w * h
    return d.reshape(h, w, 2)


def test_executables_exist_and_print_usage():
    for exe in EXE.values():
        assert os.path.exists(exe), f"{exe} missing: python -m of_dis_amd.build"
        r = subprocess.run([exe], capture_output=True, text=True)
        assert r.returncode == 2 and "usage" in r.stderr


def _oracle_flo(ia, ib, p, w, h):
    O = oracle.c_oracle()
    O.set_reduce_order(True)
    pa, pb = O.build_pyramid(p, ia), O.build_pyramid(p, ib)
    return O.upsample_crop(p, O.flow(p, pa[0], pa[1], pa[2], pb[0]), w, h)


@pytest.mark.gpu
def test_run_of_int_three_variants(gpu, tmp_path):
    w, h = 1024, 436     # needs padding to 1024x448: exercises the pad/crop path
    ia, ib, _ = gen_synth.make_pair(w, h, 31)
    fa, fb = str(tmp_path / "a.pgm"), str(tmp_path / "b.pgm")
    gen_synth.write_pgm(fa, ia)
    gen_synth.write_pgm(fb, ib)
    outs = []
    variants = [[], ["2"], "5 3 12 12 0.05 0.95 0 8 0.40 0 1 0 1 10 10 5 1 3 1.6 2".split()]
    for k, extra in enumerate(variants):
        fo = str(tmp_path / f"o{k}.flo")
        r = subprocess.run([EXE[1], fa, fb, fo] + extra, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        # reference stdout contract at verbosity 2 (run_dense.cpp:318,351,427; oflow.cpp:179,303,359)
        for tag in ("TIME (Image loading     ) (ms):", "TIME (Pyramide+Gradients) (ms):", "TIME (Grid Memo. Alloc. ) (ms):",
                    "TIME (Sc: 5, #p:    32, pconst, pinit, poptim, cflow, tvopt, total):",
                    "TIME (Sc: 3, #p:   448,", "TIME (O.Flow Run-Time   ) (ms):", "TIME (Saving flow file  ) (ms):"):
            assert tag in r.stdout, (tag, r.stdout)
        outs.append(read_flo(fo))
    assert outs[0].shape == (h, w, 2)
    assert_bits_equal(outs[1], outs[0], "variant 2 == variant 1")
    assert_bits_equal(outs[2], outs[0], "variant 3 == variant 1 (README.md:51-67)")
    assert_bits_equal(outs[0], _oracle_flo(ia, ib, oppoint(2, w, h), w, h), ".flo vs oracle pipeline")


@pytest.mark.gpu
def test_run_of_rgb_and_png(gpu, tmp_path):
    w, h = 320, 240
    ia, ib, _ = gen_synth.make_pair(w, h, 32, channels=3)   # arrays are in the order the binary sees: B,G,R
    fa, fb = str(tmp_path / "a.ppm"), str(tmp_path / "b.ppm")
    gen_synth.write_pgm(fa, ia[..., ::-1])                  # PPM stores R,G,B
    gen_synth.write_pgm(fb, ib[..., ::-1])
    fo = str(tmp_path / "o.flo")
    args = "3 1 8 8 0.05 0.95 0 12 0.75 0 1 1 1 10 10 5 1 3 1.6 0".split()
    r = subprocess.run([EXE[3], fa, fb, fo] + args, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout == ""                                   # verbosity 0
    p = oppoint(3, w, h, noc=3).copy(sc_f=3, sc_l=1, max_iter=8, min_iter=8, costfct=1)
    p.width, p.height = 320, 240
    ref = _oracle_flo(ia, ib, p, w, h)
    assert_bits_equal(read_flo(fo), ref, "rgb .flo vs oracle pipeline")
    try:
        from PIL import Image
    except Exception:
        return
    pa, pb = str(tmp_path / "a.png"), str(tmp_path / "b.png")
    Image.fromarray(ia[..., ::-1].copy()).save(pa)
    Image.fromarray(ib[..., ::-1].copy()).save(pb)
    fo2 = str(tmp_path / "o2.flo")
    r = subprocess.run([EXE[3], pa, pb, fo2] + args, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert_bits_equal(read_flo(fo2), ref, "PNG input == PPM input")


@pytest.mark.gpu
def test_run_of_int_forward_backward(gpu, tmp_path):
    """CLI parameter 10 (usefbcon, README.md:64): the binary builds the second image's gradient pyramid as well and the
    result equals the reference core run on the oracle's pyramids, upsampled by the oracle."""
    if oracle.need_ref("int", True) is None:
        pytest.skip("comparison against the compiled reference skipped: neither /root/reference nor oracle/_ref exists here")
    w, h = 640, 480
    ia, ib, _ = gen_synth.make_pair(w, h, 33)
    fa, fb, fo = str(tmp_path / "a.pgm"), str(tmp_path / "b.pgm"), str(tmp_path / "o.flo")
    gen_synth.write_pgm(fa, ia)
    gen_synth.write_pgm(fb, ib)
    args = "5 3 12 12 0.05 0.95 0 8 0.40 1 1 0 1 10 10 5 1 3 1.6 0".split()
    r = subprocess.run([EXE[1], fa, fb, fo] + args, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    p = oppoint(2, w, h).copy(usefbcon=1, sc_f=5, sc_l=3)
    O = oracle.c_oracle()
    pa, pb = O.build_pyramid(p, ia), O.build_pyramid(p, ib)
    core = oracle.ref("int", True).flow(p, pa[0], pa[1], pa[2], pb[0], pyr_b_dx=pb[1], pyr_b_dy=pb[2])
    assert_bits_equal(read_flo(fo), O.upsample_crop(p, core, w, h), "usefbcon .flo vs reference core + oracle pipeline")


def read_pfm(path):
    with open(path, "rb") as f:
        assert f.readline() == b"Pf\n"
        w, h = map(int, f.readline().split())
        assert f.readline() == b"-1.000000\n"            # fprintf("%f", -1.0f): little endian (run_dense.cpp:70)
        d = np.frombuffer(f.read(), np.float32)
    assert d.size == w * h
    return -d.reshape(h, w)[::-1]                         # rows bottom-up, values negated (run_dense.cpp:72-79)


@pytest.mark.gpu
def test_run_de_int_stereo_binary(gpu, tmp_path):
    """run_DE_INT (the reference's SELECTMODE=2 binary): .pfm of the horizontal displacement at full resolution."""
    if oracle.need_ref("de_int", True) is None:
        pytest.skip("comparison against the compiled reference skipped: neither /root/reference nor oracle/_ref exists here")
    w, h = 1024, 436
    ia, ib, _ = gen_synth.make_pair(w, h, 34)
    fa, fb, fo = str(tmp_path / "a.pgm"), str(tmp_path / "b.pgm"), str(tmp_path / "o.pfm")
    gen_synth.write_pgm(fa, ib)                           # second image first: negative horizontal motion
    gen_synth.write_pgm(fb, ia)
    r = subprocess.run([os.path.join(ROOT, "of_dis_amd", "lib", "run_DE_INT"), fa, fb, fo, "2"], capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    p = oppoint(2, w, h).copy(selectmode=2)
    O = oracle.c_oracle()
    pa, pb = O.build_pyramid(p, ib), O.build_pyramid(p, ia)
    core = oracle.ref("de_int", True).flow(p, pa[0], pa[1], pa[2], pb[0])
    assert core.min() < -0.5                              # a real displacement field, not all clamped to zero
    two = np.concatenate([core, core], axis=-1)           # the oracle's resize is per channel
    expect = O.upsample_crop(p.copy(selectmode=0), two, w, h)[..., 0]
    assert_bits_equal(read_pfm(fo), expect, "run_DE_INT .pfm vs reference core + oracle upsample")


def _write_case(path, p, pa, pb, expect, init=None):
    import ctypes as C
    import struct
    with open(path, "wb") as f:
        f.write(struct.pack("<ii", 0x4F464449, C.sizeof(p)))
        f.write(bytes(p))
        f.write(struct.pack("<i", 1 if init is not None else 0))
        for l in range(p.sc_l, p.sc_f + 1):
            for pl in (pa[0][l], pa[1][l], pa[2][l], pb[0][l]):
                f.write(np.ascontiguousarray(pl, np.float32).tobytes())
        if init is not None:
            f.write(np.ascontiguousarray(init, np.float32).tobytes())
        f.write(np.ascontiguousarray(expect, np.float32).tobytes())


def test_c_program_is_built():
    assert os.access(os.path.join(LIB, "dropin_test"), os.X_OK)
    r = subprocess.run([os.path.join(LIB, "dropin_test")], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr


@pytest.mark.gpu
def test_c_program_through_the_abi(gpu, tmp_path):
    """tests/c/dropin_test.c: a C99 program that includes include/ofdis.h and calls ofdis_flow() the way the reference's
    run_dense.cpp would call the constructor, on pyramids built by the oracle; the bytes of its result must be the
    oracle's (gray with TV, RGB, and a warm start through the cached context)."""
    import oracle
    from common import synth_case
    O = oracle.c_oracle()
    O.set_reduce_order(True)
    cases = []
    p, pa, pb, _, _ = synth_case(1024, 436, 2100, 1, 2, 1)
    cases.append(("gray", p, pa, pb, None))
    w, h = p.level_size(p.sc_f)
    init = (np.random.default_rng(9).standard_normal((h // 2, w // 2, 2)) * 0.5).astype(np.float32)
    cases.append(("warm", p, pa, pb, init))
    p3, qa, qb, _, _ = synth_case(320, 240, 2101, 3, 3, 1)
    cases.append(("rgb", p3, qa, qb, None))
    for name, pp, xa, xb, ini in cases:
        expect = O.flow(pp, xa[0], xa[1], xa[2], xb[0], initflow=ini)
        path = str(tmp_path / f"{name}.bin")
        _write_case(path, pp, xa, xb, expect, ini)
        r = subprocess.run([os.path.join(LIB, "dropin_test"), path, "5"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (name, r.stdout, r.stderr)
        assert "OK" in r.stdout and "identical" in r.stdout, r.stdout


# ------------------------------------------------------------------------------------------------ sequence driver
SEQ = {1: os.path.join(LIB, "run_OF_INT_seq"), 3: os.path.join(LIB, "run_OF_RGB_seq")}


def test_sequence_driver_usage_and_bad_lists(tmp_path):
    for exe in SEQ.values():
        assert os.path.exists(exe), f"{exe} missing: python -m of_dis_amd.build"
        r = subprocess.run([exe], capture_output=True, text=True)
        assert r.returncode == 2 and "pairs.txt" in r.stderr
    lst = tmp_path / "bad.txt"
    lst.write_text("only_two fields\n")
    r = subprocess.run([SEQ[1], str(lst)], capture_output=True, text=True)
    assert r.returncode == 2 and "expected" in r.stderr
    lst.write_text("# nothing\n\n")
    r = subprocess.run([SEQ[1], str(lst)], capture_output=True, text=True)
    assert r.returncode == 1 and "no pairs" in r.stderr
    r = subprocess.run([SEQ[1], str(tmp_path / "missing.txt")], capture_output=True, text=True)
    assert r.returncode == 1 and "cannot read" in r.stderr


@pytest.mark.parametrize("total,shares", [(512, 8), (513, 8), (7, 8), (66, 2), (100, 3), (1, 1)])
def test_sequence_driver_partition_is_frame_range(tmp_path, total, shares):
    """the C++ driver cuts its list exactly as of_dis_amd.shard.frame_range cuts bench.py's batch (contiguous
    shares, sizes differing by at most one, earlier shares take the remainder) -- checked without a GPU (--dry-run)."""
    from of_dis_amd.shard import frame_range
    lst = tmp_path / "pairs.txt"
    lst.write_text("".join(f"a{k}.pgm b{k}.pgm o{k}.flo\n" for k in range(total)))
    r = subprocess.run([SEQ[1], str(lst), "--gpus", str(shares), "--dry-run", "1"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = [tuple(map(int, __import__("re").match(r"share (\d+): device (\d+) pairs (-?\d+)\.\.(-?\d+)", l).groups()))
           for l in r.stdout.splitlines()]
    assert len(got) == shares
    covered = []
    for rank, (sr, dev, lo, hi) in enumerate(got):
        assert (sr, dev) == (rank, rank)
        assert (lo, hi + 1) == frame_range(total, rank, shares)
        covered += list(range(lo, hi + 1))
    assert covered == list(range(total))


def _write_pairs(tmp_path, n, w, h, channels=1, seed0=500):
    pairs = []
    for k in range(n):
        ia, ib, _ = gen_synth.make_pair(w, h, seed0 + k, channels=channels)
        ext = "pgm" if channels == 1 else "ppm"
        fa, fb = str(tmp_path / f"a{k:03d}.{ext}"), str(tmp_path / f"b{k:03d}.{ext}")
        gen_synth.write_pgm(fa, ia if channels == 1 else ia[..., ::-1])
        gen_synth.write_pgm(fb, ib if channels == 1 else ib[..., ::-1])
        pairs.append((fa, fb))
    return pairs


@pytest.mark.gpu
def test_sequence_driver_matches_the_single_pair_binary(gpu, tmp_path):
    """SURVEY 8(e) in the host language of the reference: run_OF_INT_seq over 66 pairs (a chunk size that does not divide the
    share: the last chunk is short) writes, pair for pair, the BYTES the single-pair run_OF_INT writes (exact contract: the
    chunk context runs other kernel mappings than the one-pair context, same bits); the split over two shares (two host
    threads, two sets of contexts: on DISTINCT devices whenever the box has two -- the shares' PCI bus ids must then differ --
    else --devices 0,0, both on the one GPU of the test box), another chunk size and other numbers of chunks in flight
    (--depth 1 / 2 / 3: the device stage's slots, include/ofdis.h version 3) give the same files."""
    import re
    w, h, n = 320, 192, 66
    pairs = _write_pairs(tmp_path, n, w, h)
    single = []
    for k, (fa, fb) in enumerate(pairs):
        fo = str(tmp_path / f"single{k:03d}.flo")
        r = subprocess.run([EXE[1], fa, fb, fo] + "5 3 12 12 0.05 0.95 0 8 0.40 0 1 0 1 10 10 5 1 3 1.6 0".split(),
                           capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        single.append(open(fo, "rb").read())
    args = "5 3 12 12 0.05 0.95 0 8 0.40 0 1 0 1 10 10 5 1 3 1.6 1".split()
    two_gpus = gpu.lib().ofdis_device_count() >= 2
    devs = "0,1" if two_gpus else "0,0"
    for tag, opts in (("one", ["--chunk", "32"]), ("two", ["--devices", devs, "--chunk", "20", "--depth", "3"]),
                      ("d1", ["--chunk", "16", "--depth", "1"])):
        lst = tmp_path / f"{tag}.txt"
        lst.write_text("# pairs of the test\n" + "".join(f"{fa} {fb} {tmp_path}/{tag}{k:03d}.flo\n" for k, (fa, fb) in enumerate(pairs)))
        r = subprocess.run([SEQ[1], str(lst)] + opts + args[:-1] + ["2"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (r.stdout, r.stderr)
        assert f"TIME ({n} pairs on {2 if tag == 'two' else 1} device share(s)" in r.stdout
        ids = re.findall(r"TIME \(share \d+: device \d+ \[([0-9a-fA-F:.]+)\]", r.stdout)
        assert len(ids) == (2 if tag == "two" else 1), r.stdout
        if tag == "two" and two_gpus:  # one GPU per share, really
            assert len(set(ids)) == 2, f"two shares on {ids}: not two devices"
        for k in range(n):
            got = open(tmp_path / f"{tag}{k:03d}.flo", "rb").read()
            assert got == single[k], f"{tag}: pair {k} differs from the single-pair binary's .flo"
    # and against the oracle's pipeline, so that "the same" is also "right" (first and last pair)
    p = oppoint(2, w, h).copy(sc_f=5, sc_l=3)
    for k in (0, n - 1):
        ia, ib, _ = gen_synth.make_pair(w, h, 500 + k)
        assert_bits_equal(read_flo(tmp_path / f"one{k:03d}.flo"), _oracle_flo(ia, ib, p, w, h), f"pair {k} vs oracle pipeline")


@pytest.mark.gpu
@pytest.mark.parametrize("channels,opp", [(1, "2"), (3, "3")])
def test_stereo_sequence_driver_matches_the_single_pair_binary(gpu, tmp_path, channels, opp):
    """run_DE_INT_seq / run_DE_RGB_seq (the sequence driver compiled for the reference's SELECTMODE=2): every .pfm is, byte for
    byte, the file the single-pair run_DE_* writes for that pair -- over two shares and a chunk size that does not divide them."""
    w, h, n = 320, 192, 11
    pairs = [(fb, fa) for fa, fb in _write_pairs(tmp_path, n, w, h, channels=channels, seed0=900)]  # negative horizontal motion
    kind = "INT" if channels == 1 else "RGB"
    single = []
    for k, (fa, fb) in enumerate(pairs):
        fo = str(tmp_path / f"single{k:03d}.pfm")
        r = subprocess.run([os.path.join(LIB, f"run_DE_{kind}"), fa, fb, fo, opp], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        single.append(open(fo, "rb").read())
    assert len(set(single)) > 1
    lst = tmp_path / "de.txt"
    lst.write_text("".join(f"{fa} {fb} {tmp_path}/seq{k:03d}.pfm\n" for k, (fa, fb) in enumerate(pairs)))
    r = subprocess.run([os.path.join(LIB, f"run_DE_{kind}_seq"), str(lst), "--devices", "0,0", "--chunk", "4", opp],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    for k in range(n):
        assert open(tmp_path / f"seq{k:03d}.pfm", "rb").read() == single[k], f"pair {k} differs from run_DE_{kind}'s .pfm"


@pytest.mark.gpu
def test_sequence_driver_rgb_and_unreadable_pairs(gpu, tmp_path):
    """run_OF_RGB_seq (operating point by number); a pair whose image is missing or has another size is reported, the others
    are still written, and the exit status says that not everything went through."""
    w, h, n = 256, 128, 5
    pairs = _write_pairs(tmp_path, n, w, h, channels=3, seed0=700)
    ia, ib, _ = gen_synth.make_pair(128, 128, 800, channels=3)
    gen_synth.write_pgm(str(tmp_path / "odd_a.ppm"), ia[..., ::-1])
    gen_synth.write_pgm(str(tmp_path / "odd_b.ppm"), ib[..., ::-1])
    lines = [f"{fa} {fb} {tmp_path}/o{k}.flo" for k, (fa, fb) in enumerate(pairs)]
    lines.insert(2, f"{tmp_path}/nope.ppm {pairs[0][1]} {tmp_path}/nope.flo")
    lines.insert(4, f"{tmp_path}/odd_a.ppm {tmp_path}/odd_b.ppm {tmp_path}/odd.flo")
    lst = tmp_path / "rgb.txt"
    lst.write_text("\n".join(lines) + "\n")
    r = subprocess.run([SEQ[3], str(lst), "--chunk", "3", "3"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 1, (r.stdout, r.stderr)
    assert "nope.ppm" in r.stderr and "like the first readable pair" in r.stderr
    assert not os.path.exists(tmp_path / "nope.flo") and not os.path.exists(tmp_path / "odd.flo")
    for k, (fa, fb) in enumerate(pairs):
        fo = str(tmp_path / f"s{k}.flo")
        r1 = subprocess.run([EXE[3], fa, fb, fo, "3"], capture_output=True, text=True, timeout=120)
        assert r1.returncode == 0, r1.stderr
        assert open(fo, "rb").read() == open(tmp_path / f"o{k}.flo", "rb").read(), f"rgb pair {k}"
    # an unreadable FIRST pair does not end the run (the geometry comes from the first readable image, or from --size), and
    # the count of failed pairs is the number of .flo files that were not written
    for k in range(n):
        os.remove(tmp_path / f"o{k}.flo")
    for extra in ([], ["--size", str(w), str(h)]):
        lst.write_text("\n".join([lines[2]] + lines[:2] + lines[3:]) + "\n")
        r = subprocess.run([SEQ[3], str(lst), "--chunk", "3"] + extra + ["3"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 1, (r.stdout, r.stderr)
        assert "nope.ppm" in r.stderr
        for k in range(n):
            assert open(tmp_path / f"s{k}.flo", "rb").read() == open(tmp_path / f"o{k}.flo", "rb").read(), f"rgb pair {k} ({extra})"
            os.remove(tmp_path / f"o{k}.flo")


@pytest.mark.gpu
def test_rgb_sequence_driver_chunks_take_the_fused_tv_kernel(gpu, tmp_path):
    """run_OF_RGB_seq with its RGB levels on the fused system + SOR kernel (OFDIS_FUSED_RGB_MIN=1: under the CLI's default,
    exact, contract only contexts of 512 frames and more take it by themselves) against the single-pair binary's per-stage
    kernels: the .flo files are the same bytes (default operating point, TV on; a full chunk of 16 and a short last one)."""
    w, h, n = 320, 192, 19
    pairs = _write_pairs(tmp_path, n, w, h, channels=3, seed0=950)
    lst = tmp_path / "rgb16.txt"
    lst.write_text("".join(f"{fa} {fb} {tmp_path}/q{k:02d}.flo\n" for k, (fa, fb) in enumerate(pairs)))
    r = subprocess.run([SEQ[3], str(lst), "2"], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, OFDIS_FUSED_RGB_MIN="1"))
    assert r.returncode == 0, (r.stdout, r.stderr)
    for k in (0, 7, 15, 16, 18):
        fa, fb = pairs[k]
        fo = str(tmp_path / f"one{k:02d}.flo")
        r1 = subprocess.run([EXE[3], fa, fb, fo, "2"], capture_output=True, text=True, timeout=120)
        assert r1.returncode == 0, r1.stderr
        assert open(fo, "rb").read() == open(tmp_path / f"q{k:02d}.flo", "rb").read(), f"rgb pair {k}"
