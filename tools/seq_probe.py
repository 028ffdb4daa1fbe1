"""Developer helper: what run_OF_INT_seq achieves on a list of 1024x436 pairs (decode + device + .flo writing), one GPU.
    python tools/seq_probe.py [npairs] [chunk] [depths=1,2,3]     -> prints the driver's TIME lines per --depth
The per-share line is the DEVICE stage alone (upload + pyramids + path + upsample + download of the full-resolution flow, chunks
overlapping at --depth > 1): pairs / that time is the rate VERDICT r05 item 4 asks for (>= 12 k pairs/s; link bound 15 k)."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
chunk = sys.argv[2] if len(sys.argv) > 2 else "64"
depths = (sys.argv[3] if len(sys.argv) > 3 else "1,2,3").split(",")
tmp = tempfile.mkdtemp(dir="/tmp")
distinct = 32
for k in range(distinct):
    ia, ib, _ = gen_synth.make_pair(1024, 436, 100 + k)
    gen_synth.write_pgm(f"{tmp}/a{k}.pgm", ia)
    gen_synth.write_pgm(f"{tmp}/b{k}.pgm", ib)
with open(f"{tmp}/pairs.txt", "w") as f:
    for i in range(n):
        f.write(f"{tmp}/a{i % distinct}.pgm {tmp}/b{i % distinct}.pgm {tmp}/o{i}.flo\n")
exe = os.path.join(ROOT, "of_dis_amd", "lib", "run_OF_INT_seq")
args = "5 3 12 12 0.05 0.95 0 8 0.40 0 1 0 1 10 10 5 1 3 1.6 2".split()
import re
# the device stage alone (no decoding, no .flo): the first chunk pushed through it 64 times
for ck in sorted({chunk, "16", "64"}, key=int):
    for depth in depths:
        r = subprocess.run([exe, f"{tmp}/pairs.txt", "--chunk", ck, "--depth", depth, "--device-bench", str(max(8, 8192 // int(ck)))] + args[:-1] + ["0"],
                           capture_output=True, text=True)
        print(f"chunk {ck} --depth {depth}:", r.stdout.strip(), r.stderr.strip()[:200])
for depth in depths:
    for rep in range(2):
        r = subprocess.run([exe, f"{tmp}/pairs.txt", "--chunk", chunk, "--depth", depth] + args, capture_output=True, text=True)
        print(f"--depth {depth}:", r.stdout.strip(), r.stderr.strip()[:200])
        m = re.search(r"in flight\) \(ms\): *([0-9.e+]+)", r.stdout)
        if m:
            print(f"    device stage: {n / (float(m.group(1)) * 1e-3):.0f} pairs/s")
subprocess.run(["rm", "-rf", tmp])
