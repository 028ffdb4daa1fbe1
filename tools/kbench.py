"""Developer helper: run bench.py with extra env vars / args and print a compact per-kernel table.

    python tools/kbench.py [--env K=V ...] [-- bench args]
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
env = dict(os.environ)
args = sys.argv[1:]
bargs = []
if "--" in args:
    i = args.index("--")
    bargs = args[i + 1:]
    args = args[:i]
tag = []
for a in args:
    if a.startswith("--env"):
        continue
    k, v = a.split("=", 1)
    env[k] = v
    tag.append(a)
r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-seconds", "0", "--no-parity"] + bargs,
                   env=env, capture_output=True, text=True)
line = [l for l in r.stdout.splitlines() if l.startswith("{")]
if not line:
    print("FAILED", " ".join(tag), r.stderr[-2000:])
    sys.exit(1)
d = json.loads(line[-1])
print(" ".join(tag), "| fps", d["value"], "ms/step", d["ms_per_step"], "|",
      " ".join(f"{k}={v['ms_per_step']:.3f}({v['achieved_GBs']:.0f})" for k, v in d["kernels"].items()))
