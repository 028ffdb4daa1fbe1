"""Developer helper: build a variant of libofdis_hip.so with extra per-file hipcc flags for A/B timing.

    python tools/ab_build.py NAME file.hip:-flag[,-flag...] [...]      ->  of_dis_amd/lib/ab_NAME/libofdis_hip.so
    OFDIS_LIB=of_dis_amd/lib/ab_NAME/libofdis_hip.so python tools/kbench.py ...
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from of_dis_amd import build as B  # noqa: E402

CSRC = os.environ.get("OFDIS_CSRC", B.CSRC)  # a patched copy of the sources (timing-only experiments stay out of the tree)
name = sys.argv[1]
extra = {}
for a in sys.argv[2:]:
    f, fl = a.split(":", 1)
    extra[f] = [x for x in fl.split(",") if x]
out = os.path.join(B.LIBDIR, "ab_" + name)
os.makedirs(out, exist_ok=True)
objs = []
for src in B.HIP_SOURCES:
    src = src if src.endswith(".hip") else src + ".hip"
    obj = os.path.join(out, src.replace(".hip", ".o"))
    flags = [f for f in B.PER_FILE_FLAGS.get(src, [])] + extra.get(src, [])
    drop = [f[1:] for f in flags if f.startswith("!")]
    flags = [f for f in flags if not f.startswith("!") and f not in drop]
    subprocess.check_call([B._hipcc()] + B.HIPFLAGS + flags + ["-c", os.path.join(CSRC, src), "-o", obj])
    objs.append(obj)
so = os.path.join(out, "libofdis_hip.so")
subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs)
print(so)
