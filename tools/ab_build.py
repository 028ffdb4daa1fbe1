"""Developer helper: build a variant of libofdis_hip.so (patched sources and / or extra per-file hipcc flags) for A/B timing.

    python tools/ab_build.py NAME [file.hip:-flag[,-flag...] ...]      ->  of_dis_amd/lib/ab_NAME/libofdis_hip.so
    OFDIS_CSRC=/path/to/patched/csrc python tools/ab_build.py NAME      (a patched copy of of_dis_amd/csrc: timing-only
                                                                        experiments stay out of the tree)
    OFDIS_LIB=of_dis_amd/lib/ab_NAME/libofdis_hip.so python bench.py ...
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from of_dis_amd import build as B  # noqa: E402

csrc = os.environ.get("OFDIS_CSRC", B.CSRC)
name = sys.argv[1]
extra = {}
for a in sys.argv[2:]:
    f, fl = a.split(":", 1)
    extra[f] = [x for x in fl.split(",") if x]
out = os.path.join(B.LIBDIR, "ab_" + name)
os.makedirs(out, exist_ok=True)
objs, _ = B.compile_units(csrc, out, force=True, extra_flags=extra)
so = os.path.join(out, "libofdis_hip.so")
subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs)
print(so)
