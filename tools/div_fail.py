"""Developer helper: print the operands for which the fused TV kernel's quotient (rcp_newton) differs from a / b."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from of_dis_amd import capi as gpu
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
n = 1 << 24
def draw(lo, hi):
    m = rng.integers(0, 1 << 23, n, dtype=np.uint32)
    e = rng.integers(127 + lo, 127 + hi, n, dtype=np.uint32)
    sgn = rng.integers(0, 2, n, dtype=np.uint32)
    return ((sgn << 31) | (e << 23) | m).view(np.float32)
tot = 0
for rep in range(4):
    a, b = draw(-40, 40), draw(-40, 40)
    got = gpu.div_sqrt_test(a, b)
    q = (a / b).astype(np.float32)
    bad = got[4] != q
    rr = (np.float32(1) / b).astype(np.float32)
    offr = got[5] != rr
    bad0 = got[0] != q
    tot += n
    sq = np.sqrt(np.abs(a)).astype(np.float32)
    print("   sqrt_newton mismatches", int((got[6] != sq).sum()), " b / sqrt(|a|) mismatches", int((got[7] != (b / sq).astype(np.float32)).sum()))
    print("rep", rep, "quotient mismatches", int(bad.sum()), "(div_rn:", int(bad0.sum()), ") reciprocal not correctly rounded:", int(offr.sum()),
          "of", n, "; mismatching quotients whose reciprocal IS correctly rounded:", int((bad & ~offr).sum()))
    for i in np.nonzero(bad)[0][:6]:
        print("   a=%r (%08x) b=%r (%08x) got %08x want %08x  rcp %08x want %08x" % (a[i], a[i:i+1].view(np.uint32)[0], b[i], b[i:i+1].view(np.uint32)[0],
              got[4][i:i+1].view(np.uint32)[0], q[i:i+1].view(np.uint32)[0], got[5][i:i+1].view(np.uint32)[0], rr[i:i+1].view(np.uint32)[0]))
