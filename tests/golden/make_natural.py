"""Generates tests/golden/natural.json: for every natural pair tests/natural.py can find here, the SHA-256 of the two input
frames as decoded (so that another decoder / package version is noticed and the entry skipped, not failed) and of the flow
the REFERENCE BUILD (oracle/_ref: the reference's own sources compiled in place, defined summation order) computes for it.
Run in the build container:  python tests/golden/make_natural.py"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)

import natural  # noqa: E402
import oracle  # noqa: E402
from test_natural import CASES, case_inputs  # noqa: E402


def main():
    out = {}
    for name, noc, opp, mode in CASES:
        got = case_inputs(name, noc, opp, mode)
        if got is None:
            print("missing:", name)
            continue
        p, ia, ib, pa, pb, _ = got
        kind = ("de_" if mode == 2 else "") + ("int" if noc == 1 else "rgb")
        R = oracle.need_ref(kind, True)
        if R is None:
            raise SystemExit("the reference build is needed (oracle/_ref): make -C oracle")
        ref = R.flow(p, pa[0], pa[1], pa[2], pb[0])
        out[f"{name}|{noc}|{opp}|{mode}"] = {"input_sha256": [natural.sha(ia), natural.sha(ib)], "shape": list(ref.shape),
                                            "flow_sha256": natural.sha(ref)}
        print(name, noc, opp, mode, ref.shape, out[f"{name}|{noc}|{opp}|{mode}"]["flow_sha256"][:16])
    with open(os.path.join(HERE, "natural.json"), "w") as f:
        json.dump({"made_by": "tests/golden/make_natural.py from the reference build (oracle/_ref, OFDIS_SHIM_WAVE64 order)",
                   "cases": out}, f, indent=1, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    main()
