"""Turn two rocprofv3 PMC passes (FETCH_SIZE; WRITE_SIZE) of `bench.py --steps 1 --warmup 0` into
profiles/traffic.json: HBM bytes per step per kernel class.

    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <batch> <tv> <nrun_steps>
                                [<sq_counter_collection.csv>]

The optional third file is a pass with SQ_INSTS_VALU (+ GRBM_GUI_ACTIVE): wave64 VALU instructions issued per step per
kernel class (every instruction counts once) and the shader clock
sustained under the dominant kernel (GRBM_GUI_ACTIVE cycles / its kernel-trace duration), both read by bench.py for
`roofline_valu`.

Corrections (MI355X_MICROARCH.md "HBM", re-measured in profiles/r01_pmc_calibration.txt on known byte counts):
FETCH_SIZE is in KiB and reports exactly half of the bytes read (4 B/lane and 16 B/lane alike); WRITE_SIZE is in
KiB and exact.  bench.py runs `nrun_steps` = warmup + steps + 3 timing passes of the pipeline; counters are
summed over all dispatches of a kernel class and divided by that number.
"""
import csv
import json
import os
import sys
from collections import defaultdict

# kernel-name prefix -> kernel class of bench.py (tv_prep = warp + derivatives of the fused path, reported as "derivatives")
CLASS = {"warp": "warp", "derivatives": "derivatives", "tv_prep": "derivatives", "tv_system": "tv_system", "sor_": "sor",
         "tv_fused": "tv_fused", "patch_optimize": "patch_optimize", "densify": "densify", "tv_finish": "tv_finish"}


NLEVELS = 3  # operating point 2 at 1024x436: levels 5, 4, 3 -- a pass launches every class once per level, coarsest first
SC_F = 5
LEVELS = {}   # counter -> class -> level -> value summed over the passes


def collect(path, counter, scale):
    """Sum of `counter` per kernel class; also per pyramid LEVEL (LEVELS[counter]): the dispatches of a class in dispatch
    order are level 5, 4, 3, 5, 4, 3, ... (classes launched exactly once per level and pass)."""
    acc = defaultdict(float)
    per = defaultdict(dict)  # class -> dispatch id -> [sum over the rows of that dispatch, kernel name]
    for n_row, r in enumerate(csv.DictReader(open(path))):
        if r["Counter_Name"] != counter:
            continue
        name = r.get("Kernel_Name", "")
        for key, cls in CLASS.items():
            if any(("ofdis::" + ns + key) in name for ns in ("", "exact::", "fused::")):
                v = float(r["Counter_Value"]) * scale
                acc[cls] += v
                e = per[cls].setdefault(int(r.get("Dispatch_Id") or n_row), [0.0, name.split("(")[0].replace("void ", "")])
                e[0] += v
    lv = {}
    for cls, by_id in per.items():
        rows = [(i, v, k) for i, (v, k) in sorted(by_id.items())]
        if len(rows) % NLEVELS:
            continue
        lv[cls] = {}
        for i, (_, v, kname) in enumerate(rows):
            e = lv[cls].setdefault(str(SC_F - i % NLEVELS), {"sum": 0.0, "kernel": kname})
            e["sum"] += v
    LEVELS[counter] = lv
    return acc


fetch = collect(sys.argv[1], "FETCH_SIZE", 2 * 1024.0)
write = collect(sys.argv[2], "WRITE_SIZE", 1024.0)
batch, tv, nsteps = int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
sq_file = sys.argv[6] if len(sys.argv) > 6 and sys.argv[6] not in ("", "-") else None
contract = sys.argv[7] if len(sys.argv) > 7 else "exact"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from of_dis_amd import capi  # noqa: E402  (the library the counters were collected on: its build id goes into the file)

out = {"batch": batch, "tv": tv, "contract": contract, "build_id": capi.build_id(),
       "what": "one un-pipelined pass over `batch` pairs = ONE sub-batch of the headline run (bench.py --batch 2*batch --pipeline 2): "
               "same kernel selection and strip lengths; bench.py multiplies by the number of sub-batches",
       "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, FETCH x2 (calibrated)",
       "bytes_per_step": {k: (fetch[k] + write[k]) / nsteps for k in sorted(set(fetch) | set(write))},
       "read_bytes_per_step": {k: fetch[k] / nsteps for k in sorted(fetch)},
       "write_bytes_per_step": {k: write[k] / nsteps for k in sorted(write)}}
# the same per pyramid level (the classes launched once per level): bytes and, below, VALU instructions
per_level = {}
for cls in sorted(set(LEVELS.get("FETCH_SIZE", {})) & set(LEVELS.get("WRITE_SIZE", {}))):
    per_level[cls] = {l: {"kernel": LEVELS["FETCH_SIZE"][cls][l]["kernel"],
                          "bytes_per_step": (LEVELS["FETCH_SIZE"][cls][l]["sum"] + LEVELS["WRITE_SIZE"][cls][l]["sum"]) / nsteps}
                      for l in LEVELS["FETCH_SIZE"][cls] if l in LEVELS["WRITE_SIZE"][cls]}
out["per_level"] = per_level
if sq_file:
    valu = collect(sq_file, "SQ_INSTS_VALU", 1.0)
    out["valu_insts_per_step"] = {k: valu[k] / nsteps for k in sorted(valu)}
    for cls, lv in LEVELS.get("SQ_INSTS_VALU", {}).items():
        for l, e in lv.items():
            if cls in per_level and l in per_level[cls]:
                per_level[cls][l]["valu_insts_per_step"] = e["sum"] / nsteps
    # sustained shader clock under the dominant kernel: busy cycles / duration of the same dispatches
    cyc, dur = defaultdict(float), defaultdict(float)
    for r in csv.DictReader(open(sq_file)):
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
            continue
        for key, cls in CLASS.items():
            if any(("ofdis::" + ns + key) in r.get("Kernel_Name", "") for ns in ("", "exact::", "fused::")):
                cyc[cls] += float(r["Counter_Value"])
                if r.get("Start_Timestamp") and r.get("End_Timestamp"):
                    dur[cls] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    dom = max(valu, key=valu.get) if valu else None
    if dom and dur.get(dom):
        ghz = cyc[dom] / dur[dom] / 8.0  # the counter is reported summed over the 8 XCDs; timestamps are ns
        if 0.8 < ghz < 2.6:
            out["sustained_clock_ghz"] = round(ghz, 3)
            out["sustained_clock_kernel"] = dom
            out["sustained_clock_source"] = "GRBM_GUI_ACTIVE (sum over 8 XCDs) / 8 / dispatch duration, serialised PMC pass"
prof = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
json.dump(out, open(os.path.join(prof, "traffic_%s.json" % contract), "w"), indent=1)  # one file per arithmetic contract
if contract == "fused":  # ... and the default name for the contract the headline normally runs under
    json.dump(out, open(os.path.join(prof, "traffic.json"), "w"), indent=1)
print(json.dumps(out["bytes_per_step"], indent=1))
