"""CPU: the C-ABI library loads and exports every symbol include/ofdis.h declares; host-side parameter
logic (operating points, geometry) matches the reference's derivations.  No compute calls (no GPU here)."""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest

from of_dis_amd import capi
from of_dis_amd.params import OfdisParams, auto_first_scale, oppoint, padded_size

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "ofdis.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ofdis_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    assert _declared_functions() == sorted(capi.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol():
    L = capi.lib()
    for name in _declared_functions():
        assert hasattr(L, name), f"{name} declared in include/ofdis.h but not exported"
    hdr = open(os.path.join(ROOT, "include", "ofdis.h")).read()
    assert L.ofdis_version() == capi.OFDIS_VERSION == int(re.search(r"#define OFDIS_VERSION (\d+)", hdr).group(1))


def test_build_id_is_the_hash_of_the_sources():
    """ofdis_build_id(): the library says which kernel sources + compiler flags it was built from (of_dis_amd.build.source_id);
    profiles/traffic_*.json carries the id of the library its counters were collected on and bench.py attaches it only to that
    library.  Here: the built library's id is the id of the tree it sits in (i.e. the build is not stale), and an id changes
    when a flag does."""
    from of_dis_amd import build
    assert capi.build_id() == build.source_id(), "libofdis_hip.so is stale: python -m of_dis_amd.build"
    assert build.source_id(extra_flags={"ofdis_dis.hip": ["-DX"]}) != build.source_id()
    import json
    for name in ("traffic_fused.json", "traffic_exact.json"):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            tj = json.load(open(path))
            assert "bytes_per_step" in tj and isinstance(tj.get("build_id", ""), str)


def test_binding_refuses_a_library_of_another_abi_version(monkeypatch):
    """The struct layouts of the binding belong to one ABI version: a stale libofdis_hip.so must not be handed our structs."""
    monkeypatch.setattr(capi, "_lib", None)
    monkeypatch.setattr(capi, "OFDIS_VERSION", capi.OFDIS_VERSION + 1)
    with pytest.raises(capi.OfdisError, match="ABI version"):
        capi.lib()


def test_library_exports_nothing_but_the_header():
    """The shipped library's dynamic symbol table is exactly include/ofdis.h: no test hooks, no kernel handles, no C++
    internals (of_dis_amd/build.py links with a version script generated from the header; the parity tests' helper
    kernels live in the separate libofdis_testhooks.so)."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", capi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    assert exported == _declared_functions(), sorted(set(exported) ^ set(_declared_functions()))


def test_tuning_struct_layout_matches_header():
    src = open(os.path.join(ROOT, "include", "ofdis.h")).read()
    body = re.search(r"typedef struct ofdis_tuning \{(.*?)\} ofdis_tuning;", src, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = [d.split()[-1] for d in body.split(";") if d.strip()]
    assert names == [n for n, _ in capi.OfdisTuning._fields_]
    t = capi.get_tuning()   # host only: read from the environment, no device work
    assert (t.gray8, t.fused_tv, t.rgb12_lpp, t.contract, t.fused_tp_pipe) == (1, 1, 0, 0, 1)


def test_tuning_environment_variables_match_header():
    """Every knob of ofdis_tuning names the environment variable that initialises it; the library must read exactly those
    (and only in its one-time initialisation: no getenv on a launch path)."""
    src = open(os.path.join(ROOT, "include", "ofdis.h")).read()
    body = re.search(r"typedef struct ofdis_tuning \{(.*?)\} ofdis_tuning;", src, re.S).group(1)
    documented = set(re.findall(r"OFDIS_[A-Z0-9_]+", body))
    csrc = os.path.join(ROOT, "of_dis_amd", "csrc")
    capi_src = open(os.path.join(csrc, "ofdis_capi.hip")).read()
    init = capi_src[capi_src.index("void tuning_init_locked()"):capi_src.index("ofdis_tuning tuning(unsigned* epoch)")]
    assert set(re.findall(r'"(OFDIS_[A-Z0-9_]+)"', init)) == documented
    for name in os.listdir(csrc):
        if name.endswith((".hip", ".h")):
            text = open(os.path.join(csrc, name)).read()
            if name == "ofdis_capi.hip":
                text = text.replace(init, "")
            code = re.sub(r"//[^\n]*", "", text)  # (comments may mention it)
            # the one exception: the scratch-poisoning TEST HOOK, read where a context allocates its memory (never on a launch
            # path; the tests switch it on and off while the library is loaded)
            code = code.replace('getenv("OFDIS_POISON_SCRATCH")', "")
            assert "getenv(" not in code, f"{name} reads the environment outside the tuning initialisation"


def test_params_struct_layout_matches_header():
    src = open(os.path.join(ROOT, "include", "ofdis.h")).read()
    body = re.search(r"typedef struct ofdis_params \{(.*?)\} ofdis_params;", src, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        typ, names = decl.split(None, 1)
        fields += [(n.strip(), typ) for n in names.split(",")]
    assert [n for n, _ in fields] == [n for n, _ in OfdisParams._fields_]
    for (n, typ), (_, ct) in zip(fields, OfdisParams._fields_):
        assert (typ == "float") == (ct is C.c_float), n
    assert C.sizeof(OfdisParams) == 4 * len(fields)


@pytest.mark.parametrize("op", [1, 2, 3, 4])
@pytest.mark.parametrize("width", [1024, 640, 1920, 320])
def test_oppoint_table_matches_c_abi(op, width):
    """ofdis_params_oppoint (C++) and params.oppoint (Python) both restate run_dense.cpp:225-265."""
    q = OfdisParams()
    capi.check(capi.lib().ofdis_params_oppoint(C.byref(q), op, width, 1))
    p = oppoint(op, width, 436, verbosity=2)
    for name in ("sc_f", "sc_l", "max_iter", "min_iter", "p_samp_s", "usetvref", "imgpadding", "costfct", "patnorm",
                 "usefbcon", "tv_innerit", "tv_solverit", "verbosity", "noc"):
        assert getattr(p, name) == getattr(q, name), name
    for name in ("dp_thresh", "dr_thresh", "res_thresh", "patove", "tv_alpha", "tv_gamma", "tv_delta", "tv_sor"):
        assert np.float32(getattr(p, name)) == np.float32(getattr(q, name)), name


def test_readme_operating_point_2():
    """README.md:51-67: op-point 2 == `5 3 12 12 0.05 0.95 0 8 0.40 0 1 0 1 10 10 5 1 3 1.6 2` at width 1024."""
    p = oppoint(2, 1024, 436, verbosity=2)
    assert (p.sc_f, p.sc_l, p.max_iter, p.min_iter) == (5, 3, 12, 12)
    assert (p.p_samp_s, p.usefbcon, p.patnorm, p.costfct, p.usetvref) == (8, 0, 1, 0, 1)
    assert (p.tv_innerit, p.tv_solverit, p.verbosity) == (1, 3, 2)
    assert np.float32(p.patove) == np.float32(0.4) and np.float32(p.tv_sor) == np.float32(1.6)
    assert (p.width, p.height) == (1024, 448)
    assert auto_first_scale(1024, 5, 8) == 5


def test_geometry_matches_survey_table():
    """level sizes and patch counts of the BASELINE configs."""
    p = oppoint(2, 1024, 436)
    assert [p.level_size(l) for l in (5, 4, 3)] == [(32, 14), (64, 28), (128, 56)]
    assert p.steps == 4
    assert [p.grid(l)[0] * p.grid(l)[1] for l in (5, 4, 3)] == [32, 112, 448]
    p = oppoint(2, 640, 480)
    assert (p.width, p.height) == (640, 480)
    assert [p.level_size(l) for l in (5, 4, 3)] == [(20, 15), (40, 30), (80, 60)]
    assert [p.grid(l)[0] * p.grid(l)[1] for l in (5, 4, 3)] == [20, 80, 300]
    assert oppoint(1, 1024, 436).steps == 5          # 8*(1-0.3f) = 5.6 -> 5
    assert oppoint(4, 1920, 1080, noc=3).steps == 3  # 12*(1-0.75f) = 3
    assert padded_size(1920, 1080, 6) == (1920, 1088)


def test_status_codes_without_a_gpu():
    """Argument validation happens before any device work, so it is testable here."""
    L = capi.lib()
    h = C.c_void_p()
    p = oppoint(2, 1024, 436)
    for bad, code in ((p.copy(width=1000), -1), (p.copy(imgpadding=4), -1),
                      (p.copy(noc=2), -1), (p.copy(sc_l=6), -1), (p.copy(costfct=10), -2)):
        rc = L.ofdis_batch_create(C.byref(h), C.byref(bad), 4)
        assert rc == code, (bad.as_dict(), rc, L.ofdis_last_error())
        assert L.ofdis_last_error()
    assert L.ofdis_batch_create(C.byref(h), C.byref(p), 0) == -1


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(capi, "_lib", None)
    monkeypatch.setattr(capi, "LIB_PATH", "/nonexistent/libofdis_hip.so")
    with pytest.raises(capi.OfdisError):
        capi.lib()


def test_outlier_threshold_is_the_exact_square_root_boundary():
    """The patch kernels test ||d||^2 > X instead of sqrt(||d||^2) > t (patch.cpp:199): X must be the largest float
    whose correctly rounded square root is <= t, for every patch size's t = P/2 and for awkward values."""
    import numpy as np
    from common import testhooks
    L = testhooks()
    f32 = np.float32
    for t in [1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 13.0, 0.1, 0.7, 1e-3, 123.456, 3.4e18, 1e-19, 0.0]:
        t = f32(t)
        X = f32(L.ofdis_test_outlier_sq(t))
        up = np.nextafter(X, f32(np.inf))
        assert np.sqrt(X) <= t, (t, X)
        assert np.sqrt(up) > t, (t, X, up)
    assert np.isnan(L.ofdis_test_outlier_sq(float("nan")))
