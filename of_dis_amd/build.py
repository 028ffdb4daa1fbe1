"""Build the product: libofdis_hip.so (HIP kernels + C ABI, gfx950) and the run_OF_* executables.

    python -m of_dis_amd.build          # incremental
    python -m of_dis_amd.build --force

Everything is compiled in-tree (of_dis_amd/lib/) so that the built .so travels with the repo
snapshot to the GPU box.  hipcc cross-compiles gfx950 without a GPU.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
ROOT = os.path.dirname(HERE)

# Kernel files: compiled once per ARITHMETIC CONTRACT (csrc/ofdis_dev.h) from the same source, into ofdis::exact and
# ofdis::fused; the pyramid (exact by construction for 8-bit input) and the C ABI are compiled once.
KERNEL_SOURCES = ["ofdis_dis.hip", "ofdis_tv.hip", "ofdis_prep.hip", "ofdis_sor.hip", "ofdis_fused.hip", "ofdis_fused_xcu.hip",
                  "ofdis_fused_tall.hip", "ofdis_de.hip"]
COMMON_SOURCES = ["ofdis_pyr.hip", "ofdis_capi.hip"]
HIP_SOURCES = KERNEL_SOURCES + COMMON_SOURCES
# -fvisibility=hidden: the shared library exports the C ABI of include/ofdis.h (marked in ofdis_capi.hip) and nothing else.
BASEFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
             "-Wall", "-Wno-unused-function", "-Wno-bitwise-instead-of-logical"]
# exact: -ffp-contract=off, every fp32 operation separately rounded, like the reference's SSE path (bit-identical results).
# fused: the tolerance contract (EPE < 1e-3 px): a * b + c contracts into one v_fma_f32, also across statements.
CONTRACT_FLAGS = {"exact": ["-DOFDIS_CONTRACT=0", "-ffp-contract=off"],
                  "fused": ["-DOFDIS_CONTRACT=1", "-ffp-contract=fast"]}
HIPFLAGS = BASEFLAGS + CONTRACT_FLAGS["exact"]


# No SLP vectorisation: gfx950's SIMDs are 32 lanes wide, so a packed v_pk_*_f32 occupies the issue port about as
# long as the two scalar ops it replaces (tools/probes/valu_probe.hip: 4.2 vs 2 x 2.5 cycles per wave64) and the
# moves that assemble register pairs are pure overhead; in ofdis_dis.hip it also splits the DPP reduction chains.
_NO_SLP = ["-fno-slp-vectorize"]
PER_FILE_FLAGS = {"ofdis_dis.hip": _NO_SLP, "ofdis_tv.hip": _NO_SLP, "ofdis_prep.hip": _NO_SLP, "ofdis_fused.hip": _NO_SLP,
                  "ofdis_fused_xcu.hip": _NO_SLP, "ofdis_fused_tall.hip": _NO_SLP,
                  "ofdis_sor.hip": _NO_SLP}


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def abi_symbols():
    """Every function include/ofdis.h declares: the export list of the shared library."""
    import re
    src = open(os.path.join(ROOT, "include", "ofdis.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ofdis_[a-z0-9_]+)\s*\(", src)))


def source_id(csrc=None, extra_flags=None):
    """Identity of what the kernels of a build are made of: a hash over every kernel / header source under csrc/ (not the
    host/ mains), include/ofdis.h and the compiler flags above.  It is compiled into the library (ofdis_build_id()) and
    stamped into profiles/traffic_*.json when the PMC counters are collected: bench.py attaches counter-derived figures only
    to a library with the SAME id -- a kernel change without a PMC re-run cannot keep a stale roofline fraction."""
    import hashlib
    csrc = csrc or CSRC
    h = hashlib.sha256()
    names = sorted(f for f in os.listdir(csrc) if f.endswith((".hip", ".h", ".inc")))
    for name in names:
        h.update(name.encode() + b"\0")
        h.update(open(os.path.join(csrc, name), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "ofdis.h"), "rb").read())
    h.update(repr((BASEFLAGS, sorted(CONTRACT_FLAGS.items()), sorted(PER_FILE_FLAGS.items()),
                   sorted((extra_flags or {}).items()))).encode())
    return h.hexdigest()[:16]


def _build_id_header(csrc, outdir, extra_flags=None):
    """outdir/ofdis_build_id.h (included by ofdis_capi.hip), rewritten only when the id changes."""
    path = os.path.join(outdir, "ofdis_build_id.h")
    text = f'#define OFDIS_BUILD_ID "{source_id(csrc, extra_flags)}"\n'
    if not os.path.exists(path) or open(path).read() != text:
        open(path, "w").write(text)
    return path


def _version_script():
    """Linker version script exporting include/ofdis.h and nothing else (kernel handles, inline members of the context
    struct and the runtime's registration symbols stay local)."""
    path = os.path.join(LIBDIR, "ofdis.map")
    text = "{\n  global:\n" + "".join(f"    {n};\n" for n in abi_symbols()) + "  local: *;\n};\n"
    if not os.path.exists(path) or open(path).read() != text:
        open(path, "w").write(text)
    return path


def _run(cmd, verbose):
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("build failed: " + " ".join(cmd) + "\n" + r.stdout + r.stderr)
    if verbose and (r.stdout or r.stderr):
        print(r.stdout + r.stderr)


def lib_path():
    return os.path.join(LIBDIR, "libofdis_hip.so")


def testhooks_path():
    """The TEST library (tests/csrc/ofdis_testhooks.hip): kernels that expose header-only helpers to the parity tests.
    Built next to the product, never linked into it."""
    return os.path.join(LIBDIR, "libofdis_testhooks.so")


def compile_units(csrc, outdir, force=False, verbose=False, extra_flags=None):
    """Compile every translation unit of the library (kernel files once per contract) from `csrc` into `outdir`; returns
    the object list.  extra_flags: {file name: [flags]} on top of PER_FILE_FLAGS (developer A/B builds, tools/ab_build.py)."""
    hipcc = _hipcc()
    headers = [os.path.join(csrc, h) for h in os.listdir(csrc) if h.endswith((".h", ".inc"))]
    headers.append(os.path.join(ROOT, "include", "ofdis.h"))
    headers.append(os.path.abspath(__file__))  # the flags live here
    build_id_h = _build_id_header(csrc, outdir, extra_flags)  # changes whenever any kernel source or flag does
    objs, jobs = [], []
    units = [(src, c) for c in ("exact", "fused") for src in KERNEL_SOURCES] + [(src, "exact") for src in COMMON_SOURCES]
    for src, contract in units:
        sp = os.path.join(csrc, src)
        suffix = ".o" if contract == "exact" else "." + contract + ".o"
        obj = os.path.join(outdir, src.replace(".hip", suffix))
        deps = [sp] + headers + ([build_id_h] if src == "ofdis_capi.hip" else [])
        if force or _newer(obj, deps):
            flags = PER_FILE_FLAGS.get(src, []) + (extra_flags or {}).get(src, [])
            if src == "ofdis_capi.hip":
                flags = flags + ["-I", outdir]
            jobs.append([hipcc] + BASEFLAGS + CONTRACT_FLAGS[contract] + flags + ["-c", sp, "-o", obj])
        objs.append(obj)
    if jobs:  # the translation units are independent: compile them side by side
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as ex:
            list(ex.map(lambda cmd: _run(cmd, verbose), jobs))
    return objs, headers


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    objs, headers = compile_units(CSRC, LIBDIR, force, verbose)
    so = lib_path()
    vs = _version_script()
    if force or _newer(so, objs + [vs]):
        _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", f"-Wl,--version-script={vs}", "-o", so] + objs, verbose)
    # host executables: the reference's CLI contract (run_OF_INT / run_OF_RGB and the stereo-depth run_DE_INT / run_DE_RGB,
    # one pair per process: host/run_dense_main.cpp) and the sequence drivers (run_OF_*_seq / run_DE_*_seq: many pairs,
    # one host thread per GPU: host/run_seq_main.cpp); every other .cpp under host/ is shared by both mains
    host_dir = os.path.join(CSRC, "host")
    mains = {"run_dense_main.cpp": (("run_OF_INT", 1, 1), ("run_OF_RGB", 3, 1), ("run_DE_INT", 1, 2), ("run_DE_RGB", 3, 2)),
             "run_seq_main.cpp": (("run_OF_INT_seq", 1, 1), ("run_OF_RGB_seq", 3, 1), ("run_DE_INT_seq", 1, 2),
                                  ("run_DE_RGB_seq", 3, 2))}
    if os.path.isdir(host_dir):
        common = [os.path.join(host_dir, f) for f in sorted(os.listdir(host_dir)) if f.endswith(".cpp") and f not in mains]
        host_hdrs = [os.path.join(host_dir, f) for f in os.listdir(host_dir) if f.endswith(".h")]
        for main_name, exes in mains.items():
            main_cpp = os.path.join(host_dir, main_name)
            if not os.path.exists(main_cpp):
                continue
            for name, noc, mode in exes:
                exe = os.path.join(LIBDIR, name)
                if force or _newer(exe, [main_cpp] + common + host_hdrs + headers + [so]):
                    _run(["g++", "-O2", "-std=c++17", "-Wall", "-pthread", f"-DOFDIS_NOC={noc}", f"-DOFDIS_MODE={mode}", "-I",
                          os.path.join(ROOT, "include"), main_cpp] + common
                         + ["-o", exe, "-L", LIBDIR, "-lofdis_hip", "-lz", "-Wl,-rpath,$ORIGIN"], verbose)
    # test-only artefacts: a failure here must not take the product down with it (it is reported, the tests that need
    # them fail on their own)
    try:
        # the plain-C caller of the drop-in boundary (tests/c/dropin_test.c): compiled as C99 against include/ofdis.h
        c_test = os.path.join(ROOT, "tests", "c", "dropin_test.c")
        if os.path.exists(c_test):
            exe = os.path.join(LIBDIR, "dropin_test")
            if force or _newer(exe, [c_test, os.path.join(ROOT, "include", "ofdis.h"), so]):
                _run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-O2", "-D_POSIX_C_SOURCE=199309L", "-I",
                      os.path.join(ROOT, "include"), c_test, "-o", exe, "-L", LIBDIR, "-lofdis_hip", "-Wl,-rpath,$ORIGIN"],
                     verbose)
        hooks = os.path.join(ROOT, "tests", "csrc", "ofdis_testhooks.hip")
        if os.path.exists(hooks):
            tso = testhooks_path()
            if force or _newer(tso, [hooks] + headers):
                _run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
                      hooks, "-o", tso], verbose)
    except RuntimeError as e:
        print("of_dis_amd.build: test artefacts not built:", str(e)[:2000], file=sys.stderr)
    return so


if __name__ == "__main__":
    p = build(force="--force" in sys.argv, verbose=True)
    print("built", p)
