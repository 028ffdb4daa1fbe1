"""Developer helper: per-kernel ISA summary (largest loop: instruction counts by class, 8-byte encodings) of a HIP object.

    python tools/isa_count.py of_dis_amd/lib/ofdis_fused.o [kernel-name-substring ...]

Extracts the gfx950 code object from the (bundled) object / shared library, disassembles it and, for every kernel
whose demangled name contains one of the substrings, finds the longest backward-branch loop and counts its
instructions: VALU (4-byte / 8-byte encodings, DPP, transcendental), SALU, VMEM, LDS, waitcnt, s_nop.
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def code_object(path):
    tmp = tempfile.mkdtemp()
    out = os.path.join(tmp, "gfx950.co")
    kind = "o" if path.endswith(".o") else "o"
    r = subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=" + kind, "--input=" + path,
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + out], capture_output=True, text=True)
    if r.returncode != 0 or not os.path.exists(out) or os.path.getsize(out) == 0:
        # shared library: the fat binary sits in .hip_fatbin
        fb = os.path.join(tmp, "fatbin")
        subprocess.check_call([LLVM + "/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", path, fb])
        subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fb,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + out])
    return out


def disasm(co):
    return subprocess.run([LLVM + "/llvm-objdump", "-d", "--demangle", co], capture_output=True, text=True).stdout


def kernels(text):
    cur, body = None, []
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if m:
            if cur:
                yield cur, body
            cur, body = m.group(1), []
        elif cur and re.match(r"^\s+[sv]_|^\s+(buffer|global|flat|ds|scratch)_", line):
            body.append(line)
    if cur:
        yield cur, body


def parse(line):
    # "\tv_add_f32_e32 v1, v2, v3    // 000000001234: 02060702"
    m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):\s*((?:[0-9A-Fa-f]{8}\s*)+)", line)
    if not m:
        return None
    return m.group(1), m.group(2), int(m.group(3), 16), len(m.group(4).split()) * 4


def classify(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    return "vmem"


# back-to-back issue costs of tools/probes/issue_probe.hip (2+ wavefronts per SIMD); in a mixed stream compares, selects,
# SGPR and literal operands pair like plain instructions (2.1) and every DPP / transcendental costs its wavefront the
# pairing for ~100 instructions (profiles/README.md, snapshot r02_c) -- the estimate below is a lower bound
COST = {"plain": 2.1, "inline": 2.1, "vop3": 2.2, "dpp": 4.1, "sgpr": 2.1, "literal": 2.1, "trans": 8.0,
        "cmp": 2.1, "cndmask_sgpr": 2.1, "cndmask_vcc": 2.1}


def valu_class(op, args, nbytes):
    srcs = [x.strip() for x in args.split(",")][1:]
    if re.match(r"v_(rcp|sqrt|rsq|exp|log|sin|cos)", op):
        return "trans"
    if op.endswith("_dpp"):
        return "dpp"
    if op.startswith("v_cndmask"):
        return "cndmask_sgpr" if nbytes >= 8 else "cndmask_vcc"
    if op.startswith("v_cmp"):
        return "cmp"
    if any(re.match(r"^-?\|?(s\d+|s\[|vcc|exec|ttmp|m0)", x) for x in srcs):
        return "sgpr"
    if any(re.match(r"^0x", x) for x in srcs):
        return "literal"
    if nbytes >= 8:
        return "vop3"
    if any(re.match(r"^-?\d", x) for x in srcs):
        return "inline"
    return "plain"


def summarize(name, body, quiet=False):
    ins = [p for p in (parse(l) for l in body) if p]
    addr = {a: i for i, (_, _, a, _) in enumerate(ins)}
    best = None
    for i, (op, args, a, n) in enumerate(ins):
        if op.startswith("s_cbranch") or op == "s_branch":
            mm = re.match(r"(-?\d+)", args)  # simm16, printed unsigned
            if mm:
                off = int(mm.group(1))
                if off >= 32768:
                    off -= 65536
                tgt = a + 4 + off * 4
                if tgt in addr and addr[tgt] <= i:
                    span = (addr[tgt], i)
                    if best is None or span[1] - span[0] > best[1] - best[0]:
                        best = span
    if best is None:
        if not quiet:
            print(name, ": no loop found,", len(ins), "instructions")
        return None
    loop = ins[best[0]:best[1] + 1]
    if quiet:
        c = {}
        for op, args, a, n in loop:
            c[classify(op)] = c.get(classify(op), 0) + 1
        return {"loop_instructions": len(loop), "loop_valu": c.get("valu", 0), "loop_vmem": c.get("vmem", 0),
                "loop_lds": c.get("lds", 0), "loop_salu": c.get("salu", 0)}
    c = {}
    v8 = dpp = trans = 0
    ops = {}
    for op, args, a, n in loop:
        k = classify(op)
        c[k] = c.get(k, 0) + 1
        if k == "valu":
            base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
            ops[base] = ops.get(base, 0) + 1
            if n >= 8:
                v8 += 1
            if op.endswith("_dpp"):
                dpp += 1
            if re.match(r"v_(rcp|sqrt|rsq|exp|log|sin|cos)", op):
                trans += 1
    print(f"{name}\n  loop of {len(loop)} instructions: " + ", ".join(f"{k} {v}" for k, v in sorted(c.items())) +
          f"; VALU 8-byte encoded {v8}, DPP {dpp}, transcendental {trans}")
    print("  " + ", ".join(f"{k} {v}" for k, v in sorted(ops.items(), key=lambda kv: -kv[1])[:14]))
    # operand classes with the issue costs measured by tools/probes/class_probe.hip (SIMD clocks per wave64 instruction)
    cl = {}
    for op, args, a, n in loop:
        if classify(op) != "valu":
            continue
        k = valu_class(op, args, n)
        cl[k] = cl.get(k, 0) + 1
    est = sum(COST.get(k, 2.1) * v for k, v in cl.items())
    print("  classes: " + ", ".join(f"{k} {v}" for k, v in sorted(cl.items(), key=lambda kv: -kv[1])) +
          f"; paired-issue lower bound {est:.0f} clocks per loop pass")


# the kernels whose iteration loops bench.py prices against the VALU issue peak (static counts of the SHIPPED objects):
# key -> (object file, kernel name as disassembled, patches per wavefront)
PROFILE_KERNELS = {
    "patch_optimize_rgb12_fused": ("ofdis_dis.fused.o", "ofdis::fused::patch_optimize_rgb12_kernel<1, 3, false, 3>(", 4),
    "patch_optimize_rgb12_exact": ("ofdis_dis.o", "ofdis::exact::patch_optimize_rgb12x_kernel<1, 3, false, 3>(", 4),
    "patch_optimize_gray8_fused": ("ofdis_dis.fused.o", "ofdis::fused::patch_optimize_gray8_kernel<0, false>(", 16),
    "patch_optimize_gray8_exact": ("ofdis_dis.o", "ofdis::exact::patch_optimize_gray8_kernel<0, false>(", 16),
}


def profile_counts(libdir):
    """{key: {kernel, patches_per_wavefront, loop_instructions, ...}} for PROFILE_KERNELS (profiles/isa_counts.json)."""
    out, cache = {}, {}
    for key, (obj, kname, ppw) in PROFILE_KERNELS.items():
        if obj not in cache:
            cache[obj] = list(kernels(disasm(code_object(os.path.join(libdir, obj)))))
        for name, body in cache[obj]:
            if kname in name:
                c = summarize(name, body, quiet=True)
                if c:
                    out[key] = dict(kernel=name.split("(")[0].replace("void ", ""), patches_per_wavefront=ppw, **c)
    return out


if __name__ == "__main__":
    if sys.argv[1] == "--write-profile":  # python tools/isa_count.py --write-profile  ->  profiles/isa_counts.json
        import json
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        d = profile_counts(os.path.join(root, "of_dis_amd", "lib"))
        d["sustained_clock_ghz"] = 2.157  # profiles/traffic_fused.json: GRBM_GUI_ACTIVE / dispatch duration under this load
        d["what"] = ("instructions of one pass of the iteration loop (largest backward-branch loop) of the shipped code objects, "
                     "tools/isa_count.py; tests/test_isa.py checks that the file matches the build")
        json.dump(d, open(os.path.join(root, "profiles", "isa_counts.json"), "w"), indent=1, sort_keys=True)
        print(json.dumps(d, indent=1, sort_keys=True))
        sys.exit(0)
    co = code_object(sys.argv[1])
    text = disasm(co)
    pats = sys.argv[2:]
    for name, body in kernels(text):
        if not pats or any(p in name for p in pats):
            summarize(name, body)
