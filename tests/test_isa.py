"""Static checks on the built gfx950 code object (no GPU needed): no kernel spills registers or uses scratch memory, and the
hot kernels keep the register budgets their occupancy figures in DESIGN.md rest on."""
import os
import re
import subprocess
import tempfile

import pytest

from of_dis_amd import build as B

LLVM = "/opt/rocm/lib/llvm/bin"


def _kernel_notes():
    so = B.lib_path()
    if not os.path.exists(so):
        B.build()
    tmp = tempfile.mkdtemp()
    fb = os.path.join(tmp, "fatbin")
    subprocess.check_call([LLVM + "/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, fb])
    blob = open(fb, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)]   # one bundle per translation unit
    kernels = {}
    for i, st in enumerate(starts):
        part, co = os.path.join(tmp, "bundle%d" % i), os.path.join(tmp, "gfx950_%d.co" % i)
        with open(part, "wb") as f:
            f.write(blob[st:starts[i + 1] if i + 1 < len(starts) else len(blob)])
        subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + part,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
        cur = None
        for line in notes.splitlines():
            m = re.match(r"\s*-?\s*\.(\w+):\s+(\S+)", line)
            if not m:
                continue
            k, v = m.group(1), m.group(2)
            if k == "agpr_count":      # first key of a kernel's metadata map
                cur = {}
            if cur is not None:
                cur[k] = v
                if k == "wavefront_size":  # last key of the (alphabetically ordered) map
                    sym = cur["symbol"]
                    kernels[sym[:-3] if sym.endswith(".kd") else sym] = cur   # mangled name
                    cur = None
    return kernels


@pytest.fixture(scope="module")
def kernels():
    if not os.path.exists(LLVM + "/llvm-readelf"):
        pytest.skip("ROCm LLVM tools not found")
    k = _kernel_notes()
    assert len(k) > 40, len(k)
    return k


# the 1024-thread block SOR (levels of 641 ... 1024 rows: 16 wavefronts of one workgroup on a CU = 128 VGPRs each) is the one
# kernel allowed to spill; every kernel of the benchmarked configurations must not
SPILL_OK = ("sor_block_kernelILi3ELi3ELi1024EE",)


def test_no_kernel_spills_or_uses_scratch(kernels):
    for name, m in kernels.items():
        if any(s in name for s in SPILL_OK):
            continue
        assert int(m["vgpr_spill_count"]) == 0, (name, m)
        assert int(m["sgpr_spill_count"]) == 0, (name, m)
        assert int(m["private_segment_fixed_size"]) == 0, (name, m)


@pytest.mark.parametrize("pattern,max_vgprs,what", [
    ("tv_fused_kernelILi3ELb1ELi0EE", 168, "tv_fused_kernel<3, true, 0>, throughput mapping: three wavefronts per SIMD"),
    ("tv_fused_kernelILi3ELb1ELi2EE", 168, "tv_fused_kernel<3, true, 2>, split mapping: 12 wavefronts of a workgroup on one CU"),
    ("patch_optimize_gray8_kernelILi0EE", 128, "gray 8x8 patch kernel: four wavefronts per SIMD"),
    ("patch_optimize_kernelILi7ELi64ELi432ELi1EE", 84, "RGB 12x12 patch kernel, L1 cost: six wavefronts per SIMD"),
    ("densify_kernelILb1EE", 64, "densify_kernel<true>: eight wavefronts per SIMD"),
])
def test_register_budgets(kernels, pattern, max_vgprs, what):
    hits = [(n, m) for n, m in kernels.items() if pattern in n]
    assert hits, pattern
    for n, m in hits:
        assert int(m["vgpr_count"]) <= max_vgprs, (what, n, m["vgpr_count"])
