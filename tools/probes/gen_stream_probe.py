"""Generate tools/probes/stream_probe.hip: the VALU instruction stream of a kernel's main loop (taken from the
disassembly of the built object), replayed without memory traffic, to separate the issue cost of the stream itself
from everything else.  Variants of the same stream answer "what would it cost if ...":

    asis      the loop's VALU instructions (and s_nop) in program order
    shuffle   the same instructions in random order (true dependencies destroyed: per-instruction cost only)
    vconst    as is, SGPR / literal source operands replaced by VGPRs (constants kept in registers)
    rot       as is, every VGPR number rotated by a constant (different register-bank assignment)

    python tools/probes/gen_stream_probe.py of_dis_amd/lib/ofdis_fused.o "tv_fused_kernel<3, true, 0>" > tools/probes/stream_probe.hip
"""
import os
import random
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import isa_count as I  # noqa: E402


def main_loop(obj, pat):
    text = I.disasm(I.code_object(obj))
    for name, body in I.kernels(text):
        if pat in name:
            ins = [p for p in (I.parse(l) for l in body) if p]
            addr = {a: i for i, (_, _, a, _) in enumerate(ins)}
            best = None
            for i, (op, args, a, n) in enumerate(ins):
                if op.startswith("s_cbranch"):
                    off = int(args.split()[0])
                    off = off - 65536 if off >= 32768 else off
                    t = a + 4 + off * 4
                    if t in addr and addr[t] <= i and (best is None or i - addr[t] > best[1] - best[0]):
                        best = (addr[t], i)
            return name, ins[best[0]:best[1] + 1]
    raise SystemExit("kernel not found")


def regs(args, kind):
    out = set()
    for m in re.finditer(kind + r"(\d+)\b", args):
        out.add(int(m.group(1)))
    for m in re.finditer(kind + r"\[(\d+):(\d+)\]", args):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def variant(stream, kind, seed=1):
    if "__" in kind:  # combination: applied left to right
        for kd in kind.split("__"):
            stream = variant(stream, kd, seed)
        return stream
    rnd = random.Random(seed)
    if kind == "lds":
        return list(stream)
    if kind == "ldsnowait":
        return [(op, a) for op, a in stream if op != "s_waitcnt"]
    out = []
    for op, args in stream:
        if kind == "vconst" and op.startswith("v_") and not op.startswith("v_cndmask") and not op.startswith("v_cmp") \
                and not op.startswith("v_readfirstlane"):
            parts = [x.strip() for x in args.split(",")]
            for i in range(1, len(parts)):
                tok = parts[i].split()[0] if parts[i] else parts[i]
                if re.match(r"^-?\|?s\d+\|?$", tok) or re.match(r"^0x[0-9a-fA-F]+$", tok):
                    parts[i] = parts[i].replace(tok, "v%d" % (152 + rnd.randrange(8)), 1)
            args = ", ".join(parts)
            if op.endswith("_e32") is False and op.endswith("_e64") and not re.search(r"\bs\d|\bs\[|\|-|-v|\|", args) \
                    and op[:-4] in ("v_mul_f32", "v_add_f32", "v_sub_f32"):
                op = op[:-4] + "_e32"
        if kind == "rot":
            args = re.sub(r"\bv(\d+)\b", lambda m: "v%d" % ((int(m.group(1)) + 1) % 160), args)
        out.append((op, args))
    if kind in ("nop_after", "salu_nop", "nop_cluster"):  # a scalar instruction right after every DPP / transcendental
        o2 = []
        for i, (op, a) in enumerate(out):
            o2.append((op, a))
            slow = op.endswith("_dpp") or re.match(r"v_(rcp|sqrt|rsq)", op)
            nxt = out[i + 1][0] if i + 1 < len(out) else ""
            nslow = nxt.endswith("_dpp") or re.match(r"v_(rcp|sqrt|rsq)", nxt)
            if slow and not (kind == "nop_cluster" and nslow):
                o2.append(("s_nop", "0"))
        out = o2
    m = re.match(r"cl(\d+)(nonop)?(dt)?$", kind)
    if m:  # timing only: the DPP / transcendental instructions of every sixth of the loop (= one diagonal step) gathered
        # into C clusters spread evenly over the step, one scalar instruction after each cluster
        C = int(m.group(1))
        seg = (len(out) + 5) // 6
        o2 = []
        for s0 in range(0, len(out), seg):
            part = out[s0:s0 + seg]
            isslow = lambda op: op.endswith("_dpp") or re.match(r"v_(rcp|sqrt|rsq)", op)
            slow = [x for x in part if isslow(x[0])]
            if m.group(3):  # DPP first, transcendental after: separate clusters
                slow = [x for x in slow if x[0].endswith("_dpp")] + [x for x in slow if not x[0].endswith("_dpp")]
            fast = [x for x in part if not isslow(x[0])]
            per = (len(slow) + C - 1) // C
            gap = len(fast) // C
            for c in range(C):
                o2 += slow[c * per:(c + 1) * per]
                if not m.group(2):
                    o2.append(("s_nop", "0"))
                o2 += fast[c * gap:(c + 1) * gap] if c < C - 1 else fast[c * gap:]
        out = o2
    if kind == "shuffle":
        rnd.shuffle(out)
    if kind == "perm":  # random renaming of the VGPRs: other register-bank relations between an instruction's operands
        vs = sorted({v for op, a in out for v in regs(a, "v")})
        pm = vs[:]
        rnd.shuffle(pm)
        mp = dict(zip(vs, pm))
        out = [(op, re.sub(r"\bv(\d+)\b", lambda m: "v%d" % mp[int(m.group(1))], a)) for op, a in out]
    drop = {"plainonly": ("dpp", "trans", "cmp", "cndmask_sgpr", "cndmask_vcc", "sgpr"), "nodpp": ("dpp",),
            "notrans": ("trans",), "nocmpcnd": ("cmp", "cndmask_sgpr", "cndmask_vcc"), "nosgpr": ("sgpr",),
            "nolit": ("literal",), "novop3": ("vop3",)}.get(kind)
    if drop:
        out = [(op, a) for op, a in out if not op.startswith("v_") or I.valu_class(op, a, 8 if (op.endswith("_e64") or
               op.startswith("v_fma_") or op.endswith("_dpp")) else 4) not in drop]
    return out


def emit(name, tag, stream):
    vmax = 0
    swr = set()
    for op, args in stream:
        v = regs(args, "v")
        if v:
            vmax = max(vmax, max(v))
        if op.startswith("v_cmp") or op.startswith("v_readfirstlane") or op.startswith("v_div_scale"):
            first = args.split(",")[0]
            swr |= regs(first, "s")
    nv = sum(1 for op, _ in stream if op.startswith("v_"))
    clob = ['"v%d"' % i for i in range(0, max(vmax, 159) + 1)] + ['"s%d"' % i for i in sorted(swr)] + ['"vcc"', '"memory"']
    body = "\\n\\t".join(f"{op} {args}".strip() for op, args in stream)
    return nv, f'''
// {name}: {tag}, {nv} VALU instructions per pass
__global__ __launch_bounds__(512) void k_{tag}(unsigned long long* clk, int iters, int delay) {{
  if (delay && threadIdx.x >= 256) for (int d = 0; d < delay; ++d) __builtin_amdgcn_s_sleep(1);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {{
    asm volatile("{body}" ::: {", ".join(clob)});
  }}
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) atomicMax(clk, t1 - t0);
}}
'''


if __name__ == "__main__":
    name, loop = main_loop(sys.argv[1], sys.argv[2])
    def keep(op, args):
        if op.startswith("v_") or op.startswith("s_nop") or op.startswith("ds_bpermute"):
            return (op, args)
        if op == "s_waitcnt" and "lgkmcnt" in args:
            return (op, re.search(r"lgkmcnt\(\d+\)", args).group(0))
        return None
    full = [keep(op, args) or ("s_nop", "0") for op, args, a, n in loop]
    lds = [keep(op, args) for op, args, a, n in loop if keep(op, args)]
    lds = [(op, args) for op, args in lds if not op.startswith("v_readfirstlane") and not op.startswith("v_cmpx")]
    full = [(op, args) for op, args in full if not op.startswith("v_readfirstlane") and not op.startswith("v_cmpx")]
    stream = [(op, args) for op, args, a, n in loop if op.startswith("v_") or op.startswith("s_nop")]
    # exec-mask writers / lane-crossing ops with SGPR results that the replay cannot keep meaningful are still plain VALU
    # issue; v_readfirstlane writes an SGPR the compiler may own: drop it
    stream = [(op, args) for op, args in stream if not op.startswith("v_readfirstlane") and not op.startswith("v_cmpx")]
    kinds = ["asis", "lds", "ldsnowait", "salu", "notrans"]
    print("// GENERATED by tools/probes/gen_stream_probe.py from", sys.argv[1], "--", name)
    print("// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/stream_probe.hip -o tools/probes/stream_probe")
    print("#include <hip/hip_runtime.h>\n#include <stdio.h>")
    counts = {}
    for kd in kinds:
        base = full if kd in ("salu", "salu_nop") else (lds if kd.startswith("lds") else stream)
        nv, src = emit(name, kd, variant(base, kd))
        counts[kd] = nv
        print(src)
    print('''
template <typename K>
void run(const char* tag, K kern, int nvalu, unsigned long long* clk) {
  const int iters = 200;
  printf("%-8s (%d VALU/pass):", tag, nvalu);
  for (int delay : {0, 1}) {  // the second wavefront of every SIMD enters the loop `delay` x s_sleep 1 later
    const int wps = 2;
    const int blocks = 256;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, clk, 3, delay);
    hipDeviceSynchronize();
    hipMemset(clk, 0, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, clk, iters, delay);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    // longest-resident wavefront's clocks = the launch in shader clocks; per SIMD: wps wavefronts x iters passes
    printf("  delay %2d: %.2f clk/instr/SIMD (%.0f clk/pass/SIMD, %.0f MHz, %.3f ms)", delay, (double)c / ((double)iters * nvalu * wps),
           (double)c / ((double)iters * wps), (double)c / (ms * 1e3), ms);
  }
  printf("\\n");
}
int main() {
  unsigned long long* clk; hipMalloc(&clk, 8);''')
    for kd in kinds:
        print(f'  run("{kd}", k_{kd}, {counts[kd]}, clk);')
    print("  return 0;\n}")
