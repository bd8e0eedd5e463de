"""Developer tool: the ISA of one trace_stack_kernel instantiation, its main loop tabulated per source line by instruction group
(VERDICT r3 next 1a).  Groups as measured by tools/issue_rate.hip on gfx950 at 6 waves per SIMD:
  F  f32 add / sub / mul / fma / fmac (also with clamp, modifiers, literals): 2 cycles, issue beside an S instruction for nothing
  I  v_add_u32 / v_sub_u32 / v_and / v_or / v_xor / v_not / v_mov / right shifts: 2 cycles alone, ~4 next to S instructions
  S  everything else in the vector ALU (compares, selects, min / max, conversions, v_bfe, shift-or / and-or / bfi / add-shift forms,
     left shifts, ldexp, floor, ffbh, readlane, mad_u24 ...): 4 cycles
  SALU / LDS / VMEM / SMEM by opcode prefix.
usage: python tools/isa_table.py [--mangled-suffix ILi256ELi12ELi3ELb0ELb0ELb0ELb0E] [--out-prefix profiles/r04_hot_loop]
writes <prefix>_isa.s (the instantiation's ISA) and <prefix>_isa_table.md."""
import argparse
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "octree-tracer_amd", "csrc", "svo_kernels.hip")
FAST_F = {"v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mac_f32", "v_fmamk_f32", "v_fmaak_f32"}
FAST_I = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_xor_b32", "v_and_b32", "v_or_b32", "v_mov_b32", "v_not_b32", "v_lshrrev_b32", "v_ashrrev_i32"}


def group(op):
    base = op.replace("_e32", "").replace("_e64", "")
    if base.startswith("v_"):
        return "F" if base in FAST_F else ("I" if base in FAST_I else "S")
    if base.startswith("s_waitcnt") or base.startswith("s_nop"):
        return "wait"
    if base.startswith("s_load") or base.startswith("s_buffer_load") or base.startswith("s_memtime"):
        return "SMEM"
    if base.startswith("s_"):
        return "SALU"
    if base.startswith("ds_"):
        return "LDS"
    return "VMEM"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mangled-suffix", default="ILi256ELi12ELi3ELb0ELb0ELb0ELb0E", help="template arguments of the instantiation (BLOCK, NS, K, GE, DBG, CNT, SHD)")
    ap.add_argument("--out-prefix", default=os.path.join(ROOT, "profiles", "r04_hot_loop"))
    ap.add_argument("--flags", default="")
    a = ap.parse_args()
    asm = "/tmp/isa_table.s"
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-slp-vectorize", "-gline-tables-only",
           "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", SRC, "-o", asm] + a.flags.split()
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    lines = open(asm).read().split("\n")
    name = "_ZN3svo18trace_stack_kernel" + a.mangled_suffix + "EEvNS_9TraceArgsEjPjS2_"
    start = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end + 1]
    files = {}
    for l in lines:
        m = re.match(r"\s*\.file\s+(\d+)\s+\"([^\"]*)\"(?:\s+\"([^\"]*)\")?", l)
        if m:
            files[int(m.group(1))] = os.path.basename(m.group(3) or m.group(2))
    src_lines = open(SRC).read().split("\n")
    loop = max(i for i, l in enumerate(body) if "Loop Header: Depth=1" in l)
    cur = ("?", 0)
    per_line = collections.OrderedDict()
    totals = collections.Counter()
    out_isa = []
    for i, l in enumerate(body):
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
        if m:
            cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
            continue
        if l.strip().startswith(".cfi") or l.strip().startswith(";;#") or re.match(r"\s*;", l):
            if "Loop Header" in l or "Parent Loop" in l:
                out_isa.append(l)
            continue
        out_isa.append(l if not (l.startswith("\t") and not l.startswith("\t.")) else f"{l:<64s} ; {cur[0]}:{cur[1]}")
        if i <= loop or not l.startswith("\t") or l.startswith("\t."):
            continue
        g = group(l.split()[0])
        if g == "wait":
            continue
        per_line.setdefault(cur, collections.Counter())[g] += 1
        totals[g] += 1
    with open(a.out_prefix + "_isa.s", "w") as f:
        f.write("\n".join(out_isa) + "\n")
    cols = ["F", "I", "S", "SALU", "LDS", "VMEM", "SMEM"]
    with open(a.out_prefix + "_isa_table.md", "w") as f:
        f.write(f"# trace_stack_kernel<{a.mangled_suffix}>: main loop, instructions per source line (static: every instruction once)\n\n")
        f.write("Groups (tools/issue_rate.hip, profiles/r04_issue_rate*.log): F = f32 add/mul/fma, 2 cycles, free beside an S instruction; I = simple integer / move,\n"
                "2 cycles; S = the rest of the vector ALU, 4 cycles.  The walk's loop body runs ~2.5 times per round on top of its first pass, the refill and\n"
                "generation blocks in about one round of ten.\n\n")
        f.write("| source line | " + " | ".join(cols) + " | text |\n|---|" + "---|" * (len(cols) + 1) + "\n")
        for (fn, ln), c in per_line.items():
            text = src_lines[ln - 1].strip()[:90].replace("|", "\\|") if fn == os.path.basename(SRC) and 0 < ln <= len(src_lines) else ""
            f.write(f"| {fn}:{ln} | " + " | ".join(str(c.get(k, "")) for k in cols) + f" | `{text}` |\n")
        f.write("| **total (static)** | " + " | ".join(str(totals.get(k, 0)) for k in cols) + " | |\n")
    # per section of the round, by where the source line lies (lambdas are inlined: their lines say what they belong to)
    def find(marker):
        return next(i + 1 for i, l in enumerate(src_lines) if marker in l)
    marks = [("restart_at (end of the step: where the next walk starts)", find("auto restart_at_rb = [&]")),
             ("flush_record (refill: the record of a finished ray)", find("auto flush_record = [&]")),
             ("walk (descend)", find("auto descend = [&]")),
             ("camera shortcut set-up (before the loop)", find("// Camera shortcut.")),
             ("round head / walk call", find("// ---- 1. descent")),
             ("refill: claim, generation, pick-up", find("// ---- 2. refill")),
             ("counting (CNT only)", find("// ---- 3a. hit counters")),
             ("step (hit test / DDA step)", find("// ---- 3. hit test")),
             ("after the loop", find("if (CNT) cq_flush();"))]
    marks.sort(key=lambda m: m[1])
    sect = collections.OrderedDict()
    for (fn, ln), c in per_line.items():
        if fn != os.path.basename(SRC) or ln == 0:
            key = "helpers (svo_trace_fn.h, HIP headers: ray generation, item decoding, atomics)"
        else:
            key = "before the kernel's lambdas"
            for name_, at in marks:
                if ln >= at:
                    key = name_
        sect.setdefault(key, collections.Counter()).update(c)
    with open(a.out_prefix + "_isa_table.md", "a") as f:
        f.write("\n## By section (static counts inside the main loop)\n\n| section | " + " | ".join(cols) + " |\n|---|" + "---|" * len(cols) + "\n")
        for k, c in sect.items():
            f.write(f"| {k} | " + " | ".join(str(c.get(x, 0)) for x in cols) + " |\n")
    for k, c in sect.items():
        print(f"{k:70s}", {x: c[x] for x in cols if c.get(x)})
    v = totals["F"] + totals["I"] + totals["S"]
    print("main loop, static:", dict(totals), "slow-group share of vector instructions:", round(totals["S"] / max(v, 1), 3))


if __name__ == "__main__":
    main()
