#!/usr/bin/env python3
"""Static ISA instruction histogram of one drone_kernel instantiation, per source phase.

Compiles csrc/drone_kernel_k.hip (one k_closest value, -DDRONESIM_K) for gfx950 with `-save-temps -gline-tables-only` (line tables do not
change code generation), takes the body of the requested kernel from the device assembly, maps every instruction to
the source line its `.loc` names and buckets the lines by the `// @phase <name>` markers in the kernel source.
Counts are STATIC (a loop body counts once); loop bodies are listed per phase so they can be weighted by hand.

    python tools/isa_hist.py [--kernel drone_kernelILi2ELb0ELi0ELi1EE] [--part 2] [--out profiles/x.md] [--extra=-D...]
"""
import argparse
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "scalable_collision_avoidance_rl_amd", "csrc")
SRC = os.path.join(CSRC, "drone_kernel.hpp")          # the kernel source (phase markers, .loc lines)
TU = os.path.join(CSRC, "drone_kernel_k.hip")         # the translation unit that instantiates it

TRANS = ("v_sqrt", "v_log", "v_rsq", "v_rcp", "v_exp", "v_sin", "v_cos")
CROSS = ("v_readlane", "v_readfirstlane", "v_writelane", "v_permlane", "v_bpermute", "ds_bpermute", "ds_permute", "ds_swizzle")


def classify(op):
    if op.startswith(TRANS):
        return "valu_trans"
    if op.startswith(CROSS):
        return "cross_lane"
    if op.startswith("v_cmp") or op.startswith("v_cmpx"):
        return "valu_cmp"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_load", "buffer_load", "flat_load")):
        return "vmem_load"
    if op.startswith(("global_store", "buffer_store", "flat_store", "global_atomic")):
        return "vmem_store"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith(("s_nop", "s_endpgm", "s_sleep", "s_setprio", "s_code_end")):
        return "other"
    if op.startswith("s_"):
        return "salu"
    return "other"


def phases_of_source():
    """[(first_line, name)] from `// @phase name` markers, ascending."""
    out = []
    for n, line in enumerate(open(SRC), 1):
        m = re.search(r"//\s*@phase\s+(\S+)", line)
        if m:
            out.append((n, m.group(1)))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="drone_kernelILi2ELb0ELi0ELi1E")
    ap.add_argument("--part", default="2", help="k_closest value (-DDRONESIM_K)")
    ap.add_argument("--out", default=None)
    ap.add_argument("--extra", default="")
    ap.add_argument("--title", default=None)
    args = ap.parse_args()
    work = os.path.join(ROOT, "build", "isa")
    os.makedirs(work, exist_ok=True)
    cmd = ["hipcc", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-mllvm", "-amdgpu-kernarg-preload-count=8",
           "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), f"-DDRONESIM_K={args.part}", "-mllvm", "-amdgpu-sched-strategy=max-ilp",
           "-gline-tables-only", "-save-temps", "-c", "-o", "isa_part.o", TU] + ([args.extra] if args.extra else [])
    subprocess.check_call(cmd, cwd=work)
    asm = os.path.join(work, "drone_kernel_k-hip-amdgcn-amd-amdhsa-gfx950.s")
    lines = open(asm).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(rf"^_Z\w*{re.escape(args.kernel)}\w*:", l))
    name = lines[start].split(":")[0]
    file_ids = {}
    for l in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
        if m:
            file_ids[int(m.group(1))] = (m.group(3) or m.group(2))
    src_ids = {k for k, v in file_ids.items() if v.endswith("drone_kernel.hpp")}
    ph = phases_of_source()

    def phase_of(line_no):
        cur = "prologue"
        for first, nm in ph:
            if line_no >= first:
                cur = nm
        return cur

    hist = collections.OrderedDict()
    cur_line, cur_file = 0, None
    n_total = 0
    loops = collections.defaultdict(int)
    for l in lines[start + 1:]:
        s = l.strip()
        if s.startswith(".loc"):
            p = s.split()
            cur_file, cur_line = int(p[1]), int(p[2])
            continue
        if not s or s.startswith((".", ";", "//")) or s.endswith(":"):
            continue
        op = s.split()[0]
        if op == "s_endpgm":
            n_total += 1
            break
        phase = phase_of(cur_line) if cur_file in src_ids else "inlined-header"
        hist.setdefault(phase, collections.Counter())[classify(op)] += 1
        if op.startswith("s_cbranch") and re.search(r"\.LBB\d+_\d+", s):
            loops[phase] += 1
        n_total += 1
    # resource usage
    meta = {}
    for i, l in enumerate(lines):
        if re.match(rf"\s*\.amdhsa_kernel\s+{re.escape(name)}\s*$", l):
            for l2 in lines[i:i + 60]:
                m = re.match(r"\s*\.amdhsa_(next_free_vgpr|next_free_sgpr|accum_offset)\s+(\d+)", l2)
                if m:
                    meta[m.group(1)] = int(m.group(2))
            break
    cols = ["valu", "valu_cmp", "valu_trans", "cross_lane", "salu", "lds", "vmem_load", "vmem_store", "smem", "waitcnt",
            "branch", "barrier", "other"]
    out = []
    out.append(f"# {args.title or 'ISA histogram'}\n")
    out.append(f"kernel `{name}`  (static instruction counts; {n_total} instructions; "
               f"next_free_vgpr {meta.get('next_free_vgpr')}, next_free_sgpr {meta.get('next_free_sgpr')})\n")
    out.append("| phase | " + " | ".join(cols) + " | total | cond. branches |")
    out.append("|---|" + "---|" * (len(cols) + 2))
    tot = collections.Counter()
    order = ["prologue"] + [nm for _, nm in ph if nm in hist] + [k for k in hist if k not in dict((b, a) for a, b in ph) and k != "prologue"]
    seen = set()
    for phase in order:
        if phase in seen or phase not in hist:
            continue
        seen.add(phase)
        c = hist[phase]
        tot.update(c)
        out.append(f"| {phase} | " + " | ".join(str(c.get(k, 0)) for k in cols) + f" | {sum(c.values())} | {loops.get(phase, 0)} |")
    out.append("| **all** | " + " | ".join(str(tot.get(k, 0)) for k in cols) + f" | {sum(tot.values())} | {sum(loops.values())} |")
    text = "\n".join(out) + "\n"
    print(text)
    if args.out:
        with open(os.path.join(ROOT, args.out) if not os.path.isabs(args.out) else args.out, "w") as f:
            f.write(text)


if __name__ == "__main__":
    sys.exit(main())
