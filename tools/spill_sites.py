#!/usr/bin/env python3
"""Where does a kernel of a built libdronesim.so touch scratch?  Lists every scratch_load / scratch_store of the chosen
drone_kernel instantiations with the loops (backward branches of the ISA) that enclose it, so that "spills" can be split
into hot-path ones (inside the per-step loop of a fused rollout, outside any rarely-taken region) and cold ones.

    python tools/spill_sites.py [lib.so] K FAR MODE GEO EPI"""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_resources import code_objects

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def disasm(path, pat):
    for _, co in code_objects(open(path, "rb").read()):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co); f.flush()
            txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True).stdout
        name, cur = None, {}
        for line in txt.split("\n"):
            m = re.match(r"^([0-9a-f]+) <(.+)>:$", line)
            if m:
                name = m.group(2)
                if pat in name:
                    cur[name] = []
                continue
            if name in cur and line.strip():
                m = re.match(r"^\s*([0-9a-f]+):\s*(.*)$", line.split("//")[0])
                if not m:   # objdump prints "  s_xxx   // 000000001234: ..." with --no-show-raw-insn
                    m2 = re.match(r"^\s*(\S.*?)\s*//\s*([0-9A-Fa-f]+):", line)
                    if m2:
                        cur[name].append((int(m2.group(2), 16), m2.group(1).strip()))
                    continue
                cur[name].append((int(m.group(1), 16), m.group(2).strip()))
        for k, v in cur.items():
            yield k, v


def main():
    args = sys.argv[1:]
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scalable_collision_avoidance_rl_amd", "libdronesim.so")
    if args and args[0].endswith(".so"):
        lib = args.pop(0)
    K, FAR, MODE, GEO, EPI = args[:5]
    pat = f"drone_kernelILi{K}ELb{FAR}ELi{MODE}ELi{GEO}ELb{EPI}E"
    for name, ins in disasm(lib, pat):
        addr = [a for a, _ in ins]
        loops = []                                           # (target index, branch index) of every backward branch
        for i, (a, t) in enumerate(ins):
            m = re.match(r"s_cbranch_\w+\s+(\d+)|s_branch\s+(\d+)", t)
            if m:
                off = int(m.group(1) or m.group(2))
                if off >= 32768:
                    off -= 65536
                tgt = a + 4 + 4 * off
                if tgt <= a and tgt in addr:
                    loops.append((addr.index(tgt), i))
        print(name[:80], len(ins), "instructions;", len(loops), "loops")
        for i, (a, t) in enumerate(ins):
            if t.startswith("scratch_"):
                enc = sorted((j - s, s, j) for s, j in loops if s <= i <= j)
                print(f"  [{i:5d}] {t:60s} loops: " + ", ".join(f"{s}-{j} ({n + 1} ins)" for n, s, j in enc))


if __name__ == "__main__":
    main()


def hot_loop_scratch(lib, K, FAR, MODE, GEO, EPI):
    """(instructions, (start, end) of the largest loop by layout = the per-step loop of a fused rollout, scratch
    instructions inside it but outside the out-of-line in-kernel reset, scratch instructions in all) for one
    drone_kernel instantiation, or None if absent."""
    pat = f"drone_kernelILi{K}ELb{FAR}ELi{MODE}ELi{GEO}ELb{EPI}E"
    for name, ins in disasm(lib, pat):
        addr = {a: i for i, (a, _) in enumerate(ins)}
        loops = []
        for i, (a, t) in enumerate(ins):
            m = re.match(r"s_cbranch_\w+\s+(\d+)|s_branch\s+(\d+)", t)
            if m:
                off = int(m.group(1) or m.group(2))
                if off >= 32768:
                    off -= 65536
                tgt = a + 4 + 4 * off
                if tgt <= a and tgt in addr:
                    loops.append((addr[tgt], i))
        if not loops:
            return len(ins), None, 0, sum(t.startswith("scratch_") for _, t in ins)
        s, e = max(loops, key=lambda l: l[1] - l[0])
        sc = [i for i, (_, t) in enumerate(ins) if t.startswith("scratch_")]
        # the out-of-line in-kernel reset of the episode layer is bracketed by `s_nop 13` / `s_nop 14` in the source
        cb = [i for i, (_, t) in enumerate(ins) if re.fullmatch(r"s_nop 13", t)]
        ce = [i for i, (_, t) in enumerate(ins) if re.fullmatch(r"s_nop 14", t)]
        # (a block laid out behind the loop ends with the jump back to the loop header: its end marker then sits at the top
        # of the loop, BEFORE the begin marker in layout order -- the block runs to the end of the function)
        cold = None if not cb else (min(cb), max(ce)) if ce and min(cb) < max(ce) else (min(cb), len(ins))
        hot = [i for i in sc if s <= i <= e and not (cold and cold[0] <= i <= cold[1])]
        return len(ins), (s, e), len(hot), len(sc)
    return None
