#!/usr/bin/env python3
"""Compare the gfx950 code objects of two builds of libdronesim.so kernel by kernel: instruction streams
(llvm-objdump -d, addresses and encodings stripped) must be identical for a refactoring that claims "no code change".

    python tools/co_diff.py before.so after.so"""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_resources import code_objects

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def kernels(path):
    out = {}
    for _, co in code_objects(open(path, "rb").read()):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co); f.flush()
            txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True).stdout
        name = None
        for line in txt.split("\n"):
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                name = m.group(1); out[name] = []
            elif name and line.strip():
                ins = re.sub(r"^\s*[0-9a-f]+:\s*", "", line.split("//")[0]).strip()
                ins = re.sub(r"<[^>]*\+0x[0-9a-f]+>", "<L>", ins)
                if ins:
                    out[name].append(ins)
    return out


def main():
    a, b = kernels(sys.argv[1]), kernels(sys.argv[2])
    same = diff = 0
    for name in sorted(set(a) | set(b)):
        if name not in a or name not in b:
            print("only in", "after " if name in b else "before", name[:110]); diff += 1
        elif a[name] != b[name]:
            n = sum(x != y for x, y in zip(a[name], b[name])) + abs(len(a[name]) - len(b[name]))
            print(f"DIFFERENT ({len(a[name])} vs {len(b[name])} instructions, {n} differing lines)", name[:110]); diff += 1
        else:
            same += 1
    print(f"{same} kernels identical, {diff} different")
    return 1 if diff else 0


if __name__ == "__main__":
    sys.exit(main())
