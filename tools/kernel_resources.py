#!/usr/bin/env python3
"""Registers / scratch / LDS of every kernel in a built libdronesim.so (or object file): finds the gfx950 code objects in
the clang offload bundles and reads their metadata notes with llvm-readelf.

    python tools/kernel_resources.py [lib.so] [--scratch-only]"""
import os
import re
import struct
import subprocess
import sys
import tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(blob):
    pos = 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            return
        n, = struct.unpack_from("<Q", blob, pos + len(MAGIC))
        q = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, q)
            triple = blob[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if "gfx" in triple and size:
                yield triple, blob[pos + off:pos + off + size]
        pos = q


def resources(path):
    """[(short name, dict(k, far, mode, geo, epi) or None, vgpr, sgpr, scratch bytes, static LDS bytes)] of every kernel."""
    blob = open(path, "rb").read()
    rows = []
    for triple, co in code_objects(blob):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co); f.flush()
            txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
        for blk in txt.split("  - .agpr_count")[1:]:
            g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
            name = g("name")
            m = re.search(r"drone_kernelILi(\d)ELb(\d)ELi(\d)ELi(\d)ELb(\d)", name)
            short = ("drone_kernel<K=%s,FAR=%s,MODE=%s,GEO=%s,EPI=%s>" % m.groups()) if m else name[:60]
            inst = dict(zip(("k", "far", "mode", "geo", "epi"), map(int, m.groups()))) if m else None
            num = lambda k: int(g(k)) if g(k).isdigit() else -1
            rows.append((short, inst, num("vgpr_count"), num("sgpr_count"), num("private_segment_fixed_size"),
                         num("group_segment_fixed_size")))
    return rows


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    path = args[0] if args else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                             "scalable_collision_avoidance_rl_amd", "libdronesim.so")
    scratch_only = "--scratch-only" in sys.argv
    for short, _, vgpr, sgpr, scratch, lds in resources(path):
        if scratch_only and scratch == 0:
            continue
        print(f"{short:48s} vgpr {vgpr:>4d} sgpr {sgpr:>4d} scratch {scratch:>5d} static-lds {lds:>6d}")


if __name__ == "__main__":
    main()
