#!/usr/bin/env python3
"""Developer probe: per-launch time (hipGraph of 200 dependent launches) of trivial kernels with the
step kernel's grid shape and I/O volume -- the floor any step kernel at that shape can reach."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "tools", "libcalib.so"))
lib.probe_launch_empty.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
lib.probe_launch_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]


def timeit(fn, steps=200, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(steps):
            fn()
    g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / steps * 1e3)
    return float(np.median(ts))


def main():
    out = torch.zeros(1 << 20, dtype=torch.int32, device="cuda")
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for blocks, threads in [(64, 64), (4096, 64), (1024, 256), (512, 512), (32768, 64), (8192, 256)]:
        us = timeit(lambda: lib.probe_launch_empty(out.data_ptr(), blocks, threads, st()))
        print(f"empty  {blocks:6d} x {threads:4d}: {us:6.2f} us/launch", flush=True)
    n = 64 * 4096
    a = torch.rand(n, 2, device="cuda"); b = torch.rand(n, 2, device="cuda")
    o = torch.empty(8, n, 2, device="cuda")
    for blocks, threads in [(4096, 64), (1024, 256)]:
        for nout in (1, 4, 7):       # 7 x 8 B = 56 B written per agent (step writes 60), 16 B read
            for nt in (0, 1):
                us = timeit(lambda: lib.probe_launch_stream(a.data_ptr(), b.data_ptr(), o.data_ptr(), blocks, threads, nout, nt, st()))
                mb = n * (16 + 8 * nout) / 1e6
                print(f"stream {blocks:6d} x {threads:4d} nout={nout} nt={nt}: {us:6.2f} us/launch  {mb:5.1f} MB -> {mb/us*1e-3*1e3:7.1f} GB/s", flush=True)


if __name__ == "__main__":
    main()
