#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes (run once per counter):

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT/fetch -o p -- python tools/pmc_run.py c3
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d OUT/write -o p -- python tools/pmc_run.py c3
    python tools/pmc_parse.py OUT c3 > profiles/<round>_<cfg>_pmc_traffic.json

It launches calibration copies of KNOWN byte counts in the step kernel's access shapes (8 B/lane and
16 B/lane streams, 24 B/lane + 12 B/lane strided stores) and then `n` step launches of the workload."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from scalable_collision_avoidance_rl_amd import drones
from tools.kbench import PRESETS

CALIB_BYTES = (20 * 2 ** 20, 2 ** 30)      # cache-resident (like one step's 20 MB) and HBM-sized


def main():
    spec = sys.argv[1] if len(sys.argv) > 1 else "c3"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    N, E, G, delta = PRESETS[spec]
    lib = C.CDLL(os.path.join(ROOT, "tools", "libcalib.so"))
    lib.calib_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    lib.calib_write.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for nbytes in CALIB_BYTES:
        src = torch.ones(nbytes // 4, dtype=torch.float32, device="cuda")
        dst = torch.empty_like(src)
        torch.cuda.synchronize()
        for width in (8, 16):
            for _ in range(3):
                assert lib.calib_copy(src.data_ptr(), dst.data_ptr(), nbytes, width, st) == 0
        torch.cuda.synchronize()
        del src, dst
    na = 64 * 4096
    z = torch.empty(na * 6, dtype=torch.float32, device="cuda"); nb = torch.empty(na * 3, dtype=torch.int32, device="cuda")
    for _ in range(3):
        assert lib.calib_write(z.data_ptr(), nb.data_ptr(), na, st) == 0
    torch.cuda.synchronize()
    layer = not os.environ.get("PLAIN")            # the graded path: per-step statistic + in-kernel auto-reset
    env = drones(N, 0, [G, G], "O", deltas=np.ones(N) * delta, simplify_zstate=True, n_envs=E, batched=True, seed=1,
                 auto_reset=layer, track_episodes=layer)
    g = torch.Generator(device="cuda").manual_seed(0)
    pool = torch.rand(n, E, N, 2, device="cuda", generator=g) * 2 - 1
    for s in range(n):
        env.step(pool[s])
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
