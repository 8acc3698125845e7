"""Shared helpers for the parity tests (test infrastructure)."""
import glob
import os

import numpy as np

from oracle.oracle import Oracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# fp32 parity bar of BASELINE.md section 4 / north_star: |a-b| <= 1e-5 * max(1, |ref|)
RTOL = 1e-5
ATOL = 1e-5


def atol_coord(G):
    """Absolute tolerance for outputs that are DIFFERENCES OF COORDINATES (z rows: x_j - x_i, x_i - xF_i).
    The state is stored in float32, so each coordinate of magnitude <= G carries up to half an ulp32(G) of
    rounding after the integrator and the difference up to one ulp32(G) -- 4.8e-7 at G=5, 1.9e-6 at G=28 but
    3.1e-5 at G=256, above a flat 1e-5.  The bar is therefore 1e-5 + ulp32(G)."""
    return ATOL + float(np.spacing(np.float32(G)))


# discrete outputs (n_coll, done, neighbour ids) are compared exactly where every
# decision is at least this far from its threshold
MARGIN = 1e-4


def single_step_files():
    return sorted(glob.glob(os.path.join(GOLDEN, "single_step_*.npz")))


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def oracle_for(fx, threads=1):
    """Oracle configured like the reference env that produced fixture `fx`."""
    N, k, c = int(fx["N"]), int(fx["k"]), int(fx["c"])
    G = float(fx["G"])
    o = Oracle(N, [G, G], k_closest=k, deltas=fx["deltas"], simplify_zstate=(c == 2),
               collision_weight=float(fx["collision_weight"]), threads=threads)
    return o


def assert_close(a, b, what, rtol=RTOL, atol=ATOL):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    bad = ~(np.abs(a - b) <= atol + rtol * np.abs(b))
    bad &= ~(np.isnan(a) & np.isnan(b))
    if bad.any():
        idx = np.argwhere(bad)[0]
        raise AssertionError(f"{what}: {bad.sum()} / {bad.size} outside tolerance; first at {tuple(idx)}: "
                             f"got {a[tuple(idx)]!r} want {b[tuple(idx)]!r}")


def z_compare_mask(nbr_idx, row_tie_free, c):
    """Which z entries are defined by the reference independent of argsort tie order.

    Ghost rows (nbr slot == -1) carry v,l of a tie-ordered agent in columns 2..4 when
    the row's sorted prefix contains tied (clipped) entries -> compare only columns 0-1 there."""
    E, N, K1 = nbr_idx.shape
    mask = np.ones((E, N, K1, c), bool)
    if c == 5:
        ghost = nbr_idx < 0
        ghost[..., 0] = False
        tied = ~np.asarray(row_tie_free, bool)[..., None] & ghost
        mask[..., 2:][tied] = False
    return mask
