"""CPU-only: host-side logic of the product package and the C-ABI surface (no compute calls)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import scalable_collision_avoidance_rl_amd as pkg
from scalable_collision_avoidance_rl_amd import _native
from tests import helpers as H


def test_module_constants_match_reference():
    # drone_env.py:27-30
    assert (pkg.dim, pkg.dt, pkg.max_time_steps) == (2, 0.05, 200)


def test_formation_matches_reference_golden():
    fx = H.load("formation.npz")
    for tag in [k[3:] for k in fx.files if k.startswith("xF_")]:
        n, g = tag.split("_")
        grid = [float(x) for x in g.split("x")] if "x" in g else [float(g), float(g)]
        end_points, d_safety = pkg.formation_O(int(n), grid)
        assert end_points.shape == (2 * int(n), 1)                      # column vector, drone_env.py:127
        H.assert_close(end_points.reshape(-1, 2), fx[f"xF_{tag}"], tag, rtol=1e-13, atol=1e-13)
        np.testing.assert_array_equal(d_safety, fx[f"dhat_{tag}"])


def test_clip_deltas_and_warning(capsys):
    d_hat = np.array([2.98] * 4)
    assert pkg.clip_deltas(None, d_hat) is d_hat                        # drone_env.py:85-87
    out = pkg.clip_deltas(np.array([1.0, 3.5, 2.0, 2.98]), d_hat)
    np.testing.assert_array_equal(out, [1.0, 2.98, 2.0, 2.98])
    assert "Some deltas are greater" in capsys.readouterr().out        # drone_env.py:91
    pkg.clip_deltas(np.ones(4), d_hat)
    assert capsys.readouterr().out == ""


def test_lattice_divisions():
    # floor(G / 0.22): SURVEY 8a6 -> 484 nodes at G=5, 16129 at G=28, 1352569 at G=256
    assert pkg.lattice_divisions([5, 5]) == (22, 22)
    assert np.prod(pkg.lattice_divisions([28, 28])) == 16129
    assert np.prod(pkg.lattice_divisions([256, 256])) == 1352569
    assert pkg.lattice_divisions([7, 4]) == (31, 18)


def test_shard_range_partitions_env_axis():
    for E in (1, 7, 4096, 32768):
        for W in (1, 2, 3, 8):
            spans = [pkg.shard_range(E, r, W) for r in range(W)]
            assert spans[0][0] == 0 and spans[-1][1] == E
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        pkg.shard_range(8, 2, 2)


def test_library_loads_and_exports_every_declared_symbol():
    lib = _native.lib()
    header = open(_native.HEADER_PATH).read()
    declared = sorted(set(re.findall(r"\b(dronesim_[a-z0-9_]+)\s*\(", header)))
    assert declared == sorted(_native.SYMBOLS)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.dronesim_version() == 600
    assert lib.dronesim_error_string(0) == b"ok"
    assert b"invalid" in lib.dronesim_error_string(_native.EINVAL)


def test_product_library_exports_the_product_only():
    """The float64 verification variant (test infrastructure) lives in libdronesim_verify.so / include/dronesim_verify.h:
    the product library exports no *_f64 symbol, and the verification library exports what its header declares."""
    import subprocess
    vheader = open(_native.VERIFY_HEADER_PATH).read()
    declared = sorted(set(re.findall(r"\b(dronesim_[a-z0-9_]+)\s*\(", vheader)))
    assert declared == sorted(_native.VERIFY_SYMBOLS)
    vlib = _native.verify_lib()
    for name in declared:
        assert getattr(vlib, name) is not None
    out = subprocess.run(["nm", "-D", "--defined-only", _native.LIB_PATH], capture_output=True, text=True).stdout
    exported = [l.split()[-1] for l in out.splitlines() if l.strip()]
    assert exported and not [n for n in exported if "f64" in n.lower()], [n for n in exported if "f64" in n.lower()]
    assert not hasattr(_native.lib(), "dronesim_step_f64")


def test_params_struct_layout_matches_header():
    header = open(_native.HEADER_PATH).read()
    body = header[header.index("typedef struct DroneParams {"):header.index("} DroneParams;")]
    names = re.findall(r"^\s*(?:const\s+)?(?:int32_t|float)\s*\*?\s*(\w+);", body, re.M)
    assert names == [f[0] for f in _native.DroneParams._fields_]
    assert C.sizeof(_native.DroneParams) == 4 * 4 + 11 * 4 + 4 + 5 * 8     # 4-byte pad before the pointers


def test_episode_structs_match_header():
    """ctypes mirrors of DroneEpisodeAcc / DroneEpisodeCtl (field order, sizes) against include/dronesim.h."""
    header = open(_native.HEADER_PATH).read()
    vheader = open(_native.VERIFY_HEADER_PATH).read()
    for cls, name in ((_native.DroneEpisodeAcc, "DroneEpisodeAcc"), (_native.DroneEpisodeCtl, "DroneEpisodeCtl"),
                      (_native.DroneParamsF64, "DroneParamsF64")):
        header = vheader if name == "DroneParamsF64" else open(_native.HEADER_PATH).read()
        body = header[header.index("typedef struct %s {" % name):header.index("} %s;" % name)]
        names = re.findall(r"^\s*(?:const\s+)?(?:int32_t|int64_t|uint64_t|float|double|DroneEpisodeAcc)\s*\*?\s*(\w+);", body, re.M)
        assert names == [f[0] for f in cls._fields_], name
    header = open(_native.HEADER_PATH).read()
    body = header[header.index("typedef struct DroneStepCall {"):header.index("} DroneStepCall;")]
    names = re.findall(r"^\s*(?:const\s+)?(?:int32_t|uint8_t|float|DroneParams|DroneEpisodeCtl)\s*\*?\s*(\w+);", body, re.M)
    assert names == [f[0] for f in _native.DroneStepCall._fields_] and C.sizeof(_native.DroneStepCall) == 11 * 8 + 8
    assert C.sizeof(_native.DroneEpisodeAcc) == 64 and C.sizeof(_native.DroneEpisodeCtl) == 72
    assert _native.DroneEpisodeAcc.done_return.offset == 32 and _native.DroneEpisodeAcc.ep_len.offset == 20
    assert _native.DroneEpisodeCtl.seed.offset == 24 and _native.DroneEpisodeCtl.episode.offset == 40
    assert _native.DroneEpisodeCtl.z_final.offset == 48 and _native.DroneEpisodeCtl.pos_final.offset == 64


def test_philox_restatement_matches_the_published_known_answers():
    """The oracle's Philox4x32-10 (reset stream, in-kernel RandomAgent actions) against the Random123 known-answer
    vectors (Salmon et al., SC'11, kat_vectors: philox4x32 10 rounds)."""
    from oracle import oracle as O
    assert O.philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert O.philox([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert O.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    assert O.philox_word0(0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344, 0xa4093822, 0x299f31d0) == 0xd16cfe09
    # RandomAgent stream: values on the 2^24-point grid of [-1, 1), exact in float32, pairs (t even / odd) share a block
    a0 = O.rand_actions(4, np.array([6, 7], np.int32), np.array([2, 2], np.int32), 99, env_base=5)
    assert a0.min() >= -1 and a0.max() < 1 and np.array_equal(a0, a0.astype(np.float32).astype(np.float64))
    w = O.philox([1, 5, 3, 2], [99 ^ 0x52414E44, 0])
    assert a0[0, 1, 0] == -1 + (w[0] >> 8) * 2.0 ** -23 and a0[0, 1, 1] == -1 + (w[1] >> 8) * 2.0 ** -23
    w1 = O.philox([1, 6, 3, 2], [99 ^ 0x52414E44, 0])
    assert a0[1, 1, 0] == -1 + (w1[2] >> 8) * 2.0 ** -23 and a0[1, 1, 1] == -1 + (w1[3] >> 8) * 2.0 ** -23


def test_argument_errors_are_reported_without_a_gpu():
    """Argument validation happens before any HIP call (error behaviour of the boundary)."""
    lib = _native.lib()
    p = _native.DroneParams()
    nul = None
    assert lib.dronesim_step(None, nul, nul, nul, nul, nul, nul, nul, nul, nul, nul, 1, nul) == _native.EINVAL
    p.N, p.k, p.c = 5000, 2, 2
    assert lib.dronesim_step(C.byref(p), nul, nul, nul, nul, nul, nul, nul, nul, nul, nul, 1, nul) == _native.EUNSUPPORTED
    p.N, p.k = 5, 5
    assert lib.dronesim_observe(C.byref(p), nul, nul, nul, nul, nul, nul, nul, nul, 1, nul) == _native.EINVAL
    assert b"k_closest" in lib.dronesim_last_error()
    p.k, p.c = 2, 3
    assert lib.dronesim_step(C.byref(p), nul, nul, nul, nul, nul, nul, nul, nul, nul, nul, 1, nul) == _native.EINVAL
    with pytest.raises(_native.DroneSimError):
        _native.check(_native.EINVAL, "unit-test")


def test_f16x2_rowtile_boundary_without_a_gpu():
    """dronesim_mlp_forward_f16x2_rt: argument errors are reported before any HIP call (NULL struct / arrays, a block count that is
    not dronesim_mlp_rt16_blocks, d_in > 16, a misaligned stream); E = 0 is a no-op; nout > 4 needs no w3p."""
    lib = _native.lib()
    fn, nul = lib.dronesim_mlp_forward_f16x2_rt, None
    assert fn(None, nul, nul, nul, nul, 0, 0, 0, nul, nul, 4, nul) == _native.EINVAL
    m = _native.DroneMlpBf16()
    m.N, m.d_in, m.h1, m.h2, m.nout, m.out_kind, m.sample_kind = 2, 6, 40, 72, 4, 0, 0
    one = C.c_void_p(16)
    m.w1p, m.b1, m.b2, m.b3 = one, one, one, one
    m.reserved = lib.dronesim_mlp_rt16_blocks(40, 72, 4)
    assert m.reserved == ((1 * 2 + 2 * 3 + 3) // 4 + 3) * 4
    assert lib.dronesim_mlp_rt16_blocks(40, 72, 16) == ((1 * 2 + 2 * 3 + 3 + 3) // 4 + 3) * 4   # + one L3 block per chunk
    assert fn(C.byref(m), one, nul, nul, nul, 0, 0, 0, nul, nul, 4, nul) == _native.EINVAL and b"NULL weight" in lib.dronesim_last_error()
    m.w3p = one                                                   # (layer 3 reads the plain float32 [N, h2, nout] array here)
    assert fn(C.byref(m), one, nul, nul, nul, 0, 0, 0, nul, nul, 0, nul) == _native.OK
    m.reserved += 4
    assert fn(C.byref(m), one, nul, nul, nul, 0, 0, 0, nul, nul, 4, nul) == _native.EINVAL and b"rt16_blocks" in lib.dronesim_last_error()
    m.reserved -= 4
    m.w1p = C.c_void_p(24)
    assert fn(C.byref(m), one, nul, nul, nul, 0, 0, 0, nul, nul, 4, nul) == _native.EINVAL and b"aligned" in lib.dronesim_last_error()
    m.w1p = one
    m.d_in = 17
    assert fn(C.byref(m), one, nul, nul, nul, 0, 0, 0, nul, nul, 4, nul) == _native.EUNSUPPORTED
    m.d_in = 6
    m.nout, m.w3p, m.reserved = 16, None, lib.dronesim_mlp_rt16_blocks(40, 72, 16)       # nout > 4: layer 3 rides in the stream
    assert fn(C.byref(m), one, nul, nul, nul, 0, 0, 0, nul, nul, 0, nul) == _native.OK
    assert lib.dronesim_mlp_rt16_blocks(0, 5, 1) == 0


def test_split_policy_boundary_without_a_gpu():
    """dronesim_mlp_forward_bf16x3 / _f16x2: argument errors are reported before any HIP call; the stage count of the
    weight streams; `pack_split_streams` lays the fragments out in the order include/dronesim.h states, and the parts
    of both schemes add up to the float32 weights (exactly for bf16x3, to 2^-22 for f16x2)."""
    import torch
    from scalable_collision_avoidance_rl_amd import policies as P
    lib = _native.lib()
    nul = None
    for fn in (lib.dronesim_mlp_forward_bf16x3, lib.dronesim_mlp_forward_f16x2):
        assert fn(None, nul, nul, nul, nul, 0, 0, 0, nul, nul, 4, nul) == _native.EINVAL
        m = _native.DroneMlpBf16()
        m.N, m.d_in, m.h1, m.h2, m.nout, m.out_kind, m.sample_kind = 2, 6, 40, 72, 4, 0, 0
        one = C.c_void_p(16)                                      # any non-NULL address: validation never dereferences
        m.w1p, m.b1, m.b2, m.b3 = one, one, one, one
        m.reserved = 0
        assert fn(C.byref(m), one, nul, nul, nul, 0, 0, 0, nul, nul, 4, nul) == _native.EINVAL
        assert b"stages" in lib.dronesim_last_error()
        m.reserved = lib.dronesim_mlp_bf16x3_stages(40, 72)
        assert fn(C.byref(m), one, nul, nul, nul, 0, 0, 0, nul, nul, 0, nul) == _native.OK       # E = 0: nothing to do
        m.d_in = 17
        assert fn(C.byref(m), one, nul, nul, nul, 0, 0, 0, nul, nul, 4, nul) == _native.EUNSUPPORTED
    # h1 = 40 -> 2 chunks, h2 = 72 -> 3 chunks: waves 0..2 own one chunk, wave 3 none; wave 0's stream is
    # W1(0) W2(0,0) W1(1) W2(0,1) W2(1,0) W2(1,1) W3(0,0) W3(0,1) + 8 stages of padding
    stages = lib.dronesim_mlp_bf16x3_stages(40, 72)
    assert stages == 2 * (1 + 2 * 1) + 2 * 1 + 8
    g = torch.Generator().manual_seed(5)
    w1, w2, w3 = (torch.rand(*s, generator=g) * 2 - 1 for s in ((2, 6, 40), (2, 40, 72), (2, 72, 4)))
    for scheme, nparts, dtype in (("bf16x3", 3, torch.bfloat16), ("f16x2", 2, torch.float16)):
        st = P.pack_split_streams(w1, w2, w3, stages, scheme)
        assert tuple(st.shape) == (2, 4, stages, nparts, 64, 8) and st.dtype == dtype
        f1 = P.pack_split_fragments(w1, 1, 2, "linear", scheme)           # [N, c1, 1, P, 64, 8]
        f2 = P.pack_split_fragments(w2, 4, 3, "accumulator", scheme)      # [N, c2, s, P, 64, 8]
        f3 = P.pack_split_fragments(w3, 6, 1, "accumulator", scheme)      # [N, 1, s, P, 64, 8]
        for w in range(3):
            want = [f1[:, 0, 0], f2[:, w, 0], f1[:, 1, 0], f2[:, w, 1], f2[:, w, 2], f2[:, w, 3], f3[:, 0, 2 * w], f3[:, 0, 2 * w + 1]]
            for j, fr in enumerate(want):
                assert torch.equal(st[:, w, j], fr), (scheme, w, j)
            assert not st[:, w, len(want):].float().abs().sum()           # zero padding
        assert torch.equal(st[:, 3, 1], f1[:, 1, 0]) and not st[:, 3, 2:].float().abs().sum()   # wave 3 owns no chunk (the kernel skips it)
        total = f2.float().sum(dim=3)                                      # parts add up to the weights, fragment by fragment
        ref = P.pack_bf16_fragments(w2, 4, 3, "accumulator", dtype=torch.float32)
        err = (total - ref).abs().max().item()
        assert err == 0.0 if scheme == "bf16x3" else err < 2.0 ** -21


def test_mlp_forward_validates_w2_layout_before_any_hip_call():
    lib = _native.lib()
    m = _native.DroneMlp()
    m.N, m.d_in, m.h1, m.h2, m.nout, m.out_kind, m.sample_kind = 2, 6, 40, 72, 4, 0, 0
    one = C.c_void_p(16)                                          # any non-NULL address: validation never dereferences
    m.w1 = m.b1 = m.w2 = m.b2 = m.w3 = m.b3 = one
    for layout, want in ((0, _native.OK), (1, _native.OK), (2, _native.OK), (3, _native.EINVAL), (-1, _native.EINVAL)):
        m.w2_layout = layout
        assert lib.dronesim_mlp_forward(C.byref(m), one, None, None, None, 0, 0, 0, None, None, 0, None) == want   # E = 0
    assert b"w2_layout" in lib.dronesim_last_error()
    m.w2_layout, m.d_in = 2, 15                                   # the row-tile stream holds at most 14 inputs
    assert lib.dronesim_mlp_forward(C.byref(m), one, None, None, None, 0, 0, 0, None, None, 0, None) == _native.EUNSUPPORTED
    m.d_in, m.w2 = 6, C.c_void_p(20)                              # ... and must be 16-byte aligned
    assert lib.dronesim_mlp_forward(C.byref(m), one, None, None, None, 0, 0, 0, None, None, 0, None) == _native.EINVAL


def test_pack_f32_rowtile_stream_layout():
    """`pack_f32_rowtile_stream` lays all three layers out as include/dronesim.h states for DroneMlp.w2_layout = 2, block by block in
    the kernel's consumption order, and `dronesim_mlp_rt_blocks` counts the same blocks."""
    import torch
    from scalable_collision_avoidance_rl_amd import policies as P
    lib = _native.lib()
    g = torch.Generator().manual_seed(5)
    for (n, d, h1, h2, no) in ((2, 6, 70, 250, 4), (1, 14, 32, 32, 16), (2, 3, 200, 200, 1), (1, 6, 400, 400, 4), (1, 5, 33, 449, 32)):
        w1, b1 = torch.rand(n, d, h1, generator=g), torch.rand(n, h1, generator=g)
        w2, w3 = torch.rand(n, h1, h2, generator=g), torch.rand(n, h2, no, generator=g)
        st = P.pack_f32_rowtile_stream(w1, b1, w2, w3)
        nc1, nc2 = (h1 + 31) // 32, (h2 + 31) // 32
        passes = (nc2 + P.RT_CHUNKS - 1) // P.RT_CHUNKS
        per = (nc2 + passes - 1) // passes
        assert st.shape == (n, int(lib.dronesim_mlp_rt_blocks(h1, h2, no)), 4, 64, 4) and st.dtype == torch.float32 and st.is_contiguous()
        el = lambda t, a, k, c: float(t[a, k, c]) if k < t.shape[1] and c < t.shape[2] else 0.0
        rng = np.random.default_rng(1)
        blk = 0
        for p in range(passes):
            chunks = range(p * per, min(nc2, (p + 1) * per))
            for c1 in range(nc1):
                B = st[:, blk]; blk += 1
                for _ in range(12):
                    a, lane, r = int(rng.integers(0, n)), int(rng.integers(0, 64)), int(rng.integers(0, 7))
                    half, i = lane >> 5, lane & 31
                    piece, j = (0, r) if r < 4 else (1, r - 4)
                    assert float(B[a, piece, lane, j]) == el(w1, a, 2 * r + half, 32 * c1 + i)
                    assert float(B[a, 1, lane, 3]) == (float(b1[a, 32 * c1 + i]) if half == 0 and 32 * c1 + i < h1 else 0.0)
                    assert float(B[a, 2 + (r & 1), lane, j & 3]) == 0.0
                for c2 in chunks:
                    B = st[:, blk]; blk += 1
                    for _ in range(12):
                        a, lane, q, j = (int(rng.integers(0, m)) for m in (n, 64, 4, 4))
                        assert float(B[a, q, lane, j]) == el(w2, a, 32 * c1 + 8 * q + 4 * (lane >> 5) + j, 32 * c2 + (lane & 31))
            for c2 in (chunks if no > 4 else ()):                  # (nout <= 4: no L3 blocks, layer 3 reads the plain w3 array)
                B = st[:, blk]; blk += 1
                for _ in range(12):
                    a, lane, q, j = (int(rng.integers(0, m)) for m in (n, 64, 4, 4))
                    assert float(B[a, q, lane, j]) == el(w3, a, 32 * c2 + 8 * q + 4 * (lane >> 5) + j, lane & 31)
        assert blk + P.RT_PAD == st.shape[1] and float(st[:, blk:].abs().max()) == 0.0


def test_pack_f16_rowtile_stream_layout():
    """`pack_f16_rowtile_stream` (the weight stream of dronesim_mlp_forward_f16x2_rt): blocks of four 1-KiB float16 pieces in the
    kernel's consumption order -- per pass, per in-chunk c1: (W1 hi, W1 lo, 0, 0), then per out-chunk of the pass (hi, lo of k-step
    2 c1; hi, lo of k-step 2 c1 + 1) in the accumulator's k order; nout > 4: the pass ends with one such block of W3 per out-chunk;
    hi + lo add up to the float32 weight to 2^-22; zero blocks up to `dronesim_mlp_rt16_blocks`, which is a multiple of 4 (the ring opens
    four blocks at a time) with >= 12 trailing zero blocks."""
    import torch
    from scalable_collision_avoidance_rl_amd import policies as P
    lib = _native.lib()
    g = torch.Generator().manual_seed(6)
    for (n, d, h1, h2, no) in ((2, 6, 70, 250, 4), (1, 14, 32, 32, 1), (2, 3, 200, 200, 16), (1, 6, 400, 400, 2), (1, 16, 33, 449, 32)):
        w1, w2 = torch.rand(n, d, h1, generator=g) - 0.5, torch.rand(n, h1, h2, generator=g) - 0.5
        w3 = torch.rand(n, h2, no, generator=g) - 0.5
        blocks = int(lib.dronesim_mlp_rt16_blocks(h1, h2, no))
        st = P.pack_f16_rowtile_stream(w1, w2, blocks, w3 if no > 4 else None)
        nc1, nc2 = (h1 + 31) // 32, (h2 + 31) // 32
        passes = (nc2 + P.RT_CHUNKS - 1) // P.RT_CHUNKS
        per = (nc2 + passes - 1) // passes
        real = passes * nc1 + nc1 * nc2 + (nc2 if no > 4 else 0)
        assert blocks % 4 == 0 and blocks == ((real + 3) // 4 + 3) * 4
        assert st.shape == (n, blocks, 4, 64, 8) and st.dtype == torch.float16 and st.is_contiguous()
        el = lambda t, a, k, c: float(t[a, k, c]) if k < t.shape[1] and c < t.shape[2] else 0.0
        close = lambda got, want: abs(got - want) <= 2.0 ** -21 * abs(want) + 2.0 ** -30
        rng = np.random.default_rng(2)
        blk = 0
        for p in range(passes):
            chunks = range(p * per, min(nc2, (p + 1) * per))
            for c1 in range(nc1):
                B = st[:, blk].float(); blk += 1
                assert float(B[:, 2:].abs().max()) == 0.0
                for _ in range(12):
                    a, lane, j = (int(rng.integers(0, m)) for m in (n, 64, 8))
                    assert close(float(B[a, 0, lane, j]) + float(B[a, 1, lane, j]), el(w1, a, 8 * (lane >> 5) + j, 32 * c1 + (lane & 31)))   # "linear" k order
                for c2 in chunks:
                    B = st[:, blk].float(); blk += 1
                    for _ in range(12):
                        a, lane, s, j = (int(rng.integers(0, m)) for m in (n, 64, 2, 8))
                        k = 32 * c1 + 16 * s + 8 * (j >> 2) + 4 * (lane >> 5) + (j & 3)                  # "accumulator" k order
                        assert close(float(B[a, 2 * s, lane, j]) + float(B[a, 2 * s + 1, lane, j]), el(w2, a, k, 32 * c2 + (lane & 31)))
            for c2 in (chunks if no > 4 else ()):
                B = st[:, blk].float(); blk += 1
                for _ in range(12):
                    a, lane, s, j = (int(rng.integers(0, m)) for m in (n, 64, 2, 8))
                    k = 32 * c2 + 16 * s + 8 * (j >> 2) + 4 * (lane >> 5) + (j & 3)
                    assert close(float(B[a, 2 * s, lane, j]) + float(B[a, 2 * s + 1, lane, j]), el(w3, a, k, lane & 31))
        assert blk == real and float(st[:, blk:].float().abs().max()) == 0.0


def test_pack_f32_fragments_layout():
    """`pack_f32_fragments` lays layer 2's float32 weights out as include/dronesim.h states for DroneMlp.w2_layout = 1:
    frag[a][c][s][q][l][j] = W[a][16 s + 8 (l >> 5) + 4 q + j][32 c + (l & 31)], zero beyond K / F."""
    import torch
    from scalable_collision_avoidance_rl_amd import policies as P
    g = torch.Generator().manual_seed(3)
    for (n, k, f) in ((2, 40, 72), (1, 16, 32), (3, 5, 1), (1, 400, 400)):
        w = torch.rand(n, k, f, generator=g)
        fr = P.pack_f32_fragments(w)
        ns, nc = (k + 15) // 16, (f + 31) // 32
        assert fr.shape == (n, nc, ns, 2, 64, 4) and fr.dtype == torch.float32 and fr.is_contiguous()
        rng = np.random.default_rng(0)
        for _ in range(200):
            a, c, s_, q, l, j = (int(rng.integers(0, m)) for m in (n, nc, ns, 2, 64, 4))
            kk, col = 16 * s_ + 8 * (l >> 5) + 4 * q + j, 32 * c + (l & 31)
            want = float(w[a, kk, col]) if kk < k and col < f else 0.0
            assert float(fr[a, c, s_, q, l, j]) == want
        assert float(fr.sum()) == pytest.approx(float(w.sum()), rel=1e-5)


def test_env_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pkg.drones(5, 0, [5, 5], "O", deltas=np.ones(5), simplify_zstate=True)


def test_product_package_never_imports_the_oracle():
    root = os.path.dirname(os.path.abspath(pkg.__file__))
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "libdrone_oracle" not in src, f


def test_bf16_fragment_packing_layouts():
    """pack_bf16_fragments against the index formulas of include/dronesim.h (both k orders), on CPU tensors."""
    import torch
    from scalable_collision_avoidance_rl_amd.policies import pack_bf16_fragments
    g = torch.Generator().manual_seed(0)
    n, k, f, ks, nc = 2, 40, 50, 3, 2                      # ragged: K = 40 < 48, F = 50 < 64
    w = torch.randint(-100, 100, (n, k, f), generator=g).float()       # exactly representable in bf16
    for order, kmap in (("linear", lambda s, h, j: 16 * s + 8 * h + j),
                        ("accumulator", lambda s, h, j: 16 * s + 8 * (j >> 2) + 4 * h + (j & 3))):
        frag = pack_bf16_fragments(w, ks, nc, k_order=order).float().numpy()
        assert frag.shape == (n, nc, ks, 64, 8)
        for a, c, s, l, j in [(0, 0, 0, 0, 0), (1, 1, 2, 63, 7), (0, 1, 1, 37, 5), (1, 0, 2, 31, 6), (0, 1, 2, 40, 3)]:
            kk, ff = kmap(s, l >> 5, j), 32 * c + (l & 31)
            want = float(w[a, kk, ff]) if kk < k and ff < f else 0.0
            assert frag[a, c, s, l, j] == want, (order, a, c, s, l, j)
        # every weight appears exactly once
        assert np.isclose(np.abs(frag).sum(), float(w.abs().sum()))


def test_no_step_kernel_spills():
    """Registers / scratch of every instantiation of the BUILT library (tools/kernel_resources.py reads the gfx950 code
    objects): the single-step and observe kernels -- the BASELINE configs are k = 2 -- use no scratch at all (a spill on
    the hot path is a silent 2x; a register-budget change must not introduce one; this test caught two in round 4), bar the
    k >= 7 episode-layer kernels of N > 256, and the headline kernels fit their occupancy targets."""
    import shutil
    from tools import kernel_resources as KR
    if not os.path.exists(KR.READELF) and not shutil.which(KR.READELF):
        pytest.skip("llvm-readelf not available")
    lib = os.path.join(os.path.dirname(os.path.abspath(pkg.__file__)), "libdronesim.so")
    rows = [r for r in KR.resources(lib) if r[1] is not None]
    assert len(rows) >= 200                                   # 8 k x geometries x modes x episode layer
    step = [r for r in rows if r[1]["mode"] in (0, 1)]
    spilled = [(r[0], r[4]) for r in step if r[4] != 0]
    # the only step kernels that touch scratch: N > 256 (1024-thread workgroups: 128 registers are the hard cap there) with
    # the episode layer at k >= 7, a few dwords
    assert all(r[1]["k"] >= 7 and r[1]["geo"] == 3 and r[1]["epi"] == 1 and r[4] <= 64 for r in step if r[4] != 0), spilled
    by = {(r[1]["k"], r[1]["far"], r[1]["mode"], r[1]["geo"], r[1]["epi"]): r for r in rows}
    # C3's graded kernel (kSym64 step, episode layer) and its plain form: 8 waves per SIMD = at most 64 VGPRs
    assert by[(2, 0, 0, 1, 1)][2] <= 64 and by[(2, 0, 0, 1, 0)][2] <= 64
    # C5's kernels (workgroup-per-env, N <= 256): plain 8 waves per SIMD, episode layer 6 (<= 80 VGPRs)
    assert by[(2, 0, 0, 2, 0)][2] <= 64 and by[(2, 0, 0, 2, 1)][2] <= 80


def test_no_rollout_kernel_spills_on_its_hot_path():
    """Fused rollouts of the BUILT library at k = 2 without far agents for every BASELINE shape -- kPacked (C2), kSym64
    (C3 / C4), kBlockU256 (C5) -- with and without the episode layer, and the plain kBlock256 (65 ... 255 agents): NO scratch
    instruction on the per-step path (tools/spill_sites.py: inside the largest loop of the ISA, outside the out-of-line
    in-kernel reset that `s_nop 13` / `s_nop 14` bracket).  What scratch an episode-layer kernel does use sits in that
    reset (once per episode and env).  (kBlock256 WITH the episode layer keeps 18 scratch accesses per step on purpose:
    spill-free at 168 registers it measured 3 ... 13 % slower -- csrc/drone_kernel.hpp, kPackedRolloutEpiWaves.)"""
    import shutil
    from tools import kernel_resources as KR
    from tools import spill_sites as SS
    if not os.path.exists(KR.READELF) and not shutil.which(KR.READELF):
        pytest.skip("llvm-readelf not available")
    lib = os.path.join(os.path.dirname(os.path.abspath(pkg.__file__)), "libdronesim.so")
    for geo in (0, 1, 2, 4):
        for epi in ((0,) if geo == 2 else (0, 1)):
            # (kSym64 / kBlockU256 with the episode layer: one kernel per action source, MODE 3 = pool, 4 = in-kernel)
            for mode in ((3, 4) if (epi and geo in (1, 4)) else (2,)):
                r = SS.hot_loop_scratch(lib, 2, 0, mode, geo, epi)
                assert r is not None, (geo, epi, mode)
                n_ins, loop, hot, total = r
                assert loop is not None and loop[1] - loop[0] > 500, (geo, epi, mode, loop)      # the per-step loop was found
                assert hot == 0, f"GEO={geo} EPI={epi} MODE={mode}: {hot} scratch instructions inside the per-step loop {loop}"
                if geo in (1, 4) and epi == 0:
                    assert total == 0, (geo, epi, total)
                if geo == 1:                                  # kSym64 (bench.py's fused_rollout lines): no scratch anywhere
                    assert total == 0, (geo, epi, mode, total)


def test_policy_kernels_keep_their_occupancy():
    """The batched policy kernels of the BUILT library: no scratch, and the register budgets their launch geometry
    assumes -- the exact-f32 and split kernels run two workgroups of four waves per CU (<= 256 registers per wave, which
    is also what makes hipcc pick the VGPR form of the matrix instructions: `__launch_bounds__(256, 2)`), the exact-f32
    instance of the reference's observation width (d_in <= 6) leaves room for a third workgroup (<= 168) where LDS
    allows it (h <= 200), the plain-bf16 kernels three to four (<= 168 / <= 128)."""
    import shutil
    from tools import kernel_resources as KR
    if not os.path.exists(KR.READELF) and not shutil.which(KR.READELF):
        pytest.skip("llvm-readelf not available")
    lib = os.path.join(os.path.dirname(os.path.abspath(pkg.__file__)), "libdronesim.so")
    rows = [r for r in KR.resources(lib) if "mlp3" in r[0]]
    f32 = [r for r in rows if "mlp3_kernel" in r[0]]
    split = [r for r in rows if "mlp3_split_kernel" in r[0]]
    bf16 = [r for r in rows if "mlp3_bf16_kernel" in r[0]]
    assert len(f32) == 5 and len(split) == 2 and len(bf16) == 16, [r[0] for r in rows]
    assert all(r[4] == 0 for r in rows), [(r[0], r[4]) for r in rows if r[4]]
    assert all(r[2] <= 256 for r in f32 + split), [(r[0], r[2]) for r in f32 + split]
    assert all(r[2] <= 168 for r in f32 if "ILb1ELb1ELi3E" in r[0]), [(r[0], r[2]) for r in f32]
    for r in bf16:
        nc1 = int(r[0].split("mlp3_bf16_kernelILi")[1].split("E")[0])
        assert r[2] <= (128 if nc1 <= 8 else 168), (r[0], r[2])


def test_bench_refuses_a_multi_gpu_run_it_cannot_start():
    """`python bench.py --gpus N` with fewer than N visible GPUs (none in the build container) exits non-zero with a
    message instead of printing a one-rank line."""
    import subprocess
    import sys
    import torch
    n = torch.cuda.device_count() + 2
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BENCH_ONE_DEVICE")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", "3"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "refusing" in r.stderr
    assert not any(l.startswith("{") for l in r.stdout.splitlines())


def test_f16x2_weight_factors_are_powers_of_two_that_keep_low_parts_normal():
    """policies.f16_weight_scales (DroneMlpBf16.wscale, include/dronesim.h): per (agent, layer) a power of two that puts the
    layer's largest weight into [2^13, 2^14) -- inside the float16 range, with the low part of every weight above
    2^-3 x (largest / 2^14) a NORMAL float16; an all-zero layer gets 1.  And the ctypes mirror carries the field."""
    import torch
    from scalable_collision_avoidance_rl_amd.policies import f16_weight_scales, split2_f16
    g = torch.Generator().manual_seed(3)
    w1 = (torch.rand(3, 6, 40, generator=g) * 2 - 1) * 0.08
    w2 = (torch.rand(3, 40, 40, generator=g) * 2 - 1) * torch.tensor([1e-6, 0.3, 900.0])[:, None, None]
    w3 = torch.zeros(3, 40, 4)
    sc = f16_weight_scales(w1, w2, w3)
    assert sc.shape == (3, 3) and sc.dtype == torch.float32
    m, e = torch.frexp(sc)
    assert torch.all(m == 0.5) and torch.all(sc[:, 2] == 1.0)                 # exact powers of two; zero layer -> 1
    for l, w in enumerate((w1, w2)):
        top = (w.abs().flatten(1).amax(1) * sc[:, l])
        assert torch.all(top >= 2.0 ** 13) and torch.all(top < 2.0 ** 14)
    # what the factor buys: relative error of hi + lo over the weights of a layer of the reference's size
    w = w1[0]
    err = lambda x: float(((sum(split2_f16(x)).double() - x.double()).abs() / x.double().abs().clamp_min(1e-30))[x.abs() > 1e-3].max())
    assert err(w) > 2.0 ** -19 and err(w * sc[0, 0]) <= 2.0 ** -21
    header = open(_native.HEADER_PATH).read()
    body = header[header.index("typedef struct DroneMlpBf16 {"):header.index("} DroneMlpBf16;")]
    assert "const float *wscale;" in body and _native.DroneMlpBf16._fields_[-1][0] == "wscale"
    assert C.sizeof(_native.DroneMlpBf16) == 8 * 4 + 7 * 8
