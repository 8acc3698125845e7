"""Batched per-agent policies and critics.

The reference keeps one small torch module per agent and evaluates them one by one in Python every step
(SAC_agents.py:170-180: ``actors[i].sample_action(z_states[i].flatten(), N[i])``).  `BatchedMLP` stacks the
N networks' weights and evaluates all of them on the batched observation ``z [E,N,d_in]`` in ONE launch
of the HIP kernel in csrc/policy.hip (exact float32 on the matrix cores), including the sampling of
``sample_action``.  Architectures mirrored:

  DiscreteSoftmaxNN  utils.py:255-309   BatchedMLP.from_discrete_softmax(modules)
  NormalActorNN      utils.py:55-117    BatchedMLP.from_normal_actor(modules)
  CriticNN           utils.py:14-53     BatchedMLP.from_critic(modules)
  saved ``*.pth`` lists               BatchedMLP.from_reference_file(path)

`modules` are any objects exposing the reference's attribute names (`input_layer`, `hidden_layer1`, ...,
each with `.weight [out,in]` and `.bias`), e.g. the reference's own classes or plain namespaces of tensors.
No CPU fallback: the forward pass needs the built HIP library and a GPU."""
from __future__ import annotations

import ctypes as C

OUT_IDENTITY, OUT_SOFTMAX, OUT_TANH_SIGMOID = 0, 1, 2
SAMPLE_NONE, SAMPLE_CATEGORICAL, SAMPLE_GAUSSIAN = 0, 1, 2


def _wt(layer):
    """torch Linear stores [out, in]; the kernel reads [in, out]."""
    return layer.weight.detach().t().contiguous().float(), layer.bias.detach().contiguous().float()


def pack_bf16_fragments(w, ksteps, nchunks, k_order="linear", dtype=None):
    """[N, K, F] float weights -> matrix-core fragments [N, nchunks, ksteps, 64, 8] in bf16, the layout
    `dronesim_mlp_forward_bf16` reads with one 16-byte load per lane:
        frag[a, c, s, l, j] = w[a, kmap(s, l >> 5, j), 32 c + (l & 31)]      (zero beyond K / F)
    k_order "linear" (layers 1, 2): kmap = 16 s + 8 h + j.
    k_order "accumulator" (layer 3): kmap = 16 s + 8 (j >> 2) + 4 h + (j & 3) -- the order in which a lane of
    the previous layer's accumulator tile holds its features, so that tile feeds layer 3 from registers."""
    import torch
    n, k, f = w.shape
    pad = torch.zeros(n, ksteps * 16, nchunks * 32, dtype=torch.float32, device=w.device)
    pad[:, :k, :f] = w
    if k_order == "linear":
        frag = pad.view(n, ksteps, 2, 8, nchunks, 32).permute(0, 4, 1, 2, 5, 3)           # [N, c, s, h, i, j]
    elif k_order == "accumulator":
        # k = 16 s + 8 jh + 4 h + jl  -> axes (s, jh, h, jl); wanted [N, c, s, h, i, (jh, jl)]
        frag = pad.view(n, ksteps, 2, 2, 4, nchunks, 32).permute(0, 5, 1, 3, 6, 2, 4)
    else:
        raise ValueError("k_order must be 'linear' or 'accumulator'")
    return frag.reshape(n, nchunks, ksteps, 64, 8).to(dtype or torch.bfloat16).contiguous()


def split3_bf16(w):
    """float32 tensor -> its three bf16 parts by truncation, ``w == hi + mid + lo`` exactly (each part is a float32
    tensor whose low 16 bits are zero, i.e. exactly representable in bf16)."""
    import torch
    w = w.float().contiguous()
    hi = (w.view(torch.int32) & -65536).view(torch.float32)
    rem = w - hi
    mid = (rem.view(torch.int32) & -65536).view(torch.float32)
    lo = rem - mid
    return hi, mid, lo


def split2_f16(w):
    """float32 tensor -> (hi, lo) with hi = float16(w), lo = float16(w - hi): w == hi + lo to 2^-22 relative
    (float32 tensors holding float16-representable values; |w| must stay below 65504)."""
    import torch
    w = w.float().contiguous()
    hi = w.to(torch.float16).float()
    return hi, (w - hi).to(torch.float16).float()


def pack_split_fragments(w, ksteps, nchunks, k_order, scheme="bf16x3"):
    """[N, K, F] float32 weights -> ``[N, nchunks, ksteps, P, 64, 8]``: the P parts' fragments of the split
    schemes side by side per (agent, chunk, k-step): bf16 hi / mid / lo ("bf16x3") or float16 hi / lo ("f16x2")."""
    import torch
    if scheme == "bf16x3":
        parts = [pack_bf16_fragments(p, ksteps, nchunks, k_order) for p in split3_bf16(w)]
    elif scheme == "f16x2":
        parts = [pack_bf16_fragments(p, ksteps, nchunks, k_order, dtype=torch.float16) for p in split2_f16(w)]
    else:
        raise ValueError("scheme must be 'bf16x3' or 'f16x2'")
    return torch.stack(parts, dim=3).contiguous()


def pack_bf16x3_fragments(w, ksteps, nchunks, k_order):
    return pack_split_fragments(w, ksteps, nchunks, k_order, "bf16x3")


def stack_reference_modules(modules, kind=None):
    """Per-agent modules -> ``(w1, b1, w2, b2, w3, b3, out_kind, sample_kind)`` with a leading agent axis
    (CPU tensors; pure torch, no GPU needed).  ``kind``: 'discrete_softmax' | 'normal_actor' | 'critic',
    default: recognised from the first module's attribute names."""
    import torch
    from .compat import network_kind
    modules = list(modules)
    kind = kind or network_kind(modules[0])
    st = torch.stack
    if kind in ("discrete_softmax", "critic"):
        last = "out_1" if kind == "discrete_softmax" else "output_layer"
        parts = [(_wt(m.input_layer), _wt(m.hidden_layer1), _wt(getattr(m, last))) for m in modules]
        tensors = [st([p[layer][j] for p in parts]) for layer in range(3) for j in range(2)]
        return (*tensors, OUT_SOFTMAX, SAMPLE_CATEGORICAL) if kind == "discrete_softmax" else (*tensors, OUT_IDENTITY, SAMPLE_NONE)
    if kind != "normal_actor":
        raise ValueError(f"unknown network kind {kind!r}")
    w1, b1, w2, b2, w3, b3 = [], [], [], [], [], []
    for m in modules:
        a, ab = _wt(m.input_layer)
        h1w, h1b = _wt(m.hidden_layer1); h2w, h2b = _wt(m.hidden_layer2)
        o1w, o1b = _wt(m.out_1); o2w, o2b = _wt(m.out_2)
        w1.append(a); b1.append(ab)
        w2.append(torch.cat([h1w, h2w], dim=1)); b2.append(torch.cat([h1b, h2b]))
        d = o1w.shape[1]
        blk = torch.zeros(h1w.shape[1] + h2w.shape[1], 2 * d)
        blk[:h1w.shape[1], :d] = o1w; blk[h1w.shape[1]:, d:] = o2w
        w3.append(blk); b3.append(torch.cat([o1b, o2b]))
    return st(w1), st(b1), st(w2), st(b2), st(w3), st(b3), OUT_TANH_SIGMOID, SAMPLE_GAUSSIAN


def pack_split_streams(w1, w2, w3, stages, scheme="bf16x3"):
    """The weight image of `dronesim_mlp_forward_bf16x3` / `_f16x2` (include/dronesim.h): per (agent, wave) one
    stream of `stages` stages -- the P parts' fragments of one (chunk, k-step): P x 1 KiB per stage (3 KiB for bf16x3, 2 KiB for f16x2) -- in the kernel's
    consumption order.  Returns ``[N, 4, stages, P, 64, 8]`` bf16 (P = 3) or float16 (P = 2)."""
    import torch
    n, _, h1 = w1.shape
    h2 = w2.shape[2]
    nc1, nc2 = (h1 + 31) // 32, (h2 + 31) // 32
    P = 3 if scheme == "bf16x3" else 2
    f1 = pack_split_fragments(w1, 1, nc1, "linear", scheme).reshape(n, nc1, P, 64, 8)                      # [c1]
    f2 = pack_split_fragments(w2, 2 * nc1, nc2, "accumulator", scheme).reshape(n, nc2 * 2 * nc1, P, 64, 8)  # [c2 * KS2 + s]
    f3 = pack_split_fragments(w3, 2 * nc2, 1, "accumulator", scheme).reshape(n, 2 * nc2, P, 64, 8)         # [s]
    table = torch.cat([f1, f2, f3, torch.zeros_like(f1[:, :1])], dim=1)
    o2, o3, zero = nc1, nc1 + nc2 * 2 * nc1, nc1 + nc2 * 2 * nc1 + 2 * nc2
    streams = []
    for w in range(4):
        mine = list(range(w, nc2, 4))
        w2i = lambda c1, ss: [o2 + c2 * 2 * nc1 + 2 * c1 + ss for c2 in mine]
        seq = [0]
        for c1 in range(nc1):
            seq += w2i(c1, 0)
            if c1 + 1 < nc1:
                seq.append(c1 + 1)
            seq += w2i(c1, 1)
        for c2 in mine:
            seq += [o3 + 2 * c2, o3 + 2 * c2 + 1]
        assert len(seq) <= stages
        streams.append(seq + [zero] * (stages - len(seq)))
    idx = torch.tensor(streams, device=table.device)                                             # [4, stages]
    return table[:, idx].contiguous()


def f16_weight_scales(w1, w2, w3, target_exp=14):
    """Power-of-two factors ``[N, 3]`` (float32) for the f16x2 weight image (include/dronesim.h: DroneMlpBf16.wscale):
    layer l of agent i is multiplied by ``2^e`` with ``max |w| * 2^e`` in ``[2^(target_exp-1), 2^target_exp)`` before it
    is split into float16 hi + lo, so that the low parts of all but the very smallest weights are NORMAL float16 numbers
    (22 significant bits in all); unscaled, a weight of 0.05 has a subnormal low part and keeps an absolute 2^-25 only.
    An all-zero layer gets 1."""
    import torch
    out = []
    for w in (w1, w2, w3):
        m = w.float().abs().flatten(1).amax(1)                              # [N]
        e = torch.where(m > 0, target_exp - 1 - torch.floor(torch.log2(m.clamp_min(1e-37))), torch.zeros_like(m))
        out.append(torch.exp2(e.clamp(-100, 100)))
    return torch.stack(out, 1).contiguous()


def pack_bf16x3_streams(w1, w2, w3, stages):
    return pack_split_streams(w1, w2, w3, stages, "bf16x3")


def pack_f32_fragments(w):
    """[N, K, F] float32 weights -> float32 matrix-core fragments [N, ceil(F/32), ceil(K/16), 2, 64, 4], the layout
    `dronesim_mlp_forward` reads with two coalesced 16-byte loads per lane and 16-k stage (`DroneMlp.w2_layout = 1`):
        frag[a, c, s, q, l, j] = w[a, 16 s + 8 (l >> 5) + 4 q + j, 32 c + (l & 31)]       (zero beyond K / F)"""
    import torch
    n, k, f = w.shape
    ns, nc = (k + 15) // 16, (f + 31) // 32
    pad = torch.zeros(n, ns * 16, nc * 32, dtype=torch.float32, device=w.device)
    pad[:, :k, :f] = w
    # k = 16 s + 8 h + 4 q + j -> axes (s, h, q, j); column = 32 c + i; lane = 32 h + i
    frag = pad.view(n, ns, 2, 2, 4, nc, 32).permute(0, 5, 1, 3, 2, 6, 4)                  # [N, c, s, q, h, i, j]
    return frag.reshape(n, nc, ns, 2, 64, 4).contiguous()


RT_CHUNKS, RT_PAD = 7, 4          # csrc/policy.hip: kRtChunks (output chunks a wave keeps per pass), kRtPad (zero blocks behind a stream)


def pack_f32_rowtile_stream(w1, b1, w2, w3):
    """[N, d_in, h1], [N, h1], [N, h1, h2], [N, h2, nout] float32 -> the weight stream of the row-tile exact-f32 kernel
    (`DroneMlp.w2_layout = 2`, csrc/policy.hip: mlp3_rt_kernel): ``[N, blocks, 4, 64, 4]`` float32, blocks of four 1-KiB pieces
    ``[64 lanes][4 floats]`` (lane = 32 half + i) in the kernel's consumption order:

        per pass p (output chunks S_p):  for c1: L1(c1), L2(c1, c2) for c2 in S_p;  then (nout > 4 only) L3(c2) for c2 in S_p;  RT_PAD zero blocks
        L1(c1):      piece 0 = W1[2 r + half, 32 c1 + i] for r = 0..3;  piece 1 = the same for r = 4..6, then b1[32 c1 + i] (lanes < 32)
        L2(c1, c2):  piece q = W2[32 c1 + 8 q + 4 half + j, 32 c2 + i],  j = 0..3
        L3(c2):      piece q = W3[32 c2 + 8 q + 4 half + j, i]
    (zero beyond d_in / h1 / h2 / nout).  The k order 8 q + 4 half + j is the order in which a lane of the float32 matrix
    instruction's accumulator tile holds its features, so each layer's output feeds the next one from registers."""
    import torch
    n, d_in, h1 = w1.shape
    h2, nout = w3.shape[1], w3.shape[2]
    if d_in > 14 or nout > 32:
        raise ValueError("the row-tile stream needs d_in <= 14 and nout <= 32")
    nc1, nc2 = (h1 + 31) // 32, (h2 + 31) // 32
    dev = w1.device
    W1 = torch.zeros(n, 16, nc1 * 32, device=dev); W1[:, :d_in, :h1] = w1
    B1 = torch.zeros(n, nc1 * 32, device=dev); B1[:, :h1] = b1
    W2 = torch.zeros(n, nc1 * 32, nc2 * 32, device=dev); W2[:, :h1, :h2] = w2
    W3 = torch.zeros(n, nc2 * 32, 32, device=dev); W3[:, :h2, :nout] = w3
    # L1 blocks [n, nc1, 4 pieces, 64 lanes, 4]: k = 2 r + half with r = 4 piece + j
    l1 = torch.zeros(n, nc1, 4, 2, 32, 4, device=dev)                            # [.., piece, half, i, j]
    kk = W1.view(n, 8, 2, nc1, 32)                                               # [n, r, half, c1, i]
    l1[:, :, 0] = kk[:, 0:4].permute(0, 3, 2, 4, 1)                              # r = 0..3 -> j
    l1[:, :, 1, :, :, 0:3] = kk[:, 4:7].permute(0, 3, 2, 4, 1)                   # r = 4..6
    l1[:, :, 1, 0, :, 3] = B1.view(n, nc1, 32)                                   # bias in lanes 0..31 (half 0)
    l1 = l1.reshape(n, nc1, 4, 64, 4)
    # L2 blocks [n, c1, c2, q, half, i, j]: k = 32 c1 + 8 q + 4 half + j
    l2 = W2.view(n, nc1, 4, 2, 4, nc2, 32).permute(0, 1, 5, 2, 3, 6, 4).reshape(n, nc1, nc2, 4, 64, 4)
    l3 = W3.view(n, nc2, 4, 2, 4, 32).permute(0, 1, 2, 3, 5, 4).reshape(n, nc2, 4, 64, 4)
    passes = (nc2 + RT_CHUNKS - 1) // RT_CHUNKS
    per = (nc2 + passes - 1) // passes
    seq = []
    for p in range(passes):
        chunks = range(p * per, min(nc2, (p + 1) * per))
        for c1 in range(nc1):
            seq.append(l1[:, c1])
            seq += [l2[:, c1, c2] for c2 in chunks]
        if nout > 4:                                   # (nout <= 4: layer 3 runs on the vector ALU from the plain w3 array)
            seq += [l3[:, c2] for c2 in chunks]
    seq += [torch.zeros_like(l1[:, 0])] * RT_PAD
    return torch.stack(seq, dim=1).contiguous()


def pack_f16_rowtile_stream(w1, w2, blocks, w3=None):
    """[N, d_in, h1], [N, h1, h2] (and, for nout > 4, [N, h2, nout]) float32 weights ALREADY multiplied by their power-of-two factors
    -> the weight stream of the float16 row-tile kernel (`dronesim_mlp_forward_f16x2_rt`, csrc/policy.hip: mlp3_rt16_kernel):
    ``[N, blocks, 4, 64, 8]`` float16, blocks of four 1-KiB pieces in the kernel's consumption order -- per pass (output chunks S_p),
    for every in-chunk c1: L1(c1) = (W1 hi, W1 lo, 0, 0), then L2(c1, c2) = (hi, lo of k-step 2 c1; hi, lo of k-step 2 c1 + 1) for c2 in
    S_p; with `w3` the pass ends with L3(c2) = (hi, lo of k-step 2 c2; hi, lo of k-step 2 c2 + 1) of W3 for c2 in S_p -- zero blocks up
    to `blocks`."""
    import torch
    n, d_in, h1 = w1.shape
    h2 = w2.shape[2]
    nc1, nc2 = (h1 + 31) // 32, (h2 + 31) // 32
    f1 = pack_split_fragments(w1, 1, nc1, "linear", "f16x2")                    # [N, c1, 1, P, 64, 8]
    f2 = pack_split_fragments(w2, 2 * nc1, nc2, "accumulator", "f16x2")         # [N, c2, 2 c1 + s, P, 64, 8]
    f3 = None if w3 is None else pack_split_fragments(w3, 2 * nc2, 1, "accumulator", "f16x2")   # [N, 1, 2 c2 + s, P, 64, 8]
    zero = torch.zeros_like(f1[:, 0, 0, 0])
    passes = (nc2 + RT_CHUNKS - 1) // RT_CHUNKS
    per = (nc2 + passes - 1) // passes
    seq = []
    for p in range(passes):
        chunks = range(p * per, min(nc2, (p + 1) * per))
        for c1 in range(nc1):
            seq.append(torch.stack([f1[:, c1, 0, 0], f1[:, c1, 0, 1], zero, zero], dim=1))
            for c2 in chunks:
                seq.append(torch.stack([f2[:, c2, 2 * c1, 0], f2[:, c2, 2 * c1, 1], f2[:, c2, 2 * c1 + 1, 0], f2[:, c2, 2 * c1 + 1, 1]], dim=1))
        for c2 in (chunks if f3 is not None else ()):
            seq.append(torch.stack([f3[:, 0, 2 * c2, 0], f3[:, 0, 2 * c2, 1], f3[:, 0, 2 * c2 + 1, 0], f3[:, 0, 2 * c2 + 1, 1]], dim=1))
    assert len(seq) <= blocks - 12
    pad = torch.zeros_like(seq[0])
    seq += [pad] * (blocks - len(seq))
    return torch.stack(seq, dim=1).contiguous()


class BatchedMLP:
    def __init__(self, w1, b1, w2, b2, w3, b3, out_kind, sample_kind, device=None, seed=0, precision="f32", pack_w2=True,
                 split_kernel=False):
        """w1 [N,d_in,h1], b1 [N,h1], w2 [N,h1,h2], b2 [N,h2], w3 [N,h2,nout], b3 [N,nout] (float32).
        ``precision="f32"`` (default) is exact float32 on the matrix cores; ``"bf16x3"`` and ``"f16x2"`` give
        float32-accurate results (same 1e-5 bar) from three-part bfloat16 / two-part float16 splits of weights and
        activations on the 16-bit matrix instructions (six / three partial products); f16x2 is the faster one and
        needs every weight, input and hidden activation below 65504 in magnitude; ``"bf16"`` runs weights and
        activations in plain bfloat16 with float32 accumulation (~1e-2 relative agreement, fastest).
        ``pack_w2`` (f32 only): True (default) hands the kernel ONE packed stream of all three layers (`pack_f32_rowtile_stream`,
        `DroneMlp.w2_layout = 2`: the row-tile kernel of round 6; d_in <= 14), "fragments" layer 2 as matrix-core fragments
        (`pack_f32_fragments`, `w2_layout = 1`: the kernel of rounds 3-5); False keeps the [N, h1, h2] array of the plain C ABI
        (`w2_layout = 0`).
        ``split_kernel`` (f16x2 only): False (default) runs the row-tile kernel of round 6 (`dronesim_mlp_forward_f16x2_rt`: ONE
        stream per agent, `pack_f16_rowtile_stream`), True the split kernel of rounds 2-5 (`dronesim_mlp_forward_f16x2`).
        NOTE: the packed images are SNAPSHOTS of the weights: after an in-place update of ``w1 .. b3`` call
        `refresh_weights()` (re-packs into the same device buffers)."""
        import torch
        from . import _native
        self._torch, self._native = torch, _native
        self._lib = _native.lib()
        if not torch.cuda.is_available():
            raise RuntimeError("BatchedMLP needs a ROCm GPU (MI355X); there is no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:       # "cuda" -> "cuda:<current>": tensors carry an index
            self.device = torch.device("cuda", torch.cuda.current_device())
        f = lambda t: torch.as_tensor(t, dtype=torch.float32).to(self.device).contiguous()
        self.w1, self.b1, self.w2, self.b2, self.w3, self.b3 = (f(t) for t in (w1, b1, w2, b2, w3, b3))
        self.n_agents, self.d_in, self.h1 = self.w1.shape
        self.h2, self.nout = self.w3.shape[1], self.w3.shape[2]
        assert self.w2.shape == (self.n_agents, self.h1, self.h2) and self.b3.shape == (self.n_agents, self.nout)
        self.out_kind, self.sample_kind = int(out_kind), int(sample_kind)
        self.seed, self.counter = int(seed), 0
        m = _native.DroneMlp()
        m.N, m.d_in, m.h1, m.h2, m.nout = self.n_agents, self.d_in, self.h1, self.h2, self.nout
        m.out_kind, m.sample_kind = self.out_kind, self.sample_kind
        m.w1, m.b1, m.w2 = self.w1.data_ptr(), self.b1.data_ptr(), self.w2.data_ptr()
        m.b2, m.w3, m.b3 = self.b2.data_ptr(), self.w3.data_ptr(), self.b3.data_ptr()
        self._m = m
        self.precision = precision
        self._rowtile = False
        if precision == "f32" and pack_w2 == "fragments":
            # round 3-5 fast path: layer 2 reads its weights as matrix-core fragments (`DroneMlp.w2_layout = 1`)
            self._w2p = pack_f32_fragments(self.w2)
            m.w2, m.w2_layout = self._w2p.data_ptr(), 1
        elif precision == "f32" and pack_w2 and self.d_in <= 14:
            # round 6: ONE stream holding all three layers in the row-tile kernel's consumption order (`DroneMlp.w2_layout = 2`)
            self._w2p = pack_f32_rowtile_stream(self.w1, self.b1, self.w2, self.w3)
            assert self._w2p.shape[1] == int(self._lib.dronesim_mlp_rt_blocks(self.h1, self.h2, self.nout))
            m.w2, m.w2_layout = self._w2p.data_ptr(), 2
            self._rowtile = True
        elif precision == "f32" and pack_w2:
            self._w2p = pack_f32_fragments(self.w2)
            m.w2, m.w2_layout = self._w2p.data_ptr(), 1
        if precision == "bf16":
            if self.d_in > 16:
                raise ValueError("the bf16 path supports d_in <= 16")
            nc1, nc2 = (self.h1 + 31) // 32, (self.h2 + 31) // 32
            self._w1p = pack_bf16_fragments(self.w1, 1, nc1)
            self._w2p = pack_bf16_fragments(self.w2, 2 * nc1, nc2)
            self._w3p = pack_bf16_fragments(self.w3, 2 * nc2, 1, k_order="accumulator")
            mb = _native.DroneMlpBf16()
            mb.N, mb.d_in, mb.h1, mb.h2, mb.nout = self.n_agents, self.d_in, self.h1, self.h2, self.nout
            mb.out_kind, mb.sample_kind = self.out_kind, self.sample_kind
            mb.w1p, mb.w2p, mb.w3p = self._w1p.data_ptr(), self._w2p.data_ptr(), self._w3p.data_ptr()
            mb.b1, mb.b2, mb.b3 = self.b1.data_ptr(), self.b2.data_ptr(), self.b3.data_ptr()
            self._m = mb
        elif precision in ("bf16x3", "f16x2"):
            if self.d_in > 16:
                raise ValueError(f"the {precision} path supports d_in <= 16")
            stages = int(self._lib.dronesim_mlp_bf16x3_stages(self.h1, self.h2))
            self._wscale = None
            if precision == "f16x2":                               # power-of-two factors: the low parts stay normal float16
                self._wscale = f16_weight_scales(self.w1, self.w2, self.w3)
            # round 6: f16x2 takes the ROW-TILE kernel (one stream per agent, a ring per workgroup; layer 3 in exact float32 on the
            # vector ALU for nout <= 4, on the matrix cores otherwise); `split_kernel=True` keeps the split kernel of rounds 2-5
            self._rt16 = precision == "f16x2" and not split_kernel
            if self._rt16:
                stages = int(self._lib.dronesim_mlp_rt16_blocks(self.h1, self.h2, self.nout))
            self._w1p = self._split_image(stages)
            mb = _native.DroneMlpBf16()
            mb.N, mb.d_in, mb.h1, mb.h2, mb.nout = self.n_agents, self.d_in, self.h1, self.h2, self.nout
            mb.out_kind, mb.sample_kind, mb.reserved = self.out_kind, self.sample_kind, stages
            mb.w1p, mb.w2p, mb.w3p = self._w1p.data_ptr(), None, (self.w3.data_ptr() if self._rt16 and self.nout <= 4 else None)
            mb.b1, mb.b2, mb.b3 = self.b1.data_ptr(), self.b2.data_ptr(), self.b3.data_ptr()
            mb.wscale = None if self._wscale is None else self._wscale.data_ptr()
            self._m = mb
        elif precision != "f32":
            raise ValueError("precision must be 'f32', 'f16x2', 'bf16x3' or 'bf16'")

    def refresh_weights(self, w1=None, b1=None, w2=None, b2=None, w3=None, b3=None):
        """Call after every weight update.  The kernels read PACKED images of the weights (`_w2p` for f32 layer 2;
        `_w1p/_w2p/_w3p` for the 16-bit paths) that are snapshots taken when the object was built: an in-place update
        of ``self.w2`` alone (optimizer step) would otherwise leave the kernel on a mix of new and stale weights.
        Optional arguments are copied into the live tensors first; every packed image is then re-made IN PLACE (same
        device addresses), so a captured hipGraph that holds them stays valid."""
        for name, val in (("w1", w1), ("b1", b1), ("w2", w2), ("b2", b2), ("w3", w3), ("b3", b3)):
            if val is not None:
                getattr(self, name).copy_(self._torch.as_tensor(val, dtype=self._torch.float32))
        if self.precision == "f32":
            if getattr(self, "_w2p", None) is not None:
                self._w2p.copy_(pack_f32_rowtile_stream(self.w1, self.b1, self.w2, self.w3) if self._rowtile
                                else pack_f32_fragments(self.w2))
        elif self.precision == "bf16":
            nc1, nc2 = (self.h1 + 31) // 32, (self.h2 + 31) // 32
            self._w1p.copy_(pack_bf16_fragments(self.w1, 1, nc1))
            self._w2p.copy_(pack_bf16_fragments(self.w2, 2 * nc1, nc2))
            self._w3p.copy_(pack_bf16_fragments(self.w3, 2 * nc2, 1, k_order="accumulator"))
        else:
            stages = int(self._lib.dronesim_mlp_rt16_blocks(self.h1, self.h2, self.nout) if getattr(self, "_rt16", False)
                         else self._lib.dronesim_mlp_bf16x3_stages(self.h1, self.h2))
            if self._wscale is not None:
                self._wscale.copy_(f16_weight_scales(self.w1, self.w2, self.w3))
            self._w1p.copy_(self._split_image(stages))

    load_weights = refresh_weights

    def _split_image(self, stages):
        """The packed weight streams of the split precisions; f16x2: of the weights times their power-of-two factors."""
        if getattr(self, "_rt16", False):
            sc = self._wscale
            return pack_f16_rowtile_stream(self.w1 * sc[:, 0, None, None], self.w2 * sc[:, 1, None, None], stages,
                                           self.w3 * sc[:, 2, None, None] if self.nout > 4 else None)
        if self._wscale is None:
            return pack_split_streams(self.w1, self.w2, self.w3, stages, self.precision)
        sc = self._wscale
        return pack_split_streams(self.w1 * sc[:, 0, None, None], self.w2 * sc[:, 1, None, None], self.w3 * sc[:, 2, None, None],
                                  stages, self.precision)

    # ------------------------------------------------------------------ constructors
    @classmethod
    def from_discrete_softmax(cls, modules, **kw):
        """N x DiscreteSoftmaxNN: input_layer -> hidden_layer1 -> out_1, softmax (utils.py:271-302)."""
        return cls(*stack_reference_modules(modules, "discrete_softmax"), **kw)

    @classmethod
    def from_critic(cls, modules, **kw):
        """N x CriticNN: input_layer -> hidden_layer1 -> output_layer, no activation (utils.py:22-53)."""
        return cls(*stack_reference_modules(modules, "critic"), **kw)

    @classmethod
    def from_normal_actor(cls, modules, **kw):
        """N x NormalActorNN: the two heads (hidden_layer1 -> out_1 tanh, hidden_layer2 -> out_2 sigmoid,
        utils.py:64-108) become one concatenated second layer and a block-diagonal output layer."""
        return cls(*stack_reference_modules(modules, "normal_actor"), **kw)

    @classmethod
    def from_reference_file(cls, path, **kw):
        """All agents' networks of one of the reference's saved files (``torch.save`` of a list of modules,
        SAC_agents.py:404-406, e.g. ``models/discrete-A2Cactors.pth``), opened without the reference's code
        (`compat.load_reference_modules`); the network class is recognised by its attribute names."""
        from .compat import load_reference_modules
        return cls(*stack_reference_modules(load_reference_modules(path)), **kw)

    # ------------------------------------------------------------------ evaluation
    def _given(self, t, shape, dtype, what):
        """A caller-provided output tensor (e.g. a slot of a `RolloutStorage`): the kernel writes straight into it."""
        torch = self._torch
        if not (torch.is_tensor(t) and t.device.type == self.device.type and t.device.index == self.device.index
                and t.dtype == dtype and t.is_contiguous()
                and t.numel() == int(torch.Size(shape).numel())):
            raise ValueError(f"{what} must be a contiguous {dtype} tensor of {tuple(shape)} elements on {self.device}")
        return t

    def _run(self, z, want_out, sample, env_base=0, env=None, out=None, act_out=None, idx_out=None):
        torch = self._torch
        z = z.to(device=self.device, dtype=torch.float32).contiguous()
        E = z.shape[0]
        if tuple(z.shape[:2]) != (E, self.n_agents) or z[0, 0].numel() != self.d_in:
            raise ValueError(f"z must be [E,{self.n_agents},{self.d_in}], got {tuple(z.shape)}")
        if out is not None:
            out = self._given(out, (E, self.n_agents, self.nout), torch.float32, "out")
        elif want_out:
            out = torch.empty(E, self.n_agents, self.nout, device=self.device)
        act = idx = None
        m = self._m
        m.sample_kind = self.sample_kind if sample else SAMPLE_NONE
        if sample:
            act = (torch.empty(E, self.n_agents, 2, device=self.device) if act_out is None
                   else self._given(act_out, (E, self.n_agents, 2), torch.float32, "act_out"))
            if self.sample_kind == SAMPLE_CATEGORICAL:
                idx = (torch.empty(E, self.n_agents, dtype=torch.int32, device=self.device) if idx_out is None
                       else self._given(idx_out, (E, self.n_agents), torch.int32, "idx_out"))
        entry = {"bf16": self._lib.dronesim_mlp_forward_bf16, "bf16x3": self._lib.dronesim_mlp_forward_bf16x3,
                 "f16x2": (self._lib.dronesim_mlp_forward_f16x2_rt if getattr(self, "_rt16", False) else self._lib.dronesim_mlp_forward_f16x2),
                 "f32": self._lib.dronesim_mlp_forward}[self.precision]
        with torch.cuda.device(self.device):
            rc = entry(C.byref(m), z.data_ptr(), None if out is None else out.data_ptr(),
                                                None if act is None else act.data_ptr(),
                                                None if idx is None else idx.data_ptr(), self.seed, self.counter,
                                                int(env_base), None if env is None else env.t.data_ptr(),
                                                None if env is None else env.episode.data_ptr(), E,
                                                C.c_void_p(torch.cuda.current_stream().cuda_stream))
        self._native.check(rc, "dronesim_mlp_forward")
        if sample and env is None:
            self.counter += 1
        return out, act, idx

    def forward(self, z, out=None):
        """Post-activation outputs ``[E,N,nout]``: action probabilities / (mu_x, mu_y, var_x, var_y) / value.
        ``out``: write them into this tensor (e.g. ``storage.values[t]`` for a critic: ``[E,N]`` = ``[E,N,1]``)."""
        return self._run(z, True, False, out=out)[0]

    def sample_action(self, z, env_base=0, return_outputs=False, env=None, act_out=None, idx_out=None):
        """Batched ``sample_action`` (utils.py:304-309 / 110-117): actions ``[E,N,2]`` ready for ``env.step``;
        for the categorical policy also the chosen indices.  Every call advances the Philox counter; with
        ``env=`` (a batched `drones`) the stream is keyed by the env's own device-side ``t`` / ``episode``
        counters instead, which keeps a captured hipGraph drawing fresh numbers on every replay.
        ``act_out`` / ``idx_out``: tensors the kernel writes the actions / indices into (``storage.actions[t]``)."""
        if env is not None:
            env_base = env.env_lo
        out, act, idx = self._run(z, return_outputs, True, env_base, env, act_out=act_out, idx_out=idx_out)
        return (act, idx, out) if return_outputs else (act, idx)
