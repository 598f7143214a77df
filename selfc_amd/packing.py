"""Repack reference-layout conv weights into the MFMA A-fragment layout the
kernels of selfc_amd/csrc/dense_conv.hip read.

Reference layouts (state_dict contract):
  conv1..4   (32, cin + 32*(i-1), 3, 3)  [DenseBlock]  or (.., 1, 3, 3) [D2DTInput]
  conv5      (cout, cin + 128, 3, 3)     [DenseBlock]  or (cout, cin+128, 3, 1, 1) [D2DTInput]
Input-channel order of every conv is the reference's ``torch.cat((x, x1, x2, x3, x4), 1)``
(Subnet_constructor.py:28-31): first the cin inputs, then the 32-channel features.

Kernel K order of a 3x3 conv ("stages", see build_stages() in dense_conv.hip):
  cin <= 3 : [im2col stage: k = tap*cin + c, zero padded to 32] + features
  cin  > 3 : [input channels in 32-wide pieces (last one 16 wide if roundup(cin,16)%32==16)] + features
  every non-im2col stage is tap-major: for tap in 0..8: for k in 0..width-1.
A 32x32x16 A fragment f holds W[outch = lane&31][k = 16 f + 8 (lane>>5) + j], j = 0..7.

Temporal conv5 (tconv5_kernel): fragments [tap][net][kstep][otile][lane][8] of the
16x16x32 MFMA, W[outch = 16 otile + (lane&15)][k = 32 kstep + 8 (lane>>4) + j]; k-step 0
is the 3 input channels when cin <= 3, then the dense buffer's channels in order.
"""
from __future__ import annotations

from typing import List, Sequence

import torch

from ._lib import operand_dtype

F16 = operand_dtype()     # MFMA operand dtype of the loaded library (float16, or bfloat16 under SELFC_OPERAND=bf16)

# Every pack_* function is a pure re-ordering (gather + zero fill) of its inputs.  PackPlan (below) exploits that: it
# runs the functions ONCE on index-valued stand-ins with _RAW set (no 16-bit rounding) to learn the gather map, after
# which re-packing changed weights - every optimizer step in training - is one concatenation, one gather and one cast.
_RAW = False


def _operand(t: torch.Tensor) -> torch.Tensor:
    return t.contiguous() if _RAW else t.to(F16).contiguous()


def roundup(v: int, m: int) -> int:
    return (v + m - 1) // m * m


def dense_channels(cin: int) -> int:
    """Channel stride of a subnet's f16 dense buffer (dense_conv.hip: dense_channels)."""
    return 128 if cin <= 3 else roundup(cin, 32) + 128


def _k_expand_3x3(w: torch.Tensor, cin: int, nfeat: int) -> torch.Tensor:
    """w (O, cin + 32*nfeat, 3, 3) fp32 -> (O, K) in kernel K order."""
    o = w.shape[0]
    w9 = w.reshape(o, w.shape[1], 9)                       # tap = ky*3 + kx
    cols: List[torch.Tensor] = []
    if cin <= 3:
        st = torch.zeros(o, 32, dtype=w.dtype, device=w.device)
        st[:, : 9 * cin] = w9[:, :cin, :].permute(0, 2, 1).reshape(o, 9 * cin)   # k = tap*cin + c
        cols.append(st)
    else:
        cin16 = roundup(cin, 16)
        for c0 in range(0, cin16, 32):
            width = 32 if cin16 - c0 >= 32 else 16
            st = torch.zeros(o, 9, width, dtype=w.dtype, device=w.device)
            real = max(0, min(width, cin - c0))
            st[:, :, :real] = w9[:, c0:c0 + real, :].permute(0, 2, 1)
            cols.append(st.reshape(o, 9 * width))
    for i in range(nfeat):
        st = w9[:, cin + 32 * i: cin + 32 * (i + 1), :].permute(0, 2, 1)            # (O, 9, 32)
        cols.append(st.reshape(o, 9 * 32))
    return torch.cat(cols, dim=1)


def pack_conv3x3(weight: torch.Tensor, cin: int, layer: int) -> torch.Tensor:
    """conv `layer` (1..5) of a dense block -> f16 [nfrag, 64, 8] (32 output rows,
    zero padded when the conv has fewer, i.e. the 2-D conv5)."""
    w = weight.detach().float()
    if w.dim() == 5:                                       # Conv3d (1,3,3)
        assert w.shape[2] == 1
        w = w[:, :, 0]
    assert w.shape[2:] == (3, 3) and w.shape[1] == cin + 32 * (layer - 1), (tuple(w.shape), cin, layer)
    assert w.shape[0] <= 32
    if w.shape[0] < 32:
        w = torch.cat((w, w.new_zeros(32 - w.shape[0], *w.shape[1:])), 0)
    wk = _k_expand_3x3(w, cin, layer - 1)                  # (32, K)
    nfrag = wk.shape[1] // 16
    frag = wk.reshape(32, nfrag, 2, 8).permute(1, 2, 0, 3).reshape(nfrag, 64, 8)
    return _operand(frag)


def pack_tconv5(weights: Sequence[torch.Tensor], cin: int) -> torch.Tensor:
    """1 or 2 (G, H) temporal conv5 weights (cout, cin+128, 3, 1, 1) -> f16
    [3, nets, KS, OT, 64, 8]."""
    nets = len(weights)
    cout = weights[0].shape[0]
    ot = roundup(cout, 16) // 16
    hasx = cin <= 3
    kd = dense_channels(cin) // 32
    ks = kd + (1 if hasx else 0)
    dev = weights[0].device
    wk = torch.zeros(nets, 3, ot * 16, ks * 32, dtype=torch.float32, device=dev)
    for n, w in enumerate(weights):
        w = w.detach().float()
        assert w.shape[1:] == (cin + 128, 3, 1, 1) and w.shape[0] == cout, tuple(w.shape)
        w3 = w[:, :, :, 0, 0].permute(2, 0, 1)             # (tap, cout, C)
        if hasx:
            wk[n, :, :cout, :cin] = w3[:, :, :cin]
            wk[n, :, :cout, 32:32 + 128] = w3[:, :, cin:]
        else:
            cin32 = roundup(cin, 32)
            wk[n, :, :cout, :cin] = w3[:, :, :cin]
            wk[n, :, :cout, cin32:cin32 + 128] = w3[:, :, cin:]
    frag = wk.reshape(nets, 3, ot, 16, ks, 4, 8).permute(1, 0, 4, 2, 5, 3, 6).reshape(3, nets, ks, ot, 64, 8)
    return _operand(frag)


def widen_dense_params(weights: Sequence[torch.Tensor], biases: Sequence[torch.Tensor], cin: int, cout: int, gc: int,
                       cin_v: int = None, cout_v: int = None):
    """An EXACTLY equivalent dense block in the kernels' layout (growth 32, cin_v inputs, cout_v outputs) of a block with
    growth gc <= 32 (codec variant: stp_denseblock_innerc = 12, SelfC_Codec_arch_inv.py:248-252), cin <= cin_v and
    cout <= cout_v: real rows / columns keep their place in the reference's concat order [x | f1 | f2 | f3 | f4], every
    added row, column and bias is zero - the padded features are LeakyReLU(0) = 0 and meet zero weights, so no rounding
    differs.  Pure placement + zero fill (PackPlan-compatible).  Returns (weights, biases) of conv1..conv5."""
    cin_v = cin if cin_v is None else cin_v
    cout_v = cout if cout_v is None else cout_v
    assert gc <= 32 and cin <= cin_v and cout <= cout_v
    ws, bs = [], []
    for k in range(1, 6):
        w, b = weights[k - 1], biases[k - 1]
        nfeat = k - 1
        o_real = gc if k < 5 else cout
        o_v = 32 if k < 5 else cout_v
        assert w.shape[0] == o_real and w.shape[1] == cin + gc * nfeat, (tuple(w.shape), cin, gc, k)
        wv = torch.zeros((o_v, cin_v + 32 * nfeat) + tuple(w.shape[2:]), dtype=w.dtype, device=w.device)
        wv[:o_real, :cin] = w[:, :cin]
        for j in range(nfeat):
            wv[:o_real, cin_v + 32 * j: cin_v + 32 * j + gc] = w[:, cin + gc * j: cin + gc * (j + 1)]
        bv = torch.zeros(o_v, dtype=b.dtype, device=b.device)
        bv[:o_real] = b
        ws.append(wv)
        bs.append(bv)
    return ws, bs


def pad_bias(bias, n: int = 64, device=None) -> torch.Tensor:
    out = torch.zeros(n, dtype=torch.float32, device=device if bias is None else bias.device)
    if bias is not None:
        out[: bias.numel()] = bias.detach().float()
    return out


def pack_pointwise(weight: torch.Tensor) -> torch.Tensor:
    """1x1(x1) conv / Linear weight (cout, cin[,1,1[,1]]) -> f16 [OT, KS, 64, 8] fragments of the
    16x16x32 MFMA: W[16 o + (lane&15)][32 ks + 8 (lane>>4) + j] (stp.hip: pwconv_kernel, gagg_mix_kernel)."""
    w = weight.detach().float().reshape(weight.shape[0], -1)
    cout, cin = w.shape
    assert cin % 32 == 0, cin
    ot, ks = roundup(cout, 16) // 16, cin // 32
    wp = torch.zeros(ot * 16, cin, dtype=torch.float32, device=w.device)
    wp[:cout] = w
    frag = wp.reshape(ot, 16, ks, 4, 8).permute(0, 2, 3, 1, 4).reshape(ot, ks, 64, 8)
    return _operand(frag)


def gagg_row_perm(device=None) -> torch.Tensor:
    """Output-row order of GlobalAgg's proj1 for gagg_mix_kernel (stp.hip): MFMA tile o, row 4 kq + i holds output channel
    32 (o >> 1) + 8 kq + 4 (o & 1) + i - the lane that computes it (k octet kq) already holds that channel of x in registers,
    so the residual needs no second read.  Returns perm with packed_row[16 o + r] = W[perm[16 o + r]]."""
    idx = torch.arange(64)
    o, r = idx // 16, idx % 16
    perm = 32 * (o // 2) + 8 * (r // 4) + 4 * (o % 2) + (r % 4)
    return perm.to(device) if device is not None else perm


def gmm_head_perm(hf_dim: int, K: int, device=None) -> torch.Tensor:
    """Output-channel permutation of the GMM head's last conv for the fused head + sampler kernel (stp.hip:
    stp_head_gmm_kernel): new channel (3 k + j) * hf_dim + c  <-  reference channel (c * K + k) * 3 + j
    (SelfC_GMM_arch_inv.py:382-386: parameters viewed as (hf_dim, K, 3))."""
    k, j, c = torch.meshgrid(torch.arange(K), torch.arange(3), torch.arange(hf_dim), indexing="ij")
    return ((c * K + k) * 3 + j).reshape(-1).to(device)


def head_row_perm(cout: int, device=None) -> torch.Tensor:
    """Output-channel permutation of a hidden layer of the GMM head for the whole-head kernel (stp.hip: stp_head_gmm_kernel):
    the 16x16x32 MFMA leaves rows 4 kq + e of an output tile in lane (pixel, kq) and wants k-channels 8 kq + 0..7 of a
    32-channel k-step there as the next layer's operand, so tile 2 s + h, row 4 kq + e computes channel 32 s + 8 kq + 4 h + e:
    two consecutive tiles ARE one operand fragment, without any cross-lane movement."""
    assert cout % 32 == 0, cout
    t, kq, e = torch.meshgrid(torch.arange(cout // 16), torch.arange(4), torch.arange(4), indexing="ij")
    return ((t // 2) * 32 + 8 * kq + 4 * (t % 2) + e).reshape(-1).to(device)


def pool_weight_map(fc_weight: torch.Tensor, h: int, w: int) -> torch.Tensor:
    """Fold ``fc(adaptive_avg_pool2d(x, (32,32)).flatten())`` (SelfC_GMM_arch_inv.py:269-271) into one
    (h*w,) map: g = sum_px x[px] * wmap[px] + fc.bias.  adaptive_avg_pool2d bin i covers
    [floor(i*L/32), ceil((i+1)*L/32)) - bins overlap when 32 does not divide L and replicate when L < 32."""
    dev = fc_weight.device
    fcw = fc_weight.detach().double().reshape(32, 32)
    return (pool_bins(h, dev).t() @ fcw @ pool_bins(w, dev)).reshape(h * w).float().contiguous()


_BINS = {}


def pool_bins(length: int, device) -> torch.Tensor:
    """(32, length) float64 averaging matrix of adaptive_avg_pool over one axis: bin i = [floor(i*L/32), ceil((i+1)*L/32))."""
    key = (length, str(device))
    if key not in _BINS:
        m = torch.zeros(32, length, dtype=torch.float64)
        for i in range(32):
            s, e = (i * length) // 32, -((-(i + 1) * length) // 32)
            m[i, s:e] = 1.0 / (e - s)
        _BINS[key] = m.to(device)
    return _BINS[key]


def pool_weight_map_grad(dwmap: torch.Tensor, h: int, w: int) -> torch.Tensor:
    """Adjoint of pool_weight_map: gradient of the (h*w,) map -> gradient of fc.weight (1, 1024)."""
    d = dwmap.double().reshape(h, w)
    return (pool_bins(h, d.device) @ d @ pool_bins(w, d.device).t()).reshape(1, 32 * 32).float()


def pool_weight_map_batch(fc_weights: Sequence[torch.Tensor], h: int, w: int) -> torch.Tensor:
    """pool_weight_map of several GlobalAgg blocks in one batched product -> (G, h*w) (an STP chain has six; module by module
    that is four small launches each, every training step)."""
    dev = fc_weights[0].device
    fcw = torch.stack([f.detach().reshape(32, 32) for f in fc_weights]).double()
    return torch.matmul(torch.matmul(pool_bins(h, dev).t(), fcw), pool_bins(w, dev)).reshape(len(fc_weights), h * w).float().contiguous()


def pool_weight_map_grad_batch(dwmaps: torch.Tensor, h: int, w: int) -> torch.Tensor:
    """pool_weight_map_grad of a (G, h*w) stack of map gradients -> (G, 1024)."""
    d = dwmaps.double().reshape(-1, h, w)
    return torch.matmul(torch.matmul(pool_bins(h, d.device), d), pool_bins(w, d.device).t()).reshape(-1, 32 * 32).float()


def pack_fused_gh(weights: Sequence[torch.Tensor], cin: int = 3) -> torch.Tensor:
    """conv1..conv4 weights of a cin == 3 dense block -> the fragment stream of csrc/fused_gh.hip:
    per conv [im2col48: K = 12 taps x 4 (c0 c1 c2 0), 3 fragments][feature j = 1..: tap-major, 18
    fragments each]  -> f16 [120, 64, 8]."""
    assert cin == 3 and len(weights) == 4
    frags = []
    for layer, wt in enumerate(weights, start=1):
        w = wt.detach().float()
        if w.dim() == 5:
            w = w[:, :, 0]
        assert w.shape == (32, cin + 32 * (layer - 1), 3, 3), tuple(w.shape)
        w9 = w.reshape(32, w.shape[1], 9)
        im = torch.zeros(32, 12, 4, dtype=torch.float32, device=w.device)
        im[:, :9, :3] = w9[:, :3, :].permute(0, 2, 1)                    # k = tap*4 + c
        cols = [im.reshape(32, 48)]
        for i in range(layer - 1):
            cols.append(w9[:, cin + 32 * i: cin + 32 * (i + 1), :].permute(0, 2, 1).reshape(32, 288))
        wk = torch.cat(cols, dim=1)
        nfrag = wk.shape[1] // 16
        frags.append(wk.reshape(32, nfrag, 2, 8).permute(1, 2, 0, 3).reshape(nfrag, 64, 8))
    out = torch.cat(frags, dim=0)
    assert out.shape[0] == 120
    return _operand(out)


def pack_fused_f(weights: Sequence[torch.Tensor], cin: int = 48) -> torch.Tensor:
    """conv1..conv4 weights of a cin == 48 dense block -> the fragment stream of csrc/fused_f.hip (two pairwise-fused
    launches).  Per pair (conv a, conv b) = (1, 2), (3, 4) with nin = cin + 64*pair shared input channels:
    [merged steps - source group (x2, f1, f2) major, tap-major inside a group, 16-channel k-step minor: fragment of
    conv a, fragment of conv b (its first nin inputs)]
    + [conv b's last 32 inputs (the feature conv a produced), tap-major: 18 fragments]  -> f16 [72 + 144, 64, 8]."""
    assert cin == 48 and len(weights) == 4

    def to_frags(wk):
        nfrag = wk.shape[1] // 16
        return wk.reshape(32, nfrag, 2, 8).permute(1, 2, 0, 3).reshape(nfrag, 64, 8)

    out = []
    for pair in (0, 1):
        nin = cin + 64 * pair
        ws = []
        for j in (0, 1):
            w = weights[2 * pair + j].detach().float()
            if w.dim() == 5:
                w = w[:, :, 0]
            assert w.shape == (32, nin + 32 * j, 3, 3), tuple(w.shape)
            ws.append(w.reshape(32, nin + 32 * j, 9))
        groups = [(0, cin)] + [(cin + 32 * i, cin + 32 * (i + 1)) for i in range(2 * pair)]     # x2, then f1, f2

        def kmajor(w9):          # source group major, tap-major inside a group, channel minor
            return torch.cat([w9[:, lo:hi].permute(0, 2, 1).reshape(32, 9 * (hi - lo)) for lo, hi in groups], dim=1)

        fa = to_frags(kmajor(ws[0]))
        fb = to_frags(kmajor(ws[1]))
        ff = to_frags(ws[1][:, nin:].permute(0, 2, 1).reshape(32, 9 * 32))
        out.append(torch.stack((fa, fb), dim=1).reshape(2 * fa.shape[0], 64, 8))
        out.append(ff)
    res = torch.cat(out, dim=0)
    assert res.shape[0] == 216
    return _operand(res)


# ---- fragment streams of csrc/fused_f16.hip (the pair kernels laid out for v_mfma_f32_16x16x32) -------------------------
def f16_steps(pair: int, cin: int = 48):
    """K schedule of the 16x16x32 pair kernels: (merged steps, FM steps), each step = four octet descriptors (tap, first input
    channel) | None, one per lane group q = lane >> 4 (8 input channels each).  x part (cin = 48): nine steps of 32 channels
    of one tap, then the taps' last 16 channels paired (taps 3r / 3r+1: next pixel; taps 2 / 5: next row; tap 8 alone, upper
    half zero) - the byte offsets of a lane group then differ by per-lane constants only (csrc/fused_f16.hip: bA / bP / bR /
    bS).  f1, f2 (pair 1) and the FM part: one 32-channel step per tap."""
    assert cin == 48
    merged = [[(t, 0), (t, 8), (t, 16), (t, 24)] for t in range(9)]
    merged += [[(3 * r, 32), (3 * r, 40), (3 * r + 1, 32), (3 * r + 1, 40)] for r in range(3)]
    merged += [[(2, 32), (2, 40), (5, 32), (5, 40)], [(8, 32), (8, 40), None, None]]
    for f in range(2 * pair):
        merged += [[(t, cin + 32 * f + 8 * q) for q in range(4)] for t in range(9)]
    nin = cin + 64 * pair
    fm = [[(t, nin + 8 * q) for q in range(4)] for t in range(9)]
    return merged, fm


def _frag16(w9: torch.Tensor, step, rb: int) -> torch.Tensor:
    """One A fragment [64 lanes][8] of v_mfma_f32_16x16x32: lane (q, i): output row i of block rb = channel 8 (i >> 2) + 4 rb + (i & 3)
    (so that a lane's accumulators of the two blocks are 8 CONSECUTIVE channels), k = the 8 input channels of octet q."""
    rows = torch.tensor([8 * (i >> 2) + 4 * rb + (i & 3) for i in range(16)], device=w9.device)
    out = torch.zeros(4, 16, 8, dtype=w9.dtype, device=w9.device)
    for q, d in enumerate(step):
        if d is not None:
            tap, c0 = d
            out[q] = w9[rows][:, c0:c0 + 8, tap]
    return out.reshape(64, 8)


_F16_GATHER: dict = {}


def pack_fused_f16(weights: Sequence[torch.Tensor], cin: int = 48) -> torch.Tensor:
    """conv1..conv4 of a cin == 48 dense block -> the fragment stream of csrc/fused_f16.hip: per pair (conv a, conv b) the merged
    steps [a rb0, a rb1, b rb0, b rb1] followed by the FM steps [b rb0, b rb1]: 74 + 146 fragments -> f16 [220, 64, 8].
    The layout is a fixed gather of the four weight tensors: its index table is learnt once per process from index-valued
    stand-ins (`_pack_fused_f16_steps`, a fragment-by-fragment construction) and applied as ONE gather afterwards."""
    assert cin == 48 and len(weights) == 4
    ws = []
    for k, w in enumerate(weights):
        w = w.detach().float()
        if w.dim() == 5:
            w = w[:, :, 0]
        assert w.shape == (32, cin + 32 * k, 3, 3), tuple(w.shape)
        ws.append(w)
    idx = _F16_GATHER.get("idx")
    if idx is None:
        fakes, off = [], 0
        for w in ws:
            fakes.append(torch.arange(off + 1, off + w.numel() + 1, dtype=torch.float64).reshape(w.shape))
            off += w.numel()
        idx = _pack_fused_f16_steps(fakes, cin).reshape(-1).round().long() - 1            # -1: zero padding
        _F16_GATHER["idx"] = idx = torch.where(idx < 0, torch.full_like(idx, off), idx)   # -> the appended zero
    dev = ws[0].device
    if _F16_GATHER.get("dev") != dev:
        _F16_GATHER["dev"], _F16_GATHER["idx_dev"] = dev, idx.to(dev)
    flat = torch.cat([w.reshape(-1) for w in ws] + [torch.zeros(1, dtype=torch.float32, device=dev)])
    return _operand(flat[_F16_GATHER["idx_dev"]].reshape(220, 64, 8))


def _pack_fused_f16_steps(weights: Sequence[torch.Tensor], cin: int = 48) -> torch.Tensor:
    """the layout of pack_fused_f16, built fragment by fragment (values of `weights` pass through unchanged) -> [220, 64, 8]"""
    out = []
    for pair in (0, 1):
        nin = cin + 64 * pair
        ws = []
        for j in (0, 1):
            w = weights[2 * pair + j].detach().float()
            if w.dim() == 5:
                w = w[:, :, 0]
            assert w.shape == (32, nin + 32 * j, 3, 3), tuple(w.shape)
            ws.append(w.reshape(32, nin + 32 * j, 9))
        merged, fm = f16_steps(pair, cin)
        for st in merged:
            out += [_frag16(ws[0], st, 0), _frag16(ws[0], st, 1), _frag16(ws[1], st, 0), _frag16(ws[1], st, 1)]
        for st in fm:
            out += [_frag16(ws[1], st, 0), _frag16(ws[1], st, 1)]
    res = torch.stack(out)
    assert res.shape[0] == 220
    return res


def pack_f5_partial16(w5: torch.Tensor, cin: int = 48) -> torch.Tensor:
    """F's temporal conv5 (cout <= 3, cin + 128, 3, 1, 1) as the A fragments of the 16x16x32 partial products: row 4 tap + oc
    (12 of 16 used), K = 32 input channels in the reference's concat order: x2[0:32], x2[32:48] (upper half zero), f1, f2 (pair
    0), f3, f4 (pair 1) -> f16 [6, 64, 8]."""
    w = w5.detach().float()
    cout, ctot = w.shape[0], w.shape[1]
    assert w.shape[2:] == (3, 1, 1) and cout <= 3 and ctot == cin + 128 and cin == 48, tuple(w.shape)
    wk = torch.zeros(16, ctot + 16, dtype=torch.float32, device=w.device)       # + 16 zero columns behind the last channel
    for tap in range(3):
        wk[tap * 4: tap * 4 + cout, :ctot] = w[:, :, tap, 0, 0]
    zero = ctot
    frags = []
    for c0, nreal in ((0, 32), (32, 16), (48, 32), (80, 32), (112, 32), (144, 32)):
        cols = [c0 + k if k < nreal else zero for k in range(32)]
        frags.append(wk[:, cols].reshape(16, 4, 8).permute(1, 0, 2).reshape(64, 8))
    return _operand(torch.stack(frags))


def pack_f5_partial(w5: torch.Tensor, cin: int = 48) -> torch.Tensor:
    """Temporal conv5 of F (cout, cin + 128, 3, 1, 1), cout <= 3, as the A fragments of the conv5 partial products the
    fused F launches emit (csrc/fused_f.hip): row = 4 tap + oc (rows 0-2, 4-6, 8-10 of 32 used), one 32x32x16 fragment per 16 input
    channels in the reference's concat order [x2 | f1 | f2 | f3 | f4] -> f16 [(cin + 128) / 16, 64, 8]."""
    w = w5.detach().float()
    cout, ctot = w.shape[0], w.shape[1]
    assert w.shape[2:] == (3, 1, 1) and cout <= 3 and ctot == cin + 128 and ctot % 16 == 0, tuple(w.shape)
    wk = torch.zeros(32, ctot, dtype=torch.float32, device=w.device)
    for tap in range(3):
        wk[tap * 4: tap * 4 + cout] = w[:, :, tap, 0, 0]
    nfrag = ctot // 16
    return _operand(wk.reshape(32, nfrag, 2, 8).permute(1, 2, 0, 3).reshape(nfrag, 64, 8))


def pack_conv_planes(weight: torch.Tensor, cin: int) -> torch.Tensor:
    """Conv3d weight (cout, cin + 128*j, kt, 3, 3), kt in {1, 3}, of a FeatureCalapseBlock-style dense block
    (inputs first, then 128-channel features) -> f16 [cout/32, nstages*18, 64, 8] for selfc_conv_planes_run.
    Buffer planes: the cin inputs zero-padded to whole 32-channel planes, then the features (32 | 128).
    K order per output group: for temporal tap: for plane: for spatial tap: for 32 channels."""
    w = weight.detach().float()
    cout, ctot, kt = w.shape[0], w.shape[1], w.shape[2]
    assert w.shape[3:] == (3, 3) and kt in (1, 3) and cout % 32 == 0 and (ctot - cin) % 32 == 0
    pin = roundup(cin, 32) // 32
    nplanes = pin + (ctot - cin) // 32
    wp = torch.zeros(cout, nplanes * 32, kt, 9, dtype=torch.float32, device=w.device)
    wp[:, :cin] = w[:, :cin].reshape(cout, cin, kt, 9)
    wp[:, pin * 32:] = w[:, cin:].reshape(cout, ctot - cin, kt, 9)
    # (cout, plane, 32, kt, 9) -> (cout, kt, plane, 9, 32)
    wk = wp.reshape(cout, nplanes, 32, kt, 9).permute(0, 3, 1, 4, 2).reshape(cout, kt * nplanes * 9 * 32)
    nfrag = wk.shape[1] // 16
    z = cout // 32
    frag = wk.reshape(z, 32, nfrag, 2, 8).permute(0, 2, 3, 1, 4).reshape(z, nfrag, 64, 8)
    return _operand(frag)


# ----------------------------------------------------------------------------------------------------------
# backward (csrc/backward.hip): transposed, tap-flipped weights for the generic plane-list conv
# ----------------------------------------------------------------------------------------------------------

def pack_planes_generic(wt: torch.Tensor) -> torch.Tensor:
    """wt (G*32 out, P*32 in, kt, ks) fp32, ks = 9 (3x3 taps) or 1 (centre tap only) -> f16
    [G][kt*P*(18|2)][64][8]: per 32-channel output group, K order = temporal tap, input plane, spatial tap,
    32 channels (the stage order of conv3x3_kernel's generic mode)."""
    cout, cin, kt, ks = wt.shape
    assert cout % 32 == 0 and cin % 32 == 0 and ks in (1, 9) and kt in (1, 3)
    z, pl = cout // 32, cin // 32
    wk = wt.reshape(z, 32, pl, 32, kt, ks).permute(0, 1, 4, 2, 5, 3).reshape(z, 32, kt * pl * ks * 32)
    nfrag = wk.shape[2] // 16
    frag = wk.reshape(z, 32, nfrag, 2, 8).permute(0, 2, 3, 1, 4).reshape(z, nfrag, 64, 8)
    return _operand(frag)


def pack_t5_bwd(wt: torch.Tensor) -> torch.Tensor:
    """conv5^T of a temporal dense block for the frame-walking temporal-conv kernel (dense_conv.hip: EPI_T5B).
    wt (Z*32 out, KS*32 in, 3 taps) -> f16 [Z][3][KS][2][64][8]: per 32-channel output plane, 16x16x32 A fragments
    W[16 o + (lane & 15)][32 ks + 8 (lane >> 4) + j] of out tile o (0, 1), k-step ks, tap."""
    cout, cin, kt = wt.shape
    assert cout % 32 == 0 and cin % 32 == 0 and kt == 3
    z, ks = cout // 32, cin // 32
    frag = wt.reshape(z, 2, 16, ks, 4, 8, 3).permute(0, 6, 3, 1, 4, 2, 5).reshape(z, 3, ks, 2, 64, 8)
    return _operand(frag)


def pack_subnet_bwd(weights: Sequence[torch.Tensor], cin: int, cout: int, temporal: bool):
    """conv1..conv5 weights of a DenseBlock (temporal=False) / D2DTInput (temporal=True) -> the five packed
    data-gradient convs of selfc_subnet_bwd: (wt5, [wtd3, wtd2, wtd1], wtx).

    y[o][p] = sum W[o][c][tap] x[c][p + off(tap)]  =>  dx[c][q] = sum W[o][c][ntap-1-tap'] dy[o][q + off(tap')]:
    the gradient conv swaps in/out channels and reverses the taps.  Gradient planes are ordered
    [dpre4, dpre3, dpre2, dpre1] (plane p <-> conv 4-p); output channel groups of conv5^T are
    [x (cin, zero padded to 32s) | f1 | f2 | f3 | f4]."""
    dev = weights[0].device
    nx, ng = roundup(cin, 32) // 32, roundup(cout, 32) // 32
    w3 = []
    for k in range(1, 5):
        w = weights[k - 1].detach().float()
        if w.dim() == 5:
            w = w[:, :, 0]
        assert w.shape == (32, cin + 32 * (k - 1), 3, 3), tuple(w.shape)
        w3.append(w.reshape(32, -1, 9).flip(2))                      # [o][c][tap'] = W[o][c][8 - tap']
    w5 = weights[4].detach().float()
    if temporal:
        assert w5.shape == (cout, cin + 128, 3, 1, 1), tuple(w5.shape)
        w5 = w5[:, :, :, 0, 0].flip(2)                                # (cout, C, 3) temporal taps reversed
        kt, ks = 3, 1
    else:
        assert w5.shape == (cout, cin + 128, 3, 3), tuple(w5.shape)
        w5 = w5.reshape(cout, cin + 128, 9).flip(2)
        kt, ks = 1, 9
    # conv5^T: out (nx+4)*32, in ng*32
    t5 = torch.zeros((nx + 4) * 32, ng * 32, kt, ks, dtype=torch.float32, device=dev)
    w5t = w5.permute(1, 0, 2)                                         # (C, cout, taps)
    w5t = w5t.reshape(cin + 128, cout, kt, ks)
    t5[:cin, :cout] = w5t[:cin]
    t5[nx * 32:, :cout] = w5t[cin:]
    wt5 = pack_t5_bwd(t5[:, :, :, 0]) if temporal else pack_planes_generic(t5)
    # dpre_j, j = 3, 2, 1: in planes dpre4..dpre_{j+1}
    wtd = []
    for j in (3, 2, 1):
        npl = 4 - j
        t = torch.zeros(32, npl * 32, 1, 9, dtype=torch.float32, device=dev)
        for pl_i in range(npl):
            k = 4 - pl_i
            sl = w3[k - 1][:, cin + 32 * (j - 1): cin + 32 * j, :]    # (o, r, tap')
            t[:, pl_i * 32:(pl_i + 1) * 32, 0, :] = sl.permute(1, 0, 2)
        wtd.append(pack_planes_generic(t))
    # dx: in planes dpre4..dpre1, out nx*32
    t = torch.zeros(nx * 32, 4 * 32, 1, 9, dtype=torch.float32, device=dev)
    for pl_i in range(4):
        k = 4 - pl_i
        t[:cin, pl_i * 32:(pl_i + 1) * 32, 0, :] = w3[k - 1][:, :cin, :].permute(1, 0, 2)
    wtx = pack_planes_generic(t)
    return wt5, wtd, wtx


def pack_pointwise_T(weight: torch.Tensor) -> torch.Tensor:
    """1x1(x1) conv weight (cout, cin, ...) -> its data-gradient conv for selfc_bwd_conv_planes (sp1): out = cin padded
    to 32-channel groups, in = cout padded to 32-channel planes."""
    w = weight.detach().float().reshape(weight.shape[0], -1)
    cout, cin = w.shape
    t = torch.zeros(roundup(cin, 32), roundup(cout, 32), 1, 1, dtype=torch.float32, device=w.device)
    t[:cin, :cout, 0, 0] = w.t()
    return pack_planes_generic(t)


# ----------------------------------------------------------------------------------------------------------
# gather plans
# ----------------------------------------------------------------------------------------------------------

class PackPlan:
    """Kernel-layout tensors of a module as ONE gather from its flat parameter vector.

    build(params) -> {name: (tensor, 'w' | 'b')} must be composed of the pack_* functions of this file ('w': MFMA
    operand dtype, 'b': fp32).  The plan is learnt once from index-valued stand-ins; run() then costs three device ops
    whatever the number of packed tensors."""

    ALIGN = 128      # elements: keeps every output segment 256-byte aligned

    def __init__(self, params: Sequence[torch.Tensor], build):
        global _RAW
        dev = params[0].device
        total = sum(p.numel() for p in params)
        if total + 2 >= (1 << 24):
            raise ValueError("PackPlan: too many parameters for exact float32 indices")
        fakes, off = [], 0
        for p in params:
            fakes.append(torch.arange(off + 1, off + p.numel() + 1, dtype=torch.float32, device=dev).reshape(p.shape))
            off += p.numel()
        prev, _RAW = _RAW, True
        try:
            outs = build(fakes)
        finally:
            _RAW = prev
        self.total = total
        self.items = []                       # (name, kind, start, numel, shape)
        idx_w, idx_b = [], []
        cur = {"w": 0, "b": 0}
        for name, (t, kind) in outs.items():
            idx = t.reshape(-1).round().long() - 1
            idx = torch.where(idx < 0, torch.full_like(idx, total), idx)          # zero fill -> the appended 0
            if idx.numel() and int(idx.max()) > total:       # a pack function that rounded its stand-ins (not built on _operand): refuse
                raise ValueError(f"PackPlan: {name} is not a pure gather of the parameters (index {int(idx.max())} > {total})")
            pad = (-idx.numel()) % self.ALIGN
            if pad:
                idx = torch.cat((idx, torch.full((pad,), total, dtype=torch.long, device=dev)))
            (idx_w if kind == "w" else idx_b).append(idx)
            self.items.append((name, kind, cur[kind], t.numel(), tuple(t.shape)))
            cur[kind] += idx.numel()
        self.nw = cur["w"]
        self.idx = torch.cat(idx_w + idx_b)

    def run(self, params: Sequence[torch.Tensor]):
        flat = torch.cat([p.detach().reshape(-1).float() for p in params] + [torch.zeros(1, dtype=torch.float32, device=self.idx.device)])
        g = flat[self.idx]
        gw, gb = g[:self.nw].to(F16), g[self.nw:]
        out = {}
        for name, kind, start, numel, shape in self.items:
            src = gw if kind == "w" else gb
            out[name] = src[start:start + numel].view(shape)
        return out


def subnet_pack_entries(prefix: str, weights: Sequence[torch.Tensor], biases: Sequence[torch.Tensor], cin: int, cout: int,
                        temporal: bool, partner_w5: torch.Tensor = None, with_bwd: bool = True):
    """{prefix + name: (tensor, kind)} of everything the kernels need from one DenseBlock / D2DTInput: forward
    fragments (runtime.PackedSubnet) and, with_bwd, the gradient convs (autograd.PackedSubnetBwd)."""
    dev = weights[0].device
    e = {}
    for i in range(4):
        e[f"{prefix}w3_{i}"] = (pack_conv3x3(weights[i], cin, i + 1), "w")
        e[f"{prefix}b3_{i}"] = (pad_bias(biases[i], 64, dev), "b")
    if temporal:
        e[f"{prefix}w5"] = (pack_tconv5([weights[4]] + ([partner_w5] if partner_w5 is not None else []), cin), "w")
    else:
        e[f"{prefix}w5"] = (pack_conv3x3(weights[4], cin, 5), "w")
    e[f"{prefix}b5"] = (pad_bias(biases[4], 64, dev), "b")
    if cin == 3 and temporal:
        e[f"{prefix}wfused"] = (pack_fused_gh(list(weights[:4]), 3), "w")
    if cin == 48:
        # [the 32x32x16 stream (csrc/fused_f.hip): 216 fragments | the 16x16x32 stream (csrc/fused_f16.hip): 220 fragments]
        e[f"{prefix}wfused"] = (torch.cat((pack_fused_f(list(weights[:4]), 48), pack_fused_f16(list(weights[:4]), 48))), "w")
        if temporal and cout <= 3:
            e[f"{prefix}w5p"] = (torch.cat((pack_f5_partial(weights[4], 48), pack_f5_partial16(weights[4], 48))), "w")
    if with_bwd and cout <= 96:
        wt5, wtd, wtx = pack_subnet_bwd(weights, cin, cout, temporal)
        e[f"{prefix}wt5"] = (wt5, "w")
        for i in range(3):
            e[f"{prefix}wtd_{i}"] = (wtd[i], "w")
        e[f"{prefix}wtx"] = (wtx, "w")
    return e
