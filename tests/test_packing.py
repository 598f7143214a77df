"""Host logic: the MFMA fragment packers (selfc_amd/packing.py).  The kernels'
K order and lane maps are re-stated here in plain torch ("what the kernel
contracts"), fed with the packed fragments, and compared with F.conv2d / the
oracle on the reference-layout weights."""
import pytest
import torch
import torch.nn.functional as F

from oracle import selfc_oracle as O
from selfc_amd import packing as P


def stages(cin, layer):
    """mirror of build_stages()/stage_of() in csrc/dense_conv.hip"""
    st = []
    if cin <= 3:
        st.append(("im2col", 0, 32))
        fbase = 0
    else:
        cin16 = P.roundup(cin, 16)
        for c0 in range(0, cin16, 32):
            st.append(("dense", c0, 32 if cin16 - c0 >= 32 else 16))
        fbase = P.roundup(cin, 32)
    for i in range(layer - 1):
        st.append(("dense", fbase + 32 * i, 32))
    return st


def unpack_a32(frag):
    """[nfrag,64,8] -> W[32][K] with W[lane&31][16 f + 8 (lane>>5) + j]"""
    nfrag = frag.shape[0]
    return frag.float().reshape(nfrag, 2, 32, 8).permute(2, 0, 1, 3).reshape(32, nfrag * 16)


def act_k(dense, x1, cin, layer):
    """dense (N,H,W,C), x1 (N,H,W,4) -> (N,H,W,K) in kernel K order (zero padded halo)"""
    n, h, w, _ = dense.shape
    dp = F.pad(dense, (0, 0, 1, 1, 1, 1))
    xp = F.pad(x1, (0, 0, 1, 1, 1, 1)) if x1 is not None else None
    cols = []
    for kind, coff, width in stages(cin, layer):
        if kind == "im2col":
            v = torch.zeros(n, h, w, 32)
            for tap in range(9):
                ky, kx = divmod(tap, 3)
                v[..., tap * cin:(tap + 1) * cin] = xp[:, ky:ky + h, kx:kx + w, :cin]
            cols.append(v)
        else:
            for tap in range(9):
                ky, kx = divmod(tap, 3)
                cols.append(dp[:, ky:ky + h, kx:kx + w, coff:coff + width])
    return torch.cat(cols, dim=-1)


def make_dense(x, feats, cin):
    """NCHW x and list of NCHW 32-ch features -> NHWC dense buffer in the kernel layout"""
    n, _, h, w = x.shape
    d = torch.zeros(n, h, w, P.dense_channels(cin))
    fbase = 0
    if cin > 3:
        d[..., :cin] = x.permute(0, 2, 3, 1)
        fbase = P.roundup(cin, 32)
    for i, f in enumerate(feats):
        d[..., fbase + 32 * i: fbase + 32 * (i + 1)] = f.permute(0, 2, 3, 1)
    return d


@pytest.mark.parametrize("cin,layer,cout,conv3d", [(48, 1, 32, True), (48, 4, 32, True), (3, 1, 32, True),
                                                   (3, 3, 32, True), (9, 2, 32, False), (9, 5, 3, False),
                                                   (3, 5, 9, False), (64, 3, 32, True), (12, 4, 32, False), (2, 2, 32, False)])
def test_pack_conv3x3(cin, layer, cout, conv3d):
    g = torch.Generator().manual_seed(cin * 10 + layer)
    ctot = cin + 32 * (layer - 1)
    w = torch.randn(cout, ctot, 3, 3, generator=g) * 0.1
    x = torch.randn(2, cin, 6, 7, generator=g)
    feats = [torch.randn(2, 32, 6, 7, generator=g) for _ in range(layer - 1)]
    ref = F.conv2d(torch.cat([x] + feats, 1), w.half().float(), None, 1, 1)          # f16-rounded weights
    frag = P.pack_conv3x3(w.unsqueeze(2) if conv3d else w, cin, layer)
    assert frag.dtype == P.F16 and frag.shape[1:] == (64, 8)
    wk = unpack_a32(frag)
    x1 = None
    if cin <= 3:
        x1 = torch.zeros(2, 6, 7, 4)
        x1[..., :cin] = x.permute(0, 2, 3, 1)
    ak = act_k(make_dense(x, feats, cin), x1, cin, layer)
    assert ak.shape[-1] == wk.shape[1]
    out = torch.einsum("ok,nhwk->nohw", wk, ak)
    assert torch.allclose(out[:, :cout], ref, atol=1e-4, rtol=1e-4)
    assert out[:, cout:].abs().max() == 0 if cout < 32 else True


@pytest.mark.parametrize("cin,cout,nets", [(48, 3, 1), (3, 48, 2), (3, 64, 1), (64, 64, 1), (12, 3, 1), (3, 12, 2)])
def test_pack_tconv5(cin, cout, nets):
    g = torch.Generator().manual_seed(cin + cout)
    T, B, h, w = 5, 2, 3, 4
    ws = [torch.randn(cout, cin + 128, 3, 1, 1, generator=g) * 0.1 for _ in range(nets)]
    frag = P.pack_tconv5(ws, cin)
    hasx = cin <= 3
    kd = P.dense_channels(cin) // 32
    ks = kd + (1 if hasx else 0)
    ot = P.roundup(cout, 16) // 16
    assert frag.shape == (3, nets, ks, ot, 64, 8)
    # W[tap][net][16 o + (lane&15)][32 ks + 8 (lane>>4) + j]
    wk = frag.float().reshape(3, nets, ks, ot, 4, 16, 8).permute(1, 0, 3, 5, 2, 4, 6).reshape(nets, 3, ot * 16, ks * 32)
    for n in range(nets):
        x = torch.randn(B * T, cin, h, w, generator=g)
        feats = [torch.randn(B * T, 32, h, w, generator=g) for _ in range(4)]
        d = torch.cat([x] + feats, 1)                                            # reference cat order
        p = {"conv5.weight": ws[n].half().float()}
        # oracle's temporal conv on the concatenated features (bypass conv1-4): restate with einsum
        d5 = d.reshape(B, T, -1, h, w)
        ref = torch.zeros(B, T, cout, h, w)
        for dt in (-1, 0, 1):
            lo, hi = max(0, -dt), min(T, T - dt)
            ref[:, lo:hi] += torch.einsum("oc,btchw->btohw", p["conv5.weight"][:, :, dt + 1, 0, 0], d5[:, lo + dt:hi + dt])
        # kernel-side K vector: [x padded to 32 if cin<=3] + dense buffer channels
        dense = make_dense(x, feats, cin)
        kv = dense
        if hasx:
            xk = torch.zeros(B * T, h, w, 32)
            xk[..., :cin] = x.permute(0, 2, 3, 1)
            kv = torch.cat((xk, dense), -1)
        kv = kv.reshape(B, T, h, w, -1)
        out = torch.zeros(B, T, ot * 16, h, w)
        for tap in range(3):
            dt = tap - 1
            lo, hi = max(0, -dt), min(T, T - dt)
            out[:, lo:hi] += torch.einsum("ok,bthwk->btohw", wk[n, tap], kv[:, lo + dt:hi + dt])
        assert torch.allclose(out[:, :, :cout], ref, atol=1e-4, rtol=1e-4)


def test_pad_bias():
    b = P.pad_bias(torch.arange(3.0))
    assert b.shape == (64,) and b[:3].tolist() == [0.0, 1.0, 2.0] and b[3:].abs().sum() == 0


@pytest.mark.parametrize("h,w", [(16, 16), (64, 112), (20, 36), (8, 12), (33, 65)])
def test_pool_weight_map_equals_fc_of_adaptive_pool(h, w):
    """GlobalAgg's fc(adaptive_avg_pool2d(x,32x32)) folded into one HxW map (overlapping and replicating bins)."""
    g = torch.Generator().manual_seed(h * 100 + w)
    fcw = torch.randn(1, 1024, generator=g)
    x = torch.randn(3, 5, h, w, generator=g)
    ref = (F.adaptive_avg_pool2d(x, (32, 32)).reshape(3, 5, 1024) @ fcw.t()).squeeze(-1)
    got = (x.reshape(3, 5, h * w) * P.pool_weight_map(fcw, h, w)).sum(-1)
    assert torch.allclose(got, ref, atol=2e-5, rtol=1e-5)


def test_pack_pointwise_lane_map():
    g = torch.Generator().manual_seed(3)
    w = torch.randn(720, 256, 1, 1, 1, generator=g)
    fr = P.pack_pointwise(w)
    assert fr.shape == (45, 8, 64, 8) and fr.dtype == P.F16
    # W[16 o + (lane&15)][32 ks + 8 (lane>>4) + j]
    wk = fr.float().reshape(45, 8, 4, 16, 8).permute(0, 3, 1, 2, 4).reshape(720, 256)
    assert torch.equal(wk, w.reshape(720, 256).half().float())
    fr = P.pack_pointwise(torch.randn(48, 64, generator=g))        # cout padded to whole tiles
    assert fr.shape == (3, 2, 64, 8)


@pytest.mark.parametrize("layer", [1, 2, 4])
def test_pack_fused_gh_stream(layer):
    """fused conv1..4 stream of csrc/fused_gh.hip: per conv [im2col48: k = tap*4 + c][features tap-major]."""
    g = torch.Generator().manual_seed(40 + layer)
    ws = [torch.randn(32, 3 + 32 * i, 1, 3, 3, generator=g) * 0.1 for i in range(4)]
    stream = P.pack_fused_gh(ws, 3)
    assert stream.shape == (120, 64, 8)
    layer_off = [0, 3, 24, 63, 120]
    wk = unpack_a32(stream[layer_off[layer - 1]:layer_off[layer]])          # (32, 48 + 288*(layer-1))
    x = torch.randn(2, 3, 6, 7, generator=g)
    feats = [torch.randn(2, 32, 6, 7, generator=g) for _ in range(layer - 1)]
    ref = F.conv2d(torch.cat([x] + feats, 1), ws[layer - 1][:, :, 0].half().float(), None, 1, 1)
    xp = F.pad(x.permute(0, 2, 3, 1), (0, 0, 1, 1, 1, 1))
    cols = [torch.zeros(2, 6, 7, 48)]
    for tap in range(9):
        ky, kx = divmod(tap, 3)
        cols[0][..., tap * 4: tap * 4 + 3] = xp[:, ky:ky + 6, kx:kx + 7, :]
    for f in feats:
        fp = F.pad(f.permute(0, 2, 3, 1), (0, 0, 1, 1, 1, 1))
        for tap in range(9):
            ky, kx = divmod(tap, 3)
            cols.append(fp[:, ky:ky + 6, kx:kx + 7, :])
    ak = torch.cat(cols, -1)
    assert ak.shape[-1] == wk.shape[1]
    out = torch.einsum("ok,nhwk->nohw", wk, ak)
    assert torch.allclose(out, ref, atol=1e-4, rtol=1e-4)


def test_gmm_head_perm():
    """output-channel permutation of the GMM head for the fused head + sampler kernel: new channel (3k + j)*48 + c holds the
    reference's channel (c*K + k)*3 + j, i.e. `parameters.reshape(b, hf_dim, K, 3, ...)` (SelfC_GMM_arch_inv.py:382-386)."""
    perm = P.gmm_head_perm(48, 5)
    assert perm.shape == (720,) and sorted(perm.tolist()) == list(range(720))
    raw = torch.arange(720.0)
    p3 = raw.reshape(48, 5, 3)                       # [c][k][pi | log-sigma | mu]
    new = raw[perm].reshape(5, 3, 48)                # [k][j][c]
    assert torch.equal(new, p3.permute(1, 2, 0))


@pytest.mark.parametrize("pair", [0, 1])
def test_pack_fused_f_stream(pair):
    """pairwise-fused F stream of csrc/fused_f.hip: per pair [merged steps: (conv a, conv b) fragment pairs, source
    group major / tap-major / 16-channel k-step minor][conv b's 18 FM fragments, tap-major].  The kernel's step decode
    (merged_steps / fm_steps) is restated here on an image whose pixel holds [x2 48 | f1 32 | f2 32] channels."""
    g = torch.Generator().manual_seed(70 + pair)
    ws = [torch.randn(32, 48 + 32 * i, 1, 3, 3, generator=g) * 0.1 for i in range(4)]
    stream = P.pack_fused_f(ws, 48)
    assert stream.shape == (216, 64, 8)
    nin = 48 + 64 * pair
    s1 = 9 * nin // 16
    base = 0 if pair == 0 else 72
    merged = stream[base: base + 2 * s1].reshape(s1, 2, 64, 8)
    wa, wb = unpack_a32(merged[:, 0].contiguous()), unpack_a32(merged[:, 1].contiguous())       # (32, 16*s1)
    wf = unpack_a32(stream[base + 2 * s1: base + 2 * s1 + 18])                                    # (32, 288)
    n, h, w = 2, 5, 6
    img = torch.randn(n, nin, h, w, generator=g)            # x2 | f1 | f2 in the reference's concat order
    fm = torch.randn(n, 32, h, w, generator=g)
    ip = F.pad(img.permute(0, 2, 3, 1), (0, 0, 1, 1, 1, 1))
    cols = []
    for st in range(s1):                                    # kernel decode of merged step st
        if st < 27:
            tap, ks = st // 3, st % 3
        elif st < 45:
            tap, ks = (st - 27) >> 1, 3 + ((st - 27) & 1)
        else:
            tap, ks = (st - 45) >> 1, 5 + ((st - 45) & 1)
        ky, kx = divmod(tap, 3)
        cols.append(ip[:, ky:ky + h, kx:kx + w, 16 * ks:16 * ks + 16])
    ak = torch.cat(cols, -1)
    fp = F.pad(fm.permute(0, 2, 3, 1), (0, 0, 1, 1, 1, 1))
    fk = torch.cat([fp[:, (st >> 1) // 3:(st >> 1) // 3 + h, (st >> 1) % 3:(st >> 1) % 3 + w, 16 * (st & 1):16 * (st & 1) + 16]
                    for st in range(18)], -1)
    out_a = torch.einsum("ok,nhwk->nohw", wa, ak)
    out_b = torch.einsum("ok,nhwk->nohw", wb, ak) + torch.einsum("ok,nhwk->nohw", wf, fk)
    ref_a = F.conv2d(img, ws[2 * pair][:, :, 0].half().float(), None, 1, 1)
    ref_b = F.conv2d(torch.cat((img, fm), 1), ws[2 * pair + 1][:, :, 0].half().float(), None, 1, 1)
    assert torch.allclose(out_a, ref_a, atol=1e-4, rtol=1e-4)
    assert torch.allclose(out_b, ref_b, atol=1e-4, rtol=1e-4)


def test_pack_f5_partial():
    """conv5 partial products of the fused F launches: rows 4 tap + oc, one fragment per 16 channels of [x2 | f1..f4];
    summing P[t-1][tap 0] + P[t][tap 1] + P[t+1][tap 2] reproduces the temporal conv5 (Subnet_constructor.py:106,130)."""
    g = torch.Generator().manual_seed(91)
    w5 = torch.randn(3, 176, 3, 1, 1, generator=g) * 0.1
    fr = P.pack_f5_partial(w5, 48)
    assert fr.shape == (11, 64, 8)
    wk = unpack_a32(fr)                                        # (32, 176)
    assert torch.count_nonzero(wk[11:]) == 0 and torch.count_nonzero(wk[3::4]) == 0
    d = torch.randn(1, 176, 5, 4, 3, generator=g)              # (b, C, T, h, w)
    ref = F.conv3d(d, w5.half().float(), None, 1, (1, 0, 0))   # (1, 3, 5, 4, 3)
    pp = torch.einsum("rc,bcthw->brthw", wk[:12], d)           # (1, 12, T, h, w)
    out = torch.zeros_like(ref)
    for t in range(5):
        out[:, :, t] += pp[:, 4:7, t]
        if t > 0:
            out[:, :, t] += pp[:, 0:3, t - 1]
        if t + 1 < 5:
            out[:, :, t] += pp[:, 8:11, t + 1]
    assert torch.allclose(out, ref, atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("cin,nfeat,kt,cout", [(48, 0, 3, 128), (48, 2, 1, 128), (192, 4, 3, 64)])
def test_pack_conv_planes(cin, nfeat, kt, cout):
    """generic plane-list conv (FeatureCalapseBlock): K order = temporal tap, plane, spatial tap, 32 channels."""
    g = torch.Generator().manual_seed(cin + nfeat + kt)
    ctot = cin + 128 * nfeat
    w = torch.randn(cout, ctot, kt, 3, 3, generator=g) * 0.05
    frag = P.pack_conv_planes(w, cin)
    pin = P.roundup(cin, 32) // 32
    nplanes = pin + 4 * nfeat
    assert frag.shape == (cout // 32, kt * nplanes * 18, 64, 8)
    B, T, h, wd = 1, 3, 4, 5
    x = torch.randn(B, ctot, T, h, wd, generator=g)
    pad = (1, 1, 1) if kt == 3 else (0, 1, 1)
    ref = F.conv3d(x, w.half().float(), None, 1, pad)                       # (B, cout, T, h, w)
    # plane buffer: inputs zero padded to whole planes, then features
    buf = torch.zeros(B, T, h, wd, nplanes * 32)
    buf[..., :cin] = x[:, :cin].permute(0, 2, 3, 4, 1)
    buf[..., pin * 32:] = x[:, cin:].permute(0, 2, 3, 4, 1)
    bp = F.pad(buf, (0, 0, 1, 1, 1, 1, 1, 1) if kt == 3 else (0, 0, 1, 1, 1, 1))
    cols = []
    for ti in range(kt):
        for pl in range(nplanes):
            for tap in range(9):
                ky, kx = divmod(tap, 3)
                src = bp[:, ti:ti + T] if kt == 3 else bp
                cols.append(src[:, :, ky:ky + h, kx:kx + wd, 32 * pl:32 * pl + 32])
    ak = torch.cat(cols, -1)
    for z in range(cout // 32):
        wk = unpack_a32(frag[z])
        out = torch.einsum("ok,bthwk->bothw", wk, ak)
        assert torch.allclose(out, ref[:, 32 * z:32 * z + 32], atol=2e-4, rtol=1e-4)


def test_pack_plan_equals_direct_packing():
    """PackPlan learns the gather map from index-valued stand-ins; its output must be bit-identical to calling the
    pack functions on the real weights (forward fragments, biases and the gradient convs)."""
    import torch
    from selfc_amd import packing as P
    torch.manual_seed(0)
    for cin, cout, temporal, partner in [(48, 3, True, False), (3, 48, True, True), (9, 3, False, False), (3, 9, False, False),
                                         (64, 64, True, False)]:
        ws = [torch.randn(32, cin + 32 * k, *((1, 3, 3) if temporal else (3, 3))) for k in range(4)]
        ws.append(torch.randn(cout, cin + 128, *((3, 1, 1) if temporal else (3, 3))))
        bs = [torch.randn(32) for _ in range(4)] + [torch.randn(cout)]
        pw5 = torch.randn_like(ws[4]) if partner else None
        params = [t for pair in zip(ws, bs) for t in pair] + ([pw5] if partner else [])

        def build(ps):
            return P.subnet_pack_entries("s.", ps[0:10:2], ps[1:10:2], cin, cout, temporal, partner_w5=ps[10] if partner else None)

        plan = P.PackPlan(params, build)
        got = plan.run(params)
        want = build(params)
        assert set(got) == set(want)
        for name, (t, kind) in want.items():
            assert got[name].dtype == (P.F16 if kind == "w" else torch.float32), name
            assert got[name].shape == t.shape and torch.equal(got[name], t), name
            assert got[name].data_ptr() % 16 == 0, name
        # changed weights -> same plan, new values
        params2 = [p * 0.5 for p in params]
        got2 = plan.run(params2)
        assert torch.equal(got2["s.w3_0"], P.pack_conv3x3(params2[0], cin, 1))


def test_widen_dense_params_is_an_exact_equivalent():
    """packing.widen_dense_params: a dense block with growth 12 and 24 channels (the codec variant's STP subnets) packed into the
    kernels' growth-32 / 64-channel layout computes the same function (float64: the added terms are exact zeros)."""
    import torch.nn.functional as F
    from selfc_amd.packing import widen_dense_params
    torch.manual_seed(0)
    cin, cout, gc = 24, 24, 12
    ws = [torch.randn(gc, cin + gc * k, 1, 3, 3, dtype=torch.float64) for k in range(4)] + [torch.randn(cout, cin + 4 * gc, 3, 1, 1, dtype=torch.float64)]
    bs = [torch.randn(gc, dtype=torch.float64) for _ in range(4)] + [torch.randn(cout, dtype=torch.float64)]

    def dense(x, ws, bs):
        feats = [x]
        for k in range(4):
            feats.append(F.leaky_relu(F.conv3d(torch.cat(feats, 1), ws[k], bs[k], padding=(0, 1, 1)), 0.2))
        return F.conv3d(torch.cat(feats, 1), ws[4], bs[4], padding=(1, 0, 0))
    x = torch.randn(1, cin, 3, 6, 5, dtype=torch.float64)
    wv, bv = widen_dense_params(ws, bs, cin, cout, gc, 64, 64)
    assert [tuple(w.shape[:2]) for w in wv] == [(32, 64), (32, 96), (32, 128), (32, 160), (64, 192)]
    xv = torch.zeros(1, 64, 3, 6, 5, dtype=torch.float64)
    xv[:, :cin] = x
    yv = dense(xv, wv, bv)
    y = dense(x, ws, bs)
    assert (yv[:, :cout] - y).abs().max() < 1e-13 * y.abs().max() and yv[:, cout:].abs().max() == 0.0


def test_weight_epoch_invalidates_every_packed_cache_key():
    """runtime.invalidate_weights: params_key / weights_stamp change although no tensor version did - what a
    replayed optimiser step or a `.data` write needs."""
    from selfc_amd import runtime as rt
    m = torch.nn.Conv2d(3, 4, 3)
    k0, s0 = rt.params_key(m), rt.weights_stamp(rt.plist(m))
    assert rt.params_key(m) == k0
    m.weight.data.mul_(2.0)                      # bypasses the version counter
    assert rt.params_key(m) == k0
    rt.invalidate_weights()
    assert rt.params_key(m) != k0 and rt.weights_stamp(rt.plist(m)) != s0
    with torch.no_grad():
        m.weight.mul_(2.0)                       # a tracked in-place update changes the key by itself
    k1 = rt.params_key(m)
    with torch.no_grad():
        m.weight.add_(1.0)
    assert rt.params_key(m) != k1


def test_grad_sink_views_and_zero():
    """autograd.GradSink (host logic only): one flat buffer, a 256-byte aligned view per parameter as its `.grad`, zero()
    is one fill and re-attaches views somebody replaced, view_of() only vouches for views that are still attached."""
    from selfc_amd import autograd as ag
    ps = [torch.nn.Parameter(torch.randn(3, 5)), torch.nn.Parameter(torch.randn(7)), torch.nn.Parameter(torch.randn(2, 2), requires_grad=False)]
    sink = ag.GradSink(ps)
    assert len(sink.params) == 2 and sink.flat.numel() == 128              # 15 -> 64, 7 -> 64 floats
    assert ps[0].grad.shape == (3, 5) and ps[1].grad.shape == (7,) and ps[2].grad is None
    assert ps[0].grad.data_ptr() == sink.flat.data_ptr() and ps[1].grad.data_ptr() == sink.flat.data_ptr() + 64 * 4
    ps[0].grad.add_(1.0)
    assert float(sink.flat.sum()) == 15.0                                   # the pads stay zero
    assert sink.view_of(ps[0]) is ps[0].grad and sink.view_of(ps[2]) is None
    ps[1].grad = None                                                       # e.g. optimizer.zero_grad(set_to_none=True)
    assert sink.view_of(ps[1]) is None
    sink.zero()
    assert float(sink.flat.abs().sum()) == 0.0 and sink.view_of(ps[1]) is ps[1].grad
    with ag.grad_sink(sink) as s_:
        assert ag._SINK is sink and s_ is sink
    assert ag._SINK is None


def test_pack_group_equals_per_module_packing():
    """runtime.PackGroup (one gather for a whole net, used by the trainer every step) must install, module by module, exactly the
    tensors the modules' own caches would build - blocks, the STP chain's subnets, GlobalAgg gather parts (+ the transposed proj1
    of its backward), the head and its transposed convs - under the keys those caches check (no repack afterwards)."""
    import copy
    import torch
    from selfc_amd import GlobalVar, runtime as rt
    from selfc_amd.modules.SelfC_GMM_arch_inv import SelfCInvNet
    from selfc_amd.modules.Subnet_constructor import D2DTInput
    from selfc_amd.packing import pack_pointwise_T
    GlobalVar.set_Temporal_LEN(7)
    torch.manual_seed(5)
    opt = {"global_module": "nonlocal", "stp_blk_num": 4, "fh_loss": "gmm", "scale": 4, "gmm_k": 5}
    net = SelfCInvNet(opt, 3, 3, "D2DTNet", [2, 2], 2)
    with torch.no_grad():
        for p in torch.nn.Module.parameters(net):
            p.add_(torch.randn_like(p) * 0.05)                 # INN_init zeroes conv5: make every tensor distinctive
    stp = net.stp_net
    grp = rt.PackGroup()
    for blk in net._blocks():
        rt.group_add_block(grp, blk)
    stp.add_to_pack_group(grp)
    for step in range(2):                                       # second round: same plan, new values
        with torch.no_grad():
            for p in torch.nn.Module.parameters(net):
                p.mul_(0.9)
        grp.refresh()
        ref = copy.deepcopy(net)                                # fresh addresses: every cache of the copy misses and packs eagerly
        for blk, rblk in zip(net._blocks(), ref._blocks()):
            pb = blk._pb
            assert rt.packed_block(blk) is pb                   # the group's install IS the cache entry
            want = rt.packed_block(rblk)
            for sub in "FGH":
                a, b = getattr(pb, sub), getattr(want, sub)
                for x, y in zip(a.w3 + a.b3 + [a.w5, a.b5, a.wfused, a.w5p, a.wt5, a.wtx] + a.wtd,
                                b.w3 + b.b3 + [b.w5, b.b5, b.wfused, b.w5p, b.wt5, b.wtx] + b.wtd):
                    assert (x is None) == (y is None)
                    assert x is None or (x.dtype == y.dtype and torch.equal(x, y))
        for m, rm in zip(stp._chain(), ref.stp_net._chain()):
            if isinstance(m, D2DTInput):
                pk = m._pk
                assert rt.packed_subnet(m, stp._virt(m)) is pk
                want = rt.packed_subnet(rm, stp._virt(m))
                for x, y in zip(pk.w3 + pk.b3 + [pk.w5, pk.b5], want.w3 + want.b3 + [want.w5, want.b5]):
                    assert torch.equal(x, y)
            else:
                got, want = m._packed(36, 36), rm._packed(36, 36)
                assert set(got) == set(want)
                for k_ in got:
                    assert got[k_].dtype == want[k_].dtype and torch.equal(got[k_], want[k_]), k_
                assert m._w1t_key == rt.params_key(m)
                assert torch.equal(m._w1t, rm._gather_entries(rm._gather_params())["w1t"][0])
        tail, rtail = stp._tail_packed(), ref.stp_net._tail_packed()
        for (w_, b_, ci, co), (rw, rb, rci, rco) in zip(tail, rtail):
            assert (ci, co) == (rci, rco) and torch.equal(w_, rw) and torch.equal(b_, rb)
        for conv in stp._tail_convs():
            assert conv._wt_key == rt.params_key(conv) and torch.equal(conv._wt_pk, pack_pointwise_T(conv.weight))


def test_pool_weight_map_batch_and_adjoint_batch():
    """The batched pooling-map fold and its adjoint (one product for all GlobalAgg blocks of a chain) equal the per-module
    functions, and the two are adjoint to each other: <fold(W), D> == <W, fold^T(D)>."""
    import torch
    from selfc_amd import packing as P
    g = torch.Generator().manual_seed(9)
    for h, w in ((36, 36), (64, 112), (9, 20)):
        ws = [torch.randn(1, 1024, generator=g) for _ in range(4)]
        maps = P.pool_weight_map_batch(ws, h, w)
        assert maps.shape == (4, h * w)
        for i, fw in enumerate(ws):
            assert torch.equal(maps[i], P.pool_weight_map(fw, h, w))
        d = torch.randn(4, h * w, generator=g)
        folded = P.pool_weight_map_grad_batch(d, h, w)
        for i in range(4):
            assert torch.equal(folded[i].reshape(1, 1024), P.pool_weight_map_grad(d[i], h, w))
        lhs = (maps.double() * d.double()).sum()
        rhs = (torch.stack([fw.reshape(-1) for fw in ws]).double() * folded.double()).sum()
        assert abs(float(lhs - rhs)) < 1e-4 * max(1.0, abs(float(lhs)))


def test_gagg_row_perm_keeps_a_lanes_channels():
    """proj1's output-row order for gagg_mix_kernel: a permutation of 0..63 in which rows 4 kq .. 4 kq + 3 of output tile o are the
    four consecutive channels 32 (o >> 1) + 8 kq + 4 (o & 1) + i - channels the lane with k octet kq reads as its B operand
    (k = 32 ks + 8 kq + j), so that the residual is in its registers."""
    from selfc_amd.packing import gagg_row_perm
    perm = gagg_row_perm().tolist()
    assert sorted(perm) == list(range(64))
    for o in range(4):
        for kq in range(4):
            rows = perm[16 * o + 4 * kq: 16 * o + 4 * kq + 4]
            assert rows == list(range(rows[0], rows[0] + 4)) and rows[0] % 4 == 0
            held = {32 * ks + 8 * kq + j for ks in (0, 1) for j in range(8)}        # the lane's input channels
            assert set(rows) <= held


def _unpack_a16(frags):
    """16x16x32 A fragments [n, 64, 8] -> (n, 16 rows, 32 k): lane l = (q = l >> 4, i = l & 15) holds row i, k = 8 q + j"""
    return frags.float().reshape(-1, 4, 16, 8).permute(0, 2, 1, 3).reshape(-1, 16, 32)


@pytest.mark.parametrize("pair", [0, 1])
def test_pack_fused_f16_stream(pair):
    """fragment stream of csrc/fused_f16.hip (v_mfma_f32_16x16x32): per pair [merged k32 steps: a rb0, a rb1, b rb0, b rb1]
    [9 FM steps: b rb0, b rb1]; row r of block rb = output channel 8 (r >> 2) + 4 rb + (r & 3); the k-octets of a step follow
    packing.f16_steps - the kernel's step_off / step_kind decode is restated here with BYTE offsets on an image whose pixel
    holds [x2 48 | f1 32 | f2 32] f16 channels (96 / 224 bytes, no padding)."""
    g = torch.Generator().manual_seed(170 + pair)
    ws = [torch.randn(32, 48 + 32 * i, 1, 3, 3, generator=g) * 0.1 for i in range(4)]
    stream = P.pack_fused_f16(ws, 48)
    assert stream.shape == (220, 64, 8)
    nin = 48 + 64 * pair
    nm = 14 + 18 * pair
    base = 0 if pair == 0 else 74
    merged = _unpack_a16(stream[base: base + 4 * nm]).reshape(nm, 4, 16, 32)        # [step][a0 a1 b0 b1][row][k]
    fmw = _unpack_a16(stream[base + 4 * nm: base + 4 * nm + 18]).reshape(9, 2, 16, 32)
    n, h, w = 2, 5, 6
    img = torch.randn(n, nin, h, w, generator=g)
    fm = torch.randn(n, 32, h, w, generator=g)
    ip = F.pad(img.permute(0, 2, 3, 1), (0, 0, 1, 1, 1, 1))            # [n][h+2][w+2][nin]: channel c at byte 2 c of the pixel
    pitch = 2 * nin

    def octet(y0, x0, byte):          # 8 channels starting at `byte` bytes from pixel (y0, x0)'s first byte, for every output pixel
        dx, c = divmod(byte // 2, nin) if byte >= 0 else (0, 0)
        return ip[:, y0:y0 + h, x0 + dx:x0 + dx + w, c:c + 8]

    def step_bytes(s, q):             # the kernel: lane pointer of kind(s) (relative to the top-left tap pixel) + step_off(s); rows are dy
        if s < 9:
            return s // 3, (s % 3) * pitch + 16 * q
        if s < 12:
            return s - 9, (q >> 1) * pitch + 64 + 16 * (q & 1)
        if s == 12:
            return (q >> 1), 2 * pitch + 64 + 16 * (q & 1)
        if s == 13:
            return 2, 2 * pitch + 64 + 16 * (q & 1)
        t, f = (s - 14) % 9, (s - 14) // 9
        return t // 3, (t % 3) * pitch + 96 + 64 * f + 16 * q

    acc_a = torch.zeros(n, 32, h, w)
    acc_b = torch.zeros(n, 32, h, w)
    chan = torch.tensor([[8 * (r >> 2) + 4 * rb + (r & 3) for r in range(16)] for rb in range(2)])
    for s in range(nm):
        k = torch.cat([octet(*((lambda dy, b: (dy, 0, b))(*step_bytes(s, q)))) for q in range(4)], -1)      # [n][h][w][32 k]
        for rb in range(2):
            acc_a[:, chan[rb]] += torch.einsum("rk,nhwk->nrhw", merged[s, rb], k)
            acc_b[:, chan[rb]] += torch.einsum("rk,nhwk->nrhw", merged[s, 2 + rb], k)
    fp = F.pad(fm.permute(0, 2, 3, 1), (0, 0, 1, 1, 1, 1))
    for t in range(9):
        k = fp[:, t // 3:t // 3 + h, t % 3:t % 3 + w, :]
        for rb in range(2):
            acc_b[:, chan[rb]] += torch.einsum("rk,nhwk->nrhw", fmw[t, rb], k)
    ref_a = F.conv2d(img, ws[2 * pair][:, :, 0].half().float(), None, 1, 1)
    ref_b = F.conv2d(torch.cat((img, fm), 1), ws[2 * pair + 1][:, :, 0].half().float(), None, 1, 1)
    assert torch.allclose(acc_a, ref_a, atol=1e-4, rtol=1e-4)
    assert torch.allclose(acc_b, ref_b, atol=1e-4, rtol=1e-4)


def test_pack_f5_partial16():
    """conv5 partial-product fragments of the 16x16x32 kernels: row 4 tap + oc, K = 32 channels of one feature (x2[32:48] in the
    lower half of a fragment whose upper half is zero): summed over the six fragments they are F's temporal conv5 per tap."""
    g = torch.Generator().manual_seed(171)
    w5 = torch.randn(3, 176, 3, 1, 1, generator=g) * 0.1
    fr = _unpack_a16(P.pack_f5_partial16(w5, 48))                  # (6, 16, 32)
    assert fr.shape == (6, 16, 32)
    d = torch.randn(176, generator=g)
    parts = [d[0:32], torch.cat((d[32:48], torch.zeros(16))), d[48:80], d[80:112], d[112:144], d[144:176]]
    p = sum(fr[j] @ parts[j] for j in range(6))                    # 16 rows
    for tap in range(3):
        ref = w5[:, :, tap, 0, 0].half().float() @ d
        assert torch.allclose(p[4 * tap:4 * tap + 3], ref, atol=1e-4, rtol=1e-4)
        assert float(p[4 * tap + 3].abs()) == 0.0
    assert float(p[12:].abs().max()) == 0.0
