"""Round-2 parity cases on the MI355X, through the C ABI: the codec variant's shapes, InvRescaleNet's
reverse call with the sampled HF tensor pinned, and the GMM head of STP v1.  Fixtures: tools/make_golden.py r2 (vectors
from the reference's own modules).  Tolerance: 1e-3 relative (BASELINE north star), max-norm and relative L2."""
import pytest
import torch

from conftest import load_golden, rel_err, rel_l2, subdict
from oracle import selfc_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-3
T = 7
CODEC_OPT = {"global_module": "nonlocal", "stp_blk_num": 4, "fh_loss": "l2", "scale": 2, "gmm_k": 5,
             "stp_hidden_c": 24, "stp_denseblock_innerc": 12}


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from selfc_amd import GlobalVar, _lib
    _lib.lib()
    GlobalVar.set_Temporal_LEN(T)
    return torch.device("cuda:0")


def _codec_net(dev, g):
    from selfc_amd.modules.SelfC_Codec_arch_inv import SelfCInvNet
    net = SelfCInvNet(CODEC_OPT, 3, 3, "D2DTNet", [4], 1)
    net.load_state_dict({k: v for k, v in g.items() if k.startswith(("operations.", "stp_net."))}, strict=True)
    return net.to(dev).eval()


def test_irn_reverse_with_pinned_hf(dev, monkeypatch):
    """InvRescaleNet.forward(rev=True) (Inv_arch.py:115-123): the call itself, with torch.rand pinned to the fixture's draw"""
    from selfc_amd.modules.Inv_arch import InvRescaleNet
    from selfc_amd.modules.Subnet_constructor import subnet
    g = load_golden("g14_irn_rev")
    net = InvRescaleNet(3, 3, subnet("DBNet", "xavier"), [1], 1)
    net.load_state_dict({k: v for k, v in g.items() if k.startswith("operations.")}, strict=True)
    net.to(dev).eval()
    hf = g["hf45"].to(dev)
    monkeypatch.setattr(torch, "rand", lambda *a, **k: hf.clone())
    with torch.no_grad():
        out, none = net(g["lr"].to(dev), rev=True)
    assert none is None and out.shape == g["x_rev"].shape
    assert rel_err(out.cpu(), g["x_rev"]) < TOL and rel_l2(out.cpu(), g["x_rev"]) < TOL


def test_codec_invblock_15_3(dev):
    """InvBlockExp(15 | 3) on D2DTInput(12->3) / (3->12), clips of 3 frames (SelfC_Codec_arch_inv.py:24-57,386-391)"""
    from selfc_amd import GlobalVar
    g = load_golden("g15_codec")
    net = _codec_net(dev, g)
    blk = net.operations[1]
    try:
        GlobalVar.set_Temporal_LEN(3)
        with torch.no_grad():
            y = blk(g["blk_x"].to(dev))
            assert rel_err(y.cpu(), g["blk_y"]) < TOL and rel_l2(y.cpu(), g["blk_y"]) < TOL
            assert rel_err(blk.s.cpu(), g["blk_s"]) < TOL
            xr = blk(g["blk_x"].to(dev), rev=True)
            assert rel_err(xr.cpu(), g["blk_xrev"]) < TOL
    finally:
        GlobalVar.set_Temporal_LEN(T)


def test_codec_globalagg_24_and_narrow_stp(dev):
    """GlobalAgg(24) over TEMP_LEN = 3 clips and the narrow STP (hidden 24, growth 12, l2 head of 12 channels):
    :103-131, 234-312"""
    g = load_golden("g15_codec")
    net = _codec_net(dev, g)
    stp = net.stp_net
    with torch.no_grad():
        y = stp.global_m1(g["ga_x"].to(dev))
        assert rel_err(y.cpu(), g["ga_y"]) < TOL and rel_l2(y.cpu(), g["ga_y"]) < TOL
        lr = g["stp_lr"].to(dev)
        stp(lr.reshape(2, 3, 3, 8, 12).transpose(1, 2))
        raw = stp.parameters.transpose(1, 2).reshape(6, -1, 8, 12)          # the reference's attribute name
        assert raw.shape == g["stp_raw"].shape
        assert rel_err(raw.cpu(), g["stp_raw"]) < TOL and rel_l2(raw.cpu(), g["stp_raw"]) < TOL
        # a gc = 12 dense block through the module API as well
        d = stp.local_m2
        xin = torch.randn(6, 24, 8, 12, generator=torch.Generator().manual_seed(5))
        from selfc_amd import GlobalVar
        try:
            GlobalVar.set_Temporal_LEN(3)
            yd = d(xin.to(dev))
        finally:
            GlobalVar.set_Temporal_LEN(T)
        assert rel_err(yd.cpu(), O.d2dt(subdict(g, "stp_net.local_m2"), xin, 3)) < TOL


def test_codec_forward_test_tiling(dev):
    """forward_test (:502-640) without the H.265 stream: 3-frame segments (5 frames -> 2 segments, the pad repeats the
    second-to-last frame), two column strips down, 2 x 2 tiles up; GlobalVar is restored afterwards."""
    from selfc_amd import GlobalVar
    g = load_golden("g15_codec")
    net = _codec_net(dev, g)
    GlobalVar.set_Istrain(False)
    try:
        GlobalVar.set_Temporal_LEN(5)
        with torch.no_grad():
            out = net(x=g["x"].to(dev), rev=False)
            lr = out[0]
            assert len(out) == 7 and lr.shape == g["enc_lr"].shape
            # the encoder's LR is quantised on its way out (where the reference hands it to the 8-bit video writer)
            assert (lr.cpu() - g["lr_q"]).abs().max() <= 1.0 / 255 + 1e-6
            assert ((lr.cpu() - g["lr_q"]).abs() > 1e-6).float().mean() < 0.01     # only values within 1e-3 of a rounding boundary may flip
            hr = net(x=g["lr_q"].to(dev), rev=True)
        assert GlobalVar.get_Temporal_LEN() == 5
    finally:
        GlobalVar.set_Temporal_LEN(T)
    assert rel_err(hr.cpu(), g["dec_hr"]) < TOL and rel_l2(hr.cpu(), g["dec_hr"]) < TOL
    # un-tiled halves against the oracle (encode / decode of one segment)
    x3 = g["x"][:3]
    try:
        GlobalVar.set_Temporal_LEN(3)
        with torch.no_grad():
            z = net.encode(x3.to(dev))
        assert rel_err(z.cpu(), O.large_fwd(g, x3, 3, k=2)) < TOL
    finally:
        GlobalVar.set_Temporal_LEN(T)


def test_stp_v1_gmm_head(dev):
    """STP v1 with fh_loss gmm (SelfC_arch_inv.py:118-128,151-163): raw head output and the sample with injected eps"""
    from selfc_amd.modules.SelfC_arch_inv import STPNet
    g = load_golden("g16_stp_v1_gmm")
    opt = {"stp_d2d_inner_c": 32, "stp_temporal_c": 32, "fh_loss": "gmm", "gmm_mixture_num": 5, "stp_blk_num": 2, "condition_func": "D2DTNet"}
    stp = STPNet(opt)
    stp.load_state_dict({k: v for k, v in g.items() if k.split(".")[0] in ("blk1", "blk2", "tail_gmm")}, strict=True)
    stp.to(dev).eval()
    stp.eps = g["eps"].unsqueeze(1).to(dev)                  # (K, b=1, 9, T, h, w)
    with torch.no_grad():
        stp(g["lr"].to(dev).reshape(1, T, 3, 8, 12).transpose(1, 2))
    raw = stp.parameters[0].transpose(0, 1)
    assert rel_err(raw.cpu(), g["raw"]) < TOL and rel_l2(raw.cpu(), g["raw"]) < TOL
    v = stp.sample()[0].transpose(0, 1)
    assert rel_err(v.cpu(), g["v"]) < TOL
    assert torch.isfinite(stp.neg_llh(stp.sample())).all()


def test_stp_v2_gmm_thin_head(dev):
    """fh_loss 'gmm_thin' (SelfC_GMM_arch_inv.py:345-354): ReLU between the head's pointwise layers, then the same sampler"""
    from selfc_amd.modules.SelfC_GMM_arch_inv import STPNet
    g7, g = load_golden("g7_stp_gmm"), load_golden("g17_stp_gmm_thin")
    stp = STPNet({"global_module": "nonlocal", "stp_blk_num": 6, "fh_loss": "gmm_thin", "scale": 4, "gmm_k": 5})
    sd = {k: v for k, v in g7.items() if k.split(".")[0] in ("local_m1", "local_m2", "global_m1", "global_m2", "other_stp_modules")}
    sd.update({k: v for k, v in g.items() if k.startswith("tail_gmm.")})
    stp.load_state_dict(sd, strict=True)
    stp.to(dev).eval()
    stp.eps = g["eps"].permute(1, 2, 0, 3, 4).unsqueeze(0).to(dev)        # (1,48,5,T,h,w)
    with torch.no_grad():
        stp(g["lr"].to(dev).reshape(1, T, 3, 8, 12).transpose(1, 2))
    raw = stp.parameters[0].transpose(0, 1)
    assert rel_err(raw.cpu(), g["raw"]) < TOL and rel_l2(raw.cpu(), g["raw"]) < TOL
    assert rel_err(stp.sample()[0].transpose(0, 1).cpu(), g["v"]) < TOL


def test_feature_calapse_block_3d_io(dev):
    """FeatureCalapseBlock.forward(x, io_type='3d') (Subnet_constructor.py:304-320): x is (b, c, t, h, w) and the clip length is
    x's own third axis, not GlobalVar's; the reference's SpaceToDepth unpacks four sizes, so the call only exists for scale == 1
    (scale > 1 raises the same ValueError there)."""
    from selfc_amd.modules.Subnet_constructor import FeatureCalapseBlock
    torch.manual_seed(3)
    blk = FeatureCalapseBlock(32, 32, scale=1, INN_init=False, is_res=True)
    sd = {k: v.detach().clone() for k, v in blk.state_dict().items()}
    x5 = torch.randn(2, 32, 5, 8, 12) * 0.5                        # 2 clips of FIVE frames (GlobalVar says 7)
    frames = x5.transpose(1, 2).reshape(10, 32, 8, 12)
    ref = O.feature_calapse_block(sd, frames, 5, scale=1, is_res=True).reshape(2, 5, 32, 8, 12).transpose(1, 2)
    blk.to(dev)
    with torch.no_grad():
        y = blk(x5.to(dev), io_type="3d")
    assert y.shape == ref.shape and rel_err(y.cpu(), ref) < TOL and rel_l2(y.cpu(), ref) < TOL
    with pytest.raises(ValueError):
        FeatureCalapseBlock(3, 12).to(dev)(torch.zeros(1, 3, 7, 16, 16, device=dev), io_type="3d")


@pytest.mark.parametrize("which", ["v2_scale2", "codec_gmm"])
def test_gmm_head_other_widths(dev, which):
    """The GMM head outside SelfC-large's shape: scale 2 (hf_dim = 12, head 64 -> 128 -> 256 -> 180: SelfC_GMM_arch_inv.py:331-344
    with scale 2) and the codec variant's narrow head (24 -> 48 -> 96 -> 180, SelfC_Codec_arch_inv.py:263-272) - widths are padded
    to the pointwise kernel's granules with zero rows / columns, the sample comes from the generic sampler."""
    from selfc_amd import GlobalVar
    torch.manual_seed(23)
    if which == "v2_scale2":
        from selfc_amd.modules.SelfC_GMM_arch_inv import STPNet
        stp, tlen, prefix = STPNet({"global_module": "nonlocal", "stp_blk_num": 4, "fh_loss": "gmm", "scale": 2, "gmm_k": 5}), T, "tail_gmm"
        fn = lambda p, x: O.stp_v2_parameters(p, x, tlen, stp_blk_num=4)                      # noqa: E731
    else:
        from selfc_amd.modules.SelfC_Codec_arch_inv import STPNet
        stp, tlen, prefix = STPNet(dict(CODEC_OPT, fh_loss="gmm")), 3, "tail"
        fn = lambda p, x: O.codec_stp_parameters(p, x, 3, 4)                                  # noqa: E731
    sd = {k: v.detach().clone() for k, v in stp.state_dict().items()}
    b, h, w = 2, 8, 12
    lr = torch.rand(b * tlen, 3, h, w)
    eps = torch.randn(b * tlen, 12, 5, h, w)
    raw_ref = fn(sd, lr)
    assert raw_ref.shape[1] == 180
    v_ref = O.stp_v2_gmm_sample(raw_ref, eps, hf_dim=12, k=5)
    stp.to(dev).eval()
    stp.eps = eps.reshape(b, tlen, 12, 5, h, w).permute(0, 2, 3, 1, 4, 5).to(dev)             # (b, hf, K, t, h, w)
    try:
        GlobalVar.set_Temporal_LEN(tlen)
        with torch.no_grad():
            stp(lr.to(dev).reshape(b, tlen, 3, h, w).transpose(1, 2))
    finally:
        GlobalVar.set_Temporal_LEN(T)
    raw = stp.parameters.transpose(1, 2).reshape(b * tlen, 180, h, w)
    v = stp.sample().transpose(1, 2).reshape(b * tlen, 12, h, w)
    assert rel_err(raw.cpu(), raw_ref) < TOL and rel_l2(raw.cpu(), raw_ref) < TOL
    assert rel_err(v.cpu(), v_ref) < TOL
