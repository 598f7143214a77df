"""Parity of the HIP path (through the C ABI) against the golden vectors captured
from the reference and against the CPU oracle.  Tolerances: bit-exact for the
index shuffles; "1e-3 relative fp32" (BASELINE.json) for float paths, measured as
max|a-b| / max|b| (conftest.rel_err).  Run with `-m gpu` on the MI355X box."""
import os
import sys

import pytest
import torch

from conftest import ROOT, group_err, load_golden, rel_err, rel_l2, seeded_fill, subdict
from oracle import selfc_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-3
T = 7


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from selfc_amd import _lib
    _lib.lib()           # fails loudly if libselfc_hip.so is missing
    from selfc_amd import GlobalVar
    GlobalVar.set_Temporal_LEN(T)
    return torch.device("cuda:0")


def test_library_loaded():
    from selfc_amd import _lib
    assert b"gfx950" in _lib.lib().selfc_version()


def test_haar_bit_exact(dev):
    from selfc_amd.modules.Inv_arch import HaarDownsampling
    g = load_golden("g1_haar")
    h1, h2 = HaarDownsampling(3).to(dev), HaarDownsampling(12).to(dev)
    assert torch.equal(h1.haar_weights.cpu(), g["haar_weights"])
    y = h1(g["x"].to(dev))
    assert torch.equal(y.cpu(), g["y"])
    assert abs(h1.jacobian(None) - float(g["jac_fwd"])) < 1e-9
    assert torch.equal(h2(y).cpu(), g["y2"])
    assert torch.equal(h1(g["y"].to(dev), rev=True).cpu(), g["xr"])
    assert abs(h1.jacobian(None, rev=True) - float(g["jac_rev"])) < 1e-9
    assert torch.equal(h1(g["zrand"].to(dev), rev=True).cpu(), g["zrand_inv"])


def test_freq_bit_exact(dev):
    from selfc_amd.modules.SelfC_GMM_arch_inv import FrequencyAnalyzer
    g = load_golden("g2_freq")
    fa = FrequencyAnalyzer(3)
    assert torch.equal(fa(g["x"].to(dev)).cpu(), g["y"])
    assert torch.equal(fa(g["z"].to(dev), rev=True).cpu(), g["z_rev"])
    assert torch.equal(fa(g["y"].to(dev), rev=True).cpu(), g["y_rev"])
    # integer-valued input: every sum is exact, so this pins the pure index shuffle
    xi = torch.randint(0, 64, (2, 3, 8, 12)).float() * 16
    assert torch.equal(fa(xi.to(dev)).cpu(), O.freq_fwd(xi))


def test_quant_bit_exact(dev):
    from selfc_amd.modules.Quantization import Quantization
    g = load_golden("g9_quant")
    assert torch.equal(Quantization()(g["x"].to(dev)).cpu(), g["y"])


def test_quant_other_class_settings_bit_exact(dev):
    """Quantization.quant_v / is_clip are class-level settings of the reference (Quantization.py:9-13,20-22): any grid, with or
    without the clamp, bit-exact against the oracle (incl. half-way values, negatives, > 1 and the 15-level grid)."""
    from selfc_amd.modules.Quantization import Quantization
    g = load_golden("g9_quant")
    x = torch.cat((g["x"].reshape(-1), torch.tensor([-0.7, 1.3, 0.5 / 15, 1.5 / 15, 2.5 / 15, 0.1234567]), torch.linspace(-0.5, 1.5, 37)))
    try:
        for qv, clip in ((15.0, True), (255.0, False), (1023.0, True), (7.0, False)):
            q = Quantization(qv, clip)
            assert torch.equal(q(x.to(dev)).cpu(), O.quantize(x, qv, clip)), (qv, clip)
    finally:
        Quantization(255.0, True)               # restore the shipped class-level setting


def test_denseblock(dev):
    from selfc_amd.modules.Subnet_constructor import DenseBlock
    g = load_golden("g3_denseblock")
    for tag, (ci, co) in {"f": (9, 3), "g": (3, 9)}.items():
        m = DenseBlock(ci, co, "xavier")
        m.load_state_dict(subdict(g, tag), strict=True)
        with torch.no_grad():
            y = m.to(dev)(g[f"{tag}_x"].to(dev))
        assert rel_err(y.cpu(), g[f"{tag}_y"]) < TOL


def test_d2dt(dev):
    from selfc_amd.modules.Subnet_constructor import D2DTInput
    g = load_golden("g4_d2dt")
    for tag, (ci, co) in {"f": (48, 3), "g": (3, 48)}.items():
        m = D2DTInput(ci, co, "xavier")
        m.load_state_dict(subdict(g, tag), strict=True)
        with torch.no_grad():
            y = m.to(dev)(g[f"{tag}_x"].to(dev))
        assert rel_err(y.cpu(), g[f"{tag}_y"]) < TOL
        with torch.no_grad():   # io_type='3d' path
            x3 = g[f"{tag}_x"].reshape(2, T, ci, 12, 20).transpose(1, 2).to(dev)
            y3 = m(x3, io_type="3d").transpose(1, 2).reshape(2 * T, co, 12, 20)
        assert rel_err(y3.cpu(), g[f"{tag}_y"]) < TOL


def _invblock(dev, name, kind, cnum):
    from selfc_amd.modules.Inv_arch import InvBlockExp
    from selfc_amd.modules.Subnet_constructor import subnet
    g = load_golden(name)
    blk = InvBlockExp(subnet(kind, "xavier"), cnum, 3)
    blk.load_state_dict({k: v for k, v in g.items() if k[:2] in ("F.", "G.", "H.")}, strict=True)
    blk.to(dev)
    x = g["x"].to(dev)
    with torch.no_grad():
        y = blk(x)
        assert rel_err(y.cpu(), g["y_fwd"]) < TOL
        assert rel_err(blk.s.cpu(), g["s_fwd"]) < TOL
        assert abs(blk.jacobian(x).item() - g["jac_fwd"].item()) < 2e-3 * abs(g["jac_fwd"].item()) + 1e-2
        xr = blk(y, rev=True)                       # invertibility through the kernels
        assert rel_err(xr.cpu(), g["x"]) < TOL
        yr = blk(x, rev=True)
        assert rel_err(yr.cpu(), g["y_rev"]) < TOL
        assert rel_err(blk.s.cpu(), g["s_rev"]) < TOL
        assert abs(blk.jacobian(x, rev=True).item() - g["jac_rev"].item()) < 2e-3 * abs(g["jac_rev"].item()) + 1e-2


def test_invblock_dbnet(dev):
    _invblock(dev, "g5_invblock_dbnet", "DBNet", 12)


def test_invblock_d2dt(dev):
    _invblock(dev, "g5_invblock_d2dt", "D2DTNet", 51)


OPT = {"global_module": "nonlocal", "stp_blk_num": 6, "fh_loss": "l2", "scale": 4, "gmm_k": 5}


def _large_net(dev, g, fh_loss="l2", stp=None):
    from selfc_amd.modules.SelfC_GMM_arch_inv import SelfCInvNet
    net = SelfCInvNet(dict(OPT, fh_loss=fh_loss), 3, 3, "D2DTNet", [4, 4], 2)
    sd = {k: v for k, v in g.items() if k.startswith("operations.")}
    if stp is not None:
        sd.update({k: v for k, v in stp.items() if k.startswith("stp_net.")})
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith("stp_net.") for k in missing)
    return net.to(dev).eval()


def test_large_stack_golden(dev):
    g = load_golden("g8_large_stack")
    net = _large_net(dev, g)
    with torch.no_grad():
        z, loss_c = net(x=g["x"].to(dev), rev=False)
        assert float(loss_c) == 0.0
        assert rel_err(z.cpu(), g["z"]) < TOL
        assert rel_err(z[:, :3].cpu(), g["z"][:, :3]) < TOL          # the LR video itself
        # LR (0:3) and HF (3:51) channel groups each against their OWN magnitude, max-norm and relative L2
        assert group_err(z.cpu(), g["z"]) < TOL
        xr = net.inverse_from_latent(g["z"].to(dev))
        assert rel_err(xr.cpu(), g["x_rev"]) < TOL
        assert rel_l2(xr.cpu(), g["x_rev"]) < TOL


def test_large_full_reverse_with_stp_l2(dev):
    g = load_golden("g8_large_stack")
    s = load_golden("g7_stp_l2_full_rev")
    net = _large_net(dev, g, "l2", s)
    with torch.no_grad():
        xr, hf = net(x=s["lr"].to(dev), rev=True)
    assert rel_err(hf.cpu(), s["hf"]) < TOL
    assert rel_l2(hf.cpu(), s["hf"]) < TOL
    assert rel_err(xr.cpu(), s["x_rev"]) < TOL
    assert rel_l2(xr.cpu(), s["x_rev"]) < TOL


def test_stp_gmm_injected_eps(dev):
    from selfc_amd.modules.SelfC_GMM_arch_inv import STPNet
    g = load_golden("g7_stp_gmm")
    stp = STPNet(dict(OPT, fh_loss="gmm"))
    stp.load_state_dict({k: v for k, v in g.items() if k.split(".")[0] in
                         ("local_m1", "local_m2", "global_m1", "global_m2", "other_stp_modules", "tail_gmm")}, strict=True)
    stp.to(dev).eval()
    eps = g["eps"].permute(1, 2, 0, 3, 4).unsqueeze(0).to(dev)        # (1,48,5,T,h,w)
    stp.eps = eps
    with torch.no_grad():
        stp(g["lr"].to(dev).reshape(1, T, 3, 8, 12).transpose(1, 2))
    raw = stp.stp_parameters[0].transpose(0, 1)
    assert rel_err(raw.cpu(), g["raw"]) < TOL
    v = stp.sample()[0].transpose(0, 1)
    assert rel_err(v.cpu(), g["v"]) < TOL


def test_stp_gmm_fused_head_and_sampler(dev):
    """sampling path of SelfCModel.test(): the whole GMM head and the GMM sample run as ONE kernel (selfc_stp_head_gmm,
    output channels permuted to [k][pi | log-sigma | mu][c]) - same golden sample as the unfused pair."""
    from selfc_amd.modules.SelfC_GMM_arch_inv import STPNet
    g = load_golden("g7_stp_gmm")
    stp = STPNet(dict(OPT, fh_loss="gmm"))
    stp.load_state_dict({k: v for k, v in g.items() if k.split(".")[0] in
                         ("local_m1", "local_m2", "global_m1", "global_m2", "other_stp_modules", "tail_gmm")}, strict=True)
    stp.to(dev).eval()
    stp.eps = g["eps"].permute(1, 2, 0, 3, 4).unsqueeze(0).to(dev)        # (1,48,5,T,h,w)
    lr = g["lr"].to(dev).reshape(T, 3, 8, 12)
    x1 = torch.zeros(T, 8, 12, 4, device=dev)
    x1[..., :3] = lr.permute(0, 2, 3, 1)
    hf = torch.zeros(T, 8, 12, 48, device=dev)
    with torch.no_grad():
        raw = stp.run_nhwc(x1, hf, T, T, 8, 12)
    assert raw is None and stp._head_fused() is not None
    assert rel_err(hf.permute(0, 3, 1, 2).cpu(), g["v"]) < TOL


def test_stp_two_clips_ragged_against_oracle(dev):
    """Two clips of 7 frames at 20x36 (not a multiple of the kernels' 64 / 128 / 256-pixel workgroups) through the whole STP
    sampling path with injected noise, against the oracle: the per-clip attention inside the mix kernel (clip index, clip-local
    softmax), the f16 hand-over into the next D2DTInput's operand planes with N = 14 and the partial last workgroup of the
    one-kernel GMM head."""
    from selfc_amd.modules.SelfC_GMM_arch_inv import STPNet
    g = load_golden("g7_stp_gmm")
    keys = ("local_m1", "local_m2", "global_m1", "global_m2", "other_stp_modules", "tail_gmm")
    sd = {k: v for k, v in g.items() if k.split(".")[0] in keys}
    stp = STPNet(dict(OPT, fh_loss="gmm"))
    stp.load_state_dict(sd, strict=True)
    stp.to(dev).eval()
    gen = torch.Generator().manual_seed(21)
    n, h, w = 2 * T, 20, 36
    lr = torch.rand(n, 3, h, w, generator=gen)
    eps = torch.randn(n, 48, 5, h, w, generator=gen)
    raw_ref = O.stp_v2_parameters(sd, lr, T)
    v_ref = O.stp_v2_gmm_sample(raw_ref, eps)
    stp.eps = eps.reshape(2, T, 48, 5, h, w).permute(0, 2, 3, 1, 4, 5).to(dev)       # (b, 48, 5, T, h, w)
    x1 = torch.zeros(n, h, w, 4, device=dev)
    x1[..., :3] = lr.to(dev).permute(0, 2, 3, 1)
    hf = torch.zeros(n, h, w, 48, device=dev)
    with torch.no_grad():
        assert stp.run_nhwc(x1, hf, n, T, h, w) is None                    # fused head: nothing but the sample is written
        assert rel_err(hf.permute(0, 3, 1, 2).cpu(), v_ref) < TOL
        assert rel_l2(hf.permute(0, 3, 1, 2).cpu(), v_ref) < TOL
        raw = stp.run_nhwc(x1, torch.zeros_like(hf), n, T, h, w, keep_raw=True)   # layer-wise head, raw output kept
    assert rel_err(raw.reshape(n, h, w, -1).permute(0, 3, 1, 2).cpu(), raw_ref) < TOL


def test_globalagg(dev):
    from selfc_amd.modules.SelfC_GMM_arch_inv import GlobalAgg
    g = load_golden("g6_globalagg")
    ga = GlobalAgg(64)
    ga.load_state_dict({k: v for k, v in g.items() if k.split(".")[0] in ("fc", "proj1", "proj2", "proj3")}, strict=True)
    ga.to(dev)
    for tag in ("a", "b"):          # 16x16: replicating pooling bins; 20x36: overlapping bins
        with torch.no_grad():
            y = ga(g[f"{tag}_x"].to(dev))
        assert rel_err(y.cpu(), g[f"{tag}_y"]) < TOL


def test_haar_net(dev):
    from selfc_amd.modules.Inv_arch import InvRescaleNet
    from selfc_amd.modules.Subnet_constructor import subnet
    g = load_golden("g8_haar_net")
    irn = InvRescaleNet(3, 3, subnet("DBNet", "xavier"), [1], 1)
    irn.load_state_dict({k: v for k, v in g.items() if k.startswith("operations.")}, strict=True)
    irn.to(dev)
    with torch.no_grad():
        lr, hfm = irn(g["x"].to(dev))
        assert rel_err(lr.cpu(), g["lr"]) < TOL
        assert abs(hfm.item() - g["hf_meansq"].item()) < 1e-3 * g["hf_meansq"].item()
        out = g["z"].to(dev)
        for op in reversed(irn.operations):
            out = op(out, rev=True)
        assert rel_err(out.cpu(), g["x_rev"]) < TOL


@pytest.mark.parametrize("b,h,w", [(1, 36, 52), (2, 20, 44), (3, 64, 64), (1, 136, 200), (2, 68, 132)])
def test_large_vs_oracle_ragged_sizes(dev, b, h, w):
    """latent sizes that are not multiples of the 16x16 tile / 128-pixel strip; several clips."""
    g = load_golden("g8_large_stack")
    net = _large_net(dev, g)
    gen = torch.Generator().manual_seed(b * 1000 + h)
    x = torch.rand(b * T, 3, h, w, generator=gen)
    z_ref = O.large_fwd(g, x, T)
    with torch.no_grad():
        z, _ = net(x=x.to(dev), rev=False)
        assert rel_err(z.cpu(), z_ref) < TOL
        assert group_err(z.cpu(), z_ref) < TOL
        xr = net.inverse_from_latent(z_ref.to(dev))
    assert rel_err(xr.cpu(), O.large_inv_from_latent(g, z_ref, T)) < TOL


_CHILD = """
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + '/tests')
from conftest import load_golden
from selfc_amd import GlobalVar
GlobalVar.set_Temporal_LEN(7)
from test_gpu_parity import _large_net
dev = torch.device('cuda:0')
g = load_golden('g8_large_stack')
net = _large_net(dev, g)
x = torch.rand(14, 3, 72, 100, generator=torch.Generator().manual_seed(77))
with torch.no_grad():
    z, _ = net(x=x.to(dev), rev=False)
    xr = net.inverse_from_latent(z)
np.savez(sys.argv[2], z=z.cpu().numpy(), xr=xr.cpu().numpy())
"""


@pytest.mark.parametrize("switch", ["SELFC_NO_FUSE_F=1", "SELFC_NO_F5P=1", "SELFC_NO_FUSE=1", "SELFC_F_MFMA32=1"])
def test_fused_paths_agree_with_alternative_paths(dev, tmp_path, switch):
    """The default kernels (F's conv1-4 as two pairwise-fused 16x16x32 launches with conv5 as partial products, G/H's conv1-4 as
    one depth-4 fused launch) against the alternative paths of the library - layer-wise conv3x3 / temporal-conv5 kernels, the
    32x32x16 pair kernels of F (SELFC_F_MFMA32: also what a dense buffer beyond the 16x16x32 kernels' 2-GiB buffer addressing
    takes) - run in a child process with the developer switch set: same f16 operands,
    different fp32 summation order (which flips some f16 roundings of the features): the two paths are each within
    ~3.5e-4 of the fp32 oracle and must agree with each other inside the parity tolerance.  Ragged size (18 x 25 latent)."""
    import os
    import subprocess
    import sys
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "child.npz")
    env = dict(os.environ, **dict([switch.split("=")]))
    subprocess.run([sys.executable, "-c", _CHILD, root, out], check=True, env=env, timeout=300)
    g = load_golden("g8_large_stack")
    net = _large_net(dev, g)
    x = torch.rand(14, 3, 72, 100, generator=torch.Generator().manual_seed(77))
    with torch.no_grad():
        z, _ = net(x=x.to(dev), rev=False)
        xr = net.inverse_from_latent(z)
    with np.load(out) as c:
        assert rel_err(z.cpu(), torch.from_numpy(c["z"])) < 6e-4
        assert rel_err(xr.cpu(), torch.from_numpy(c["xr"])) < TOL


def test_dense_buffers_beyond_2gib_take_the_flat_kernels(dev):
    """The 16x16x32 kernels of F address the dense buffer through 32-bit buffer-resource offsets (six planes below 2 GiB);
    a call beyond that - 44 frames at the 1080p latent size, 5.7 M pixel-frames - must fall back to the flat-addressed
    32x32x16 kernels inside the library (csrc/fused_f.hip launch_fused_f).  One coupling block (Inv_arch.py:20-31) over the
    whole batch against the same block over its two halves (clips of 2 frames, so the halves are whole clips): different
    kernels, same f16 operands - inside the parity tolerance, and the second half proves no offset wrapped."""
    from selfc_amd import GlobalVar
    from selfc_amd.modules.Inv_arch import InvBlockExp
    from selfc_amd.modules.Subnet_constructor import subnet
    torch.manual_seed(23)
    blk = InvBlockExp(subnet("D2DTNet", "xavier"), 51, 3).to(dev).eval()
    with torch.no_grad():
        for sub in (blk.F, blk.G, blk.H):
            sub.conv5.weight.normal_(0, 0.02)
    n, h, w = 44, 270, 480
    assert n * h * w * 64 * 6 >= 0x7fff0000
    x = torch.randn(n, 51, h, w, device=dev, generator=torch.Generator(device=dev).manual_seed(3)) * 0.5
    try:
        GlobalVar.set_Temporal_LEN(2)
        with torch.no_grad():
            for rev in (False, True):
                whole = blk(x, rev=rev)
                for lo in (0, n // 2):
                    part = blk(x[lo:lo + n // 2].contiguous(), rev=rev)
                    assert rel_err(whole[lo:lo + n // 2], part) < TOL, (rev, lo)
                    del part
                # ... and against the ORACLE (not only against itself): the LAST clip - the far end of every plane - on a crop that
                # holds the frame's bottom-right corner (rows 256..269 are the partial tile row).  One block sees 8 pixels per
                # direction (four 3x3 convs of F, then four of G / H), so the crop's interior beyond 8 pixels of its cut edges is exact.
                r0, c0, m = h - 80, w - 80, 8
                p_or = {k: v.detach().cpu() for k, v in blk.state_dict().items()}
                want = O.invblock("D2DTNet", p_or, x[n - 2:n, :, r0:, c0:].cpu(), 3, 2, rev=rev)[0]
                got = whole[n - 2:n, :, r0:, c0:].cpu()
                assert rel_err(got[:, :, m:, m:], want[:, :, m:, m:]) < TOL, ("oracle crop", rev)
                del whole
    finally:
        GlobalVar.set_Temporal_LEN(T)
    torch.cuda.empty_cache()


def test_temporal_len_one_and_clip_isolation(dev):
    """T=1 (every frame its own clip: temporal taps see only zero padding) and clip isolation."""
    from selfc_amd import GlobalVar
    g = load_golden("g8_large_stack")
    net = _large_net(dev, g)
    x = torch.rand(3, 3, 32, 32, generator=torch.Generator().manual_seed(5))
    try:
        GlobalVar.set_Temporal_LEN(1)
        with torch.no_grad():
            z, _ = net(x=x.to(dev), rev=False)
        assert rel_err(z.cpu(), O.large_fwd(g, x, 1)) < TOL
    finally:
        GlobalVar.set_Temporal_LEN(T)
    xa = torch.rand(2 * T, 3, 32, 32, generator=torch.Generator().manual_seed(6))
    xb = xa.clone()
    xb[T:] = torch.rand(T, 3, 32, 32, generator=torch.Generator().manual_seed(7))
    with torch.no_grad():
        za, _ = net(x=xa.to(dev), rev=False)
        zb, _ = net(x=xb.to(dev), rev=False)
    assert torch.equal(za[:T], zb[:T])          # clip 0 never sees clip 1


def test_full_size_properties(dev):
    """BASELINE config 2 shape (4 x 7x3x256x448): stack invertibility on the device
    (latent -> inverse blocks -> same latent) and one septuplet against the oracle."""
    from selfc_amd import runtime as rt, _lib
    g = load_golden("g8_large_stack")
    net = _large_net(dev, g)
    x = torch.rand(4 * T, 3, 256, 448, generator=torch.Generator().manual_seed(1234))
    xd = x.to(dev)
    with torch.no_grad():
        z, _ = net(x=xd, rev=False)
        ws = net._workspace(xd, 4 * T, 64, 112)
        arr, nblk = net._stack()
        rt.call("selfc_invstack_run", arr, nblk, ws.latent(), 1, _lib.stream_ptr())
        z0 = rt.latent_to_nchw(ws)
        fa = net.operations[0](xd)
    assert rel_err(z0.cpu(), fa.cpu()) < TOL
    z_ref = O.large_fwd(g, x[:T], T)
    assert rel_err(z[:T].cpu(), z_ref) < TOL


def test_psnr_y_parity_on_synthetic_clip(dev):
    """PSNR within 0.02 dB of the reference.  Vid4 and the pretrained weights are not
    available offline, so the check runs test_rescaling.py's pipeline (netG fwd -> Quantization -> netG rev,
    Y-channel PSNR) on a synthetic clip with the golden weights, HIP path vs CPU oracle."""
    from selfc_amd.modules.Quantization import Quantization
    g = load_golden("g8_large_stack")
    s = load_golden("g7_stp_l2_full_rev")
    net = _large_net(dev, g, "l2", s)
    gen = torch.Generator().manual_seed(77)
    low = torch.rand(T, 3, 9, 13, generator=gen)
    x = torch.nn.functional.interpolate(low, size=(64, 96), mode="bicubic", align_corners=False).clamp(0, 1)
    x = (x + 0.02 * torch.randn(x.shape, generator=gen)).clamp(0, 1)
    with torch.no_grad():
        z, _ = net(x=x.to(dev), rev=False)
        lr = Quantization()(z[:, :3])
        xr, _ = net(x=lr, rev=True)
    z_ref = O.large_fwd(g, x, T)
    lr_ref = O.quantize(z_ref[:, :3])
    hf_ref = O.stp_v2_parameters(subdict(s, "stp_net"), lr_ref, T)
    xr_ref = O.large_inv_from_latent(g, torch.cat((lr_ref, hf_ref), 1), T)
    p_hip = O.psnr_per_frame(O.rgb_to_y(xr.cpu()), O.rgb_to_y(x))
    p_ref = O.psnr_per_frame(O.rgb_to_y(xr_ref), O.rgb_to_y(x))
    assert max(abs(a - b) for a, b in zip(p_hip, p_ref)) < 0.02, (p_hip, p_ref)
    assert abs(sum(p_hip) / T - sum(p_ref) / T) < 0.02
    # the LR frames themselves: a pre-quantisation difference d flips a pixel with probability ~255*|d|
    # (never by more than one 1/255 step); with the measured ~1e-4 differences that is a few percent
    dl = (lr.cpu() - lr_ref).abs()
    flips = (dl > 1e-6).float().mean().item()
    assert dl.max() <= 1.0 / 255 + 1e-6, dl.max()
    assert flips < 0.06, flips


def test_harness_test_loop_and_device_psnr(dev):
    """selfc_amd.harness: SelfCModel.test()'s GOP loop (with and without the reference's redundant tail pass)
    and the device-side Y-PSNR against the oracle's restatement of the metric."""
    from selfc_amd import harness
    g = load_golden("g8_large_stack")
    s = load_golden("g7_stp_l2_full_rev")
    net = _large_net(dev, g, "l2", s)
    x = torch.rand(2 * T, 3, 32, 48, generator=torch.Generator().manual_seed(21))
    fl, fh = harness.rescale_test(net, x.to(dev))
    fl2, fh2 = harness.rescale_test(net, x.to(dev), reference_tail_pass=True)
    assert torch.equal(fl, fl2) and torch.equal(fh, fh2)           # the extra pass is discarded
    z_ref = O.large_fwd(g, x, T)
    lr_ref = O.quantize(z_ref[:, :3])
    hf_ref = O.stp_v2_parameters(subdict(s, "stp_net"), lr_ref, T)
    xr_ref = O.large_inv_from_latent(g, torch.cat((lr_ref, hf_ref), 1), T)
    assert (fl.cpu() - lr_ref).abs().max() <= 1.0 / 255 + 1e-6
    p_dev = harness.psnr_y(fh, x.to(dev))
    p_ora = O.psnr_per_frame(O.rgb_to_y(fh.cpu()), O.rgb_to_y(x))
    assert max(abs(a - b) for a, b in zip(p_dev, p_ora)) < 1e-4      # same images: the metric kernel itself
    p_ref = O.psnr_per_frame(O.rgb_to_y(xr_ref), O.rgb_to_y(x))
    assert max(abs(a - b) for a, b in zip(p_dev, p_ref)) < 0.05      # HIP path vs oracle path through the quantiser
    assert harness.psnr_y(x.to(dev), x.to(dev))[0] == float("inf")


def test_rescale_video_ragged_length_and_sharding(dev):
    """A 17-frame clip (3 GOPs, the last padded) processed whole and as two rank shards gives the same frames."""
    from selfc_amd import harness
    g = load_golden("g8_large_stack")
    s = load_golden("g7_stp_l2_full_rev")
    net = _large_net(dev, g, "l2", s)
    v = torch.rand(17, 3, 32, 32, generator=torch.Generator().manual_seed(31)).to(dev)
    whole = harness.rescale_video(net, v)
    assert whole["frames"] == list(range(17)) and whole["rec"].shape == (17, 3, 32, 32)
    parts = [harness.rescale_video(net, v, rank=r, world=2) for r in range(2)]
    assert sorted(parts[0]["frames"] + parts[1]["frames"]) == list(range(17))
    for p in parts:
        for j, fi in enumerate(p["frames"]):
            assert torch.equal(p["rec"][j], whole["rec"][fi]) and torch.equal(p["lr"][j], whole["lr"][fi])
    # first GOP equals the plain 7-frame test loop
    fl, fh = harness.rescale_test(net, v[:7])
    assert torch.equal(fh, whole["rec"][:7])


def test_ssim_y(dev):
    from selfc_amd import harness
    g = load_golden("g12_ssim_y")
    got = torch.tensor(harness.ssim_y(g["a"].to(dev), g["b"].to(dev)), dtype=torch.float64)
    assert rel_err(got, g["ssim"]) < 1e-5
    assert abs(harness.ssim_y(g["a"].to(dev), g["a"].to(dev))[1] - 1.0) < 1e-6
    x = torch.rand(2, 3, 37, 61)                       # ragged block edges
    y = (x + 0.1 * torch.randn_like(x)).clamp(0, 1)
    ref = O.ssim_per_frame(O.rgb_to_y(x), O.rgb_to_y(y))
    assert rel_err(torch.tensor(harness.ssim_y(x.to(dev), y.to(dev))), torch.tensor(ref)) < 1e-5


def test_rescale_metrics(dev):
    """test_rescaling.py's per-batch metrics through the harness against the same quantities from the oracle."""
    from selfc_amd import harness
    g = load_golden("g8_large_stack")
    net = _large_net(dev, g, stp=load_golden("g7_stp_l2_full_rev"))
    x = torch.rand(T, 3, 64, 64, generator=torch.Generator().manual_seed(77))      # LR 16x16: the 11-tap SSIM window fits
    m = harness.rescale_metrics(net, x.to(dev))
    z = O.large_fwd(g, x, T)
    lr = O.quantize(z[:, :3])
    params = {**{k: v for k, v in g.items() if k.startswith("operations.")},
              **{k: v for k, v in load_golden("g7_stp_l2_full_rev").items() if k.startswith("stp_net.")}}
    rec, _ = O.large_rev_l2(params, lr, T)
    ref_l = O.gaussian_downsample(x)
    want_psnr = sum(O.psnr_per_frame(O.rgb_to_y(rec), O.rgb_to_y(x))) / T
    want_ssim = sum(O.ssim_per_frame(O.rgb_to_y(rec), O.rgb_to_y(x))) / T
    assert abs(m["psnr_y"] - want_psnr) < 0.05 and abs(m["ssim_y"] - want_ssim) < 2e-3
    want_lr_psnr = sum(O.psnr_per_frame(O.rgb_to_y(lr), O.rgb_to_y(ref_l))) / T
    want_lr_ssim = sum(O.ssim_per_frame(O.rgb_to_y(lr), O.rgb_to_y(ref_l))) / T
    assert abs(m["lr_psnr_y"] - want_lr_psnr) < 0.05 and abs(m["lr_ssim_y"] - want_lr_ssim) < 2e-3


def test_full_test_path_pipeline(dev):
    """pipeline.FullTestPath (stack fwd, quantise, STP, stack rev in the latent layout; eager, captured, and split over
    two streams) against the module API calls SelfCModel.test makes."""
    from selfc_amd.modules.Quantization import Quantization
    from selfc_amd.pipeline import FullTestPath, MultiStreamRoundTrip
    g = load_golden("g8_large_stack")
    net = _large_net(dev, g, stp=load_golden("g7_stp_l2_full_rev"))          # l2 head: deterministic
    x = torch.rand(2 * T, 3, 32, 48, generator=torch.Generator().manual_seed(5)).to(dev)
    with torch.no_grad():
        z, _ = net(x=x, rev=False)
        lr = Quantization()(z[:, :3])
        want, _ = net(x=lr, rev=True)
        p = FullTestPath(net, 2 * T, 32, 48, dev)
        got = p.run(x).clone()
        assert torch.equal(p.lr, lr)
        assert rel_err(got.cpu(), want.cpu()) < 1e-6
        ms = MultiStreamRoundTrip(net, 2 * T, 32, 48, dev, 2, part_cls=FullTestPath)
        ms.capture(x)
        assert rel_err(ms.replay().cpu(), want.cpu()) < 1e-6


def test_gaussian_downsample_ref_L(dev):
    from selfc_amd import harness
    g = load_golden("g10_gauss")
    x = g["x"]                                          # [C,T,H,W] planes
    y = harness.gaussian_downsample(x.reshape(1, -1, x.shape[2], x.shape[3]).to(dev)).reshape(g["y"].shape)
    assert rel_err(y.cpu(), g["y"]) < 1e-5
    with pytest.raises(RuntimeError):
        harness.gaussian_downsample(torch.zeros(1, 3, 6, 8, device=dev))     # not a multiple of 4 / too small


def test_selfc_haar_variant(dev):
    """model "SelfC": Haar + InvBlockExp(DBNet) + STP v1 (D2DTNet conditioner, l2 head), fwd (incl. neg_llh) and rev."""
    from selfc_amd.modules.SelfC_arch_inv import SelfCInvNet
    g = load_golden("g8_selfc_haar")
    opt1 = {"stp_d2d_inner_c": 32, "stp_temporal_c": 32, "fh_loss": "l2", "gmm_mixture_num": 5,
            "stp_blk_num": 2, "condition_func": "D2DTNet"}
    net = SelfCInvNet(opt1, 3, 3, "DBNet", [1], 1)
    net.load_state_dict({k: v for k, v in g.items() if k.startswith(("operations.", "stp_net."))}, strict=True)
    net.to(dev).eval()
    with torch.no_grad():
        z, loss = net(x=g["x"].to(dev), rev=False)
        assert rel_err(z.cpu(), g["z"]) < TOL
        assert abs(loss.item() - g["loss_c"].item()) < 5e-3 * g["loss_c"].item()
        xr, hf = net(x=g["lr"].to(dev), rev=True)
        assert rel_err(hf.cpu(), g["hf"]) < TOL
        assert rel_err(xr.cpu(), g["x_rev"]) < TOL


def test_selfc_haar_variant_feature_calapse_block(dev):
    """The default STP v1 conditioner: FeatureCalapseBlock = SpaceToDepth + dense block with gc = 128 and
    (3,3,3) conv1 / conv5 (generic plane-list conv kernel) + PixelShuffle.  16 M STP parameters are filled
    with the fixture's seeded routine instead of being stored."""
    from selfc_amd.modules.SelfC_arch_inv import SelfCInvNet
    g = load_golden("g8_selfc_haar_fcb")
    opt2 = {"stp_d2d_inner_c": 32, "stp_temporal_c": 32, "fh_loss": "l2", "gmm_mixture_num": 5,
            "stp_blk_num": 2, "condition_func": "FeatureCalapseBlock"}
    net = SelfCInvNet(opt2, 3, 3, "DBNet", [1], 1)
    missing, unexpected = net.load_state_dict({k: v for k, v in g.items() if k.startswith("operations.")}, strict=False)
    assert not unexpected and all(k.startswith("stp_net.") for k in missing)
    seeded_fill(dict(net.stp_net.named_parameters()), int(g["stp_fill_seed"]))
    net.to(dev).eval()
    with torch.no_grad():
        y1 = net.stp_net.blk1(g["lr"].to(dev))
        assert rel_err(y1.cpu(), g["blk1_y"]) < TOL
        z, loss = net(x=g["x"].to(dev), rev=False)
        assert rel_err(z.cpu(), g["z"]) < TOL
        assert abs(loss.item() - g["loss_c"].item()) < 5e-3 * g["loss_c"].item()
        xr, hf = net(x=g["lr"].to(dev), rev=True)
        assert rel_err(hf.cpu(), g["hf"]) < TOL
        assert rel_err(xr.cpu(), g["x_rev"]) < TOL


def test_1080p_tile_invertibility(dev):
    """BASELINE config 5 shape (7x3x1080x1920, latent 270x480: not a multiple of the 16x16 tile):
    the stack inverts its own output on the device and the LR channels stay finite."""
    from selfc_amd import runtime as rt, _lib
    g = load_golden("g8_large_stack")
    net = _large_net(dev, g)
    x = torch.rand(T, 3, 1080, 1920, generator=torch.Generator().manual_seed(99)).to(dev)
    with torch.no_grad():
        z, _ = net(x=x, rev=False)
        assert torch.isfinite(z).all()
        ws = net._workspace(x, T, 270, 480)
        arr, nblk = net._stack()
        rt.call("selfc_invstack_run", arr, nblk, ws.latent(), 1, _lib.stream_ptr())
        z0 = rt.latent_to_nchw(ws)
        fa = net.operations[0](x)
    assert rel_err(z0.cpu(), fa.cpu()) < TOL


def test_1080p_against_the_oracle_on_corner_crops(dev):
    """VERDICT r4 item 3 - config 5's size held against the ORACLE, not against the stack's own inverse (a tile-edge or addressing
    bug applied symmetrically in both directions passes test_1080p_tile_invertibility).  The HIP path runs the whole frame
    (7x3x1080x1920, latent 270x480 = 16.9 x 30 tiles: the bottom tile row is 14 rows); the oracle runs four 640x640 HR crops
    that each hold one true corner of the frame.  A block's output depends on its input within 8 latent pixels (four 3x3 convs
    of F, then four of G / H; the temporal conv5 is pointwise in space), the eight blocks on 64: inside a crop, everything
    further than 64 latent pixels from the two CUT edges is exactly what the full frame gives - a 96x96 latent corner per crop,
    incl. the frame border on two sides, the partial tile row and the last tile column.  Forward latent and inverse
    reconstruction (the same quantised latent into both sides), 1e-3 (SelfC_model.py:199-250 is the caller of this size)."""
    g = load_golden("g8_large_stack")
    net = _large_net(dev, g)
    Hh, Ww, C, M = 1080, 1920, 160, 64                      # latent crop side, dependency radius of the stack
    h, w = Hh // 4, Ww // 4
    x = torch.rand(T, 3, Hh, Ww, generator=torch.Generator().manual_seed(1080))
    with torch.no_grad():
        z, _ = net(x=x.to(dev), rev=False)
        z = z.cpu()
        zq = torch.cat((O.quantize(z[:, :3]), z[:, 3:]), 1)             # the latent both inverses start from
        xr = net.inverse_from_latent(zq.to(dev)).cpu()
    worst = {}
    for name, r0, c0 in (("top-left", 0, 0), ("top-right", 0, w - C), ("bottom-left", h - C, 0), ("bottom-right", h - C, w - C)):
        # valid interior in crop coordinates: away from the cut edges (the frame's own edges are real borders)
        rs = slice(0, C - M) if r0 == 0 else slice(M, C)
        cs = slice(0, C - M) if c0 == 0 else slice(M, C)
        xc = x[:, :, 4 * r0:4 * (r0 + C), 4 * c0:4 * (c0 + C)].contiguous()
        z_or = O.large_fwd(g, xc, T)
        e_f = rel_err(z[:, :, r0:r0 + C, c0:c0 + C][:, :, rs, cs], z_or[:, :, rs, cs])
        e_g = group_err(z[:, :, r0:r0 + C, c0:c0 + C][:, :, rs, cs], z_or[:, :, rs, cs])
        x_or = O.large_inv_from_latent(g, zq[:, :, r0:r0 + C, c0:c0 + C].contiguous(), T)
        hrs = slice(4 * rs.start, 4 * rs.stop)
        hcs = slice(4 * cs.start, 4 * cs.stop)
        e_i = rel_err(xr[:, :, 4 * r0:4 * (r0 + C), 4 * c0:4 * (c0 + C)][:, :, hrs, hcs], x_or[:, :, hrs, hcs])
        worst[name] = (e_f, e_g, e_i)
        # the method's own check: INSIDE the cut margin the crop must differ from the frame (else the margin proves nothing)
        if name == "bottom-right":
            cut = rel_err(z[:, :, r0:r0 + C, c0:c0 + C][:, :, :8, :], z_or[:, :, :8, :])
            assert cut > 10 * TOL, cut
    assert all(max(v) < TOL for v in worst.values()), worst


def test_freq_k2_and_clip_len_3(dev):
    """FrequencyAnalyzer(k=2) (the codec variant's split, SelfC_Codec_arch_inv.py:78-98) and a 3-frame clip length."""
    from selfc_amd import GlobalVar
    from selfc_amd.modules.SelfC_GMM_arch_inv import FrequencyAnalyzer
    x = torch.rand(3, 3, 16, 24, generator=torch.Generator().manual_seed(3))
    fa2 = FrequencyAnalyzer(3, k=2)
    assert torch.equal(fa2(x.to(dev)).cpu(), O.freq_fwd(x, 2))
    z = torch.randn(3, 15, 8, 12, generator=torch.Generator().manual_seed(4))
    assert torch.equal(fa2(z.to(dev), rev=True).cpu(), O.freq_inv(z, 2))
    g = load_golden("g8_large_stack")
    net = _large_net(dev, g)
    xx = torch.rand(6, 3, 32, 48, generator=torch.Generator().manual_seed(8))
    try:
        GlobalVar.set_Temporal_LEN(3)
        with torch.no_grad():
            zz, _ = net(x=xx.to(dev), rev=False)
        assert rel_err(zz.cpu(), O.large_fwd(g, xx, 3)) < TOL
    finally:
        GlobalVar.set_Temporal_LEN(T)


@pytest.mark.parametrize("h,w", [(16, 16), (20, 28), (4, 8), (68, 4)])
def test_tiny_and_odd_latent_sizes(dev, h, w):
    """latent sizes below / across the 16x16 conv tile, the 128-pixel temporal strip and the pooling bins
    (HR = 4 x latent): forward, inverse and the STP reverse against the oracle."""
    g = load_golden("g8_large_stack")
    s = load_golden("g7_stp_l2_full_rev")
    net = _large_net(dev, g, "l2", s)
    x = torch.rand(T, 3, 4 * h, 4 * w, generator=torch.Generator().manual_seed(h * 31 + w))
    z_ref = O.large_fwd(g, x, T)
    with torch.no_grad():
        z, _ = net(x=x.to(dev), rev=False)
        assert rel_err(z.cpu(), z_ref) < TOL
        lr = O.quantize(z_ref[:, :3])
        xr, hf = net(x=lr.to(dev), rev=True)
    hf_ref = O.stp_v2_parameters(subdict(s, "stp_net"), lr, T)
    assert rel_err(hf.cpu(), hf_ref) < TOL
    assert rel_err(xr.cpu(), O.large_inv_from_latent(g, torch.cat((lr, hf_ref), 1), T)) < TOL


def test_rejects_bad_arguments(dev):
    from selfc_amd import _lib
    L = _lib.lib()
    assert L.selfc_haar_fwd_nchw(None, None, 1, 3, 8, 8, None) == -1
    x = torch.zeros(1, 3, 7, 8, device=dev)
    assert L.selfc_haar_fwd_nchw(x.data_ptr(), x.data_ptr(), 1, 3, 7, 8, None) == -1   # odd height
    assert L.selfc_invstack_run(None, 1, None, 0, None) == -1
    from selfc_amd.modules.Subnet_constructor import D2DTInput
    with pytest.raises(RuntimeError):
        D2DTInput(48, 3).to(dev)(torch.zeros(7, 3, 8, 8, device=dev))   # wrong channel count
    with pytest.raises(RuntimeError):
        D2DTInput(48, 3)(torch.zeros(7, 48, 8, 8))                      # CPU tensor: no fallback


def test_globalagg_scratch_grows_per_buffer(dev):
    """one GlobalAgg used first on few clips of many pixels, then on many clips of few pixels - its scratch must
    follow the call (a shared attention buffer used to be written past its end; since the attention became the mix kernel's
    prologue only the pooling partials are left, sized per call)."""
    from selfc_amd.modules.SelfC_GMM_arch_inv import GlobalAgg
    g = load_golden("g6_globalagg")
    ga = GlobalAgg(64)
    ga.load_state_dict({k: v for k, v in g.items() if k.split(".")[0] in ("fc", "proj1", "proj2", "proj3")}, strict=True)
    ga.to(dev)
    sd = {k: v for k, v in g.items() if k.split(".")[0] in ("fc", "proj1", "proj2", "proj3")}
    gen = torch.Generator().manual_seed(3)
    for n, h, w in ((T, 64, 112), (4 * T, 16, 28), (T, 20, 36)):
        x = torch.randn(n, 64, h, w, generator=gen)
        with torch.no_grad():
            y = ga(x.to(dev))
        assert rel_err(y.cpu(), O.global_agg(sd, x, T)) < TOL


def test_pipeline_follows_weight_changes(dev):
    """RescaleRoundTrip re-packs when the net's weights change (eager) and refuses to replay a graph that was
    captured on the old weights."""
    from selfc_amd.pipeline import RescaleRoundTrip
    g = load_golden("g8_large_stack")
    net = _large_net(dev, g)
    x = torch.rand(T, 3, 32, 48, generator=torch.Generator().manual_seed(1)).to(dev)
    with torch.no_grad():
        rt_ = RescaleRoundTrip(net, T, 32, 48, dev)
        a = rt_.run(x).clone()
        rt_.capture(x)
        assert torch.equal(rt_.replay(), a)
        for p in net.operations[1].F.parameters():
            p.mul_(1.5)
        b = rt_.run(x).clone()                           # eager: repacked
        assert not torch.equal(a, b)
        assert torch.equal(b, RescaleRoundTrip(net, T, 32, 48, dev).run(x))
        with pytest.raises(RuntimeError, match="capture"):
            rt_.replay()


def test_latent_flags_and_fd_next_abi(dev):
    """selfc_latent.flags / fd_next (ABI 7 / 8) through the C ABI on one SelfC-large block: without SELFC_LAT_KEEP_FEATURES the
    pairwise-fused F launches leave the f3 / f4 planes of `fd` untouched (poison survives) and the block's outputs are the
    same bits as with it (where f3 / f4 equal the layer-wise path's features); with fd_next the f16 copy of the updated x2
    lands in the other buffer and this block's own F input planes keep x2."""
    from selfc_amd import _lib, runtime as rt
    g = load_golden("g8_large_stack")
    net = _large_net(dev, g)
    blk = net._blocks()[0]
    pb = rt.packed_block(blk)
    n, h, w = T, 16, 24
    z = torch.randn(n, 51, h, w, generator=torch.Generator().manual_seed(12)) * 0.5
    outs, fds = {}, {}
    for name, keep, use_next in (("inference", False, False), ("keep", True, False), ("next", True, True)):
        ws = rt.Workspace(dev, blk.F.kind, n, T, h, w, 3, 48)
        rt.nchw_to_latent(z.to(dev), ws)
        x2_planes = ws.fd[:2].clone()
        ws.fd[4:6].fill_(777.0)                                          # poison f3 / f4
        nxt = torch.full_like(ws.fd, -5.0) if use_next else None
        lat = _lib.Latent(ws.kind, n, T, h, w, 3, 48, ws.x1.data_ptr(), ws.x2.data_ptr(), ws.fd.data_ptr(), ws.gd.data_ptr(),
                          ws.hd.data_ptr(), None, ws.pf.data_ptr(), _lib.LAT_KEEP_FEATURES if keep else 0,
                          None if nxt is None else nxt.data_ptr())
        rt.call("selfc_invblock_run", pb.struct(), lat, 0, _lib.stream_ptr())
        outs[name] = rt.latent_to_nchw(ws).cpu()
        fds[name] = (ws.fd.clone(), x2_planes, nxt)
    assert torch.equal(outs["inference"], outs["keep"]) and torch.equal(outs["keep"], outs["next"])
    assert rel_err(outs["keep"], O.invblock("D2DTNet", {k[len("operations.1."):]: v for k, v in g.items() if k.startswith("operations.1.")}, z, 3, T)[0]) < TOL
    fd_inf, fd_keep = fds["inference"][0], fds["keep"][0]
    assert bool((fd_inf[4:6] == 777.0).all())                            # nothing stored f3 / f4
    assert not bool((fd_keep[4:6] == 777.0).any())                       # the training forward has them
    assert torch.equal(fd_inf[2:4], fd_keep[2:4])                        # f1 / f2 are written either way (pair 1 reads them)
    # without fd_next the G/H epilogue replaced the block's own x2 planes by y2; with it they still hold x2 and y2 is in `nxt`
    fd_next, x2_before, nxt = fds["next"]
    assert torch.equal(fd_next[:2], x2_before)
    assert torch.equal(nxt[0], fd_keep[0]) and torch.equal(nxt[1, ..., :16], fd_keep[1, ..., :16])      # the 48 real channels of y2
    assert bool((nxt[1, ..., 16:] == -5.0).all()) and bool((nxt[2:] == -5.0).all())                      # nothing else is touched


@pytest.mark.parametrize("rev", [0, 1])
def test_latent_out_of_place_abi(dev, rev):
    """selfc_latent.x1_out / x2_out (ABI 9): the block writes its updated x1 / x2 into the named buffers, bit for bit what the
    in-place call produces, and leaves its inputs untouched (the training forward keeps them for the backward pass)."""
    from selfc_amd import _lib, runtime as rt
    g = load_golden("g8_large_stack")
    net = _large_net(dev, g)
    blk = net._blocks()[0]
    pb = rt.packed_block(blk)
    n, h, w = T, 16, 24
    z = torch.randn(n, 51, h, w, generator=torch.Generator().manual_seed(13)) * 0.5
    res = {}
    for mode in ("in_place", "out_of_place"):
        ws = rt.Workspace(dev, blk.F.kind, n, T, h, w, 3, 48)
        rt.nchw_to_latent(z.to(dev), ws)
        x1_in, x2_in = ws.x1.clone(), ws.x2.clone()
        y1, y2 = torch.full_like(ws.x1, 9.0), torch.full_like(ws.x2, 9.0)
        lat = _lib.Latent(ws.kind, n, T, h, w, 3, 48, ws.x1.data_ptr(), ws.x2.data_ptr(), ws.fd.data_ptr(), ws.gd.data_ptr(),
                          ws.hd.data_ptr(), None, ws.pf.data_ptr(), _lib.LAT_KEEP_FEATURES, None,
                          y1.data_ptr() if mode == "out_of_place" else None, y2.data_ptr() if mode == "out_of_place" else None)
        rt.call("selfc_invblock_run", pb.struct(), lat, rev, _lib.stream_ptr())
        if mode == "out_of_place":
            assert torch.equal(ws.x1, x1_in) and torch.equal(ws.x2, x2_in)           # inputs intact
            res[mode] = (y1.clone(), y2.clone())
        else:
            assert bool((y1 == 9.0).all()) and bool((y2 == 9.0).all())
            res[mode] = (ws.x1.clone(), ws.x2.clone())
    assert torch.equal(res["in_place"][0][..., :3], res["out_of_place"][0][..., :3])
    assert torch.equal(res["in_place"][1], res["out_of_place"][1])


def test_latent_out_of_place_on_the_layerwise_kernels(dev):
    """The same ABI-9 contract on the layer-wise kernels (conv3x3 / tconv5 epilogues instead of the fused launches' f_couple and
    conv5 epilogue): the developer switches are read once per process, hence a child process."""
    import subprocess
    env = dict(os.environ, SELFC_NO_FUSE="1", SELFC_NO_FUSE_F="1")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k", "test_latent_out_of_place_abi"],
                         env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0 and "2 passed" in out.stdout, out.stdout[-1500:] + out.stderr[-1500:]


def test_profile_calibration_exports(dev):
    """selfc_profile_calibrate / selfc_profile_clock_sample (bench.py's box_calibration): plausible figures on an MI355X, argument
    checks, and the clock sampler ends after the time it was given."""
    import ctypes as C
    import time
    from selfc_amd import _lib
    L = _lib.lib()
    m, c = C.c_double(0.0), C.c_double(0.0)
    assert L.selfc_profile_calibrate(C.byref(m), C.byref(c), _lib.stream_ptr()) == 0
    assert 300.0 < m.value < 4000.0 and 500.0 < c.value < 12000.0, (m.value, c.value)
    assert L.selfc_profile_calibrate(None, C.byref(c), _lib.stream_ptr()) == -1
    out = torch.zeros(2, dtype=torch.int64, device=dev)
    assert L.selfc_profile_clock_sample(out.data_ptr(), 0, _lib.stream_ptr()) == -1
    assert L.selfc_profile_clock_sample(out.data_ptr(), 600000, _lib.stream_ptr()) == -1
    t0 = time.perf_counter()
    assert L.selfc_profile_clock_sample(out.data_ptr(), 2000, _lib.stream_ptr()) == 0
    torch.cuda.synchronize()
    assert time.perf_counter() - t0 < 0.5
    cycles, ticks = int(out[0]), int(out[1])
    assert 200000 <= ticks < 400000                                  # 2,000 us of the 100 MHz counter (the wave checks it every ~64 sleep units)
    assert 0.3 < 0.1 * cycles / ticks < 3.0                          # GHz


def test_module_call_is_the_cached_graph_and_stays_a_drop_in(dev):
    """The reference's callers only do netG(x=..., rev=...) (SelfC_model.py:213-230).  In eval / no_grad that call goes through a
    cached two-stream hipGraph from its second use on (pipeline.ModuleGraph).  Held to: (1) the graph path returns what the eager
    first call returns, bit for bit, and what the oracle-pinned fixtures say; (2) outputs are FRESH tensors - a later call does not
    touch what an earlier one returned; (3) changed weights re-capture; (4) a call made while the caller captures falls back to
    the eager launches (no nested capture); (5) SELFC-side state: the cache holds hipGraphs, so it is weak on the net and a
    deepcopy of the net works."""
    import copy
    from selfc_amd import pipeline
    g = load_golden("g8_large_stack")
    s = load_golden("g7_stp_l2_full_rev")
    net = _large_net(dev, g, "l2", s)                       # l2 head: deterministic reverse
    gen = torch.Generator().manual_seed(11)
    xs = [torch.rand(2 * T, 3, 32, 48, generator=gen).to(dev) for _ in range(3)]
    with torch.no_grad():
        assert pipeline.MODULE_GRAPH
        pipeline.MODULE_GRAPH = False
        want = [(net(x=x, rev=False)[0], *net(x=net(x=x, rev=False)[0][:, :3], rev=True)) for x in xs]     # eager references
        lat = [net.inverse_from_latent(w[0]) for w in want]
        pipeline.MODULE_GRAPH = True
        got = []
        for x in xs:                                          # call 1 eager (shape seen once), 2 captures, 3 replays
            z, zero = net(x=x, rev=False)
            assert float(zero) == 0.0
            xr, hf = net(x=z[:, :3], rev=True)
            got.append((z, xr, hf, net.inverse_from_latent(z)))
        cache = pipeline._MODULE_GRAPHS[net]
        live = {k[0]: v for k, v in cache.items() if isinstance(v, pipeline.ModuleGraph)}
        assert set(live) == {"fwd", "rev", "revlat"} and all(v.graph is not None and v.nstreams == 2 for v in live.values())
        for (z, xr, hf, xl), (wz, wxr, whf), wl in zip(got, want, lat):
            assert torch.equal(z, wz) and torch.equal(xr, wxr) and torch.equal(hf, whf) and torch.equal(xl, wl)
        ptrs = {t.data_ptr() for tup in got for t in tup}
        assert len(ptrs) == 12                                # (2): twelve live results, twelve buffers
        # the fixtures (one clip -> a single-stream graph)
        for _ in range(3):
            z1, _ = net(x=g["x"].to(dev), rev=False)
            xr1, hf1 = net(x=s["lr"].to(dev), rev=True)
        assert rel_err(z1.cpu(), g["z"]) < TOL and rel_err(xr1.cpu(), s["x_rev"]) < TOL and rel_err(hf1.cpu(), s["hf"]) < TOL
        # (3) weights change between two calls: the next call re-captures and follows them
        for p in net.operations[2].G.parameters():
            p.mul_(1.25)
        z_new, _ = net(x=xs[0], rev=False)
        pipeline.MODULE_GRAPH = False
        z_ref, _ = net(x=xs[0], rev=False)
        pipeline.MODULE_GRAPH = True
        assert torch.equal(z_new, z_ref) and not torch.equal(z_new, got[0][0])
        sd = {k: v for k, v in net.state_dict().items()}
        z_or = O.large_fwd({k: v.cpu() for k, v in sd.items()}, xs[0].cpu(), T)
        assert rel_err(z_new.cpu(), z_or) < TOL
        # (4) inside somebody else's capture the module launches eagerly into that capture
        static_x = xs[1].clone()
        net(x=static_x, rev=False)
        torch.cuda.synchronize()
        cg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(cg):
            z_cap, _ = net(x=static_x, rev=False)
        cg.replay()
        pipeline.MODULE_GRAPH = False
        assert torch.equal(z_cap, net(x=static_x, rev=False)[0])
        pipeline.MODULE_GRAPH = True
    # (5)
    net2 = copy.deepcopy(net)
    assert net2 not in pipeline._MODULE_GRAPHS


def test_module_graph_follows_re_registered_parameters(dev):
    """load_state_dict(assign=True) (and parametrize, or `conv.weight = nn.Parameter(...)`) puts NEW Parameter objects
    on the net; the old ones keep their version counters and addresses.  A cached ModuleGraph / pre-bound pipeline that stamped
    the objects it saw at construction would replay the old packed weights for ever - the eager path re-derives its list per call
    and is safe.  The graph paths must re-capture: third call == eager on the new weights, for the module API and for
    RescaleRoundTrip (eager run and refused stale replay)."""
    from selfc_amd import pipeline
    g = load_golden("g8_large_stack")
    net = _large_net(dev, g, "l2", load_golden("g7_stp_l2_full_rev"))
    x = torch.rand(2 * T, 3, 32, 48, generator=torch.Generator().manual_seed(5)).to(dev)
    with torch.no_grad():
        z1, _ = net(x=x, rev=False)            # seen once
        z2, _ = net(x=x, rev=False)            # captured
        assert torch.equal(z1, z2)
        rtp = pipeline.RescaleRoundTrip(net, 2 * T, 32, 48, dev)
        y_old = rtp.run(x).clone()
        rtp.capture(x)
        new_sd = {k: (v * 1.1 if k.startswith("operations.3.") and v.is_floating_point() else v.clone()) for k, v in net.state_dict().items()}
        old_first = net.operations[3].F.conv1.weight
        net.load_state_dict(new_sd, assign=True)
        assert net.operations[3].F.conv1.weight is not old_first            # re-registered, not copied into
        z3, _ = net(x=x, rev=False)            # must notice and re-capture
        pipeline.MODULE_GRAPH = False
        try:
            z_eager, _ = net(x=x, rev=False)
        finally:
            pipeline.MODULE_GRAPH = True
        assert torch.equal(z3, z_eager) and not torch.equal(z3, z2)
        with pytest.raises(RuntimeError, match="capture\\(\\) again"):
            rtp.replay()
        y_new = rtp.run(x)
        assert not torch.equal(y_new, y_old)
        z_or = O.large_fwd({k: v.cpu() for k, v in new_sd.items()}, x.cpu(), T)
        assert rel_err(z3.cpu(), z_or) < TOL
        # ONE parameter in the middle of a block replaced (not the block's first one): every cache keyed on the parameter
        # list must notice - the module graph re-captures, the pre-bound pipeline refuses its stale replay
        rtp.capture(x)
        z4, _ = net(x=x, rev=False)                                          # cached graph on the current weights
        blk = net.operations[5]
        blk.G.conv3.weight = torch.nn.Parameter(blk.G.conv3.weight.detach() * 0.9)
        z5, _ = net(x=x, rev=False)
        pipeline.MODULE_GRAPH = False
        try:
            z5_eager, _ = net(x=x, rev=False)
        finally:
            pipeline.MODULE_GRAPH = True
        assert torch.equal(z5, z5_eager) and not torch.equal(z5, z4)
        with pytest.raises(RuntimeError, match="capture\\(\\) again"):
            rtp.replay()
        sd5 = {k: v.cpu() for k, v in net.state_dict().items()}
        assert rel_err(z5.cpu(), O.large_fwd(sd5, x.cpu(), T)) < TOL
