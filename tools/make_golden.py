"""Generate tests/golden/*.npz by running the REAL reference modules on CPU.

Runs only in the build container (needs /root/reference, which never travels to
the GPU box).  The reference is imported read-only (no bytecode written); only
its inputs / seeded weights / outputs are stored - no reference source text.

    python tools/make_golden.py            # (re)writes tests/golden/*.npz

Vectors:
  g1_haar          HaarDownsampling fwd/rev, (7,3,32,32) and 2nd level C=12
  g2_freq          FrequencyAnalyzer fwd and rev (not inverses - trap 3)
  g3_denseblock    DenseBlock(9,3), DenseBlock(3,9) with non-zero conv5
  g4_d2dt          D2DTInput(48,3), (3,48) at T=7, b=2 (clip boundaries)
  g5_invblock_*    InvBlockExp fwd / rev / s / jacobian (DBNet 12ch, D2DT 51ch)
  g8_large_stack   SelfC-large FrequencyAnalyzer + 8 InvBlockExp fwd, and the op
                   loop reversed on the forward's own latent
  g8_haar_nets     InvRescaleNet / Haar SelfC op stack (DBNet,[1],1)
  g9_quant         Quantization on edge values
  g6_globalagg     GlobalAgg(64) at 16x16 (replicating bins) and 20x36
  g7_stp           STPNet v2 raw head output + GMM sample with injected eps (7,3,8,12); l2 full reverse
"""
import os
import sys
import types

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
REF = "/root/reference/codes"
sys.path.insert(0, REF)
_tv = types.ModuleType("torchvision")
_tvo = types.ModuleType("torchvision.ops")
_tv.ops = _tvo
sys.modules["torchvision"] = _tv
sys.modules["torchvision.ops"] = _tvo

import numpy as np  # noqa: E402
import torch  # noqa: E402

from global_var import GlobalVar  # noqa: E402
import models.modules.Inv_arch as IA  # noqa: E402
import models.modules.Subnet_constructor as SC  # noqa: E402
import models.modules.SelfC_GMM_arch_inv as GA  # noqa: E402
from models.modules.Quantization import Quantization  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
T = 7


def sd_np(mod, prefix=""):
    return {prefix + k: v.detach().numpy().copy() for k, v in mod.state_dict().items()}


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"{name}: {os.path.getsize(path) / 1e6:.2f} MB")


def rerandomise_conv5(mod, gen, std=0.02):
    """DenseBlock zero-inits conv5 (Subnet_constructor.py:22): make it non-zero so
    the golden exercises the layer."""
    with torch.no_grad():
        mod.conv5.weight.copy_(torch.randn(mod.conv5.weight.shape, generator=gen) * std)
        mod.conv5.bias.copy_(torch.randn(mod.conv5.bias.shape, generator=gen) * std)


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    GlobalVar.set_Temporal_LEN(T)
    g = torch.Generator().manual_seed(1234)

    with torch.no_grad():
        # ---- G1 Haar
        x = torch.rand(7, 3, 32, 32, generator=g)
        h1 = IA.HaarDownsampling(3)
        y = h1(x)
        jac_f = h1.jacobian(x)
        h2 = IA.HaarDownsampling(12)
        y2 = h2(y)
        xr = h1(y, rev=True)
        jac_r = h1.jacobian(y, rev=True)
        zrand = torch.randn(2, 12, 5, 7, generator=g)
        save("g1_haar", x=x.numpy(), y=y.numpy(), y2=y2.numpy(), xr=xr.numpy(),
             zrand=zrand.numpy(), zrand_inv=h1(zrand, rev=True).numpy(),
             jac_fwd=np.float64(jac_f), jac_rev=np.float64(jac_r),
             haar_weights=h1.haar_weights.numpy())

        # ---- G2 FrequencyAnalyzer
        fa = GA.FrequencyAnalyzer(3)
        x = torch.rand(7, 3, 32, 48, generator=g)
        yf = fa(x)
        zr = torch.randn(7, 51, 8, 12, generator=g)
        save("g2_freq", x=x.numpy(), y=yf.numpy(), z=zr.numpy(), z_rev=fa(zr, rev=True).numpy(),
             y_rev=fa(yf, rev=True).numpy())

        # ---- G3 DenseBlock
        torch.manual_seed(3)
        arrs = {}
        for tag, (ci, co) in {"f": (9, 3), "g": (3, 9)}.items():
            m = SC.DenseBlock(ci, co, "xavier")
            rerandomise_conv5(m, g)
            x = torch.randn(2, ci, 16, 20, generator=g)
            arrs.update(sd_np(m, f"{tag}."))
            arrs[f"{tag}_x"] = x.numpy()
            arrs[f"{tag}_y"] = m(x).numpy()
        save("g3_denseblock", **arrs)

        # ---- G4 D2DTInput
        torch.manual_seed(4)
        arrs = {}
        for tag, (ci, co) in {"f": (48, 3), "g": (3, 48)}.items():
            m = SC.D2DTInput(ci, co, "xavier")
            x = torch.randn(2 * T, ci, 12, 20, generator=g)
            arrs.update(sd_np(m, f"{tag}."))
            arrs[f"{tag}_x"] = x.numpy()
            arrs[f"{tag}_y"] = m(x).numpy()
        save("g4_d2dt", **arrs)

        # ---- G5 InvBlockExp
        torch.manual_seed(5)
        blk = IA.InvBlockExp(SC.subnet("DBNet", "xavier"), 12, 3)
        for sub in (blk.F, blk.G, blk.H):
            rerandomise_conv5(sub, g)
        x = torch.randn(T, 12, 16, 16, generator=g) * 0.5
        yf = blk(x)
        s_f = blk.s.clone()
        jf = blk.jacobian(x)
        yr = blk(x, rev=True)
        s_r = blk.s.clone()
        jr = blk.jacobian(x, rev=True)
        save("g5_invblock_dbnet", x=x.numpy(), y_fwd=yf.numpy(), s_fwd=s_f.numpy(), jac_fwd=jf.numpy(),
             y_rev=yr.numpy(), s_rev=s_r.numpy(), jac_rev=jr.numpy(), **sd_np(blk))

        blk = GA.InvBlockExp(SC.subnet("D2DTNet", "xavier"), 51, 3)
        x = torch.randn(2 * T, 51, 12, 16, generator=g) * 0.5
        yf = blk(x)
        s_f = blk.s.clone()
        jf = blk.jacobian(x)
        yr = blk(x, rev=True)
        s_r = blk.s.clone()
        jr = blk.jacobian(x, rev=True)
        save("g5_invblock_d2dt", x=x.numpy(), y_fwd=yf.numpy(), s_fwd=s_f.numpy(), jac_fwd=jf.numpy(),
             y_rev=yr.numpy(), s_rev=s_r.numpy(), jac_rev=jr.numpy(), **sd_np(blk))

        # ---- G8 SelfC-large stack (seed 10 = yml manual_seed)
        torch.manual_seed(10)
        opt = {"global_module": "nonlocal", "stp_blk_num": 6, "fh_loss": "l2", "scale": 4, "gmm_k": 5}
        net = GA.SelfCInvNet(opt, 3, 3, "D2DTNet", [4, 4], 2).eval()
        x = torch.rand(T, 3, 32, 48, generator=g)
        z, loss_c = net(x=x, rev=False)
        out = z
        for op in reversed(net.operations):
            out = op.forward(out, True)
        ops_sd = {k: v for k, v in sd_np(net).items() if k.startswith("operations.")}
        save("g8_large_stack", x=x.numpy(), z=z.numpy(), loss_c=loss_c.numpy(), x_rev=out.numpy(), **ops_sd)
        # full reverse through STP(l2): pins the caller-visible reverse call
        lrq = Quantization()(z[:, :3])
        xs, hf = net(x=lrq, rev=True)
        stp_sd = {k: v for k, v in sd_np(net).items() if k.startswith("stp_net.")}
        save("g7_stp_l2_full_rev", lr=lrq.numpy(), x_rev=xs.numpy(), hf=hf.numpy(), **stp_sd)

        # ---- G11 one optimize_parameters step (SelfC_model.py:153-176) on the same net (weights: g8_large_stack +
        # g7_stp_l2_full_rev), restated by hand around the reference modules because SelfC_model itself needs cv2:
        # l2 forward fit + l1 (eps 1e-6) reconstruction (loss.py:12-21, train yml :106-107), x 144*144*3, clip 10,
        # Adam(1e-4, (0.9, 0.999), wd 1e-14), sr_bd LR target (SelfC_model.py:128, Guassian.py).
        from models.Guassian import Guassian_downsample as _gd
        from models.lr_scheduler import MultiStepLR_Restart as _MS
        with torch.enable_grad():
            net.train()
            real_h = x
            ref_l = _gd(real_h.transpose(0, 1)).transpose(0, 1)
            prms = [p_ for p_ in net.parameters() if p_.requires_grad]
            optim = torch.optim.Adam(prms, lr=1e-4, weight_decay=1e-14, betas=(0.9, 0.999))
            optim.zero_grad()
            out_f, loss_c = net(x=real_h, rev=False)
            lr_bq = out_f[:, :3]
            l_fit = 1.0 * ((lr_bq - ref_l.detach()) ** 2).mean(-1).mean(-1).mean(-1).mean(-1)
            y_ = Quantization()(lr_bq)
            x_s, _ = net(x=y_, rev=True)
            d_ = real_h - x_s[:, :3]
            l_rec = 1.0 * torch.sqrt(d_ * d_ + 1e-6).mean(-1).mean(-1).mean(-1).mean(-1)
            loss = (l_fit + l_rec + loss_c.mean() * 0) * 144 * 144 * 3
            loss.backward()
            names = [n_ for n_, p_ in net.named_parameters() if p_.requires_grad]
            gnorms = np.array([float(p_.grad.norm()) for p_ in prms], dtype=np.float64)
            total = float(torch.nn.utils.clip_grad_norm_(prms, 10))
            before = [p_.detach().clone() for p_ in prms]
            optim.step()
            dsum = np.array([float((p_.detach() - b_).double().sum()) for p_, b_ in zip(prms, before)], dtype=np.float64)
            sample_g = prms[names.index("operations.1.F.conv1.weight")].grad.detach().numpy().copy()   # after clipping
        save("g11_train_step", ref_l=ref_l.numpy(), l_forw_fit=np.float64(l_fit.item()), l_back_rec=np.float64(l_rec.item()),
             loss=np.float64(loss.item()), grad_norm=np.float64(total), grad_norms=gnorms, names=np.array(names),
             step_delta_sum=dsum, grad_F1_conv1_clipped=sample_g)
        # MultiStepLR_Restart (lr_scheduler.py:8-31): lr trace of a toy schedule incl. a restart
        po = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=1e-4)
        sch = _MS(po, [3, 6, 9], restarts=[5], weights=[0.5], gamma=0.5, clear_state=False)
        trace = []
        for _ in range(12):
            po.step()
            sch.step()
            trace.append(po.param_groups[0]["lr"])
        save("g11_lr_trace", lr=np.array(trace, dtype=np.float64))

        # ---- G8 Haar nets (config C1)
        torch.manual_seed(8)
        irn = IA.InvRescaleNet(3, 3, SC.subnet("DBNet", "xavier"), [1], 1)
        for sub in (irn.operations[1].F, irn.operations[1].G, irn.operations[1].H):
            rerandomise_conv5(sub, g)
        x = torch.rand(T, 3, 64, 64, generator=g)
        lr, hfm = irn(x)
        zfull = irn.operations[1](irn.operations[0](x))
        xinv = irn.operations[0](irn.operations[1](zfull, rev=True), rev=True)
        save("g8_haar_net", x=x.numpy(), lr=lr.numpy(), hf_meansq=hfm.numpy(), z=zfull.numpy(),
             x_rev=xinv.numpy(), **sd_np(irn))

        # ---- G8 Haar-variant SelfCInvNet (model "SelfC", SelfC_arch_inv.py:276-338) with the D2DTNet-conditioned
        # STP v1 (l2 head).  Its forward also runs STP + neg_llh (:312-313).
        import models.modules.SelfC_arch_inv as SA
        torch.manual_seed(18)
        opt1 = {"stp_d2d_inner_c": 32, "stp_temporal_c": 32, "fh_loss": "l2", "gmm_mixture_num": 5,
                "stp_blk_num": 2, "condition_func": "D2DTNet"}
        hnet = SA.SelfCInvNet(opt1, 3, 3, "DBNet", [1], 1).eval()
        for sub in (hnet.operations[1].F, hnet.operations[1].G, hnet.operations[1].H):
            rerandomise_conv5(sub, g)
        x = torch.rand(T, 3, 32, 32, generator=g)
        out, loss = hnet(x=x, rev=False)
        lrq = Quantization()(out[:, :3])
        xr, hf = hnet(x=lrq, rev=True)
        save("g8_selfc_haar", x=x.numpy(), z=out.numpy(), loss_c=loss.numpy(), lr=lrq.numpy(), x_rev=xr.numpy(),
             hf=hf.numpy(), **sd_np(hnet))

        # ---- G8b the same net with the default FeatureCalapseBlock conditioner (space-to-depth + (3,3,3) convs).
        # Its STP has ~16 M parameters: they are not stored but filled by tests/conftest.py:seeded_fill(seed).
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
        from conftest import seeded_fill
        torch.manual_seed(19)
        opt2 = dict(opt1, condition_func="FeatureCalapseBlock")
        fnet = SA.SelfCInvNet(opt2, 3, 3, "DBNet", [1], 1).eval()
        for sub in (fnet.operations[1].F, fnet.operations[1].G, fnet.operations[1].H):
            rerandomise_conv5(sub, g)
        seeded_fill(dict(fnet.stp_net.named_parameters()), 4242)
        x = torch.rand(T, 3, 48, 32, generator=g)             # LR 24x16 -> space-to-depth 6x4
        out, loss = fnet(x=x, rev=False)
        lrq = Quantization()(out[:, :3])
        xr, hf = fnet(x=lrq, rev=True)
        blk1_y = fnet.stp_net.blk1(lrq)
        save("g8_selfc_haar_fcb", x=x.numpy(), z=out.numpy(), loss_c=loss.numpy(), lr=lrq.numpy(), x_rev=xr.numpy(),
             hf=hf.numpy(), blk1_y=blk1_y.numpy(), stp_fill_seed=np.int64(4242),
             **{k: v for k, v in sd_np(fnet).items() if k.startswith("operations.")})

        # ---- G10 Guassian_downsample (models/Guassian.py:7-52): the LR target ref_L of feed_data (SelfC_model.py:128)
        from models.Guassian import Guassian_downsample
        xg = torch.rand(3, 5, 32, 48, generator=g)                 # [C, T, H, W]
        save("g10_gauss", x=xg.numpy(), y=Guassian_downsample(xg).numpy())

        # ---- G12 Y-channel SSIM of test_rescaling.py:110-122 (utils/util.py:361-470 `ssim`, data/util.py:239-245
        # rgb_to_ycbcr).  utils.util / data.util import cv2 and torchvision.utils at module level only for their image
        # I/O helpers: stub those names so the pure-torch metric code can be imported.
        _cv2 = types.ModuleType("cv2")
        _tvu = types.ModuleType("torchvision.utils")
        _tvu.make_grid = None
        sys.modules.setdefault("cv2", _cv2)
        sys.modules.setdefault("imageio", types.ModuleType("imageio"))
        sys.modules["torchvision.utils"] = _tvu
        _tv.utils = _tvu
        import importlib.util as _ilu

        def _load(name, rel):          # the module file itself: the package __init__ pulls in the image datasets (imageio, lmdb)
            spec = _ilu.spec_from_file_location(name, os.path.join(REF, rel))
            m = _ilu.module_from_spec(spec)
            spec.loader.exec_module(m)
            return m

        ref_util = _load("ref_utils_util", "utils/util.py")
        ref_dutil = _load("ref_data_util", "data/util.py")
        g12 = torch.Generator().manual_seed(1212)         # own stream: the fixtures below keep their historical draws
        a_img = torch.rand(3, 3, 40, 56, generator=g12)
        b_img = (a_img + 0.05 * torch.randn(3, 3, 40, 56, generator=g12)).clamp(0, 1)
        ya, yb = ref_dutil.rgb_to_ycbcr(a_img), ref_dutil.rgb_to_ycbcr(b_img)
        ssims = [float(ref_util.ssim(ya[i:i + 1], yb[i:i + 1], data_range=1.0)) for i in range(3)]
        save("g12_ssim_y", a=a_img.numpy(), b=b_img.numpy(), ssim=np.array(ssims, dtype=np.float64))

        # ---- G13 DistIterSampler (data/data_sampler.py:12-59): index streams of a few (dataset size, world, rank, epoch, ratio)
        from data.data_sampler import DistIterSampler as _DIS
        cfgs = [(10, 2, 0, 0, 3), (10, 2, 1, 0, 3), (7, 4, 3, 5, 2), (64, 8, 5, 1, 1)]
        streams = {}
        for i, (n, world, rk, ep, ratio) in enumerate(cfgs):
            smp = _DIS(list(range(n)), num_replicas=world, rank=rk, ratio=ratio)
            smp.set_epoch(ep)
            streams[f"idx{i}"] = np.array(list(iter(smp)), dtype=np.int64)
        save("g13_sampler", cfgs=np.array(cfgs, dtype=np.int64), **streams)

        # ---- G9 Quantization
        q = Quantization()
        v = torch.tensor([-0.3, 0.0, 0.001, 0.00196, 0.00197, 0.5, 0.50196, 0.998, 1.0, 1.7,
                          1.5 / 255, 2.5 / 255, 0.4999 / 255, 254.5 / 255], dtype=torch.float32)
        v = torch.cat([v, torch.rand(200, generator=g) * 1.2 - 0.1])
        save("g9_quant", x=v.numpy(), y=q(v).numpy())

        # ---- G6 GlobalAgg
        torch.manual_seed(6)
        ga = GA.GlobalAgg(64)
        arrs = sd_np(ga)
        for tag, (hh, ww) in {"a": (16, 16), "b": (20, 36)}.items():
            x = torch.randn(T, 64, hh, ww, generator=g)
            arrs[f"{tag}_x"] = x.numpy()
            arrs[f"{tag}_y"] = ga(x).numpy()
        save("g6_globalagg", **arrs)

        # ---- G7 STP v2 raw head (gmm): sampling needs the device RNG in the
        # reference (trap 5), so pin the pre-sampling tensor and the formula with
        # an injected eps (computed here with the reference's own expression).
        torch.manual_seed(7)
        opt = {"global_module": "nonlocal", "stp_blk_num": 6, "fh_loss": "gmm", "scale": 4, "gmm_k": 5}
        stp = GA.STPNet(opt).eval()
        stp.reparametrize = lambda mu, logvar: eps_holder["eps"].mul(torch.exp(logvar)).add_(mu)
        lr = torch.rand(T, 3, 8, 12, generator=g)
        eps = torch.randn(1, 48, 5, T, 8, 12, generator=g)
        eps_holder = {"eps": eps}
        stp(lr.reshape(1, T, 3, 8, 12).transpose(1, 2))
        raw = stp.parameters                          # (1,720,T,8,12)
        v = stp.gmm_v                                 # (1,48,T,8,12)
        save("g7_stp_gmm", lr=lr.numpy(), raw=raw[0].transpose(0, 1).numpy(),
             eps=eps[0].permute(2, 0, 1, 3, 4).numpy(), v=v[0].transpose(0, 1).numpy(), **sd_np(stp))


def dump_state_dict_contract():
    """Checkpoint contract: state_dict keys and shapes of every reference net the drop-in modules
    mirror, plus the init facts the traps depend on (trap 4)."""
    import json
    import models.modules.SelfC_arch_inv as SA
    GlobalVar.set_Temporal_LEN(T)
    torch.manual_seed(0)
    nets = {
        "selfc_large_gmm": GA.SelfCInvNet({"global_module": "nonlocal", "stp_blk_num": 6, "fh_loss": "gmm", "scale": 4, "gmm_k": 5},
                                          3, 3, "D2DTNet", [4, 4], 2),
        "selfc_large_l2": GA.SelfCInvNet({"global_module": "nonlocal", "stp_blk_num": 6, "fh_loss": "l2", "scale": 4, "gmm_k": 5},
                                         3, 3, "D2DTNet", [4, 4], 2),
        "irn_dbnet": IA.InvRescaleNet(3, 3, SC.subnet("DBNet", "xavier"), [2, 1], 2),
        "selfc_haar_d2dt": SA.SelfCInvNet({"stp_d2d_inner_c": 32, "stp_temporal_c": 32, "fh_loss": "l2", "gmm_mixture_num": 5,
                                           "stp_blk_num": 2, "condition_func": "D2DTNet"}, 3, 3, "DBNet", [1], 1),
        "selfc_haar_fcb": SA.SelfCInvNet({"stp_d2d_inner_c": 32, "stp_temporal_c": 32, "fh_loss": "l2", "gmm_mixture_num": 5,
                                          "stp_blk_num": 2, "condition_func": "FeatureCalapseBlock"}, 3, 3, "DBNet", [1], 1),
    }
    out = {}
    for name, net in nets.items():
        sd = net.state_dict()
        out[name] = {"shapes": {k: list(v.shape) for k, v in sd.items()},
                     "n_params": int(sum(p.numel() for p in net.parameters())),
                     "frozen": sorted(k for k, p in net.named_parameters() if not p.requires_grad)}
    db = SC.DenseBlock(9, 3, "xavier")
    d2 = SC.D2DTInput(48, 3, "xavier")
    out["init_facts"] = {"denseblock_conv5_abs_sum": float(db.conv5.weight.abs().sum()),
                         "denseblock_bias_abs_sum": float(sum(getattr(db, f"conv{i}").bias.abs().sum() for i in range(1, 6))),
                         "d2dt_conv5_is_nonzero": bool(d2.conv5.weight.abs().sum() > 0),
                         "denseblock_conv1_std_over_xavier": float(db.conv1.weight.std() / (2.0 / (9 * 9 + 32 * 9)) ** 0.5)}
    with open(os.path.join(OUT, "state_dict_contract.json"), "w") as fh:
        json.dump(out, fh, indent=0, sort_keys=True)
    print("state_dict_contract.json:", {k: v.get("n_params") for k, v in out.items() if k != "init_facts"})


def _stub_io_modules():
    """cv2 / imageio / skvideo / lmdb / thop / torchvision.utils are imported at module level by utils/util.py and the codec
    arch file only for image / video I/O: empty stand-ins let the pure-torch code import (none of their names is called)."""
    class _Any(types.ModuleType):
        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            return None
    for name in ["torchvision.utils", "cv2", "imageio", "skvideo", "skvideo.io", "lmdb", "thop"]:
        sys.modules.setdefault(name, _Any(name))
    _tv.utils = sys.modules["torchvision.utils"]
    _tv.__dict__["__getattr__"] = lambda k: None          # `from torchvision import transforms` etc. (PEP 562)
    sys.modules["skvideo"].io = sys.modules["skvideo.io"]


def main_r2():
    """Round-2 fixtures (own generators: the round-1 files above keep their historical draws and are NOT rewritten).
      g14_irn_rev      InvRescaleNet.forward(rev=True) (Inv_arch.py:115-123) with the torch.rand HF tensor pinned
      g15_codec        codec variant (SelfC_Codec_arch_inv.py): FrequencyAnalyzer(k=2), InvBlockExp(15|3) x 4, narrow STP
                       (hidden 24, growth 12, TEMP_LEN 3), and forward_test's segmenting / tiling restated around the
                       reference's own modules (forward_test itself needs cuda(0) + an ffmpeg writer)
      g16_stp_v1_gmm   STP v1 GMM head (SelfC_arch_inv.py:118-128,151-177) with injected eps
      api_contract.json  public methods + __init__ / forward signatures of every mirrored class
    """
    import inspect
    import json
    _stub_io_modules()
    import models.modules.SelfC_arch_inv as SA
    import models.modules.SelfC_Codec_arch_inv as CA
    import importlib.util as _ilu
    spec = _ilu.spec_from_file_location("ref_utils_util_r2", os.path.join(REF, "utils/util.py"))
    ref_util = _ilu.module_from_spec(spec)
    spec.loader.exec_module(ref_util)
    torch.set_num_threads(8)
    with torch.no_grad():
        # ---- G14
        GlobalVar.set_Temporal_LEN(T)
        g = torch.Generator().manual_seed(1400)
        torch.manual_seed(14)
        irn = IA.InvRescaleNet(3, 3, SC.subnet("DBNet", "xavier"), [1], 1)
        for sub in (irn.operations[1].F, irn.operations[1].G, irn.operations[1].H):
            rerandomise_conv5(sub, g)
        lr = torch.rand(T, 3, 16, 24, generator=g)
        hf45 = torch.rand(T, 45, 16, 24, generator=g)
        real_rand = torch.rand
        torch.rand = lambda *a, **k: hf45.clone()          # the reference draws torch.rand((b,45,h,w)) (:117)
        try:
            out, none = irn(lr, rev=True)
        finally:
            torch.rand = real_rand
        assert none is None
        save("g14_irn_rev", lr=lr.numpy(), hf45=hf45.numpy(), x_rev=out.numpy(), **sd_np(irn))

        # ---- G15 codec variant
        g = torch.Generator().manual_seed(1500)
        torch.manual_seed(15)
        SEG = 3
        GlobalVar.set_Temporal_LEN(SEG)
        fa = CA.FrequencyAnalyzer(3, 2)
        blocks = [CA.InvBlockExp(SC.subnet("D2DTNet", "xavier"), 15, 3) for _ in range(4)]
        ops = torch.nn.ModuleList([fa] + blocks)
        opt = {"global_module": "nonlocal", "stp_blk_num": 4, "fh_loss": "l2", "scale": 2, "gmm_k": 5,
               "stp_hidden_c": 24, "stp_denseblock_innerc": 12}
        stp = CA.STPNet(opt).eval()
        arrs = {f"operations.{k}": v for k, v in sd_np(ops).items()}
        arrs.update({f"stp_net.{k}": v for k, v in sd_np(stp).items()})
        # pieces
        xb = torch.randn(2 * SEG, 15, 10, 12, generator=g) * 0.5
        yb = blocks[0](xb)
        s_f = blocks[0].s.clone()
        xbr = blocks[0](xb, rev=True)
        lr_t = torch.rand(2 * SEG, 3, 8, 12, generator=g)
        stp(lr_t.reshape(2, SEG, 3, 8, 12).transpose(1, 2))
        stp_raw = stp.parameters.transpose(1, 2).reshape(2 * SEG, -1, 8, 12)
        ga = stp.global_m1
        xg = torch.randn(2 * SEG, 24, 8, 12, generator=g)
        yg = ga(xg)
        # whole-path restatement of forward_test (:502-640) around the reference's modules, b = 1, 5 frames -> 2 segments
        t_all = 5
        x = torch.rand(t_all, 3, 32, 48, generator=g)

        def enc(seg):
            o = seg
            for op in ops:
                o = op.forward(o, False)
            return o

        def dec(lr_tile):
            bt, _, hh, ww = lr_tile.shape
            l5 = lr_tile.reshape(bt // SEG, SEG, 3, hh, ww).transpose(1, 2)
            stp(l5)
            o = torch.cat((l5, stp.sample()), 1).transpose(1, 2).reshape(bt, -1, hh, ww)
            for op in reversed(ops):
                o = op.forward(o, True)
            return o
        vid, pad = ref_util.seg_add_pad(x.reshape(1, t_all, 3, 32, 48), SEG)
        outs = []
        for si in range(vid.size(1)):
            seg = vid[:, si].reshape(-1, 3, 32, 48)
            outs.append(torch.cat([enc(seg[:, :, :, i * 24:(i + 1) * 24])[:, 0:3] for i in range(2)], dim=-1))
        enc_lr = ref_util.seg_remove_pad(torch.cat(outs, dim=0).reshape(1, -1, SEG, 3, 16, 24), pad, SEG).reshape(-1, 3, 16, 24)   # :559-563
        lrq = Quantization()(enc_lr)
        vid, pad = ref_util.seg_add_pad(lrq.reshape(1, t_all, 3, 16, 24), SEG)
        outs = []
        for si in range(vid.size(1)):
            seg = vid[:, si].reshape(-1, 3, 16, 24)
            l5 = seg.reshape(1, SEG, 3, 16, 24).transpose(1, 2)                              # b c t h w
            lt = l5.reshape(1, 3, SEG, 2, 8, 2, 12).permute(0, 3, 5, 1, 2, 4, 6).reshape(1, 4, 3, SEG, 8, 12)   # :590-592
            here = []
            for i in range(4):
                tile = lt[:, i].transpose(1, 2).reshape(SEG, 3, 8, 12)
                here.append(dec(tile))
            o = torch.stack(here, dim=1)                                                     # bt p c h w (:608)
            o = o.reshape(1, SEG, 2, 2, 3, 16, 24).permute(0, 4, 1, 2, 5, 3, 6).reshape(1, 3, SEG, 32, 48).transpose(1, 2)   # :612-616
            outs.append(o)
        dec_hr = ref_util.seg_remove_pad(torch.stack(outs, dim=1), pad, SEG).reshape(-1, 3, 32, 48)
        GlobalVar.set_Temporal_LEN(T)
        save("g15_codec", blk_x=xb.numpy(), blk_y=yb.numpy(), blk_s=s_f.numpy(), blk_xrev=xbr.numpy(),
             stp_lr=lr_t.numpy(), stp_raw=stp_raw.numpy(), ga_x=xg.numpy(), ga_y=yg.numpy(),
             x=x.numpy(), enc_lr=enc_lr.numpy(), lr_q=lrq.numpy(), dec_hr=dec_hr.numpy(), **arrs)

        # ---- G16 STP v1 GMM head (CUDA-only in the reference: `.cuda(device)` at :161 - made a no-op for the capture)
        g = torch.Generator().manual_seed(1600)
        torch.manual_seed(16)
        opt1 = {"stp_d2d_inner_c": 32, "stp_temporal_c": 32, "fh_loss": "gmm", "gmm_mixture_num": 5,
                "stp_blk_num": 2, "condition_func": "D2DTNet"}
        s1 = SA.STPNet(opt1).eval()
        lr1 = torch.rand(T, 3, 8, 12, generator=g)
        eps1 = torch.randn(5, 1, 9, T, 8, 12, generator=g)          # one draw per mixture component (:162-163)
        it = iter(range(5))
        s1.reparametrize = lambda mu, logvar: eps1[next(it)].mul(logvar.mul(0.5).exp()).add(mu)     # :179-186 with eps pinned
        real_cuda = torch.Tensor.cuda
        torch.Tensor.cuda = lambda self, *a, **k: self
        try:
            s1(lr1.reshape(1, T, 3, 8, 12).transpose(1, 2))
        finally:
            torch.Tensor.cuda = real_cuda
        save("g16_stp_v1_gmm", lr=lr1.numpy(), eps=eps1[:, 0].numpy(), raw=s1.parameters[0].transpose(0, 1).numpy(),
             v=s1.gmm_v[0].transpose(0, 1).numpy(), **sd_np(s1))

    # ---- G17 STP v2 'gmm_thin' head (ReLU between the head's layers, SelfC_GMM_arch_inv.py:345-354), eps injected as in G7
    with torch.no_grad():
        GlobalVar.set_Temporal_LEN(T)
        g = torch.Generator().manual_seed(1700)
        torch.manual_seed(17)
        st = GA.STPNet({"global_module": "nonlocal", "stp_blk_num": 6, "fh_loss": "gmm_thin", "scale": 4, "gmm_k": 5}).eval()
        with np.load(os.path.join(OUT, "g7_stp_gmm.npz")) as z7:     # the chain in front of the head: G7's weights (not stored twice)
            chain = {k: torch.from_numpy(np.asarray(z7[k])) for k in z7.files if k.split(".")[0] in
                     ("local_m1", "local_m2", "global_m1", "global_m2", "other_stp_modules")}
        missing, unexpected = st.load_state_dict(chain, strict=False)
        assert not unexpected and all(k.startswith("tail_gmm.") for k in missing)
        lr7 = torch.rand(T, 3, 8, 12, generator=g)
        eps7 = torch.randn(1, 48, 5, T, 8, 12, generator=g)
        st.reparametrize = lambda mu, logvar: eps7.mul(torch.exp(logvar)).add_(mu)
        st(lr7.reshape(1, T, 3, 8, 12).transpose(1, 2))
        head = {k: v for k, v in sd_np(st).items() if k.startswith("tail_gmm.")}
        save("g17_stp_gmm_thin", lr=lr7.numpy(), raw=st.parameters[0].transpose(0, 1).numpy(),
             eps=eps7[0].permute(2, 0, 1, 3, 4).numpy(), v=st.gmm_v[0].transpose(0, 1).numpy(), **head)

    # ---- public API contract
    def api(cls):
        pub = sorted(n for n, v in vars(cls).items() if callable(v) and not n.startswith("_"))
        sig = lambda f: [(p.name, None if p.default is inspect._empty else repr(p.default)) for p in inspect.signature(f).parameters.values()]  # noqa: E731
        return {"methods": pub, "init": sig(cls.__init__), "forward": sig(cls.forward) if hasattr(cls, "forward") else None}
    classes = {
        "Inv_arch.InvBlockExp": IA.InvBlockExp, "Inv_arch.HaarDownsampling": IA.HaarDownsampling, "Inv_arch.InvRescaleNet": IA.InvRescaleNet,
        "Subnet_constructor.DenseBlock": SC.DenseBlock, "Subnet_constructor.D2DTInput": SC.D2DTInput,
        "Subnet_constructor.FeatureCalapseBlock": SC.FeatureCalapseBlock, "Subnet_constructor.SpaceToDepth": SC.SpaceToDepth,
        "SelfC_GMM_arch_inv.FrequencyAnalyzer": GA.FrequencyAnalyzer, "SelfC_GMM_arch_inv.PixelUnshuffle": GA.PixelUnshuffle,
        "SelfC_GMM_arch_inv.GlobalAgg": GA.GlobalAgg, "SelfC_GMM_arch_inv.STPNet": GA.STPNet, "SelfC_GMM_arch_inv.SelfCInvNet": GA.SelfCInvNet,
        "SelfC_arch_inv.STPNet": SA.STPNet, "SelfC_arch_inv.SelfCInvNet": SA.SelfCInvNet,
        "SelfC_Codec_arch_inv.FrequencyAnalyzer": CA.FrequencyAnalyzer, "SelfC_Codec_arch_inv.GlobalAgg": CA.GlobalAgg,
        "SelfC_Codec_arch_inv.STPNet": CA.STPNet,
        "Quantization.Quantization": Quantization, "global_var.GlobalVar": GlobalVar,
    }
    with open(os.path.join(OUT, "api_contract.json"), "w") as fh:
        json.dump({k: api(v) for k, v in classes.items()}, fh, indent=0, sort_keys=True)
    print("api_contract.json:", len(classes), "classes")


if __name__ == "__main__":
    if "r2" in sys.argv[1:]:
        main_r2()
    else:
        main()
        dump_state_dict_contract()
