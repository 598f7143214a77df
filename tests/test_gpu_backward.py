"""Gradients of the HIP training path (csrc/backward.hip through the C ABI and selfc_amd/autograd.py) against
torch autograd through the CPU oracle on the same seeded inputs.

Two bars.  (1) Kernel exactness: with the LeakyReLU masks and layer inputs frozen to the f16 features the HIP
forward saved, autograd through the oracle's convs is exactly what the kernels must produce; the only difference
left is the f16 rounding of gradient operands -> max|a-b|/max|b| < 3e-3.  (2) End to end against the fp32 oracle
the two forwards differ by f16 operand rounding, so a pre-activation within ~1e-3 of zero can land on the other
side of the LeakyReLU kink and its gradient changes by 5x at that pixel: a max-norm is meaningless there, the bar
is the relative L2 error < 3e-2.  Index shuffles (FrequencyAnalyzer) are fp32: 1e-6."""
import pytest
import torch

from conftest import load_golden, rel_err, subdict
from oracle import selfc_oracle as O

pytestmark = pytest.mark.gpu
STRICT = 3e-3
L2TOL = 3e-2
T = 7


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from selfc_amd import _lib
    _lib.lib()
    from selfc_amd import GlobalVar
    GlobalVar.set_Temporal_LEN(T)
    return torch.device("cuda:0")


def _oracle_grads(fn, params, x, gy):
    """autograd through the oracle: returns (y, dx, {name: dparam})."""
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    xr = x.clone().requires_grad_(True)
    y = fn(p, xr)
    y.backward(gy)
    return y.detach(), xr.grad, {k: v.grad for k, v in p.items()}


def _check_module_grads(mod, ref_grads, prefix="", tol=L2TOL, metric=rel_l2, scale=1.0):
    worst = 0.0
    for name, prm in mod.named_parameters():
        key = prefix + name
        if key not in ref_grads or ref_grads[key] is None:
            continue
        assert prm.grad is not None, f"no gradient for {name}"
        e = metric(prm.grad.cpu(), scale * ref_grads[key])
        assert e < tol, f"{name}: {e}"
        worst = max(worst, e)
    return worst


def _frozen_subnet(p, x, saved, temporal):
    """The oracle's dense block with LeakyReLU masks and layer inputs frozen to the saved f16 features: values are the
    saved ones, gradients flow through the convs (what csrc/backward.hip computes)."""
    import torch.nn.functional as F
    feats = [x]
    for k in range(1, 5):
        w = p[f"conv{k}.weight"]
        pre = F.conv2d(torch.cat(feats, 1), w[:, :, 0] if w.dim() == 5 else w, p[f"conv{k}.bias"], 1, 1)
        f = pre * torch.where(saved[k - 1] > 0, 1.0, 0.2)
        feats.append(saved[k - 1] + (f - f.detach()))
    d = torch.cat(feats, 1)
    if not temporal:
        return F.conv2d(d, p["conv5.weight"], p["conv5.bias"], 1, 1)
    n, c, h, w_ = d.shape
    d5 = d.reshape(n // T, T, c, h, w_).transpose(1, 2)
    return F.conv3d(d5, p["conv5.weight"], p["conv5.bias"], 1, (1, 0, 0)).transpose(1, 2).reshape(n, -1, h, w_)


@pytest.mark.parametrize("cls,ci,co,hw", [("D2DTInput", 48, 3, (12, 20)), ("D2DTInput", 3, 48, (12, 20)),
                                          ("DenseBlock", 9, 3, (16, 16)), ("DenseBlock", 3, 9, (16, 16)),
                                          ("D2DTInput", 3, 64, (36, 36)), ("D2DTInput", 64, 64, (20, 12)),
                                          ("D2DTInput", 48, 3, (64, 112)), ("DenseBlock", 12, 3, (18, 50))])
def test_subnet_backward(dev, cls, ci, co, hw):
    from selfc_amd.modules import Subnet_constructor as S
    torch.manual_seed(5)
    h, w = hw
    n = T * 2 if cls == "D2DTInput" else 3
    m = getattr(S, cls)(ci, co, "xavier")
    with torch.no_grad():                       # non-trivial conv5 / biases (the 2-D block zero-inits conv5)
        for prm in m.parameters():
            prm.copy_(torch.randn_like(prm) * (0.05 if prm.dim() > 1 else 0.1))
    params = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x = torch.randn(n, ci, h, w)
    gy = torch.randn(n, co, h, w) * 0.01
    fn = (lambda p, xx: O.d2dt(p, xx, T)) if cls == "D2DTInput" else (lambda p, xx: O.dense_block(p, xx))
    y_ref, dx_ref, g_ref = _oracle_grads(fn, params, x, gy)
    m.to(dev)
    xd = x.to(dev).requires_grad_(True)
    y = m(xd)
    assert rel_err(y.detach().cpu(), y_ref) < 1e-3
    dense = y.grad_fn.dense.float().cpu()                       # [planes][N][H][W][32]
    y.backward(gy.to(dev))
    # (2) end to end against the fp32 oracle
    assert rel_l2(xd.grad.cpu(), dx_ref) < L2TOL
    _check_module_grads(m, g_ref)
    # (1) kernel exactness with frozen masks / inputs
    nx = 0 if ci <= 3 else (ci + 31) // 32
    saved = [dense[nx + k].permute(0, 3, 1, 2) for k in range(4)]
    xs = x if ci <= 3 else torch.cat([dense[i] for i in range(nx)], -1)[..., :ci].permute(0, 3, 1, 2).contiguous()
    _, dx_fz, g_fz = _oracle_grads(lambda p, xx: _frozen_subnet(p, xx, saved, cls == "D2DTInput"), params, xs, gy)
    assert rel_err(xd.grad.cpu(), dx_fz) < STRICT
    _check_module_grads(m, g_fz, tol=STRICT, metric=rel_err)


@pytest.mark.parametrize("name,kind,cnum", [("g5_invblock_d2dt", "D2DTNet", 51), ("g5_invblock_dbnet", "DBNet", 12)])
@pytest.mark.parametrize("rev", [False, True])
def test_invblock_backward(dev, name, kind, cnum, rev):
    from selfc_amd.modules.Inv_arch import InvBlockExp
    from selfc_amd.modules.Subnet_constructor import subnet
    g = load_golden(name)
    torch.manual_seed(11)
    blk = InvBlockExp(subnet(kind, "xavier"), cnum, 3)
    sd = {k: v for k, v in g.items() if k[:2] in ("F.", "G.", "H.")}
    blk.load_state_dict(sd, strict=True)
    x = g["x"]
    gy = torch.randn_like(x) * 0.02
    y_ref, dx_ref, g_ref = _oracle_grads(lambda p, xx: O.invblock(kind, p, xx, 3, T, rev=rev)[0], sd, x, gy)
    blk.to(dev)
    xd = x.to(dev).requires_grad_(True)
    y = blk(xd, rev=rev)
    assert rel_err(y.detach().cpu(), y_ref) < 1e-3
    y.backward(gy.to(dev))
    assert rel_l2(xd.grad.cpu(), dx_ref) < L2TOL
    _check_module_grads(blk, g_ref)
    # parameters only (input does not require grad), twice: gradients accumulate like stock autograd
    first = {n_: p_.grad.clone() for n_, p_ in blk.named_parameters()}
    blk.zero_grad()
    for _ in range(2):
        blk(x.to(dev), rev=rev).backward(gy.to(dev))
    for nme, prm in blk.named_parameters():
        assert rel_err(prm.grad, 2 * first[nme]) < 1e-5, nme


@pytest.mark.parametrize("rev", [False, True])
def test_freq_backward(dev, rev):
    from selfc_amd.modules.SelfC_GMM_arch_inv import FrequencyAnalyzer
    torch.manual_seed(2)
    fa = FrequencyAnalyzer(3)
    x = torch.randn(3, 51, 6, 10) if rev else torch.randn(3, 3, 24, 40)
    xr = x.clone().requires_grad_(True)
    y_ref = O.freq_inv(xr) if rev else O.freq_fwd(xr)
    gy = torch.randn_like(y_ref)
    y_ref.backward(gy)
    xd = x.to(dev).requires_grad_(True)
    y = fa(xd, rev=rev)
    assert torch.equal(y.detach().cpu(), y_ref.detach())
    y.backward(gy.to(dev))
    assert rel_err(xd.grad.cpu(), xr.grad) < 1e-6


def test_stack_backward_chain(dev):
    """FrequencyAnalyzer + 2 InvBlockExp forward, quantise (identity gradient), the same blocks reversed: the
    gradient of an l2 + l1 loss w.r.t. every weight, as SelfCModel.optimize_parameters composes it
    (SelfC_model.py:153-171) minus the STP sampling."""
    from selfc_amd.modules.Inv_arch import InvBlockExp
    from selfc_amd.modules.Quantization import Quantization
    from selfc_amd.modules.SelfC_GMM_arch_inv import FrequencyAnalyzer
    from selfc_amd.modules.Subnet_constructor import subnet
    torch.manual_seed(3)
    blocks = torch.nn.ModuleList([InvBlockExp(subnet("D2DTNet", "xavier"), 51, 3) for _ in range(2)])
    sd = {k: v.detach().clone() for k, v in blocks.state_dict().items()}
    x = torch.rand(T, 3, 48, 32)
    ref_l = torch.rand(T, 3, 12, 8)
    noise = torch.randn(T, 48, 12, 8) * 0.05       # stands in for the STP sample: keeps |rec - x| away from the l1 kink

    def loss_fn(fwd_blk, quant, freq_f, freq_r, xx):
        z = freq_f(xx)
        for i in range(2):
            z = fwd_blk(i, z, False)
        lr = z[:, :3]
        l_fit = ((lr - ref_l.to(xx.device)) ** 2).mean()
        zq = torch.cat((quant(lr), z[:, 3:] * 0.7 + noise.to(xx.device)), 1)
        for i in (1, 0):
            zq = fwd_blk(i, zq, True)
        rec = freq_r(zq)
        d = rec - xx
        l_rec = torch.sqrt(d * d + 1e-6).mean()
        return (l_fit + l_rec) * 144 * 144 * 3

    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}

    class STE(torch.autograd.Function):
        @staticmethod
        def forward(ctx, v):
            return O.quantize(v)

        @staticmethod
        def backward(ctx, gg):
            return gg

    loss_ref = loss_fn(lambda i, z, rev: O.invblock("D2DTNet", {k[len(f"{i}."):]: v for k, v in p.items() if k.startswith(f"{i}.")}, z, 3, T, rev=rev)[0],
                       STE.apply, O.freq_fwd, O.freq_inv, x)
    loss_ref.backward()
    blocks.to(dev)
    fa, q = FrequencyAnalyzer(3), Quantization()
    loss = loss_fn(lambda i, z, rev: blocks[i](z, rev=rev), q, lambda v: fa(v), lambda v: fa(v, rev=True), x.to(dev))
    assert abs(float(loss.detach()) - float(loss_ref.detach())) < 2e-3 * abs(float(loss_ref.detach()))
    loss.backward()
    gref = {k: v.grad for k, v in p.items()}
    errs = {n_: rel_l2(p_.grad.cpu(), gref[n_]) for n_, p_ in blocks.named_parameters()}
    print("worst", sorted(errs.items(), key=lambda kv: -kv[1])[:5])
    # Two bars.  (1) All 60 tensors together: ||g - g_ref|| / ||g_ref|| over the concatenated gradients - the quantity an
    # optimizer step sees.  (2) Per tensor 5e-2: the worst single tensor sits at ~4.4e-2 on this 12x8 latent, and that figure is
    # a kink lottery, not kernel error - the f16 forward puts a handful of pre-activations (|v| < ~1e-3) on the other side of
    # LeakyReLU's kink than the fp32 oracle, each flip changes that pixel's contribution to a 32-channel gradient row by a factor
    # 5, and a bias gradient of a late conv sums only 7 x 96 pixels.  With the masks frozen to the forward's own
    # (test_subnet_backward, bar (1): <= 3e-3 max-norm) and against the reference's whole training
    # step on 36x36 latents (G11, test_gpu_train: gradient norms 1.7e-4) the same kernels are two orders tighter.
    from conftest import record
    num = sum(float((p_.grad.cpu().double() - gref[n_].double()).pow(2).sum()) for n_, p_ in blocks.named_parameters())
    den = sum(float(gref[n_].double().pow(2).sum()) for n_, _ in blocks.named_parameters())
    total = (num / den) ** 0.5
    record("stack backward chain (2 blocks fwd + rev), relative L2 over ALL parameter gradients", total)
    record("stack backward chain, worst single tensor relative L2 (f16-forward LeakyReLU kink lottery)", max(errs.values()))
    assert total < 1e-2, total          # measured 3.4e-3
    assert max(errs.values()) < 5e-2, sorted(errs.items(), key=lambda kv: -kv[1])[:5]


OPT = {"global_module": "nonlocal", "stp_blk_num": 6, "fh_loss": "l2", "scale": 4, "gmm_k": 5}
STP_KEYS = ("local_m1", "local_m2", "global_m1", "global_m2", "other_stp_modules", "tail_gmm")


def test_globalagg_backward(dev):
    """No kinks in GlobalAgg: the gradient is smooth, so both a max-norm and an L2 bar apply."""
    from selfc_amd.modules.SelfC_GMM_arch_inv import GlobalAgg
    g = load_golden("g6_globalagg")
    sd = {k: v for k, v in g.items() if k.split(".")[0] in ("fc", "proj1", "proj2", "proj3")}
    for tag in ("a", "b"):
        ga = GlobalAgg(64)
        ga.load_state_dict(sd, strict=True)
        x = g[f"{tag}_x"]
        torch.manual_seed(4)
        gy = torch.randn_like(x) * 0.01
        y_ref, dx_ref, g_ref = _oracle_grads(lambda p, xx: O.global_agg(p, xx, T), sd, x, gy)
        ga.to(dev)
        xd = x.to(dev).requires_grad_(True)
        y = ga(xd)
        assert rel_err(y.detach().cpu(), y_ref) < 1e-3
        y.backward(gy.to(dev))
        assert rel_err(xd.grad.cpu(), dx_ref) < 5e-3
        for name, prm in ga.named_parameters():
            if name == "proj3.bias":
                # softmax over the key axis is invariant to a shift of every key: the true gradient is exactly zero
                assert float(prm.grad.abs().max()) < 1e-3 * float(ga.proj2.bias.grad.abs().max())
                continue
            e = rel_l2(prm.grad.cpu(), g_ref[name])
            assert e < 1e-2, f"{tag} {name}: {e}"


@pytest.mark.parametrize("fh_loss", ["gmm", "l2"])
def test_stp_backward(dev, fh_loss):
    """STPNet chain + head + (GMM) sample as one differentiable op vs autograd through the oracle."""
    from selfc_amd.modules.SelfC_GMM_arch_inv import STPNet
    g = load_golden("g7_stp_gmm" if fh_loss == "gmm" else "g7_stp_l2_full_rev")
    prefix = "" if fh_loss == "gmm" else "stp_net."
    sd = {k[len(prefix):]: v for k, v in g.items() if k.startswith(prefix) and k[len(prefix):].split(".")[0] in STP_KEYS}
    stp = STPNet(dict(OPT, fh_loss=fh_loss))
    stp.load_state_dict(sd, strict=True)
    torch.manual_seed(8)
    h, w = 8, 12
    lr = torch.rand(T, 3, h, w)
    eps = torch.randn(T, 48, 5, h, w)
    gy = torch.randn(T, 48, h, w) * 0.01

    def fn(p, xx):
        raw = O.stp_v2_parameters(p, xx, T)
        return O.stp_v2_gmm_sample(raw, eps) if fh_loss == "gmm" else raw

    y_ref, dx_ref, g_ref = _oracle_grads(fn, sd, lr, gy)
    stp.to(dev)
    stp.eps = eps.permute(1, 2, 0, 3, 4).unsqueeze(0).to(dev)            # (1,48,5,T,h,w)
    xd = lr.to(dev).requires_grad_(True)
    stp(xd.reshape(1, T, 3, h, w).transpose(1, 2))
    v = stp.sample()[0].transpose(0, 1)                                   # (T,48,h,w)
    assert rel_err(v.detach().cpu(), y_ref) < 3e-3
    v.backward(gy.to(dev))
    # proj3.bias has an exactly-zero gradient (key shift invariance of the softmax); fc.bias is one scalar obtained
    # by heavy cancellation, so it is judged together with fc.weight (the same Linear layer) as one vector
    named = dict(stp.named_parameters())
    errs = {}
    for n_, p_ in named.items():
        if n_.endswith("proj3.bias") or n_.endswith("fc.bias"):
            continue
        a, b_ = p_.grad.cpu().reshape(-1), g_ref[n_].reshape(-1)
        if n_.endswith("fc.weight"):
            nb = n_[:-len("weight")] + "bias"
            a, b_ = torch.cat((a, named[nb].grad.cpu().reshape(-1))), torch.cat((b_, g_ref[nb].reshape(-1)))
        errs[n_] = rel_l2(a, b_)
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:6]
    print(fh_loss, "dx", rel_l2(xd.grad.cpu(), dx_ref), "worst", worst)
    assert rel_l2(xd.grad.cpu(), dx_ref) < 5e-2
    assert worst[0][1] < 5e-2, worst


def test_haar_backward_and_irn_training(dev):
    """HaarDownsampling adjoints, then InvRescaleNet (Haar + DBNet blocks, Inv_arch.py:87-127) end to end: gradients
    of its training-style outputs against autograd through the oracle."""
    from selfc_amd.modules.Inv_arch import HaarDownsampling, InvRescaleNet
    from selfc_amd.modules.Subnet_constructor import subnet
    torch.manual_seed(6)
    hd = HaarDownsampling(3)
    for rev, shape in ((False, (2, 3, 8, 12)), (True, (2, 12, 4, 6))):
        x = torch.randn(*shape)
        xr = x.clone().requires_grad_(True)
        y_ref = O.haar_inv(xr) if rev else O.haar_fwd(xr)
        gy = torch.randn_like(y_ref)
        y_ref.backward(gy)
        xd = x.to(dev).requires_grad_(True)
        y = hd.to(dev)(xd, rev=rev)
        assert torch.equal(y.detach().cpu(), y_ref.detach())
        y.backward(gy.to(dev))
        assert rel_err(xd.grad.cpu(), xr.grad) < 1e-6
    g = load_golden("g8_haar_net")
    irn = InvRescaleNet(3, 3, subnet("DBNet", "xavier"), [1], 1)
    sd = {k: v for k, v in g.items() if k.startswith("operations.")}
    irn.load_state_dict(sd, strict=True)
    x = g["x"]
    p = {k: v.clone().requires_grad_(k.endswith(("weight", "bias"))) for k, v in sd.items()}
    z = O.haar_net_fwd(p, x, [1], T, kind="DBNet")
    target = torch.rand(T, 3, 32, 32)
    loss_ref = ((z[:, :3] - target) ** 2).mean() + (z[:, 3:] ** 2).mean()
    loss_ref.backward()
    irn.to(dev)
    lr, hfm = irn(x.to(dev))
    loss = ((lr - target.to(dev)) ** 2).mean() + hfm
    assert abs(float(loss.detach()) - float(loss_ref.detach())) < 1e-3 * float(loss_ref.detach())
    loss.backward()
    for name, prm in irn.named_parameters():
        if prm.requires_grad:
            e = rel_l2(prm.grad.cpu(), p[name].grad)
            assert e < L2TOL, f"{name}: {e}"


def test_ddp_single_rank_training_step(dev):
    """The reference wraps netG in DistributedDataParallel (SelfC_model.py:42).  World size 1 over RCCL on this GPU:
    the autograd.Functions must cooperate with DDP's hooks (every parameter receives its gradient exactly once)."""
    import os
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel
    from selfc_amd import train
    from selfc_amd.modules.SelfC_GMM_arch_inv import SelfCInvNet
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    own = not dist.is_initialized()
    if own:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        torch.manual_seed(10)
        opt = {"global_module": "nonlocal", "stp_blk_num": 6, "fh_loss": "gmm", "scale": 4, "gmm_k": 5}
        net = SelfCInvNet(opt, 3, 3, "D2DTNet", [4, 4], 2).to(dev)
        ref = {n: p.detach().clone() for n, p in net.named_parameters()}
        ddp = DistributedDataParallel(net, device_ids=[dev.index], find_unused_parameters=False)
        tr = train.RescaleTrainer(ddp, dict(train.TRAIN_OPT_LARGE))
        gt = torch.rand(1, 3, T, 32, 48, generator=torch.Generator().manual_seed(1)).to(dev)
        real_h, ref_l, _ = train.feed_data(gt, "sr_bd", 4)
        log = tr.optimize_parameters(real_h, ref_l)
        assert log["loss"] == log["loss"] and log["loss"] > 0
        # GlobalAgg.proj3.bias has an exactly-zero gradient (softmax is invariant to a shift of every key), so Adam
        # leaves those six tensors where they are; everything else must have moved
        still = [n for n, p in net.named_parameters() if torch.equal(p.detach(), ref[n])]
        assert all(n.endswith("proj3.bias") for n in still), still
        assert len(still) <= 6 and len(ref) == 354
    finally:
        if own:
            dist.destroy_process_group()


def test_gradient_scale_invariance(dev):
    """Gradients travel through the MFMA as f16 scaled by a power of two taken from max|dOut| on the device: the result must
    not depend on the magnitude of the incoming gradient (1e-8 would underflow f16 entirely, 1e5 would overflow it)."""
    from selfc_amd.modules.Inv_arch import InvBlockExp
    from selfc_amd.modules.Subnet_constructor import subnet
    g = load_golden("g5_invblock_d2dt")
    blk = InvBlockExp(subnet("D2DTNet", "xavier"), 51, 3)
    blk.load_state_dict({k: v for k, v in g.items() if k[:2] in ("F.", "G.", "H.")}, strict=True)
    blk.to(dev)
    x = g["x"].to(dev)
    torch.manual_seed(3)
    gy = torch.randn_like(x)
    base = None
    for scale in (1.0, 1e-8, 1e5):
        blk.zero_grad()
        xd = x.clone().requires_grad_(True)
        blk(xd).backward(gy * scale)
        cur = [xd.grad / scale] + [p.grad / scale for p in blk.parameters()]
        assert all(torch.isfinite(c).all() for c in cur)
        if base is None:
            base = cur
        else:
            for a, b in zip(cur, base):
                assert rel_err(a, b) < 2e-3, scale


def test_selfc_haar_variant_training_gradients(dev):
    """model "SelfC" (Haar + InvBlockExp(DBNet) + STP v1 with the D2DTNet conditioner and l2 head, SelfC_arch_inv.py): the
    training-style objective fit(LR) + neg_llh + reconstruction through rev, every gradient on the HIP path."""
    from selfc_amd.modules.SelfC_arch_inv import SelfCInvNet
    g = load_golden("g8_selfc_haar")
    opt1 = {"stp_d2d_inner_c": 32, "stp_temporal_c": 32, "fh_loss": "l2", "gmm_mixture_num": 5,
            "stp_blk_num": 2, "condition_func": "D2DTNet"}
    sd = {k: v for k, v in g.items() if k.startswith(("operations.", "stp_net."))}
    net = SelfCInvNet(opt1, 3, 3, "DBNet", [1], 1)
    net.load_state_dict(sd, strict=True)
    x = g["x"]
    target = torch.rand(x.shape[0], 3, x.shape[2] // 2, x.shape[3] // 2, generator=torch.Generator().manual_seed(12))
    noise = torch.randn(x.shape[0], 3, x.shape[2] // 2, x.shape[3] // 2, generator=torch.Generator().manual_seed(13)) * 0.02

    def objective(fwd, rev, xx, tgt, nz):
        z, nll = fwd(xx)
        lr = z[:, :3]
        rec, _ = rev(lr + nz)                      # a perturbed LR keeps |rec - x| away from the l1 kink
        d = rec - xx
        return ((lr - tgt) ** 2).mean() + nll + torch.sqrt(d * d + 1e-6).mean()

    p = {k: v.clone().requires_grad_(k.endswith(("weight", "bias"))) for k, v in sd.items()}
    loss_ref = objective(lambda v: O.selfc_haar_fwd(p, v, [1], T), lambda v: O.selfc_haar_rev(p, v, [1], T), x, target, noise)
    loss_ref.backward()
    net.to(dev)
    loss = objective(lambda v: net(x=v, rev=False), lambda v: net(x=v, rev=True), x.to(dev), target.to(dev), noise.to(dev))
    assert abs(float(loss.detach()) - float(loss_ref.detach())) < 2e-3 * abs(float(loss_ref.detach()))
    loss.backward()
    errs = {n_: rel_l2(p_.grad.cpu(), p[n_].grad) for n_, p_ in net.named_parameters() if p_.requires_grad}
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    assert worst[0][1] < 5e-2, worst
    # the default FeatureCalapseBlock conditioner trains too since round 2 (autograd.FCBFn; parity: test_feature_calapse_block_backward)
    opt2 = dict(opt1, condition_func="FeatureCalapseBlock")
    net2 = SelfCInvNet(opt2, 3, 3, "DBNet", [1], 1).to(dev)
    z2, nll2 = net2(x=x.to(dev), rev=False)
    (z2[:, :3].mean() + nll2).backward()
    missing = [n_ for n_, p_ in net2.named_parameters() if p_.requires_grad and p_.grad is None]
    assert not missing, missing


@pytest.mark.parametrize("t", [1, 3])
def test_d2dt_backward_other_clip_lengths(dev, t):
    """io_type='3d' call with clips of 1 and 3 frames: the temporal taps of conv5 and of its weight gradient fall outside the
    clip (zero padding per clip, not per batch)."""
    from selfc_amd.modules.Subnet_constructor import D2DTInput
    torch.manual_seed(21)
    m = D2DTInput(3, 48)
    with torch.no_grad():
        for prm in m.parameters():
            prm.copy_(torch.randn_like(prm) * (0.05 if prm.dim() > 1 else 0.1))
    params = {k: v.detach().clone() for k, v in m.state_dict().items()}
    b, h, w = 3, 12, 20
    x = torch.randn(b * t, 3, h, w)
    gy = torch.randn(b * t, 48, h, w) * 0.01
    y_ref, dx_ref, g_ref = _oracle_grads(lambda p, xx: O.d2dt(p, xx, t), params, x, gy)
    m.to(dev)
    x5 = x.reshape(b, t, 3, h, w).transpose(1, 2).to(dev).requires_grad_(True)
    y5 = m(x5, io_type="3d")
    y = y5.transpose(1, 2).reshape(b * t, 48, h, w)
    assert rel_err(y.detach().cpu(), y_ref) < 1e-3
    y.backward(gy.to(dev))
    dx = x5.grad.transpose(1, 2).reshape(b * t, 3, h, w)
    assert rel_l2(dx.cpu(), dx_ref) < L2TOL
    _check_module_grads(m, g_ref)


@pytest.mark.parametrize("hw", [(5, 7), (9, 13), (17, 33), (4, 52)])
def test_invblock_backward_ragged_sizes(dev, hw):
    """Odd latent sizes: partial 16x16 / 12x16 conv tiles, partial 4x4 weight-gradient patches, single-tile frames."""
    from selfc_amd.modules.Inv_arch import InvBlockExp
    from selfc_amd.modules.Subnet_constructor import subnet
    g = load_golden("g5_invblock_d2dt")
    sd = {k: v for k, v in g.items() if k[:2] in ("F.", "G.", "H.")}
    blk = InvBlockExp(subnet("D2DTNet", "xavier"), 51, 3)
    blk.load_state_dict(sd, strict=True)
    blk.to(dev)
    torch.manual_seed(hw[0] * 100 + hw[1])
    x = torch.randn(T, 51, *hw) * 0.5
    gy = torch.randn_like(x) * 0.02
    for rev in (False, True):
        y_ref, dx_ref, g_ref = _oracle_grads(lambda p, xx: O.invblock("D2DTNet", p, xx, 3, T, rev=rev)[0], sd, x, gy)
        blk.zero_grad()
        xd = x.to(dev).requires_grad_(True)
        y = blk(xd, rev=rev)
        assert rel_err(y.detach().cpu(), y_ref) < 1e-3
        y.backward(gy.to(dev))
        assert rel_l2(xd.grad.cpu(), dx_ref) < 5e-2
        _check_module_grads(blk, g_ref, tol=6e-2)


@pytest.mark.parametrize("ci,co,hw,scale", [(3, 12, (16, 24), 4), (12, 8, (8, 16), 4)])
def test_feature_calapse_block_backward(dev, ci, co, hw, scale):
    """FeatureCalapseBlock (Subnet_constructor.py:280-324: space-to-depth, (3,3,3) conv1 / conv5, gc = 128) - the default
    conditioner of STP v1: gradients of the input and of all ten parameters against autograd through the oracle.
    End to end vs the fp32 oracle the bar is the relative L2 error (f16 forward, LeakyReLU kinks); the (3,3,3) temporal
    taps are covered by both clips (clip-boundary zero padding in the weight gradient)."""
    from selfc_amd.modules import Subnet_constructor as S
    torch.manual_seed(7)
    h, w = hw
    n = 2 * T
    m = S.FeatureCalapseBlock(ci, co, scale)
    with torch.no_grad():
        for prm in m.parameters():
            prm.copy_(torch.randn_like(prm) * ((1.0 / prm[0].numel()) ** 0.5 if prm.dim() > 1 else 0.05))
    params = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x = torch.randn(n, ci, h, w)
    gy = torch.randn(n, co, h, w) * 0.01
    y_ref, dx_ref, g_ref = _oracle_grads(lambda p, xx: O.feature_calapse_block(p, xx, T, scale), params, x, gy)
    m.to(dev)
    xd = x.to(dev).requires_grad_(True)
    y = m(xd)
    assert rel_err(y.detach().cpu(), y_ref) < 1e-3
    y.backward(gy.to(dev))
    assert rel_l2(xd.grad.cpu(), dx_ref) < L2TOL
    _check_module_grads(m, g_ref)
    # parameters only, accumulating like stock autograd
    first = {n_: p_.grad.clone() for n_, p_ in m.named_parameters()}
    m(x.to(dev)).backward(gy.to(dev))
    for n_, p_ in m.named_parameters():
        assert rel_err(p_.grad, 2 * first[n_]) < 1e-5, n_


def test_stp_v1_default_conditioner_trains(dev):
    """STP v1 with its default FeatureCalapseBlock conditioner (SelfC_arch_inv.py:107-108) and the l2 head is differentiable
    end to end: neg_llh's gradient reaches every parameter and matches the oracle in relative L2."""
    from selfc_amd.modules.SelfC_arch_inv import STPNet
    torch.manual_seed(9)
    opt = {"stp_d2d_inner_c": 32, "stp_temporal_c": 32, "fh_loss": "l2", "gmm_mixture_num": 5, "stp_blk_num": 2, "condition_func": "FeatureCalapseBlock"}
    stp = STPNet(opt)
    with torch.no_grad():
        for prm in stp.parameters():
            prm.copy_(torch.randn_like(prm) * ((1.0 / prm[0].numel()) ** 0.5 if prm.dim() > 1 else 0.05))
    params = {k: v.detach().clone() for k, v in stp.state_dict().items()}
    lr = torch.rand(T, 3, 16, 16)
    hf = torch.randn(T, 9, 16, 16) * 0.1
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    loss_ref = torch.mean((hf - O.stp_v1_parameters(p, lr, T)) ** 2)
    loss_ref.backward()
    stp.to(dev).train()
    stp(lr.to(dev).reshape(1, T, 3, 16, 16).transpose(1, 2))
    loss = stp.neg_llh(hf.to(dev).reshape(1, T, 9, 16, 16).transpose(1, 2))
    assert abs(loss.item() - loss_ref.item()) < 1e-3 * abs(loss_ref.item()) + 1e-6
    loss.backward()
    import torch.nn as nn
    for name, prm in nn.Module.named_parameters(stp):
        assert prm.grad is not None, name
        assert rel_l2(prm.grad.cpu(), p[name].grad) < 5e-2, name


# ---- round 3: the shapes that were inference-only (narrow dense blocks / GlobalAgg / STP via zero-padded shadows, ReLU head, STP v1 GMM head)

CODEC_OPT = {"global_module": "nonlocal", "stp_blk_num": 4, "fh_loss": "l2", "scale": 2, "gmm_k": 5,
             "stp_hidden_c": 24, "stp_denseblock_innerc": 12}


def _all_tensor_rel_l2(mod, g_ref):
    """||g - g_ref|| / ||g_ref|| over the concatenation of every parameter gradient: what an optimizer step sees"""
    num = sum(float((p_.grad.cpu().double() - g_ref[n_].double()).pow(2).sum()) for n_, p_ in mod.named_parameters())
    den = sum(float(g_ref[n_].double().pow(2).sum()) for n_, _ in mod.named_parameters())
    return (num / den) ** 0.5


class _f16_forward_oracle:
    """Inside: the oracle's convs compute on f16-ROUNDED operands (values as the HIP forward sees them; the cast's gradient is the
    identity), so its LeakyReLU / ReLU masks fall where the HIP path's fall.  A per-tensor gradient error that shrinks against THIS
    reference is kink-dominated (a pre-activation within ~1e-3 of zero on the other side of the kink), not a kernel error."""

    def __enter__(self):
        import torch.nn.functional as F
        self.F, self.orig = F, (F.conv2d, F.conv3d)
        r16 = lambda t_: t_.half().float()      # noqa: E731
        c2, c3 = self.orig
        F.conv2d = lambda x, w, *a, **k: c2(r16(x), r16(w), *a, **k)
        F.conv3d = lambda x, w, *a, **k: c3(r16(x), r16(w), *a, **k)
        return self

    def __exit__(self, *exc):
        self.F.conv2d, self.F.conv3d = self.orig


def _bar_table(tag, mod, errs, errs16, g_ref):
    """per tensor (worst six): share of the whole gradient's norm, relative L2 error against the fp32 oracle and against the
    f16-forward oracle - appended to gpurun_out/gradient_bars.txt (copied to profiles/rN/) and logged with the parity figures"""
    import os
    from conftest import ROOT, record
    tot = sum(float(v.double().pow(2).sum()) for v in g_ref.values()) ** 0.5
    rows = []
    for n_, e in sorted(errs.items(), key=lambda kv: -kv[1])[:6]:
        share = float(g_ref[n_].double().norm()) / tot
        rows.append((n_, share, e, errs16.get(n_, float("nan"))))
        record(f"{tag}: {n_} gradient rel L2 vs fp32 oracle (norm share {share:.1e}; vs f16-forward oracle {errs16.get(n_, float('nan')):.1e})", e)
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "gradient_bars.txt"), "a") as fh:
            fh.write(f"{tag}\n  {'tensor':44s} {'share of |grad|':>16s} {'err vs fp32 oracle':>20s} {'err vs f16-forward oracle':>26s}\n")
            for n_, share, e, e16 in rows:
                fh.write(f"  {n_:44s} {share:16.2e} {e:20.2e} {e16:26.2e}\n")
    except OSError:
        pass
    return rows


def _floored(errs, g_ref, mod, floor=1e-5):
    """the same per-tensor errors with the denominator floored at `floor` x the WHOLE gradient's norm: a tensor that carries 1e-7 of the
    gradient (a GlobalAgg's fc / proj vectors: sums of cancelling terms) is judged against what it could move, not against its own noise"""
    named = dict(mod.named_parameters())
    tot = sum(float(v.double().pow(2).sum()) for v in g_ref.values()) ** 0.5
    out = {}
    for n_, e in errs.items():
        nb = float(g_ref[n_].double().norm())
        if n_.endswith("fc.weight"):
            nb = (nb ** 2 + float(g_ref[n_[:-len("weight")] + "bias"].double().norm()) ** 2) ** 0.5
        out[n_] = e * nb / (nb + floor * tot)
    return out


def _fc_merged_errs(mod, g_ref, skip=("proj3.bias",)):
    """per-tensor relative L2 with fc.weight / fc.bias of a GlobalAgg judged as one vector (fc.bias is a single scalar obtained
    by heavy cancellation) and proj3.bias skipped (exactly zero: the softmax is invariant to a key shift)"""
    named = dict(mod.named_parameters())
    errs = {}
    for n_, p_ in named.items():
        if n_.endswith(skip) or n_.endswith("fc.bias"):
            continue
        assert p_.grad is not None, n_
        a, b_ = p_.grad.cpu().reshape(-1), g_ref[n_].reshape(-1)
        if n_.endswith("fc.weight"):
            nb = n_[:-len("weight")] + "bias"
            a, b_ = torch.cat((a, named[nb].grad.cpu().reshape(-1))), torch.cat((b_, g_ref[nb].reshape(-1)))
        errs[n_] = rel_l2(a, b_)
    return errs


@pytest.mark.parametrize("cls,ci,co,gc", [("D2DTInput", 24, 24, 12), ("D2DTInput", 3, 24, 12), ("DenseBlock", 9, 3, 16)])
def test_narrow_growth_dense_block_backward(dev, cls, ci, co, gc):
    """Dense blocks with growth < 32 (the codec variant's STP: stp_denseblock_innerc = 12, SelfC_Codec_arch_inv.py:248-252) train
    through the exactly equivalent growth-32 block (selfc_amd/shadow.py): gradients against autograd through the oracle."""
    from selfc_amd.modules import Subnet_constructor as SC
    torch.manual_seed(11)
    mod = getattr(SC, cls)(ci, co, gc=gc, INN_init=False)
    sd = {k: v.detach().clone() for k, v in mod.state_dict().items()}
    x = torch.randn(T, ci, 12, 8) * 0.5
    gy = torch.randn(T, co, 12, 8) * 0.02
    fn = (lambda p, xx: O.d2dt(p, xx, T)) if cls == "D2DTInput" else (lambda p, xx: O.dense_block(p, xx))
    y_ref, dx_ref, g_ref = _oracle_grads(fn, sd, x, gy)
    mod.to(dev)
    xd = x.to(dev).requires_grad_(True)
    y = mod(xd)
    assert rel_err(y.detach().cpu(), y_ref) < 1e-3
    y.backward(gy.to(dev))
    assert rel_l2(xd.grad.cpu(), dx_ref) < L2TOL
    # per tensor 5e-2 (the end-to-end bar of the chained tests): a 12-element bias gradient on a 12x8 frame is the f16-forward
    # LeakyReLU kink lottery at its noisiest (see test_stack_backward_chain); all tensors together stay far below
    worst = _check_module_grads(mod, g_ref, tol=5e-2)
    tot = (sum(float((p_.grad.cpu().double() - g_ref[n_].double()).pow(2).sum()) for n_, p_ in mod.named_parameters()) /
           sum(float(g_ref[n_].double().pow(2).sum()) for n_, _ in mod.named_parameters())) ** 0.5
    print(cls, ci, co, gc, "worst parameter-gradient rel L2", worst, "all tensors", tot)
    assert tot < 1.5e-2, tot
    # a second call accumulates into .grad, and the shadow follows an optimizer step (weights changed -> re-synced)
    with torch.no_grad():
        for p_ in mod.parameters():
            p_.mul_(1.01)
    mod.zero_grad()
    mod(xd).backward(gy.to(dev))
    _, _, g2 = _oracle_grads(fn, {k: v * 1.01 for k, v in sd.items()}, x, gy)
    _check_module_grads(mod, g2, tol=5e-2)


def test_narrow_globalagg_backward(dev):
    """GlobalAgg(24) over 3-frame clips (codec variant, SelfC_Codec_arch_inv.py:103-131): the 64-channel shadow with the softmax
    temperature 1/24 folded into proj2."""
    from selfc_amd.modules.SelfC_Codec_arch_inv import GlobalAgg
    g = load_golden("g15_codec")
    sd = subdict(g, "stp_net.global_m1")
    ga = GlobalAgg(24)
    ga.load_state_dict(sd, strict=True)
    x = g["ga_x"]
    torch.manual_seed(4)
    gy = torch.randn_like(x) * 0.01
    y_ref, dx_ref, g_ref = _oracle_grads(lambda p, xx: O.global_agg(p, xx, 3), sd, x, gy)
    ga.to(dev)
    xd = x.to(dev).requires_grad_(True)
    y = ga(xd)
    assert rel_err(y.detach().cpu(), y_ref) < 1e-3 and rel_err(y.detach().cpu(), g["ga_y"]) < 1e-3
    y.backward(gy.to(dev))
    assert rel_err(xd.grad.cpu(), dx_ref) < 5e-3
    errs = _fc_merged_errs(ga, g_ref)
    print("GlobalAgg(24) worst", sorted(errs.items(), key=lambda kv: -kv[1])[:4])
    assert max(errs.values()) < L2TOL, errs


@pytest.mark.parametrize("hw", [(8, 12), (24, 32)])
def test_codec_stp_trains(dev, hw):
    """The codec variant's STP (hidden 24, growth 12, l2 head of 12 channels, 3-frame clips; SelfC_Codec_arch_inv.py:234-312,
    which the reference trains through its H.265 surrogate) under autograd: every parameter gradient against the oracle.  Two
    sizes: the fixture's 8x12 frames and 24x32 ones - the error of a correct backward is the f16-forward kink lottery and falls
    with the number of pixels a gradient sums over; a wrong placement in the shadow would not."""
    from selfc_amd.modules.SelfC_Codec_arch_inv import STPNet
    g = load_golden("g15_codec")
    sd = subdict(g, "stp_net")
    stp = STPNet(CODEC_OPT)
    stp.load_state_dict(sd, strict=True)
    h, w = hw
    lr = g["stp_lr"] if hw == (8, 12) else torch.rand(6, 3, h, w, generator=torch.Generator().manual_seed(21))      # 2 clips of 3 frames
    torch.manual_seed(9)
    gy = torch.randn(6, 12, h, w) * 0.01
    y_ref, dx_ref, g_ref = _oracle_grads(lambda p, xx: O.codec_stp_parameters(p, xx, 3, 4), sd, lr, gy)
    stp.to(dev)
    xd = lr.to(dev).requires_grad_(True)
    stp(xd.reshape(2, 3, 3, h, w).transpose(1, 2))
    v = stp.sample().transpose(1, 2).reshape(6, 12, h, w)
    assert rel_err(v.detach().cpu(), y_ref) < 1e-3
    if hw == (8, 12):
        assert rel_err(v.detach().cpu(), g["stp_raw"]) < 1e-3
    v.backward(gy.to(dev))
    assert rel_l2(xd.grad.cpu(), dx_ref) < 5e-2
    errs = _fc_merged_errs(stp, g_ref)
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    tot = _all_tensor_rel_l2(stp, g_ref)
    print("codec STP", hw, "worst", worst, "all tensors", tot)
    with _f16_forward_oracle():
        _, _, g16 = _oracle_grads(lambda p, xx: O.codec_stp_parameters(p, xx, 3, 4), sd, lr, gy)
    _bar_table(f"codec STP (24 / 12) backward at {h}x{w}", stp, errs, _fc_merged_errs(stp, g16), g_ref)
    from conftest import record
    record(f"codec STP (24 / 12) backward at {h}x{w}, relative L2 over all parameter gradients", tot)
    # 24-channel rows / 12-channel growth on 2 clips of 3 frames: the per-tensor figures of the small vectors (a 12-element
    # bias, the 1024-tap pooled map fc.weight whose gradient is a sum of cancelling terms) are the f16-forward kink lottery;
    # the bar that binds is the one over all tensors
    assert tot < (2.5e-2 if hw == (8, 12) else 1.2e-2), tot
    # per tensor: measured worst 2.2e-2 (8x12) / 2.6e-2 (24x32), a third to a fifth of it left against the f16-forward oracle, i.e. mostly
    # kinks (profiles/r6/gradient_bars.txt); bar = 3 x measured
    assert worst[0][1] < 8e-2, worst
    assert len(list(stp.parameters())) == len(sd)         # `parameters` (the head output) is still callable


def test_stp_v2_gmm_thin_backward(dev):
    """fh_loss 'gmm_thin' (SelfC_GMM_arch_inv.py:345-354): ReLU between the head's layers - its mask in the HIP backward."""
    from selfc_amd.modules.SelfC_GMM_arch_inv import STPNet
    g7, g = load_golden("g7_stp_gmm"), load_golden("g17_stp_gmm_thin")
    sd = {k: v for k, v in g7.items() if k.split(".")[0] in ("local_m1", "local_m2", "global_m1", "global_m2", "other_stp_modules")}
    sd.update({k: v for k, v in g.items() if k.startswith("tail_gmm.")})
    stp = STPNet(dict(OPT, fh_loss="gmm_thin"))
    stp.load_state_dict(sd, strict=True)
    lr, eps = g["lr"], g["eps"]                           # (T,3,8,12), (T,48,5,8,12)
    torch.manual_seed(5)
    gy = torch.randn(T, 48, 8, 12) * 0.01
    y_ref, dx_ref, g_ref = _oracle_grads(lambda p, xx: O.stp_v2_gmm_sample(O.stp_v2_parameters(p, xx, T, thin=True), eps), sd, lr, gy)
    stp.to(dev)
    stp.eps = eps.permute(1, 2, 0, 3, 4).unsqueeze(0).to(dev)
    xd = lr.to(dev).requires_grad_(True)
    stp(xd.reshape(1, T, 3, 8, 12).transpose(1, 2))
    v = stp.sample()[0].transpose(0, 1)
    assert rel_err(v.detach().cpu(), y_ref) < 3e-3
    v.backward(gy.to(dev))
    errs = _fc_merged_errs(stp, g_ref)
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    tot = _all_tensor_rel_l2(stp, g_ref)
    print("gmm_thin dx", rel_l2(xd.grad.cpu(), dx_ref), "worst", worst, "all tensors", tot)
    with _f16_forward_oracle():
        _, _, g16 = _oracle_grads(lambda p, xx: O.stp_v2_gmm_sample(O.stp_v2_parameters(p, xx, T, thin=True), eps), sd, lr, gy)
    _bar_table("STP v2 gmm_thin backward at 8x12", stp, errs, _fc_merged_errs(stp, g16), g_ref)
    # ReLU kinks are hard zeros: a hidden unit the f16 forward puts on the other side loses (or gains) its whole contribution,
    # which shows in the ill-conditioned small vectors of the LAST GlobalAgg (fc.weight: a sum of cancelling terms); all tensors
    # together and dx are what bind
    assert rel_l2(xd.grad.cpu(), dx_ref) < 5e-2 and tot < 2e-2, (tot, worst)
    # per tensor (profiles/r6/gradient_bars.txt): the five worst (6.8e-2 ... 1.3e-1) each carry < 3e-7 of the gradient's norm and are no
    # better against the f16-forward oracle - cancellation noise of vectors that are sums of cancelling terms, not kinks; judged against
    # a floor of 1e-5 of the whole gradient they vanish, and every tensor with a share above 1e-5 is within 2.4e-2.  Bars = 3 x measured
    fl = _floored(errs, g_ref, stp)
    from conftest import record
    record("STP v2 gmm_thin backward: worst per-tensor relative L2 with the denominator floored at 1e-5 of the whole gradient", max(fl.values()))
    assert max(fl.values()) < 7.5e-2, sorted(fl.items(), key=lambda kv: -kv[1])[:4]
    assert worst[0][1] < 4e-1, worst


def test_stp_v1_gmm_head_trains(dev):
    """STP v1 with fh_loss gmm (SelfC_arch_inv.py:118-128,151-177): the three-layer head (HeadFn) and the reparameterised sample
    (GmmSampleFn, std = exp(0.5 ls)) are differentiable; `parameters` feeds the likelihood, `gmm_v` the reverse pass."""
    from selfc_amd.modules.SelfC_arch_inv import STPNet
    g = load_golden("g16_stp_v1_gmm")
    opt = {"stp_d2d_inner_c": 32, "stp_temporal_c": 32, "fh_loss": "gmm", "gmm_mixture_num": 5, "stp_blk_num": 2, "condition_func": "D2DTNet"}
    sd = {k: v for k, v in g.items() if k.split(".")[0] in ("blk1", "blk2", "tail_gmm")}
    stp = STPNet(opt)
    stp.load_state_dict(sd, strict=True)
    lr, eps = g["lr"], g["eps"]                           # (T,3,8,12), (K,9,T,8,12)
    torch.manual_seed(6)
    gv = torch.randn(T, 9, 8, 12) * 0.01
    graw = torch.randn(T, 135, 8, 12) * 0.001

    def fn(p, xx):                                        # both consumers of the head at once: sample + raw parameters
        raw = O.stp_v1_parameters(p, xx, T)
        return torch.cat((O.stp_v1_gmm_sample(raw, eps), raw), 1)
    y_ref, dx_ref, g_ref = _oracle_grads(fn, sd, lr, torch.cat((gv, graw), 1))
    stp.to(dev)
    stp.eps = eps.unsqueeze(1).to(dev)                    # (K, b=1, 9, T, h, w)
    xd = lr.to(dev).requires_grad_(True)
    stp(xd.reshape(1, T, 3, 8, 12).transpose(1, 2))
    v = stp.sample()[0].transpose(0, 1)
    raw = stp.parameters[0].transpose(0, 1)
    assert rel_err(v.detach().cpu(), g["v"]) < 1e-3 and rel_err(raw.detach().cpu(), g["raw"]) < 1e-3
    ((v * gv.to(dev)).sum() + (raw * graw.to(dev)).sum()).backward()
    assert rel_l2(xd.grad.cpu(), dx_ref) < 5e-2
    worst = _check_module_grads(stp, g_ref, tol=5e-2)
    print("STP v1 gmm worst parameter-gradient rel L2", worst)
    assert torch.isfinite(stp.neg_llh(stp.sample().detach())).all()


@pytest.mark.parametrize("kind,cnum,split", [("DBNet", 12, 6), ("D2DTNet", 20, 8), ("D2DTNet", 51, 4), ("D2DTNet", 67, 3)])
@pytest.mark.parametrize("rev", [False, True])
def test_invblock_wide_split(dev, kind, cnum, split, rev):
    """InvBlockExp with channel_split_num > 3 (Inv_arch.py:12-13 takes any split) or, for D2DTNet, an x2 wider than 48: composed from stand-alone subnets and the
    stand-alone coupling pass - output, s, jacobian, exact-inverse property and every gradient against the oracle."""
    from selfc_amd.modules.Inv_arch import InvBlockExp
    from selfc_amd.modules.Subnet_constructor import subnet
    torch.manual_seed(17)
    blk = InvBlockExp(subnet(kind, "xavier"), cnum, split)
    with torch.no_grad():                     # DBNet zero-initialises conv5: give the coupling something to do
        for sub in (blk.F, blk.G, blk.H):
            sub.conv5.weight.normal_(0, 0.02)
            sub.conv5.bias.normal_(0, 0.02)
    sd = {k: v.detach().clone() for k, v in blk.state_dict().items()}
    x = torch.randn(T, cnum, 12, 8) * 0.5
    gy = torch.randn_like(x) * 0.02
    (y_ref, s_ref) = O.invblock(kind, sd, x, split, T, rev=rev)
    _, dx_ref, g_ref = _oracle_grads(lambda p, xx: O.invblock(kind, p, xx, split, T, rev=rev)[0], sd, x, gy)
    blk.to(dev)
    with torch.no_grad():
        y0 = blk(x.to(dev), rev=rev)
        assert rel_err(y0.cpu(), y_ref) < 1e-3 and rel_err(blk.s.cpu(), s_ref) < 1e-3
        jac = blk.jacobian(x.to(dev), rev=rev)
        assert abs(float(jac) - float(O.invblock_jacobian(s_ref, T, rev))) < 1e-3 * (abs(float(O.invblock_jacobian(s_ref, T, rev))) + 1)
        back = blk(y0, rev=not rev)           # the coupling is exactly invertible
        assert rel_err(back.cpu(), x) < 1e-3
    xd = x.to(dev).requires_grad_(True)
    y = blk(xd, rev=rev)
    y.backward(gy.to(dev))
    assert rel_l2(xd.grad.cpu(), dx_ref) < L2TOL
    worst = _check_module_grads(blk, g_ref, tol=5e-2)
    tot = _all_tensor_rel_l2(blk, g_ref)
    print(kind, cnum, split, rev, "worst", worst, "all tensors", tot)
    assert tot < 1.5e-2, tot


def test_rowsum_accum_kernel(dev):
    """selfc_rowsum_accum: dst_i = beta_i * dst_i + column sums of up to eight small row-major matrices in one launch (the per-clip
    GlobalAgg gradients summed straight into the flat gradient buffer) against torch."""
    import ctypes as C
    from selfc_amd import _lib, runtime as rt
    g = torch.Generator().manual_seed(21)
    shapes = [(1, 4096), (8, 64), (8, 4096), (8, 64), (8, 4096), (8, 64), (8, 1), (8, 1296)]
    betas = [1.0, 1.0, 0.0, 1.0, 1.0, 0.0, 1.0, 0.0]
    srcs = [torch.randn(s, generator=g).to(dev) for s in shapes]
    dsts = [torch.randn(s[1], generator=g).to(dev) for s in shapes]
    want = [b * d + s.sum(0) for s, d, b in zip(srcs, dsts, betas)]
    job = _lib.RowSum()
    for i, (s, d, b) in enumerate(zip(srcs, dsts, betas)):
        job.src[i], job.dst[i], job.len[i], job.rows[i], job.beta[i] = s.data_ptr(), d.data_ptr(), s.shape[1], s.shape[0], b
    job.n = len(shapes)
    rt.call("selfc_rowsum_accum", C.byref(job), _lib.stream_ptr())
    for d, w_ in zip(dsts, want):
        assert torch.allclose(d, w_, rtol=1e-5, atol=1e-5)
    job.n = 9
    assert _lib.lib().selfc_rowsum_accum(C.byref(job), _lib.stream_ptr()) == -1          # SELFC_EINVAL, nothing launched
