"""SelfC-large (model "SelfC_GMM") on MI355X: FrequencyAnalyzer + N x InvBlockExp + STP.

Mirrors codes/models/modules/SelfC_GMM_arch_inv.py: PixelUnshuffle (:46-60),
FrequencyAnalyzer (:62-82), GlobalAgg (:257-285), STPNet (:289-430), SelfCInvNet
(:432-494).  ``SelfCInvNet.forward`` keeps the whole op loop in the kernels'
latent layout: one split kernel, one selfc_invstack_run, one merge kernel.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib, runtime as rt
from ..global_var import GlobalVar
from .Inv_arch import InvBlockExp  # noqa: F401  (same class, as in the reference's three copies)
from .Subnet_constructor import D2DTInput, subnet


class PixelUnshuffle(nn.Module):
    """(N,C,H,W) -> (N,C*S*S,H/S,W/S) with out channel (sy*S+sx)*C + c (:51-60); a pure view shuffle."""

    def __init__(self, scale=4):
        super().__init__()
        self.scale = scale

    def forward(self, x):
        n, c, h, w = x.size()
        s = self.scale
        x = x.view(n, c, h // s, s, w // s, s).permute(0, 3, 5, 1, 2, 4).contiguous()
        return x.view(n, c * s * s, h // s, w // s)


class FrequencyAnalyzer(nn.Module):
    """lo = 4x4 block mean, hi = PixelUnshuffle(x - up(lo)); reverse = up(lo) + nn.PixelShuffle(hi)
    with ITS channel order, which is not the forward's inverse (:73-82, SURVEY trap 3)."""

    def __init__(self, channel_in, k=4):
        super().__init__()
        self.k = k
        self.channel_in = channel_in

    def forward(self, x, rev=False):
        x = rt.as_input(x)
        k, sp = self.k, _lib.stream_ptr()
        c2 = 3 * k * k
        if not rev:
            n, c, h, w = x.shape
            if c != 3 or h % k or w % k:
                raise RuntimeError(f"FrequencyAnalyzer forward expects (N,3,{k}a,{k}b), got {tuple(x.shape)}")
            ws = rt.workspace(x.device, rt.SUBNET_D2DT, n, 1, h // k, w // k, 3, c2)
            rt.call("selfc_freq_fwd", x.data_ptr(), ws.x1.data_ptr(), ws.x2.data_ptr(), None, ws.FC, n, h, w, k, sp)
            return rt.latent_to_nchw(ws)
        n, c, h, w = x.shape
        if c != 3 + c2:
            raise RuntimeError(f"FrequencyAnalyzer reverse expects {3 + c2} channels, got {c}")
        ws = rt.workspace(x.device, rt.SUBNET_D2DT, n, 1, h, w, 3, c2)
        rt.nchw_to_latent(x, ws, with_fd=False)
        y = torch.empty((n, 3, h * k, w * k), dtype=torch.float32, device=x.device)
        rt.call("selfc_freq_inv", ws.x1.data_ptr(), ws.x2.data_ptr(), y.data_ptr(), n, h, w, k, sp)
        return y


class GlobalAgg(nn.Module):
    """Temporal 7x7 attention over globally pooled descriptors (:257-285).
    NOTE (round 1): parameters/state_dict match the reference; the arithmetic still
    runs as stock torch ops on the GPU - native kernels are the next STP milestone."""

    def __init__(self, c):
        super().__init__()
        self.fc = nn.Linear(32 * 32, 1)
        self.proj1 = nn.Conv2d(c, c, 1, 1, 0)
        self.proj2 = nn.Linear(c, c)
        self.proj3 = nn.Linear(c, c)

    def forward(self, x):
        t = GlobalVar.get_Temporal_LEN()
        bt, c, h, w = x.size()
        b = bt // t
        p1 = self.proj1(x)
        g = self.fc(F.adaptive_avg_pool2d(x, (32, 32)).reshape(bt, c, 32 * 32)).squeeze(-1).reshape(b, t, c)
        a = F.softmax(torch.matmul(self.proj2(g), self.proj3(g).transpose(1, 2)) / c, dim=-1)
        v = p1.reshape(b, t, c, h, w).permute(0, 2, 3, 4, 1).reshape(b, c * h * w, t)
        mixed = torch.matmul(v, a).reshape(b, c, h, w, t).permute(0, 4, 1, 2, 3).reshape(bt, c, h, w)
        return x + mixed


class STPNet(nn.Module):
    """Self-conditioned latent predictor v2 (:289-430): 6 x [D2DTInput + GlobalAgg] + 1x1x1 MLP
    head; GMM sample v = sum_k pi*(eps*exp(clamp(logsigma,-7,7)) + mu) with pi = softmax over the
    hf_dim axis (trap 6).  The D2DTInput subnets run on the HIP kernels; GlobalAgg / head are
    torch ops this round.  ``eps`` can be injected (``self.eps``) for reproducible parity."""

    def __init__(self, opt):
        super().__init__()
        self.global_module = opt["global_module"]
        self.fh_loss = opt["fh_loss"]
        self.scale = opt["scale"]
        self.K = opt["gmm_k"]
        self.stp_blk_num = opt["stp_blk_num"] - 2
        c = 64
        if self.global_module not in (None, 'nonlocal'):
            raise NotImplementedError("selfc_amd covers global_module: nonlocal (the shipped configs); "
                                      "the deform aggregators need torchvision.ops.deform_conv2d")
        self.local_m1 = D2DTInput(3, c, INN_init=False)
        self.local_m2 = D2DTInput(c, c, INN_init=False)
        if self.global_module == 'nonlocal':
            self.global_m1 = GlobalAgg(c)
            self.global_m2 = GlobalAgg(c)
        others = []
        for _ in range(self.stp_blk_num):
            others.append(D2DTInput(c, c, INN_init=False))
            if self.global_module == 'nonlocal':
                others.append(GlobalAgg(c))
        self.other_stp_modules = nn.Sequential(*others)
        self.hf_dim = 3 * (self.scale ** 2)
        lre = lambda: nn.LeakyReLU(negative_slope=0.2, inplace=True)  # noqa: E731
        if self.fh_loss == "l2":
            self.tail_gmm = nn.Sequential(lre(), nn.Conv3d(c, self.hf_dim, 1, 1, 0, bias=True))
        elif self.fh_loss == "gmm":
            self.tail_gmm = nn.Sequential(lre(), nn.Conv3d(c, c * 2, 1, 1, 0, bias=True),
                                          lre(), nn.Conv3d(c * 2, c * 4, 1, 1, 0, bias=True),
                                          lre(), nn.Conv3d(c * 4, self.hf_dim * self.K * 3, 1, 1, 0, bias=True))
        elif self.fh_loss == "gmm_thin":
            self.tail_gmm = nn.Sequential(lre(), nn.Conv3d(c, c, 1, 1, 0, bias=True),
                                          nn.ReLU(inplace=True), nn.Conv3d(c, c, 1, 1, 0, bias=True),
                                          nn.ReLU(inplace=True), nn.Conv3d(c, self.hf_dim * self.K * 3, 1, 1, 0, bias=True))
        self.eps = None   # optional injected noise (b, hf_dim, K, t, h, w)

    def forward(self, x):
        b, c, t, h, w = x.size()
        temp = x.transpose(1, 2).reshape(b * t, c, h, w)
        temp = self.local_m1(temp)
        if self.global_module:
            temp = self.global_m1(temp)
        temp = self.local_m2(temp)
        if self.global_module:
            temp = self.global_m2(temp)
        temp = self.other_stp_modules(temp)
        bt, c, hh, ww = temp.size()
        t = GlobalVar.get_Temporal_LEN()
        b = bt // t
        temp = temp.reshape(b, t, c, hh, ww).transpose(1, 2)
        # the reference stores this as `self.parameters` (shadowing nn.Module.parameters, :377);
        # kept under a non-clashing name here
        self.stp_parameters = self.tail_gmm(temp)
        if self.fh_loss == "l2":
            return
        p = self.stp_parameters.reshape(b, self.hf_dim, self.K, 3, t, hh, ww)
        pi = F.softmax(p[:, :, :, 0], dim=1)
        log_scale = torch.clamp(p[:, :, :, 1], -7, 7)
        mean = p[:, :, :, 2]
        self.gmm_v = (pi * self.reparametrize(mean, log_scale)).sum(2)

    def reparametrize(self, mu, logvar):
        std = torch.exp(logvar)
        eps = self.eps if self.eps is not None else torch.randn_like(std)
        return eps.mul(std).add_(mu)

    def neg_llh(self, hf):
        if self.fh_loss == "l2":
            return torch.mean((hf - self.stp_parameters) ** 2)
        b, c, t, h, w = hf.size()
        p = self.stp_parameters.reshape(b, self.hf_dim, self.K, 3, t, h, w).permute(0, 1, 4, 5, 6, 2, 3).reshape(-1, self.K, 3)
        # index convention of the likelihood differs from sampling (:399-405): mean = idx1, log-sigma = idx2
        mix = torch.distributions.Categorical(F.softmax(p[:, :, 0], dim=1))
        comp = torch.distributions.Normal(p[:, :, 1], torch.exp(torch.clamp(p[:, :, 2], -7, 7)))
        return -torch.distributions.MixtureSameFamily(mix, comp).log_prob(hf.reshape(-1))

    def sample(self):
        return self.stp_parameters if self.fh_loss == "l2" else self.gmm_v


class SelfCInvNet(nn.Module):
    """SelfC-large: [FrequencyAnalyzer] + sum(block_num) x InvBlockExp(51|3) + STPNet (:432-494)."""

    def __init__(self, opt, channel_in, channel_out, subnet_type, block_num, down_num):
        super().__init__()
        operations = [FrequencyAnalyzer(channel_in)]
        current_channel = channel_in * 17
        sc = subnet(subnet_type, "xavier")
        for i in range(down_num):
            for _ in range(block_num[i]):
                operations.append(InvBlockExp(sc, current_channel, channel_out))
        self.operations = nn.ModuleList(operations)
        self.stp_net = STPNet(opt)

    # -- fused latent-layout pipeline ---------------------------------------------------
    def _blocks(self):
        return [op for op in self.operations if isinstance(op, InvBlockExp)]

    def _stack(self):
        blocks = self._blocks()
        key = tuple(rt.params_key(b) for b in blocks)
        if getattr(self, "_stack_key", None) != key:
            self._stack_arr, self._stack_keep = rt.block_array(blocks)
            self._stack_key = key
        return self._stack_arr, len(blocks)

    def _workspace(self, x, n, h, w):
        t = GlobalVar.get_Temporal_LEN()
        if not t or n % t:
            raise RuntimeError(f"GlobalVar temporal length {t!r} does not divide the {n} input frames")
        blk = self._blocks()[0]
        return rt.workspace(x.device, blk.F.kind, n, t, h, w, blk.split_len1, blk.split_len2)

    def forward(self, x, rev=False, cal_jacobian=False, lr_before_distor=None):
        x = rt.as_input(x)
        rt.no_autograd_guard(x, *[p for b in self._blocks() for p in b.parameters()])
        sp = _lib.stream_ptr()
        arr, nblk = self._stack()
        k = self.operations[0].k
        if not rev:
            n, c, H, W = x.shape
            if c != 3 or H % k or W % k:
                raise RuntimeError(f"SelfCInvNet forward expects (N,3,{k}a,{k}b), got {tuple(x.shape)}")
            ws = self._workspace(x, n, H // k, W // k)
            rt.call("selfc_freq_fwd", x.data_ptr(), ws.x1.data_ptr(), ws.x2.data_ptr(), ws.fd.data_ptr(), ws.FC, n, H, W, k, sp)
            lat = ws.latent()
            rt.call("selfc_invstack_run", arr, nblk, lat, 0, sp)
            out = rt.latent_to_nchw(ws)
            return out, out.new_zeros(())          # loss_c = out.mean()*0 (:466)
        n, c, h, w = x.shape
        t = GlobalVar.get_Temporal_LEN()
        b = n // t
        lr_input = x[:, 0:3].reshape(b, t, 3, h, w).transpose(1, 2)
        self.stp_net(lr_input)
        recon_hf = self.stp_net.sample().transpose(1, 2).reshape(b * t, -1, h, w)
        ws = self._workspace(x, n, h, w)
        rt.nchw_to_latent(torch.cat((x[:, 0:3], recon_hf), dim=1).contiguous(), ws, with_fd=False)
        lat = ws.latent()
        rt.call("selfc_invstack_run", arr, nblk, lat, 1, sp)
        out = torch.empty((n, 3, h * k, w * k), dtype=torch.float32, device=x.device)
        rt.call("selfc_freq_inv", ws.x1.data_ptr(), ws.x2.data_ptr(), out.data_ptr(), n, h, w, k, sp)
        return out, recon_hf

    def inverse_from_latent(self, z):
        """Reversed op loop (:486-489) on a full (N,51,h,w) latent, STP bypassed - the
        'inv' half of BASELINE.json's metric."""
        z = rt.as_input(z)
        n, c, h, w = z.shape
        arr, nblk = self._stack()
        ws = self._workspace(z, n, h, w)
        sp = _lib.stream_ptr()
        k = self.operations[0].k
        rt.nchw_to_latent(z, ws, with_fd=False)
        lat = ws.latent()
        rt.call("selfc_invstack_run", arr, nblk, lat, 1, sp)
        out = torch.empty((n, 3, h * k, w * k), dtype=torch.float32, device=z.device)
        rt.call("selfc_freq_inv", ws.x1.data_ptr(), ws.x2.data_ptr(), out.data_ptr(), n, h, w, k, sp)
        return out
