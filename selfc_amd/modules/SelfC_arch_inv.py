"""Haar-variant SelfC net (model "SelfC") on MI355X.

Mirrors codes/models/modules/SelfC_arch_inv.py: ``STPNet`` v1 (:90-198) and ``SelfCInvNet`` (:276-338) =
[HaarDownsampling, block_num[i] x InvBlockExp] per level + STP; ``hf_dim`` is hard-coded to 9 there, so the
net only works for one Haar level (scale 2).  The reference's temporal length for this file is the module
constant ``TEMP_LEN = 7`` (:6); here GlobalVar is used when set, else 7.

Both conditioners are built: ``condition_func: "D2DTNet"`` (a chain of D2DTInput subnets) and the default
``FeatureCalapseBlock`` pair (space-to-depth + (3,3,3) Conv3d with gc=128, Subnet_constructor.py:280-324), with both
heads: ``fh_loss: "l2"`` (:110-116) and ``"gmm"`` (:118-128,151-177: three pointwise layers on pwconv_kernel, then the
sampler with this file's ``std = exp(0.5 logvar)``, selfc_gmm_sample_generic).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib, runtime as rt
from ..global_var import GlobalVar
from .module_util import HeadOutput, cache_free_state
from .Inv_arch import HaarDownsampling, InvBlockExp
from .Subnet_constructor import D2DTInput, FeatureCalapseBlock, subnet

TEMP_LEN = 7


def _tlen():
    return GlobalVar.get_Temporal_LEN() or TEMP_LEN


class STPNet(nn.Module):
    __getstate__ = cache_free_state      # deepcopy / pickle leave the runtime's caches behind (module_util)
    def __init__(self, opt):
        super().__init__()
        self.stp_d2d_inner_c = opt["stp_d2d_inner_c"]
        self.stp_temporal_c = opt["stp_temporal_c"]
        self.fh_loss = opt["fh_loss"]
        self.K = opt["gmm_mixture_num"]
        self.stp_blk_num = opt["stp_blk_num"]
        self.condition_func = opt["condition_func"]
        if self.fh_loss not in ("l2", "gmm"):
            raise NotImplementedError(f"STP v1 has the heads 'l2' and 'gmm' (:110-128), not {self.fh_loss!r}")
        if self.stp_temporal_c % 32 or not 32 <= self.stp_temporal_c <= 64:
            raise NotImplementedError("stp_temporal_c must be 32 or 64 for the pointwise head kernel")
        if self.condition_func == "D2DTNet":
            self.blk1 = nn.Sequential(D2DTInput(3, 12), D2DTInput(12, 24), D2DTInput(24, 48))
            self.blk2 = D2DTInput(48, self.stp_temporal_c)
        else:
            self.blk1 = FeatureCalapseBlock(3, 12)
            self.blk2 = FeatureCalapseBlock(12, self.stp_temporal_c)
        self.hf_dim = 9
        c = self.stp_temporal_c
        lre = lambda: nn.LeakyReLU(negative_slope=0.2, inplace=True)  # noqa: E731
        if self.fh_loss == "l2":
            self.tail = nn.Sequential(lre(), nn.Conv3d(c, self.hf_dim, 1, 1, 0, bias=True))
        else:
            self.tail_gmm = nn.Sequential(lre(), nn.Conv3d(c, c, 1, 1, 0, bias=True), lre(), nn.Conv3d(c, c, 1, 1, 0, bias=True),
                                          lre(), nn.Conv3d(c, self.hf_dim * self.K * 3, 1, 1, 0, bias=True))
        self.eps = None    # optional injected noise (K, b, 9, t, h, w): one draw per mixture component, as :162-163 makes them

    def _tail_packed(self):
        if self.fh_loss == "gmm":
            convs = [m for m in self.tail_gmm if isinstance(m, nn.Conv3d)]
            key = rt.params_key(*convs)
            if getattr(self, "_tail_key", None) != key:
                from ..packing import pack_pointwise, pad_bias, roundup
                self._tail = [(pack_pointwise(m.weight), pad_bias(m.bias, roundup(m.out_channels, 16)), m.in_channels,
                               roundup(m.out_channels, 16)) for m in convs]
                self._tail_key = key
            return self._tail
        conv = self.tail[1]
        key = rt.params_key(conv)
        if getattr(self, "_tail_key", None) != key:
            from ..packing import pack_pointwise, pad_bias
            self._tail = (pack_pointwise(conv.weight), pad_bias(conv.bias, 16))   # 9 outputs padded to one 16-row tile
            self._tail_key = key
        return self._tail

    def _publish(self, raw5d):
        """as the reference: ``self.parameters`` = the head output (shadowing nn.Module.parameters, :149,153), plus
        ``stp_parameters``"""
        self.stp_parameters = raw5d
        self.parameters = HeadOutput.wrap(raw5d, self)     # a tensor, as in the reference - and still callable (module_util.HeadOutput)

    def _gmm_head(self, feat, b, t, h, w):
        """feat fp32 NHWC (n,h,w,c) -> publishes parameters (b,135,t,h,w) and gmm_v (b,9,t,h,w) (:151-163)"""
        n, npix, sp, dev = b * t, b * t * h * w, _lib.stream_ptr(), feat.device
        (w0, b0, ci0, co0), (w1, b1, ci1, co1), (w2, b2, ci2, co2) = self._tail_packed()
        h1 = torch.empty((npix, co0), dtype=_lib.operand_dtype(), device=dev)
        h2 = torch.empty((npix, co1), dtype=_lib.operand_dtype(), device=dev)
        raw = torch.empty((npix, co2), dtype=torch.float32, device=dev)
        rt.call("selfc_pwconv_run", feat.data_ptr(), 1, h1.data_ptr(), 0, w0.data_ptr(), b0.data_ptr(), npix, ci0, co0, co0, 1, 1, sp)
        rt.call("selfc_pwconv_run", h1.data_ptr(), 0, h2.data_ptr(), 0, w1.data_ptr(), b1.data_ptr(), npix, ci1, co1, co1, 0, 1, sp)
        rt.call("selfc_pwconv_run", h2.data_ptr(), 0, raw.data_ptr(), 1, w2.data_ptr(), b2.data_ptr(), npix, ci2, co2, co2, 0, 0, sp)
        hf, K = self.hf_dim, self.K
        if self.eps is not None:           # (K, b, 9, t, h, w) -> rows [npix][c*K + k]
            eps = self.eps.reshape(K, b, hf, t, h, w).permute(1, 3, 4, 5, 2, 0).reshape(npix, hf * K).to(device=dev, dtype=torch.float32).contiguous()
        else:
            eps = torch.randn((npix, hf * K), dtype=torch.float32, device=dev)
        v = torch.empty((npix, hf), dtype=torch.float32, device=dev)
        rt.call("selfc_gmm_sample_generic", raw.data_ptr(), eps.data_ptr(), v.data_ptr(), npix, hf, K, co2, hf, 0.5, sp)
        self._publish(raw[:, : hf * K * 3].reshape(b, t, h, w, hf * K * 3).permute(0, 4, 1, 2, 3))
        self.gmm_v = v.reshape(b, t, h, w, hf).permute(0, 4, 1, 2, 3)

    @property
    def gmm(self):
        """mixture of the likelihood path (:165-177): weight = softmax over K, mean = idx1, scale = exp(clamp(idx2))"""
        p = self.stp_parameters
        b, _, t, h, w = p.shape
        p = p.reshape(b, self.hf_dim, self.K, 3, t, h, w).permute(0, 1, 4, 5, 6, 2, 3).reshape(-1, self.K, 3)
        mix = torch.distributions.Categorical(F.softmax(p[:, :, 0], dim=1))
        comp = torch.distributions.Normal(p[:, :, 1], torch.exp(torch.clamp(p[:, :, 2], -7, 7)))
        return torch.distributions.MixtureSameFamily(mix, comp)

    def reparametrize(self, mu, logvar):
        """eps * exp(0.5 logvar) + mu (:179-186)"""
        return torch.randn_like(mu).mul(logvar.mul(0.5).exp()).add_(mu)

    def forward(self, x):
        """x (b,3,t,h,w); sets ``stp_parameters`` (the reference's ``self.parameters``) = (b,9,t,h,w)."""
        b, c, t, h, w = x.size()
        temp = x.transpose(1, 2).reshape(b * t, c, h, w)
        from .. import autograd as ag
        if ag.module_needs_grad(temp, self):
            feat = self.blk2(self.blk1(temp))                        # differentiable D2DTInput / FeatureCalapseBlock chain
            if self.fh_loss == "gmm":
                # the three-layer head and the reparameterised sample as differentiable HIP ops: `parameters` feeds neg_llh
                # (torch.distributions, :165-177), `gmm_v` the reverse pass (:151-163)
                convs = [m for m in self.tail_gmm if isinstance(m, nn.Conv3d)]
                prm = [q for m in convs for q in (m.weight, m.bias)]
                raw = ag.HeadFn.apply(feat, convs, self._tail_packed(), t, *prm)          # (b*t, 9*K*3, h, w)
                hf, K, npix = self.hf_dim, self.K, b * t * h * w
                if self.eps is not None:
                    eps = self.eps.reshape(K, b, hf, t, h, w).permute(1, 3, 4, 5, 2, 0).reshape(npix, hf * K).to(device=raw.device, dtype=torch.float32).contiguous()
                else:
                    eps = torch.randn((npix, hf * K), dtype=torch.float32, device=raw.device)
                v = ag.GmmSampleFn.apply(raw, eps, hf, K, 0.5)
                self._publish(raw.reshape(b, t, hf * K * 3, h, w).transpose(1, 2))
                self.gmm_v = v.reshape(b, t, hf, h, w).transpose(1, 2)
                return
            conv = self.tail[1]
            v = ag.PointwiseHeadFn.apply(feat, conv, self._tail_packed(), t, conv.weight, conv.bias)
            self._publish(v.reshape(b, t, self.hf_dim, h, w).transpose(1, 2))
            return
        temp = self.blk2(self.blk1(temp))
        n, cc = b * t, self.stp_temporal_c
        sp = _lib.stream_ptr()
        feat = torch.empty((n, h, w, cc), dtype=torch.float32, device=temp.device)
        rt.call("selfc_nchw_to_nhwc4", temp.data_ptr(), feat.data_ptr(), n, cc, h, w, sp)
        if self.fh_loss == "gmm":
            return self._gmm_head(feat, b, t, h, w)
        wp, bp = self._tail_packed()
        out16 = torch.empty((n, h, w, 16), dtype=torch.float32, device=temp.device)
        rt.call("selfc_pwconv_run", feat.data_ptr(), 1, out16.data_ptr(), 1, wp.data_ptr(), bp.data_ptr(),
                n * h * w, cc, 16, 16, 1, 0, sp)
        self._publish(out16[..., : self.hf_dim].reshape(b, t, h, w, self.hf_dim).permute(0, 4, 1, 2, 3))

    def neg_llh(self, hf):
        if self.fh_loss == "gmm":
            return -self.gmm.log_prob(hf.reshape(-1))
        return torch.mean((hf - self.stp_parameters) ** 2)

    def sample(self):
        return self.gmm_v if self.fh_loss == "gmm" else self.stp_parameters


class SelfCInvNet(nn.Module):
    __getstate__ = cache_free_state      # deepcopy / pickle leave the runtime's caches behind (module_util)
    def __init__(self, opt, channel_in, channel_out, subnet_type, block_num, down_num):
        super().__init__()
        operations = []
        current_channel = channel_in
        sc = subnet(subnet_type, "xavier")
        for i in range(down_num):
            operations.append(HaarDownsampling(current_channel))
            current_channel *= 4
            for _ in range(block_num[i]):
                operations.append(InvBlockExp(sc, current_channel, channel_out))
        self.operations = nn.ModuleList(operations)
        self.stp_net = STPNet(opt)

    def forward(self, x, rev=False, cal_jacobian=False, lr_before_distor=None):
        out = x
        jacobian = 0
        t = _tlen()
        if not rev:
            for op in self.operations:
                out = op.forward(out, rev)
                if cal_jacobian:
                    jacobian += op.jacobian(out, rev)
            bt, c, h, w = out.size()
            b = bt // t
            o5 = out.reshape(b, t, c, h, w).transpose(1, 2)
            self.stp_net(o5[:, 0:3])
            loss_c = self.stp_net.neg_llh(o5[:, 3:])
            return out, loss_c
        bt, c, h, w = x.size()
        b = bt // t
        lr_input = x[:, 0:3].reshape(b, t, 3, h, w).transpose(1, 2)
        self.stp_net(lr_input)
        recon_hf = self.stp_net.sample()
        out = torch.cat((lr_input, recon_hf), dim=1).transpose(1, 2).reshape(b * t, -1, h, w)
        for op in reversed(self.operations):
            out = op.forward(out, rev)
            if cal_jacobian:
                jacobian += op.jacobian(out, rev)
        return out, recon_hf.transpose(1, 2).reshape(b * t, -1, h, w)
