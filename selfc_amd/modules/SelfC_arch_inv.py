"""Haar-variant SelfC net (model "SelfC") on MI355X.

Mirrors codes/models/modules/SelfC_arch_inv.py: ``STPNet`` v1 (:90-198) and ``SelfCInvNet`` (:276-338) =
[HaarDownsampling, block_num[i] x InvBlockExp] per level + STP; ``hf_dim`` is hard-coded to 9 there, so the
net only works for one Haar level (scale 2).  The reference's temporal length for this file is the module
constant ``TEMP_LEN = 7`` (:6); here GlobalVar is used when set, else 7.

Both conditioners are built: ``condition_func: "D2DTNet"`` (a chain of D2DTInput subnets) and the default
``FeatureCalapseBlock`` pair (space-to-depth + (3,3,3) Conv3d with gc=128, Subnet_constructor.py:280-324),
with the ``fh_loss: "l2"`` head (the reference's GMM branch of this file is CUDA-only, :161).
"""
import torch
import torch.nn as nn

from .. import _lib, runtime as rt
from ..global_var import GlobalVar
from .Inv_arch import HaarDownsampling, InvBlockExp
from .Subnet_constructor import D2DTInput, FeatureCalapseBlock, subnet

TEMP_LEN = 7


def _tlen():
    return GlobalVar.get_Temporal_LEN() or TEMP_LEN


class STPNet(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.stp_d2d_inner_c = opt["stp_d2d_inner_c"]
        self.stp_temporal_c = opt["stp_temporal_c"]
        self.fh_loss = opt["fh_loss"]
        self.K = opt["gmm_mixture_num"]
        self.stp_blk_num = opt["stp_blk_num"]
        self.condition_func = opt["condition_func"]
        if self.fh_loss != "l2":
            raise NotImplementedError("selfc_amd builds the l2 head of STP v1 (its GMM branch is CUDA-only in the reference, :161)")
        if self.stp_temporal_c % 32 or not 32 <= self.stp_temporal_c <= 64:
            raise NotImplementedError("stp_temporal_c must be 32 or 64 for the pointwise head kernel")
        if self.condition_func == "D2DTNet":
            self.blk1 = nn.Sequential(D2DTInput(3, 12), D2DTInput(12, 24), D2DTInput(24, 48))
            self.blk2 = D2DTInput(48, self.stp_temporal_c)
        else:
            self.blk1 = FeatureCalapseBlock(3, 12)
            self.blk2 = FeatureCalapseBlock(12, self.stp_temporal_c)
        self.hf_dim = 9
        self.tail = nn.Sequential(nn.LeakyReLU(negative_slope=0.2, inplace=True),
                                  nn.Conv3d(self.stp_temporal_c, self.hf_dim, 1, 1, 0, bias=True))

    def _tail_packed(self):
        conv = self.tail[1]
        key = rt.params_key(conv)
        if getattr(self, "_tail_key", None) != key:
            from ..packing import pack_pointwise, pad_bias
            self._tail = (pack_pointwise(conv.weight), pad_bias(conv.bias, 16))   # 9 outputs padded to one 16-row tile
            self._tail_key = key
        return self._tail

    def forward(self, x):
        """x (b,3,t,h,w); sets ``stp_parameters`` (the reference's ``self.parameters``) = (b,9,t,h,w)."""
        b, c, t, h, w = x.size()
        temp = x.transpose(1, 2).reshape(b * t, c, h, w)
        from .. import autograd as ag
        if ag.module_needs_grad(temp, self):
            if self.condition_func != "D2DTNet":
                rt.no_autograd_guard(temp, *self.parameters())      # FeatureCalapseBlock has no backward kernels yet
            feat = self.blk2(self.blk1(temp))                        # differentiable D2DTInput chain
            conv = self.tail[1]
            v = ag.PointwiseHeadFn.apply(feat, conv, self._tail_packed(), t, conv.weight, conv.bias)
            self.stp_parameters = v.reshape(b, t, self.hf_dim, h, w).transpose(1, 2)
            return
        temp = self.blk2(self.blk1(temp))
        n, cc = b * t, self.stp_temporal_c
        sp = _lib.stream_ptr()
        feat = torch.empty((n, h, w, cc), dtype=torch.float32, device=temp.device)
        rt.call("selfc_nchw_to_nhwc4", temp.data_ptr(), feat.data_ptr(), n, cc, h, w, sp)
        wp, bp = self._tail_packed()
        out16 = torch.empty((n, h, w, 16), dtype=torch.float32, device=temp.device)
        rt.call("selfc_pwconv_run", feat.data_ptr(), 1, out16.data_ptr(), 1, wp.data_ptr(), bp.data_ptr(),
                n * h * w, cc, 16, 16, 1, 0, sp)
        self.stp_parameters = out16[..., : self.hf_dim].reshape(b, t, h, w, self.hf_dim).permute(0, 4, 1, 2, 3)

    def neg_llh(self, hf):
        return torch.mean((hf - self.stp_parameters) ** 2)

    def sample(self):
        return self.stp_parameters


class SelfCInvNet(nn.Module):
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
