"""Invertible rescaling blocks on MI355X: InvBlockExp, HaarDownsampling, InvRescaleNet.

Mirrors codes/models/modules/Inv_arch.py (:8-41, :44-84, :87-127); the same
InvBlockExp is what SelfC_arch_inv.py:8-41 and SelfC_GMM_arch_inv.py:8-41 define.
"""
import numpy as np
import torch
import torch.nn as nn

from .module_util import cache_free_state

from .. import _lib, runtime as rt
from ..global_var import GlobalVar


class InvBlockExp(nn.Module):
    __getstate__ = cache_free_state      # deepcopy / pickle leave the runtime's caches behind (module_util)
    """Affine coupling layer (Inv_arch.py:8-41).

    forward : y1 = x1 + F(x2); s = clamp*(2*sigmoid(H(y1))-1); y2 = x2*exp(s) + G(y1)
    reverse : s = clamp*(2*sigmoid(H(x1))-1); y2 = (x2-G(x1))/exp(s); y1 = x1 - F(y2)
    One call = selfc_invblock_run: 8 dense 3x3 conv launches + 2 conv5 launches with
    the coupling fused into their epilogues."""

    def __init__(self, subnet_constructor, channel_num, channel_split_num, clamp=1.):
        super().__init__()
        self.split_len1 = channel_split_num
        self.split_len2 = channel_num - channel_split_num
        self.clamp = clamp
        self.F = subnet_constructor(self.split_len2, self.split_len1)
        self.G = subnet_constructor(self.split_len1, self.split_len2)
        self.H = subnet_constructor(self.split_len1, self.split_len2)

    # InvBlockExp.s (Inv_arch.py:27,30) is kept in the kernels' NHWC layout and converted to NCHW on first access
    @property
    def s(self):
        v = self.__dict__.get("_s_nchw")
        if v is None:
            ws = self.__dict__.get("_s_ws")
            if ws is None:
                raise AttributeError("InvBlockExp.s is set by forward()")
            v = self.__dict__["_s_nchw"] = rt.s_to_nchw(ws)
        return v

    @s.setter
    def s(self, value):
        self.__dict__["_s_nchw"] = value
        self.__dict__["_s_ws"] = None

    def _set_s_lazy(self, ws):
        # keep only what s_to_nchw needs (the NHWC s buffer and its dims), not the block's whole saved workspace
        from types import SimpleNamespace
        self.__dict__["_s_ws"] = SimpleNamespace(s=ws.s, N=ws.N, c2=ws.c2, H=ws.H, W=ws.W, device=ws.device)
        self.__dict__["_s_nchw"] = None

    def _temporal_len(self):
        if self.F.kind == rt.SUBNET_D2DT:
            t = GlobalVar.get_Temporal_LEN()
            if not t:
                raise RuntimeError("GlobalVar.set_Temporal_LEN(T) must be called before an InvBlockExp(D2DTNet) forward")
            return t
        return 1

    def forward(self, x, rev=False):
        x = rt.as_input(x)
        n, c, h, w = x.shape
        if c != self.split_len1 + self.split_len2:
            raise RuntimeError(f"InvBlockExp expects {self.split_len1 + self.split_len2} channels, got {c}")
        t = self._temporal_len()
        if n % t:
            raise RuntimeError(f"{n} frames are not a multiple of the temporal length {t}")
        from .. import autograd as ag
        if self.split_len1 > 3 or (self.F.kind == rt.SUBNET_D2DT and self.split_len2 > 48):
            return self._forward_composed(x, bool(rev))
        if ag.module_needs_grad(x, self):          # training: buffers kept for the HIP backward (autograd.py)
            return ag.InvBlockFn.apply(x, self, bool(rev), t, *ag.block_params(self))
        ws = rt.workspace(x.device, self.F.kind, n, t, h, w, self.split_len1, self.split_len2)
        pb = rt.packed_block(self)
        rt.nchw_to_latent(x, ws)
        bw, lat = pb.struct(), ws.latent(want_s=True)
        rt.call("selfc_invblock_run", bw, lat, 1 if rev else 0, _lib.stream_ptr())
        self.s = rt.s_to_nchw(ws)
        return rt.latent_to_nchw(ws)

    def _forward_composed(self, x, rev):
        """channel_split_num > 3 (Inv_arch.py:12-13 accepts any split), or a D2DTNet block whose x2 is wider than the 48 channels
        the fused temporal conv5 + coupling kernel is built for: the fused block kernels keep x1 in a 4-float pixel, so a
        wider split runs as the reference composes it - F, G, H as stand-alone subnets (selfc_subnet_run, HIP backward through
        their autograd Functions) and the affine coupling as its own elementwise HIP pass (autograd.CouplingFn)."""
        from .. import autograd as ag
        s1 = self.split_len1
        x1, x2 = x[:, :s1].contiguous(), x[:, s1:].contiguous()
        if not rev:
            y1 = x1 + self.F(x2)
            y2, s = ag.CouplingFn.apply(x2, self.G(y1), self.H(y1), self.clamp, False)
        else:
            y2, s = ag.CouplingFn.apply(x2, self.G(x1), self.H(x1), self.clamp, True)
            y1 = x1 - self.F(y2)
        self.s = s.detach()
        return torch.cat((y1, y2), 1)

    def jacobian(self, x, rev=False):
        jac = torch.sum(self.s)
        return (-jac if rev else jac) / x.shape[0]


class HaarDownsampling(nn.Module):
    """2x2 Haar butterfly + band shuffle (Inv_arch.py:44-84).  ``haar_weights`` is
    kept as a frozen parameter only because it is part of the checkpoint layout;
    the kernel hard-codes the +-1 pattern it encodes."""

    def __init__(self, channel_in):
        super().__init__()
        self.channel_in = channel_in
        w = torch.ones(4, 1, 2, 2)
        w[1, 0, 0, 1] = w[1, 0, 1, 1] = -1
        w[2, 0, 1, 0] = w[2, 0, 1, 1] = -1
        w[3, 0, 1, 0] = w[3, 0, 0, 1] = -1
        self.haar_weights = nn.Parameter(torch.cat([w] * channel_in, 0), requires_grad=False)

    def forward(self, x, rev=False):
        from .. import autograd as ag
        if ag.needs_grad(x):
            return ag.HaarFn.apply(x, self, bool(rev))
        return self._run(x, rev)

    def _run(self, x, rev=False, track=True):
        x = rt.as_input(x)
        n, c, h, w = x.shape
        sp = _lib.stream_ptr()
        if not track:               # adjoint evaluation inside a backward: leave elements / last_jac alone
            if not rev:
                y = torch.empty((n, 4 * c, h // 2, w // 2), dtype=x.dtype, device=x.device)
                rt.call("selfc_haar_fwd_nchw", x.data_ptr(), y.data_ptr(), n, c, h, w, sp)
            else:
                y = torch.empty((n, c // 4, 2 * h, 2 * w), dtype=x.dtype, device=x.device)
                rt.call("selfc_haar_inv_nchw", x.data_ptr(), y.data_ptr(), n, c // 4, h, w, sp)
            return y
        self.elements = c * h * w
        if not rev:
            if c != self.channel_in:
                raise RuntimeError(f"HaarDownsampling({self.channel_in}) got {c} channels")
            self.last_jac = self.elements / 4 * np.log(1 / 16.)
            y = torch.empty((n, 4 * c, h // 2, w // 2), dtype=x.dtype, device=x.device)
            rt.call("selfc_haar_fwd_nchw", x.data_ptr(), y.data_ptr(), n, c, h, w, sp)
            return y
        if c != 4 * self.channel_in:
            raise RuntimeError(f"HaarDownsampling({self.channel_in}) reverse got {c} channels")
        self.last_jac = self.elements / 4 * np.log(16.)
        y = torch.empty((n, c // 4, 2 * h, 2 * w), dtype=x.dtype, device=x.device)
        rt.call("selfc_haar_inv_nchw", x.data_ptr(), y.data_ptr(), n, c // 4, h, w, sp)
        return y

    def jacobian(self, x, rev=False):
        return self.last_jac


class InvRescaleNet(nn.Module):
    __getstate__ = cache_free_state      # deepcopy / pickle leave the runtime's caches behind (module_util)
    """IRN-style [Haar, block_num[i] x InvBlockExp] x down_num (Inv_arch.py:87-127)."""

    def __init__(self, channel_in=3, channel_out=3, subnet_constructor=None, block_num=[], down_num=2):
        super().__init__()
        operations = []
        current_channel = channel_in
        for i in range(down_num):
            operations.append(HaarDownsampling(current_channel))
            current_channel *= 4
            for _ in range(block_num[i]):
                operations.append(InvBlockExp(subnet_constructor, current_channel, channel_out))
        self.operations = nn.ModuleList(operations)

    def forward(self, x, rev=False, cal_jacobian=False):
        out = x
        jacobian = 0
        if not rev:
            for op in self.operations:
                out = op.forward(out, rev)
                if cal_jacobian:
                    jacobian += op.jacobian(out, rev)
            return out[:, 0:3], (out[:, 3:] ** 2).mean()
        b, c, h, w = out.size()
        # the reference concatenates 45 uniform-random HF channels whatever the net needs (:116-118)
        out = torch.cat([out, torch.rand((b, 45, h, w), device=out.device)], dim=1)
        last = self.operations[-1]
        need = last.split_len1 + last.split_len2 if isinstance(last, InvBlockExp) else 4 * last.channel_in
        out = out[:, :need]   # the reference's narrow() silently ignores the excess channels
        for op in reversed(self.operations):
            out = op.forward(out, rev)
            if cal_jacobian:
                jacobian += op.jacobian(out, rev)
        return out, None
