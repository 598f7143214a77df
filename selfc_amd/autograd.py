"""Training through the HIP path: torch.autograd.Functions of the boundary modules.

The reference trains with stock autograd (SelfC_model.py:153-176); here every differentiable boundary op keeps
what its backward needs (the f16 dense feature buffers the forward kernels leave behind, the coupling's s and
one side of the latent) and calls the gradient entry points of csrc/backward.hip.  Parameter gradients come
back in the reference's own tensor layouts, so optimizers / DDP see ordinary ``.grad`` tensors.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib, runtime as rt
from .packing import dense_channels, pack_subnet_bwd, roundup

_SCRATCH: Dict[Tuple, torch.Tensor] = {}


def _scratch(device, n, h, w, cin, cout) -> torch.Tensor:
    need = _lib.lib().selfc_subnet_bwd_scratch_bytes(n, h, w, cin, cout)
    if need == 0:
        raise RuntimeError("selfc_subnet_bwd_scratch_bytes: invalid shape")
    key = str(device)
    buf = _SCRATCH.get(key)
    if buf is None or buf.numel() < need:
        buf = _SCRATCH[key] = torch.empty(need, dtype=torch.uint8, device=device)
    return buf


class PackedSubnetBwd:
    def __init__(self, mod):
        ws = [getattr(mod, f"conv{i}").weight for i in range(1, 6)]
        self.wt5, self.wtd, self.wtx = pack_subnet_bwd(ws, mod.channel_in, mod.channel_out, mod.kind == rt.SUBNET_D2DT)

    def struct(self) -> _lib.SubnetBW:
        s = _lib.SubnetBW()
        s.wt5 = self.wt5.data_ptr()
        for i in range(3):
            s.wtd[i] = self.wtd[i].data_ptr()
        s.wtx = self.wtx.data_ptr()
        return s


def packed_bwd(mod) -> PackedSubnetBwd:
    key = rt.params_key(mod)
    if getattr(mod, "_pkb_key", None) != key:
        mod._check()
        if mod.channel_out > 96:
            raise NotImplementedError("subnet backward covers channel_out <= 96")
        mod._pkb = PackedSubnetBwd(mod)
        mod._pkb_key = key
    return mod._pkb


def subnet_params(mod) -> List[torch.Tensor]:
    """conv1.weight, conv1.bias, ..., conv5.weight, conv5.bias (the order the Functions take and return)."""
    out = []
    for i in range(1, 6):
        conv = getattr(mod, f"conv{i}")
        if conv.bias is None:
            raise NotImplementedError("subnet backward expects bias=True convs (every shipped config)")
        out += [conv.weight, conv.bias]
    return out


def subnet_bwd(mod, dense: torch.Tensor, xin: Optional[torch.Tensor], dout: torch.Tensor, sign: float,
               dx: Optional[torch.Tensor], accumulate_dx: bool, n: int, t: int, h: int, w: int,
               want_params: bool = True) -> List[Optional[torch.Tensor]]:
    """Backward of one DenseBlock / D2DTInput on kernel-layout buffers; returns the 10 parameter gradients
    (reference layouts) or Nones."""
    cin, cout = mod.channel_in, mod.channel_out
    pk = packed_bwd(mod)
    dev = dout.device
    grads: List[Optional[torch.Tensor]] = [None] * 10
    wg = (C.c_void_p * 5)()
    bg = (C.c_void_p * 5)()
    if want_params:
        for i in range(5):
            conv = getattr(mod, f"conv{i + 1}")
            gw = torch.empty(conv.weight.shape, dtype=torch.float32, device=dev)
            gb = torch.empty(conv.bias.shape, dtype=torch.float32, device=dev)
            grads[2 * i], grads[2 * i + 1] = gw, gb
            wg[i], bg[i] = gw.data_ptr(), gb.data_ptr()
    scratch = _scratch(dev, n, h, w, cin, cout)
    bw = pk.struct()
    rt.call("selfc_subnet_bwd", bw, mod.kind, dense.data_ptr(), None if xin is None else xin.data_ptr(), dout.data_ptr(),
            float(sign), None if dx is None else dx.data_ptr(), 1 if accumulate_dx else 0,
            wg if want_params else None, bg if want_params else None, 0.0,
            scratch.data_ptr(), scratch.numel(), n, t, h, w, cin, cout, _lib.stream_ptr())
    return grads


class SubnetFn(torch.autograd.Function):
    """DenseBlock.forward / D2DTInput.forward (Subnet_constructor.py:26-34,119-133) with a HIP backward."""

    @staticmethod
    def forward(ctx, x, mod, t, *params):
        x = rt.as_input(x)
        n, cin, h, w = x.shape
        pk = mod.packed()
        dev, sp = x.device, _lib.stream_ptr()
        cinp, coutp = roundup(cin, 4), roundup(mod.channel_out, 4)
        xin = torch.empty((n, h, w, cinp), dtype=torch.float32, device=dev)
        rt.call("selfc_nchw_to_nhwc4", x.data_ptr(), xin.data_ptr(), n, cin, h, w, sp)
        dense = torch.zeros((dense_channels(cin) // 32, n, h, w, 32), dtype=_lib.operand_dtype(), device=dev)
        yout = torch.empty((n, h, w, coutp), dtype=torch.float32, device=dev)
        sw = pk.struct()
        rt.call("selfc_subnet_run", sw, mod.kind, xin.data_ptr(), yout.data_ptr(), dense.data_ptr(),
                n, t, h, w, cin, mod.channel_out, sp)
        y = torch.empty((n, mod.channel_out, h, w), dtype=torch.float32, device=dev)
        rt.call("selfc_nhwc4_to_nchw", yout.data_ptr(), y.data_ptr(), n, mod.channel_out, h, w, sp)
        ctx.mod, ctx.t, ctx.shape = mod, t, (n, cin, h, w)
        ctx.dense = dense
        ctx.xin = xin if cin <= 3 else None
        return y

    @staticmethod
    def backward(ctx, gy):
        mod, t = ctx.mod, ctx.t
        n, cin, h, w = ctx.shape
        gy = gy.contiguous().float()
        dev, sp = gy.device, _lib.stream_ptr()
        cout = mod.channel_out
        dout = torch.empty((n, h, w, roundup(cout, 4)), dtype=torch.float32, device=dev)
        rt.call("selfc_nchw_to_nhwc4", gy.data_ptr(), dout.data_ptr(), n, cout, h, w, sp)
        dxl = torch.empty((n, h, w, roundup(cin, 4)), dtype=torch.float32, device=dev)
        grads = subnet_bwd(mod, ctx.dense, ctx.xin, dout, 1.0, dxl, False, n, t, h, w,
                           want_params=any(ctx.needs_input_grad[3:]))
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((n, cin, h, w), dtype=torch.float32, device=dev)
            rt.call("selfc_nhwc4_to_nchw", dxl.data_ptr(), dx.data_ptr(), n, cin, h, w, sp)
        return (dx, None, None, *grads)


class InvBlockFn(torch.autograd.Function):
    """InvBlockExp.forward(x, rev) (Inv_arch.py:21-33) with a HIP backward.

    Saved per call: the three dense feature buffers, s, and the side of the latent the gradient formulas need
    (forward: x2 in and y1 out; reverse: x1 in and y2 out)."""

    @staticmethod
    def forward(ctx, x, blk, rev, t, *params):
        x = rt.as_input(x)
        n, c, h, w = x.shape
        c1, c2 = blk.split_len1, blk.split_len2
        ws = rt.Workspace(x.device, blk.F.kind, n, t, h, w, c1, c2)      # private: kept for backward
        pb = rt.packed_block(blk)
        rt.nchw_to_latent(x, ws)
        keep = (ws.x1 if rev else ws.x2).clone()                          # the input side the kernels overwrite
        bw, lat = pb.struct(), ws.latent(want_s=True)
        rt.call("selfc_invblock_run", bw, lat, 1 if rev else 0, _lib.stream_ptr())
        blk.s = rt.s_to_nchw(ws)
        ctx.blk, ctx.rev, ctx.t, ctx.ws, ctx.keep = blk, bool(rev), t, ws, keep
        return rt.latent_to_nchw(ws)

    @staticmethod
    def backward(ctx, gy):
        blk, rev, t, ws, keep = ctx.blk, ctx.rev, ctx.t, ctx.ws, ctx.keep
        n, h, w, c1, c2 = ws.N, ws.H, ws.W, ws.c1, ws.c2
        gy = gy.contiguous().float()
        dev, sp = gy.device, _lib.stream_ptr()
        want = any(ctx.needs_input_grad[4:])
        d1 = torch.empty((n, h, w, 4), dtype=torch.float32, device=dev)
        d2 = torch.empty((n, h, w, ws.c2p), dtype=torch.float32, device=dev)
        rt.call("selfc_nchw_to_latent", gy.data_ptr(), d1.data_ptr(), d2.data_ptr(), None, ws.FC, n, c1, c2, h, w, sp)
        dx2 = torch.empty_like(d2)
        dh = torch.empty_like(d2)
        nel = d2.numel()
        clamp = float(blk.clamp)
        if not rev:
            # y1 = x1 + F(x2); y2 = x2*e^s + G(y1), s = s(H(y1)).  keep = x2 (input), ws.x1 = y1
            rt.call("selfc_coupling_bwd", 0, keep.data_ptr(), ws.s.data_ptr(), d2.data_ptr(), dx2.data_ptr(), dh.data_ptr(), clamp, nel, sp)
            gG = subnet_bwd(blk.G, ws.gd, ws.x1, d2, 1.0, d1, True, n, t, h, w, want)
            gH = subnet_bwd(blk.H, ws.hd, ws.x1, dh, 1.0, d1, True, n, t, h, w, want)
            # the forward's epilogue replaced F's f16 input copy by y2: put x2 back before F's weight gradients
            rt.call("selfc_nhwc_to_planes", keep.data_ptr(), ws.fd.data_ptr(), n * h * w, c2, sp)
            gF = subnet_bwd(blk.F, ws.fd, None, d1, 1.0, dx2, True, n, t, h, w, want)
        else:
            # y2 = (x2 - G(x1))*e^-s, s = s(H(x1)); y1 = x1 - F(y2).  keep = x1 (input), ws.x2 = y2 (also in fd)
            gF = subnet_bwd(blk.F, ws.fd, None, d1, -1.0, d2, True, n, t, h, w, want)
            rt.call("selfc_coupling_bwd", 1, ws.x2.data_ptr(), ws.s.data_ptr(), d2.data_ptr(), dx2.data_ptr(), dh.data_ptr(), clamp, nel, sp)
            gG = subnet_bwd(blk.G, ws.gd, keep, dx2, -1.0, d1, True, n, t, h, w, want)
            gH = subnet_bwd(blk.H, ws.hd, keep, dh, 1.0, d1, True, n, t, h, w, want)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((n, c1 + c2, h, w), dtype=torch.float32, device=dev)
            rt.call("selfc_latent_to_nchw", d1.data_ptr(), dx2.data_ptr(), dx.data_ptr(), n, c1, c2, h, w, sp)
        return (dx, None, None, None, *gF, *gG, *gH)


def block_params(blk) -> List[torch.Tensor]:
    return subnet_params(blk.F) + subnet_params(blk.G) + subnet_params(blk.H)


class FreqFn(torch.autograd.Function):
    """FrequencyAnalyzer.forward(x, rev) (SelfC_GMM_arch_inv.py:69-82) with its adjoint as backward."""

    @staticmethod
    def forward(ctx, x, mod, rev):
        ctx.rev = bool(rev)
        ctx.shape = tuple(x.shape)
        with torch.no_grad():
            return mod._run(x, rev)

    @staticmethod
    def backward(ctx, gy):
        gy = gy.contiguous().float()
        dev, sp = gy.device, _lib.stream_ptr()
        if not ctx.rev:
            n, _, hh, ww = ctx.shape                                   # input (N,3,H,W), gy (N,51,H/4,W/4)
            h, w = hh // 4, ww // 4
            d1 = torch.empty((n, h, w, 4), dtype=torch.float32, device=dev)
            d2 = torch.empty((n, h, w, 48), dtype=torch.float32, device=dev)
            rt.call("selfc_nchw_to_latent", gy.data_ptr(), d1.data_ptr(), d2.data_ptr(), None, 0, n, 3, 48, h, w, sp)
            dx = torch.empty((n, 3, hh, ww), dtype=torch.float32, device=dev)
            rt.call("selfc_freq_fwd_bwd", d1.data_ptr(), d2.data_ptr(), dx.data_ptr(), n, hh, ww, sp)
            return dx, None, None
        n, _, h, w = ctx.shape                                          # input (N,51,h,w), gy (N,3,4h,4w)
        d1 = torch.empty((n, h, w, 4), dtype=torch.float32, device=dev)
        d2 = torch.empty((n, h, w, 48), dtype=torch.float32, device=dev)
        rt.call("selfc_freq_inv_bwd", gy.data_ptr(), d1.data_ptr(), d2.data_ptr(), n, 4 * h, 4 * w, sp)
        dx = torch.empty((n, 51, h, w), dtype=torch.float32, device=dev)
        rt.call("selfc_latent_to_nchw", d1.data_ptr(), d2.data_ptr(), dx.data_ptr(), n, 3, 48, h, w, sp)
        return dx, None, None


def needs_grad(*tensors) -> bool:
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)
