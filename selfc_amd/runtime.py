"""Host-side runtime shared by the drop-in modules: workspace cache, packed-weight
cache, and thin wrappers over the C ABI (include/selfc_hip.h)."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import torch

from . import _lib
from .packing import dense_channels, pack_conv3x3, pack_fused_gh, pack_tconv5, pad_bias, roundup

SUBNET_D2DT = _lib.SUBNET_D2DT
SUBNET_DB2D = _lib.SUBNET_DB2D


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def no_autograd_guard(*tensors):
    """The HIP path is forward-only this round: refuse loudly instead of silently
    returning tensors without a graph."""
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors):
        raise NotImplementedError(
            "selfc_amd: backward kernels are not implemented yet - call under torch.no_grad() "
            "(training through the HIP path is a later milestone, see DESIGN.md)")


class Workspace:
    """Latent state + dense feature buffers for one (kind, N, H, W, c1, c2) problem.

    Layout: include/selfc_hip.h.  Buffers are zero-initialised once: the pad
    channels of x2 / fd are never written by any kernel and must read as 0."""

    def __init__(self, device, kind: int, N: int, T: int, H: int, W: int, c1: int, c2: int):
        self.kind, self.N, self.T, self.H, self.W, self.c1, self.c2 = kind, N, T, H, W, c1, c2
        self.c2p = roundup(c2, 4)
        self.FC = dense_channels(c2)
        f32, f16 = torch.float32, _lib.operand_dtype()
        self.x1 = torch.zeros((N, H, W, 4), dtype=f32, device=device)
        self.x2 = torch.zeros((N, H, W, self.c2p), dtype=f32, device=device)
        # dense buffers are plane-blocked: [C/32][N][H][W][32] (every 32-channel group contiguous per pixel)
        self.fd = torch.zeros((self.FC // 32, N, H, W, 32), dtype=f16, device=device)
        self.gd = torch.zeros((4, N, H, W, 32), dtype=f16, device=device)
        self.hd = torch.zeros((4, N, H, W, 32), dtype=f16, device=device)
        self.s: Optional[torch.Tensor] = None
        self.device = device

    def latent(self, want_s: bool = False) -> _lib.Latent:
        if want_s and self.s is None:
            self.s = torch.zeros((self.N, self.H, self.W, self.c2p), dtype=torch.float32, device=self.device)
        return _lib.Latent(self.kind, self.N, self.T, self.H, self.W, self.c1, self.c2,
                           _ptr(self.x1), _ptr(self.x2), _ptr(self.fd), _ptr(self.gd), _ptr(self.hd),
                           _ptr(self.s) if want_s else None)

    def nbytes(self) -> int:
        ts = [self.x1, self.x2, self.fd, self.gd, self.hd] + ([self.s] if self.s is not None else [])
        return sum(t.numel() * t.element_size() for t in ts)


_WS: Dict[Tuple, Workspace] = {}


def workspace(device, kind, N, T, H, W, c1, c2) -> Workspace:
    key = (str(device), kind, N, T, H, W, c1, c2)
    ws = _WS.get(key)
    if ws is None:
        if len(_WS) >= 4:          # a handful of resolutions at most; drop the oldest
            _WS.pop(next(iter(_WS)))
        ws = _WS[key] = Workspace(device, kind, N, T, H, W, c1, c2)
    return ws


class PackedSubnet:
    """Kernel-layout weights of one DenseBlock / D2DTInput (kept alive with the struct)."""

    def __init__(self, mod, partner=None):
        dev = mod.conv1.weight.device
        self.cin, self.cout, self.kind = mod.channel_in, mod.channel_out, mod.kind
        self.w3 = [pack_conv3x3(getattr(mod, f"conv{i}").weight, self.cin, i) for i in range(1, 5)]
        self.b3 = [pad_bias(getattr(mod, f"conv{i}").bias, 64, dev) for i in range(1, 5)]
        if self.kind == SUBNET_D2DT:
            ws = [mod.conv5.weight] + ([partner.conv5.weight] if partner is not None else [])
            self.w5 = pack_tconv5(ws, self.cin)
        else:
            self.w5 = pack_conv3x3(mod.conv5.weight, self.cin, 5)
        self.b5 = pad_bias(mod.conv5.bias, 64, dev)
        # cin == 3 temporal subnets (G / H of the coupling): fused conv1..4 stream
        self.wfused = None
        if self.cin == 3 and self.kind == SUBNET_D2DT:
            self.wfused = pack_fused_gh([getattr(mod, f"conv{i}").weight for i in range(1, 5)], 3)

    def struct(self) -> _lib.SubnetW:
        s = _lib.SubnetW()
        for i in range(4):
            s.w3[i] = _ptr(self.w3[i])
            s.b3[i] = _ptr(self.b3[i])
        s.w5 = _ptr(self.w5)
        s.b5 = _ptr(self.b5)
        s.wfused = _ptr(self.wfused)
        return s


def params_key(*mods) -> Tuple:
    return tuple((p.data_ptr(), p._version, str(p.device)) for m in mods for p in m.parameters())


def call(name: str, *args):
    rc = getattr(_lib.lib(), name)(*args)
    _lib.check(rc, name)


def nchw_to_latent(x: torch.Tensor, ws: Workspace, with_fd: bool = True):
    call("selfc_nchw_to_latent", _ptr(x), _ptr(ws.x1), _ptr(ws.x2), _ptr(ws.fd) if with_fd else None, ws.FC,
         ws.N, ws.c1, ws.c2, ws.H, ws.W, _lib.stream_ptr())


def latent_to_nchw(ws: Workspace) -> torch.Tensor:
    y = torch.empty((ws.N, ws.c1 + ws.c2, ws.H, ws.W), dtype=torch.float32, device=ws.device)
    call("selfc_latent_to_nchw", _ptr(ws.x1), _ptr(ws.x2), _ptr(y), ws.N, ws.c1, ws.c2, ws.H, ws.W, _lib.stream_ptr())
    return y


def s_to_nchw(ws: Workspace) -> torch.Tensor:
    """InvBlockExp.s (N,c2,H,W) from the kernel's NHWC s buffer."""
    y = torch.empty((ws.N, ws.c2, ws.H, ws.W), dtype=torch.float32, device=ws.device)
    call("selfc_nhwc4_to_nchw", _ptr(ws.s), _ptr(y), ws.N, ws.c2, ws.H, ws.W, _lib.stream_ptr())
    return y


def as_input(x: torch.Tensor) -> torch.Tensor:
    """Boundary contract: NCHW fp32, possibly a non-contiguous view."""
    _lib.require_gpu(x)
    if x.dtype != torch.float32:
        raise TypeError(f"selfc_amd expects float32 NCHW tensors at the module boundary, got {x.dtype}")
    return x.contiguous()


class PackedBlock:
    """Kernel-layout weights of one InvBlockExp (F, G, H) as a selfc_invblock_w."""

    def __init__(self, blk):
        self.F = PackedSubnet(blk.F)
        self.G = PackedSubnet(blk.G, partner=blk.H if blk.G.kind == SUBNET_D2DT else None)
        self.H = PackedSubnet(blk.H)
        self.clamp = float(blk.clamp)

    def struct(self) -> _lib.InvBlockW:
        s = _lib.InvBlockW()
        s.F, s.G, s.H = self.F.struct(), self.G.struct(), self.H.struct()
        s.clamp = self.clamp
        return s


def packed_block(blk) -> PackedBlock:
    key = params_key(blk) + (float(blk.clamp),)
    if getattr(blk, "_pb_key", None) != key:
        for sub in (blk.F, blk.G, blk.H):
            sub._check()
        if blk.split_len1 > 3:
            raise NotImplementedError("selfc_amd coupling kernels cover channel_split_num <= 3 (every shipped config uses 3)")
        blk._pb = PackedBlock(blk)
        blk._pb_key = key
    return blk._pb


def block_array(blocks):
    """(selfc_invblock_w[n], keep-alive list) for selfc_invstack_run."""
    packs = [packed_block(b) for b in blocks]
    arr = (_lib.InvBlockW * len(packs))(*[p.struct() for p in packs])
    return arr, packs
