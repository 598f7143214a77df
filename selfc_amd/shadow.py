"""Training narrow modules on the wide kernels: exact zero-padded shadows.

The gradient kernels (csrc/backward.hip, csrc/stp.hip) are built for the kernels' native widths: dense blocks with growth
32, GlobalAgg / STP rows of 64 channels.  The reference also trains narrower instances - the codec variant's STP has hidden
width 24 and dense growth 12 (SelfC_Codec_arch_inv.py:234-376).  Inference already runs those on the wide kernels by
zero-padding the weights at packing time (packing.widen_dense_params: padded features are LeakyReLU(0) = 0 and meet zero
weights, so nothing differs).  Training uses the same fact one level up:

  * a *shadow* module of the native width holds leaf parameters that are the real ones placed into zeros (``sync``);
  * the op runs on the shadow under a nested autograd graph, through the ordinary HIP autograd.Functions;
  * the shadow's gradients are narrowed back to the real parameters' shapes - the adjoint of the placement (``narrow``).

Both directions are pure placement (plus one constant factor for GlobalAgg's softmax temperature, which is ``1 / C`` of the
REAL width, SelfC_GMM_arch_inv.py:277), so the result is what autograd through the narrow reference module gives, up to the
f16 operand rounding every HIP path has."""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple

import torch
import torch.nn as nn

from . import runtime as rt

#: one placement: wide[dst] = scale * real[src]  (index tuples of slices)
Place = Tuple[tuple, tuple, float]


def _sl(*pairs) -> tuple:
    return tuple(slice(a, b) for a, b in pairs)


class Shadow:
    def __init__(self, real: nn.Module, wide: nn.Module, pairs: Sequence[Tuple[torch.Tensor, torch.Tensor, List[Place]]]):
        self.real, self.wide = real, wide
        self.pairs = list(pairs)
        self.real_params = [p for p, _, _ in self.pairs]
        self.wide_params = [q for _, q, _ in self.pairs]
        mapped = {id(q) for q in self.wide_params}
        extra = [n for n, q in wide.named_parameters() if id(q) not in mapped]
        if extra:
            raise AssertionError(f"shadow parameters without a source: {extra}")
        self._key = None

    def sync(self):
        """wide <- placement of real (no autograd); only when the real parameters changed."""
        key = rt.params_key(self.real) + (str(self.real_params[0].device),)
        if key == self._key:
            return
        with torch.no_grad():
            for p, q, places in self.pairs:
                if q.device != p.device:
                    q.data = q.data.to(p.device)
                q.zero_()
                for dst, src, scale in places:
                    q[dst] = p.detach()[src] * scale if scale != 1.0 else p.detach()[src]
        self._key = key

    def narrow(self, wide_grads: Sequence[torch.Tensor]) -> List[torch.Tensor]:
        """adjoint of the placement: gradients w.r.t. the real parameters (None where the shadow got none)."""
        out = []
        for (p, _, places), g in zip(self.pairs, wide_grads):
            if g is None:
                out.append(None)
                continue
            r = torch.zeros_like(p)
            for dst, src, scale in places:
                r[src] += g[dst] * scale if scale != 1.0 else g[dst]
            out.append(r)
        return out


class ShadowFn(torch.autograd.Function):
    """y = call(x) evaluated on the shadow's wide module; differentiable w.r.t. x and the REAL parameters.

    ``call(x)`` must be built from autograd-capable ops (the HIP Functions of selfc_amd.autograd and torch view / pad ops) and
    use the shadow's wide module; it runs under a nested graph that backward() differentiates with torch.autograd.grad."""

    @staticmethod
    def forward(ctx, x, shadow: Shadow, call: Callable, *real_params):
        shadow.sync()
        with torch.enable_grad():
            xw = x.detach().requires_grad_(x.requires_grad)
            yw = call(xw)
        ctx.shadow, ctx.xw, ctx.yw = shadow, xw, yw
        return yw.detach()

    @staticmethod
    def backward(ctx, gy):
        sh = ctx.shadow
        wants = [q for q in sh.wide_params]
        inputs = ([ctx.xw] if ctx.xw.requires_grad else []) + wants
        grads = torch.autograd.grad(ctx.yw, inputs, gy.contiguous(), allow_unused=True)
        dx = None
        if ctx.xw.requires_grad:
            dx, grads = grads[0], grads[1:]
        return (dx, None, None, *sh.narrow(grads))


def shadow_apply(x: torch.Tensor, shadow: Shadow, call: Callable) -> torch.Tensor:
    return ShadowFn.apply(x, shadow, call, *shadow.real_params)


# ---- placement maps --------------------------------------------------------------------------------------------------

def dense_pairs(real, wide, cin: int, cout: int, gc: int, cin_v: int, cout_v: int):
    """conv1..conv5 of a dense block with growth gc into the growth-32 block (packing.widen_dense_params, same layout):
    real input channels keep their place, feature j's gc channels sit at the start of the wide block's j-th 32-channel group."""
    pairs = []
    for k in range(1, 6):
        rc, wc = getattr(real, f"conv{k}"), getattr(wide, f"conv{k}")
        o_real = gc if k < 5 else cout
        tail = (slice(None),) * (rc.weight.dim() - 2)
        places = [((slice(0, o_real), slice(0, cin)) + tail, (slice(None), slice(0, cin)) + tail, 1.0)]
        for j in range(k - 1):
            places.append(((slice(0, o_real), slice(cin_v + 32 * j, cin_v + 32 * j + gc)) + tail,
                           (slice(None), slice(cin + gc * j, cin + gc * (j + 1))) + tail, 1.0))
        pairs.append((rc.weight, wc.weight, places))
        pairs.append((rc.bias, wc.bias, [(_sl((0, o_real)), _sl((0, o_real)), 1.0)]))
    return pairs


def globalagg_pairs(real, wide):
    """GlobalAgg(c) into GlobalAgg(64): top-left blocks; proj2 carries 64 / c so that the wide module's softmax((q k^T) / 64)
    equals the real module's softmax((q k^T) / c)."""
    c = real.c
    tq = 64.0 / c
    sq, v = _sl((0, c), (0, c)), _sl((0, c))
    full = (slice(None),)
    return [
        (real.fc.weight, wide.fc.weight, [(full * 2, full * 2, 1.0)]),
        (real.fc.bias, wide.fc.bias, [(full, full, 1.0)]),
        (real.proj1.weight, wide.proj1.weight, [(sq + full * 2, sq + full * 2, 1.0)]),
        (real.proj1.bias, wide.proj1.bias, [(v, v, 1.0)]),
        (real.proj2.weight, wide.proj2.weight, [(sq, sq, tq)]),
        (real.proj2.bias, wide.proj2.bias, [(v, v, tq)]),
        (real.proj3.weight, wide.proj3.weight, [(sq, sq, 1.0)]),
        (real.proj3.bias, wide.proj3.bias, [(v, v, 1.0)]),
    ]


# ---- shadows of the three narrow module kinds ------------------------------------------------------------------------------

def _cached(real: nn.Module, build: Callable[[], Shadow]) -> Shadow:
    sh = real.__dict__.get("_shadow")
    dev = next(nn.Module.parameters(real)).device
    if sh is None or sh.wide_params[0].device != dev:
        sh = real.__dict__["_shadow"] = build()
    return sh


def dense_shadow(mod) -> Shadow:
    """DenseBlock / D2DTInput with growth < 32 as the growth-32 block of the same class."""
    def build():
        wide = type(mod)(mod.channel_in, mod.channel_out, gc=32, INN_init=False).to(next(nn.Module.parameters(mod)).device)
        wide.requires_grad_(True)
        return Shadow(mod, wide, dense_pairs(mod, wide, mod.channel_in, mod.channel_out, mod.gc, mod.channel_in, mod.channel_out))
    return _cached(mod, build)


def globalagg_shadow(mod) -> Shadow:
    def build():
        wide = type(mod)(64).to(next(nn.Module.parameters(mod)).device)      # same class: keeps the codec copy's TEMP_LEN
        return Shadow(mod, wide, globalagg_pairs(mod, wide))
    return _cached(mod, build)


def stp_shadow(stp) -> Shadow:
    """An STP v2 net with hidden width c < 64 and / or dense growth < 32 (the codec variant: 24 / 12) as the native
    64 / 32 net.  l2 head only (what the codec variant ships): its Conv3d(c, hf_dim) becomes Conv3d(64, roundup(hf_dim, 16))."""
    from .modules.SelfC_GMM_arch_inv import STPNet as WideSTP
    from .modules.Subnet_constructor import D2DTInput
    from .packing import roundup

    def build():
        if stp.fh_loss != "l2":
            raise NotImplementedError("selfc_amd: training a GMM head on a hidden width other than 64 is not built (the codec variant ships fh_loss: l2)")
        dev = next(nn.Module.parameters(stp)).device
        opt = {"global_module": stp.global_module, "stp_blk_num": stp.stp_blk_num + 2, "fh_loss": "l2", "scale": 4, "gmm_k": stp.K}
        wide = WideSTP(opt)
        hf16 = roundup(stp.hf_dim, 16)
        wide.tail_gmm = nn.Sequential(nn.LeakyReLU(negative_slope=0.2, inplace=True), nn.Conv3d(64, hf16, 1, 1, 0, bias=True))
        wide.hf_dim = hf16
        wide = wide.to(dev)
        pairs = []
        for rm, wm in zip(stp._chain(), wide._chain()):
            if isinstance(rm, D2DTInput):
                cin_v, cout_v = (rm.channel_in if rm.channel_in <= 3 else 64), 64
                pairs += dense_pairs(rm, wm, rm.channel_in, rm.channel_out, rm.gc, cin_v, cout_v)
            else:
                pairs += globalagg_pairs(rm, wm)
        rc = [m for m in stp._tail_seq() if isinstance(m, nn.Conv3d)][0]
        wc = wide.tail_gmm[1]
        o, c = rc.out_channels, rc.in_channels
        full3 = (slice(None),) * 3
        pairs.append((rc.weight, wc.weight, [(_sl((0, o), (0, c)) + full3, _sl((0, o), (0, c)) + full3, 1.0)]))
        pairs.append((rc.bias, wc.bias, [(_sl((0, o)), _sl((0, o)), 1.0)]))
        return Shadow(stp, wide, pairs)
    return _cached(stp, build)
