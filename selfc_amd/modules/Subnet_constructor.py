"""Dense-block subnets and the ``subnet()`` plugin factory on MI355X.

Mirrors codes/models/modules/Subnet_constructor.py: ``DenseBlock`` (:8-34),
``D2DTInput`` (:98-133) and ``subnet`` (:719-788) keep their constructor
signatures, attribute names (conv1..conv5, lrelu) and therefore their
state_dict keys / tensor shapes, so reference checkpoints load strict=True.
The nn.Conv layers only own the parameters; forward repacks them once into MFMA
fragments (selfc_amd/packing.py) and runs csrc/dense_conv.hip through the C ABI.
"""
import torch
import torch.nn as nn

from .module_util import cache_free_state

from .. import _lib, runtime as rt
from ..global_var import GlobalVar
from ..packing import dense_channels, roundup
from . import module_util as mutil


class _DenseSubnet(nn.Module):
    __getstate__ = cache_free_state      # deepcopy / pickle leave the runtime's caches behind (module_util)
    kind = None  # SUBNET_DB2D / SUBNET_D2DT

    def _check(self):
        if not 1 <= self.gc <= 32:
            raise NotImplementedError("selfc_amd dense-block kernels cover growth channels gc <= 32 (gc < 32 is zero-padded to 32)")
        if self.channel_in > 96 or self.channel_out > (32 if self.kind == rt.SUBNET_DB2D else 64):
            raise NotImplementedError(f"subnet {self.channel_in}->{self.channel_out} is outside the compiled kernel set")

    def packed(self) -> rt.PackedSubnet:
        return rt.packed_subnet(self)

    def _run(self, x: torch.Tensor, T: int) -> torch.Tensor:
        """x NCHW (N,cin,H,W) -> NCHW (N,cout,H,W) through selfc_subnet_run."""
        x = rt.as_input(x)
        n, cin, h, w = x.shape
        if cin != self.channel_in:
            raise RuntimeError(f"expected {self.channel_in} input channels, got {cin}")
        if n % T:
            raise RuntimeError(f"{n} frames are not a multiple of the temporal length {T}")
        from .. import autograd as ag
        if ag.module_needs_grad(x, self):          # training: same kernels, buffers kept for the HIP backward
            if self.gc != 32:
                # growth < 32 (the codec variant's dense blocks): train the exactly equivalent growth-32 block (shadow.py)
                from .. import shadow
                sh = shadow.dense_shadow(self)
                return shadow.shadow_apply(x, sh, lambda xw: ag.SubnetFn.apply(xw, sh.wide, T, *ag.subnet_params(sh.wide)))
            return ag.SubnetFn.apply(x, self, T, *ag.subnet_params(self))
        pk = self.packed()
        dev, sp = x.device, _lib.stream_ptr()
        cinp, coutp = roundup(cin, 4), roundup(self.channel_out, 4)
        xin = torch.empty((n, h, w, cinp), dtype=torch.float32, device=dev)
        rt.call("selfc_nchw_to_nhwc4", x.data_ptr(), xin.data_ptr(), n, cin, h, w, sp)
        dense = torch.zeros((dense_channels(cin) // 32, n, h, w, 32), dtype=_lib.operand_dtype(), device=dev)
        yout = torch.empty((n, h, w, coutp), dtype=torch.float32, device=dev)
        sw = pk.struct()
        rt.call("selfc_subnet_run", sw, self.kind, xin.data_ptr(), yout.data_ptr(), dense.data_ptr(),
                n, T, h, w, cin, self.channel_out, sp)
        y = torch.empty((n, self.channel_out, h, w), dtype=torch.float32, device=dev)
        rt.call("selfc_nhwc4_to_nchw", yout.data_ptr(), y.data_ptr(), n, self.channel_out, h, w, sp)
        return y


class DenseBlock(_DenseSubnet):
    """2-D dense block, 5x Conv2d 3x3 (Subnet_constructor.py:8-34)."""
    kind = rt.SUBNET_DB2D

    def __init__(self, channel_in, channel_out, init='xavier', gc=32, bias=True, INN_init=True, is_res=False):
        super().__init__()
        self.channel_in, self.channel_out, self.gc = channel_in, channel_out, gc
        self.conv1 = nn.Conv2d(channel_in, gc, 3, 1, 1, bias=bias)
        self.conv2 = nn.Conv2d(channel_in + gc, gc, 3, 1, 1, bias=bias)
        self.conv3 = nn.Conv2d(channel_in + 2 * gc, gc, 3, 1, 1, bias=bias)
        self.conv4 = nn.Conv2d(channel_in + 3 * gc, gc, 3, 1, 1, bias=bias)
        self.conv5 = nn.Conv2d(channel_in + 4 * gc, channel_out, 3, 1, 1, bias=bias)
        self.lrelu = nn.LeakyReLU(negative_slope=0.2, inplace=True)
        first4 = [self.conv1, self.conv2, self.conv3, self.conv4]
        if INN_init:
            # xavier*0.1 (or kaiming*0.1) on conv1-4, conv5 = 0: identity coupling at init (:17-22)
            (mutil.initialize_weights_xavier if init == 'xavier' else mutil.initialize_weights)(first4, 0.1)
            mutil.initialize_weights(self.conv5, 0)
        else:
            mutil.initialize_weights_xavier(first4 + [self.conv5], 1)
        self.is_res = is_res

    def forward(self, x):
        y = self._run(x, 1)
        return y + x if self.is_res else y


class D2DTInput(_DenseSubnet):
    """Per-frame 3x3 dense convs + 3-tap temporal conv5 (Subnet_constructor.py:98-133).

    The INN_init branch only matches nn.Conv2d in the reference's helpers, so for
    these Conv3d layers it is a no-op and PyTorch's default init stays (trap 4);
    ``is_res`` is accepted and ignored, as in the reference."""
    kind = rt.SUBNET_D2DT

    def __init__(self, channel_in, channel_out, init='xavier', gc=32, bias=True, INN_init=True, is_res=False):
        super().__init__()
        self.channel_in, self.channel_out, self.gc = channel_in, channel_out, gc
        self.conv1 = nn.Conv3d(channel_in, gc, (1, 3, 3), 1, (0, 1, 1), bias=bias)
        self.conv2 = nn.Conv3d(channel_in + gc, gc, (1, 3, 3), 1, (0, 1, 1), bias=bias)
        self.conv3 = nn.Conv3d(channel_in + 2 * gc, gc, (1, 3, 3), 1, (0, 1, 1), bias=bias)
        self.conv4 = nn.Conv3d(channel_in + 3 * gc, gc, (1, 3, 3), 1, (0, 1, 1), bias=bias)
        self.conv5 = nn.Conv3d(channel_in + 4 * gc, channel_out, (3, 1, 1), 1, (1, 0, 0), bias=bias)
        self.lrelu = nn.LeakyReLU(negative_slope=0.2, inplace=True)
        if INN_init:
            first4 = [self.conv1, self.conv2, self.conv3, self.conv4]
            (mutil.initialize_weights_xavier if init == 'xavier' else mutil.initialize_weights)(first4, 0.1)
            mutil.initialize_weights(self.conv5, 0)

    def forward(self, x, io_type="2d"):
        if io_type == '3d':                       # (b,c,t,h,w) in and out (:117-118,131)
            b, c, t, h, w = x.shape
            y = self._run(x.transpose(1, 2).reshape(b * t, c, h, w), t)
            return y.reshape(b, t, -1, h, w).transpose(1, 2)
        t = GlobalVar.get_Temporal_LEN()
        if not t:
            raise RuntimeError("GlobalVar.set_Temporal_LEN(T) must be called before a D2DTInput forward")
        return self._run(x, t)


class SpaceToDepth(nn.Module):
    """(N,C,H,W) -> (N,C*S*S,H/S,W/S), out channel (sy*S+sx)*C + c (Subnet_constructor.py:242-257); pure view shuffle."""

    def __init__(self, block_size=4):
        super().__init__()
        assert block_size in {2, 4}, "Space2Depth only supports blocks size = 4 or 2"
        self.block_size = block_size

    def forward(self, x):
        n, c, h, w = x.size()
        s = self.block_size
        x = x.view(n, c, h // s, s, w // s, s).permute(0, 3, 5, 1, 2, 4).contiguous()
        return x.view(n, c * s * s, h // s, w // s)


class FeatureCalapseBlock(nn.Module):
    __getstate__ = cache_free_state      # deepcopy / pickle leave the runtime's caches behind (module_util)
    """Space-to-depth dense block of STP v1 (Subnet_constructor.py:280-324): at 1/scale resolution, channels
    scale^2*cin -> [4 x (scale*gc) features] -> scale^2*cout, conv1 and conv5 are (3,3,3) Conv3d, conv2-4
    (1,3,3); then PixelShuffle back.  The five convs run on selfc_conv_planes_run (dense_conv.hip in its
    generic plane-list mode: temporal taps + 32-channel output groups); the two shuffles are views."""

    def __init__(self, channel_in, channel_out, scale=4, init='xavier', gc=32, bias=True, INN_init=True, is_res=False):
        super().__init__()
        self.scale = scale
        self.is_res = is_res
        if scale > 1:
            self.ds = SpaceToDepth(scale)
            self.us = nn.PixelShuffle(scale)
        self.cin = (scale ** 2) * channel_in
        self.cout = (scale ** 2) * channel_out
        self.gc = scale * gc
        ci, co, g = self.cin, self.cout, self.gc
        self.conv1 = nn.Conv3d(ci, g, (3, 3, 3), 1, (1, 1, 1), bias=bias)
        self.conv2 = nn.Conv3d(ci + g, g, (1, 3, 3), 1, (0, 1, 1), bias=bias)
        self.conv3 = nn.Conv3d(ci + 2 * g, g, (1, 3, 3), 1, (0, 1, 1), bias=bias)
        self.conv4 = nn.Conv3d(ci + 3 * g, g, (1, 3, 3), 1, (0, 1, 1), bias=bias)
        self.conv5 = nn.Conv3d(ci + 4 * g, co, (3, 3, 3), 1, (1, 1, 1), bias=bias)
        self.lrelu = nn.LeakyReLU(negative_slope=0.2, inplace=True)
        if INN_init:   # only matches nn.Conv2d in the reference's helpers: a no-op for these Conv3d layers
            first4 = [self.conv1, self.conv2, self.conv3, self.conv4]
            (mutil.initialize_weights_xavier if init == 'xavier' else mutil.initialize_weights)(first4, 0.1)
            mutil.initialize_weights(self.conv5, 0)

    def _packed(self):
        key = rt.params_key(self)
        if getattr(self, "_pk_key", None) != key:
            from ..packing import pack_conv_planes, pad_bias
            if self.gc % 32 or self.cout % 32:
                raise NotImplementedError("FeatureCalapseBlock kernels need scale*gc and scale^2*cout to be multiples of 32")
            convs = [getattr(self, f"conv{i}") for i in range(1, 6)]
            self._pk = [(pack_conv_planes(c.weight, self.cin), pad_bias(c.bias, c.out_channels, c.weight.device)) for c in convs]
            self._pk_key = key
        return self._pk

    def _run_planes(self, xs, t):
        """xs (n, cin, h, w) NCHW at 1/scale resolution -> (y (n, cout, h, w), the block's f16 plane buffer [x | f1..f4])"""
        n, c, h, w = xs.shape
        dev, sp = xs.device, _lib.stream_ptr()
        pk = self._packed()
        pin = roundup(self.cin, 32) // 32
        gp = self.gc // 32
        nhwc = torch.empty((n, h, w, roundup(c, 4)), dtype=torch.float32, device=dev)
        rt.call("selfc_nchw_to_nhwc4", xs.data_ptr(), nhwc.data_ptr(), n, c, h, w, sp)
        dense = torch.zeros((pin + 4 * gp, n, h, w, 32), dtype=_lib.operand_dtype(), device=dev)
        rt.call("selfc_nhwc_to_planes", nhwc.data_ptr(), dense.data_ptr(), n * h * w, c, sp)
        for i in range(4):          # conv1 (3,3,3), conv2-4 (1,3,3); LeakyReLU fused, features appended as planes
            wp, bp = pk[i]
            nin = pin + i * gp
            rt.call("selfc_conv_planes_run", dense.data_ptr(), nin, 3 if i == 0 else 1, wp.data_ptr(), bp.data_ptr(),
                    self.gc, nin, None, n, t, h, w, sp)
        wp, bp = pk[4]
        out = torch.empty((n, h, w, self.cout), dtype=torch.float32, device=dev)
        rt.call("selfc_conv_planes_run", dense.data_ptr(), pin + 4 * gp, 3, wp.data_ptr(), bp.data_ptr(),
                self.cout, -1, out.data_ptr(), n, t, h, w, sp)
        y = torch.empty((n, self.cout, h, w), dtype=torch.float32, device=dev)
        rt.call("selfc_nhwc4_to_nchw", out.data_ptr(), y.data_ptr(), n, self.cout, h, w, sp)
        return y, dense

    def forward(self, x, io_type="2d"):
        _lib.require_gpu(x)
        if io_type != "2d":
            # any other io_type: x is already (b, c, t, h, w) (Subnet_constructor.py:304-320 skips its reshapes).  SpaceToDepth
            # unpacks four sizes (:247), so the reference itself only admits this call with scale == 1
            if self.scale > 1:
                raise ValueError("too many values to unpack (expected 4)")       # what SpaceToDepth.forward raises on a 5-D input
            b, c, t, h, w = x.shape
            y = self._forward_frames(x.transpose(1, 2).reshape(b * t, c, h, w), t)
            return y.reshape(b, t, -1, h, w).transpose(1, 2)          # the residual (is_res) was added on the frame view
        t = GlobalVar.get_Temporal_LEN() or 7
        return self._forward_frames(x, t)

    def _forward_frames(self, x, t):
        res = x
        xs = self.ds(x) if self.scale > 1 else x                     # SpaceToDepth / PixelShuffle: torch view ops (differentiable)
        n, c, h, w = xs.shape
        if c != self.cin or n % t:
            raise RuntimeError(f"FeatureCalapseBlock expects (b*{t},{self.cin // self.scale ** 2},H,W), got {tuple(x.shape)}")
        from .. import autograd as ag
        if ag.module_needs_grad(xs, self):
            y = ag.FCBFn.apply(xs, self, t, *ag.subnet_params(self))
        else:
            y, _ = self._run_planes(rt.as_input(xs), t)
        y = self.us(y) if self.scale > 1 else y
        return y + res if self.is_res else y


def subnet(net_structure, init='xavier'):
    """String-keyed factory, Subnet_constructor.py:719-788.  Live names: 'DBNet', 'D2DTNet' and 'FeatureCalapseBlock'
    (:731-735); any other name returns None exactly like the reference (it then fails at first use)."""
    def constructor(channel_in, channel_out, gc=32):
        if net_structure == 'DBNet':
            return DenseBlock(channel_in, channel_out, init) if init == 'xavier' else DenseBlock(channel_in, channel_out)
        if net_structure == 'FeatureCalapseBlock':
            # argument positions as in the reference (:732-735): with init == 'xavier' the string lands in the `scale`
            # slot and the constructor raises TypeError there too; any other init builds the scale-4 block
            return FeatureCalapseBlock(channel_in, channel_out, init) if init == 'xavier' else FeatureCalapseBlock(channel_in, channel_out)
        if net_structure == 'D2DTNet':
            if init == 'xavier':
                return D2DTInput(channel_in, channel_out, init, gc=gc)
            return D2DTInput(channel_in, channel_out)
        return None

    return constructor
