"""Codec variant of SelfC (model "SelfC_GMM_Codec") on MI355X: the invertible rescaler that sits around an external
H.265 encoder.

Mirrors the compute of codes/models/modules/SelfC_Codec_arch_inv.py: ``FrequencyAnalyzer(k=2)`` (:78-98) +
``sum(block_num)`` x ``InvBlockExp(15 | 3)`` (:24-57, 379-416) + the narrow ``STPNet`` (:234-376: hidden width
``stp_hidden_c`` = 24, dense-block growth ``stp_denseblock_innerc`` = 12, ``GlobalAgg`` over the module constant
TEMP_LEN = 3) and the segmenting / tiling of ``forward_test`` (:502-640): 3-frame segments (``seg_add_pad``,
utils/util.py:329-354), two column strips on the way down, 2 x 2 tiles (no halo) on the way up.

Same kernels as SelfC-large, different shapes: the 12-channel dense growth and the 24-channel features are zero-padded
into the kernels' 32 / 64-channel layouts at weight-packing time (packing.widen_dense_params - exact, the padded
channels stay 0).  NOT rebuilt: the H.265 / BPG / surrogate-codec quantisers (ffmpeg / libx265 subprocesses,
SURVEY section 2 rows 13-14).  The compressed-domain round trip is a caller-supplied ``lr_codec`` callable
(frames (n,3,h,w) in [0,1] -> decoded frames); without one the 8-bit ``Quantization`` is all that happens to the LR
video.  Inference only (the reference trains this net through its H.265 surrogate).
"""
import torch
import torch.nn as nn

from .module_util import cache_free_state

from ..global_var import GlobalVar
from .Inv_arch import InvBlockExp
from .Quantization import Quantization
from .SelfC_GMM_arch_inv import FrequencyAnalyzer as _FrequencyAnalyzer
from .SelfC_GMM_arch_inv import GlobalAgg as _GlobalAgg
from .SelfC_GMM_arch_inv import STPNet as _STPNetV2
from .Subnet_constructor import D2DTInput, subnet

TEMP_LEN = 3        # SelfC_Codec_arch_inv.py:77


class FrequencyAnalyzer(_FrequencyAnalyzer):
    """k defaults to 2 in this file (:79)."""

    def __init__(self, channel_in, k=2):
        super().__init__(channel_in, k)


class GlobalAgg(_GlobalAgg):
    """:103-131 - identical to SelfC-large's except that clips are TEMP_LEN = 3 frames whatever GlobalVar says."""
    TEMP_LEN = TEMP_LEN


class STPNet(_STPNetV2):
    """:234-376.  opt keys: global_module, stp_blk_num, fh_loss, scale, gmm_k, stp_hidden_c, stp_denseblock_innerc.
    State-dict names as in the reference: local_m1/2, global_m1/2, other_stp_modules.{i}, tail.{1[,3,5]}."""

    def __init__(self, opt):
        nn.Module.__init__(self)
        self.global_module = opt["global_module"]
        self.fh_loss = opt["fh_loss"]
        self.scale = opt["scale"]
        self.K = opt["gmm_k"]
        self.stp_blk_num = opt["stp_blk_num"] - 2
        c = self.c = opt["stp_hidden_c"]
        gc = opt["stp_denseblock_innerc"]
        if self.global_module not in (None, 'nonlocal'):
            raise NotImplementedError("selfc_amd covers global_module: nonlocal (the shipped codec configs); "
                                      "the deform aggregators need torchvision.ops.deform_conv2d")
        if not 1 <= c <= 64:
            raise NotImplementedError("stp_hidden_c must be <= 64 (rows of the STP kernels)")
        self.local_m1 = D2DTInput(3, c, gc=gc, INN_init=False)
        self.local_m2 = D2DTInput(c, c, gc=gc, INN_init=False)
        if self.global_module == 'nonlocal':
            self.global_m1 = GlobalAgg(c)
            self.global_m2 = GlobalAgg(c)
        others = []
        for _ in range(self.stp_blk_num):
            others.append(D2DTInput(c, c, gc=gc, INN_init=False))
            if self.global_module == 'nonlocal':
                others.append(GlobalAgg(c))
        self.other_stp_modules = nn.Sequential(*others)
        self.hf_dim = 3 * (self.scale ** 2)
        lre = lambda: nn.LeakyReLU(negative_slope=0.2, inplace=True)  # noqa: E731
        if self.fh_loss == "l2":
            self.tail = nn.Sequential(lre(), nn.Conv3d(c, self.hf_dim, 1, 1, 0, bias=True))
        elif self.fh_loss == "gmm":
            self.tail = nn.Sequential(lre(), nn.Conv3d(c, c * 2, 1, 1, 0, bias=True),
                                      lre(), nn.Conv3d(c * 2, c * 4, 1, 1, 0, bias=True),
                                      lre(), nn.Conv3d(c * 4, self.hf_dim * self.K * 3, 1, 1, 0, bias=True))
        elif self.fh_loss == "gmm_thin":
            self.tail = nn.Sequential(lre(), nn.Conv3d(c, c, 1, 1, 0, bias=True),
                                      nn.ReLU(inplace=True), nn.Conv3d(c, c, 1, 1, 0, bias=True),
                                      nn.ReLU(inplace=True), nn.Conv3d(c, self.hf_dim * self.K * 3, 1, 1, 0, bias=True))
        self.eps = None

    def _tail_seq(self):
        return self.tail


def seg_add_pad(video, seg_len):
    """(b,t,c,h,w) -> ((b,seg_num,seg_len,c,h,w), pad): utils/util.py:329-345.  The pad frames repeat the SECOND-TO-LAST
    frame of the growing tensor (`out_video[:, -2:-1]`), exactly as the reference does."""
    b, t, c, h, w = video.shape
    pad = 0 if t % seg_len == 0 else seg_len - t % seg_len
    for _ in range(pad):
        video = torch.cat((video, video[:, -2:-1]), dim=1)
    return video.reshape(b, -1, seg_len, c, h, w), pad


def seg_remove_pad(video, pad, seg_len):
    """utils/util.py:346-354"""
    b, seg_num, seg_len, c, h, w = video.shape
    if pad == 0:
        return video.reshape(b, -1, c, h, w)
    pre = video[:, :seg_num - 1].reshape(b, (seg_num - 1) * seg_len, c, h, w)
    return torch.cat((pre, video[:, -1, :seg_len - pad]), dim=1)


class SelfCInvNet(nn.Module):
    __getstate__ = cache_free_state      # deepcopy / pickle leave the runtime's caches behind (module_util)
    """:379-416 (constructor), :446-500 (forward_train without the codec surrogates), :502-640 (forward_test).

    ``lr_codec``: optional callable applied to the 8-bit-quantised LR frames where the reference runs its H.265 stream
    writer / reader (:528-566); it is NOT part of the state dict."""

    SEG_LEN = 3            # :507
    DIVIDE_WIDTH = 2       # :508
    DIVIDE_HEIGHT = 2      # :509

    def __init__(self, opt, channel_in, channel_out, subnet_type, block_num, down_num, all_opt=None, lr_codec=None):
        super().__init__()
        operations = [FrequencyAnalyzer(channel_in, 2)]
        current_channel = channel_in * (2 ** 2 + 1)
        sc = subnet(subnet_type, "xavier")
        for i in range(down_num):
            for _ in range(block_num[i]):
                operations.append(InvBlockExp(sc, current_channel, channel_out))
        self.operations = nn.ModuleList(operations)
        self.stp_net = STPNet(opt)
        if opt.get("deart_net"):
            raise NotImplementedError("deart_net uses GroupedGlobalDeformAgg (torchvision.ops.deform_conv2d): not built")
        self.opt, self.all_opt = opt, all_opt
        self.Quantization = Quantization()
        self.lr_codec = lr_codec

    # -- the two directions on one tensor (no tiling): forward_train :446-500 minus the codec surrogates -------------
    def encode(self, x):
        out = x
        for op in self.operations:
            out = op.forward(out, False)
        return out

    def decode(self, lr, t=None):
        """lr (b*t,3,h,w) -> (b*t,3,2h,2w): STP prediction of the HF channels, then the reversed op loop (:480-498)."""
        t = t or GlobalVar.get_Temporal_LEN()
        bt, _, h, w = lr.shape
        b = bt // t
        lr5 = lr[:, 0:3].reshape(b, t, 3, h, w).transpose(1, 2)
        self.stp_net(lr5)
        out = torch.cat((lr5, self.stp_net.sample()), dim=1).transpose(1, 2).reshape(bt, -1, h, w)
        for op in reversed(self.operations):
            out = op.forward(out, True)
        return out

    def _distort(self, lr):
        lr = self.Quantization(lr)
        return self.lr_codec(lr) if self.lr_codec is not None else lr

    def forward(self, x, rev=False, cal_jacobian=False, lr_before_distor=None):
        if GlobalVar.get_Istrain():
            return self.forward_train(x, rev, cal_jacobian, lr_before_distor)
        return self.forward_test(x, rev, cal_jacobian, lr_before_distor)

    def forward_train(self, x, rev=False, cal_jacobian=False, lr_before_distor=None):
        if rev:
            return self.decode(x)
        out = self.encode(x)
        lr = out[:, 0:3]
        zero = torch.zeros(1, device=x.device)
        return lr, self._distort(lr), lr.mean() * 0, zero, zero, zero, zero

    def forward_test(self, x, rev=False, cal_jacobian=False, lr_before_distor=None):
        t_all = GlobalVar.get_Temporal_LEN()
        bt, c, h, w = x.shape
        b = bt // t_all
        video, pad = seg_add_pad(x.reshape(b, t_all, c, h, w), self.SEG_LEN)
        seg_num = video.shape[1]
        GlobalVar.set_Temporal_LEN(self.SEG_LEN)
        try:
            if not rev:
                # two column strips per segment, each through the whole op loop; only the LR channels leave (:537-552)
                outs = []
                for s in range(seg_num):
                    seg = video[:, s].reshape(-1, c, h, w)
                    strips = [self.encode(seg[:, :, :, i * (w // self.DIVIDE_WIDTH):(i + 1) * (w // self.DIVIDE_WIDTH)].contiguous())[:, 0:3]
                              for i in range(self.DIVIDE_WIDTH)]
                    outs.append(self._distort(torch.cat(strips, dim=-1)))
                lr = torch.stack(outs, dim=0)                                   # (seg, b*seg_len, 3, h', w')
                hh, ww = lr.shape[-2:]
                lr = lr.reshape(seg_num, b, self.SEG_LEN, 3, hh, ww).permute(1, 0, 2, 3, 4, 5)
                lr = seg_remove_pad(lr, pad, self.SEG_LEN).reshape(-1, 3, hh, ww)
                zero = torch.zeros(1)
                return lr, lr, zero, zero, zero, zero, None
            # decode: 2 x 2 tiles of every segment, each through STP + the reversed op loop, no halo (:575-626)
            dh, dw = self.DIVIDE_HEIGHT, self.DIVIDE_WIDTH
            hd, wd = h // dh, w // dw
            outs = []
            for s in range(seg_num):
                seg = video[:, s].reshape(-1, c, h, w)[:, 0:3]
                tiles = seg.reshape(-1, 3, dh, hd, dw, wd).permute(2, 4, 0, 1, 3, 5)          # (dh, dw, b*seg_len, 3, hd, wd)
                rec = [[self.decode(tiles[i, j].contiguous(), self.SEG_LEN) for j in range(dw)] for i in range(dh)]
                rows = [torch.cat(r, dim=-1) for r in rec]
                outs.append(torch.cat(rows, dim=-2))                                          # (b*seg_len, 3, H, W)
            hr = torch.stack(outs, dim=0)
            H, W = hr.shape[-2:]
            hr = hr.reshape(seg_num, b, self.SEG_LEN, 3, H, W).permute(1, 0, 2, 3, 4, 5)
            return seg_remove_pad(hr, pad, self.SEG_LEN).reshape(-1, 3, H, W)
        finally:
            GlobalVar.set_Temporal_LEN(t_all)
