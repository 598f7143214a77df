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
from .module_util import HeadOutput, cache_free_state
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
        from .. import autograd as ag
        if ag.needs_grad(x):
            if self.k != 4:
                raise NotImplementedError("FrequencyAnalyzer backward is built for k = 4")
            return ag.FreqFn.apply(x, self, bool(rev))
        return self._run(x, rev)

    def _run(self, x, rev=False):
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
    __getstate__ = cache_free_state      # deepcopy / pickle leave the runtime's caches behind (module_util)
    """Temporal TxT attention over globally pooled descriptors (:257-285) on the kernels of
    csrc/stp.hip: weighted pooling (fc folded through adaptive_avg_pool2d into one HxW map),
    a tiny per-clip attention kernel, and a fused temporal-mix + 1x1 projection + residual."""

    #: frames per clip; None = GlobalVar.get_Temporal_LEN() (SelfC_GMM_arch_inv.py:276).  The codec variant's copy of this
    #: class uses the module constant TEMP_LEN = 3 instead (SelfC_Codec_arch_inv.py:77,118).
    TEMP_LEN = None

    def __init__(self, c):
        super().__init__()
        if not 1 <= c <= 64:
            raise NotImplementedError("selfc_amd GlobalAgg kernels hold c <= 64 channels (64 in SelfC-large, 24 in the codec variant)")
        self.c = c
        self.fc = nn.Linear(32 * 32, 1)
        self.proj1 = nn.Conv2d(c, c, 1, 1, 0)
        self.proj2 = nn.Linear(c, c)
        self.proj3 = nn.Linear(c, c)

    def _gather_params(self):
        return [self.fc.bias, self.proj1.weight, self.proj1.bias, self.proj2.weight, self.proj2.bias, self.proj3.weight, self.proj3.bias]

    def _gather_entries(self, ps):
        """everything of the packed set that is a pure re-ordering of the parameters ({name: (tensor, kind)}, packing.PackPlan terms);
        `wmap` (a weighted sum of fc.weight) is not"""
        from ..packing import gagg_row_perm, pack_planes_generic, pack_pointwise, pad_bias
        c = self.c
        fcb, w1, b1, w2, b2, w3, b3 = ps

        def sq64(wt):        # (c,c[,1,1]) -> zero-padded (64,64): the kernels' rows are 64 channels wide, pads stay 0
            out = torch.zeros(64, 64, dtype=torch.float32, device=wt.device)
            out[:c, :c] = wt.detach().float().reshape(c, c)
            return out
        e = dict(fcb=(fcb.detach().float().contiguous(), "b"), w1=(pack_pointwise(sq64(w1)[self._row_perm(w1.device)]), "w"), b1=(pad_bias(b1, 64), "b"),
                 w2=(sq64(w2), "b"), b2=(pad_bias(b2, 64), "b"), w3=(sq64(w3), "b"), b3=(pad_bias(b3, 64), "b"))
        if c == 64:          # the gradient kernels' transposed proj1 (autograd.globalagg_bwd)
            e["w1t"] = (pack_planes_generic(w1.detach().float().reshape(64, 64).t().reshape(64, 64, 1, 1).contiguous()), "w")
        return e

    def _row_perm(self, dev):
        """packing.gagg_row_perm on the weights' device, made once (an index tensor's host -> device copy cannot be captured)"""
        cache = self.__dict__.setdefault("_row_perms", {})
        if str(dev) not in cache:
            from ..packing import gagg_row_perm
            cache[str(dev)] = gagg_row_perm(dev)
        return cache[str(dev)]

    def _install_gathered(self, d):
        """runtime.PackGroup: the gather parts of this module arrived with the group's refresh"""
        self._pkg, self._pkg_key = d, rt.params_key(self)
        if "w1t" in d:
            self._w1t, self._w1t_key = d["w1t"], self._pkg_key

    def _packed(self, h, w):
        pkey = rt.params_key(self)
        key = pkey + (h, w)
        if getattr(self, "_pk_key", None) != key:
            from ..packing import pool_weight_map, pool_weight_map_batch
            sib = self.__dict__.get("_siblings")          # the GlobalAgg blocks of one STP chain fold their pooling maps together
            fkey = (rt.params_key(self.fc), h, w)
            if sib and self.__dict__.get("_wmap_key") != fkey:
                maps = pool_weight_map_batch([m.fc.weight for m in sib], h, w)
                for i, m in enumerate(sib):
                    m.__dict__["_wmap"], m.__dict__["_wmap_key"] = maps[i], (rt.params_key(m.fc), h, w)
            if getattr(self, "_pkg_key", None) == pkey:
                g = {k: v for k, v in self._pkg.items() if k != "w1t"}
            else:
                g = {k: v[0] for k, v in self._gather_entries(self._gather_params()).items() if k != "w1t"}
            wmap = self.__dict__["_wmap"] if self.__dict__.get("_wmap_key") == fkey else pool_weight_map(self.fc.weight, h, w)
            self._pk = dict(g, wmap=wmap)
            self._pk_key = key
        return self._pk

    def run_nhwc(self, x, y, n, t, h, w, scratch, dense_out=None):
        """x, y: fp32 NHWC [n][h*w][64] (distinct buffers); scratch: dict cache for the partial sums.
        dense_out (instead of y): the f16 operand buffer [planes][n][h*w][32] of the D2DTInput that consumes the result -
        its two input planes are written directly and that subnet runs with xin = NULL."""
        pk = self._packed(h, w)
        L = _lib.lib()
        need = L.selfc_globalagg_partial_floats(n, h * w)
        buf = scratch.get("gagg_partial")        # grows with the call; never outlives a device change
        if buf is None or buf.numel() < need or buf.device != x.device:
            scratch["gagg_partial"] = torch.empty(need, dtype=torch.float32, device=x.device)
        rt.call("selfc_globalagg_run_d", x.data_ptr(), None if y is None else y.data_ptr(),
                None if dense_out is None else dense_out.data_ptr(), pk["wmap"].data_ptr(), pk["fcb"].data_ptr(),
                pk["w1"].data_ptr(), pk["b1"].data_ptr(), pk["w2"].data_ptr(), pk["b2"].data_ptr(),
                pk["w3"].data_ptr(), pk["b3"].data_ptr(), scratch["gagg_partial"].data_ptr(), None,
                n, t, h * w, self.c, _lib.stream_ptr())

    def forward(self, x):
        x = rt.as_input(x)
        t = self.TEMP_LEN or GlobalVar.get_Temporal_LEN()
        n, c, h, w = x.shape
        if not t or n % t or c != self.c:
            raise RuntimeError(f"GlobalAgg expects (b*T,{self.c},h,w) with T={t!r}, got {tuple(x.shape)}")
        from .. import autograd as ag
        if ag.module_needs_grad(x, self):
            if c != 64:
                # c < 64 (the codec variant: 24): train the exactly equivalent 64-channel module on zero-padded rows (shadow.py)
                from .. import shadow
                sh = shadow.globalagg_shadow(self)
                return shadow.shadow_apply(x, sh, lambda xw: ag.GlobalAggFn.apply(F.pad(xw, (0, 0, 0, 0, 0, 64 - c)), sh.wide, t,
                                                                                  *rt.plist(sh.wide))[:, :c])
            return ag.GlobalAggFn.apply(x, self, t, *rt.plist(self))
        sp = _lib.stream_ptr()
        if c != 64:                    # zero-pad the rows to the kernels' 64 channels (host-side view ops, not a hot path)
            xin = torch.zeros((n, h, w, 64), dtype=torch.float32, device=x.device)
            xin[..., :c] = x.permute(0, 2, 3, 1)
            yout = torch.empty_like(xin)
            self.run_nhwc(xin, yout, n, t, h, w, self.__dict__.setdefault("_scratch", {}))
            return yout[..., :c].permute(0, 3, 1, 2).contiguous()
        xin = torch.empty((n, h, w, 64), dtype=torch.float32, device=x.device)
        rt.call("selfc_nchw_to_nhwc4", x.data_ptr(), xin.data_ptr(), n, 64, h, w, sp)
        yout = torch.empty_like(xin)
        self.run_nhwc(xin, yout, n, t, h, w, self.__dict__.setdefault("_scratch", {}))
        y = torch.empty_like(x)
        rt.call("selfc_nhwc4_to_nchw", yout.data_ptr(), y.data_ptr(), n, 64, h, w, sp)
        return y


class STPNet(nn.Module):
    __getstate__ = cache_free_state      # deepcopy / pickle leave the runtime's caches behind (module_util)
    """Self-conditioned latent predictor v2 (:289-430): 6 x [D2DTInput + GlobalAgg] + 1x1x1 MLP
    head; GMM sample v = sum_k pi*(eps*exp(clamp(logsigma,-7,7)) + mu) with pi = softmax over the
    hf_dim axis (trap 6).  Runs end to end on HIP kernels in fp32 NHWC (``run_nhwc``): the dense
    blocks through selfc_subnet_run, GlobalAgg / head / sampler through csrc/stp.hip.  Noise is
    torch's device Philox stream unless ``self.eps`` (b, hf_dim, K, t, h, w) is injected."""

    def __init__(self, opt):
        super().__init__()
        self.global_module = opt["global_module"]
        self.fh_loss = opt["fh_loss"]
        self.scale = opt["scale"]
        self.K = opt["gmm_k"]
        self.stp_blk_num = opt["stp_blk_num"] - 2
        c = self.c = 64
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

    # -- native pipeline ------------------------------------------------------------------
    def _chain(self):
        mods = [self.local_m1]
        if self.global_module:
            mods.append(self.global_m1)
        mods.append(self.local_m2)
        if self.global_module:
            mods.append(self.global_m2)
        return mods + list(self.other_stp_modules)

    def _tail_seq(self):
        """the head's nn.Sequential (`tail_gmm` here, `tail` in the codec variant's STPNet)"""
        return self.tail_gmm

    def _needs_shadow(self) -> bool:
        """narrower than the kernels' native widths (64-channel rows, growth 32)?"""
        return self.c != 64 or any(isinstance(m, D2DTInput) and m.gc != 32 for m in self._chain())

    def _virt(self, m):
        """(cin, cout) the chain's kernels see for subnet m: features travel in 64-channel fp32 rows whatever c is"""
        return (m.channel_in if m.channel_in <= 3 else 64, 64)

    def _tail_packed(self):
        """[(fragments, bias, cin, cout)] of the head's pointwise convs in the kernels' terms: the l2 head as it is (cout =
        hf_dim); a GMM head with every width rounded up to what selfc_pwconv_run takes - input rows of 64 channels (the chain's
        rows, zero beyond c), hidden widths to 32 / 64 / 128 / 256, the last layer to a multiple of 16 - by zero rows / columns, which is
        exact (padded hidden units are act(0 + 0) = 0 and meet zero columns).  c = 64 with hf_dim = 48 (SelfC-large) needs no pad."""
        convs = self._tail_convs()
        key = rt.params_key(*convs)
        if getattr(self, "_tail_key", None) != key:
            self._install_tail({k: v[0] for k, v in self._tail_entries([p_ for m in convs for p_ in (m.weight, m.bias)]).items()})
        return self._tail

    def _tail_convs(self):
        return [m for m in self._tail_seq() if isinstance(m, nn.Conv3d)]

    def _tail_widths(self):
        """[(cin_p, cout_p)] per conv of the head: the padded widths documented at _tail_packed"""
        from ..packing import roundup
        convs = self._tail_convs()
        out, prev = [], 64
        for i, m in enumerate(convs):
            if i == len(convs) - 1:
                cout_p = m.out_channels if len(convs) == 1 else roundup(m.out_channels, 16)
            else:         # a hidden width is the next layer's K: the pointwise kernel takes 32, 64, 128 or 256 input channels
                fits = [v for v in (32, 64, 128, 256) if v >= m.out_channels]
                if not fits:
                    raise NotImplementedError("selfc_amd: hidden layers of the STP head hold at most 256 channels")
                cout_p = fits[0]
            out.append((prev, cout_p))
            prev = cout_p
        return out

    def _tail_entries(self, ps):
        """{name: (tensor, kind)} of the head's packed convs from ps = [w0, b0, w1, b1, ..] (packing.PackPlan terms)"""
        from ..packing import pack_pointwise, pad_bias, roundup
        e = {}
        for i, (m, (cin_p, cout_p)) in enumerate(zip(self._tail_convs(), self._tail_widths())):
            wt = torch.zeros(max(cout_p, m.out_channels), cin_p, dtype=torch.float32, device=ps[2 * i].device)
            wt[:m.out_channels, :m.in_channels] = ps[2 * i].detach().float().reshape(m.out_channels, m.in_channels)
            e[f"w{i}"] = (pack_pointwise(wt), "w")
            e[f"b{i}"] = (pad_bias(ps[2 * i + 1], roundup(cout_p, 16)), "b")
        return e

    def _install_tail(self, d):
        self._tail = [(d[f"w{i}"], d[f"b{i}"], cin_p, cout_p) for i, (cin_p, cout_p) in enumerate(self._tail_widths())]
        self._tail_fused = self._tail_fused_key = None
        self._tail_key = rt.params_key(*self._tail_convs())

    def add_to_pack_group(self, group):
        """runtime.PackGroup membership of everything this net packs per set of weights: the chain's subnets (as the kernels see
        them), the GlobalAgg blocks' gather parts, the head."""
        from ..packing import pack_pointwise_T
        aggs = [m for m in self._chain() if not isinstance(m, D2DTInput)]
        for m in self._chain():
            if isinstance(m, D2DTInput):
                rt.group_add_subnet(group, m, self._virt(m))
            else:
                m.__dict__["_siblings"] = aggs
                group.add(m._gather_params(), m._gather_entries, m._install_gathered)
        convs = self._tail_convs()
        group.add([p_ for m in convs for p_ in (m.weight, m.bias)], self._tail_entries, self._install_tail)

        def install_t(d):          # the transposed head weights of autograd._head_bwd
            for i, m in enumerate(convs):
                m.__dict__["_wt_pk"], m.__dict__["_wt_key"] = d[f"wt{i}"], rt.params_key(m)
        group.add([m.weight for m in convs], lambda ps: {f"wt{i}": (pack_pointwise_T(p_), "w") for i, p_ in enumerate(ps)}, install_t)

    def _head_fused(self):
        """(fragment stream, bias vector) of the whole-head + sampler kernel, or None when the head is not the shipped
        64 -> 128 -> 256 -> 720 GMM head.  Built on first use per set of weights (the training path never asks for it)."""
        convs = [m for m in self._tail_seq() if isinstance(m, nn.Conv3d)]
        if not (self.fh_loss == "gmm" and len(convs) == 3 and self.hf_dim == 48 and self.K == 5
                and [(m.in_channels, m.out_channels) for m in convs] == [(64, 128), (128, 256), (256, 720)]):
            return None
        self._tail_packed()
        if self._tail_fused_key != self._tail_key:
            # hidden layers with their output rows in operand order, the last layer's output channels as
            # [k][pi | log-sigma | mu][c]; one fragment stream, one bias vector
            from ..packing import gmm_head_perm, head_row_perm, pack_pointwise
            dev = convs[0].weight.device
            perms = self.__dict__.setdefault("_head_perms", {}).get(str(dev))
            if perms is None:            # index tensors: made once per device (a host -> device copy cannot be captured)
                perms = self._head_perms[str(dev)] = [head_row_perm(128, dev), head_row_perm(256, dev), gmm_head_perm(self.hf_dim, self.K, dev)]
            ws = [pack_pointwise(m.weight.detach().reshape(m.out_channels, -1)[pm]).reshape(-1) for m, pm in zip(convs, perms)]
            bs = [m.bias.detach().float()[pm] for m, pm in zip(convs, perms)]
            self._tail_fused = (torch.cat(ws).contiguous(), torch.cat(bs).contiguous())
            self._tail_fused_key = self._tail_key
        return self._tail_fused

    def run_nhwc(self, x1, hf_out, n, t, h, w, keep_raw=False, scratch=None, eps=None):
        """x1: fp32 NHWC4 [n][h*w][4] (LR frames); hf_out: fp32 [n][h*w][hf_dim] (e.g. the latent x2
        buffer).  Returns the raw head output [n][h*w][Cp] when keep_raw (else None).
        scratch: a caller-owned dict for the intermediate buffers (one per stream when several calls overlap);
        eps: pre-allocated noise rows [n*h*w][hf_dim*K] to fill in place (hipGraph capture) instead of a fresh randn."""
        dev, sp = x1.device, _lib.stream_ptr()
        sc = scratch if scratch is not None else self.__dict__.setdefault("_scratch", {})
        shape_key = (n, h, w, str(dev))
        if sc.get("key") != shape_key:
            sc.clear()
            sc["key"] = shape_key
            sc["feat"] = [torch.empty((n, h * w, 64), dtype=torch.float32, device=dev) for _ in range(2)]
            sc["dense4"] = torch.zeros((4, n, h, w, 32), dtype=_lib.operand_dtype(), device=dev)
            sc["dense6"] = torch.zeros((6, n, h, w, 32), dtype=_lib.operand_dtype(), device=dev)
        cur, nxt = None, 0
        chain = self._chain()
        in_dense = False          # the previous GlobalAgg wrote this subnet's input planes itself
        for i, m in enumerate(chain):
            dst = sc["feat"][nxt]
            if isinstance(m, D2DTInput):
                src = x1 if cur is None else sc["feat"][cur]
                dense = sc["dense4"] if m.channel_in <= 3 else sc["dense6"]
                cin_v, cout_v = self._virt(m)
                sw = rt.packed_subnet(m, (cin_v, cout_v)).struct()
                rt.call("selfc_subnet_run", sw, m.kind, None if in_dense else src.data_ptr(), dst.data_ptr(), dense.data_ptr(),
                        n, t, h, w, cin_v, cout_v, sp)
                in_dense = False
            else:
                # a GlobalAgg in front of a wide D2DTInput hands over f16 operand planes instead of an fp32 row (the same
                # rounding, one conversion pass and half the store traffic less); the last one feeds the head in fp32
                in_dense = i + 1 < len(chain) and isinstance(chain[i + 1], D2DTInput) and chain[i + 1].channel_in > 3
                if in_dense:
                    m.run_nhwc(sc["feat"][cur], None, n, t, h, w, sc, dense_out=sc["dense6"])
                    continue              # nothing new in feat: cur / nxt stay
                m.run_nhwc(sc["feat"][cur], dst, n, t, h, w, sc)
            cur, nxt = nxt, 1 - nxt
        feat = sc["feat"][cur]
        npix = n * h * w
        tail = self._tail_packed()
        if self.fh_loss == "l2":
            wp, bp, cin, cout = tail[0]
            if cout % 16:             # whole 16-channel tiles are stored: go through a padded row, then narrow (scale 2: 12)
                c16 = (cout + 15) // 16 * 16
                if "hf16" not in sc:
                    sc["hf16"] = torch.empty((npix, c16), dtype=torch.float32, device=dev)
                rt.call("selfc_pwconv_run", feat.data_ptr(), 1, sc["hf16"].data_ptr(), 1, wp.data_ptr(), bp.data_ptr(),
                        npix, cin, c16, c16, 1, 0, sp)
                hf_out.view(npix, -1)[:, :cout].copy_(sc["hf16"][:, :cout])
            else:
                rt.call("selfc_pwconv_run", feat.data_ptr(), 1, hf_out.data_ptr(), 1, wp.data_ptr(), bp.data_ptr(),
                        npix, cin, cout, cout, 1, 0, sp)
            return hf_out if keep_raw else None
        (w0, b0, ci0, co0), (w1_, b1_, ci1, co1), (w2_, b2_, ci2, co2) = tail
        head = None if keep_raw else self._head_fused()             # sampling path: no activation of the head is ever written
        fused = head is not None
        if not fused:
            if "h1" not in sc:
                sc["h1"] = torch.empty((npix, tail[0][3]), dtype=_lib.operand_dtype(), device=dev)
                sc["h2"] = torch.empty((npix, tail[1][3]), dtype=_lib.operand_dtype(), device=dev)
                # 720 fp32 channels per pixel-frame (578 MB at 4 x 7 x 64 x 112): only when somebody wants it
                sc["raw"] = torch.empty((npix, tail[2][3]), dtype=torch.float32, device=dev)
            # tail_gmm = [lrelu, conv, act, conv, act, conv]: each activation is fused into the producer's epilogue (act =
            # LeakyReLU for 'gmm', ReLU for 'gmm_thin', :334-354; activation flag 1 / 2 of selfc_pwconv_run)
            act = 2 if self.fh_loss == "gmm_thin" else 1
            rt.call("selfc_pwconv_run", feat.data_ptr(), 1, sc["h1"].data_ptr(), 0, w0.data_ptr(), b0.data_ptr(), npix, ci0, co0, co0, 1, act, sp)
            rt.call("selfc_pwconv_run", sc["h1"].data_ptr(), 0, sc["h2"].data_ptr(), 0, w1_.data_ptr(), b1_.data_ptr(), npix, ci1, co1, co1, 0, act, sp)
            rt.call("selfc_pwconv_run", sc["h2"].data_ptr(), 0, sc["raw"].data_ptr(), 1, w2_.data_ptr(), b2_.data_ptr(), npix, ci2, co2, co2, 0, 0, sp)
        # noise rows: [npix][c*K + k] for selfc_gmm_sample, [npix][k*hf_dim + c] for the fused kernel
        if self.eps is not None:
            b = n // t
            e6 = self.eps.reshape(b, self.hf_dim, self.K, t, h, w)
            eps = (e6.permute(0, 3, 4, 5, 2, 1) if fused else e6.permute(0, 3, 4, 5, 1, 2)).reshape(npix, self.hf_dim * self.K)
            eps = eps.to(device=dev, dtype=torch.float32).contiguous()
        elif eps is not None:
            eps.normal_()
        else:
            eps = torch.randn((npix, self.hf_dim * self.K), dtype=torch.float32, device=dev)
        if fused:
            wf, bfz = head
            rt.call("selfc_stp_head_gmm", feat.data_ptr(), wf.data_ptr(), bfz.data_ptr(), eps.data_ptr(), hf_out.data_ptr(),
                    npix, self.hf_dim, self.K, hf_out.shape[-1], 1, sp)
            return None
        creal = self.hf_dim * self.K * 3
        if self.hf_dim == 48 and co2 == creal and self.K in (1, 3, 5):
            rt.call("selfc_gmm_sample", sc["raw"].data_ptr(), eps.data_ptr(), hf_out.data_ptr(), npix, self.hf_dim, self.K, sp)
        else:         # any other scale / mixture size (hf_dim = 3 scale^2): the one-thread-per-pixel sampler, padded raw rows
            rt.call("selfc_gmm_sample_generic", sc["raw"].data_ptr(), eps.data_ptr(), hf_out.data_ptr(), npix, self.hf_dim, self.K,
                    co2, hf_out.shape[-1], 1.0, sp)
        if not keep_raw:
            return None
        return sc["raw"] if co2 == creal else sc["raw"][:, :creal]

    def _eps_rows(self, n, t, h, w, dev):
        """Injected noise (b, hf_dim, K, t, h, w) as kernel rows [npix][hf_dim*K], or None (device RNG)."""
        if self.eps is None:
            return None
        eps = self.eps.reshape(n // t, self.hf_dim, self.K, t, h, w).permute(0, 3, 4, 5, 1, 2).reshape(n * h * w, self.hf_dim * self.K)
        return eps.to(device=dev, dtype=torch.float32).contiguous()

    # -- reference-shaped API ---------------------------------------------------------------
    def forward(self, x):
        """x (b,3,t,h,w); side effects as in the reference: ``stp_parameters`` (the reference's
        ``self.parameters``, (b,Cp,t,h,w)) and, for GMM heads, ``gmm_v`` (b,hf_dim,t,h,w)."""
        b, c, t, h, w = x.size()
        xf = rt.as_input(x.transpose(1, 2).reshape(b * t, c, h, w))
        from .. import autograd as ag
        if ag.module_needs_grad(xf, self):
            # training: one differentiable op for chain + head + sample; `stp_parameters` (the raw head output) is only
            # exposed for the l2 head here - the reference's GMM likelihood path (neg_llh) is not used by its trainer
            if self._needs_shadow():
                # hidden width < 64 and / or dense growth < 32 (the codec variant): the native-width shadow net (shadow.py)
                from .. import shadow
                sh = shadow.stp_shadow(self)
                v = shadow.shadow_apply(xf, sh, lambda xw: ag.STPSampleFn.apply(xw, sh.wide, t, None, *rt.plist(sh.wide))[:, :self.hf_dim])
            else:
                v = ag.STPSampleFn.apply(xf, self, t, self._eps_rows(b * t, t, h, w, xf.device), *rt.plist(self))
            v5 = v.reshape(b, t, -1, h, w).transpose(1, 2)
            if self.fh_loss == "l2":
                self._publish(v5)
            else:
                self.gmm_v = v5
            return
        n, sp = b * t, _lib.stream_ptr()
        x1 = torch.empty((n, h, w, 4), dtype=torch.float32, device=xf.device)
        rt.call("selfc_nchw_to_nhwc4", xf.data_ptr(), x1.data_ptr(), n, 3, h, w, sp)
        hf = torch.empty((n, h * w, self.hf_dim), dtype=torch.float32, device=xf.device)
        raw = self.run_nhwc(x1, hf, n, t, h, w, keep_raw=True)
        to5d = lambda a: a.reshape(b, t, h, w, -1).permute(0, 4, 1, 2, 3)  # noqa: E731
        self._publish(to5d(raw))
        if self.fh_loss != "l2":
            self.gmm_v = to5d(hf)

    def _publish(self, raw5d):
        """The raw head output (b,Cp,t,h,w) under both names: ``stp_parameters`` and - exactly as the reference does at
        :377 - ``parameters``, which shadows nn.Module.parameters on this instance (callers such as the reference's
        ``neg_llh`` read ``stp_net.parameters`` as a tensor; nothing in selfc_amd calls ``stp_net.parameters()``)."""
        self.stp_parameters = raw5d
        self.parameters = HeadOutput.wrap(raw5d, self)     # a tensor, as in the reference - and still callable (module_util.HeadOutput)

    @property
    def gmm(self):
        """torch.distributions mixture of the likelihood path (:396-411), built on demand from the raw head output; note
        its index convention differs from sampling (:399-405): mean = idx1, log-sigma = idx2."""
        p = self.stp_parameters
        b, _, t, h, w = p.shape
        p = p.reshape(b, self.hf_dim, self.K, 3, t, h, w).permute(0, 1, 4, 5, 6, 2, 3).reshape(-1, self.K, 3)
        mix = torch.distributions.Categorical(F.softmax(p[:, :, 0], dim=1))
        comp = torch.distributions.Normal(p[:, :, 1], torch.exp(torch.clamp(p[:, :, 2], -7, 7)))
        return torch.distributions.MixtureSameFamily(mix, comp)

    def reparametrize(self, mu, logvar):
        """eps * exp(logvar) + mu with eps ~ N(0,1) drawn on mu's device (:410-417; the reference allocates it with
        torch.cuda.FloatTensor, trap 5).  Host-side helper kept for the reference's signature - the sampling path of
        forward() runs inside selfc_stp_head_gmm / selfc_gmm_sample."""
        eps = self.eps.to(mu) if self.eps is not None and self.eps.shape == mu.shape else torch.randn_like(mu)
        return eps.mul(torch.exp(logvar)).add_(mu)

    def neg_llh(self, hf):
        if self.fh_loss == "l2":
            return torch.mean((hf - self.stp_parameters) ** 2)
        return -self.gmm.log_prob(hf.reshape(-1))

    def sample(self):
        return self.stp_parameters if self.fh_loss == "l2" else self.gmm_v


class SelfCInvNet(nn.Module):
    __getstate__ = cache_free_state      # deepcopy / pickle leave the runtime's caches behind (module_util)
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
        from .. import autograd as ag
        if ag.module_needs_grad(x, self):
            return self._forward_train(x, rev)
        # eval / no_grad: the cached two-stream hipGraph of this call (pipeline.ModuleGraph) from its second use on; the eager
        # single-stream path below is the first-use / fallback path and computes the same thing with the same kernels
        from ..pipeline import module_graph
        k = self.operations[0].k
        if not rev:
            n, c, H, W = x.shape
            if c != 3 or H % k or W % k:
                raise RuntimeError(f"SelfCInvNet forward expects (N,3,{k}a,{k}b), got {tuple(x.shape)}")
            mg = module_graph(self, "fwd", n, H // k, W // k, x.device)
            if mg is not None:
                out = mg(x)
                return out, out.new_zeros(())
        else:
            n, c, h, w = x.shape
            if c < 3:
                raise RuntimeError(f"SelfCInvNet reverse expects the 3 LR channels, got {tuple(x.shape)}")
            mg = module_graph(self, "rev", n, h, w, x.device)
            if mg is not None:
                return mg(x if c == 3 else x[:, 0:3].contiguous())
        sp = _lib.stream_ptr()
        arr, nblk = self._stack()
        if not rev:
            ws = self._workspace(x, n, H // k, W // k)
            rt.call("selfc_freq_fwd", x.data_ptr(), ws.x1.data_ptr(), ws.x2.data_ptr(), ws.fd.data_ptr(), ws.FC, n, H, W, k, sp)
            lat = ws.latent()
            rt.call("selfc_invstack_run", arr, nblk, lat, 0, sp)
            out = rt.latent_to_nchw(ws)
            return out, out.new_zeros(())          # loss_c = out.mean()*0 (:466)
        n, c, h, w = x.shape
        if c < 3:
            raise RuntimeError(f"SelfCInvNet reverse expects the 3 LR channels, got {tuple(x.shape)}")
        lr = x[:, 0:3].contiguous()
        ws = self._workspace(x, n, h, w)
        # LR frames -> latent x1; STP predicts the HF channels straight into the latent x2 buffer (:475-485)
        rt.call("selfc_nchw_to_nhwc4", lr.data_ptr(), ws.x1.data_ptr(), n, 3, h, w, sp)
        self.stp_net.run_nhwc(ws.x1, ws.x2, n, ws.T, h, w)
        recon_hf = torch.empty((n, ws.c2, h, w), dtype=torch.float32, device=x.device)
        rt.call("selfc_nhwc4_to_nchw", ws.x2.data_ptr(), recon_hf.data_ptr(), n, ws.c2, h, w, sp)
        lat = ws.latent()
        rt.call("selfc_invstack_run", arr, nblk, lat, 1, sp)
        out = torch.empty((n, 3, h * k, w * k), dtype=torch.float32, device=x.device)
        rt.call("selfc_freq_inv", ws.x1.data_ptr(), ws.x2.data_ptr(), out.data_ptr(), n, h, w, k, sp)
        return out, recon_hf

    def _forward_train(self, x, rev):
        """The reference's op loops (:452-490) composed from the differentiable boundary ops (autograd.py): same HIP
        kernels as inference, one block per call so that each keeps its buffers for the backward."""
        from .. import autograd as ag
        t = GlobalVar.get_Temporal_LEN()
        if not t or x.shape[0] % t:
            raise RuntimeError(f"GlobalVar temporal length {t!r} does not divide the {x.shape[0]} input frames")
        # one op for FrequencyAnalyzer + every block (state stays in the latent layout between blocks) where its kernels
        # apply: k = 4 (the split's adjoint kernels), channel split <= 3; otherwise the op loop, one differentiable op per module
        blocks = self._blocks()
        stack = self.operations[0].k == 4 and all(b.split_len1 <= 3 for b in blocks) and len(blocks) == len(self.operations) - 1
        prm = [p for b in blocks for p in ag.block_params(b)] if stack else None
        if not rev:
            if stack:
                out = ag.InvStackFn.apply(x, self, False, t, *prm)
                return out, out.new_zeros(())
            out = x
            for op in self.operations:
                out = op.forward(out, False)
            return out, out.new_zeros(())
        n, c, h, w = x.shape
        lr = x[:, 0:3]
        stp = self.stp_net
        recon_hf = ag.STPSampleFn.apply(lr, stp, t, stp._eps_rows(n, t, h, w, x.device), *rt.plist(stp))
        out = torch.cat((lr, recon_hf), dim=1)
        if stack:
            return ag.InvStackFn.apply(out, self, True, t, *prm), recon_hf
        for op in reversed(self.operations):
            out = op.forward(out, True)
        return out, recon_hf

    def inverse_from_latent(self, z):
        """Reversed op loop (:486-489) on a full (N,51,h,w) latent, STP bypassed - the
        'inv' half of BASELINE.json's metric."""
        z = rt.as_input(z)
        n, c, h, w = z.shape
        if not (torch.is_grad_enabled() and z.requires_grad):
            from ..pipeline import module_graph
            mg = module_graph(self, "revlat", n, h, w, z.device) if c == 3 + self._blocks()[0].split_len2 else None
            if mg is not None:
                return mg(z)
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
