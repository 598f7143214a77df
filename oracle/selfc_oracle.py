"""CPU oracle for the SelfC invertible-rescaling hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  The product path (``selfc_amd``) never routes through this file and
fails loudly when its HIP extension is missing.

It is a functional restatement (plain fp32 torch ops on CPU, parameters passed
as flat ``{state_dict-key: tensor}`` dicts) of the reference algorithm; every
function cites the reference file:line it follows (paths relative to
``/root/reference/codes``).  Parity is PINNED: ``tools/make_golden.py`` imports
the real reference modules in the build container and stores their
inputs/outputs under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks
this restatement against those vectors (bit-exact for the index shuffles,
<=1e-6 for the float paths).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]


def _sub(params: Params, prefix: str) -> Params:
    """View of ``params`` restricted to keys under ``prefix.`` (prefix stripped)."""
    if not prefix:
        return params
    p = prefix + "."
    return {k[len(p):]: v for k, v in params.items() if k.startswith(p)}


def lrelu(x: torch.Tensor) -> torch.Tensor:
    # nn.LeakyReLU(negative_slope=0.2): Subnet_constructor.py:16,107
    return torch.where(x >= 0, x, x * 0.2)


# ----------------------------------------------------------------------------
# a1  HaarDownsampling  (models/modules/Inv_arch.py:44-84)
# ----------------------------------------------------------------------------

def haar_fwd(x: torch.Tensor) -> torch.Tensor:
    """(N,C,H,W) -> (N,4C,H/2,W/2); out[:, k*C+c] = band k of channel c.

    Inv_arch.py:49-58 builds the four +-1 2x2 kernels, :69 applies them as a
    stride-2 depthwise conv and divides by 4, :70-72 transposes (C,4)->(4,C).
    The summation order a+b+c+d (row-major over the 2x2 window) reproduces the
    reference's fp32 bits (checked against the golden vectors with torch.equal).
    """
    a = x[:, :, 0::2, 0::2]
    b = x[:, :, 0::2, 1::2]
    c = x[:, :, 1::2, 0::2]
    d = x[:, :, 1::2, 1::2]
    ll = (((a + b) + c) + d) / 4.0
    hl = (((a - b) + c) - d) / 4.0
    lh = (((a + b) - c) - d) / 4.0
    hh = (((a - b) - c) + d) / 4.0
    return torch.cat((ll, hl, lh, hh), dim=1)


def haar_inv(y: torch.Tensor) -> torch.Tensor:
    """(N,4C,h,w) -> (N,C,2h,2w): Inv_arch.py:78-81 (conv_transpose2d, no /4)."""
    n, c4, h, w = y.shape
    c = c4 // 4
    ll, hl, lh, hh = y[:, 0:c], y[:, c:2 * c], y[:, 2 * c:3 * c], y[:, 3 * c:4 * c]
    out = y.new_empty((n, c, 2 * h, 2 * w))
    out[:, :, 0::2, 0::2] = ((ll + hl) + lh) + hh
    out[:, :, 0::2, 1::2] = ((ll - hl) + lh) - hh
    out[:, :, 1::2, 0::2] = ((ll + hl) - lh) - hh
    out[:, :, 1::2, 1::2] = ((ll - hl) - lh) + hh
    return out


def haar_jacobian(shape: Sequence[int], rev: bool) -> float:
    """Inv_arch.py:66-67,75-76: elements/4*log(1/16) (fwd) or *log(16) (rev);
    ``shape`` is the shape of the tensor handed to that call."""
    elements = shape[1] * shape[2] * shape[3]
    return elements / 4 * (math.log(16.0) if rev else math.log(1 / 16.0))


# ----------------------------------------------------------------------------
# a6  PixelUnshuffle / FrequencyAnalyzer (models/modules/SelfC_GMM_arch_inv.py:46-82)
# ----------------------------------------------------------------------------

def pixel_unshuffle_ref(x: torch.Tensor, s: int) -> torch.Tensor:
    """SelfC_GMM_arch_inv.py:51-60: out channel = (sy*S+sx)*C + c."""
    n, c, h, w = x.shape
    x = x.reshape(n, c, h // s, s, w // s, s).permute(0, 3, 5, 1, 2, 4)
    return x.reshape(n, c * s * s, h // s, w // s)


def freq_fwd(x: torch.Tensor, k: int = 4) -> torch.Tensor:
    """(N,3,H,W) -> (N,3+3k^2,H/k,W/k): SelfC_GMM_arch_inv.py:75-78.

    ``nn.Upsample(1/k, mode='area')`` == k x k block mean; ``nn.Upsample(k,
    mode='area')`` == nearest replicate."""
    lo = F.avg_pool2d(x, k)
    up = lo.repeat_interleave(k, dim=2).repeat_interleave(k, dim=3)
    hi = pixel_unshuffle_ref(x - up, k)
    return torch.cat((lo, hi), dim=1)


def freq_inv(y: torch.Tensor, k: int = 4) -> torch.Tensor:
    """(N,3+3k^2,h,w) -> (N,3,kh,kw): SelfC_GMM_arch_inv.py:80-82.

    Uses nn.PixelShuffle channel order c*k^2+sy*k+sx, which is NOT the inverse
    of the forward's order (SURVEY trap 3) - reproduced as is."""
    lo = y[:, 0:3]
    hi = y[:, 3:]
    up = lo.repeat_interleave(k, dim=2).repeat_interleave(k, dim=3)
    return up + F.pixel_shuffle(hi, k)


# ----------------------------------------------------------------------------
# a12 Quantization (models/modules/Quantization.py:4-26)
# ----------------------------------------------------------------------------

def quantize(x: torch.Tensor, quant_v: float = 255.0, is_clip: bool = True) -> torch.Tensor:
    if is_clip:
        x = torch.clamp(x, 0, 1)
    return (x * quant_v).round() / quant_v


# ----------------------------------------------------------------------------
# a3  DenseBlock (2-D)  (models/modules/Subnet_constructor.py:8-34)
# ----------------------------------------------------------------------------

def dense_block(p: Params, x: torch.Tensor, is_res: bool = False) -> torch.Tensor:
    feats = [x]
    for i in range(1, 5):
        inp = feats[0] if i == 1 else torch.cat(feats, dim=1)
        feats.append(lrelu(F.conv2d(inp, p[f"conv{i}.weight"], p.get(f"conv{i}.bias"), 1, 1)))
    out = F.conv2d(torch.cat(feats, dim=1), p["conv5.weight"], p.get("conv5.bias"), 1, 1)
    return out + x if is_res else out


# ----------------------------------------------------------------------------
# a4  D2DTInput  (models/modules/Subnet_constructor.py:98-133)
# ----------------------------------------------------------------------------

def d2dt(p: Params, x: torch.Tensor, t: int) -> torch.Tensor:
    """(B*T,Cin,H,W) -> (B*T,Cout,H,W).

    conv1-4 are Conv3d (1,3,3) pad (0,1,1) == per-frame 3x3 convs
    (Subnet_constructor.py:102-105,126-129); conv5 is Conv3d (3,1,1) pad (1,0,0)
    == 3-tap temporal conv inside each clip with zero padding at the clip ends
    (:106,130).  Frames are laid out clip-major (b*T + t), :119-124."""
    bt, cin, h, w = x.shape
    assert bt % t == 0, "frame count must be a multiple of the temporal length"
    feats = [x]
    for i in range(1, 5):
        wgt = p[f"conv{i}.weight"]           # (gc, cin_i, 1, 3, 3)
        inp = feats[0] if i == 1 else torch.cat(feats, dim=1)
        feats.append(lrelu(F.conv2d(inp, wgt[:, :, 0], p.get(f"conv{i}.bias"), 1, 1)))
    d = torch.cat(feats, dim=1)              # (B*T, cin+128, H, W)
    w5 = p["conv5.weight"]                   # (cout, cin+128, 3, 1, 1)
    cout = w5.shape[0]
    d5 = d.reshape(bt // t, t, d.shape[1], h, w)
    out = torch.zeros((bt // t, t, cout, h, w), dtype=x.dtype)
    for dt in (-1, 0, 1):
        wk = w5[:, :, dt + 1, 0, 0]          # (cout, C)
        lo, hi = max(0, -dt), min(t, t - dt)  # output frames whose tap t+dt is inside the clip
        if hi <= lo:
            continue
        contrib = torch.einsum("oc,btchw->btohw", wk, d5[:, lo + dt:hi + dt])
        out[:, lo:hi] += contrib
    if p.get("conv5.bias") is not None:
        out = out + p["conv5.bias"].view(1, 1, -1, 1, 1)
    return out.reshape(bt, cout, h, w)


def subnet_apply(kind: str, p: Params, x: torch.Tensor, t: int) -> torch.Tensor:
    """Dispatch on the ``subnet()`` factory name (Subnet_constructor.py:719-788)."""
    if kind == "DBNet":
        return dense_block(p, x)
    if kind == "D2DTNet":
        return d2dt(p, x, t)
    raise ValueError(f"oracle covers DBNet and D2DTNet only, got {kind!r}")


# ----------------------------------------------------------------------------
# a2  InvBlockExp  (models/modules/Inv_arch.py:8-41)
# ----------------------------------------------------------------------------

def invblock(kind: str, p: Params, x: torch.Tensor, split1: int, t: int,
             rev: bool = False, clamp: float = 1.0) -> Tuple[torch.Tensor, torch.Tensor]:
    """Returns (cat(y1,y2), s).  Inv_arch.py:21-33."""
    x1, x2 = x[:, :split1], x[:, split1:]
    pf, pg, ph = _sub(p, "F"), _sub(p, "G"), _sub(p, "H")
    if not rev:
        y1 = x1 + subnet_apply(kind, pf, x2, t)
        s = clamp * (torch.sigmoid(subnet_apply(kind, ph, y1, t)) * 2 - 1)
        y2 = x2 * torch.exp(s) + subnet_apply(kind, pg, y1, t)
    else:
        s = clamp * (torch.sigmoid(subnet_apply(kind, ph, x1, t)) * 2 - 1)
        y2 = (x2 - subnet_apply(kind, pg, x1, t)) / torch.exp(s)
        y1 = x1 - subnet_apply(kind, pf, y2, t)
    return torch.cat((y1, y2), dim=1), s


def invblock_jacobian(s: torch.Tensor, n: int, rev: bool) -> torch.Tensor:
    """Inv_arch.py:35-41."""
    return (-torch.sum(s) if rev else torch.sum(s)) / n


# ----------------------------------------------------------------------------
# a10 SelfC-large InvBlock stack (models/modules/SelfC_GMM_arch_inv.py:432-490)
# ----------------------------------------------------------------------------

def large_block_indices(params: Params) -> List[int]:
    idx = sorted({int(k.split(".")[1]) for k in params if k.startswith("operations.")})
    return idx


def large_fwd(params: Params, x: torch.Tensor, t: int = 7, kind: str = "D2DTNet",
              split1: int = 3, k: int = 4) -> torch.Tensor:
    """FrequencyAnalyzer then every InvBlockExp in order: SelfC_GMM_arch_inv.py:454-469.
    Returns the (N,51,h,w) latent (``loss_c`` is identically 0 there, :466)."""
    out = freq_fwd(x, k)
    for i in large_block_indices(params):
        out, _ = invblock(kind, _sub(params, f"operations.{i}"), out, split1, t, rev=False)
    return out


def large_inv_from_latent(params: Params, z: torch.Tensor, t: int = 7, kind: str = "D2DTNet",
                          split1: int = 3, k: int = 4) -> torch.Tensor:
    """Reversed InvBlockExp stack then FrequencyAnalyzer reverse on a full
    51-channel latent (the op loop of SelfC_GMM_arch_inv.py:486-489)."""
    out = z
    for i in reversed(large_block_indices(params)):
        out, _ = invblock(kind, _sub(params, f"operations.{i}"), out, split1, t, rev=True)
    return freq_inv(out, k)


def large_roundtrip(params: Params, x: torch.Tensor, t: int = 7) -> Tuple[torch.Tensor, torch.Tensor]:
    """The BASELINE metric's unit of work: FA.fwd -> 8x fwd -> Quantization of
    the LR channels -> 8x inv -> FA.inv, with the forward's own HF channels fed
    back.  Returns (latent, reconstruction)."""
    z = large_fwd(params, x, t)
    zq = torch.cat((quantize(z[:, :3]), z[:, 3:]), dim=1)
    return z, large_inv_from_latent(params, zq, t)


# ----------------------------------------------------------------------------
# a11 Haar nets (Inv_arch.py:87-127, SelfC_arch_inv.py:276-338)
# ----------------------------------------------------------------------------

def haar_net_fwd(params: Params, x: torch.Tensor, block_num: Sequence[int], t: int = 7,
                 kind: str = "DBNet", split1: int = 3) -> torch.Tensor:
    """[Haar, block_num[i] x InvBlockExp] per level: Inv_arch.py:93-102,108-111.
    Returns the full latent; IRN's forward then slices it (:112-114)."""
    out = x
    op = 0
    for nb in block_num:
        out = haar_fwd(out)
        op += 1
        for _ in range(nb):
            out, _ = invblock(kind, _sub(params, f"operations.{op}"), out, split1, t, rev=False)
            op += 1
    return out


def haar_net_inv(params: Params, z: torch.Tensor, block_num: Sequence[int], t: int = 7,
                 kind: str = "DBNet", split1: int = 3) -> torch.Tensor:
    ops: List[Tuple[str, int]] = []
    op = 0
    for nb in block_num:
        ops.append(("haar", op))
        op += 1
        for _ in range(nb):
            ops.append(("blk", op))
            op += 1
    out = z
    for typ, i in reversed(ops):
        if typ == "haar":
            out = haar_inv(out)
        else:
            out, _ = invblock(kind, _sub(params, f"operations.{i}"), out, split1, t, rev=True)
    return out


# ----------------------------------------------------------------------------
# a7  GlobalAgg  (models/modules/SelfC_GMM_arch_inv.py:257-285)
# ----------------------------------------------------------------------------

def global_agg(p: Params, x: torch.Tensor, t: int) -> torch.Tensor:
    bt, c, h, w = x.shape
    b = bt // t
    proj1 = F.conv2d(x, p["proj1.weight"], p["proj1.bias"])                       # :266
    pooled = F.adaptive_avg_pool2d(x, (32, 32)).reshape(bt, c, 32 * 32)           # :269-270
    g = (pooled @ p["fc.weight"].t() + p["fc.bias"]).squeeze(-1).reshape(b, t, c)  # :271-272
    q = g @ p["proj2.weight"].t() + p["proj2.bias"]                               # :273
    k = g @ p["proj3.weight"].t() + p["proj3.bias"]                               # :274
    a = torch.softmax((q @ k.transpose(1, 2)) / c, dim=-1)                        # :275-277
    v = proj1.reshape(b, t, c, h, w).permute(0, 2, 3, 4, 1).reshape(b, c * h * w, t)  # :279-280
    mixed = (v @ a).reshape(b, c, h, w, t).permute(0, 4, 1, 2, 3).reshape(bt, c, h, w)  # :281-284
    return x + mixed


# ----------------------------------------------------------------------------
# a8  STPNet v2  (models/modules/SelfC_GMM_arch_inv.py:289-430)
# ----------------------------------------------------------------------------

def stp_v2_parameters(params: Params, lr: torch.Tensor, t: int, stp_blk_num: int = 6, thin: bool = False) -> torch.Tensor:
    """lr (B*T,3,h,w) -> raw head output (B*T, Cp, h, w) (``self.parameters`` of
    the reference, frame-major here instead of (b,Cp,t,h,w)); :358-376."""
    x = d2dt(_sub(params, "local_m1"), lr, t)
    x = global_agg(_sub(params, "global_m1"), x, t)
    x = d2dt(_sub(params, "local_m2"), x, t)
    x = global_agg(_sub(params, "global_m2"), x, t)
    for i in range(stp_blk_num - 2):
        x = d2dt(_sub(params, f"other_stp_modules.{2 * i}"), x, t)
        x = global_agg(_sub(params, f"other_stp_modules.{2 * i + 1}"), x, t)
    tail = sorted({int(k.split(".")[1]) for k in params if k.startswith("tail_gmm.")})
    for n_, j in enumerate(tail):                    # tail_gmm = [lrelu, conv1x1x1]* (:327-344); 'gmm_thin': ReLU after the first (:345-354)
        x = torch.relu(x) if (thin and n_ > 0) else lrelu(x)
        wgt = params[f"tail_gmm.{j}.weight"]
        x = F.conv2d(x, wgt.reshape(wgt.shape[0], wgt.shape[1], 1, 1), params[f"tail_gmm.{j}.bias"])
    return x


def stp_v2_gmm_sample(raw: torch.Tensor, eps: torch.Tensor, hf_dim: int = 48, k: int = 5) -> torch.Tensor:
    """raw (N, hf_dim*K*3, h, w), eps (N, hf_dim, K, h, w) -> (N, hf_dim, h, w).

    :382-394: reshape (hf_dim,K,3); pi = softmax over the hf_dim axis (trap 6),
    log-scale = clamp(idx1,-7,7), mean = idx2; v = sum_k pi*(eps*exp(ls)+mean).
    The reference draws eps on the device (:412-417); here it is injected."""
    n, _, h, w = raw.shape
    r = raw.reshape(n, hf_dim, k, 3, h, w)
    pi = torch.softmax(r[:, :, :, 0], dim=1)
    ls = torch.clamp(r[:, :, :, 1], -7, 7)
    mu = r[:, :, :, 2]
    return (pi * (eps * torch.exp(ls) + mu)).sum(2)


# ----------------------------------------------------------------------------
# a9  STPNet v1, condition_func == "D2DTNet", l2 head  (models/modules/SelfC_arch_inv.py:90-198)
# ----------------------------------------------------------------------------

def feature_calapse_block(p: Params, x: torch.Tensor, t: int, scale: int = 4, is_res: bool = False) -> torch.Tensor:
    """FeatureCalapseBlock.forward (Subnet_constructor.py:306-324): SpaceToDepth (:242-257, channel
    (sy*S+sx)*C + c), 3-D dense block with (3,3,3) conv1 / conv5 and (1,3,3) conv2-4, PixelShuffle."""
    bt, c, hh, ww = x.shape
    xs = pixel_unshuffle_ref(x, scale) if scale > 1 else x
    h, w = xs.shape[2], xs.shape[3]
    v = xs.reshape(bt // t, t, xs.shape[1], h, w).transpose(1, 2)          # (b, C, t, h, w)
    feats = [v]
    for i in range(1, 5):
        inp = feats[0] if i == 1 else torch.cat(feats, dim=1)
        pad = (1, 1, 1) if i == 1 else (0, 1, 1)
        feats.append(lrelu(F.conv3d(inp, p[f"conv{i}.weight"], p.get(f"conv{i}.bias"), 1, pad)))
    y = F.conv3d(torch.cat(feats, dim=1), p["conv5.weight"], p.get("conv5.bias"), 1, (1, 1, 1))
    y = y.transpose(1, 2).reshape(bt, -1, h, w)
    y = F.pixel_shuffle(y, scale) if scale > 1 else y
    return y + x if is_res else y


def stp_v1_parameters(params: Params, lr: torch.Tensor, t: int) -> torch.Tensor:
    """lr (B*T,3,h,w) -> (B*T,9,h,w).  condition_func "D2DTNet": blk1 = 3 x D2DTInput (3->12->24->48),
    blk2 = D2DTInput(48->c) (:99-105); default: blk1 = FeatureCalapseBlock(3,12), blk2 =
    FeatureCalapseBlock(12,c) (:107-108); tail = LeakyReLU + Conv3d(c,9,1) (:110-116,148)."""
    x = lr
    if "blk1.0.conv1.weight" in params:
        for i in range(3):
            x = d2dt(_sub(params, f"blk1.{i}"), x, t)
        x = d2dt(_sub(params, "blk2"), x, t)
    else:
        x = feature_calapse_block(_sub(params, "blk1"), x, t)
        x = feature_calapse_block(_sub(params, "blk2"), x, t)
    if "tail.1.weight" in params:
        wgt = params["tail.1.weight"]
        return F.conv2d(lrelu(x), wgt.reshape(wgt.shape[0], wgt.shape[1], 1, 1), params["tail.1.bias"])
    for j in (1, 3, 5):                              # fh_loss "gmm": tail_gmm = [lrelu, conv1x1x1] x 3 (:118-128) -> 9*K*3 channels
        wgt = params[f"tail_gmm.{j}.weight"]
        x = F.conv2d(lrelu(x), wgt.reshape(wgt.shape[0], wgt.shape[1], 1, 1), params[f"tail_gmm.{j}.bias"])
    return x


def stp_v1_gmm_sample(raw: torch.Tensor, eps: torch.Tensor, hf_dim: int = 9, k: int = 5) -> torch.Tensor:
    """raw (N, 9*K*3, h, w), eps (K, 9, N, h, w) [one draw per mixture component] -> (N, 9, h, w).  :151-163: pi = softmax
    over the hf_dim axis, log-scale = clamp(idx1,-7,7), mean = idx2, v = sum_i pi_i * reparametrize(mean_i, ls_i) with
    std = exp(0.5 * ls) in THIS file's reparametrize (:179-186; v2 uses exp(ls))."""
    n, _, h, w = raw.shape
    r = raw.reshape(n, hf_dim, k, 3, h, w)
    pi = torch.softmax(r[:, :, :, 0], dim=1)
    ls = torch.clamp(r[:, :, :, 1], -7, 7)
    mu = r[:, :, :, 2]
    e = eps.permute(2, 1, 0, 3, 4)                   # (N, 9, K, h, w)
    return (pi * (e * torch.exp(0.5 * ls) + mu)).sum(2)


def irn_rev(params: Params, lr: torch.Tensor, hf: torch.Tensor, block_num: Sequence[int], t: int = 7) -> torch.Tensor:
    """InvRescaleNet.forward(rev=True) (Inv_arch.py:115-123): cat(lr, 45 sampled channels), reversed op loop.  The
    InvBlockExp's narrow() takes the channels it needs and silently drops the rest (:22)."""
    need = 3 * 4 ** len(block_num)
    return haar_net_inv(params, torch.cat((lr, hf), 1)[:, :need], block_num, t)


def selfc_haar_fwd(params: Params, x: torch.Tensor, block_num: Sequence[int], t: int = 7,
                   kind: str = "DBNet") -> Tuple[torch.Tensor, torch.Tensor]:
    """SelfC_arch_inv.SelfCInvNet.forward(rev=False) (:300-314): op loop, then STP on the LR channels and
    neg_llh = mean((hf - stp)^2) over the HF channels (l2 head, :189-190)."""
    z = haar_net_fwd(params, x, block_num, t, kind)
    pred = stp_v1_parameters(_sub(params, "stp_net"), z[:, :3], t)
    return z, torch.mean((z[:, 3:] - pred) ** 2)


def selfc_haar_rev(params: Params, lr: torch.Tensor, block_num: Sequence[int], t: int = 7,
                   kind: str = "DBNet") -> Tuple[torch.Tensor, torch.Tensor]:
    """forward(rev=True) (:315-333): hf = STP(lr); reversed op loop on cat(lr, hf)."""
    hf = stp_v1_parameters(_sub(params, "stp_net"), lr, t)
    return haar_net_inv(params, torch.cat((lr, hf), 1), block_num, t, kind), hf


# ----------------------------------------------------------------------------
# quality metric of test_rescaling.py: Y channel (data/util.py:239-245) and PSNR (utils/util.py:198-221)
# ----------------------------------------------------------------------------

def rgb_to_y(x: torch.Tensor) -> torch.Tensor:
    """(N,3,H,W) RGB in [0,1] -> (N,1,H,W): (65.481 R + 128.553 G + 24.966 B + 16) / 255."""
    return ((x[:, 0:1] * 65.481 + x[:, 1:2] * 128.553 + x[:, 2:3] * 24.966 + 16.0) / 255.0)


def psnr_per_frame(a: torch.Tensor, b: torch.Tensor) -> List[float]:
    """20 log10(1 / sqrt(mean((a-b)^2))) per leading index, as calculate_psnr does."""
    out = []
    for i in range(a.shape[0]):
        mse = torch.mean((a[i].double() - b[i].double()) ** 2)
        out.append(float("inf") if mse == 0 else (20.0 * torch.log10(1.0 / torch.sqrt(mse))).item())
    return out


# ----------------------------------------------------------------------------
# Guassian_downsample (models/Guassian.py:7-52), scale 4: the "sr_bd" LR target of SelfCModel.feed_data
# ----------------------------------------------------------------------------

def gaussian_kernel_13(sigma: float = 1.6) -> torch.Tensor:
    """scipy.ndimage.gaussian_filter of a 13x13 dirac (Guassian.py:16-22): truncate 4 sigma -> radius 6, so the
    mask is the outer product of the normalised 1-D weights exp(-k^2 / (2 sigma^2)), k = -6..6 (float64)."""
    k = torch.arange(-6, 7, dtype=torch.float64)
    w = torch.exp(-0.5 * (k / sigma) ** 2)
    w = w / w.sum()
    return torch.outer(w, w)


def gaussian_downsample(x: torch.Tensor) -> torch.Tensor:
    """x (P,H,W) or (..,H,W) planes -> (..,H/4,W/4): reflect-pad 14, 13x13 Gaussian (sigma 1.6) at stride 4,
    crop 2 (Guassian.py:34-51); equivalently out[oy,ox] = sum_ij g[i,j] x[refl(4oy+i-6), refl(4ox+j-6)]."""
    shp = x.shape
    h, w = shp[-2], shp[-1]
    v = x.reshape(-1, 1, h, w)
    v = F.pad(v, [14, 14, 14, 14], "reflect")
    k = gaussian_kernel_13().to(x.dtype).reshape(1, 1, 13, 13)
    y = F.conv2d(v, k, stride=4)[:, :, 2:-2, 2:-2]
    return y.reshape(*shp[:-2], y.shape[-2], y.shape[-1])


# ----------------------------------------------------------------------------
# f1  one training step of SelfCModel.optimize_parameters (models/SelfC_model.py:153-176), large net, l2 STP head
# ----------------------------------------------------------------------------

def reconstruction_loss(x: torch.Tensor, target: torch.Tensor, losstype: str = "l2", eps: float = 1e-6) -> torch.Tensor:
    """ReconstructionLoss.forward (models/modules/loss.py:12-21): mean over all four axes, one after the other."""
    if losstype == "l2":
        v = (x - target) ** 2
    elif losstype == "l1":
        d = x - target
        v = torch.sqrt(d * d + eps)
    else:
        raise ValueError(losstype)
    return v.mean(-1).mean(-1).mean(-1).mean(-1)


class _QuantSTE(torch.autograd.Function):
    """Quant (Quantization.py:4-17): clamp + 8-bit rounding forward, identity backward."""

    @staticmethod
    def forward(ctx, v):
        return quantize(v)

    @staticmethod
    def backward(ctx, g):
        return g


def large_rev_l2(params: Params, lr: torch.Tensor, t: int = 7) -> Tuple[torch.Tensor, torch.Tensor]:
    """SelfCInvNet.forward(x=LR, rev=True) with the l2 head (SelfC_GMM_arch_inv.py:470-490): STP predicts the HF
    channels from the LR frames, then the op loop runs reversed."""
    hf = stp_v2_parameters(_sub(params, "stp_net"), lr, t)
    return large_inv_from_latent(params, torch.cat((lr, hf), dim=1), t), hf


def train_step_losses(params: Params, real_h: torch.Tensor, ref_l: torch.Tensor, t: int = 7,
                      lambda_fit_forw: float = 1.0, lambda_rec_back: float = 1.0):
    """(l_forw_fit, l_back_rec, loss) of optimize_parameters for criterion forw l2 / back l1 (train yml :106-117);
    loss_c of the large net is identically 0 (SelfC_GMM_arch_inv.py:466)."""
    z = large_fwd(params, real_h, t)
    lr_bq = z[:, :3]
    l_fit = lambda_fit_forw * reconstruction_loss(lr_bq, ref_l.detach(), "l2")
    x_s, _ = large_rev_l2(params, _QuantSTE.apply(lr_bq), t)
    l_rec = lambda_rec_back * reconstruction_loss(real_h, x_s[:, :3], "l1")
    return l_fit, l_rec, (l_fit + l_rec) * 144 * 144 * 3


def multistep_lr_restart(base_lr: float, steps: int, milestones: Sequence[int], restarts: Sequence[int] = (0,),
                         weights: Sequence[float] = (1,), gamma: float = 0.1) -> List[float]:
    """lr after each of `steps` scheduler.step() calls of MultiStepLR_Restart (models/lr_scheduler.py:8-31)."""
    lr, out = base_lr, []
    for epoch in range(1, steps + 1):
        if epoch in restarts:
            lr = base_lr * weights[list(restarts).index(epoch)]
        elif epoch in milestones:
            lr = lr * gamma ** list(milestones).count(epoch)
        out.append(lr)
    return out


# ----------------------------------------------------------------------------
# f2  SSIM on the Y channel (utils/util.py:361-470 `ssim`, called per frame by calculate_ssim :597-605)
# ----------------------------------------------------------------------------

def ssim_gauss_1d(size: int = 11, sigma: float = 1.5) -> torch.Tensor:
    """_fspecial_gauss_1d (utils/util.py:362-376)."""
    c = torch.arange(size, dtype=torch.float32) - size // 2
    gk = torch.exp(-(c ** 2) / (2 * sigma ** 2))
    return gk / gk.sum()


def ssim_per_frame(x: torch.Tensor, y: torch.Tensor, data_range: float = 1.0) -> List[float]:
    """x, y (N,1,H,W): SSIM of every frame with an 11-tap sigma-1.5 Gaussian window applied separably without padding
    (gaussian_filter :379-393), K1 = 0.01, K2 = 0.03, mean of the (H-10)x(W-10) map (_ssim :396-441)."""
    win = ssim_gauss_1d().reshape(1, 1, 1, 11)
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2

    def blur(v):
        return F.conv2d(F.conv2d(v, win), win.transpose(2, 3))

    mu1, mu2 = blur(x), blur(y)
    s1 = blur(x * x) - mu1 * mu1
    s2 = blur(y * y) - mu2 * mu2
    s12 = blur(x * y) - mu1 * mu2
    cs = (2 * s12 + c2) / (s1 + s2 + c2)
    m = ((2 * mu1 * mu2 + c1) / (mu1 * mu1 + mu2 * mu2 + c1)) * cs
    return [float(v) for v in m.mean(-1).mean(-1).mean(-1)]


# ----------------------------------------------------------------------------
# f4  codec variant (models/modules/SelfC_Codec_arch_inv.py): FrequencyAnalyzer(k=2) + 4 x InvBlockExp(15 | 3) + the
#     narrow STP (hidden 24, growth 12, clips of TEMP_LEN = 3) + the segmenting / tiling of forward_test.  The H.265
#     stream between the two halves is external and not restated.
# ----------------------------------------------------------------------------

def seg_add_pad(video: torch.Tensor, seg_len: int):
    """(b,t,c,h,w) -> ((b,seg_num,seg_len,c,h,w), pad): utils/util.py:329-345; pads repeat out_video[:, -2:-1]."""
    b, t, c, h, w = video.shape
    pad = 0 if t % seg_len == 0 else seg_len - t % seg_len
    for _ in range(pad):
        video = torch.cat((video, video[:, -2:-1]), dim=1)
    return video.reshape(b, -1, seg_len, c, h, w), pad


def seg_remove_pad(video: torch.Tensor, pad: int, seg_len: int) -> torch.Tensor:
    """utils/util.py:346-354"""
    b, seg_num, seg_len, c, h, w = video.shape
    if pad == 0:
        return video.reshape(b, -1, c, h, w)
    pre = video[:, :seg_num - 1].reshape(b, (seg_num - 1) * seg_len, c, h, w)
    return torch.cat((pre, video[:, -1, :seg_len - pad]), dim=1)


def codec_stp_parameters(params: Params, lr: torch.Tensor, t: int = 3, stp_blk_num: int = 4) -> torch.Tensor:
    """STPNet.forward of the codec file (:292-312): the same chain as stp_v2_parameters on narrower modules (weights
    carry the widths), GlobalAgg over clips of TEMP_LEN = 3 (:77,118), head = `tail` (:257-262).  lr (B*T,3,h,w)."""
    x = d2dt(_sub(params, "local_m1"), lr, t)
    x = global_agg(_sub(params, "global_m1"), x, 3)
    x = d2dt(_sub(params, "local_m2"), x, t)
    x = global_agg(_sub(params, "global_m2"), x, 3)
    for i in range(stp_blk_num - 2):
        x = d2dt(_sub(params, f"other_stp_modules.{2 * i}"), x, t)
        x = global_agg(_sub(params, f"other_stp_modules.{2 * i + 1}"), x, 3)
    for j in sorted({int(k.split(".")[1]) for k in params if k.startswith("tail.")}):
        x = lrelu(x)
        wgt = params[f"tail.{j}.weight"]
        x = F.conv2d(x, wgt.reshape(wgt.shape[0], wgt.shape[1], 1, 1), params[f"tail.{j}.bias"])
    return x


def codec_decode(params: Params, lr: torch.Tensor, t: int = 3) -> torch.Tensor:
    """forward_train(rev=True) (:480-498) with the l2 head: hf = STP(lr); reversed op loop on cat(lr, hf); k = 2."""
    hf = codec_stp_parameters(_sub(params, "stp_net"), lr, t)
    return large_inv_from_latent(params, torch.cat((lr, hf), 1), t, k=2)


def codec_encode_tiled(params: Params, x: torch.Tensor, t_all: int, seg_len: int = 3, wdiv: int = 2) -> torch.Tensor:
    """forward_test(rev=False) up to the H.265 writer (:510-552): 3-frame segments, each column strip through the whole op
    loop on its own, LR channels concatenated along the width.  x (b*t,3,H,W) -> (b*t,3,H/2,W/2), unquantised."""
    bt, c, h, w = x.shape
    b = bt // t_all
    video, pad = seg_add_pad(x.reshape(b, t_all, c, h, w), seg_len)
    outs = []
    for s in range(video.shape[1]):
        seg = video[:, s].reshape(-1, c, h, w)
        strips = [large_fwd(params, seg[:, :, :, i * (w // wdiv):(i + 1) * (w // wdiv)], seg_len, k=2)[:, 0:3] for i in range(wdiv)]
        outs.append(torch.cat(strips, dim=-1))
    lr = torch.stack(outs, 0)
    hh, ww = lr.shape[-2:]
    lr = lr.reshape(video.shape[1], b, seg_len, 3, hh, ww).permute(1, 0, 2, 3, 4, 5)
    return seg_remove_pad(lr, pad, seg_len).reshape(-1, 3, hh, ww)


def codec_decode_tiled(params: Params, lr: torch.Tensor, t_all: int, seg_len: int = 3, hdiv: int = 2, wdiv: int = 2) -> torch.Tensor:
    """forward_test(rev=True) (:575-640): every 3-frame segment is cut into hdiv x wdiv tiles (no halo), each tile runs
    STP + the reversed op loop on its own, the tiles are put back side by side.  lr (b*t,3,h,w) -> (b*t,3,2h,2w)."""
    bt, c, h, w = lr.shape
    b = bt // t_all
    video, pad = seg_add_pad(lr.reshape(b, t_all, c, h, w), seg_len)
    hd, wd = h // hdiv, w // wdiv
    outs = []
    for s in range(video.shape[1]):
        seg = video[:, s].reshape(-1, c, h, w)[:, 0:3]
        rows = []
        for i in range(hdiv):
            rows.append(torch.cat([codec_decode(params, seg[:, :, i * hd:(i + 1) * hd, j * wd:(j + 1) * wd], seg_len)
                                   for j in range(wdiv)], dim=-1))
        outs.append(torch.cat(rows, dim=-2))
    hr = torch.stack(outs, 0)
    H, W = hr.shape[-2:]
    hr = hr.reshape(video.shape[1], b, seg_len, 3, H, W).permute(1, 0, 2, 3, 4, 5)
    return seg_remove_pad(hr, pad, seg_len).reshape(-1, 3, H, W)
