"""Caller-side restatement of the reference's evaluation loop on the HIP path.

``rescale_test`` follows ``SelfCModel.test()`` (codes/models/SelfC_model.py:185-250): the clip is cut into
GOPs of 7 frames, each GOP goes through ``netG(x)`` -> ``Quantization`` of the three LR channels ->
``netG(x=LR, rev=True)``; a trailing partial GOP is padded by repeating the last frame.  When the length is
a multiple of 7 the reference still runs one extra pass on seven copies of the last frame and discards it
(:203-209,236-243) - reproduced only on request (``reference_tail_pass``), the outputs are identical.
``psnr_y`` is test_rescaling.py's metric (Y channel, per frame) with the reduction on the device.
"""
from __future__ import annotations

import math
from typing import List, Tuple

import torch

from . import _lib, runtime as rt
from .modules.Quantization import Quantization

GOP = 7   # SelfC_model.py:196,199 hard-code t = 7 and gop = 7


def rescale_test(net, real_H: torch.Tensor, reference_tail_pass: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """real_H (b*7, 3, h, w) on the device -> (forw_L (b*7,3,h/4,w/4), fake_H (b*7,3,h,w))."""
    real_H = rt.as_input(real_H)
    bt, c, h, w = real_H.shape
    t = GOP
    if bt % t:
        raise RuntimeError(f"{bt} frames are not a multiple of {t} (SelfCModel.test reshapes to (b,7,c,h,w))")
    b = bt // t
    clip = real_H.reshape(b, t, c, h, w)
    quant = Quantization()
    n_gop = t // GOP
    forw_L: List[torch.Tensor] = []
    fake_H: List[torch.Tensor] = []
    with torch.no_grad():
        for i in range(n_gop + 1):
            if i == n_gop:
                keep = t % GOP
                if keep == 0 and not reference_tail_pass:
                    continue
                idx = [i * GOP + j for j in range(keep)] + [t - 1] * (GOP - keep)
                inp = clip[:, idx]
            else:
                keep = GOP
                inp = clip[:, i * GOP:(i + 1) * GOP]
            _b, _t, _c, _h, _w = inp.shape
            out, _ = net(x=inp.reshape(_b * _t, _c, _h, _w))
            lr = quant(out[:, :3])
            rec, _ = net(x=lr, rev=True)
            lr5 = lr.reshape(b, _t, c, h // 4, w // 4)
            rec5 = rec[:, :3].reshape(b, _t, c, h, w)
            for j in range(keep):
                forw_L.append(lr5[:, j])
                fake_H.append(rec5[:, j])
    fl = torch.stack(forw_L, dim=1)
    fh = torch.stack(fake_H, dim=1)
    return fl.reshape(b * t, c, h // 4, w // 4), fh.reshape(b * t, c, h, w)


def psnr_y(img1: torch.Tensor, img2: torch.Tensor) -> List[float]:
    """Per-frame Y-channel PSNR of two (N,3,H,W) RGB tensors in [0,1] (rgb_to_ycbcr + calculate_psnr of the
    reference); squared-error reduction by selfc_y_sse, deterministic two-stage sum."""
    a, b = rt.as_input(img1), rt.as_input(img2)
    if a.shape != b.shape or a.shape[1] != 3:
        raise RuntimeError(f"psnr_y expects two (N,3,H,W) tensors, got {tuple(a.shape)} and {tuple(b.shape)}")
    n, _, h, w = a.shape
    L = _lib.lib()
    nblk = L.selfc_y_sse_blocks(h * w)
    partial = torch.empty((n, nblk), dtype=torch.float64, device=a.device)
    rt.call("selfc_y_sse", a.data_ptr(), b.data_ptr(), partial.data_ptr(), n, h * w, _lib.stream_ptr())
    mse = (partial.sum(dim=1) / (h * w)).cpu().tolist()
    return [float("inf") if m == 0 else 20.0 * math.log10(1.0 / math.sqrt(m)) for m in mse]


_SSIM_WIN = {}


def ssim_y(img1: torch.Tensor, img2: torch.Tensor) -> List[float]:
    """Per-frame Y-channel SSIM of two (N,3,H,W) RGB tensors in [0,1]: rgb_to_ycbcr + calculate_ssim of the reference
    (data/util.py:239-245, utils/util.py:396-441,597-605; 11-tap sigma-1.5 Gaussian window, no padding, data_range 1)."""
    a, b = rt.as_input(img1), rt.as_input(img2)
    if a.shape != b.shape or a.shape[1] != 3:
        raise RuntimeError(f"ssim_y expects two (N,3,H,W) tensors, got {tuple(a.shape)} and {tuple(b.shape)}")
    n, _, h, w = a.shape
    if h < 11 or w < 11:
        raise RuntimeError("ssim_y needs frames of at least 11x11 pixels")
    key = str(a.device)
    if key not in _SSIM_WIN:
        c = torch.arange(11, dtype=torch.float32) - 5
        gk = torch.exp(-(c ** 2) / (2 * 1.5 ** 2))
        _SSIM_WIN[key] = (gk / gk.sum()).to(a.device)
    nbx, nby = (w - 10 + 15) // 16, (h - 10 + 15) // 16
    partial = torch.empty((n, nby * nbx), dtype=torch.float64, device=a.device)
    rt.call("selfc_y_ssim", a.data_ptr(), b.data_ptr(), _SSIM_WIN[key].data_ptr(), partial.data_ptr(), n, h, w, _lib.stream_ptr())
    return (partial.sum(dim=1) / ((h - 10) * (w - 10))).cpu().tolist()


def rescale_metrics(net, real_H: torch.Tensor, ref_L: torch.Tensor = None) -> dict:
    """The per-batch numbers of test_rescaling.py:82-122 on the device: rescale_test, then Y-channel PSNR / SSIM of the
    reconstruction against real_H and of the LR video against ref_L (default: the sr_bd Gaussian LR target)."""
    forw_L, fake_H = rescale_test(net, real_H)
    if ref_L is None:
        ref_L = gaussian_downsample(real_H)
    avg = lambda v: sum(v) / len(v)  # noqa: E731
    psnr, ssim = psnr_y(fake_H, real_H), ssim_y(fake_H, real_H)
    lr_psnr, lr_ssim = psnr_y(forw_L, ref_L), ssim_y(forw_L, ref_L)
    return {"psnr_y": avg(psnr), "ssim_y": avg(ssim), "lr_psnr_y": avg(lr_psnr), "lr_ssim_y": avg(lr_ssim),
            "per_frame": {"psnr_y": psnr, "ssim_y": ssim, "lr_psnr_y": lr_psnr, "lr_ssim_y": lr_ssim}}


def gop_slices(n_frames: int, gop: int = GOP) -> List[List[int]]:
    """Frame indices of consecutive GOPs; the last GOP is padded by repeating the final frame (the padding
    rule of SelfCModel.test, SelfC_model.py:203-209).  100 frames -> 15 GOPs, the last one [98, 99, 99, 99, 99, 99, 99]."""
    out = []
    for s in range(0, n_frames, gop):
        idx = list(range(s, min(s + gop, n_frames)))
        out.append(idx + [n_frames - 1] * (gop - len(idx)))
    return out


def shard_gops(n_gops: int, rank: int, world: int) -> List[int]:
    """GOPs are independent units: round-robin over ranks, no exchange step."""
    return list(range(rank, n_gops, world))


def rescale_video(net, frames: torch.Tensor, rank: int = 0, world: int = 1, gops_per_call: int = 4):
    """frames (F,3,H,W) on the device, any F (e.g. a 100-frame UVG group at 1080p) -> dict with, for this rank's
    GOPs, the frame indices, the quantised LR frames and the reconstructions (padding frames dropped)."""
    frames = rt.as_input(frames)
    n = frames.shape[0]
    slices = gop_slices(n)
    mine = shard_gops(len(slices), rank, world)
    quant = Quantization()
    idx_out, lr_out, rec_out = [], [], []
    with torch.no_grad():
        for s in range(0, len(mine), gops_per_call):
            batch = [slices[g] for g in mine[s:s + gops_per_call]]
            flat = [i for g in batch for i in g]
            x = frames[flat]
            out, _ = net(x=x)
            lr = quant(out[:, :3])
            rec, _ = net(x=lr, rev=True)
            pos = 0
            for g in batch:
                seen = set()
                for j, fi in enumerate(g):
                    if fi not in seen:                 # drop the repeated padding frames
                        seen.add(fi)
                        idx_out.append(fi)
                        lr_out.append(lr[pos + j])
                        rec_out.append(rec[pos + j, :3])
                pos += len(g)
    return {"frames": idx_out, "lr": torch.stack(lr_out) if lr_out else None, "rec": torch.stack(rec_out) if rec_out else None}


_GK = {}


def gaussian_downsample(x: torch.Tensor) -> torch.Tensor:
    """(N,C,H,W) -> (N,C,H/4,W/4): the reference's Guassian_downsample(scale=4) (models/Guassian.py:7-52), i.e. the
    "sr_bd" LR target ref_L that SelfCModel.feed_data builds (SelfC_model.py:128).  13x13 mask = outer product of
    the normalised 1-D weights exp(-k^2/(2*1.6^2)), k = -6..6 (scipy's gaussian_filter of a dirac, 4-sigma truncation)."""
    x = rt.as_input(x)
    n, c, h, w = x.shape
    key = str(x.device)
    if key not in _GK:
        k = torch.arange(-6, 7, dtype=torch.float64)
        w1 = torch.exp(-0.5 * (k / 1.6) ** 2)
        w1 = w1 / w1.sum()
        _GK[key] = torch.outer(w1, w1).float().reshape(169).contiguous().to(x.device)
    y = torch.empty((n, c, h // 4, w // 4), dtype=torch.float32, device=x.device)
    rt.call("selfc_gauss_down4", x.data_ptr(), y.data_ptr(), _GK[key].data_ptr(), n * c, h, w, _lib.stream_ptr())
    return y


# ----------------------------------------------------------------------------------------------------------------------
# config 4 of BASELINE.json: test_rescaling.py on folders of 7-frame groups (Vid4), pretrained weights
# ----------------------------------------------------------------------------------------------------------------------

def load_pretrained(net, path: str, strict: bool = True):
    """BaseModel.load_network (base_model.py:87-107): torch.load on CPU, strip DistributedDataParallel's "module." prefix,
    drop the codec-surrogate keys, load_state_dict(strict)."""
    sd = torch.load(path, map_location="cpu")
    clean = {}
    for k, v in sd.items():
        if "Quantization_H265_Suggrogate" in k:
            continue
        clean[k[7:] if k.startswith("module.") else k] = v
    net.load_state_dict(clean, strict=strict)
    return net


def evaluate_folder(net, dataroot_GT: str, dataroot_list: str, device, max_groups: int = None, eps_seed: int = None) -> dict:
    """cal_metric of test_rescaling.py:65-153 for one dataset entry of the yml (`dataroot_GT` + `dataroot_list`, one 7-frame
    group per list line, batch size 1): frames come through selfc_amd.data (the reference's folder convention), every group
    goes through rescale_metrics on the device.  eps_seed: fh_loss gmm samples its HF channels; a seed makes the draw a
    fixed, device-independent tensor (``STPNet.eps``) so that another implementation can be fed the same noise.
    Returns the averages of the four metrics over the groups and the per-group values (+ the noise used)."""
    from .data import SeptupletDataset
    ds = SeptupletDataset({"dataroot_GT": dataroot_GT, "dataroot_list": dataroot_list, "phase": "val", "video_len": GOP})
    n = len(ds) if max_groups is None else min(len(ds), max_groups)
    groups, stp = [], getattr(net, "stp_net", None)
    for i in range(n):
        gt = ds[i]["GT"]                                       # (C,T,H,W) RGB [0,1]
        real_H = gt.transpose(0, 1).contiguous().to(device)    # feed_data: (T,C,H,W) frames (SelfC_model.py:117-124)
        if real_H.shape[0] != GOP:
            raise RuntimeError(f"{ds[i]['GT_path']}: {real_H.shape[0]} frames, expected groups of {GOP}")
        eps = None
        if eps_seed is not None and stp is not None and getattr(stp, "fh_loss", "l2") != "l2":
            h, w = real_H.shape[2] // 4, real_H.shape[3] // 4
            eps = torch.randn((1, stp.hf_dim, stp.K, GOP, h, w), generator=torch.Generator().manual_seed(eps_seed + i))
            stp.eps = eps
        try:
            m = rescale_metrics(net, real_H)
        finally:
            if eps is not None:
                stp.eps = None
        m["path"], m["eps"] = ds[i]["GT_path"], eps
        groups.append(m)
    keys = ("psnr_y", "ssim_y", "lr_psnr_y", "lr_ssim_y")
    out = {k: sum(g[k] for g in groups) / max(1, len(groups)) for k in keys}
    out["groups"] = groups
    return out
