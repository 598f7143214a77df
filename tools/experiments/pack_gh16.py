"""Packing for tools/experiments/fused_gh_16x16x32.hip (not used by the product)."""
from typing import List, Sequence

import torch


def _operand(t):          # the product hook replaces this with selfc_amd.packing._operand (which PackPlan switches to raw indices)
    return t.to(torch.float16).contiguous()

def frag16_rows(h: int) -> List[int]:
    """Output channels behind the 16 rows of output-half h's 16x16x32 A fragments in the fused kernels: row m <-> channel
    8 (m // 4) + 4 h + m % 4, so that a lane's accumulator rows 4 oct + i of half 0 and of half 1 are the 8 consecutive
    channels 8 oct .. 8 oct + 7 - one 16-byte piece of the pixel, exactly what the next conv reads as its B operand."""
    return [8 * (m // 4) + 4 * h + m % 4 for m in range(16)]


def _frags16(wk: torch.Tensor) -> torch.Tensor:
    """wk (32 out, nsteps, 32 k) -> 16x16x32 A fragments [2 nsteps, 64, 8]: step-major, output half minor;
    lane l holds row l % 16, k = 8 (l // 16) + j."""
    nst = wk.shape[1]
    halves = [wk[frag16_rows(h)] for h in (0, 1)]                                   # (16, nst, 32) each
    t = torch.stack(halves, dim=0).reshape(2, 16, nst, 4, 8).permute(2, 0, 3, 1, 4)  # (nst, h, oct, row, j)
    return t.reshape(2 * nst, 64, 8)


def pack_fused_gh(weights: Sequence[torch.Tensor], cin: int = 3) -> torch.Tensor:
    """conv1..conv4 weights of a cin == 3 dense block -> the fragment stream of csrc/fused_gh.hip (v_mfma_f32_16x16x32,
    k = 32 per step, two output halves per step): per conv [im2col: 16 tap slots x 4 (c0 c1 c2 0) = 2 steps]
    [feature j = 1..: 9 taps = 9 steps each]  -> f16 [4 + 22 + 40 + 58 = 124, 64, 8]."""
    assert cin == 3 and len(weights) == 4
    frags = []
    for layer, wt in enumerate(weights, start=1):
        w = wt.detach().float()
        if w.dim() == 5:
            w = w[:, :, 0]
        assert w.shape == (32, cin + 32 * (layer - 1), 3, 3), tuple(w.shape)
        w9 = w.reshape(32, w.shape[1], 9)
        im = torch.zeros(32, 16, 4, dtype=torch.float32, device=w.device)
        im[:, :9, :3] = w9[:, :3, :].permute(0, 2, 1)                    # k = (tap - 8 step) * 4 + c
        steps = [im.reshape(32, 2, 32)]
        for i in range(layer - 1):
            steps.append(w9[:, cin + 32 * i: cin + 32 * (i + 1), :].permute(0, 2, 1))       # (32, 9 taps, 32 channels)
        frags.append(_frags16(torch.cat(steps, dim=1)))
    out = torch.cat(frags, dim=0)
    assert out.shape[0] == 124
    return _operand(out)


