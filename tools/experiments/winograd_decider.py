"""would Winograd F(2x2, 3x3) with f16 TRANSFORMED operands hold the 1e-3 bar for conv3 / conv4 of G and H?

CPU emulation on the oracle (no kernel): the whole SelfC-large stack (8 blocks, g8_large_stack weights) forward and inverse on one
7x3x64x96 clip, with every 3x3 conv computing on f16-rounded operands and fp32 accumulation - (a) all convs direct (what the HIP
path does), (b) conv3 + conv4 of every G and H through Winograd F(2x2,3x3): input tiles transformed in fp32 from the f16-rounded
activations and ROUNDED TO f16 (they are the MFMA's B operand), filters transformed in fp32 from the fp32 weights and rounded to f16
(the A operand), products accumulated in fp32, output transform in fp32.  Errors against the fp32 oracle, conftest.rel_err's metric.

    python tools/experiments/winograd_decider.py > profiles/r6/winograd_decider.txt
"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden  # noqa: E402
from oracle import selfc_oracle as O  # noqa: E402

MODE = {"v": "fp32"}
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=torch.float32)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def r16(t):
    return t.half().float()


def winograd_conv(x, w, b):
    """3x3 pad-1 conv, F(2x2,3x3); x (N,C,H,W) fp32 holding f16-representable values, w (O,C,3,3) fp32."""
    n, c, h, wd = x.shape
    hp, wp = (h + 1) // 2 * 2, (wd + 1) // 2 * 2
    xp = F.pad(x, (1, 1 + wp - wd, 1, 1 + hp - h))
    tiles = xp.unfold(2, 4, 2).unfold(3, 4, 2)                      # (N,C,th,tw,4,4)
    V = r16(torch.einsum("ij,nctujk,lk->nctuil", BT, tiles, BT))   # B^T d B, rounded: the MFMA operand
    U = r16(torch.einsum("ij,ocjk,lk->ocil", G, w, G))             # G g G^T, rounded
    M = torch.einsum("ocil,nctuil->notuil", U, V)                  # fp32 accumulate over channels
    Y = torch.einsum("ij,notujk,lk->notuil", AT, M, AT)            # (N,O,th,tw,2,2)
    y = Y.permute(0, 1, 2, 4, 3, 5).reshape(n, w.shape[0], hp, wp)[:, :, :h, :wd]
    return y + b.view(1, -1, 1, 1)


_conv2d = F.conv2d
STATE = {"layer": 0, "wino_layers": (), "in_gh": False}


def d2dt_emul(p, x, t):
    """oracle.d2dt with emulated operand rounding (Subnet_constructor.py:115-133)"""
    bt, c, h, w = x.shape
    feats = [x]
    for k in range(1, 5):
        inp = torch.cat(feats, 1)
        wk, bk = p[f"conv{k}.weight"][:, :, 0], p[f"conv{k}.bias"]
        if MODE["v"] == "fp32":
            y = _conv2d(inp, wk, bk, padding=1)
        elif MODE["v"] == "wino" and STATE["in_gh"] and k in (3, 4):
            y = winograd_conv(r16(inp), wk, bk)
        else:
            y = _conv2d(r16(inp), r16(wk), bk, padding=1)
        feats.append(O.lrelu(y))
    inp = torch.cat(feats, 1)
    b = bt // t
    v = inp.reshape(b, t, -1, h, w).transpose(1, 2)
    w5 = p["conv5.weight"]
    if MODE["v"] != "fp32":
        v, w5 = r16(v), r16(w5)
    y = F.conv3d(v, w5, p["conv5.bias"], padding=(1, 0, 0))
    return y.transpose(1, 2).reshape(bt, -1, h, w)


def invblock_emul(p, x, split1, t, rev):
    x1, x2 = x[:, :split1], x[:, split1:]
    sub = lambda n, v: d2dt_emul(O._sub(p, n), v, t)   # noqa: E731
    if not rev:
        STATE["in_gh"] = False
        y1 = x1 + sub("F", x2)
        STATE["in_gh"] = True
        s = 2 * torch.sigmoid(sub("H", y1)) - 1
        y2 = x2 * torch.exp(s) + sub("G", y1)
    else:
        STATE["in_gh"] = True
        s = 2 * torch.sigmoid(sub("H", x1)) - 1
        y2 = (x2 - sub("G", x1)) / torch.exp(s)
        STATE["in_gh"] = False
        y1 = x1 - sub("F", y2)
    return torch.cat((y1, y2), 1)


def stack(g, x, rev):
    idx = O.large_block_indices(g)
    z = x if rev else O.freq_fwd(x)
    for i in (reversed(idx) if rev else idx):
        z = invblock_emul(O._sub(g, f"operations.{i}"), z, 3, 7, rev)
    return O.freq_inv(z) if rev else z


def err(a, b):
    return float((a - b).abs().max() / b.abs().max()), float((a - b).norm() / b.norm())


def main():
    torch.set_num_threads(8)
    g = load_golden("g8_large_stack")
    x = torch.rand(7, 3, 64, 96, generator=torch.Generator().manual_seed(1234))
    # self-check of the Winograd emulation in fp32 (no rounding): equals the direct conv
    xx, ww, bb = torch.randn(2, 5, 9, 11), torch.randn(4, 5, 3, 3), torch.randn(4)
    global r16
    keep = r16
    r16 = lambda t: t   # noqa: E731
    d = float((winograd_conv(xx, ww, bb) - F.conv2d(xx, ww, bb, padding=1)).abs().max())
    r16 = keep
    print(f"winograd emulation self-check (fp32 operands) max abs diff vs conv2d: {d:.2e}")
    res = {}
    for mode in ("fp32", "f16", "wino"):
        MODE["v"] = mode
        with torch.no_grad():
            z = stack(g, x, False)
            res[mode] = (z, None)
    zq = torch.cat((O.quantize(res["fp32"][0][:, :3]), res["fp32"][0][:, 3:]), 1)
    for mode in ("fp32", "f16", "wino"):
        MODE["v"] = mode
        with torch.no_grad():
            res[mode] = (res[mode][0], stack(g, zq, True))
    print("SelfC-large stack, 7x3x64x96, g8_large_stack weights; errors against the fp32 oracle: max|a-b|/max|b| , ||a-b||/||b||")
    for mode, label in (("f16", "direct convs, f16 operands (the shipped arithmetic)"),
                        ("wino", "conv3 + conv4 of G and H as Winograd F(2x2,3x3) with f16 transformed operands, everything else direct f16")):
        ef, ei = err(res[mode][0], res["fp32"][0]), err(res[mode][1], res["fp32"][1])
        print(f"  {label}:\n      forward latent {ef[0]:.2e} , {ef[1]:.2e}      inverse {ei[0]:.2e} , {ei[1]:.2e}      (bar 1e-3)")
    ef, ei = err(res["wino"][0], res["fp32"][0]), err(res["wino"][1], res["fp32"][1])
    go = max(ef[0], ei[0]) < 1e-3
    print("DECISION:", "GO (inside the bar)" if go else "NO-GO: outside the 1e-3 bar before a single kernel is written",
          "- F(2x2,3x3) would cut conv3 + conv4's MFMA work by 2.25x (81 % of G/H's MACs)")


if __name__ == "__main__":
    main()
