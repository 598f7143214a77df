"""Config 4 of BASELINE.json: test_rescaling.py's metrics (Y-channel PSNR / SSIM of the reconstruction and of the LR video)
of SelfC-large on folders of 7-frame groups, HIP path, optionally next to the CPU oracle on the first groups.

    python tools/eval_vid4.py --root <Vid4 root> --pretrain <selfc_large_pretrain.pth> [--oracle-groups 1]

<root>/<seq>/ds_7_to_7_new/{testlist.txt, <group>/im1.png..im7.png} for seq in city, walk, calendar, foliage (the layout of
options/test/rescaling/test_SelfC_large_vid4.yml:10-40).  Neither the frames nor the checkpoint ship with the reference
(.MISSING_LARGE_BLOBS, README.md:36-58); tests/test_gpu_data.py runs this same code on a synthetic folder, and on the real
thing when SELFC_VID4_ROOT and SELFC_PRETRAIN point at it.  The oracle import is for the comparison only."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch  # noqa: E402

OPT = {"global_module": "nonlocal", "stp_blk_num": 6, "fh_loss": "gmm", "scale": 4, "gmm_k": 5}   # test_SelfC_large_vid4.yml:44-54


def oracle_psnr(sd, real_H, eps, fh_loss):
    """the same group through the CPU oracle with the same noise: (psnr_y per frame, lr psnr per frame)"""
    from oracle import selfc_oracle as O
    t = real_H.shape[0]
    z = O.large_fwd(sd, real_H, t)
    lr = O.quantize(z[:, :3])
    stp = {k[len("stp_net."):]: v for k, v in sd.items() if k.startswith("stp_net.")}
    raw = O.stp_v2_parameters(stp, lr, t)
    hf = raw if fh_loss == "l2" else O.stp_v2_gmm_sample(raw, eps[0].permute(2, 0, 1, 3, 4))
    rec = O.large_inv_from_latent(sd, torch.cat((lr, hf), 1), t)
    ref_l = O.gaussian_downsample(real_H)
    return O.psnr_per_frame(O.rgb_to_y(rec), O.rgb_to_y(real_H)), O.psnr_per_frame(O.rgb_to_y(lr), O.rgb_to_y(ref_l))


def run(root, pretrain, oracle_groups=0, seqs=("city", "walk", "calendar", "foliage"), sub="ds_7_to_7_new", opt=OPT, max_groups=None,
        device="cuda:0"):
    from selfc_amd import GlobalVar, harness
    from selfc_amd.modules.SelfC_GMM_arch_inv import SelfCInvNet
    GlobalVar.set_Temporal_LEN(7)
    dev = torch.device(device)
    net = SelfCInvNet(opt, 3, 3, "D2DTNet", [4, 4], 2)
    harness.load_pretrained(net, pretrain, strict=True)
    net.to(dev).eval()
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    report = {}
    for seq in seqs:
        d = os.path.join(root, seq, sub) if sub else os.path.join(root, seq)
        res = harness.evaluate_folder(net, d, os.path.join(d, "testlist.txt"), dev, max_groups=max_groups, eps_seed=4000)
        entry = {k: res[k] for k in ("psnr_y", "ssim_y", "lr_psnr_y", "lr_ssim_y")}
        entry["groups"] = len(res["groups"])
        diffs = []
        for g in res["groups"][:oracle_groups]:
            from selfc_amd.data import _read_frame
            import numpy as np
            gd = os.path.dirname(g["path"])
            frames = torch.from_numpy(np.stack([_read_frame(os.path.join(gd, f"im{i}.png")) for i in range(1, 8)])).permute(0, 3, 1, 2).contiguous()
            p_ref, _ = oracle_psnr(sd, frames, g["eps"], opt["fh_loss"])
            diffs.append(max(abs(a - b) for a, b in zip(g["per_frame"]["psnr_y"], p_ref)))
        if diffs:
            entry["max_psnr_diff_vs_oracle_dB"] = max(diffs)
        report[seq] = entry
    return report


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--root", default=os.environ.get("SELFC_VID4_ROOT"))
    ap.add_argument("--pretrain", default=os.environ.get("SELFC_PRETRAIN"))
    ap.add_argument("--oracle-groups", type=int, default=1, help="groups per sequence also run through the CPU oracle (slow)")
    a = ap.parse_args()
    if not a.root or not a.pretrain:
        raise SystemExit("--root / --pretrain (or SELFC_VID4_ROOT / SELFC_PRETRAIN) are required: the assets do not ship with the reference")
    print(json.dumps(run(a.root, a.pretrain, a.oracle_groups), indent=1))
