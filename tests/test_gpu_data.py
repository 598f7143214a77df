"""Frame I/O into the HIP path: folders of 7-frame groups in the reference's layout ->
selfc_amd.data -> SelfCInvNet on the device -> test_rescaling.py's metrics, next to the CPU oracle on the same frames and
the same STP noise.  Runs on a synthetic folder always, and on the real Vid4 frames + selfc_large_pretrain.pth when
SELFC_VID4_ROOT and SELFC_PRETRAIN point at them (the assets do not ship with the reference: .MISSING_LARGE_BLOBS)."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from selfc_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def _write_groups(root, seq, n_groups, h, w, seed):
    from PIL import Image
    d = os.path.join(root, seq, "ds_7_to_7_new")
    os.makedirs(d)
    gen = torch.Generator().manual_seed(seed)
    names = []
    for gi in range(n_groups):
        name = f"{gi:04d}"
        os.makedirs(os.path.join(d, name))
        low = torch.rand(1, 3, h // 8, w // 8, generator=gen)
        base = torch.nn.functional.interpolate(low, size=(h, w), mode="bicubic", align_corners=False)[0]
        for i in range(7):                                   # a slowly drifting, lightly textured clip
            fr = (torch.roll(base, shifts=i, dims=2) + 0.03 * torch.randn(3, h, w, generator=gen)).clamp(0, 1)
            Image.fromarray((fr.permute(1, 2, 0).numpy() * 255).round().astype(np.uint8)).save(os.path.join(d, name, f"im{i + 1}.png"))
        names.append(name)
    with open(os.path.join(d, "testlist.txt"), "w") as fh:
        fh.write("\n".join(names) + "\n")


def test_folder_of_groups_through_the_hip_path(dev, tmp_path):
    import eval_vid4
    g = load_golden("g8_large_stack")
    s = load_golden("g7_stp_gmm")
    from selfc_amd import GlobalVar
    from selfc_amd.modules.SelfC_GMM_arch_inv import SelfCInvNet
    GlobalVar.set_Temporal_LEN(7)
    net = SelfCInvNet(eval_vid4.OPT, 3, 3, "D2DTNet", [4, 4], 2)
    sd = {k: v for k, v in g.items() if k.startswith("operations.")}
    sd.update({"stp_net." + k: v for k, v in s.items() if k.split(".")[0] in ("local_m1", "local_m2", "global_m1", "global_m2", "other_stp_modules", "tail_gmm")})
    net.load_state_dict(sd, strict=True)
    ckpt = str(tmp_path / "pretrain.pth")
    torch.save({"module." + k: v for k, v in net.state_dict().items()}, ckpt)          # as DistributedDataParallel saves it
    for seq, (h, w) in {"city": (64, 96), "walk": (48, 80)}.items():
        _write_groups(str(tmp_path), seq, 2, h, w, seed=len(seq))
    rep = eval_vid4.run(str(tmp_path), ckpt, oracle_groups=1, seqs=("city", "walk"))
    for seq in ("city", "walk"):
        r = rep[seq]
        assert r["groups"] == 2 and 5.0 < r["psnr_y"] < 80.0 and 0.0 < r["ssim_y"] <= 1.0
        assert r["max_psnr_diff_vs_oracle_dB"] < 0.02, r          # BASELINE: PSNR within 0.02 dB of the reference path


def test_vid4_config4_if_assets_present(dev):
    root, ckpt = os.environ.get("SELFC_VID4_ROOT"), os.environ.get("SELFC_PRETRAIN")
    if not (root and ckpt and os.path.isdir(root) and os.path.isfile(ckpt)):
        pytest.skip("Vid4 frames / selfc_large_pretrain.pth are not available (set SELFC_VID4_ROOT and SELFC_PRETRAIN)")
    import eval_vid4
    rep = eval_vid4.run(root, ckpt, oracle_groups=1)
    for seq, r in rep.items():
        assert r["max_psnr_diff_vs_oracle_dB"] < 0.02, (seq, r)
