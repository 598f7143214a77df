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


def test_host_fed_training_step_keeps_up_with_device_generated_data(dev):
    """SURVEY 8 f3, second half: a loader that does not starve the GPU.  The same eager training step fed (a) by batches drawn
    on the device and (b) by host batches through a DataLoader-style iterable + DevicePrefetcher must take about the same
    time, and the prefetcher must deliver the loader's batches bit-exactly and in order.  (The first DevicePrefetcher pinned
    every batch from its thread: torch's pinned-memory allocator used from a second thread stalled the main thread's
    launches and the step took 2.2x as long - `tools/loader_probe.py`; the bar below is where that regression shows.)"""
    import time
    from selfc_amd import GlobalVar, data, train
    from selfc_amd.modules.SelfC_GMM_arch_inv import SelfCInvNet
    GlobalVar.set_Temporal_LEN(7)
    torch.manual_seed(3)
    opt = {"global_module": "nonlocal", "stp_blk_num": 6, "fh_loss": "gmm", "scale": 4, "gmm_k": 5}
    net = SelfCInvNet(opt, 3, 3, "D2DTNet", [4, 4], 2).to(dev)
    tr = train.RescaleTrainer(net, dict(train.TRAIN_OPT_LARGE))
    gen = torch.Generator().manual_seed(5)
    pool = [torch.rand((8, 3, 7, 144, 144), generator=gen) for _ in range(3)]

    def host_batches(n):
        for i in range(n):
            yield {"GT": pool[i % 3], "index": torch.tensor([i])}

    got = list(data.DevicePrefetcher(host_batches(5), dev, depth=2))
    assert [int(b["index"]) for b in got] == list(range(5))
    assert all(torch.equal(b["GT"].cpu(), pool[i % 3]) for i, b in enumerate(got))

    def time_steps(feed, n=10, skip=4):
        ts = []
        for _ in range(n):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            real_h, ref_l, _ = train.feed_data(next(feed)["GT"], "sr_bd", 4)
            tr.optimize_parameters(real_h, ref_l)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        return sorted(ts[skip:])[(n - skip) // 2]                      # median of the steady steps

    t_dev = time_steps(iter(data.SyntheticSeptuplets(8, 7, 144, dev, 1)))
    t_host = time_steps(iter(data.DevicePrefetcher(host_batches(64), dev, depth=2)))
    assert t_host <= 1.35 * t_dev, (t_host, t_dev)
