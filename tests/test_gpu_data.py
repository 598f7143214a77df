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


#: Vid4's four sequences: HR frame size and clip length (README.md:36-58 of the reference names the set; 576x720 / 576x704 / 480x720,
#: 41 / 34 / 49 / 47 frames) - the frames and selfc_large_pretrain.pth themselves are not available (.MISSING_LARGE_BLOBS)
VID4_GEOMETRY = {"calendar": (576, 720, 41), "city": (576, 704, 34), "foliage": (480, 720, 49), "walk": (480, 720, 47)}


def _synthetic_clip(h, w, n, seed):
    """n frames (n,3,h,w) of a drifting low-frequency pattern with light texture, in [0,1]"""
    gen = torch.Generator().manual_seed(seed)
    low = torch.rand(1, 3, h // 16, w // 16, generator=gen)
    base = torch.nn.functional.interpolate(low, size=(h, w), mode="bicubic", align_corners=False)[0]
    return torch.stack([(torch.roll(base, shifts=(i, 2 * i), dims=(1, 2)) + 0.04 * torch.randn(3, h, w, generator=gen)).clamp(0, 1)
                        for i in range(n)])


@pytest.mark.parametrize("seq", sorted(VID4_GEOMETRY))
def test_vid4_geometry_and_clip_lengths_against_the_oracle(dev, seq):
    """Config 4's WORKLOAD SHAPE without its assets: synthetic clips at Vid4's exact frame sizes and clip
    lengths through the restated test loop (SelfC_model.py:185-250, test_rescaling.py:65-153) - the whole clip cut into GOPs
    of 7 with the last GOP padded by repeating the final frame (harness.rescale_video), and one 7-frame group through
    harness.rescale_test incl. the reference's redundant tail pass.  Against the oracle: the TAIL GOP (the padded one) on
    the WHOLE frame - LR, HF (l2 STP head: no noise) and reconstruction at 1e-3, Y-PSNR of every frame within 0.02 dB of the
    oracle's (north star) - and the first GOP on the four corner crops (the :615 method of test_gpu_parity.py: dependency
    radius 64 latent pixels).  Latents are 144x180 / 144x176 / 120x180: partial tile rows and columns in both kernel tilings."""
    from conftest import rel_err, record, subdict
    from oracle import selfc_oracle as O
    from selfc_amd import GlobalVar, harness
    from selfc_amd.modules.SelfC_GMM_arch_inv import SelfCInvNet
    GlobalVar.set_Temporal_LEN(7)
    torch.set_num_threads(min(16, os.cpu_count() or 16))      # the oracle's CPU convs collapse on a 256-thread host (bench.py calibrates the same way)
    H, W, n = VID4_GEOMETRY[seq]
    g, s = load_golden("g8_large_stack"), load_golden("g7_stp_l2_full_rev")
    net = SelfCInvNet({"global_module": "nonlocal", "stp_blk_num": 6, "fh_loss": "l2", "scale": 4, "gmm_k": 5}, 3, 3, "D2DTNet", [4, 4], 2)
    sd = {k: v for k, v in g.items() if k.startswith("operations.")}
    sd.update({k: v for k, v in s.items() if k.startswith("stp_net.")})
    net.load_state_dict(sd, strict=True)
    net.to(dev).eval()
    clip = _synthetic_clip(H, W, n, seed=len(seq) + n)
    out = harness.rescale_video(net, clip.to(dev), gops_per_call=2)
    assert out["frames"] == list(range(n)) and out["lr"].shape == (n, 3, H // 4, W // 4) and out["rec"].shape == (n, 3, H, W)
    slices = harness.gop_slices(n)
    assert len(slices) == -(-n // 7) and slices[-1][-1] == n - 1 and len(slices[-1]) == 7
    # --- tail GOP (n % 7 real frames + repeats of the last one), whole frame, against the oracle
    tail = slices[-1]
    keep = n - tail[0]
    x = clip[tail]
    with torch.no_grad():
        z, _ = net(x=x.to(dev), rev=False)
        from selfc_amd.modules.Quantization import Quantization
        lrq = Quantization()(z[:, :3])
        xr, hf = net(x=lrq, rev=True)
    # the video loop's tail frames ARE this GOP's first `keep` frames
    assert torch.equal(out["lr"][tail[0]:].cpu(), lrq[:keep].cpu()) and torch.equal(out["rec"][tail[0]:].cpu(), xr[:keep, :3].cpu())
    if seq in ("city", "walk"):
        # (the whole-frame oracle pass is the expensive part: two of the four sequences - one per frame height - take it; these two
        # stop at the loop, the padded tail and the corner crops below)
        z_or = None
    else:
        z_or = O.large_fwd(g, x, 7)
    if z_or is not None:
        _vid4_whole_frame(seq, g, s, x, z, z_or, lrq, hf, xr, harness, O, dev)
    _vid4_corners(g, net, clip, slices, H, W, out, harness, O, dev)


def _vid4_whole_frame(seq, g, s, x, z, z_or, lrq, hf, xr, harness, O, dev):
    from conftest import rel_err, record, subdict
    e_z = rel_err(z.cpu(), z_or)
    lr_or = O.quantize(z_or[:, :3])
    # quantisation: the LR latent is within ~1e-4 absolute of the oracle's, i.e. +-0.03 of a 1/255 step - a value that close to a
    # rounding boundary lands on the neighbouring step (expected share = 2 x 255 x mean|error|: measured 1.8 %), never further.  The
    # HIP path's reverse is therefore compared on the HIP path's OWN quantised LR (what the caller feeds it) against the oracle on it
    flips = float((lrq.cpu() != lr_or).float().mean())
    record(f"Vid4 geometry {seq}: fraction of LR pixels on the neighbouring quantisation step", flips)
    assert flips < 5e-2 and float((lrq.cpu() - lr_or).abs().max()) <= 1.0 / 255 + 1e-6
    hf_or = O.stp_v2_parameters(subdict(s, "stp_net"), lrq.cpu(), 7)
    e_hf = rel_err(hf.cpu(), hf_or)
    x_or = O.large_inv_from_latent(g, torch.cat((lrq.cpu(), hf_or), 1), 7)
    e_x = rel_err(xr.cpu(), x_or)
    assert max(e_z, e_hf, e_x) < 1e-3, (e_z, e_hf, e_x)
    p_hip = harness.psnr_y(xr[:, :3], x.to(dev))
    p_or = O.psnr_per_frame(O.rgb_to_y(x_or[:, :3]), O.rgb_to_y(x))         # Y channel, as test_rescaling.py:93-96 / harness.psnr_y
    d_psnr = record(f"Vid4 geometry {seq}: max per-frame |Y-PSNR(HIP) - Y-PSNR(oracle)| dB", max(abs(a - b) for a, b in zip(p_hip, p_or)))
    assert d_psnr < 0.02, (p_hip, p_or)


def _vid4_corners(g, net, clip, slices, H, W, out, harness, O, dev):
    from conftest import rel_err
    # --- first GOP: the four corners of the frame against the oracle on crops (forward latent and inverse)
    x0 = clip[slices[0]]
    h, w, C, M = H // 4, W // 4, 112, 64
    with torch.no_grad():
        z0, _ = net(x=x0.to(dev), rev=False)
        z0 = z0.cpu()
        zq = torch.cat((O.quantize(z0[:, :3]), z0[:, 3:]), 1)
        xr0 = net.inverse_from_latent(zq.to(dev)).cpu()
    for r0, c0 in ((0, 0), (0, w - C), (h - C, 0), (h - C, w - C)):
        rs = slice(0, C - M) if r0 == 0 else slice(M, C)
        cs = slice(0, C - M) if c0 == 0 else slice(M, C)
        xc = x0[:, :, 4 * r0:4 * (r0 + C), 4 * c0:4 * (c0 + C)].contiguous()
        zc = O.large_fwd(g, xc, 7)
        assert rel_err(z0[:, :, r0:r0 + C, c0:c0 + C][:, :, rs, cs], zc[:, :, rs, cs]) < 1e-3
        xo = O.large_inv_from_latent(g, zq[:, :, r0:r0 + C, c0:c0 + C].contiguous(), 7)
        hrs, hcs = slice(4 * rs.start, 4 * rs.stop), slice(4 * cs.start, 4 * cs.stop)
        assert rel_err(xr0[:, :, 4 * r0:4 * (r0 + C), 4 * c0:4 * (c0 + C)][:, :, hrs, hcs], xo[:, :, hrs, hcs]) < 1e-3
    # --- the reference's own loop on one 7-frame group, with its discarded extra pass (SelfC_model.py:203-209): same outputs
    fl, fh = harness.rescale_test(net, x0.to(dev), reference_tail_pass=True)
    assert torch.equal(fl.cpu(), out["lr"][:7].cpu()) and torch.equal(fh.cpu(), out["rec"][:7].cpu())


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
