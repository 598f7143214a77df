"""Frame I/O (selfc_amd/data.py): the sampler against index streams captured from the reference's DistIterSampler (G13), the
folder convention / dataset / dataloader on a synthetic directory tree."""
import os
import random

import numpy as np
import torch

from conftest import load_golden


def test_dist_iter_sampler_matches_reference_streams():
    from selfc_amd.data import DistIterSampler
    g = load_golden("g13_sampler")
    for i, (n, world, rank, epoch, ratio) in enumerate(g["cfgs"].tolist()):
        s = DistIterSampler(list(range(n)), num_replicas=world, rank=rank, ratio=ratio)
        s.set_epoch(epoch)
        got = torch.tensor(list(iter(s)))
        assert torch.equal(got, g[f"idx{i}"]), (n, world, rank)
        assert len(s) == len(got)
    # the ranks of one epoch partition the enlarged index list
    parts = []
    for r in range(4):
        s = DistIterSampler(list(range(9)), num_replicas=4, rank=r, ratio=2)
        parts += list(iter(s))
    assert len(parts) == 4 * 5 and sorted(set(parts)) == list(range(9))


def _make_tree(root, nvid=3, nframes=7, h=20, w=24, npy_for=None):
    from PIL import Image
    rng = np.random.RandomState(0)
    lines = []
    for v in range(nvid):
        rel = f"{v:05d}/0001"
        d = os.path.join(root, rel)
        os.makedirs(d)
        frames = rng.randint(0, 256, size=(nframes, h, w, 3), dtype=np.uint8)
        if npy_for is not None and v == npy_for:
            np.save(os.path.join(d, "clip.npy"), frames)
            for i in range(1, nframes):              # directory entry count decides N: keep it at nframes
                open(os.path.join(d, f"pad{i}"), "w").close()
        else:
            for i in range(nframes):
                Image.fromarray(frames[i]).save(os.path.join(d, f"im{i + 1}.png"))
        lines.append(rel)
    lst = os.path.join(root, "list.txt")
    with open(lst, "w") as fh:
        fh.write("\n".join(reversed(lines)) + "\n")     # unsorted on purpose
    return lst


def test_dataset_folder_convention_and_augmentation(tmp_path):
    from selfc_amd import GlobalVar
    from selfc_amd.data import SeptupletDataset, create_dataloader, get_vid_paths
    root = str(tmp_path)
    lst = _make_tree(root, npy_for=1)
    paths = get_vid_paths(root, lst)
    assert len(paths) == 3 and paths == sorted(paths) and paths[0][0].endswith("00000/0001/im1.png") and len(paths[0]) == 7
    test = SeptupletDataset({"dataroot_GT": root, "dataroot_list": lst, "phase": "test", "video_len": 7, "GT_size": 16})
    assert GlobalVar.get_Temporal_LEN() == 7
    item = test[0]
    assert item["GT"].shape == (3, 7, 20, 24) and item["GT"].dtype == torch.float32
    assert 0.0 <= float(item["GT"].min()) and float(item["GT"].max()) <= 1.0
    assert item["GT_path"] == paths[0][0]
    # PNG and clip.npy routes give the same kind of tensor; values are k/255
    npy_item = test[1]
    assert npy_item["GT"].shape == (3, 7, 20, 24)
    assert torch.allclose(npy_item["GT"] * 255, (npy_item["GT"] * 255).round(), atol=1e-4)
    # training: one crop / flip / rot per clip -> every frame is the same window of its source frame
    random.seed(3)
    train = SeptupletDataset({"dataroot_GT": root, "dataroot_list": lst, "phase": "train", "video_len": 7, "GT_size": 16,
                              "use_flip": True, "use_rot": True})
    full = test[0]["GT"]
    for _ in range(8):
        clip = train[0]["GT"]
        assert clip.shape == (3, 7, 16, 16)
        cands = []
        for t in range(7):
            f = full[:, t]
            found = None
            for rot in (False, True):
                for vf in (False, True):
                    for hf in (False, True):
                        for y in range(0, 5):
                            for x in range(0, 9):
                                win = f[:, y:y + 16, x:x + 16]
                                if hf:
                                    win = win.flip(2)
                                if vf:
                                    win = win.flip(1)
                                if rot:
                                    win = win.transpose(1, 2)
                                if torch.equal(win, clip[:, t]):
                                    found = (rot, vf, hf, y, x)
            assert found is not None
            cands.append(found)
        assert len(set(cands)) == 1, cands
    loader = create_dataloader(train, {"phase": "train", "batch_size": 2, "n_workers": 0}, {"dist": False, "gpu_ids": [0]})
    batch = next(iter(loader))
    assert batch["GT"].shape == (2, 3, 7, 16, 16)                  # data['GT'] (B,C,T,H,W) as feed_data expects


def test_multistep_lr_restart_product_class_matches_reference_trace():
    """selfc_amd.train.MultiStepLR_Restart against the lr trace of the reference's scheduler (G11)."""
    from selfc_amd.train import MultiStepLR_Restart
    g = load_golden("g11_lr_trace")
    opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=1e-4)
    sch = MultiStepLR_Restart(opt, [3, 6, 9], restarts=[5], weights=[0.5], gamma=0.5, clear_state=False)
    trace = []
    for _ in range(12):
        opt.step()
        sch.step()
        trace.append(opt.param_groups[0]["lr"])
    assert torch.allclose(torch.tensor(trace, dtype=torch.float64), g["lr"], rtol=0, atol=1e-18)
    # repeated milestones multiply gamma once per occurrence; clear_state drops the optimizer state at a restart
    opt2 = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
    sch2 = MultiStepLR_Restart(opt2, [2, 2], restarts=[4], weights=[1.0], gamma=0.1, clear_state=True)
    lrs = []
    for _ in range(4):
        opt2.step()
        sch2.step()
        lrs.append(opt2.param_groups[0]["lr"])
    assert abs(lrs[1] - 0.01) < 1e-12 and abs(lrs[3] - 1.0) < 1e-12 and len(opt2.state) == 0


def test_prefetcher_preserves_sampler_order_and_batches():
    """DevicePrefetcher (the loader side of SURVEY 8f3) must hand out exactly the batches of the loader it wraps, in order: the
    DistIterSampler index stream pinned by G13 is unchanged by the background thread, with worker processes and at any depth."""
    import torch.utils.data as tud
    from selfc_amd.data import DevicePrefetcher, DistIterSampler

    class Idx(tud.Dataset):
        def __len__(self):
            return 11

        def __getitem__(self, i):
            return {"GT": torch.full((3, 2, 4, 4), float(i)), "index": i}

    g = load_golden("g13_sampler")
    n, world, rank, epoch, ratio = g["cfgs"].tolist()[0]
    ds = list(range(n))
    s = DistIterSampler(ds, num_replicas=world, rank=rank, ratio=ratio)
    s.set_epoch(epoch)

    class Plain(tud.Dataset):
        def __len__(self):
            return n

        def __getitem__(self, i):
            return {"GT": torch.full((2, 2), float(i)), "index": i}
    for workers, depth in ((0, 1), (2, 2), (2, 4)):
        loader = tud.DataLoader(Plain(), batch_size=2, sampler=s, num_workers=workers, drop_last=True)
        got = [b["index"].tolist() for b in DevicePrefetcher(loader, "cpu", depth=depth)]
        want = g["idx0"].tolist()
        want = [want[i:i + 2] for i in range(0, len(want) - len(want) % 2, 2)]
        assert got == want, (workers, depth)
    loader = tud.DataLoader(Idx(), batch_size=3, shuffle=False)
    pf = DevicePrefetcher(loader, "cpu")
    assert len(pf) == len(loader)
    vals = [float(b["GT"][0, 0, 0, 0, 0]) for b in pf]
    assert vals == [0.0, 3.0, 6.0, 9.0]
    it = iter(pf)                      # abandoning an iterator half way must not hang (producer blocked on a full queue)
    next(it)
    del it

    def broken():
        yield {"GT": torch.zeros(1)}
        raise ValueError("decode failed")
    try:
        list(DevicePrefetcher(broken(), "cpu"))
        raise AssertionError("loader exception was swallowed")
    except ValueError as e:
        assert "decode failed" in str(e)


def test_prefetcher_does_not_hang_on_a_stalled_source():
    """Leaving the consumer loop (break, an exception in the train step) while the producer thread is stuck INSIDE the wrapped
    loader's next() - a stalled worker, an endless source - must return after the bounded join, not spin forever."""
    import threading
    import time
    from selfc_amd.data import DevicePrefetcher
    release = threading.Event()

    def stalled():
        yield {"GT": torch.zeros(1)}
        release.wait(30.0)             # the "stalled worker": far longer than the join timeout below
        yield {"GT": torch.ones(1)}

    it = iter(DevicePrefetcher(stalled(), "cpu", depth=1, join_timeout_s=0.3))
    next(it)
    t0 = time.monotonic()
    it.close()                         # what `break` / an exception does to the generator
    assert time.monotonic() - t0 < 3.0
    release.set()
