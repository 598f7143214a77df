"""Frame I/O on the training / test side of the hot path: the reference's video-folder convention, its
iteration-oriented distributed sampler and its dataloader factory, without cv2 / lmdb.

  DistIterSampler     codes/data/data_sampler.py:12-59
  get_vid_paths       codes/data/util.py:59-86   (<root>/<list line>/im{1..N}.png, sorted)
  SeptupletDataset    codes/data/LQGTVID_dataset.py:14-231 for data_type 'img' (GT only; LR targets are generated on the
                      device by train.feed_data): RGB float [0,1], (C,T,H,W), one random crop / flip / rot90 per clip
  create_dataloader   codes/data/__init__.py:7-27

PNG decoding uses PIL (cv2.imread + the BGR->RGB swap of the reference yield the same RGB values); a clip directory may
instead hold one ``clip.npy`` (N,H,W,3) uint8 file, which avoids the PNG decode that starves 8 GPUs at 2 workers each.

Feeding the GPU (the reference: worker processes, codes/data/__init__.py:20-26, and a blocking ``.to(device)`` in feed_data,
SelfC_model.py:114-115):

  DevicePrefetcher    wraps any batch iterable: a background thread pulls batch k+1 (worker processes keep decoding) and
                      issues its host-to-device copy on a side stream while step k computes; order is preserved
  SyntheticSeptuplets the synthetic leg: uniform [0,1) clips drawn ON the device (no host work, no PCIe) from a per-rank seed"""
from __future__ import annotations

import math
import os
import queue
import random
import threading
import time
from typing import Iterable, Iterator, List, Optional

import numpy as np
import torch
import torch.utils.data as tud

from .global_var import GlobalVar


class DistIterSampler(tud.Sampler):
    """Iteration-oriented distributed sampler (behaviour of data_sampler.py:12-59): one "epoch" is `ratio` passes over the
    dataset, shuffled as a single permutation seeded by the epoch number; rank r reads positions r, r + world, ... of it, and
    every position is folded back onto a dataset index by modulo.  The loader is thus restarted only every `ratio` passes."""

    def __init__(self, dataset, num_replicas=None, rank=None, ratio=100):
        if num_replicas is None or rank is None:
            import torch.distributed as dist
            if not dist.is_available():
                raise RuntimeError("Requires distributed package to be available")
            num_replicas = dist.get_world_size() if num_replicas is None else num_replicas
            rank = dist.get_rank() if rank is None else rank
        self.dataset, self.num_replicas, self.rank, self.epoch = dataset, num_replicas, rank, 0
        self.num_samples = int(math.ceil(len(dataset) * ratio / num_replicas))       # per rank
        self.total_size = self.num_samples * num_replicas                              # padded so that it splits evenly

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        return self.num_samples

    def __iter__(self):
        perm = torch.randperm(self.total_size, generator=torch.Generator().manual_seed(self.epoch))
        mine = perm[self.rank::self.num_replicas] % len(self.dataset)
        if mine.numel() != self.num_samples:
            raise AssertionError("rank slice and num_samples disagree")
        return iter(mine.tolist())


def get_vid_paths(dataroot: str, data_list: str) -> List[List[str]]:
    """One list of frame paths per line of `data_list`: <dataroot>/<line>/im1.png .. im<N>.png with N = number of entries in
    that directory; the list of videos is sorted (data/util.py:59-86)."""
    vids = []
    with open(data_list) as fh:
        for line in fh.readlines():
            d = os.path.join(dataroot, line.strip())
            n = len(os.listdir(d))
            vids.append([os.path.join(d, "im" + str(i)) + ".png" for i in range(1, n + 1)])
    return sorted(vids)


def _read_frame(path: str) -> np.ndarray:
    """float32 HWC RGB in [0,1] (read_img1 + the channel swap of LQGTVID_dataset.py:132-134)."""
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"), dtype=np.float32) / 255.0


class SeptupletDataset(tud.Dataset):
    """opt keys as in the yml: dataroot_GT, dataroot_list, phase, GT_size, video_len, use_flip, use_rot, sample_num."""

    def __init__(self, opt: dict):
        super().__init__()
        self.opt = opt
        self.train = opt.get("phase", "train") == "train"
        self.paths_GT = get_vid_paths(opt["dataroot_GT"], opt["dataroot_list"])
        if not self.train and opt.get("sample_num"):
            self.paths_GT = self.paths_GT[0:opt["sample_num"]]
        GlobalVar.set_Temporal_LEN(opt["video_len"])            # the dataset sets the global clip length (:56)

    def __len__(self):
        return len(self.paths_GT)

    def _frames(self, paths: List[str]) -> np.ndarray:
        d = os.path.dirname(paths[0])
        npy = os.path.join(d, "clip.npy")
        if os.path.exists(npy):
            clip = np.load(npy, mmap_mode="r")[: len(paths)]
            return np.asarray(clip, dtype=np.float32) / 255.0     # (T,H,W,3)
        return np.stack([_read_frame(p) for p in paths])

    def __getitem__(self, index):
        paths = self.paths_GT[index]
        video_len = self.opt["video_len"]
        sel = paths[0:video_len] if video_len else paths         # video_len 7 / other: the first frames (:208-211)
        clip = self._frames(sel)                                   # (T,H,W,3) RGB [0,1]
        if self.train:
            gt = self.opt["GT_size"]
            t, h, w, _ = clip.shape
            if h < gt or w < gt:
                raise RuntimeError(f"frames of {paths[0]} are smaller than GT_size {gt} (the reference resizes with cv2 here)")
            rnd_h = random.randint(0, max(0, h - gt))
            rnd_w = random.randint(0, max(0, w - gt))
            clip = clip[:, rnd_h:rnd_h + gt, rnd_w:rnd_w + gt, :]
            # one draw per clip, applied to every frame (gen_aug_params :60-65, util.augment)
            if self.opt.get("use_flip") and random.random() < 0.5:
                clip = clip[:, :, ::-1, :]
            if self.opt.get("use_rot") and random.random() < 0.5:
                clip = clip[:, ::-1, :, :]
            if self.opt.get("use_rot") and random.random() < 0.5:
                clip = clip.transpose(0, 2, 1, 3)
        vid = torch.from_numpy(np.ascontiguousarray(clip.transpose(3, 0, 1, 2))).float()      # (C,T,H,W)
        return {"GT": vid, "LQ_path": paths[0], "GT_path": paths[0]}


def create_dataloader(dataset, dataset_opt: dict, opt: Optional[dict] = None, sampler=None):
    """data/__init__.py:7-27: training batch = batch_size // world when distributed, drop_last; test: batch as given.
    (pin_memory stays False as in the reference: DevicePrefetcher pins in its own thread.)"""
    phase = dataset_opt["phase"]
    if phase == "train":
        if opt and opt.get("dist"):
            world_size = torch.distributed.get_world_size()
            num_workers = dataset_opt["n_workers"]
            assert dataset_opt["batch_size"] % world_size == 0
            batch_size = dataset_opt["batch_size"] // world_size
            shuffle = False
        else:
            num_workers = dataset_opt["n_workers"] * len((opt or {}).get("gpu_ids", [0]))
            batch_size = dataset_opt["batch_size"]
            shuffle = True
        return tud.DataLoader(dataset, batch_size=batch_size, shuffle=shuffle, num_workers=num_workers, sampler=sampler,
                              drop_last=True, pin_memory=False)
    return tud.DataLoader(dataset, batch_size=dataset_opt["batch_size"], shuffle=False, num_workers=dataset_opt.get("n_workers", 0),
                          drop_last=False, pin_memory=False)


def _map_tensors(obj, fn):
    if torch.is_tensor(obj):
        return fn(obj)
    if isinstance(obj, dict):
        return {k: _map_tensors(v, fn) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_map_tensors(v, fn) for v in obj)
    return obj


class DevicePrefetcher:
    """Iterate `loader` with batch k+1 already on its way to `device` while step k runs.

    A daemon thread pulls the next batch from the loader (so DataLoader workers never wait for the training step) and
    enqueues its host-to-device copies on a side stream; ``__next__`` makes the consumer's current stream wait for that
    copy's event.  At most `depth` batches are in flight.  Batches come out in exactly the loader's order (one producer,
    one FIFO) - the DistIterSampler index streams are unchanged.  `device` "cpu": tensors pass through untouched (tests,
    dry runs); exceptions raised by the loader are re-raised in the consumer.

    A host tensor reaches the device through a plain ``.to(device)`` issued by the producer thread on the side stream: the HIP
    runtime stages the copy itself and blocks only that thread.  Measured on the training step of config 3
    (`tools/loader_probe.py`, `profiles/r3/train_loader_and_dp_step.txt`): 16.5-18.0 ms per step against 16.3-17.7 with
    device-generated data.  NOT done, on measurement: ``tensor.pin_memory()`` per batch (the first version of this class) or a
    ring of pinned buffers allocated by the producer thread - torch's caching pinned-memory allocator used from a second thread
    stalls the main thread's kernel launches, step 28-38 ms."""

    _END = object()

    def __init__(self, loader: Iterable, device, depth: int = 2, join_timeout_s: float = 5.0):
        self.loader, self.device, self.depth = loader, torch.device(device), max(1, int(depth))
        self.join_timeout_s = float(join_timeout_s)      # how long leaving the iteration waits for the producer thread
        self.cuda = self.device.type == "cuda"
        self.stream = torch.cuda.Stream(device=self.device) if self.cuda else None

    def _to_device(self, t: torch.Tensor) -> torch.Tensor:
        """producer thread, side stream current"""
        return t.to(self.device, non_blocking=True) if t.is_cuda else t.to(self.device)

    def __len__(self):
        return len(self.loader)

    def _producer(self, it: Iterator, q: "queue.Queue", stop: threading.Event):
        try:
            if self.cuda:
                torch.cuda.set_device(self.device)
            for batch in it:
                if self.cuda:
                    with torch.cuda.stream(self.stream):
                        moved = _map_tensors(batch, self._to_device)
                        ev = self.stream.record_event()
                    item = (moved, ev)
                else:
                    item = (batch, None)
                while not stop.is_set():
                    try:
                        q.put(item, timeout=0.1)
                        break
                    except queue.Full:
                        continue
                if stop.is_set():
                    return
            q.put((self._END, None))
        except BaseException as e:  # noqa: BLE001 - handed to the consumer
            q.put((e, None))

    def __iter__(self):
        q: "queue.Queue" = queue.Queue(maxsize=self.depth)
        stop = threading.Event()
        th = threading.Thread(target=self._producer, args=(iter(self.loader), q, stop), daemon=True)
        th.start()
        try:
            while True:
                item, ev = q.get()
                if item is self._END:
                    return
                if isinstance(item, BaseException):
                    raise item
                if ev is not None:
                    cur = torch.cuda.current_stream(self.device)
                    cur.wait_event(ev)
                    _map_tensors(item, lambda t: t.record_stream(cur) if t.is_cuda else None)
                yield item
        finally:
            stop.set()
            # unblock a producer waiting on a full queue - for a bounded time: one stuck inside the wrapped loader's next()
            # (a stalled worker, an endless source) cannot be interrupted, and the consumer leaving its loop (break, an
            # exception in the train step, KeyboardInterrupt) must not hang on it.  The thread is a daemon: it is abandoned.
            deadline = time.monotonic() + self.join_timeout_s
            while th.is_alive() and time.monotonic() < deadline:
                try:
                    q.get_nowait()
                except queue.Empty:
                    th.join(timeout=0.05)


class SyntheticSeptuplets:
    """Endless stream of data['GT'] batches (B,C,T,H,W), uniform [0,1), drawn on `device` from its own generator
    (seed: per rank).  What tools/train_synthetic.py and bench_train.py feed - Vimeo-shaped crops without a dataset."""

    def __init__(self, batch: int, t_len: int, size: int, device, seed: int):
        self.shape = (batch, 3, t_len, size, size)
        self.device = torch.device(device)
        self.gen = torch.Generator(device=self.device).manual_seed(seed)

    def __iter__(self):
        return self

    def __next__(self):
        return {"GT": torch.rand(self.shape, device=self.device, generator=self.gen)}
