"""Where a host-fed training step loses time (diagnostic): per step, the wait for the batch, feed_data + the step's launches,
and the device time to drain, for device-generated and host-fed data."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from train_synthetic import _HostBatches

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
from selfc_amd import GlobalVar, data, train
from selfc_amd.modules.SelfC_GMM_arch_inv import SelfCInvNet
GlobalVar.set_Temporal_LEN(7)
torch.manual_seed(10)
opt = {"global_module": "nonlocal", "stp_blk_num": 6, "fh_loss": "gmm", "scale": 4, "gmm_k": 5}
net = SelfCInvNet(opt, 3, 3, "D2DTNet", [4, 4], 2).to(dev)
tr = train.RescaleTrainer(net, dict(train.TRAIN_OPT_LARGE), capturable=False)

def run(feed, label, n=12):
    rows = []
    for i in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gt = next(feed)["GT"]
        t1 = time.perf_counter()
        real_h, ref_l, _ = train.feed_data(gt, "sr_bd", 4)
        tr.optimize_parameters(real_h, ref_l)
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        rows.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
    r = rows[4:]
    print(label, "next %.2f  launch %.2f  drain %.2f  total %.2f ms" % tuple([sum(x[k] for x in r) / len(r) for k in range(3)] + [sum(sum(x) for x in r) / len(r)]), flush=True)

import queue, threading

def threaded(make_item, finish=None, depth=2):
    """producer thread variants: make_item(cpu batch) runs in the thread, finish(item) in the consumer"""
    q = queue.Queue(maxsize=depth)
    pool = _HostBatches(8, 144, 1, 100).pool
    def prod():
        torch.cuda.set_device(dev)
        i = 0
        while True:
            q.put(make_item(pool[i % 4])); i += 1
    threading.Thread(target=prod, daemon=True).start()
    while True:
        it = q.get()
        yield {"GT": finish(it) if finish else it}

side = torch.cuda.Stream()
def a_pin_to(t):
    with torch.cuda.stream(side):
        m = t.pin_memory().to(dev, non_blocking=True); ev = side.record_event()
    return (m, ev)
def fin_ev(it):
    torch.cuda.current_stream().wait_event(it[1]); return it[0]
def b_pageable(t):
    with torch.cuda.stream(side):
        m = t.to(dev); ev = side.record_event()
    return (m, ev)
ring = [torch.empty(8, 3, 7, 144, 144).pin_memory() for _ in range(4)]
evs = [None] * 4
cnt = [0]
def c_ring(t):
    k = cnt[0] % 4; cnt[0] += 1
    if evs[k] is not None: evs[k].synchronize()
    ring[k].copy_(t)
    with torch.cuda.stream(side):
        m = ring[k].to(dev, non_blocking=True); ev = side.record_event()
    evs[k] = ev
    return (m, ev)
def d_cpu(t):
    return t.clone()

run(iter(data.SyntheticSeptuplets(8, 7, 144, dev, 1)), "device feed            ")
run(iter(data.DevicePrefetcher(_HostBatches(8, 144, 1, 100), dev, depth=2)), "DevicePrefetcher       ")
run(threaded(a_pin_to, fin_ev), "thread: pin_memory+to  ")
run(threaded(b_pageable, fin_ev), "thread: pageable .to   ")
run(threaded(c_ring, fin_ev), "thread: pinned ring    ")
run(threaded(d_cpu, lambda t: t.to(dev)), "thread: cpu clone only ")
run(threaded(lambda t: t, lambda t: t.to(dev)), "thread: pass-through   ")
