"""Which tensors differ, and at which stage, when a capturable trainer that loaded a reference-layout optimizer state is compared
with an eager one (tests/test_gpu_train.py::test_reference_optimizer_state_loads_into_a_capturable_trainer)."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch
from conftest import load_golden
from selfc_amd import GlobalVar, train
import test_gpu_train as TT
dev = torch.device("cuda:0"); GlobalVar.set_Temporal_LEN(7)
x = load_golden("g8_large_stack")["x"]
gt = x.reshape(1, 7, 3, 32, 48).transpose(1, 2).to(dev)
real_h, ref_l, _ = train.feed_data(gt, "sr_bd", 4)
def snap(net): return {k: v.detach().clone() for k, v in net.state_dict().items()}
def diff(a, b, tag):
    d = {k: float((a[k] - b[k]).abs().max()) for k in a}
    bad = sorted(d.items(), key=lambda kv: -kv[1])[:6]
    print(tag, "worst", bad[0][1], "n>1e-6:", sum(v > 1e-6 for v in d.values()), bad, flush=True)
net_a = TT._net(dev); tr_a = train.RescaleTrainer(net_a, dict(train.TRAIN_OPT_LARGE))
tr_a.optimize_parameters(real_h, ref_l); tr_a.optimize_parameters(real_h, ref_l)
w2 = snap(net_a); sd2 = TT._reference_layout(tr_a.optimizer_state_dict())
print("A flat?", tr_a.flat_optimizer, "state entries", len(sd2["state"]), "of", len(sd2["param_groups"][0]["params"]))
tr_a.optimize_parameters(real_h, ref_l); w3 = snap(net_a)
tr_a.optimize_parameters(real_h, ref_l); w4 = snap(net_a)
for mode in ("eager-capturable", "graph"):
    net_b = TT._net(dev); net_b.load_state_dict(w2)
    tr_b = train.RescaleTrainer(net_b, dict(train.TRAIN_OPT_LARGE), capturable=True)
    tr_b.load_optimizer_state_dict(sd2)
    print(mode, "B flat?", tr_b.flat_optimizer)
    if mode == "graph":
        tr_b.capture(real_h, ref_l, warmup=1)
    else:
        tr_b.optimize_parameters(real_h, ref_l)
    torch.cuda.synchronize(); diff(snap(net_b), w3, mode + " step3 vs A")
    tr_b.optimize_parameters(real_h, ref_l); torch.cuda.synchronize(); diff(snap(net_b), w4, mode + " step4 vs A")
    st = tr_b.optimizer_G.state[tr_b.optimizer_G.param_groups[0]["params"][0]]
    print(mode, "step now", float(st["step"]), "lr", float(tr_b.optimizer_G.param_groups[0]["lr"]))
# and: a fresh graph trainer WITHOUT any load, four steps, vs A's own four steps from scratch
net_c = TT._net(dev); tr_c = train.RescaleTrainer(net_c, dict(train.TRAIN_OPT_LARGE), capturable=True)
net_d = TT._net(dev); tr_d = train.RescaleTrainer(net_d, dict(train.TRAIN_OPT_LARGE))
tr_c.capture(real_h, ref_l, warmup=1); tr_d.optimize_parameters(real_h, ref_l)
torch.cuda.synchronize(); diff(snap(net_c), snap(net_d), "no-load: graph warm-up vs eager step1")
for i in range(2, 4):
    tr_c.optimize_parameters(real_h, ref_l); tr_d.optimize_parameters(real_h, ref_l)
    torch.cuda.synchronize(); diff(snap(net_c), snap(net_d), f"no-load: replay vs eager step{i}")
