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
def mk(cap):
    net = TT._net(dev); return net, train.RescaleTrainer(net, dict(train.TRAIN_OPT_LARGE), capturable=cap)
def cmp(a, b, tag): print(tag, float((a - b).abs().max()), "rel", float((a - b).norm() / (b.norm() + 1e-30)), flush=True)
nx, tx = mk(False); ny, ty = mk(True)
for step in (1, 2, 3):
    for t_ in (tx, ty):
        t_._zero_grad(); t_._forward_backward(real_h, ref_l)
    torch.cuda.synchronize()
    cmp(ty.sink.flat, tx.sink.flat, f"step {step}: flat gradient (capturable vs not)")
    for t_ in (tx, ty):
        t_._sync_grads(); t_._clip_and_step()
    torch.cuda.synchronize()
    cmp(ty.sink.flat, tx.sink.flat, f"step {step}: clipped gradient")
    cmp(ty.sink.flat_param.data, tx.sink.flat_param.data, f"step {step}: weights after")
    sx = tx.optimizer_G.state[tx.sink.flat_param]; sy = ty.optimizer_G.state[ty.sink.flat_param]
    cmp(sy["exp_avg"], sx["exp_avg"], f"step {step}: exp_avg"); cmp(sy["exp_avg_sq"], sx["exp_avg_sq"], f"step {step}: exp_avg_sq")
    print("steps", float(sx["step"]), float(sy["step"]), "norms", float(tx.grad_norm), float(ty.grad_norm))
# the same trainer kind twice: is a trainer's second step reproducible at all?
nz, tz = mk(False)
for step in (1, 2):
    tz._zero_grad(); tz._forward_backward(real_h, ref_l); tz._sync_grads(); tz._clip_and_step()
torch.cuda.synchronize()
nw, tw = mk(False)
for step in (1, 2):
    tw._zero_grad(); tw._forward_backward(real_h, ref_l); tw._sync_grads(); tw._clip_and_step()
torch.cuda.synchronize()
cmp(tz.sink.flat_param.data, tw.sink.flat_param.data, "two non-capturable trainers after 2 steps")
cmp(tz.sink.flat_param.data, tx.sink.flat_param.data, "(vs tx after 3)")
