"""Round-6 fixture: a K-step TRAINING TRAJECTORY of the reference modules on CPU (fp32, stock autograd, torch Adam).

    python tools/make_golden_r6.py            # writes tests/golden/g18_train_trajectory.npz

Runs only in the build container (imports /root/reference read-only, nothing of its source is stored).  The step is the
restatement of SelfCModel.optimize_parameters (SelfC_model.py:148-183) that tools/make_golden.py uses for G11 - l2 forward
fit + l1 (eps 1e-6) reconstruction, x 144*144*3, clip_grad_norm_ 10, Adam(1e-4, (0.9, 0.999), wd 1e-14) - repeated K = 20
times on ONE fixed clip (g8_large_stack's x, 7x3x32x48) from g8_large_stack + g7_stp_l2_full_rev's weights (fh_loss l2: no
RNG in the step).  Stored: per-step l_forw_fit / l_back_rec / loss / total gradient norm, and the weight UPDATE after K steps
(final - initial, all 350 trainable tensors in named_parameters() order) as int16 with one global scale (6.7 MB instead of
13.5 MB; resolution 3e-5 of the largest update), plus its per-tensor L2 norms in float64."""
import os
import sys
import types

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/codes")
_tv = types.ModuleType("torchvision")
_tvo = types.ModuleType("torchvision.ops")
_tv.ops = _tvo
sys.modules["torchvision"] = _tv
sys.modules["torchvision.ops"] = _tvo

import numpy as np  # noqa: E402
import torch  # noqa: E402

from global_var import GlobalVar  # noqa: E402
import models.modules.SelfC_GMM_arch_inv as GA  # noqa: E402
from models.modules.Quantization import Quantization  # noqa: E402
from models.Guassian import Guassian_downsample  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
K = 20


def load(name):
    with np.load(os.path.join(OUT, name + ".npz")) as z:
        return {k: torch.from_numpy(np.asarray(z[k])) for k in z.files}


def main():
    torch.set_num_threads(8)
    GlobalVar.set_Temporal_LEN(7)
    g8, g7 = load("g8_large_stack"), load("g7_stp_l2_full_rev")
    opt = {"global_module": "nonlocal", "stp_blk_num": 6, "fh_loss": "l2", "scale": 4, "gmm_k": 5}
    net = GA.SelfCInvNet(opt, 3, 3, "D2DTNet", [4, 4], 2)
    sd = {k: v for k, v in g8.items() if k.startswith("operations.")}
    sd.update({k: v for k, v in g7.items() if k.startswith("stp_net.")})
    net.load_state_dict(sd, strict=True)
    net.train()
    real_h = g8["x"]
    ref_l = Guassian_downsample(real_h.transpose(0, 1)).transpose(0, 1)
    names = [n for n, p in net.named_parameters() if p.requires_grad]
    prms = [p for p in net.parameters() if p.requires_grad]
    before = [p.detach().clone() for p in prms]
    optim = torch.optim.Adam(prms, lr=1e-4, weight_decay=1e-14, betas=(0.9, 0.999))
    rows = []
    for k in range(K):
        optim.zero_grad()
        out_f, loss_c = net(x=real_h, rev=False)
        lr_bq = out_f[:, :3]
        l_fit = 1.0 * ((lr_bq - ref_l.detach()) ** 2).mean(-1).mean(-1).mean(-1).mean(-1)
        y_ = Quantization()(lr_bq)
        x_s, _ = net(x=y_, rev=True)
        d_ = real_h - x_s[:, :3]
        l_rec = 1.0 * torch.sqrt(d_ * d_ + 1e-6).mean(-1).mean(-1).mean(-1).mean(-1)
        loss = (l_fit + l_rec + loss_c.mean() * 0) * 144 * 144 * 3
        loss.backward()
        total = float(torch.nn.utils.clip_grad_norm_(prms, 10))
        optim.step()
        rows.append((l_fit.item(), l_rec.item(), loss.item(), total))
        print(k + 1, rows[-1], flush=True)
    upd = torch.cat([(p.detach() - b).flatten() for p, b in zip(prms, before)])
    norms = np.array([float((p.detach() - b).double().norm()) for p, b in zip(prms, before)], dtype=np.float64)
    scale = float(upd.abs().max()) / 32767.0
    q = torch.round(upd / scale).clamp(-32767, 32767).to(torch.int16)
    back = q.float() * scale
    print("update: max %.3e, L2 %.6e, int16 round-trip relative L2 %.2e" % (float(upd.abs().max()), float(upd.norm()), float((back - upd).norm() / upd.norm())))
    rows = np.array(rows, dtype=np.float64)
    path = os.path.join(OUT, "g18_train_trajectory.npz")
    np.savez_compressed(path, l_forw_fit=rows[:, 0], l_back_rec=rows[:, 1], loss=rows[:, 2], grad_norm=rows[:, 3], names=np.array(names),
                        numels=np.array([p.numel() for p in prms], dtype=np.int64), update_q=q.numpy(), update_scale=np.float64(scale),
                        update_norms=norms, update_l2=np.float64(float(upd.double().norm())))
    print("g18_train_trajectory: %.2f MB" % (os.path.getsize(path) / 1e6))


if __name__ == "__main__":
    main()
