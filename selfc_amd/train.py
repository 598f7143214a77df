"""Training-step harness: the parts of ``SelfCModel`` (codes/models/SelfC_model.py) that drive the hot path in training.

``SelfCModel`` itself cannot be imported without cv2 / thop / lmdb, so the caller-side logic is restated here with the
same names and option keys, around the HIP-backed ``SelfCInvNet``:

  feed_data            SelfC_model.py:93-132   pad short clips with the last frame, (B,C,T,H,W) -> (B*T,C,H,W), LR target
  ReconstructionLoss   modules/loss.py:5-21
  MultiStepLR_Restart  lr_scheduler.py:8-31
  RescaleTrainer.optimize_parameters  SelfC_model.py:153-184 (forward fit + quantise + reverse reconstruction, x144*144*3,
                       clip, Adam step)

Forward, reverse and every gradient run on the HIP kernels (selfc_amd/autograd.py); the loss reductions, the clip and
Adam are the reference's own torch calls.

Multi-GPU (SelfC_model.py:41-44, data/__init__.py:13-14: DistributedDataParallel around netG, global batch split over the
ranks).  Default here: pass the PLAIN net while torch.distributed is initialised with more than one rank - the trainer
then keeps the flat gradient buffer, broadcasts rank 0's parameters once, and between backward and clip issues ONE
all-reduce (SUM, / world) of that 13.5 MB buffer over RCCL (``GradSink.all_reduce``): the same averaged gradients DDP's
hooks produce, with the kernels still accumulating in place and the step still capturable (``capture()`` records
[zero, forward, backward] and [clip, Adam] as two hipGraphs with the collective between them).  Fallback: wrap the net in
DistributedDataParallel exactly as the reference does - parameter gradients are then ordinary ``.grad`` tensors that pass
through autograd and DDP's buckets (no flat buffer, no capture).
"""
from __future__ import annotations

import os
from collections import Counter, OrderedDict, defaultdict
from typing import Optional

import torch
import torch.nn as nn
from torch.optim.lr_scheduler import _LRScheduler

from . import _lib, autograd as ag, harness, runtime as rt
from .global_var import GlobalVar
from .modules.Quantization import Quantization


#: SELFC_FUSED_LOSS=0: the loss is the reference's torch expression again (~10 element-wise / reduction launches forward and ~12
#: backward per loss, 4.5 us each inside a replayed step) instead of selfc_recon_loss (two launches, the gradient in the same pass)
_FUSED_LOSS = os.environ.get("SELFC_FUSED_LOSS", "1") != "0"
#: SELFC_FUSED_ADAM=0: gradient clipping + Adam are torch's launches again (the staged norm, the clip, 16 multi-tensor launches)
#: instead of selfc_clip_adam (two launches on the flat buffer, torch's operation order, torch.optim.Adam's own state tensors)
_FUSED_ADAM = os.environ.get("SELFC_FUSED_ADAM", "1") != "0"


def _rows(t_: torch.Tensor):
    """(tensor, n_outer, inner, row stride) of a 4-D tensor whose rows [n] are contiguous blocks of C*H*W floats (a channel slice
    out[:, :3] of an NCHW tensor is such a view), or None"""
    if t_.dim() != 4 or t_.dtype != torch.float32 or not t_.is_cuda:
        return None
    n, c, h, w = t_.shape
    if n == 0 or c * h * w == 0:
        return None
    sn, sc, sh, sw = t_.stride()
    if sw != 1 or sh != w or sc != h * w or (n > 1 and sn < c * h * w):
        return None
    return t_, n, c * h * w, (sn if n > 1 else c * h * w)


class ReconLossFn(torch.autograd.Function):
    """weight * ReconstructionLoss(x, target) (loss.py:5-21) on selfc_recon_loss: the value (deterministic two-stage sum) and
    d loss / d x in one pass over x and target; backward scales that gradient by the incoming scalar (d / d target = its negative)."""

    @staticmethod
    def forward(ctx, x, target, l1, eps, weight):
        rx, rt_ = _rows(x), _rows(target)
        dev = x.device
        need = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        # only the target needs a gradient (the reverse loss: Reconstruction_back(real_H, x_samples)): store -d/dx at once
        flip = ctx.needs_input_grad[1] and not ctx.needs_input_grad[0]
        ctx.flip = flip
        grad = torch.empty(x.shape, dtype=torch.float32, device=dev) if need else None
        nb = int(_lib.lib().selfc_recon_loss_blocks())
        partial = torch.empty(nb, dtype=torch.float64, device=dev)
        out = torch.empty((), dtype=torch.float32, device=dev)
        if flip:      # loss(x, t) is symmetric in its arguments' difference up to the sign of the gradient: swap them
            rx, rt_ = rt_, rx
        rt.call("selfc_recon_loss", rx[0].data_ptr(), rx[3], rt_[0].data_ptr(), rt_[3], rx[1], rx[2], 1 if l1 else 0, float(eps), float(weight),
                None if grad is None else grad.data_ptr(), partial.data_ptr(), out.data_ptr(), _lib.stream_ptr())
        ctx.grad = grad
        return out

    @staticmethod
    def backward(ctx, g):
        gx = ctx.grad * g if ctx.grad is not None else None
        if ctx.flip:                  # ctx.grad is d loss / d target already
            return (None, gx, None, None, None)
        return (gx if ctx.needs_input_grad[0] else None, (-gx) if ctx.needs_input_grad[1] else None, None, None, None)


class ReconstructionLoss(nn.Module):
    """loss.py:5-21: 'l2' = (x-t)^2, 'l1' = sqrt((x-t)^2 + eps); mean over the four axes one after the other."""

    def __init__(self, losstype="l2", eps=1e-6):
        super().__init__()
        self.losstype = losstype
        self.eps = eps

    def forward(self, x, target, weight: float = 1.0):
        """weight: a factor folded into the fused kernel (the trainer's lambda_fit_forw / lambda_rec_back: one launch less each)"""
        if _FUSED_LOSS and self.losstype in ("l2", "l1") and x.shape == target.shape and _rows(x) is not None and _rows(target) is not None:
            # the mean over the four axes one after the other = sum / count (equal group sizes)
            return ReconLossFn.apply(x, target, self.losstype == "l1", self.eps, float(weight))
        if weight != 1.0:
            return weight * self.forward(x, target)
        if self.losstype == "l2":
            v = (x - target) ** 2
        elif self.losstype == "l1":
            diff = x - target
            v = torch.sqrt(diff * diff + self.eps)
        else:
            print("reconstruction loss type error!")
            return 0
        # the reference's v.mean(-1).mean(-1).mean(-1).mean(-1) (loss.py), as written there.  (Rounds 3-4 folded it into ONE mean over
        # the four axes: equal-sized groups, same value - but a single reduction of 3.5 M elements to one output is torch's multi-block
        # "global reduce" (partials + a semaphore zeroed by a memset in front of the kernel), and inside a captured step on ONE stream at
        # 8 septuplets per rank that kernel left its output unwritten: the step then REPORTED l_back_rec = l_forw_fit, the value the
        # recycled output block still held (tools/experiments/loss_alias_probe.py; the gradients - and so the training - were never
        # affected, they do not read the scalar).  Each stage of the chained form has either many outputs or a short inner dimension:
        # no cross-block stage, nothing for a graph replay to get wrong.)
        if os.environ.get("SELFC_LOSS_ONE_MEAN") == "1":
            return v.mean(dim=(-4, -3, -2, -1))
        return v.mean(-1).mean(-1).mean(-1).mean(-1)


class MultiStepLR_Restart(_LRScheduler):
    """Step schedule with warm restarts (behaviour of lr_scheduler.py:8-31): at a restart iteration the rate jumps to
    initial_lr * weight (optionally dropping the optimizer state), at a milestone it is multiplied by gamma once per
    occurrence of that milestone, otherwise it is left alone.  A restart takes precedence over a milestone."""

    def __init__(self, optimizer, milestones, restarts=None, weights=None, gamma=0.1, clear_state=False, last_epoch=-1):
        self.milestones = Counter(milestones)
        self.gamma, self.clear_state = gamma, clear_state
        self.restarts = list(restarts) if restarts else [0]
        self.restart_weights = list(weights) if weights else [1]
        if len(self.restarts) != len(self.restart_weights):
            raise AssertionError("restarts and their weights do not match.")
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        it, groups = self.last_epoch, self.optimizer.param_groups
        if it in self.restarts:
            if self.clear_state:
                self.optimizer.state = defaultdict(dict)
            w = self.restart_weights[self.restarts.index(it)]
            return [g["initial_lr"] * w for g in groups]
        factor = self.gamma ** self.milestones[it] if it in self.milestones else 1
        return [g["lr"] * factor if factor != 1 else g["lr"] for g in groups]


def feed_data(gt: torch.Tensor, distortion: str = "sr_bd", scale: int = 4, lq: Optional[torch.Tensor] = None):
    """data['GT'] (B,C,T,H,W) -> (real_H (B*T,C,H,W), ref_L (B*T,C,H/scale,W/scale), clip_length) (SelfC_model.py:93-132)."""
    t_len = GlobalVar.get_Temporal_LEN()
    clip_length = gt.size(2)
    if clip_length < t_len:
        pads = torch.stack([gt[:, :, -1]] * (t_len - clip_length), dim=2)
        gt = torch.cat([gt, pads], dim=2)
    real_h = gt.transpose(1, 2).reshape(-1, 3, gt.size(3), gt.size(4)).contiguous()
    if lq is not None:
        ref_l = lq.transpose(1, 2).reshape(-1, 3, lq.size(3), lq.size(4))
    elif distortion == "sr_bd":
        if scale != 4:
            raise NotImplementedError("the Gaussian LR target kernel is built for scale 4")
        ref_l = harness.gaussian_downsample(real_h)                 # Guassian_downsample (models/Guassian.py:7-52)
    elif distortion == "pytorch_bicubic":                            # (sic) F.upsample(mode='area') in the reference = 4x4 block mean
        if scale != 4:
            raise NotImplementedError("the area LR target is built for scale 4")
        from .modules.SelfC_GMM_arch_inv import FrequencyAnalyzer
        with torch.no_grad():
            ref_l = FrequencyAnalyzer(3)(real_h)[:, :3].contiguous()   # the low band of the split kernel is that block mean
    else:
        raise NotImplementedError(f"distortion {distortion!r}")
    return real_h, ref_l, clip_length


class RescaleTrainer:
    """optimize_parameters of SelfCModel (SelfC_model.py:153-184) around a HIP-backed net.

    train_opt keys are the yml's (train_rescaling_selfc_large.yml:95-121): lr_G, beta1, beta2, weight_decay_G,
    pixel_criterion_forw/back, lambda_fit_forw, lambda_rec_back, lambda_cond_prob, gradient_clipping, lr_scheme,
    lr_steps, lr_gamma, restarts, restart_weights, clear_state."""

    def __init__(self, netG: nn.Module, train_opt: dict, capturable: bool = False, flat_grads: bool = None,
                 data_parallel: bool = None, process_group=None, flat_params: bool = None):
        """capturable: prepare the optimizer for `capture()` (device-side step counter and learning rate).
        flat_grads: the parameters' `.grad` are views of ONE buffer that the weight-gradient kernels accumulate into
        directly (autograd.GradSink: no per-tensor accumulation launches, one memset instead of zero_grad, the clip on the
        flat buffer).  Default: on, unless the net is wrapped in DistributedDataParallel (its hooks need the gradients to
        pass through autograd).
        data_parallel: average the flat gradient buffer over the ranks of `process_group` once per step (module
        docstring).  Default: on when torch.distributed is initialised with more than one rank and the net is not
        wrapped in DistributedDataParallel; needs flat_grads.
        flat_params: (with flat_grads, default on) the parameters become views of one flat buffer and Adam updates that single
        tensor; False keeps the per-tensor optimizer."""
        self.netG = netG
        self.capturable = capturable
        self.graph = None
        self.graph_tail = None
        self.train_opt = train_opt
        self.Quantization = Quantization()
        self.netG.train()
        self.Reconstruction_forw = ReconstructionLoss(losstype=train_opt["pixel_criterion_forw"])
        self.Reconstruction_back = ReconstructionLoss(losstype=train_opt["pixel_criterion_back"])
        wd = train_opt.get("weight_decay_G") or 0
        optim_params = [v for k, v in netG.named_parameters() if v.requires_grad and "opticFlow_Net" not in k]
        self.optim_params = optim_params
        lr = train_opt["lr_G"]
        if capturable:
            lr = torch.tensor(float(lr), dtype=torch.float32, device=optim_params[0].device)
        self._adam_kw = dict(lr=lr, weight_decay=wd, betas=(train_opt["beta1"], train_opt["beta2"]), capturable=capturable)
        self.optimizer_G = torch.optim.Adam(optim_params, **self._adam_kw)
        self.log_dict = OrderedDict()
        self.grad_norm = None
        is_ddp = isinstance(netG, nn.parallel.DistributedDataParallel)
        if flat_grads is None:
            flat_grads = not is_ddp and optim_params[0].is_cuda
        self.sink = ag.GradSink(optim_params) if flat_grads else None
        # flat parameters: with the flat gradient buffer the parameters become views of ONE buffer too and Adam runs on that single
        # tensor (GradSink.flatten_params: 2.8 ms -> 0.1 ms of optimizer launches per step).  It stays a plain torch.optim.Adam with
        # the same hyper-parameters; the per-tensor optimizer is rebuilt if a step ever leaves parameters without a gradient
        # (those must be skipped, which a single flat tensor cannot express).
        self.flat_optimizer = False
        if self.sink is not None and flat_params is not False and optim_params[0].is_cuda and \
                len({p_.dtype for p_ in optim_params}) == 1 and len(self.sink.params) == len(optim_params):
            self.optimizer_G = torch.optim.Adam([self.sink.flatten_params()], **self._adam_kw)
            self.flat_optimizer = True
            rt.invalidate_weights()
        self.schedulers = []
        if train_opt.get("lr_scheme", "MultiStepLR") == "MultiStepLR":
            self.schedulers.append(MultiStepLR_Restart(self.optimizer_G, train_opt.get("lr_steps", []),
                                                       restarts=train_opt.get("restarts"), weights=train_opt.get("restart_weights"),
                                                       gamma=train_opt.get("lr_gamma", 0.1), clear_state=train_opt.get("clear_state")))
        else:
            raise NotImplementedError("MultiStepLR learning rate scheme is enough.")
        self.process_group = process_group
        self.world = 1
        if data_parallel is None:
            import torch.distributed as dist
            data_parallel = (not is_ddp and self.sink is not None and dist.is_available() and dist.is_initialized()
                             and dist.get_world_size(process_group) > 1)
        self.data_parallel = bool(data_parallel)
        if self.data_parallel:
            import torch.distributed as dist
            if is_ddp:
                raise RuntimeError("data_parallel=True averages the flat gradient buffer itself: pass the plain net, not a DistributedDataParallel wrapper")
            if self.sink is None:
                raise RuntimeError("data_parallel=True needs flat_grads (the collective runs on the flat gradient buffer)")
            if not (dist.is_available() and dist.is_initialized()):
                raise RuntimeError("data_parallel=True needs an initialised torch.distributed process group")
            self.world = dist.get_world_size(process_group)
            # what DistributedDataParallel does at construction: every rank starts from rank 0's parameters and buffers
            src = dist.get_global_rank(process_group, 0) if process_group is not None else 0
            with torch.no_grad():
                for t_ in list(nn.Module.parameters(netG)) + list(netG.buffers()):
                    ag.device_collective(t_.data, lambda h_: dist.broadcast(h_, src, group=process_group), process_group)
            rt.invalidate_weights()
        self.before_clip = None          # optional callable(trainer): runs after backward, before clipping (tests, logging)

    def loss_forward(self, out, y):
        return self.Reconstruction_forw(out, y, weight=self.train_opt["lambda_fit_forw"])

    def loss_backward(self, x, y):
        x_samples, _ = self.netG(x=y, rev=True)
        return self.Reconstruction_back(x, x_samples[:, :3, :, :], weight=self.train_opt["lambda_rec_back"])

    def _forward_backward(self, real_H: torch.Tensor, ref_L: torch.Tensor):
        """optimize_parameters (SelfC_model.py:153-170) up to and including loss.backward(); returns the loss tensors."""
        if self.flat_optimizer and not self.sink.params_attached():
            raise RuntimeError("the net's parameters no longer alias the trainer's flat buffer (netG.to() / .half() / a re-wrap after "
                               "RescaleTrainer was built): construct the trainer after moving the net, or pass flat_params=False")
        self._repack()
        output, loss_c = self.netG(x=real_H, rev=False)
        lam_c = self.train_opt.get("lambda_cond_prob", 0)
        # (lambda_cond_prob: 0 in every shipped yml - SelfC_model.py:157 multiplies the term away; no launches for it then)
        loss_c = loss_c.mean() * lam_c if lam_c else None
        lr_before_quant = output[:, :3, :, :]
        l_forw_fit = self.loss_forward(lr_before_quant, ref_L.detach())
        LR = self.Quantization(lr_before_quant)
        l_back_rec = self.loss_backward(real_H, LR)
        total = l_forw_fit + l_back_rec
        loss = (total if loss_c is None else total + loss_c) * float(144 * 144 * 3)
        with ag.grad_sink(self.sink), ag.background_wgrad():      # (joins the background weight gradients on its way out)
            loss.backward()
        if self.sink is not None:
            if self.data_parallel and not self.__dict__.get("_touched_checked"):
                # every rank must decide "flat or per-tensor optimizer" (and which tensors Adam skips) the same way: compare
                # the touched bitmaps once, at the first step (they are a function of the graph, not of the data)
                self._touched_checked = True
                self.sink.assert_touched_equal(self.process_group)
            if self.sink.detach_untouched() and self.flat_optimizer:
                self._per_tensor_optimizer()   # tensors without a gradient must be skipped: one flat tensor cannot do that
        return l_forw_fit.detach(), l_back_rec.detach(), (0.0 if loss_c is None else loss_c.detach()), loss.detach()

    def _repack(self):
        """The weights changed with the last optimizer step: repack every module's kernel-layout tensors in ONE gather
        (runtime.PackGroup) instead of module by module on first use - ~250 small launches per step less.  Modules outside the
        group (other net classes) keep repacking themselves."""
        grp = self.__dict__.get("_pack_group")
        if grp is None:
            grp = self._pack_group = rt.pack_group_for(self.netG.module if hasattr(self.netG, "module") else self.netG)
        grp.refresh()

    def _per_tensor_optimizer(self):
        """Fall back from the flat Adam to the per-tensor one (same hyper-parameters; moments so far are carried over)."""
        if self.graph is not None:
            raise RuntimeError("parameters without a gradient appeared after capture(): capture again")
        old = self.optimizer_G
        new = torch.optim.Adam(self.optim_params, **self._adam_kw)
        st = old.state.get(self.sink.flat_param)
        if st:
            for p_, v in zip(self.sink.params, self.sink.views):
                off = v.storage_offset()
                new.state[p_] = {"step": st["step"].clone() if torch.is_tensor(st["step"]) else st["step"],
                                 "exp_avg": st["exp_avg"][off:off + p_.numel()].view(p_.shape).clone(),
                                 "exp_avg_sq": st["exp_avg_sq"][off:off + p_.numel()].view(p_.shape).clone()}
        for g_new, g_old in zip(new.param_groups, old.param_groups):
            for k_ in ("lr", "initial_lr"):
                if k_ in g_old:
                    g_new[k_] = g_old[k_]
        self.optimizer_G, self.flat_optimizer = new, False
        for sch in self.schedulers:
            sch.optimizer = new

    # -- optimizer state in the reference's layout ------------------------------------------------------------------
    def optimizer_state_dict(self) -> dict:
        """`optimizer_G.state_dict()` in the layout the reference saves and resumes from (base_model.py save_training_state /
        resume_training: one entry per parameter, in `optim_params` order) - also when Adam runs on the ONE flat tensor: its
        exp_avg / exp_avg_sq are split along the flat buffer's slices and `step` is repeated per parameter."""
        sd = self.optimizer_G.state_dict()
        if not self.flat_optimizer:
            return sd
        st = sd["state"].get(0)
        state = {}
        if st:
            for i, (p_, v) in enumerate(zip(self.sink.params, self.sink.views)):
                off, n_ = v.storage_offset(), p_.numel()
                state[i] = {k_: (val[off:off + n_].view(p_.shape).clone() if torch.is_tensor(val) and val.dim() == 1 and val.numel() == self.sink.flat.numel()
                                 else (val.clone() if torch.is_tensor(val) else val)) for k_, val in st.items()}
        groups = [dict(g_, params=list(range(len(self.sink.params)))) for g_ in sd["param_groups"]]
        return {"state": state, "param_groups": groups}

    def load_optimizer_state_dict(self, sd: dict):
        """Inverse of `optimizer_state_dict`: accepts a per-parameter Adam state (a reference `.state` file's 'optimizers' entry,
        or one saved here).  With the flat optimizer the per-parameter moments are merged into the flat tensor's; that needs one
        common `step` - if the file's steps differ between parameters the trainer falls back to the per-tensor optimizer.

        The file's hyper-parameter VALUES are adopted (lr, betas, eps, weight_decay, initial_lr); the flags that select the
        step's implementation (capturable, foreach, fused, differentiable) stay the trainer's own, the learning rate of a
        capturable trainer stays the device tensor the captured step reads (the value is written into it), and state tensors
        that already exist are overwritten IN PLACE - a captured step keeps updating the tensors it was recorded on.  A file
        written by torch 1.7 (float lr, integer `step`, no capturable key: what the reference saves) therefore loads into a
        capturable trainer, before or after capture()."""
        if self.flat_optimizer:
            sd = self._merge_state_for_flat(sd)
        self._load_state_keeping_own_groups(sd)

    def _merge_state_for_flat(self, sd: dict) -> dict:
        """The per-parameter state `sd` in the layout of the ONE-tensor Adam (param id 0); falls back to the per-tensor optimizer
        (and returns `sd` unchanged) when the parameters' steps differ or some parameter has no state."""
        n_par = len(self.sink.params)
        ids = [i for g_ in sd["param_groups"] for i in g_["params"]]
        if len(ids) != n_par:
            raise ValueError(f"optimizer state holds {len(ids)} parameters, the net has {n_par}")
        state = sd.get("state", {})
        steps = {float(state[i]["step"]) for i in ids if i in state}
        if state and (len(steps) != 1 or any(i not in state for i in ids)):
            self._per_tensor_optimizer()
            return sd
        out = {"param_groups": [dict(g_, params=[0]) for g_ in sd["param_groups"][:1]], "state": {}}
        if state:
            dev, first = self.sink.flat.device, state[ids[0]]
            merged = {}
            for k_, val in first.items():
                if torch.is_tensor(val) and val.shape == self.sink.params[0].shape and k_ != "step":
                    buf = torch.zeros_like(self.sink.flat)
                    for i, (p_, v) in zip(ids, zip(self.sink.params, self.sink.views)):
                        off = v.storage_offset()
                        buf[off:off + p_.numel()].view(p_.shape).copy_(state[i][k_].to(dev))
                    merged[k_] = buf
                else:
                    merged[k_] = val.clone() if torch.is_tensor(val) else val
            out["state"] = {0: merged}
        return out

    _IMPL_FLAGS = ("capturable", "foreach", "fused", "differentiable")

    def _load_state_keeping_own_groups(self, sd: dict):
        opt = self.optimizer_G
        if len(sd["param_groups"]) != len(opt.param_groups):
            raise ValueError(f"optimizer state has {len(sd['param_groups'])} parameter groups, the trainer's optimizer {len(opt.param_groups)}")
        state = sd.get("state", {})
        for own, saved in zip(opt.param_groups, sd["param_groups"]):
            if len(saved["params"]) != len(own["params"]):
                raise ValueError(f"optimizer state group holds {len(saved['params'])} parameters, the trainer's group {len(own['params'])}")
            for k_, val in saved.items():
                if k_ == "params" or k_ in self._IMPL_FLAGS:
                    continue
                if k_ == "lr" and torch.is_tensor(own.get("lr")):
                    own["lr"].fill_(float(val))                       # the tensor a captured step reads: same object, new value
                else:
                    if self.graph is not None and k_ not in ("lr", "initial_lr") and k_ in own:
                        # betas / eps / weight_decay / amsgrad are constants of the CAPTURED step (kernel arguments of the replayed
                        # launches): adopting other values here would make param_groups say one thing and the replay do another
                        same = (tuple(own[k_]) == tuple(val)) if isinstance(val, (tuple, list)) else (own[k_] == val)
                        if not same:
                            raise RuntimeError(f"optimizer state changes {k_} ({own[k_]!r} -> {val!r}), a constant of the captured step: capture again")
                    own[k_] = float(val) if torch.is_tensor(val) and val.numel() == 1 and k_ in ("lr", "initial_lr") else val
                    if k_ in ("betas", "eps", "weight_decay", "amsgrad"):
                        self._adam_kw[k_] = tuple(val) if isinstance(val, list) else val     # what a rebuilt optimizer (_per_tensor_optimizer) starts from
            for pid, p_ in zip(saved["params"], own["params"]):
                new, cur = state.get(pid), opt.state.get(p_)
                if new is None:
                    if cur:
                        if self.graph is not None:
                            raise RuntimeError("the loaded optimizer state has no entry for a parameter the captured step updates: capture again")
                        del opt.state[p_]
                    continue
                if cur and set(cur) == set(new):
                    for k_, val in new.items():                          # in place: a captured step keeps its tensors
                        if torch.is_tensor(cur[k_]):
                            cur[k_].copy_(torch.as_tensor(val).to(device=cur[k_].device, dtype=cur[k_].dtype))
                        else:
                            cur[k_] = val
                    continue
                if self.graph is not None:
                    raise RuntimeError("the loaded optimizer state does not match the tensors the captured step updates: capture again")
                fresh = {}
                for k_, val in new.items():
                    if k_ == "step":          # torch.optim.Adam._init_group: a device float32 scalar when capturable, a host one otherwise
                        fresh[k_] = torch.tensor(float(val), dtype=torch.float32, device=p_.device if own.get("capturable") or own.get("fused") else "cpu")
                    elif torch.is_tensor(val):
                        fresh[k_] = val.to(device=p_.device, dtype=p_.dtype).clone()
                    else:
                        fresh[k_] = val
                opt.state[p_] = fresh

    def _sync_grads(self):
        """The data-parallel step's one collective: all-reduce (SUM, / world) of the flat gradient buffer."""
        if self.data_parallel:
            self.sink.all_reduce(self.process_group, average=True)

    def _fused_clip_adam(self, max_norm) -> bool:
        """clip_grad_norm_ + Adam.step() of the ONE flat tensor as two launches (selfc_clip_adam) on torch.optim.Adam's own state
        tensors (step / exp_avg / exp_avg_sq, created here as torch would create them), so state_dict / load_state_dict / the LR
        scheduler see a stock optimizer.  False: not applicable (the caller takes the torch path)."""
        opt = self.optimizer_G
        grp = opt.param_groups[0]
        if len(opt.param_groups) != 1 or len(grp["params"]) != 1 or grp.get("amsgrad") or grp.get("maximize") or grp.get("differentiable"):
            return False
        p_ = grp["params"][0]
        if p_.grad is None or not p_.is_cuda or p_.dtype != torch.float32 or not p_.is_contiguous() or not p_.grad.is_contiguous():
            return False
        st = opt.state[p_]
        if len(st) == 0:
            st["step"] = torch.zeros((), dtype=torch.float32, device=p_.device) if grp.get("capturable") else torch.tensor(0.0, dtype=torch.float32)
            st["exp_avg"] = torch.zeros_like(p_, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p_, memory_format=torch.preserve_format)
        step, lr = st["step"], grp["lr"]
        if torch.is_tensor(step) and step.is_cuda:
            step_dev, step_host = step.data_ptr(), 0.0
        else:
            if torch.is_tensor(step):
                step += 1
            else:
                step = st["step"] = step + 1
            step_dev, step_host = None, float(step)
        lr_dev, lr_host = (lr.data_ptr(), 0.0) if (torch.is_tensor(lr) and lr.is_cuda) else (None, float(lr))
        tmp = self.__dict__.get("_adam_tmp")
        if tmp is None or tmp[0].device != p_.device:
            tmp = self._adam_tmp = (torch.empty(int(_lib.lib().selfc_clip_adam_blocks()), dtype=torch.float64, device=p_.device),
                                    torch.zeros((), dtype=torch.float32, device=p_.device))
        b1, b2 = grp["betas"]
        rt.call("selfc_clip_adam", p_.data_ptr(), p_.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p_.numel(),
                tmp[0].data_ptr(), float(max_norm or 0.0), lr_dev, lr_host, float(b1), float(b2), float(grp["eps"]), float(grp["weight_decay"]),
                step_dev, step_host, tmp[1].data_ptr(), _lib.stream_ptr())
        self.grad_norm = tmp[1]
        opt._opt_called = True             # what the LR scheduler's wrapper of optimizer.step() records
        return True

    def _clip_and_step(self):
        """gradient clipping + optimizer step (SelfC_model.py:172-176)."""
        if self.before_clip is not None:
            self.before_clip(self)
        max_norm = self.train_opt.get("gradient_clipping")
        if _FUSED_ADAM and self.flat_optimizer and self.sink is not None and self._fused_clip_adam(max_norm):
            rt.invalidate_weights()
            return
        if max_norm and self.sink is not None:
            # clip_grad_norm_ on the flat buffer (pads are zero): three launches instead of a foreach over 350 views
            self.grad_norm = self.sink.norm()              # staged: no multi-block reduction inside a captured step (GradSink.norm)
            self.sink.flat.mul_(torch.clamp(max_norm / (self.grad_norm + 1e-6), max=1.0))
        elif max_norm:
            self.grad_norm = nn.utils.clip_grad_norm_(self.optim_params, max_norm)
        self.optimizer_G.step()
        if self.flat_optimizer:
            rt.invalidate_weights()            # the update went through the flat tensor: per-parameter version counters did not move

    def _step(self, real_H: torch.Tensor, ref_L: torch.Tensor):
        """optimize_parameters (SelfC_model.py:153-176) up to and including optimizer.step(); returns the loss tensors."""
        losses = self._forward_backward(real_H, ref_L)
        self._sync_grads()
        self._clip_and_step()
        return losses

    def _log(self, losses):
        l_forw_fit, l_back_rec, loss_c, loss = losses
        self.log_dict["l_forw_fit"] = l_forw_fit.item()
        self.log_dict["l_back_rec"] = l_back_rec.item()
        self.log_dict["loss_c"] = float(loss_c)
        self.log_dict["loss"] = loss.item()
        gn = getattr(self, "grad_norm", None)
        if gn is not None:
            self.log_dict["grad_norm"] = float(gn)         # the total norm BEFORE clipping (clip_grad_norm_'s return value)
        return self.log_dict

    def optimize_parameters(self, real_H: torch.Tensor, ref_L: torch.Tensor, step: int = 0):
        if self.graph is not None:
            return self.replay(real_H, ref_L)
        self._zero_grad()
        return self._log(self._step(real_H, ref_L))

    def _zero_grad(self):
        if self.sink is not None:
            self.sink.zero()
        else:
            self.optimizer_G.zero_grad(set_to_none=True)

    # -- whole-step hipGraph ----------------------------------------------------------------------------------------
    def capture(self, real_H: torch.Tensor, ref_L: torch.Tensor, warmup: int = 3):
        """Record one optimisation step (forward, quantise, STP sample, reverse, every gradient kernel on both streams,
        clip, Adam) into a hipGraph; later `optimize_parameters` calls copy the batch into the captured buffers and
        replay it - the step stops being bound by ~1,500 host-side launches.  Shapes are fixed by this call; the
        `warmup` eager steps it runs are real optimisation steps.  Needs `capturable=True` at construction."""
        if not self.capturable:
            raise RuntimeError("construct RescaleTrainer(..., capturable=True) to capture the step")
        self._static_h, self._static_l = real_H.clone(), ref_L.clone()
        s = rt.warmup_stream()
        s.wait_stream(torch.cuda.current_stream())
        warm = []
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self._zero_grad()
                gn_ = getattr(self, "grad_norm", None)
                warm.append(self._step(self._static_h, self._static_l) + (None,))
                gn_ = getattr(self, "grad_norm", None)
                warm[-1] = warm[-1][:-1] + ((gn_.clone() if torch.is_tensor(gn_) else gn_),)     # (the fused step reuses one norm tensor)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        #: the warm-up steps are real optimisation steps: their log entries (read after the synchronize, not between the steps)
        self.warmup_logs = []
        for *losses, gn in warm:
            self.grad_norm = gn
            self.warmup_logs.append(dict(self._log(tuple(losses))))
        if self.sink is None:
            self.optimizer_G.zero_grad(set_to_none=True)
        self.graph, self.graph_tail = None, None      # a re-capture: the old execs die BEFORE the new one is instantiated (rt.new_graph)
        g = rt.new_graph()
        # every capture of the step: own capture stream, own side streams for its duration (runtime.graph_capture)
        if not self.data_parallel:
            with rt.graph_capture(g, self._static_h.device, fresh_side=ag._SIDE):
                if self.sink is not None:
                    self.sink.zero()                 # part of the replayed step
                self._static_losses = self._step(self._static_h, self._static_l)
            self.graph, self.graph_tail = g, None
            return self
        # data parallel: the collective stays OUTSIDE the captured regions (an RCCL call inside a hipGraph is not something
        # this stack promises): graph 1 = zero + forward + backward, eager all-reduce of the flat buffer, graph 2 = clip + Adam
        with rt.graph_capture(g, self._static_h.device, fresh_side=ag._SIDE):
            self.sink.zero()
            self._static_losses = self._forward_backward(self._static_h, self._static_l)
        self._sync_grads()
        g2 = rt.new_graph()
        with rt.graph_capture(g2, self._static_h.device, pool=g.pool()):
            self._clip_and_step()
        self.graph, self.graph_tail = g, g2
        return self

    def replay(self, real_H: torch.Tensor, ref_L: torch.Tensor):
        self._static_h.copy_(real_H)
        self._static_l.copy_(ref_L)
        self.graph.replay()
        if self.graph_tail is not None:
            self._sync_grads()
            self.graph_tail.replay()
        # the replayed Adam step rewrote the parameters without touching torch's version counters: packed-weight caches
        # keyed on (data_ptr, _version) would otherwise serve a later eager eval / validation pass stale weights
        rt.invalidate_weights()
        return self._log(self._static_losses)

    def update_learning_rate(self):
        for s in self.schedulers:
            s.step()

    def get_current_learning_rate(self):
        return self.optimizer_G.param_groups[0]["lr"]


TRAIN_OPT_LARGE = {   # train_rescaling_selfc_large.yml:95-121
    "lr_G": 1e-4, "beta1": 0.9, "beta2": 0.999, "lr_scheme": "MultiStepLR", "lr_steps": [100000, 200000, 300000],
    "lr_gamma": 0.5, "pixel_criterion_forw": "l2", "pixel_criterion_back": "l1", "lambda_cond_prob": 0,
    "lambda_fit_forw": 1, "lambda_rec_back": 1, "weight_decay_G": 1e-14, "gradient_clipping": 10,
}
