"""Training through the HIP path: torch.autograd.Functions of the boundary modules.

The reference trains with stock autograd (SelfC_model.py:153-176); here every differentiable boundary op keeps
what its backward needs (the f16 dense feature buffers the forward kernels leave behind, the coupling's s and
one side of the latent) and calls the gradient entry points of csrc/backward.hip.  Parameter gradients come
back in the reference's own tensor layouts, so optimizers / DDP see ordinary ``.grad`` tensors.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib, runtime as rt
from .packing import dense_channels, pack_planes_generic, pack_pointwise_T, pool_weight_map_grad, pool_weight_map_grad_batch, roundup

_SCRATCH: Dict[Tuple, torch.Tensor] = {}
#: (device, slot) -> event recorded on the side stream behind the LAST weight-gradient phase that read that slot's scratch.  The
#: next data phase on the same slot (any stream) waits for it before it overwrites amax / the gradient planes; cleared when the
#: side streams are joined (_join_side_streams), so no event outlives the backward (or the hipGraph capture) that recorded it.
_SLOT_BUSY: Dict[Tuple, "torch.cuda.Event"] = {}


def _scratch(device, n, h, w, cin, cout, slot="", pair: bool = False) -> torch.Tensor:
    """Scratch of one subnet backward (pair: of a G/H pair's, selfc_gh_bwd_pair).  `slot` separates calls whose weight-gradient
    phase may still be running on the side stream while the next call's data phase starts."""
    need = (_lib.lib().selfc_gh_bwd_pair_scratch_bytes if pair else _lib.lib().selfc_subnet_bwd_scratch_bytes)(n, h, w, cin, cout)
    if need == 0:
        raise RuntimeError("selfc_subnet_bwd_scratch_bytes: invalid shape")
    key = (str(device), slot)
    buf = _SCRATCH.get(key)
    if buf is None or buf.numel() < need:
        buf = _SCRATCH[key] = torch.empty(need, dtype=torch.uint8, device=device)
    return buf


#: SELFC_BWD_PAIR=0: G and H of a coupling block run their backward as two subnet calls on two streams again (round 5) instead of
#: ONE paired call (selfc_gh_bwd_pair: every launch covers both nets, one gradient scale, one input-gradient conv)
_PAIR = os.environ.get("SELFC_BWD_PAIR", "1") != "0"
#: SELFC_BWD_DEFER_FIN=0: every subnet call reduces its weight-gradient partials itself (one finish launch per subnet) instead of
#: leaving them to ONE launch per 24 jobs at the end of the block stack's backward (FinJobs)
_DEFER_FIN = os.environ.get("SELFC_BWD_DEFER_FIN", "1") != "0"
#: SELFC_BWD_DEFER_WG=0: the weight-gradient launches of a block stack run per subnet on the side stream again instead of as ONE launch
#: per kind (conv1..4 / temporal conv5) behind the stack's data-gradient chain.  Inside a replayed graph one stream and three streams
#: take the same time on this runtime (the executor maps the branches onto hardware queues its own way: the "side" work ends up in front
#: of the main chain on the same queue), and every per-subnet launch under-fills the chip - 24 subnets' jobs in one launch do not.
_DEFER_WG = os.environ.get("SELFC_BWD_DEFER_WG", "1") != "0"


#: SELFC_BWD_BG_WG=1 (measured, NOT the default: 5.50 vs 5.31 ms at one septuplet per rank, 13.4 vs 13.5 at eight - one more data point
#: that a forked branch of a replayed graph buys no overlap on this runtime; profiles/r6/ab_experiments.txt r6i): the weight gradients of
#: the FIRST block stack of a backward pass (the reverse direction's) are built thin (one long-lived workgroup per (conv, input plane)
#: pair: ~290 workgroups for a stack) and run on a background stream UNDER the rest of the backward - the STP chain and the second
#: stack's data chain, whose thin launches leave most of every CU idle; one fork and one join instead of the 48 fine-grained ones that
#: bought nothing (this runtime's graph executor, _TWO_STREAMS above).  The second stack's flush joins the background stream first
#: (both finishes accumulate into the same gradient slices: beta = 1), then runs fat on the main stream.
_BG_WG = os.environ.get("SELFC_BWD_BG_WG", "0") == "1"
_BG_ON = False                       # set by the trainer around loss.backward() (background_wgrad)
_BG_PENDING: Dict[str, "torch.cuda.Stream"] = {}


def bg_stream(device):
    key = (str(device), 2)
    if key not in _SIDE:
        _SIDE[key] = rt.distinct_streams(1, device, avoid=[v for k, v in _SIDE.items() if k[0] == str(device)])[0]
    return _SIDE[key]


def join_background(device=None):
    """main stream waits for the background weight gradients (the trainer calls it behind loss.backward(), before the clip)"""
    for key in [k for k in _BG_PENDING if device is None or k == str(device)]:
        torch.cuda.current_stream().wait_stream(_BG_PENDING.pop(key))


class background_wgrad:
    """with background_wgrad(): loss.backward() - lets the first block stack's weight gradients run in the background; the caller
    must call join_background() (the exit does) before it reads any parameter gradient"""

    def __enter__(self):
        global _BG_ON
        self.old, _BG_ON = _BG_ON, _BG_WG
        return self

    def __exit__(self, *exc):
        global _BG_ON
        _BG_ON = self.old
        join_background()


class FinJobs:
    """Deferred weight-gradient finishes of a block stack's backward (selfc_subnet_bwd_phase_d / selfc_gh_bwd_pair leave job
    descriptors here, host memory; flush() reduces all of them with one launch per 24 jobs, in the order they were left)."""

    def __init__(self, capacity: int, defer_wg: bool = False, thin: bool = False):
        self.thin = thin and defer_wg          # jobs built for a background launch (SELFC_BWD_WG_THIN)
        self.size = int(_lib.lib().selfc_fin_job_bytes())
        self.buf = (C.c_ubyte * (self.size * capacity))()
        self.cap, self.n = capacity, 0
        # defer_wg: the weight-gradient LAUNCHES are left as jobs too (selfc_wgrad_run_jobs); one per finish job
        self.defer_wg = defer_wg
        self.wsize = int(_lib.lib().selfc_wg_job_bytes())
        self.wbuf = (C.c_ubyte * (self.wsize * capacity))() if defer_wg else None
        self.wn = 0

    def take_wg(self, k: int):
        if not self.defer_wg:
            return None
        p_ = C.c_void_p(C.addressof(self.wbuf) + self.wn * self.wsize)
        self.wn += k
        return p_

    def take(self, k: int):
        if self.n + k > self.cap:
            raise RuntimeError("FinJobs: capacity exceeded")
        p_ = C.c_void_p(C.addressof(self.buf) + self.n * self.size)
        self.n += k
        return p_

    def flush(self):
        if self.wn:
            rt.call("selfc_wgrad_run_jobs", C.c_void_p(C.addressof(self.wbuf)), self.wn, _lib.stream_ptr())
            self.wn = 0
        if self.n:
            rt.call("selfc_wgrad_finish_jobs", C.c_void_p(C.addressof(self.buf)), self.n, _lib.stream_ptr())
            self.n = 0


#: SELFC_BWD_STREAMS=2: weight gradients (of the subnets outside a paired block stack: the STP chain) run on a second HIP stream and H's
#: backward of an unpaired block on a third, as in rounds 1-5.  Default since round 6: ONE stream.  Inside a replayed graph the executor
#: of this runtime maps branches onto hardware queues its own way (the "side" work lands in front of the main chain on the same queue,
#: every cross-queue dependency costs ~10 us): measured 6.0 against 6.3 ms per step at one septuplet per rank, 7.2 / 7.4 at two,
#: 9.4 / 9.5 at four, equal at eight (profiles/r6/ab_experiments.txt)
_TWO_STREAMS = os.environ.get("SELFC_BWD_STREAMS", "1") == "2"
_SIDE: Dict[Tuple, "torch.cuda.Stream"] = {}


def side_stream(device, which: int = 0):
    """which 0: the weight-gradient stream; 1: the stream that runs H's backward next to G's inside a coupling block."""
    if not _TWO_STREAMS:
        return None
    key = (str(device), which)
    if key not in _SIDE:
        # both side streams at once: different from each other and from torch's capture stream (runtime.distinct_streams)
        a, b = rt.distinct_streams(2, device)
        _SIDE[(str(device), 0)], _SIDE[(str(device), 1)] = a, b
    return _SIDE[key]


# ------------------------------------------------------------------------------------------------------------
# Flat gradient buffer.  Stock autograd hands every parameter gradient to an AccumulateGrad node: a copy or an add
# launch per tensor and call - ~400 launches of 3 us per training step for the 350 tensors of SelfC-large, every block
# being called twice per step (forward and reverse).  The weight-gradient kernels can accumulate into a caller's buffer
# themselves (`beta` of selfc_subnet_bwd), so a trainer may own ONE flat buffer with a view per parameter as its
# `.grad`, zero it once per step and let the subnet backward add into the views; those gradients are then reported to
# autograd as None.  Opt-in (RescaleTrainer does it when the net is not wrapped in DistributedDataParallel, whose
# hooks need the gradients to pass through autograd).
# ------------------------------------------------------------------------------------------------------------
class GradSink:
    """ONE flat fp32 buffer with a view per parameter as its ``.grad`` (see above).

    Data parallelism: ``all_reduce()`` is the step's ONE collective - a SUM over the ranks of the flat buffer
    (3,365,038 floats + pads = 13.5 MB for SelfC-large), divided by the world size, i.e. what DistributedDataParallel's
    bucketed hooks compute (SelfC_model.py:41-44), without its per-tensor hooks: the kernels keep accumulating straight
    into the buffer and the step stays capturable.

    Parameters the backward did not reach in a step (a frozen path, an unused head) are tracked: their views stay zero,
    and ``detach_untouched()`` sets their ``.grad`` to None before the optimizer step, so Adam skips them exactly as it does
    with stock autograd (no moment decay, no weight decay on tensors without a gradient)."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        offs, total = [], 0
        for p_ in self.params:
            offs.append(total)
            total += (p_.numel() + 63) & ~63            # 256-byte aligned slices, pads stay zero
        dev = self.params[0].device
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.views = [self.flat[o:o + p_.numel()].view(p_.shape) for o, p_ in zip(offs, self.params)]
        self.index = {id(p_): v for p_, v in zip(self.params, self.views)}
        self.pos = {id(p_): i for i, p_ in enumerate(self.params)}
        self.touched = set()
        self.flat_param: Optional[torch.nn.Parameter] = None
        # gradients that still arrive through autograd (AccumulateGrad adds them into the view in place) count as touched
        self._hooks = [p_.register_post_accumulate_grad_hook(self._mark) for p_ in self.params]
        self.attach()

    def _mark(self, p_):
        self.touched.add(self.pos[id(p_)])

    def attach(self):
        for p_, v in zip(self.params, self.views):
            if p_.grad is not v:
                p_.grad = v
        if self.flat_param is not None and self.flat_param.grad is not self.flat:
            self.flat_param.grad = self.flat

    def flatten_params(self) -> torch.nn.Parameter:
        """Make every parameter a VIEW of one flat fp32 buffer laid out exactly like the gradient buffer (same 64-float aligned
        slices, zero pads) and return that buffer as a single Parameter whose `.grad` is the flat gradient buffer.  An optimizer
        given this one tensor updates all 350 parameters of SelfC-large with a dozen element-wise launches instead of a dozen
        `_foreach` ops of 16 multi-tensor launches each (2.8 ms -> 0.1 ms of the training step); Adam is element-wise, so the
        update is the same.  The Parameter objects, their names, shapes and `state_dict()` are unchanged (load_state_dict copies
        into the views); what changes is that the optimizer writes them through the flat tensor, so torch's per-parameter version
        counters no longer move: callers invalidate the packed-weight caches after a step (runtime.invalidate_weights)."""
        if self.flat_param is None:
            buf = torch.zeros_like(self.flat)
            with torch.no_grad():
                for p_, v in zip(self.params, self.views):
                    off = v.storage_offset()
                    w = buf[off:off + p_.numel()].view(p_.shape)
                    w.copy_(p_.data)
                    p_.data = w
            self.flat_param = torch.nn.Parameter(buf, requires_grad=True)
            self.flat_param.grad = self.flat
        return self.flat_param

    def params_attached(self) -> bool:
        """With flatten_params(): do the parameters still alias the flat buffer?  (netG.to() / .half() after flattening
        re-points every `.data` at fresh storage and silently detaches the optimizer's tensor from the net.)"""
        if self.flat_param is None:
            return True
        base = self.flat_param.data_ptr()
        return all(p_.data_ptr() == base + 4 * v.storage_offset() for p_, v in ((self.params[0], self.views[0]), (self.params[-1], self.views[-1])))

    def assert_touched_equal(self, group=None):
        """Data parallel: the set of parameters that received a gradient must be the same on every rank (it selects the flat or
        the per-tensor optimizer and what Adam skips).  One MIN / MAX all-reduce of the bitmap; raises on a mismatch."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) < 2:
            return
        bits = torch.zeros(len(self.params), dtype=torch.int32, device=self.flat.device)
        if self.touched:
            bits[torch.tensor(sorted(self.touched), device=self.flat.device)] = 1
        lo, hi = bits.clone(), bits.clone()
        device_collective(lo, lambda t_: dist.all_reduce(t_, op=dist.ReduceOp.MIN, group=group), group)
        device_collective(hi, lambda t_: dist.all_reduce(t_, op=dist.ReduceOp.MAX, group=group), group)
        if not torch.equal(lo, hi):
            bad = torch.nonzero(lo != hi).flatten().tolist()
            raise RuntimeError(f"data-parallel ranks disagree on which parameters received a gradient (parameter indices {bad[:8]}...): "
                               "they would run different optimizers")

    def zero(self):
        """Once per step, instead of optimizer.zero_grad(): one memset; re-attaches views somebody replaced (or that
        detach_untouched() took away) and forgets which views were written."""
        self.flat.zero_()
        self.touched.clear()
        self.attach()

    def norm(self) -> torch.Tensor:
        """L2 norm of the whole buffer (gradient clipping) as THREE staged reductions - rows of 256, rows of 64, the rest - instead
        of one `vector_norm(flat)`.  Why: 3.4 M elements into ONE output is torch's multi-block "global reduce" (per-block partials
        plus a semaphore that a memset in front of the kernel zeroes); inside a captured step such a reduction was seen to leave its
        output unwritten on this stack (the loss scalars of a one-stream capture at 8 septuplets: train.ReconstructionLoss,
        tools/experiments/loss_alias_probe.py).  Every stage here has either thousands of outputs or a few hundred inputs, i.e. no
        cross-block stage; the value is the same norm (norm of row norms), deterministic, a handful of small launches."""
        n = self.flat.numel()
        main = n - n % (256 * 64)
        parts = []
        if main:
            r1 = torch.linalg.vector_norm(self.flat[:main].view(-1, 256), dim=1)
            parts.append(torch.linalg.vector_norm(r1.view(-1, 64), dim=1))
        if n > main:
            parts.append(torch.linalg.vector_norm(self.flat[main:].view(-1, 64), dim=1))     # < 16,384 elements (slices are 64-float aligned)
        return torch.linalg.vector_norm(torch.cat(parts) if len(parts) > 1 else parts[0])

    def view_of(self, p_) -> Optional[torch.Tensor]:
        """The view a kernel may accumulate into (marks the parameter as having received a gradient this step)."""
        v = self.index.get(id(p_))
        if v is not None and p_.grad is v:
            self.touched.add(self.pos[id(p_)])
            return v
        return None

    def untouched(self) -> List[torch.Tensor]:
        return [p_ for i, p_ in enumerate(self.params) if i not in self.touched]

    def detach_untouched(self) -> int:
        """After backward, before clip / optimizer.step(): parameters without a gradient this step get ``.grad = None``
        (their slice of the flat buffer is zero, so norms and the all-reduce are unaffected).  Returns how many."""
        n = 0
        for i, p_ in enumerate(self.params):
            if i not in self.touched and p_.grad is not None:
                p_.grad = None
                n += 1
        return n

    def all_reduce(self, group=None, average: bool = True) -> int:
        """The data-parallel step's single collective: SUM of the flat buffer over the ranks of `group` (RCCL on GPUs, gloo
        on CPU), then / world.  Call between backward and clip.  Returns the world size (1: nothing was sent)."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return 1
        world = dist.get_world_size(group)
        if world > 1:
            device_collective(self.flat, lambda t_: dist.all_reduce(t_, op=dist.ReduceOp.SUM, group=group), group)
            if average:
                self.flat.mul_(1.0 / world)
        return world


def device_collective(tensor: torch.Tensor, fn, group=None):
    """Run the in-place collective `fn(tensor)` (all_reduce / broadcast) on a tensor that may live on the GPU while the
    process group is gloo: RCCL ("nccl") takes device tensors as they are; gloo - the CPU rehearsal backend, and the one that
    lets several ranks share ONE GPU (tests/test_gpu_multi.py) - goes through a host copy.  The round trip synchronises with
    the host, which is fine where it is used: the data-parallel step keeps its collective outside the captured graphs."""
    import torch.distributed as dist
    if tensor.is_cuda and dist.get_backend(group) == "gloo":
        host = tensor.detach().to("cpu")
        fn(host)
        tensor.detach().copy_(host)
    else:
        fn(tensor)


_SINK: Optional[GradSink] = None


class grad_sink:
    """Context manager: parameter gradients of the subnet backward go into `sink` while it is active."""

    def __init__(self, sink: Optional[GradSink]):
        self.sink, self.prev = sink, None

    def __enter__(self):
        global _SINK
        self.prev, _SINK = _SINK, self.sink
        return self.sink

    def __exit__(self, *exc):
        global _SINK
        _SINK = self.prev
        return False


def subnet_params(mod) -> List[torch.Tensor]:
    """conv1.weight, conv1.bias, ..., conv5.weight, conv5.bias (the order the Functions take and return)."""
    cached = mod.__dict__.get("_conv_plist")
    if cached is not None and cached[0] is mod.conv1.weight:
        return cached
    out = mod.__dict__["_conv_plist"] = []
    for i in range(1, 6):
        conv = getattr(mod, f"conv{i}")
        if conv.bias is None:
            raise NotImplementedError("subnet backward expects bias=True convs (every shipped config)")
        out += [conv.weight, conv.bias]
    return out


def subnet_bwd(mod, dense: torch.Tensor, xin: Optional[torch.Tensor], dout: torch.Tensor, sign: float,
               dx: Optional[torch.Tensor], accumulate_dx: bool, n: int, t: int, h: int, w: int,
               want_params: bool = True, pk=None, side=None, slot: str = "", on_data_done=None,
               dout_amax: Optional[torch.Tensor] = None, dx_amax_out: Optional[torch.Tensor] = None,
               fin: Optional["FinJobs"] = None) -> List[Optional[torch.Tensor]]:
    """Backward of one DenseBlock / D2DTInput on kernel-layout buffers; returns the 10 parameter gradients
    (reference layouts) or Nones.  With `side` (a stream) the weight-gradient phase is enqueued there, ordered after the
    data phase; the caller joins the streams before it hands the gradients on and must not reuse `slot` before that.
    dout_amax: one-float tensor holding max|dout|, taken by whoever produced dout (selfc_coupling_bwd_x, selfc_add_absmax, another
    call's dx_amax_out) - the call then skips its own pass over dout; dx_amax_out: zeroed one-float tensor that receives max|dx|."""
    cin, cout = mod.channel_in, mod.channel_out
    pk = pk if pk is not None else mod.packed()             # inside an InvBlockExp the block's plan owns the tensors
    dev = dout.device
    grads: List[Optional[torch.Tensor]] = [None] * 10
    wg = (C.c_void_p * 5)()
    bg = (C.c_void_p * 5)()
    beta = 0.0
    sunk = None
    if want_params and _SINK is not None:
        sunk = [_SINK.view_of(p_) for p_ in subnet_params(mod)]
        if any(v is None for v in sunk):
            sunk = None
    if sunk is not None:
        # the kernels add into the trainer's flat buffer (beta = 1); autograd sees no gradient for these tensors
        for i, v in enumerate(sunk):
            (wg if i % 2 == 0 else bg)[i // 2] = v.data_ptr()
        beta = 1.0
    elif want_params:
        # one allocation for the ten gradients (each slice 64-float aligned), handed out as views
        prm = subnet_params(mod)
        offs, total = [], 0
        for p_ in prm:
            offs.append(total)
            total += (p_.numel() + 63) & ~63
        flat = torch.empty(total, dtype=torch.float32, device=dev)
        base = flat.data_ptr()
        for i, p_ in enumerate(prm):
            grads[i] = flat[offs[i]:offs[i] + p_.numel()].view(p_.shape)
            (wg if i % 2 == 0 else bg)[i // 2] = base + 4 * offs[i]
    scratch = _scratch(dev, n, h, w, cin, cout, slot)
    busy = _SLOT_BUSY.pop((str(dev), slot), None)
    if busy is not None:
        # an earlier call's weight-gradient phase (side stream) may still be reading this slot's scratch: order this call's
        # data phase, which overwrites it, behind that phase
        torch.cuda.current_stream().wait_event(busy)
    bw = pk.bwd_struct()
    args = (bw, mod.kind, dense.data_ptr(), None if xin is None else xin.data_ptr(), dout.data_ptr(),
            float(sign), None if dx is None else dx.data_ptr(), 1 if accumulate_dx else 0,
            wg if want_params else None, bg if want_params else None, beta,
            scratch.data_ptr(), scratch.numel(), n, t, h, w, cin, cout,
            None if dout_amax is None else dout_amax.data_ptr(), None if dx_amax_out is None else dx_amax_out.data_ptr())
    jobs = fin.take(2) if (fin is not None and want_params) else None      # deferred finish: the caller flushes (FinJobs)
    wjobs = fin.take_wg(2) if (jobs is not None and mod.kind == rt.SUBNET_D2DT) else None
    if wjobs is not None:
        # the weight-gradient launches are deferred as well: nothing of this call runs beside the data chain
        rt.call("selfc_subnet_bwd_phase_d", 3 | (8 if fin.thin else 0), *args, jobs, wjobs, _lib.stream_ptr())
        if on_data_done is not None:
            on_data_done()
        return grads
    if (side is None or not want_params) and on_data_done is None:
        rt.call("selfc_subnet_bwd_phase_d", 3, *args, jobs, None, _lib.stream_ptr())
        return grads
    rt.call("selfc_subnet_bwd_phase_d", 1, *args, None, None, _lib.stream_ptr())
    if on_data_done is not None:
        on_data_done()
    if not want_params:
        return grads
    if side is None:
        rt.call("selfc_subnet_bwd_phase_d", 2, *args, jobs, None, _lib.stream_ptr())
        return grads
    side.wait_event(torch.cuda.current_stream().record_event())
    with torch.cuda.stream(side):
        rt.call("selfc_subnet_bwd_phase_d", 2, *args, jobs, None, _lib.stream_ptr())
        _SLOT_BUSY[(str(dev), slot)] = side.record_event()
    return grads


def _grad_targets(mod, want_params: bool, dev):
    """(grads list for autograd, wg array, bg array, beta) of one subnet: views of the trainer's flat buffer (beta = 1, autograd gets
    None) or one fresh allocation handed out as views."""
    grads: List[Optional[torch.Tensor]] = [None] * 10
    wg = (C.c_void_p * 5)()
    bg = (C.c_void_p * 5)()
    if not want_params:
        return grads, None, None, 0.0
    prm = subnet_params(mod)
    if _SINK is not None:
        sunk = [_SINK.view_of(p_) for p_ in prm]
        if all(v is not None for v in sunk):
            for i, v in enumerate(sunk):
                (wg if i % 2 == 0 else bg)[i // 2] = v.data_ptr()
            return grads, wg, bg, 1.0
    offs, total = [], 0
    for p_ in prm:
        offs.append(total)
        total += (p_.numel() + 63) & ~63
    flat = torch.empty(total, dtype=torch.float32, device=dev)
    base = flat.data_ptr()
    for i, p_ in enumerate(prm):
        grads[i] = flat[offs[i]:offs[i] + p_.numel()].view(p_.shape)
        (wg if i % 2 == 0 else bg)[i // 2] = base + 4 * offs[i]
    return grads, wg, bg, 0.0


def gh_pair_bwd(blk, pb, gd: torch.Tensor, hd: torch.Tensor, xin: torch.Tensor, dout_g: torch.Tensor, dout_h: torch.Tensor,
                sign_g: float, sign_h: float, dx: torch.Tensor, n: int, t: int, h: int, w: int, want_params: bool, side=None,
                slot: str = "", amax_g: Optional[torch.Tensor] = None, amax_h: Optional[torch.Tensor] = None,
                dx_amax_out: Optional[torch.Tensor] = None, fin: Optional["FinJobs"] = None):
    """Backward of G and H of one InvBlockExp as ONE call (selfc_gh_bwd_pair): both read `xin`, dx += both input gradients.
    Returns (gG, gH): the two lists of 10 parameter gradients (or Nones).  `side`: stream of the weight-gradient phase."""
    G, H = blk.G, blk.H
    cin, cout = G.channel_in, G.channel_out
    dev = dout_g.device
    gG, wgG, bgG, betaG = _grad_targets(G, want_params, dev)
    gH, wgH, bgH, betaH = _grad_targets(H, want_params, dev)
    if want_params and betaG != betaH:
        raise RuntimeError("gh_pair_bwd: G and H must both (or neither) accumulate into the flat gradient buffer")
    scratch = _scratch(dev, n, h, w, cin, cout, slot, pair=True)
    busy = _SLOT_BUSY.pop((str(dev), slot), None)
    if busy is not None:
        torch.cuda.current_stream().wait_event(busy)
    ptr = lambda t_: None if t_ is None else t_.data_ptr()      # noqa: E731
    args = (pb.G.bwd_struct(), pb.H.bwd_struct(), gd.data_ptr(), hd.data_ptr(), xin.data_ptr(), dout_g.data_ptr(), dout_h.data_ptr(),
            float(sign_g), float(sign_h), dx.data_ptr(), 1, wgG, bgG, wgH, bgH, betaG,
            scratch.data_ptr(), scratch.numel(), n, t, h, w, cin, cout, ptr(amax_g), ptr(amax_h), ptr(dx_amax_out))
    jobs = fin.take(4) if (fin is not None and want_params) else None
    wjobs = fin.take_wg(4) if jobs is not None else None
    if side is None or not want_params or wjobs is not None:
        rt.call("selfc_gh_bwd_pair", (3 if want_params else 1) | (8 if (fin is not None and fin.thin) else 0), *args, jobs, wjobs, _lib.stream_ptr())
        return gG, gH
    rt.call("selfc_gh_bwd_pair", 1, *args, None, None, _lib.stream_ptr())
    side.wait_event(torch.cuda.current_stream().record_event())
    with torch.cuda.stream(side):
        rt.call("selfc_gh_bwd_pair", 2, *args, jobs, None, _lib.stream_ptr())
        _SLOT_BUSY[(str(dev), slot)] = side.record_event()
    return gG, gH


def _dense_buffer(cin: int, n: int, h: int, w: int, dev) -> torch.Tensor:
    """Fresh dense feature buffer of a stand-alone subnet call.  Features are fully written by the conv epilogues; a zero
    fill is only needed for the pad channels of the input planes (cin > 3 and not a multiple of 32)."""
    mk = torch.zeros if (cin > 3 and cin % 32) else torch.empty
    return mk((dense_channels(cin) // 32, n, h, w, 32), dtype=_lib.operand_dtype(), device=dev)


class SubnetFn(torch.autograd.Function):
    """DenseBlock.forward / D2DTInput.forward (Subnet_constructor.py:26-34,119-133) with a HIP backward."""

    @staticmethod
    def forward(ctx, x, mod, t, *params):
        x = rt.as_input(x)
        n, cin, h, w = x.shape
        pk = mod.packed()
        dev, sp = x.device, _lib.stream_ptr()
        cinp, coutp = roundup(cin, 4), roundup(mod.channel_out, 4)
        xin = torch.empty((n, h, w, cinp), dtype=torch.float32, device=dev)
        rt.call("selfc_nchw_to_nhwc4", x.data_ptr(), xin.data_ptr(), n, cin, h, w, sp)
        dense = _dense_buffer(cin, n, h, w, dev)
        yout = torch.empty((n, h, w, coutp), dtype=torch.float32, device=dev)
        sw = pk.struct()
        rt.call("selfc_subnet_run", sw, mod.kind, xin.data_ptr(), yout.data_ptr(), dense.data_ptr(),
                n, t, h, w, cin, mod.channel_out, sp)
        y = torch.empty((n, mod.channel_out, h, w), dtype=torch.float32, device=dev)
        rt.call("selfc_nhwc4_to_nchw", yout.data_ptr(), y.data_ptr(), n, mod.channel_out, h, w, sp)
        ctx.mod, ctx.t, ctx.shape = mod, t, (n, cin, h, w)
        ctx.dense = dense
        ctx.xin = xin if cin <= 3 else None
        return y

    @staticmethod
    def backward(ctx, gy):
        mod, t = ctx.mod, ctx.t
        n, cin, h, w = ctx.shape
        gy = gy.contiguous().float()
        dev, sp = gy.device, _lib.stream_ptr()
        cout = mod.channel_out
        dout = torch.empty((n, h, w, roundup(cout, 4)), dtype=torch.float32, device=dev)
        rt.call("selfc_nchw_to_nhwc4", gy.data_ptr(), dout.data_ptr(), n, cout, h, w, sp)
        dxl = torch.empty((n, h, w, roundup(cin, 4)), dtype=torch.float32, device=dev)
        grads = subnet_bwd(mod, ctx.dense, ctx.xin, dout, 1.0, dxl, False, n, t, h, w,
                           want_params=any(ctx.needs_input_grad[3:]))
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((n, cin, h, w), dtype=torch.float32, device=dev)
            rt.call("selfc_nhwc4_to_nchw", dxl.data_ptr(), dx.data_ptr(), n, cin, h, w, sp)
        return (dx, None, None, *grads)


# ------------------------------------------------------------------------------------------------------------
# FeatureCalapseBlock (Subnet_constructor.py:280-324): the dense block at 1/scale resolution with (3,3,3) conv1 / conv5
# ------------------------------------------------------------------------------------------------------------

def _fcb_seg(mod, j: int) -> Tuple[int, int, int]:
    """(first plane, planes, channels) of input segment j of the block's plane buffer: j = 0 the cin inputs, j = 1..4 feature j"""
    pin, gp = roundup(mod.cin, 32) // 32, mod.gc // 32
    return (0, pin, mod.cin) if j == 0 else (pin + (j - 1) * gp, gp, mod.gc)


def _fcb_bwd_pack(mod):
    """{(k, j): generic plane-conv fragments of conv k's TRANSPOSE restricted to input segment j} for k = 1..5, j < k:
    out channels = the segment's channels (zero padded to whole planes), in channels = conv k's outputs, temporal and
    spatial taps reversed (y = sum W[o][c][t] x[c][. + t]  =>  dx[c] = sum W[o][c][T-1-t'] dy[o][. + t'])."""
    key = rt.params_key(mod)
    if getattr(mod, "_bwd_key", None) != key:
        pk = {}
        for k in range(1, 6):
            wt = getattr(mod, f"conv{k}").weight.detach().float()                 # (O, C, kt, 3, 3)
            o, _, kt = wt.shape[0], wt.shape[1], wt.shape[2]
            wr = wt.reshape(o, wt.shape[1], kt, 9).flip(2).flip(3)
            for j in range(k):
                _, npl, ch = _fcb_seg(mod, j)
                c0 = 0 if j == 0 else mod.cin + (j - 1) * mod.gc
                t = torch.zeros(npl * 32, roundup(o, 32), kt, 9, dtype=torch.float32, device=wt.device)
                t[:ch, :o] = wr[:, c0:c0 + ch].permute(1, 0, 2, 3)
                pk[(k, j)] = pack_planes_generic(t)
        mod._bwd_pk, mod._bwd_key = pk, key
    return mod._bwd_pk


def fcb_bwd(mod, dense: torch.Tensor, dout: torch.Tensor, n: int, t: int, h: int, w: int, want_dx: bool, want_params: bool):
    """Backward of the five convs of a FeatureCalapseBlock on its saved plane buffer `dense` ([x planes | f1..f4], f16) given
    dout fp32 NHWC (n,h,w,cout).  The dense recursion of DESIGN.md section 4b, segment by segment, on the generic plane conv
    (selfc_bwd_conv_planes, temporal taps for conv1 / conv5); weight gradients per temporal tap on selfc_bwd_wgrad with the
    activation planes shifted by one frame inside each clip.  Returns (dx fp32 NHWC (n,h,w,cin) | None, [dW1, db1, .., dW5, db5] | Nones)."""
    dev, sp, L = dout.device, _lib.stream_ptr(), _lib.lib()
    F16 = _lib.operand_dtype()
    npix = n * h * w
    pin, gp, cin, cout = roundup(mod.cin, 32) // 32, mod.gc // 32, mod.cin, mod.cout
    pk = _fcb_bwd_pack(mod)
    amax = torch.zeros(64, dtype=torch.float32, device=dev)
    rt.call("selfc_bwd_scale", dout.data_ptr(), dout.numel(), amax.data_ptr(), sp)
    g = {5: torch.empty((cout // 32, npix, 32), dtype=F16, device=dev)}          # gradient planes of each conv's output
    rt.call("selfc_bwd_to_planes", dout.data_ptr(), g[5].data_ptr(), npix, cout, cout, 0, 1.0, amax.data_ptr(), sp)

    def conv_t(k, j, out_planes=None, add=None, mask=None, plain=None, accumulate=0):
        kt = 3 if k in (1, 5) else 1
        _, npl, _ = _fcb_seg(mod, j)
        rt.call("selfc_bwd_conv_planes", g[k].data_ptr(), g[k].shape[0], kt, 0, pk[(k, j)].data_ptr(), npl,
                None if out_planes is None else out_planes.data_ptr(), None if add is None else add.data_ptr(),
                None if mask is None else mask.data_ptr(), -2 if mask is not None else -1,
                None if plain is None else plain.data_ptr(), cin if plain is not None else 0, accumulate,
                amax.data_ptr(), n, t, h, w, sp)

    # dpre_j = LeakyReLU'(f_j) * sum_{k > j} conv_k^T(dpre_k)[f_j], j = 4..1 (dpre_5 = dout); the sum is chained through `add`
    for j in range(4, 0, -1):
        p0, npl, _ = _fcb_seg(mod, j)
        feat = dense[p0:p0 + npl]
        bufs = [torch.empty((npl, npix, 32), dtype=F16, device=dev) for _ in range(2)]
        ks = list(range(5, j, -1))
        for i, k in enumerate(ks):
            last = i == len(ks) - 1
            conv_t(k, j, out_planes=bufs[i & 1], add=bufs[(i - 1) & 1] if i else None, mask=feat if last else None)
        g[j] = bufs[(len(ks) - 1) & 1]
    dx = None
    if want_dx:
        dx = torch.empty((n, h, w, cin), dtype=torch.float32, device=dev)
        for i, k in enumerate(range(5, 0, -1)):
            conv_t(k, 0, plain=dx, accumulate=1 if i else 0)
    grads = [None] * 10
    if want_params:
        b = n // t
        for k in range(1, 6):
            conv = getattr(mod, f"conv{k}")
            o, kt = conv.out_channels, conv.weight.shape[2]
            qn = pin + (k - 1) * gp
            q = dense[:qn]
            gw = torch.empty((o, qn * 32, kt, 9), dtype=torch.float32, device=dev)
            gb = torch.empty((o,), dtype=torch.float32, device=dev)
            need = L.selfc_bwd_wgrad_scratch_bytes(n, h, w, g[k].shape[0], qn, 9)
            sc = _buf(_STP_CACHE, "fcb_wgrad", need, dev)
            for dt in range(kt):
                sh = dt - (kt // 2)                              # input frame = output frame + sh
                if sh == 0:
                    qs = q
                else:                                            # shift inside every clip, zero outside (the conv's zero padding)
                    q5 = q.reshape(qn, b, t, h * w * 32)
                    qs = torch.zeros_like(q5)
                    if sh > 0:
                        qs[:, :, : t - sh] = q5[:, :, sh:]
                    else:
                        qs[:, :, -sh:] = q5[:, :, : t + sh]
                one = torch.empty((o, qn * 32, 9), dtype=torch.float32, device=dev)
                rt.call("selfc_bwd_wgrad", g[k].data_ptr(), g[k].shape[0], qs.data_ptr(), qn, 9, one.data_ptr(), o, qn * 32,
                        gb.data_ptr() if sh == 0 else None, 0.0, amax.data_ptr(), sc.data_ptr(), sc.numel(), n, t, h, w, sp)
                gw[:, :, dt] = one
            real = torch.cat((gw[:, :cin], gw[:, pin * 32:]), dim=1) if pin * 32 != cin else gw      # drop the input planes' pad channels
            grads[2 * (k - 1)] = real.reshape(o, real.shape[1], kt, 3, 3).contiguous()
            grads[2 * (k - 1) + 1] = gb
    return dx, grads


class FCBFn(torch.autograd.Function):
    """The dense block of FeatureCalapseBlock.forward (Subnet_constructor.py:314-318) on space-to-depth'd frames
    (n, scale^2 cin, h, w) -> (n, scale^2 cout, h, w); the two shuffles around it stay torch view ops."""

    @staticmethod
    def forward(ctx, xs, mod, t, *params):
        xs = rt.as_input(xs)
        n, c, h, w = xs.shape
        y, dense = mod._run_planes(xs, t)
        ctx.mod, ctx.t, ctx.shape, ctx.dense = mod, t, (n, c, h, w), dense
        return y

    @staticmethod
    def backward(ctx, gy):
        mod, t = ctx.mod, ctx.t
        n, c, h, w = ctx.shape
        gy = gy.contiguous().float()
        dev, sp = gy.device, _lib.stream_ptr()
        dout = torch.empty((n, h, w, mod.cout), dtype=torch.float32, device=dev)
        rt.call("selfc_nchw_to_nhwc4", gy.data_ptr(), dout.data_ptr(), n, mod.cout, h, w, sp)
        dxl, grads = fcb_bwd(mod, ctx.dense, dout, n, t, h, w, ctx.needs_input_grad[0], any(ctx.needs_input_grad[3:]))
        dx = None
        if dxl is not None:
            dx = torch.empty((n, c, h, w), dtype=torch.float32, device=dev)
            rt.call("selfc_nhwc4_to_nchw", dxl.data_ptr(), dx.data_ptr(), n, c, h, w, sp)
        return (dx, None, None, *grads)


class InvBlockFn(torch.autograd.Function):
    """InvBlockExp.forward(x, rev) (Inv_arch.py:21-33) with a HIP backward.

    Saved per call: the three dense feature buffers, s, and the side of the latent the gradient formulas need
    (forward: x2 in and y1 out; reverse: x1 in and y2 out)."""

    @staticmethod
    def forward(ctx, x, blk, rev, t, *params):
        x = rt.as_input(x)
        n, c, h, w = x.shape
        c1, c2 = blk.split_len1, blk.split_len2
        ws = rt.Workspace(x.device, blk.F.kind, n, t, h, w, c1, c2, single_use=True)      # private: kept for backward
        pb = rt.packed_block(blk)
        rt.nchw_to_latent(x, ws)
        keep = (ws.x1 if rev else ws.x2).clone()                          # the input side the kernels overwrite
        bw, lat = pb.struct(), ws.latent(want_s=True)
        rt.call("selfc_invblock_run", bw, lat, 1 if rev else 0, _lib.stream_ptr())
        blk._set_s_lazy(ws)
        ctx.blk, ctx.rev, ctx.t, ctx.ws, ctx.keep = blk, bool(rev), t, ws, keep
        return rt.latent_to_nchw(ws)

    @staticmethod
    def backward(ctx, gy):
        blk, rev, t, ws, keep = ctx.blk, ctx.rev, ctx.t, ctx.ws, ctx.keep
        n, h, w, c1, c2 = ws.N, ws.H, ws.W, ws.c1, ws.c2
        gy = gy.contiguous().float()
        dev, sp = gy.device, _lib.stream_ptr()
        want = any(ctx.needs_input_grad[4:])
        d1 = torch.empty((n, h, w, 4), dtype=torch.float32, device=dev)
        d2 = torch.empty((n, h, w, ws.c2p), dtype=torch.float32, device=dev)
        rt.call("selfc_nchw_to_latent", gy.data_ptr(), d1.data_ptr(), d2.data_ptr(), None, ws.FC, n, c1, c2, h, w, sp)
        d1, dx2, gF, gG, gH, _ = _block_backward(blk, ws, keep, rev, t, d1, d2, want, restore_fd=True)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((n, c1 + c2, h, w), dtype=torch.float32, device=dev)
            rt.call("selfc_latent_to_nchw", d1.data_ptr(), dx2.data_ptr(), dx.data_ptr(), n, c1, c2, h, w, sp)
        _join_side_streams(dev, want)
        return (dx, None, None, None, *gF, *gG, *gH)


def _join_side_streams(dev, want: bool):
    main = torch.cuda.current_stream()
    if want and side_stream(dev) is not None:
        main.wait_stream(side_stream(dev))                         # the parameter gradients are complete from here on
        for key in [k for k in _SLOT_BUSY if k[0] == str(dev)]:    # every slot's last reader is behind that join
            del _SLOT_BUSY[key]
    if side_stream(dev, 1) is not None:
        main.wait_stream(side_stream(dev, 1))


#: SELFC_BWD_FOLD_AMAX=0: every subnet backward takes max|dOut| with its own pass again (A/B, and the reference for the
#: bit-equality test of the folded path)
_FOLD_AMAX = os.environ.get("SELFC_BWD_FOLD_AMAX", "1") != "0"


def _block_backward(blk, ws, keep, rev, t, d1, d2, want, restore_fd, tag="", amax_in=None, amax_slots=None, fin=None):
    """Gradient of one InvBlockExp call on the latent layout.  d1 / d2: gradients w.r.t. the block's outputs (y1, y2) as fp32
    [n][h][w][4] / [n][h][w][c2p] (d1 is updated in place); ws: what the forward left - fd / gd / hd (dense features), s, and
    the OUTPUT side the formulas need (forward: ws.x1 = y1; reverse: ws.x2 = y2); keep: the INPUT side the kernels overwrote
    (forward: x2, reverse: x1).  Returns (d1, dx2, gF, gG, gH, amax_out): gradients w.r.t. the inputs (x1, x2), the parameters,
    and (folded path) the one-float max of the gradient the NEXT block's first subnet backward scales by.
    The weight-gradient phases run on the side stream and H's chain on a third one; the caller joins them
    (_join_side_streams) before the gradients are used.  `tag` picks the scratch set: a caller that walks several blocks
    gives every block its own, so that block i's weight-gradient phases (side stream) and the data phases of the blocks
    behind it (main stream) never share a buffer and no ordering between the two streams is needed before the join
    (~70 MB per subnet at 8 x 7 x 36 x 36; a set that IS reused before a join waits for its last reader: subnet_bwd).

    max|dOut| (every subnet backward scales its f16 gradient operands by a power of two taken from it) is taken where dOut is
    PRODUCED - in the coupling gradient kernel (dh; reverse: also dx2), in the kernel that adds the two halves of y1's gradient
    (F's dOut) and in F's dx epilogue (forward: the next block's d2) - instead of a pass over dOut per call (61 launches + 61
    memsets per training step on the critical path).  amax_in: the max of this block's first dOut (forward: d2, reverse: d1)
    from the previous block's call, or None (first block: that subnet takes it itself); amax_slots: three zeroed floats."""
    n, h, w, c2 = ws.N, ws.H, ws.W, ws.c2
    dev, sp = d1.device, _lib.stream_ptr()
    dx2 = torch.empty_like(d2)
    dh = torch.empty_like(d2)
    nel = d2.numel()
    clamp = float(blk.clamp)
    pb = rt.packed_block(blk)
    side = side_stream(dev) if want else None
    side_h = side_stream(dev, 1)
    main = torch.cuda.current_stream()
    ev_h = []
    fold = amax_slots is not None
    A = [amax_slots[i:i + 1] for i in range(3)] if fold else [None, None, None]
    ptr = lambda t_: None if t_ is None else t_.data_ptr()      # noqa: E731

    if _PAIR and blk.G.kind == rt.SUBNET_D2DT and blk.G.channel_in <= 3:
        # G and H as ONE call: no third stream, no d1 += d1h; the pair's dx conv is the last writer of d1 and leaves its max
        if not rev:
            rt.call("selfc_coupling_bwd_x", 0, keep.data_ptr(), ws.s.data_ptr(), d2.data_ptr(), dx2.data_ptr(), dh.data_ptr(), clamp, nel,
                    None, ptr(A[0]), sp)
            gG, gH = gh_pair_bwd(blk, pb, ws.gd, ws.hd, ws.x1, d2, dh, 1.0, 1.0, d1, n, t, h, w, want, side, "GH" + tag,
                                 amax_g=amax_in, amax_h=A[0], dx_amax_out=A[1], fin=fin)
            if restore_fd:
                rt.call("selfc_nhwc_to_planes", keep.data_ptr(), ws.fd.data_ptr(), n * h * w, c2, sp)
            gF = subnet_bwd(blk.F, ws.fd, None, d1, 1.0, dx2, True, n, t, h, w, want, pb.F, side, "F" + tag, dout_amax=A[1], dx_amax_out=A[2], fin=fin)
            return d1, dx2, gF, gG, gH, A[2]
        gF = subnet_bwd(blk.F, ws.fd, None, d1, -1.0, d2, True, n, t, h, w, want, pb.F, side, "F" + tag, dout_amax=amax_in, fin=fin)
        rt.call("selfc_coupling_bwd_x", 1, ws.x2.data_ptr(), ws.s.data_ptr(), d2.data_ptr(), dx2.data_ptr(), dh.data_ptr(), clamp, nel,
                ptr(A[0]), ptr(A[1]), sp)
        gG, gH = gh_pair_bwd(blk, pb, ws.gd, ws.hd, keep, dx2, dh, -1.0, 1.0, d1, n, t, h, w, want, side, "GH" + tag,
                             amax_g=A[0], amax_h=A[1], dx_amax_out=A[2], fin=fin)
        return d1, dx2, gF, gG, gH, A[2]

    def h_backward(xin_gh, amax_dh, dx_amax=None):
        """H's whole backward (data chain, then its weight gradients) next to G's: own stream, own dx buffer."""
        if side_h is None:
            return subnet_bwd(blk.H, ws.hd, xin_gh, dh, 1.0, d1, True, n, t, h, w, want, pb.H, side, "H" + tag, dout_amax=amax_dh), None
        d1h = torch.empty_like(d1)
        side_h.wait_event(main.record_event())
        with torch.cuda.stream(side_h):
            g_ = subnet_bwd(blk.H, ws.hd, xin_gh, dh, 1.0, d1h, False, n, t, h, w, want, pb.H, None, "H" + tag,
                            on_data_done=lambda: ev_h.append(side_h.record_event()), dout_amax=amax_dh)
        return g_, d1h

    def join_d1(d1h, slot):
        """d1 += d1h (H's half of y1's gradient, from its own stream) with max|d1| of the sum; returns that max (or None)"""
        main.wait_event(ev_h[0])
        if fold:
            rt.call("selfc_add_absmax", d1.data_ptr(), d1h.data_ptr(), d1.numel(), ptr(slot), sp)
            return slot
        d1.add_(d1h)
        return None

    if not rev:
        # y1 = x1 + F(x2); y2 = x2*e^s + G(y1), s = s(H(y1)).  keep = x2 (input), ws.x1 = y1
        rt.call("selfc_coupling_bwd_x", 0, keep.data_ptr(), ws.s.data_ptr(), d2.data_ptr(), dx2.data_ptr(), dh.data_ptr(), clamp, nel,
                None, ptr(A[0]), sp)
        gH, d1h = h_backward(ws.x1, A[0])
        # G: dOut = d2 (max from the previous block's F epilogue).  Without the third stream H has already added its dx to d1 and
        # G's dx epilogue is the last writer of d1: its max is then F's
        gG = subnet_bwd(blk.G, ws.gd, ws.x1, d2, 1.0, d1, True, n, t, h, w, want, pb.G, side, "G" + tag, dout_amax=amax_in,
                        dx_amax_out=A[1] if (fold and d1h is None) else None)
        amax_d1 = A[1] if (fold and d1h is None) else None
        if d1h is not None:
            amax_d1 = join_d1(d1h, A[1])
        if restore_fd:
            # the forward's epilogue replaced F's f16 input copy by y2: put x2 back before F's weight gradients
            rt.call("selfc_nhwc_to_planes", keep.data_ptr(), ws.fd.data_ptr(), n * h * w, c2, sp)
        gF = subnet_bwd(blk.F, ws.fd, None, d1, 1.0, dx2, True, n, t, h, w, want, pb.F, side, "F" + tag, dout_amax=amax_d1, dx_amax_out=A[2])
        return d1, dx2, gF, gG, gH, A[2]                  # dx2 is the next block's d2
    # y2 = (x2 - G(x1))*e^-s, s = s(H(x1)); y1 = x1 - F(y2).  keep = x1 (input), ws.x2 = y2 (also in fd)
    gF = subnet_bwd(blk.F, ws.fd, None, d1, -1.0, d2, True, n, t, h, w, want, pb.F, side, "F" + tag, dout_amax=amax_in)
    rt.call("selfc_coupling_bwd_x", 1, ws.x2.data_ptr(), ws.s.data_ptr(), d2.data_ptr(), dx2.data_ptr(), dh.data_ptr(), clamp, nel,
            ptr(A[0]), ptr(A[1]), sp)
    gH, d1h = h_backward(keep, A[1])
    gG = subnet_bwd(blk.G, ws.gd, keep, dx2, -1.0, d1, True, n, t, h, w, want, pb.G, side, "G" + tag, dout_amax=A[0],
                    dx_amax_out=A[2] if (fold and d1h is None) else None)
    amax_d1 = A[2] if (fold and d1h is None) else None
    if d1h is not None:
        amax_d1 = join_d1(d1h, A[2])
    return d1, dx2, gF, gG, gH, amax_d1                   # d1 is the next block's F dOut


class InvStackFn(torch.autograd.Function):
    """The whole op loop of SelfCInvNet as ONE differentiable op (SelfC_GMM_arch_inv.py:452-456 forward, :486-490 reverse):
    FrequencyAnalyzer + every InvBlockExp, state kept in the kernels' latent layout from the first block to the last.

      rev == False:  x (N,3,H,W) image  -> (N,3+c2,h,w) latent        (selfc_freq_fwd, blocks 0..n-1)
      rev == True :  z (N,3+c2,h,w)     -> (N,3,H,W) reconstruction   (blocks n-1..0, selfc_freq_inv)

    Against one InvBlockFn per block this drops, per block call and direction, the NCHW <-> latent conversions (4 launches),
    the clone of the block's private x1 / x2 pair and the Python / autograd-node overhead; each block still owns the buffers its
    backward needs (fd / gd / hd, s, the input side its kernels overwrite, the output side its formulas read).  The G/H epilogue
    writes the f16 copy of the updated x2 straight into the NEXT block's F buffer (selfc_latent.fd_next), so F's input planes
    are still there in the backward."""

    @staticmethod
    def forward(ctx, x, net, rev, t, *params):
        from types import SimpleNamespace
        x = rt.as_input(x)
        freq = net.operations[0]
        k = freq.k
        blocks = net._blocks()
        order = list(reversed(blocks)) if rev else blocks
        c1, c2 = blocks[0].split_len1, blocks[0].split_len2
        dev, sp = x.device, _lib.stream_ptr()
        if not rev:
            n, _, H, W = x.shape
            h, w = H // k, W // k
        else:
            n, _, h, w = x.shape
            H, W = h * k, w * k
        kind = blocks[0].F.kind
        f32, F16 = torch.float32, _lib.operand_dtype()
        c2p, FC = roundup(c2, 4), dense_channels(c2)
        x1 = torch.empty((n, h, w, 4), dtype=f32, device=dev)
        x2 = torch.empty((n, h, w, c2p), dtype=f32, device=dev)
        pf = torch.empty((2, n, h, w, 12), dtype=f32, device=dev) if c2 == 48 else None

        # the blocks' F buffers as one allocation: only the pad channels of the last input plane need a zero fill (everything else is
        # fully written) - one strided fill for all blocks
        fd_all = torch.empty((len(order), FC // 32, n, h, w, 32), dtype=F16, device=dev)
        if c2 % 32:
            fd_all[:, c2 // 32].zero_()
        fds = list(fd_all.unbind(0))
        if not rev:
            rt.call("selfc_freq_fwd", x.data_ptr(), x1.data_ptr(), x2.data_ptr(), fds[0].data_ptr(), FC, n, H, W, k, sp)
        else:
            rt.call("selfc_nchw_to_latent", x.data_ptr(), x1.data_ptr(), x2.data_ptr(), None, FC, n, c1, c2, h, w, sp)
        saves = []
        for i, blk in enumerate(order):
            sv = SimpleNamespace(N=n, H=h, W=w, c1=c1, c2=c2, c2p=c2p, FC=FC, fd=fds[i],
                                 gd=torch.empty((4, n, h, w, 32), dtype=F16, device=dev),
                                 hd=torch.empty((4, n, h, w, 32), dtype=F16, device=dev),
                                 s=torch.empty((n, h, w, c2p), dtype=f32, device=dev), x1=None, x2=None, device=dev)
            # the block writes its updated x1 / x2 into fresh buffers (selfc_latent.x1_out / x2_out): the input side its kernels
            # would overwrite and the output side the gradient formulas read stay what they are - no clones
            y1 = torch.empty_like(x1)
            y2 = torch.empty_like(x2)
            keep = x1 if rev else x2
            nxt = fds[i + 1] if (not rev and i + 1 < len(order)) else None
            lat = _lib.Latent(kind, n, t, h, w, c1, c2, x1.data_ptr(), x2.data_ptr(), sv.fd.data_ptr(), sv.gd.data_ptr(),
                              sv.hd.data_ptr(), sv.s.data_ptr(), None if pf is None else pf.data_ptr(), _lib.LAT_KEEP_FEATURES,
                              None if nxt is None else nxt.data_ptr(), y1.data_ptr(), y2.data_ptr())
            rt.call("selfc_invblock_run", rt.packed_block(blk).struct(), lat, 1 if rev else 0, sp)
            x1, x2 = y1, y2
            if rev:
                sv.x2 = x2                                        # y2: read by the coupling gradient
            else:
                sv.x1 = x1                                        # y1: input of G / H
            blk._set_s_lazy(sv)
            saves.append((blk, sv, keep, nxt is not None))
        if not rev:
            out = torch.empty((n, c1 + c2, h, w), dtype=f32, device=dev)
            rt.call("selfc_latent_to_nchw", x1.data_ptr(), x2.data_ptr(), out.data_ptr(), n, c1, c2, h, w, sp)
        else:
            out = torch.empty((n, 3, H, W), dtype=f32, device=dev)
            rt.call("selfc_freq_inv", x1.data_ptr(), x2.data_ptr(), out.data_ptr(), n, h, w, k, sp)
        ctx.saves, ctx.rev, ctx.t, ctx.dims, ctx.blocks = saves, bool(rev), t, (n, h, w, H, W, c1, c2, c2p, FC), blocks
        return out

    @staticmethod
    def backward(ctx, gy):
        n, h, w, H, W, c1, c2, c2p, FC = ctx.dims
        rev, t = ctx.rev, ctx.t
        gy = gy.contiguous().float()
        dev, sp = gy.device, _lib.stream_ptr()
        want = any(ctx.needs_input_grad[4:])
        d1 = torch.empty((n, h, w, 4), dtype=torch.float32, device=dev)
        d2 = torch.empty((n, h, w, c2p), dtype=torch.float32, device=dev)
        if not rev:
            rt.call("selfc_nchw_to_latent", gy.data_ptr(), d1.data_ptr(), d2.data_ptr(), None, FC, n, c1, c2, h, w, sp)
        else:
            rt.call("selfc_freq_inv_bwd", gy.data_ptr(), d1.data_ptr(), d2.data_ptr(), n, H, W, sp)
        grads = {}
        # three max slots per block, zeroed by ONE fill (see _block_backward)
        slots = torch.zeros(3 * len(ctx.saves), dtype=torch.float32, device=dev) if _FOLD_AMAX else None
        amax = None
        # the weight-gradient partials of every subnet are reduced by ONE launch per 24 jobs behind the last block (FinJobs); each block
        # has its own scratch set (tag), so nothing is overwritten before that
        # reverse-direction stack = the first to run in a training backward: its weight gradients may go to the background
        bg = bool(_BG_ON and rev and want and _DEFER_FIN and _DEFER_WG and _PAIR)
        fin = FinJobs(6 * len(ctx.saves), defer_wg=_DEFER_WG and _PAIR, thin=bg) if (want and _DEFER_FIN) else None
        for i, (blk, sv, keep, fd_intact) in enumerate(reversed(ctx.saves)):
            d1, d2, gF, gG, gH, amax = _block_backward(blk, sv, keep, rev, t, d1, d2, want, restore_fd=not fd_intact, tag=str(i), amax_in=amax,
                                                       amax_slots=None if slots is None else slots[3 * i:3 * i + 3], fin=fin)
            grads[id(blk)] = (*gF, *gG, *gH)
        if fin is not None and fin.n:
            side = side_stream(dev)
            if fin.thin:
                # background: one fork here; the join is the next stack's flush (below) or the trainer's join_background()
                bgs = bg_stream(dev)
                bgs.wait_event(torch.cuda.current_stream().record_event())
                with torch.cuda.stream(bgs):
                    fin.flush()
                _BG_PENDING[str(dev)] = bgs
            elif side is None or fin.defer_wg:
                if side is not None:         # (a subnet kind whose launches cannot be deferred ran its weight phase there)
                    torch.cuda.current_stream().wait_stream(side)
                join_background(dev)         # an earlier stack's background finishes accumulate into the same gradient slices
                fin.flush()                  # the stack's weight gradients: two fat launches + the finishes, behind the data chain
            else:
                # behind every weight-gradient phase (the side stream is in order; H's non-paired phases, stream 1, are joined first)
                if side_stream(dev, 1) is not None:
                    side.wait_stream(side_stream(dev, 1))
                side.wait_event(torch.cuda.current_stream().record_event())
                with torch.cuda.stream(side):
                    fin.flush()
        dx = None
        if ctx.needs_input_grad[0]:
            if not rev:
                dx = torch.empty((n, 3, H, W), dtype=torch.float32, device=dev)
                rt.call("selfc_freq_fwd_bwd", d1.data_ptr(), d2.data_ptr(), dx.data_ptr(), n, H, W, sp)
            else:
                dx = torch.empty((n, c1 + c2, h, w), dtype=torch.float32, device=dev)
                rt.call("selfc_latent_to_nchw", d1.data_ptr(), d2.data_ptr(), dx.data_ptr(), n, c1, c2, h, w, sp)
        _join_side_streams(dev, want)
        flat = []
        for blk in ctx.blocks:                                    # parameter order of forward(): net._blocks() order
            flat += list(grads[id(blk)])
        return (dx, None, None, None, *flat)


class CouplingFn(torch.autograd.Function):
    """(y2, s) = affine coupling of InvBlockExp (Inv_arch.py:26-27,29-30) on NCHW tensors, as its own op: the composed path of
    a block with channel_split_num > 3 (the fused conv5 epilogues cover splits <= 3).  s is returned for InvBlockExp.s /
    jacobian and is not differentiated (as in the fused path)."""

    @staticmethod
    def forward(ctx, x2, g, h, clamp, rev):
        x2, g, h = rt.as_input(x2), rt.as_input(g), rt.as_input(h)
        n = x2.numel()
        pad = (-n) % 4                                   # the kernel works on float4 groups
        if pad:
            x2f, gf, hf = (torch.cat((t.reshape(-1), t.new_zeros(pad))) for t in (x2, g, h))
        else:
            x2f, gf, hf = x2.reshape(-1), g.reshape(-1), h.reshape(-1)
        y2 = torch.empty_like(x2f)
        s = torch.empty_like(x2f)
        rt.call("selfc_coupling_fwd", 1 if rev else 0, x2f.data_ptr(), gf.data_ptr(), hf.data_ptr(), y2.data_ptr(), s.data_ptr(),
                float(clamp), x2f.numel(), _lib.stream_ptr())
        ctx.rev, ctx.clamp, ctx.shape, ctx.n = bool(rev), float(clamp), x2.shape, n
        ctx.save_for_backward(y2 if rev else x2f, s)
        y2o, so = y2[:n].reshape(x2.shape), s[:n].reshape(x2.shape)
        ctx.mark_non_differentiable(so)
        return y2o, so

    @staticmethod
    def backward(ctx, dy2, _ds):
        v, s = ctx.saved_tensors
        n = ctx.n
        d = dy2.contiguous().float().reshape(-1)
        if v.numel() != n:
            d = torch.cat((d, d.new_zeros(v.numel() - n)))
        dx2 = torch.empty_like(v)
        dh = torch.empty_like(v)
        rt.call("selfc_coupling_bwd", 1 if ctx.rev else 0, v.data_ptr(), s.data_ptr(), d.data_ptr(), dx2.data_ptr(), dh.data_ptr(),
                ctx.clamp, v.numel(), _lib.stream_ptr())
        dx2, dh = dx2[:n].reshape(ctx.shape), dh[:n].reshape(ctx.shape)
        dg = -dx2 if ctx.rev else dy2
        return dx2, dg, dh, None, None


def block_params(blk) -> List[torch.Tensor]:
    return subnet_params(blk.F) + subnet_params(blk.G) + subnet_params(blk.H)


class FreqFn(torch.autograd.Function):
    """FrequencyAnalyzer.forward(x, rev) (SelfC_GMM_arch_inv.py:69-82) with its adjoint as backward."""

    @staticmethod
    def forward(ctx, x, mod, rev):
        ctx.rev = bool(rev)
        ctx.shape = tuple(x.shape)
        with torch.no_grad():
            return mod._run(x, rev)

    @staticmethod
    def backward(ctx, gy):
        gy = gy.contiguous().float()
        dev, sp = gy.device, _lib.stream_ptr()
        if not ctx.rev:
            n, _, hh, ww = ctx.shape                                   # input (N,3,H,W), gy (N,51,H/4,W/4)
            h, w = hh // 4, ww // 4
            d1 = torch.empty((n, h, w, 4), dtype=torch.float32, device=dev)
            d2 = torch.empty((n, h, w, 48), dtype=torch.float32, device=dev)
            rt.call("selfc_nchw_to_latent", gy.data_ptr(), d1.data_ptr(), d2.data_ptr(), None, 0, n, 3, 48, h, w, sp)
            dx = torch.empty((n, 3, hh, ww), dtype=torch.float32, device=dev)
            rt.call("selfc_freq_fwd_bwd", d1.data_ptr(), d2.data_ptr(), dx.data_ptr(), n, hh, ww, sp)
            return dx, None, None
        n, _, h, w = ctx.shape                                          # input (N,51,h,w), gy (N,3,4h,4w)
        d1 = torch.empty((n, h, w, 4), dtype=torch.float32, device=dev)
        d2 = torch.empty((n, h, w, 48), dtype=torch.float32, device=dev)
        rt.call("selfc_freq_inv_bwd", gy.data_ptr(), d1.data_ptr(), d2.data_ptr(), n, 4 * h, 4 * w, sp)
        dx = torch.empty((n, 51, h, w), dtype=torch.float32, device=dev)
        rt.call("selfc_latent_to_nchw", d1.data_ptr(), d2.data_ptr(), dx.data_ptr(), n, 3, 48, h, w, sp)
        return dx, None, None


class HaarFn(torch.autograd.Function):
    """HaarDownsampling.forward(x, rev) (Inv_arch.py:60-79).  y = W x / 4 with W^T W = 4 I, reverse x = W^T y, so the
    adjoints are the opposite-direction kernels: d/dx of forward = haar_inv(dy) / 4, of reverse = 4 haar_fwd(dx)."""

    @staticmethod
    def forward(ctx, x, mod, rev):
        ctx.mod, ctx.rev = mod, bool(rev)
        with torch.no_grad():
            return mod._run(x, rev)

    @staticmethod
    def backward(ctx, gy):
        with torch.no_grad():
            g = ctx.mod._run(gy.contiguous().float(), not ctx.rev, track=False)
        return (g * 0.25 if not ctx.rev else g * 4.0), None, None


def needs_grad(*tensors) -> bool:
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def module_needs_grad(x, mod) -> bool:
    """needs_grad(x, *mod.parameters()) on the cached parameter list."""
    if not torch.is_grad_enabled():
        return False
    return (x is not None and x.requires_grad) or any(p.requires_grad for p in rt.plist(mod))


# ------------------------------------------------------------------------------------------------------------
# STP (SelfC_GMM_arch_inv.py:289-430): GlobalAgg, the 1x1x1 head and the GMM sampler
# ------------------------------------------------------------------------------------------------------------

def _buf(cache: Dict, name: str, nbytes: int, device) -> torch.Tensor:
    b = cache.get(name)
    if b is None or b.numel() < nbytes or b.device != device:
        b = cache[name] = torch.empty(nbytes, dtype=torch.uint8, device=device)
    return b


_STP_CACHE: Dict = {}


def _sink_add(param, grad: torch.Tensor) -> Optional[torch.Tensor]:
    """A parameter gradient computed with torch ops: with a gradient sink it is ADDED into the parameter's view of the flat buffer
    here, on the calling stream, and autograd sees None (no AccumulateGrad node, no copy); without a sink it is returned unchanged."""
    if _SINK is not None:
        v = _SINK.view_of(param)
        if v is not None:
            v.add_(grad.reshape(v.shape))
            return None
    return grad


def globalagg_bwd(m, x: torch.Tensor, dy: torch.Tensor, dx: torch.Tensor, n: int, t: int, h: int, w: int, defer_fc: Optional[list] = None,
                  dy_amax: Optional[torch.Tensor] = None, dx_amax_out: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    """Backward of GlobalAgg.run_nhwc: x, dy, dx fp32 [n][h*w][64]; returns {parameter name: gradient}.  dy_amax: one-float tensor with
    max|dy| taken by dy's producer (the call then skips its own pass); dx_amax_out: zeroed one-float tensor that receives max|dx|."""
    pk = m._packed(h, w)
    key = rt.params_key(m)
    if getattr(m, "_w1t_key", None) != key:
        m._w1t = pack_planes_generic(m.proj1.weight.detach().float().reshape(64, 64).t().reshape(64, 64, 1, 1).contiguous())
        m._w1t_key = key
    dev, b = x.device, n // t
    f32 = dict(dtype=torch.float32, device=dev)
    dw1 = torch.empty((64, 64), **f32)
    db1c, db2c, db3c = (torch.empty((b, 64), **f32) for _ in range(3))
    dw2c, dw3c = (torch.empty((b, 64 * 64), **f32) for _ in range(2))
    dfcbc = torch.empty((b,), **f32)
    dwmapc = torch.empty((b, h * w), **f32)
    need = _lib.lib().selfc_globalagg_bwd_scratch_bytes(n, t, h, w)
    sc = _buf(_STP_CACHE, "gagg", need, dev)
    rt.call("selfc_globalagg_bwd_x", x.data_ptr(), dy.data_ptr(), dx.data_ptr(), pk["wmap"].data_ptr(), pk["fcb"].data_ptr(),
            m._w1t.data_ptr(), pk["b1"].data_ptr(), pk["w2"].data_ptr(), pk["b2"].data_ptr(), pk["w3"].data_ptr(), pk["b3"].data_ptr(),
            dw1.data_ptr(), db1c.data_ptr(), dw2c.data_ptr(), db2c.data_ptr(), dw3c.data_ptr(), db3c.data_ptr(),
            dfcbc.data_ptr(), dwmapc.data_ptr(), sc.data_ptr(), sc.numel(), n, t, h, w,
            None if dy_amax is None else dy_amax.data_ptr(), None if dx_amax_out is None else dx_amax_out.data_ptr(), _lib.stream_ptr())
    # the sums over the clips: ONE launch for the seven per-clip tensors (they were 7 torch sums + as many autograd accumulations);
    # with the trainer's flat gradient buffer the results are ADDED straight into the parameters' views and autograd sees None
    names = ["proj1.weight", "proj1.bias", "proj2.weight", "proj2.bias", "proj3.weight", "proj3.bias", "fc.bias"]
    srcs = [dw1, db1c, dw2c, db2c, dw3c, db3c, dfcbc]
    lens = [64 * 64, 64, 64 * 64, 64, 64 * 64, 64, 1]
    rows = [1, b, b, b, b, b, b]
    sunk = None
    if _SINK is not None:
        sunk = [_SINK.view_of(getattr(getattr(m, nm.split(".")[0]), nm.split(".")[1])) for nm in names]
        if any(v is None for v in sunk):
            sunk = None
    dwmap = torch.empty((h * w,), **f32)
    outs = sunk if sunk is not None else [torch.empty((ln,), **f32) for ln in lens]
    job = _lib.RowSum()
    for i, (src, dst, ln, rw) in enumerate(zip(srcs + [dwmapc], outs + [dwmap], lens + [h * w], rows + [b])):
        job.src[i], job.dst[i], job.len[i], job.rows[i] = src.data_ptr(), dst.data_ptr(), ln, rw
        job.beta[i] = 1.0 if (sunk is not None and i < 7) else 0.0
    job.n = 8
    rt.call("selfc_rowsum_accum", C.byref(job), _lib.stream_ptr())
    if defer_fc is not None:          # the caller folds all its blocks' map gradients through the pooling map in one batch
        defer_fc.append((m, dwmap))
        g = {"fc.weight": None}
    else:
        g = {"fc.weight": _sink_add(m.fc.weight, pool_weight_map_grad(dwmap, h, w))}
    for nm, o, p_shape in zip(names, outs, [(64, 64, 1, 1), (64,), (64, 64), (64,), (64, 64), (64,), (1,)]):
        g[nm] = None if sunk is not None else o.reshape(p_shape)
    return g


class GlobalAggFn(torch.autograd.Function):
    """GlobalAgg.forward (SelfC_GMM_arch_inv.py:265-285), NCHW in/out."""

    @staticmethod
    def forward(ctx, x, mod, t, *params):
        x = rt.as_input(x)
        n, c, h, w = x.shape
        sp = _lib.stream_ptr()
        xin = torch.empty((n, h * w, 64), dtype=torch.float32, device=x.device)
        rt.call("selfc_nchw_to_nhwc4", x.data_ptr(), xin.data_ptr(), n, 64, h, w, sp)
        yout = torch.empty_like(xin)
        mod.run_nhwc(xin, yout, n, t, h, w, mod.__dict__.setdefault("_scratch", {}))
        y = torch.empty_like(x)
        rt.call("selfc_nhwc4_to_nchw", yout.data_ptr(), y.data_ptr(), n, 64, h, w, sp)
        ctx.mod, ctx.t, ctx.xin, ctx.shape = mod, t, xin, (n, h, w)
        return y

    @staticmethod
    def backward(ctx, gy):
        mod, t, xin = ctx.mod, ctx.t, ctx.xin
        n, h, w = ctx.shape
        gy = gy.contiguous().float()
        sp = _lib.stream_ptr()
        dy = torch.empty_like(xin)
        rt.call("selfc_nchw_to_nhwc4", gy.data_ptr(), dy.data_ptr(), n, 64, h, w, sp)
        dxl = torch.empty_like(xin)
        g = globalagg_bwd(mod, xin, dy, dxl, n, t, h, w)
        dx = torch.empty((n, 64, h, w), dtype=torch.float32, device=gy.device)
        rt.call("selfc_nhwc4_to_nchw", dxl.data_ptr(), dx.data_ptr(), n, 64, h, w, sp)
        return (dx, None, None, *[g[name] for name, _ in mod.named_parameters()])


def _head_convs(stp):
    return [m for m in stp.tail_gmm if isinstance(m, torch.nn.Conv3d)]


def _head_bwd(convs, feat, acts, dlast, n, t, h, w, relu_hidden: bool = False) -> Tuple[torch.Tensor, Dict[int, Tuple[torch.Tensor, torch.Tensor]]]:
    """Backward of a [lrelu, conv1x1x1]* head given d(last conv output) `dlast` fp32 [npix][>= Cl].
    feat: fp32 [npix][C0] (input of the head, C0 a multiple of 32), acts: the saved post-activation f16 rows of the hidden
    layers (LeakyReLU; relu_hidden: ReLU, the 'gmm_thin' head of SelfC_GMM_arch_inv.py:345-354 - the activation in FRONT of
    the first conv is LeakyReLU in every head).  Returns (dfeat fp32 [npix][C0], {conv index: (dweight, dbias)})."""
    dev, sp = feat.device, _lib.stream_ptr()
    npix = n * h * w
    c0 = feat.shape[-1]
    F16 = _lib.operand_dtype()
    L = _lib.lib()
    amax = torch.zeros(64, dtype=torch.float32, device=dev)
    rt.call("selfc_bwd_scale", dlast.data_ptr(), dlast.numel(), amax.data_ptr(), sp)
    cl = convs[-1].out_channels
    gp = torch.empty((roundup(cl, 32) // 32, npix, 32), dtype=F16, device=dev)
    rt.call("selfc_bwd_to_planes", dlast.data_ptr(), gp.data_ptr(), npix, cl, dlast.shape[-1], 0, 1.0, amax.data_ptr(), sp)
    # activation planes: lrelu(feat), then the hidden activations
    inputs = [torch.empty((c0 // 32, npix, 32), dtype=F16, device=dev)]
    rt.call("selfc_bwd_to_planes", feat.data_ptr(), inputs[0].data_ptr(), npix, c0, c0, 1, 1.0, None, sp)
    for a in acts:
        c = a.shape[-1]
        pl = torch.empty((c // 32, npix, 32), dtype=F16, device=dev)
        rt.call("selfc_f16_rows_to_planes", a.data_ptr(), pl.data_ptr(), npix, c, sp)
        inputs.append(pl)
    grads: Dict[int, Tuple[torch.Tensor, torch.Tensor]] = {}
    dfeat = torch.empty((npix, c0), dtype=torch.float32, device=dev)
    for li in range(len(convs) - 1, -1, -1):
        conv, q = convs[li], inputs[li]
        cout, cin = conv.out_channels, conv.in_channels
        pn, qn = gp.shape[0], q.shape[0]
        need = L.selfc_bwd_wgrad_scratch_bytes(n, h, w, pn, qn, 1)
        sc = _buf(_STP_CACHE, "wgrad", need, dev)
        sw = sb = None
        if _SINK is not None:          # the trainer's flat gradient buffer: the kernel ADDS into the parameters' views, autograd sees None
            sw, sb = _SINK.view_of(conv.weight), _SINK.view_of(conv.bias)
        if sw is not None and sb is not None:
            rt.call("selfc_bwd_wgrad", gp.data_ptr(), pn, q.data_ptr(), qn, 1, sw.data_ptr(), cout, cin, sb.data_ptr(), 1.0,
                    amax.data_ptr(), sc.data_ptr(), sc.numel(), n, t, h, w, sp)
            grads[li] = (None, None)
        else:
            gw = torch.empty((cout, cin), dtype=torch.float32, device=dev)
            gb = torch.empty((cout,), dtype=torch.float32, device=dev)
            rt.call("selfc_bwd_wgrad", gp.data_ptr(), pn, q.data_ptr(), qn, 1, gw.data_ptr(), cout, cin, gb.data_ptr(), 0.0,
                    amax.data_ptr(), sc.data_ptr(), sc.numel(), n, t, h, w, sp)
            grads[li] = (gw.reshape(conv.weight.shape), gb)
        wt = conv.__dict__.get("_wt_pk") if conv.__dict__.get("_wt_key") == rt.params_key(conv) else pack_pointwise_T(conv.weight)
        if li > 0:
            nxt = torch.empty((cin // 32, npix, 32), dtype=F16, device=dev)
            rt.call("selfc_bwd_conv_planes", gp.data_ptr(), pn, 1, 1, wt.data_ptr(), cin // 32, nxt.data_ptr(), None,
                    q.data_ptr(), -3 if relu_hidden else -2, None, 0, 0, amax.data_ptr(), n, t, h, w, sp)
            gp = nxt
        else:
            rt.call("selfc_bwd_conv_planes", gp.data_ptr(), pn, 1, 1, wt.data_ptr(), c0 // 32, None, None, None, -1,
                    dfeat.data_ptr(), c0, 0, amax.data_ptr(), n, t, h, w, sp)
            rt.call("selfc_lrelu_bwd", dfeat.data_ptr(), feat.data_ptr(), dfeat.numel(), sp)
    return dfeat, grads


class STPSampleFn(torch.autograd.Function):
    """STPNet.forward + sample() (SelfC_GMM_arch_inv.py:358-394) as one differentiable op:
    lr (N,3,h,w) -> predicted HF (N,48,h,w) ('gmm': the reparameterised sample, 'l2': the head output)."""

    @staticmethod
    def forward(ctx, lr, stp, t, eps, *params):
        from .modules.Subnet_constructor import D2DTInput
        lr = rt.as_input(lr)
        n, _, h, w = lr.shape
        dev, sp = lr.device, _lib.stream_ptr()
        F16 = _lib.operand_dtype()
        npix = n * h * w
        x1 = torch.empty((n, h, w, 4), dtype=torch.float32, device=dev)
        rt.call("selfc_nchw_to_nhwc4", lr.data_ptr(), x1.data_ptr(), n, 3, h, w, sp)
        stages, cur = [], x1
        scratch = stp.__dict__.setdefault("_scratch_train", {})
        for m in stp._chain():
            dst = torch.empty((n, h * w, 64), dtype=torch.float32, device=dev)
            if isinstance(m, D2DTInput):
                dense = _dense_buffer(m.channel_in, n, h, w, dev)
                sw = m.packed().struct()
                rt.call("selfc_subnet_run", sw, m.kind, cur.data_ptr(), dst.data_ptr(), dense.data_ptr(),
                        n, t, h, w, m.channel_in, m.channel_out, sp)
                stages.append((m, cur if m.channel_in <= 3 else None, dense))
            else:
                m.run_nhwc(cur, dst, n, t, h, w, scratch)
                stages.append((m, cur, None))
            cur = dst
        feat = cur
        tail = stp._tail_packed()
        hf = torch.empty((npix, stp.hf_dim), dtype=torch.float32, device=dev)
        acts, raw = [], None
        if stp.fh_loss == "l2":
            wp, bp, cin, cout = tail[0]
            rt.call("selfc_pwconv_run", feat.data_ptr(), 1, hf.data_ptr(), 1, wp.data_ptr(), bp.data_ptr(), npix, cin, cout, cout, 1, 0, sp)
        else:
            (w0, b0, ci0, co0), (w1_, b1_, ci1, co1), (w2_, b2_, ci2, co2) = tail
            h1 = torch.empty((npix, co0), dtype=F16, device=dev)
            h2 = torch.empty((npix, co1), dtype=F16, device=dev)
            raw = torch.empty((npix, co2), dtype=torch.float32, device=dev)
            act = 2 if stp.fh_loss == "gmm_thin" else 1          # hidden activations: ReLU ('gmm_thin', :345-354) | LeakyReLU
            rt.call("selfc_pwconv_run", feat.data_ptr(), 1, h1.data_ptr(), 0, w0.data_ptr(), b0.data_ptr(), npix, ci0, co0, co0, 1, act, sp)
            rt.call("selfc_pwconv_run", h1.data_ptr(), 0, h2.data_ptr(), 0, w1_.data_ptr(), b1_.data_ptr(), npix, ci1, co1, co1, 0, act, sp)
            rt.call("selfc_pwconv_run", h2.data_ptr(), 0, raw.data_ptr(), 1, w2_.data_ptr(), b2_.data_ptr(), npix, ci2, co2, co2, 0, 0, sp)
            if eps is None:
                eps = torch.randn((npix, stp.hf_dim * stp.K), dtype=torch.float32, device=dev)
            if stp.hf_dim == 48 and co2 == 48 * stp.K * 3:
                rt.call("selfc_gmm_sample", raw.data_ptr(), eps.data_ptr(), hf.data_ptr(), npix, stp.hf_dim, stp.K, sp)
            else:      # other scales (hf_dim = 3 scale^2): the generic sampler on rows padded to a multiple of 16 channels
                rt.call("selfc_gmm_sample_generic", raw.data_ptr(), eps.data_ptr(), hf.data_ptr(), npix, stp.hf_dim, stp.K, co2, stp.hf_dim, 1.0, sp)
            acts = [h1, h2]
        y = torch.empty((n, stp.hf_dim, h, w), dtype=torch.float32, device=dev)
        rt.call("selfc_nhwc4_to_nchw", hf.data_ptr(), y.data_ptr(), n, stp.hf_dim, h, w, sp)
        ctx.stp, ctx.t, ctx.shape = stp, t, (n, h, w)
        ctx.stages, ctx.feat, ctx.acts, ctx.raw, ctx.eps = stages, feat, acts, raw, eps
        ctx.nparams = len(params)
        return y

    @staticmethod
    def backward(ctx, gy):
        from .modules.Subnet_constructor import D2DTInput
        stp, t = ctx.stp, ctx.t
        n, h, w = ctx.shape
        gy = gy.contiguous().float()
        dev, sp = gy.device, _lib.stream_ptr()
        npix = n * h * w
        dv = torch.empty((npix, stp.hf_dim), dtype=torch.float32, device=dev)
        rt.call("selfc_nchw_to_nhwc4", gy.data_ptr(), dv.data_ptr(), n, stp.hf_dim, h, w, sp)
        if stp.fh_loss == "l2":
            dlast = dv
        else:
            dlast = torch.empty_like(ctx.raw)
            if stp.hf_dim == 48 and ctx.raw.shape[-1] == 48 * stp.K * 3:
                rt.call("selfc_gmm_sample_bwd", ctx.raw.data_ptr(), ctx.eps.data_ptr(), dv.data_ptr(), dlast.data_ptr(), npix, stp.hf_dim, stp.K, sp)
            else:
                rt.call("selfc_gmm_sample_generic_bwd", ctx.raw.data_ptr(), ctx.eps.data_ptr(), dv.data_ptr(), dlast.data_ptr(), npix,
                        stp.hf_dim, stp.K, ctx.raw.shape[-1], stp.hf_dim, 1.0, sp)
        d, head_grads = _head_bwd(_head_convs(stp), ctx.feat, ctx.acts, dlast, n, t, h, w, relu_hidden=stp.fh_loss == "gmm_thin")
        grads: Dict[int, torch.Tensor] = {}
        for conv, (gw, gb) in zip(_head_convs(stp), [head_grads[i] for i in range(len(head_grads))]):
            grads[id(conv.weight)], grads[id(conv.bias)] = gw, gb
        d = d.reshape(n, h * w, 64)
        side = side_stream(dev)
        turn = 0                                           # two scratch slots; subnet_bwd reuses a slot only behind its last weight phase
        fc_maps: list = []
        # the chain's dense blocks leave their weight-gradient launches and finishes to ONE launch per kind behind the chain (FinJobs,
        # as the block stacks do): every subnet then needs its own scratch slot until the flush
        n_sub = sum(isinstance(m_, D2DTInput) for m_, _, _ in ctx.stages)
        fin = FinJobs(2 * n_sub, defer_wg=True) if (_DEFER_FIN and _DEFER_WG and n_sub) else None
        # max|gradient| travels with the gradient (as inside the block stacks): every stage leaves the maximum of the input gradient it
        # writes for the next stage's scale - one fill for all slots instead of a zero + a pass per stage (the head's output has none)
        slots = torch.zeros(len(ctx.stages), dtype=torch.float32, device=dev) if _FOLD_AMAX else None
        amax_prev = None
        for si, (m, xin, dense) in enumerate(reversed(ctx.stages)):
            amax_out = None if slots is None else slots[si:si + 1]
            if isinstance(m, D2DTInput):
                dxl = torch.empty((n, h, w, roundup(m.channel_in, 4)), dtype=torch.float32, device=dev)
                g = subnet_bwd(m, dense, xin, d, 1.0, dxl, False, n, t, h, w, True, None, side, f"stp{turn}", fin=fin,
                               dout_amax=amax_prev, dx_amax_out=amax_out)
                turn = turn + 1 if fin is not None else turn ^ 1
                for prm, gg in zip(subnet_params(m), g):
                    grads[id(prm)] = gg
                d = dxl
            else:
                dxl = torch.empty_like(d)
                g = globalagg_bwd(m, xin, d.reshape(n, h * w, 64), dxl, n, t, h, w, defer_fc=fc_maps, dy_amax=amax_prev, dx_amax_out=amax_out)
                for name, prm in m.named_parameters():
                    grads[id(prm)] = g[name]
                d = dxl
            amax_prev = amax_out
        if fin is not None:
            if side is not None:
                torch.cuda.current_stream().wait_stream(side)
            fin.flush()
        if fc_maps:                   # d fc.weight of every GlobalAgg: ONE batched fold (packing.pool_weight_map_grad_batch)
            folded = pool_weight_map_grad_batch(torch.stack([dm for _, dm in fc_maps]), h, w).float().contiguous()
            views = [(_SINK.view_of(m.fc.weight) if _SINK is not None else None) for m, _ in fc_maps]
            if all(v is not None for v in views) and len(fc_maps) <= 8:
                # added into the flat gradient buffer by ONE launch (selfc_rowsum_accum: up to eight (source, destination) pairs)
                job = _lib.RowSum()
                for i, v in enumerate(views):
                    job.src[i], job.dst[i], job.len[i], job.rows[i], job.beta[i] = folded[i].data_ptr(), v.data_ptr(), 32 * 32, 1, 1.0
                    grads[id(fc_maps[i][0].fc.weight)] = None
                job.n = len(views)
                rt.call("selfc_rowsum_accum", C.byref(job), sp)
            else:
                for i, (m, _) in enumerate(fc_maps):
                    grads[id(m.fc.weight)] = _sink_add(m.fc.weight, folded[i].reshape(1, 32 * 32))
        dlr = None
        if ctx.needs_input_grad[0]:
            dlr = torch.empty((n, 3, h, w), dtype=torch.float32, device=dev)
            rt.call("selfc_nhwc4_to_nchw", d.data_ptr(), dlr.data_ptr(), n, 3, h, w, sp)
        _join_side_streams(dev, True)
        return (dlr, None, None, None, *[grads.get(id(p)) for p in rt.plist(stp)])


class PointwiseHeadFn(torch.autograd.Function):
    """LeakyReLU + Conv3d 1x1x1 head of STP v1 (SelfC_arch_inv.py:139-141,170-176): x (N,C,h,w) -> (N,cout,h,w)."""

    @staticmethod
    def forward(ctx, x, conv, packed, t, weight, bias):
        x = rt.as_input(x)
        n, cc, h, w = x.shape
        dev, sp = x.device, _lib.stream_ptr()
        cout = conv.out_channels
        coutp = roundup(cout, 16)
        feat = torch.empty((n, h, w, cc), dtype=torch.float32, device=dev)
        rt.call("selfc_nchw_to_nhwc4", x.data_ptr(), feat.data_ptr(), n, cc, h, w, sp)
        wp, bp = packed
        outp = torch.empty((n, h, w, coutp), dtype=torch.float32, device=dev)
        rt.call("selfc_pwconv_run", feat.data_ptr(), 1, outp.data_ptr(), 1, wp.data_ptr(), bp.data_ptr(), n * h * w, cc, coutp, coutp, 1, 0, sp)
        ctx.conv, ctx.t, ctx.feat, ctx.shape = conv, t, feat, (n, cc, h, w)
        return outp[..., :cout].permute(0, 3, 1, 2).contiguous()

    @staticmethod
    def backward(ctx, gy):
        conv, t, feat = ctx.conv, ctx.t, ctx.feat
        n, cc, h, w = ctx.shape
        dev, sp = gy.device, _lib.stream_ptr()
        cout = conv.out_channels
        cs = roundup(cout, 4)
        dlast = torch.empty((n, h, w, cs), dtype=torch.float32, device=dev)
        rt.call("selfc_nchw_to_nhwc4", gy.contiguous().float().data_ptr(), dlast.data_ptr(), n, cout, h, w, sp)
        dfeat, grads = _head_bwd([conv], feat.reshape(n * h * w, cc), [], dlast.reshape(n * h * w, cs), n, t, h, w)
        dx = torch.empty((n, cc, h, w), dtype=torch.float32, device=dev)
        rt.call("selfc_nhwc4_to_nchw", dfeat.data_ptr(), dx.data_ptr(), n, cc, h, w, sp)
        gw, gb = grads[0]
        return dx, None, None, None, gw, gb


class HeadFn(torch.autograd.Function):
    """A whole [LeakyReLU, Conv3d 1x1x1]* head (STP v1's GMM head, SelfC_arch_inv.py:118-128,151-153): x (N,C,h,w) ->
    the raw output of the last conv (N,cout,h,w).  `packed`: [(fragments, bias, cin, cout_padded)] per conv (the module's
    _tail_packed()); params = w0, b0, w1, b1, ...  Hidden activations are kept as f16 rows for the backward."""

    @staticmethod
    def forward(ctx, x, convs, packed, t, *params):
        x = rt.as_input(x)
        n, cc, h, w = x.shape
        dev, sp = x.device, _lib.stream_ptr()
        npix = n * h * w
        feat = torch.empty((n, h, w, cc), dtype=torch.float32, device=dev)
        rt.call("selfc_nchw_to_nhwc4", x.data_ptr(), feat.data_ptr(), n, cc, h, w, sp)
        cur, cur_f32, acts = feat, 1, []
        for i, (wp, bp, ci, co) in enumerate(packed):
            last = i == len(packed) - 1
            out = torch.empty((npix, co), dtype=torch.float32 if last else _lib.operand_dtype(), device=dev)
            rt.call("selfc_pwconv_run", cur.data_ptr(), cur_f32, out.data_ptr(), 1 if last else 0, wp.data_ptr(), bp.data_ptr(),
                    npix, ci, co, co, 1 if i == 0 else 0, 0 if last else 1, sp)
            if not last:
                acts.append(out)
            cur, cur_f32 = out, 0
        cout = convs[-1].out_channels
        ctx.convs, ctx.t, ctx.feat, ctx.acts, ctx.shape, ctx.cop = convs, t, feat, acts, (n, cc, h, w), cur.shape[-1]
        return cur[:, :cout].reshape(n, h, w, cout).permute(0, 3, 1, 2).contiguous()

    @staticmethod
    def backward(ctx, gy):
        convs, t, feat = ctx.convs, ctx.t, ctx.feat
        n, cc, h, w = ctx.shape
        dev, sp = gy.device, _lib.stream_ptr()
        cout = convs[-1].out_channels
        cs = roundup(cout, 4)
        dlast = torch.empty((n, h, w, cs), dtype=torch.float32, device=dev)
        rt.call("selfc_nchw_to_nhwc4", gy.contiguous().float().data_ptr(), dlast.data_ptr(), n, cout, h, w, sp)
        dfeat, grads = _head_bwd(convs, feat.reshape(n * h * w, cc), ctx.acts, dlast.reshape(n * h * w, cs), n, t, h, w)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((n, cc, h, w), dtype=torch.float32, device=dev)
            rt.call("selfc_nhwc4_to_nchw", dfeat.data_ptr(), dx.data_ptr(), n, cc, h, w, sp)
        flat = []
        for i in range(len(convs)):
            flat += list(grads[i])
        return (dx, None, None, None, *flat)


class GmmSampleFn(torch.autograd.Function):
    """v[c] = sum_k softmax_c(raw[c,k,0]) * (eps[c,k] * exp(ls_scale * clamp(raw[c,k,1], -7, 7)) + raw[c,k,2]) for any
    (hf_dim, K): the reparameterised sample of STP v1's GMM head (SelfC_arch_inv.py:151-163,179-186: ls_scale = 0.5).
    raw (N, hf*K*3, h, w) NCHW, eps fp32 rows [N*h*w][hf*K] (not differentiated) -> v (N, hf, h, w)."""

    @staticmethod
    def forward(ctx, raw, eps, hf, k, ls_scale):
        raw = rt.as_input(raw)
        n, c, h, w = raw.shape
        dev, sp = raw.device, _lib.stream_ptr()
        npix = n * h * w
        rows = raw.permute(0, 2, 3, 1).contiguous()
        v = torch.empty((npix, hf), dtype=torch.float32, device=dev)
        rt.call("selfc_gmm_sample_generic", rows.data_ptr(), eps.data_ptr(), v.data_ptr(), npix, hf, k, c, hf, float(ls_scale), sp)
        ctx.rows, ctx.eps, ctx.args, ctx.shape = rows, eps, (hf, k, float(ls_scale)), (n, c, h, w)
        return v.reshape(n, h, w, hf).permute(0, 3, 1, 2).contiguous()

    @staticmethod
    def backward(ctx, gv):
        hf, k, ls = ctx.args
        n, c, h, w = ctx.shape
        sp = _lib.stream_ptr()
        dv = gv.permute(0, 2, 3, 1).contiguous().float()
        draw = torch.empty_like(ctx.rows)
        rt.call("selfc_gmm_sample_generic_bwd", ctx.rows.data_ptr(), ctx.eps.data_ptr(), dv.data_ptr(), draw.data_ptr(), n * h * w,
                hf, k, c, hf, ls, sp)
        return draw.reshape(n, h, w, c).permute(0, 3, 1, 2).contiguous(), None, None, None, None
