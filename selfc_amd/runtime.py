"""Host-side runtime shared by the drop-in modules: workspace cache, packed-weight
cache, and thin wrappers over the C ABI (include/selfc_hip.h)."""
from __future__ import annotations

import atexit
import os
import ctypes as C
import weakref
from typing import Dict, Optional, Tuple

import torch

from . import _lib
from .packing import PackPlan, dense_channels, roundup, subnet_pack_entries

SUBNET_D2DT = _lib.SUBNET_D2DT
SUBNET_DB2D = _lib.SUBNET_DB2D


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def no_autograd_guard(*tensors):
    """For the modules without a HIP backward yet (FeatureCalapseBlock, hence STP v1 with its default conditioner): refuse
    loudly instead of silently returning tensors without a graph."""
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors):
        raise NotImplementedError(
            "selfc_amd: this module has no backward kernels yet - call it under torch.no_grad() "
            "(differentiable today: DenseBlock, D2DTInput, InvBlockExp, HaarDownsampling, FrequencyAnalyzer, GlobalAgg, "
            "STPNet v2 / SelfCInvNet large; see DESIGN.md section 4b)")


class Workspace:
    """Latent state + dense feature buffers for one (kind, N, H, W, c1, c2) problem.

    Layout: include/selfc_hip.h.  Buffers are zero-initialised once: the pad
    channels of x2 / fd are never written by any kernel and must read as 0."""

    def __init__(self, device, kind: int, N: int, T: int, H: int, W: int, c1: int, c2: int, single_use: bool = False):
        self.kind, self.N, self.T, self.H, self.W, self.c1, self.c2 = kind, N, T, H, W, c1, c2
        self.c2p = roundup(c2, 4)
        self.FC = dense_channels(c2)
        f32, f16 = torch.float32, _lib.operand_dtype()
        # single_use (training: one workspace per block call, filled by selfc_nchw_to_latent which writes every element
        # of x1 / x2 incl. pads, then by the conv epilogues which write every feature channel): only the pad channels of
        # fd's input planes still need the zero fill
        mk = torch.empty if single_use else torch.zeros
        self.x1 = mk((N, H, W, 4), dtype=f32, device=device)
        self.x2 = mk((N, H, W, self.c2p), dtype=f32, device=device)
        # dense buffers are plane-blocked: [C/32][N][H][W][32] (every 32-channel group contiguous per pixel)
        self.fd = torch.zeros((self.FC // 32, N, H, W, 32), dtype=f16, device=device)
        self.gd = mk((4, N, H, W, 32), dtype=f16, device=device)
        self.hd = mk((4, N, H, W, 32), dtype=f16, device=device)
        # F conv5 partial products of the pairwise-fused F launches (every in-frame element is rewritten by each block call)
        self.pf = torch.empty((2, N, H, W, 12), dtype=f32, device=device) if c2 == 48 else None
        self.s: Optional[torch.Tensor] = None
        self.device = device

    def latent(self, want_s: bool = False, keep_features: Optional[bool] = None) -> _lib.Latent:
        """selfc_latent view of the buffers.  want_s: also produce InvBlockExp.s; keep_features (default: want_s, i.e. the
        callers that may run a backward): the dense feature planes are read after the call (SELFC_LAT_KEEP_FEATURES)."""
        if keep_features is None:
            keep_features = want_s
        if want_s and self.s is None:      # every element (incl. pads) is written by the coupling epilogue
            self.s = torch.empty((self.N, self.H, self.W, self.c2p), dtype=torch.float32, device=self.device)
        return _lib.Latent(self.kind, self.N, self.T, self.H, self.W, self.c1, self.c2,
                           _ptr(self.x1), _ptr(self.x2), _ptr(self.fd), _ptr(self.gd), _ptr(self.hd),
                           _ptr(self.s) if want_s else None, _ptr(self.pf), _lib.LAT_KEEP_FEATURES if keep_features else 0)

    def nbytes(self) -> int:
        ts = [self.x1, self.x2, self.fd, self.gd, self.hd] + [t for t in (self.s, self.pf) if t is not None]
        return sum(t.numel() * t.element_size() for t in ts)


_WS: Dict[Tuple, Workspace] = {}


def workspace(device, kind, N, T, H, W, c1, c2) -> Workspace:
    key = (str(device), kind, N, T, H, W, c1, c2)
    ws = _WS.get(key)
    if ws is None:
        if len(_WS) >= 4:          # a handful of resolutions at most; drop the oldest
            _WS.pop(next(iter(_WS)))
        ws = _WS[key] = Workspace(device, kind, N, T, H, W, c1, c2)
    return ws


def _conv_params(mod):
    """conv1.weight, conv1.bias, ..., conv5.weight, conv5.bias of a dense-block subnet."""
    out = []
    for i in range(1, 6):
        conv = getattr(mod, f"conv{i}")
        if conv.bias is None:
            raise NotImplementedError("selfc_amd dense-block kernels expect bias=True convs (every shipped config)")
        out += [conv.weight, conv.bias]
    return out


class PackedSubnet:
    """Kernel-layout weights of one DenseBlock / D2DTInput: views into the tensors a PackPlan produced
    (``d``: name -> tensor, names prefixed with ``prefix``).  Holds the forward fragments and, when present,
    the gradient convs of csrc/backward.hip."""

    def __init__(self, d, prefix, cin, cout, kind):
        self.cin, self.cout, self.kind = cin, cout, kind
        self.w3 = [d[f"{prefix}w3_{i}"] for i in range(4)]
        self.b3 = [d[f"{prefix}b3_{i}"] for i in range(4)]
        self.w5, self.b5 = d[f"{prefix}w5"], d[f"{prefix}b5"]
        self.wfused = d.get(f"{prefix}wfused")
        self.w5p = d.get(f"{prefix}w5p")
        self.wt5 = d.get(f"{prefix}wt5")
        self.wtd = [d.get(f"{prefix}wtd_{i}") for i in range(3)]
        self.wtx = d.get(f"{prefix}wtx")

    def struct(self) -> _lib.SubnetW:
        s = self.__dict__.get("_struct")          # the tensors of one PackedSubnet never change: build the ctypes view once
        if s is None:
            s = _lib.SubnetW()
            for i in range(4):
                s.w3[i] = _ptr(self.w3[i])
                s.b3[i] = _ptr(self.b3[i])
            s.w5 = _ptr(self.w5)
            s.b5 = _ptr(self.b5)
            s.wfused = _ptr(self.wfused)
            s.w5p = _ptr(self.w5p)
            self._struct = s
        return s

    def bwd_struct(self) -> _lib.SubnetBW:
        s = self.__dict__.get("_bwd_struct")
        if s is None:
            if self.wt5 is None:
                raise NotImplementedError("subnet backward covers channel_out <= 96")
            s = _lib.SubnetBW()
            s.wt5 = _ptr(self.wt5)
            for i in range(3):
                s.wtd[i] = _ptr(self.wtd[i])
            s.wtx = _ptr(self.wtx)
            self._bwd_struct = s
        return s


def packed_subnet(mod, virt: Tuple[int, int] = None) -> PackedSubnet:
    """Stand-alone subnet (not inside an InvBlockExp): plan learnt once per module, re-run when the weights change.
    virt = (cin_v, cout_v): pack as the equivalent block with zero-padded inputs / outputs (packing.widen_dense_params) -
    the STP chain of the codec variant keeps its 24-channel features in the kernels' 64-channel rows.  A growth gc < 32
    is widened the same way."""
    cin_v, cout_v = virt if virt is not None else (mod.channel_in, mod.channel_out)
    key = params_key(mod) + (cin_v, cout_v)
    if getattr(mod, "_pk_key", None) != key:
        mod._check()
        params = _conv_params(mod)
        temporal = mod.kind == SUBNET_D2DT
        widen = mod.gc != 32 or (cin_v, cout_v) != (mod.channel_in, mod.channel_out)
        if getattr(mod, "_plan", None) is None or mod._plan_dev != params[0].device or mod._plan_virt != (cin_v, cout_v):
            def build(ps):
                ws, bs = ps[0::2], ps[1::2]
                if widen:
                    from .packing import widen_dense_params
                    ws, bs = widen_dense_params(ws, bs, mod.channel_in, mod.channel_out, mod.gc, cin_v, cout_v)
                return subnet_pack_entries("", ws, bs, cin_v, cout_v, temporal, with_bwd=not widen)
            mod._plan = PackPlan(params, build)
            mod._plan_dev, mod._plan_virt = params[0].device, (cin_v, cout_v)
        mod._pk = PackedSubnet(mod._plan.run(params), "", cin_v, cout_v, mod.kind)
        mod._pk_key = key
    return mod._pk


#: Bumped whenever ANY module registers a parameter (`mod.weight = nn.Parameter(...)`, register_parameter, parametrize,
#: load_state_dict(assign=True) all go through nn.Module.register_parameter): plist() re-walks a module only after such an event.
_REG_EPOCH = 0


def _on_register_parameter(module, name, param):
    global _REG_EPOCH
    _REG_EPOCH += 1
    return None


torch.nn.modules.module.register_module_parameter_registration_hook(_on_register_parameter)


def plist(mod):
    """list(mod.parameters()) as ONE list object per parameter set: the cached list is kept while every Parameter object is the
    same (identity of all of them - `blk.G.conv3.weight = nn.Parameter(...)`, parametrize, load_state_dict(assign=True) replace
    objects anywhere in the module), so callers can key their own caches on the list's identity.  The walk over the module tree
    (0.7 ms for the whole net, 0.2 ms for the STP: on every module-API call it was the call's largest host cost) is only repeated
    after a parameter registration somewhere in the process (_REG_EPOCH); writing `mod._parameters[...]` directly is not seen."""
    hit = mod.__dict__.get("_plist_at")
    if hit is not None and hit[1] == _REG_EPOCH:
        return hit[0]
    cached = mod.__dict__.get("_plist")
    cur = list(torch.nn.Module.parameters(mod))             # unbound: STPNet shadows `parameters` with a tensor (as the reference does)
    if cached is None or len(cached) != len(cur) or any(a_ is not b_ for a_, b_ in zip(cached, cur)):
        cached = mod.__dict__["_plist"] = cur               # a re-registered parameter ANYWHERE in the module (not only the first one)
    mod.__dict__["_plist_at"] = (cached, _REG_EPOCH)
    return cached


#: Bumped by invalidate_weights().  Packed-weight caches key on (data_ptr, tensor._version) of every parameter, which
#: notices optimizer.step() / load_state_dict() / copy_() - but NOT writes that bypass the version counter: a replayed
#: hipGraph of a captured optimisation step (RescaleTrainer.replay) and `p.data` writes (EMA, hand-written SGD).
_WEIGHT_EPOCH = 0


def invalidate_weights() -> int:
    """Force every packed-weight cache (blocks, subnets, GlobalAgg, STP head, the nets' block arrays) to repack on its
    next use.  Call after updating parameters in a way torch's version counter does not see (`p.data.copy_()`,
    `p.data.mul_()`, a replayed graph that contains the optimizer step); RescaleTrainer.replay() does it itself."""
    global _WEIGHT_EPOCH
    _WEIGHT_EPOCH += 1
    return _WEIGHT_EPOCH


def weights_stamp(params) -> Tuple:
    """O(n) but cheap (no data_ptr calls) change detector over a fixed parameter list: weight epoch + sum of the version
    counters (+ the first tensor's address, which moves on .to() / .cuda(), + the identity of every Parameter object: a replaced
    parameter anywhere in the list - `blk.G.conv3.weight = nn.Parameter(...)` - is a different stamp).  pipeline.* compare it
    before every run."""
    return (_WEIGHT_EPOCH, sum(p._version for p in params), params[0].data_ptr() if params else 0, hash(tuple(map(id, params))))


def params_key(*mods) -> Tuple:
    return (_WEIGHT_EPOCH,) + tuple((p.data_ptr(), p._version) for m in mods for p in plist(m))


#: SELFC_KEEP_GRAPHS=1 (or rt.KEEP_GRAPHS = True): every graph of the package is created with keep_graph and its node census is appended
#: to GRAPH_LOG when its capture ends (graph_capture.__exit__) - tests ("no memset node"), bench.py (nodes per training step)
KEEP_GRAPHS = os.environ.get("SELFC_KEEP_GRAPHS") == "1"
GRAPH_LOG: list = []


def new_graph(keep: bool = False) -> "torch.cuda.CUDAGraph":
    """A fresh torch.cuda.CUDAGraph to capture into, behind a garbage collection and a device sync.  keep=True: the hipGraph_t
    outlives instantiation (torch's keep_graph), so that graph_stats() can count its nodes.

    Why: on this stack (ROCm 7.0 runtime bundled with torch 2.10) replaying a just-instantiated hipGraph crashed inside
    hip::Graph::UpdateStreams (host segfault in hipGraphLaunch) when OTHER graph execs had been destroyed between its
    instantiation and its first launch - which is what Python's cyclic garbage collector does at a random later moment to
    graphs that died inside reference cycles (optimizer <-> LR scheduler is one).  Collecting first means every dead graph is
    destroyed BEFORE the new one is instantiated; seen only in long processes (the whole GPU test suite), never in a
    fresh one.  Every capture site of the package goes through here."""
    import gc
    gc.collect()
    torch.cuda.synchronize()
    if keep or KEEP_GRAPHS:
        g = torch.cuda.CUDAGraph(keep_graph=True)
        _KEPT.add(id(g))
        return g
    return torch.cuda.CUDAGraph()


_KEPT: set = set()


def graph_stats(g) -> Dict[str, int]:
    """{"nodes", "kernel", "memset", "memcpy", "other"} of a graph captured into new_graph(keep=True) (selfc_graph_stats)."""
    counts = (C.c_longlong * 5)()
    _lib.check(_lib.lib().selfc_graph_stats(C.c_void_p(int(g.raw_cuda_graph())), counts), "selfc_graph_stats")
    return dict(zip(("nodes", "kernel", "memset", "memcpy", "other"), (int(v) for v in counts)))


#: set by _shutdown(): the package's own streams are gone (interpreter exit)
SHUT_DOWN = False

#: every live OwnStream (weak): closed by _shutdown() at interpreter exit, while the HIP runtime is still there
_LIVE_STREAMS: "weakref.WeakSet" = weakref.WeakSet()


def _shutdown():
    """atexit: destroy what must not be left to interpreter finalisation.  A hipGraph that died inside a reference cycle (a trainer:
    optimizer <-> LR scheduler) is only destroyed by the cyclic collector - if that is the FINAL collection, torch's CUDAGraph and this
    package's stream destructors run while the interpreter (and with it torch's CUDA state) is being torn down: bench.py then
    segfaulted AFTER printing its line and returning from main(), in 2 of every 14 runs (faulthandler: "Garbage-collecting, <no Python
    frame>"; profiles/r5/host_copy_repeats.txt).  atexit handlers run before that tear-down: collect now, wait for the device, close
    the package's own streams."""
    try:
        import gc
        gc.collect()
        if torch.cuda.is_available() and torch.cuda.is_initialized():
            torch.cuda.synchronize()
        # the streams that are still alive belong to objects somebody still holds (a cached ModuleGraph, a trainer, a pre-bound
        # pipeline).  atexit is LIFO: a handler registered BEFORE this package was imported runs AFTER this one - it may still call
        # the net.  From here on the module API takes its eager path on the caller's stream (pipeline.module_graph_call checks
        # SHUT_DOWN) and the multi-stream pipelines refuse loudly instead of launching on destroyed streams.
        global SHUT_DOWN
        SHUT_DOWN = True
        for s_ in list(_LIVE_STREAMS):
            s_.close()
        gc.collect()
    except Exception:      # noqa: BLE001  (never turn an exit into a traceback)
        pass


atexit.register(_shutdown)


class OwnStream:
    """A non-blocking HIP stream of the package's own (selfc_stream_create), outside torch's 32-entry round-robin pool,
    wrapped as a torch.cuda.ExternalStream (`.stream`); destroyed with `close()` / when the object dies / at interpreter exit."""

    def __init__(self, device=None):
        p = C.c_void_p()
        _lib.check(_lib.lib().selfc_stream_create(C.byref(p)), "selfc_stream_create")
        self.handle = p.value
        _LIVE_STREAMS.add(self)
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.stream = torch.cuda.ExternalStream(self.handle, device=dev)

    def close(self):
        if self.handle:
            try:
                self.stream.synchronize()
                _lib.lib().selfc_stream_destroy(C.c_void_p(self.handle))
            except Exception:      # noqa: BLE001  (interpreter shutdown)
                pass
            self.handle = None

    def __del__(self):
        self.close()


class graph_capture:
    """`with graph_capture(g, device): ...` = `with torch.cuda.graph(g): ...` on a capture stream created for THIS capture and
    destroyed behind it.

    torch.cuda.graph captures every graph of a process on ONE stream of its shared pool, and forks go to other pool streams.
    In a long process (the whole GPU test suite: ~30 captures of two- and three-branch graphs before it) the captured
    training step then segfaulted at its first replay - hip::Graph::UpdateStreams indexed past the exec's parallel streams
    (ROCm 7.0 runtime of torch 2.10; native backtrace in DESIGN.md section 4b) - although the streams, waits and events of
    the capture were call for call those of a fresh process, where it replays fine.  With a capture stream (or fork streams)
    that no earlier capture has touched the same sequence is fine too: what differs is state the runtime keeps on long-lived
    streams across captures.  So: a fresh capture stream per capture, and (`side_streams`) fresh fork streams for the
    captures that fork onto module-level streams."""

    def __init__(self, g, device=None, pool=None, fresh_side=None):
        self.g, self.device, self.pool, self.fresh_side = g, device, pool, fresh_side
        self.own, self.ctx, self.swapped = None, None, None

    def __enter__(self):
        dev = torch.device("cuda", torch.cuda.current_device()) if self.device is None else torch.device(self.device)
        with torch.cuda.device(dev):           # streams are created on the CURRENT device: make that the one they are labelled with
            self.own = OwnStream(dev)
            if self.fresh_side is not None:       # dict of module-level side streams to replace for the duration of the capture
                self.swapped = dict(self.fresh_side)
                self.side_own = {k: OwnStream(dev) for k in self.fresh_side}
                for k, o in self.side_own.items():
                    self.fresh_side[k] = o.stream
        kw = {} if self.pool is None else {"pool": self.pool}
        try:
            self.ctx = torch.cuda.graph(self.g, stream=self.own.stream, **kw)
            return self.ctx.__enter__()
        except BaseException:
            self._restore()                    # nothing was captured: undo the swap, drop the streams
            raise

    def _restore(self):
        """Put the module-level side streams back and destroy this capture's own streams - on every way out of the capture, also
        when the captured body (or capture_end) raised: the temporary streams die with this object, and a later backward must
        not launch on destroyed handles."""
        try:
            torch.cuda.synchronize()
        finally:
            try:
                if self.swapped is not None:
                    self.fresh_side.update(self.swapped)
                    self.swapped = None
                    for o in self.side_own.values():
                        o.close()
            finally:
                if self.own is not None:
                    self.own.close()
                    self.own = None

    def __exit__(self, *exc):
        try:
            r = self.ctx.__exit__(*exc)
            if exc[0] is None and id(self.g) in _KEPT:
                _KEPT.discard(id(self.g))
                GRAPH_LOG.append(graph_stats(self.g))
            return r
        finally:
            self._restore()


def warmup_stream(device=None) -> "torch.cuda.Stream":
    """A side stream for the eager warm-up run in front of a capture that is NOT the stream torch will capture on.

    torch.cuda.Stream() hands out the 32 streams of a per-device pool round-robin, and torch.cuda.graph captures on one pool
    stream it keeps for the life of the process.  In a long process (the whole GPU test suite) the warm-up stream therefore
    comes out as that very capture stream every 32nd time - and the training step captured right after such a warm-up
    segfaulted at its first replay (hip::Graph::UpdateStreams read past the exec's parallel streams; ROCm 7.0 runtime of torch
    2.10).  Warm-up work bound to the capture stream (autograd's gradient accumulators remember the stream they were created
    on) is the difference between the two cases; a distinct stream avoids it."""
    return distinct_streams(1, device)[0]


def distinct_streams(n: int, device=None, avoid=()) -> list:
    """n streams of torch's pool that are pairwise different HIP streams and different from the capture stream, the current
    stream and `avoid` (the pool is a 32-entry round robin: two torch.cuda.Stream() objects can be ONE stream - a fork onto
    "another" stream is then no fork at all, and see warmup_stream for what the capture stream itself as a side stream does)."""
    cap = torch.cuda.graphs.graph.default_capture_stream
    if cap is None:                     # torch creates it at the first capture: make that happen now, so that it can be told apart
        cap = torch.cuda.graphs.graph.default_capture_stream = torch.cuda.Stream()
    taken = {cap.cuda_stream, torch.cuda.current_stream(device).cuda_stream} | {a.cuda_stream for a in avoid}
    out = []
    for _ in range(64):
        st = torch.cuda.Stream(device=device)
        if st.cuda_stream not in taken:
            taken.add(st.cuda_stream)
            out.append(st)
            if len(out) == n:
                return out
    raise RuntimeError("torch's stream pool ran out of distinct streams")


_FN: Dict[str, object] = {}


def call(name: str, *args):
    fn = _FN.get(name)
    if fn is None:
        fn = _FN[name] = getattr(_lib.lib(), name)
    rc = fn(*args)
    if rc:
        _lib.check(rc, name)


def nchw_to_latent(x: torch.Tensor, ws: Workspace, with_fd: bool = True):
    call("selfc_nchw_to_latent", _ptr(x), _ptr(ws.x1), _ptr(ws.x2), _ptr(ws.fd) if with_fd else None, ws.FC,
         ws.N, ws.c1, ws.c2, ws.H, ws.W, _lib.stream_ptr())


def latent_to_nchw(ws: Workspace) -> torch.Tensor:
    y = torch.empty((ws.N, ws.c1 + ws.c2, ws.H, ws.W), dtype=torch.float32, device=ws.device)
    call("selfc_latent_to_nchw", _ptr(ws.x1), _ptr(ws.x2), _ptr(y), ws.N, ws.c1, ws.c2, ws.H, ws.W, _lib.stream_ptr())
    return y


def s_to_nchw(ws: Workspace) -> torch.Tensor:
    """InvBlockExp.s (N,c2,H,W) from the kernel's NHWC s buffer."""
    y = torch.empty((ws.N, ws.c2, ws.H, ws.W), dtype=torch.float32, device=ws.device)
    call("selfc_nhwc4_to_nchw", _ptr(ws.s), _ptr(y), ws.N, ws.c2, ws.H, ws.W, _lib.stream_ptr())
    return y


def as_input(x: torch.Tensor) -> torch.Tensor:
    """Boundary contract: NCHW fp32, possibly a non-contiguous view."""
    _lib.require_gpu(x)
    if x.dtype != torch.float32:
        raise TypeError(f"selfc_amd expects float32 NCHW tensors at the module boundary, got {x.dtype}")
    return x.contiguous()


class PackedBlock:
    """Kernel-layout weights of one InvBlockExp (F, G, H) as a selfc_invblock_w; G's temporal conv5 is packed
    together with H's (one launch computes both)."""

    def __init__(self, blk, d):
        self.F = PackedSubnet(d, "F.", blk.F.channel_in, blk.F.channel_out, blk.F.kind)
        self.G = PackedSubnet(d, "G.", blk.G.channel_in, blk.G.channel_out, blk.G.kind)
        self.H = PackedSubnet(d, "H.", blk.H.channel_in, blk.H.channel_out, blk.H.kind)
        self.clamp = float(blk.clamp)

    def struct(self) -> _lib.InvBlockW:
        s = self.__dict__.get("_struct")
        if s is None:
            s = _lib.InvBlockW()
            s.F, s.G, s.H = self.F.struct(), self.G.struct(), self.H.struct()
            s.clamp = self.clamp
            self._struct = s
        return s


def _block_entries(blk, ps):
    f, g, h = ps[0:10], ps[10:20], ps[20:30]
    temporal = blk.F.kind == SUBNET_D2DT
    e = subnet_pack_entries("F.", f[0::2], f[1::2], blk.F.channel_in, blk.F.channel_out, temporal)
    e.update(subnet_pack_entries("G.", g[0::2], g[1::2], blk.G.channel_in, blk.G.channel_out, temporal,
                                 partner_w5=h[8] if temporal else None))
    e.update(subnet_pack_entries("H.", h[0::2], h[1::2], blk.H.channel_in, blk.H.channel_out, temporal))
    return e


def packed_block(blk) -> PackedBlock:
    key = params_key(blk) + (float(blk.clamp),)
    if getattr(blk, "_pb_key", None) != key:
        for sub in (blk.F, blk.G, blk.H):
            sub._check()
        if blk.split_len1 > 3:
            raise NotImplementedError("the fused block kernels cover channel_split_num <= 3; wider splits run composed (InvBlockExp._forward_composed)")
        params = _conv_params(blk.F) + _conv_params(blk.G) + _conv_params(blk.H)
        if getattr(blk, "_plan", None) is None or blk._plan_dev != params[0].device:
            blk._plan = PackPlan(params, lambda ps: _block_entries(blk, ps))
            blk._plan_dev = params[0].device
        blk._pb = PackedBlock(blk, blk._plan.run(params))
        blk._pb_key = key
    return blk._pb


class PackGroup:
    """The kernel-layout weights of MANY modules as ONE gather (one PackPlan over all their parameters).

    Every module's own cache (`packed_block`, `packed_subnet`, GlobalAgg._packed, STPNet._tail_packed) repacks when its
    parameters' (address, version) key changes - in training that is every step, and module by module it costs ~250 small
    device launches (pad / copy / cast per tensor; 3 per planned module).  A group learns one plan over all members, and
    `refresh()` - three device ops whatever the number of members - installs each member's tensors together with the key
    its own cache checks, so the per-module calls that follow are cache hits.  A member is (params, build(ps) -> {name:
    (tensor, kind)}, install(dict)); build must be composed of packing.pack_* functions (pure gathers)."""

    def __init__(self):
        self.members, self.plan, self.dev = [], None, None

    def add(self, params, build, install):
        self.members.append((list(params), build, install))
        self.plan = None

    def refresh(self):
        if not self.members:
            return
        allp = [p_ for ps, _, _ in self.members for p_ in ps]
        dev = allp[0].device
        if self.plan is None or self.dev != dev:
            def build_all(ps):
                e, lo = {}, 0
                for i, (mp, build, _) in enumerate(self.members):
                    for k, v in build(ps[lo:lo + len(mp)]).items():
                        e[f"{i}|{k}"] = v
                    lo += len(mp)
                return e
            self.plan, self.dev = PackPlan(allp, build_all), dev
            self.names = [[] for _ in self.members]
            for name, *_ in self.plan.items:
                i, short = name.split("|", 1)
                self.names[int(i)].append((name, short))
        d = self.plan.run(allp)
        for (_, _, install), names in zip(self.members, self.names):
            install({short: d[full] for full, short in names})


def group_add_block(group: PackGroup, blk):
    """InvBlockExp member: what packed_block() would build, installed under the key packed_block() checks."""
    if blk.split_len1 > 3:
        return
    for sub in (blk.F, blk.G, blk.H):
        sub._check()

    def install(d):
        blk._pb = PackedBlock(blk, d)
        blk._pb_key = params_key(blk) + (float(blk.clamp),)
    group.add(_conv_params(blk.F) + _conv_params(blk.G) + _conv_params(blk.H), lambda ps: _block_entries(blk, ps), install)


def group_add_subnet(group: PackGroup, mod, virt: Tuple[int, int] = None):
    """stand-alone DenseBlock / D2DTInput member (packed_subnet's tensors and key)"""
    cin_v, cout_v = virt if virt is not None else (mod.channel_in, mod.channel_out)
    mod._check()
    temporal = mod.kind == SUBNET_D2DT
    widen = mod.gc != 32 or (cin_v, cout_v) != (mod.channel_in, mod.channel_out)

    def build(ps):
        ws, bs = ps[0::2], ps[1::2]
        if widen:
            from .packing import widen_dense_params
            ws, bs = widen_dense_params(ws, bs, mod.channel_in, mod.channel_out, mod.gc, cin_v, cout_v)
        return subnet_pack_entries("", ws, bs, cin_v, cout_v, temporal, with_bwd=not widen)

    def install(d):
        mod._pk = PackedSubnet(d, "", cin_v, cout_v, mod.kind)
        mod._pk_key = params_key(mod) + (cin_v, cout_v)
    group.add(_conv_params(mod), build, install)


def pack_group_for(net) -> PackGroup:
    """A PackGroup over everything `net` repacks when its weights change: its InvBlockExp blocks (`net._blocks()`) and, when it has
    one that can join, its STP net (`stp_net.add_to_pack_group`).  A training loop calls `.refresh()` once per step, before the
    forward (RescaleTrainer does); modules that are not members keep repacking themselves on first use."""
    grp = PackGroup()
    for blk in (net._blocks() if hasattr(net, "_blocks") else []):
        group_add_block(grp, blk)
    stp = getattr(net, "stp_net", None)
    if stp is not None and hasattr(stp, "add_to_pack_group"):
        stp.add_to_pack_group(grp)
    return grp


def block_array(blocks):
    """(selfc_invblock_w[n], keep-alive list) for selfc_invstack_run."""
    packs = [packed_block(b) for b in blocks]
    arr = (_lib.InvBlockW * len(packs))(*[p.struct() for p in packs])
    return arr, packs
