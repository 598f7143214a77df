"""The metric's unit of work as one pre-bound launch sequence.

``RescaleRoundTrip`` runs, for a batch of septuplets already resident in HBM,
FrequencyAnalyzer.fwd -> N x InvBlockExp.fwd -> Quantization (LR channels) ->
N x InvBlockExp.rev -> FrequencyAnalyzer.rev, i.e. SelfCModel.test()'s two netG
calls (SelfC_model.py:213-230) with the forward's own HF channels fed back
(the STP sampler is a separate, later stage).  Everything stays
in the kernels' latent layout; no allocation happens inside ``run`` so it can be
captured into a hipGraph (``capture()``).
"""
from __future__ import annotations

import ctypes as C
import os
import weakref

import torch

from . import _lib, runtime as rt
from .global_var import GlobalVar


class RescaleRoundTrip:
    def __init__(self, net, n_frames: int, H: int, W: int, device):
        t = GlobalVar.get_Temporal_LEN()
        if not t or n_frames % t:
            raise RuntimeError("set GlobalVar temporal length to a divisor of the frame count first")
        self.net = net
        self.k = net.operations[0].k
        self.N, self.H, self.W = n_frames, H, W
        self.h, self.w = H // self.k, W // self.k
        blk = net._blocks()[0]
        self.ws = rt.Workspace(device, blk.F.kind, n_frames, t, self.h, self.w, blk.split_len1, blk.split_len2)
        self._bind()
        self.lat = self.ws.latent()
        self.out = torch.empty((n_frames, 3, H, W), dtype=torch.float32, device=device)
        self.graph = None
        self._graph_stamp = None
        self.static_x = None

    def _extra_params(self):
        return []

    @property
    def _params(self):
        """The net's CURRENT Parameter objects, re-derived per check (rt.plist notices re-registered parameters -
        load_state_dict(assign=True), parametrize, `block.F.conv1.weight = nn.Parameter(...)` - by identity): a list taken once
        at construction would keep stamping objects the net no longer uses and never see such a change."""
        return [p for b in self.net._blocks() for p in rt.plist(b)] + self._extra_params()

    def _bind(self):
        """(Re)pack the blocks' weights and remember which weights that was."""
        self.arr, self.keep = rt.block_array(self.net._blocks())
        self.nblk = len(self.keep)
        self._stamp = rt.weights_stamp(self._params)

    def _fresh(self, replaying: bool = False):
        """Weights may have changed since __init__ / capture() (load_state_dict, an optimizer step): an eager run repacks
        them; a captured graph has the OLD packed buffers baked in, so replaying it is refused."""
        now = rt.weights_stamp(self._params)
        if replaying:
            if now != self._graph_stamp:
                raise RuntimeError("the net's weights changed after capture(): call capture() again (the graph holds the old packed weights)")
        elif now != self._stamp:
            self._bind()

    def run(self, x: torch.Tensor) -> torch.Tensor:
        self._fresh()
        ws, L, sp = self.ws, _lib.lib(), _lib.stream_ptr()
        chk = _lib.check
        chk(L.selfc_freq_fwd(x.data_ptr(), ws.x1.data_ptr(), ws.x2.data_ptr(), ws.fd.data_ptr(), ws.FC,
                             self.N, self.H, self.W, self.k, sp), "selfc_freq_fwd")
        chk(L.selfc_invstack_run(self.arr, self.nblk, C.byref(self.lat), 0, sp), "selfc_invstack_run fwd")
        chk(L.selfc_quantize_inplace(ws.x1.data_ptr(), ws.x1.numel(), sp), "selfc_quantize_inplace")
        chk(L.selfc_invstack_run(self.arr, self.nblk, C.byref(self.lat), 1, sp), "selfc_invstack_run rev")
        chk(L.selfc_freq_inv(ws.x1.data_ptr(), ws.x2.data_ptr(), self.out.data_ptr(), self.N, self.h, self.w, self.k, sp),
            "selfc_freq_inv")
        return self.out

    def forward_latent(self, x: torch.Tensor) -> torch.Tensor:
        """fwd half only; returns the (N,51,h,w) NCHW latent (for parity checks)."""
        self._fresh()
        ws, L, sp = self.ws, _lib.lib(), _lib.stream_ptr()
        _lib.check(L.selfc_freq_fwd(x.data_ptr(), ws.x1.data_ptr(), ws.x2.data_ptr(), ws.fd.data_ptr(), ws.FC,
                                    self.N, self.H, self.W, self.k, sp), "selfc_freq_fwd")
        _lib.check(L.selfc_invstack_run(self.arr, self.nblk, C.byref(self.lat), 0, sp), "selfc_invstack_run fwd")
        return rt.latent_to_nchw(ws)

    def inverse_latent(self, z: torch.Tensor) -> torch.Tensor:
        """rev half only on a given (N,51,h,w) NCHW latent; returns the (N,3,H,W) reconstruction."""
        self._fresh()
        ws, L, sp = self.ws, _lib.lib(), _lib.stream_ptr()
        rt.nchw_to_latent(z.contiguous(), ws, with_fd=False)
        _lib.check(L.selfc_invstack_run(self.arr, self.nblk, C.byref(self.lat), 1, sp), "selfc_invstack_run rev")
        _lib.check(L.selfc_freq_inv(ws.x1.data_ptr(), ws.x2.data_ptr(), self.out.data_ptr(), self.N, self.h, self.w, self.k, sp),
                   "selfc_freq_inv")
        return self.out

    def capture(self, x: torch.Tensor):
        """Record one run into a hipGraph (replay with ``replay()``); x must stay at this address."""
        self.static_x = x
        s = rt.warmup_stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self.run(x)                    # warm-up outside capture (lazy function attributes etc.)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.graph = None                  # a stale exec dies BEFORE the new one is instantiated (rt.new_graph's docstring)
        g = rt.new_graph()
        with rt.graph_capture(g, x.device):
            self.run(x)
        self.graph = g
        self._graph_stamp = self._stamp

    def replay(self):
        self._fresh(replaying=True)
        self.graph.replay()
        return self.out


class FullTestPath(RescaleRoundTrip):
    """SelfCModel.test()'s two netG calls as the reference runs them (SelfC_model.py:213-230): forward stack, Quantization
    of the LR frames, STP prediction of the HF channels from the quantised LR (fh_loss gmm: a fresh sample per call), reverse
    stack.  Same latent-layout pipeline as RescaleRoundTrip, plus the STP between the halves; capturable."""

    def __init__(self, net, n_frames: int, H: int, W: int, device):
        super().__init__(net, n_frames, H, W, device)
        stp = net.stp_net
        self.stp_scratch = {}
        self.eps = None
        if stp.fh_loss != "l2":
            self.eps = torch.empty((n_frames * self.h * self.w, stp.hf_dim * stp.K), dtype=torch.float32, device=device)
        self.lr = torch.empty((n_frames, 3, self.h, self.w), dtype=torch.float32, device=device)

    def _extra_params(self):
        return rt.plist(self.net.stp_net)      # the STP repacks itself per call (params_key); the stamp guards replay()

    def run(self, x: torch.Tensor) -> torch.Tensor:
        self._fresh()
        ws, L, sp = self.ws, _lib.lib(), _lib.stream_ptr()
        chk = _lib.check
        chk(L.selfc_freq_fwd(x.data_ptr(), ws.x1.data_ptr(), ws.x2.data_ptr(), ws.fd.data_ptr(), ws.FC,
                             self.N, self.H, self.W, self.k, sp), "selfc_freq_fwd")
        chk(L.selfc_invstack_run(self.arr, self.nblk, C.byref(self.lat), 0, sp), "selfc_invstack_run fwd")
        chk(L.selfc_quantize_inplace(ws.x1.data_ptr(), ws.x1.numel(), sp), "selfc_quantize_inplace")
        chk(L.selfc_nhwc4_to_nchw(ws.x1.data_ptr(), self.lr.data_ptr(), self.N, 3, self.h, self.w, sp), "selfc_nhwc4_to_nchw")   # forw_L
        self.net.stp_net.run_nhwc(ws.x1, ws.x2, self.N, ws.T, self.h, self.w, scratch=self.stp_scratch, eps=self.eps)
        chk(L.selfc_invstack_run(self.arr, self.nblk, C.byref(self.lat), 1, sp), "selfc_invstack_run rev")
        chk(L.selfc_freq_inv(ws.x1.data_ptr(), ws.x2.data_ptr(), self.out.data_ptr(), self.N, self.h, self.w, self.k, sp),
            "selfc_freq_inv")
        return self.out


class MultiStreamRoundTrip:
    """The same unit of work, with the batch of septuplets split over `nstreams` HIP
    streams (whole clips per stream - they are independent).  Each conv launch is
    short (one to three waves of workgroups), so its prologue / epilogue latency is
    exposed; kernels of different streams overlap and fill those gaps.  Fork/join is
    expressed with stream events so the whole step still captures into one hipGraph."""

    def __init__(self, net, n_frames: int, H: int, W: int, device, nstreams: int = 2, part_cls=None):
        t = GlobalVar.get_Temporal_LEN()
        clips = n_frames // t
        if clips % nstreams:
            raise RuntimeError(f"{clips} clips do not split evenly over {nstreams} streams")
        self.nstreams = nstreams
        self.per = n_frames // nstreams
        part_cls = part_cls or RescaleRoundTrip          # e.g. FullTestPath
        self.parts = [part_cls(net, self.per, H, W, device) for _ in range(nstreams)]
        self._own_streams = [rt.OwnStream(device) for _ in range(nstreams)]      # the package's own HIP streams (runtime.graph_capture)
        self.streams = [o.stream for o in self._own_streams]
        self.out = torch.empty((n_frames, 3, H, W), dtype=torch.float32, device=device)
        for i, p in enumerate(self.parts):           # parts write straight into slices of one output
            p.out = self.out[i * self.per:(i + 1) * self.per]
        self.graph, self.graphs = None, None

    def _alive(self):
        if rt.SHUT_DOWN:
            raise RuntimeError("selfc_amd has shut down (interpreter exit: its own HIP streams are destroyed, runtime._shutdown) - "
                               "a multi-stream pipeline cannot run from an atexit handler registered before the package was imported")

    def run(self, x: torch.Tensor) -> torch.Tensor:
        self._alive()
        cur = torch.cuda.current_stream()
        for i, (p, st) in enumerate(zip(self.parts, self.streams)):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                p.run(x[i * self.per:(i + 1) * self.per])
        for st in self.streams:
            cur.wait_stream(st)
        return self.out

    def capture(self, x: torch.Tensor, per_stream: bool = False):
        """One hipGraph with a branch per stream (default), or - per_stream - one hipGraph per stream, each replayed on its own
        stream between an event fork and join.  The second form does not depend on how the runtime maps the branches of ONE
        graph onto hardware queues (on some boxes of the pool a two-branch graph runs at the one-stream rate); bench.py times
        both before its timed region and keeps the faster."""
        self.run(x)
        torch.cuda.synchronize()
        self.graph, self.graphs = None, None
        if per_stream:
            graphs = []
            for i, p in enumerate(self.parts):
                g = rt.new_graph()
                with rt.graph_capture(g, x.device):
                    p.run(x[i * self.per:(i + 1) * self.per])
                graphs.append(g)
            self.graphs = graphs
        else:
            g = rt.new_graph()
            with rt.graph_capture(g, x.device):
                self.run(x)
            self.graph = g
        for p in self.parts:
            p._graph_stamp = p._stamp

    def replay(self):
        self._alive()
        self.parts[0]._fresh(replaying=True)
        if self.graphs is not None:
            cur = torch.cuda.current_stream()
            for g, st in zip(self.graphs, self.streams):
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    g.replay()
            for st in self.streams:
                cur.wait_stream(st)
        else:
            self.graph.replay()
        return self.out


# ----------------------------------------------------------------------------------------------------------------------
# The drop-in call as the fast call.  The reference's callers only ever do `netG(x=..., rev=...)` (SelfC_model.py:213-230,
# :141,153); in eval / no_grad `SelfCInvNet.forward` routes through a cached ModuleGraph: the call's block stack (and, on
# the reverse, the STP chain + sampler) replayed as ONE hipGraph with the clips split over two HIP streams - the launch
# configuration of the headline.  The kernels that touch the CALLER's tensors (the split of x on the way in, the NCHW
# conversions / the merge on the way out) are part of the graph too and address those tensors through pointer slots
# (include/selfc_hip.h, abi 10: selfc_set_pointers + the *_ind transforms): one tiny launch in front of the replay stores
# this call's input address and the addresses of its FRESH output tensors.  Nothing the caller holds is ever overwritten by
# a later call, and no copy is added.
# ----------------------------------------------------------------------------------------------------------------------
class ModuleGraph:
    """One (mode, shape) instance of SelfCInvNet's inference call.

      mode 'fwd'    : x (N,3,H,W)                -> (N,3+c2,h,w)                 forward(x, rev=False)
      mode 'rev'    : LR (N,>=3,h,w)             -> (N,3,H,W), recon_hf (N,c2,h,w)   forward(x, rev=True): STP sample + reversed stack
      mode 'revlat' : latent (N,3+c2,h,w)        -> (N,3,H,W)                    inverse_from_latent(z): reversed stack, STP bypassed
    """

    def __init__(self, net, mode: str, n: int, h: int, w: int, device, nstreams: int):
        t = GlobalVar.get_Temporal_LEN()
        self._net = weakref.ref(net)       # the cache is keyed weakly on the net: an instance must not keep it alive
        self.mode, self.N, self.h, self.w, self.T = mode, n, h, w, t
        self.k = net.operations[0].k
        self.H, self.W = h * self.k, w * self.k
        self.device = device
        self.nstreams = nstreams
        self.per = n // nstreams
        blk = net._blocks()[0]
        self.c1, self.c2 = blk.split_len1, blk.split_len2
        # x1 / x2 of the parts are slices of ONE whole-batch buffer each (the latent rows are frame-major); the plane-blocked
        # dense buffers are per part
        per = self.per
        self.ws = [rt.Workspace(device, blk.F.kind, per, t, h, w, self.c1, self.c2) for _ in range(nstreams)]
        c2p = self.ws[0].c2p
        self.X1 = torch.zeros((n, h, w, 4), dtype=torch.float32, device=device)
        self.X2 = torch.zeros((n, h, w, c2p), dtype=torch.float32, device=device)
        for i, ws in enumerate(self.ws):
            ws.x1, ws.x2 = self.X1[i * per:(i + 1) * per], self.X2[i * per:(i + 1) * per]
        self.lat = [ws.latent() for ws in self.ws]
        self._own_streams = [rt.OwnStream(device) for _ in range(nstreams)] if nstreams > 1 else []
        self.streams = [o.stream for o in self._own_streams]
        if mode == "rev":
            stp = net.stp_net
            self.stp_scratch = [{} for _ in range(nstreams)]
            self.eps = [torch.empty((per * h * w, stp.hf_dim * stp.K), dtype=torch.float32, device=device)
                        if stp.fh_loss != "l2" else None for _ in range(nstreams)]
            # the STP's sample goes to its own buffer and stays there for `recon_hf`: the first block of the reversed stack reads
            # x2 from it and writes its y2 into the latent x2 (selfc_latent.x2_out), the other blocks work in place
            self.HF = torch.zeros((n, h, w, c2p), dtype=torch.float32, device=device)
            self.hf = [self.HF[i * per:(i + 1) * per] for i in range(nstreams)]
            self.lat_first = [_lib.Latent(blk.F.kind, per, t, h, w, self.c1, self.c2, ws.x1.data_ptr(), hf.data_ptr(), ws.fd.data_ptr(),
                                          ws.gd.data_ptr(), ws.hd.data_ptr(), None, None if ws.pf is None else ws.pf.data_ptr(), 0, None,
                                          None, ws.x2.data_ptr()) for ws, hf in zip(self.ws, self.hf)]
        self.slots = torch.zeros(4, dtype=torch.int64, device=device)       # pointer slots the graph's *_ind transforms read
        self.graph = None
        self.stamp = None
        self.calls = 0
        self._last_x = None

    @property
    def net(self):
        return self._net()

    @property
    def _params(self):
        """The Parameter objects the net holds NOW (per call, as the eager path derives them: rt.plist is keyed on identity) -
        parameters re-registered on the net (load_state_dict(assign=True), parametrize) must change the stamp and re-capture."""
        net = self.net
        ps = [p for b in net._blocks() for p in rt.plist(b)]
        return ps + rt.plist(net.stp_net) if self.mode == "rev" else ps

    # -- the captured call ------------------------------------------------------------------------------------------------
    # slots: 0 = the caller's input tensor, 1 = the first output, 2 = recon_hf (mode 'rev')
    def _part(self, i: int):
        ws, sp, per = self.ws[i], _lib.stream_ptr(), self.per
        L, chk = _lib.lib(), _lib.check
        h, w, k = self.h, self.w, self.k
        slot = lambda j: C.c_void_p(self.slots.data_ptr() + 8 * j)      # noqa: E731
        if self.mode == "fwd":
            chk(L.selfc_freq_fwd_ind(slot(0), i * per * 3 * self.H * self.W, ws.x1.data_ptr(), ws.x2.data_ptr(), ws.fd.data_ptr(), ws.FC,
                                     per, self.H, self.W, k, sp), "selfc_freq_fwd_ind")
            chk(L.selfc_invstack_run(self.arr, self.nblk, C.byref(self.lat[i]), 0, sp), "selfc_invstack_run fwd")
            chk(L.selfc_latent_to_nchw_ind(ws.x1.data_ptr(), ws.x2.data_ptr(), slot(1), i * per * (self.c1 + self.c2) * h * w,
                                           per, self.c1, self.c2, h, w, sp), "selfc_latent_to_nchw_ind")
            return
        if self.mode == "rev":
            chk(L.selfc_nchw_to_nhwc4_ind(slot(0), i * per * 3 * h * w, ws.x1.data_ptr(), per, 3, h, w, sp), "selfc_nchw_to_nhwc4_ind")
            self.net.stp_net.run_nhwc(ws.x1, self.hf[i], per, self.T, h, w, scratch=self.stp_scratch[i], eps=self.eps[i])
            chk(L.selfc_nhwc4_to_nchw_ind(self.hf[i].data_ptr(), slot(2), i * per * self.c2 * h * w, per, self.c2, h, w, sp), "selfc_nhwc4_to_nchw_ind")
            chk(L.selfc_invblock_run(C.byref(self.arr[self.nblk - 1]), C.byref(self.lat_first[i]), 1, sp), "selfc_invblock_run rev (first)")
            chk(L.selfc_invstack_run(self.arr, self.nblk - 1, C.byref(self.lat[i]), 1, sp), "selfc_invstack_run rev")
        else:
            chk(L.selfc_nchw_to_latent_ind(slot(0), i * per * (self.c1 + self.c2) * h * w, ws.x1.data_ptr(), ws.x2.data_ptr(), None, ws.FC,
                                           per, self.c1, self.c2, h, w, sp), "selfc_nchw_to_latent_ind")
            chk(L.selfc_invstack_run(self.arr, self.nblk, C.byref(self.lat[i]), 1, sp), "selfc_invstack_run rev")
        chk(L.selfc_freq_inv_ind(ws.x1.data_ptr(), ws.x2.data_ptr(), slot(1), i * per * 3 * self.H * self.W, per, h, w, k, sp), "selfc_freq_inv_ind")

    def _run_parts(self):
        if self.nstreams == 1:
            self._part(0)
            return
        cur = torch.cuda.current_stream()
        for i, st in enumerate(self.streams):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                self._part(i)
        for st in self.streams:
            cur.wait_stream(st)

    def _io(self, x: torch.Tensor):
        """This call's fresh output tensors, and their addresses (with x's) into the pointer slots - on the current stream, i.e.
        ordered behind the previous replay's reads and in front of the next one's."""
        dev, n = self.device, self.N
        if self.mode == "fwd":
            outs = (torch.empty((n, self.c1 + self.c2, self.h, self.w), dtype=torch.float32, device=dev),)
        elif self.mode == "rev":
            outs = (torch.empty((n, 3, self.H, self.W), dtype=torch.float32, device=dev),
                    torch.empty((n, self.c2, self.h, self.w), dtype=torch.float32, device=dev))
        else:
            outs = (torch.empty((n, 3, self.H, self.W), dtype=torch.float32, device=dev),)
        _lib.check(_lib.lib().selfc_set_pointers(self.slots.data_ptr(), 1 + len(outs), x.data_ptr(), outs[0].data_ptr(),
                                                 outs[1].data_ptr() if len(outs) > 1 else None, None, _lib.stream_ptr()), "selfc_set_pointers")
        return outs

    def _capture(self, x: torch.Tensor):
        self.arr, self.keep = rt.block_array(self.net._blocks())
        self.nblk = len(self.keep)
        s = rt.warmup_stream(self.device)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            warm = self._io(x)             # warm-up outside capture: lazy packs, scratch allocation, function attributes
            self._run_parts()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize(self.device)
        del warm
        self.graph = None                  # a stale exec dies BEFORE the new one is instantiated (rt.new_graph's docstring)
        g = rt.new_graph()
        with rt.graph_capture(g, self.device):
            self._run_parts()
        self.graph = g
        self.stamp = rt.weights_stamp(self._params)

    # -- one call ---------------------------------------------------------------------------------------------------------
    def __call__(self, x: torch.Tensor):
        if self.graph is None or rt.weights_stamp(self._params) != self.stamp:
            self._capture(x)               # first use, or the weights changed: the graph holds the old packed buffers
        outs = self._io(x)
        self.graph.replay()
        # x must outlive the replay's reads: the caller's reference may die right after this returns (e.g. the contiguous copy of
        # a sliced view made at the boundary), and the caching allocator could hand its memory to a tensor written on ANOTHER
        # stream; on the calling stream itself reuse is ordered behind the replay.  Outputs are ordinary tensors of this stream.
        self._last_x = x
        return outs[0] if len(outs) == 1 else outs

    def nbytes(self) -> int:
        return sum(ws.nbytes() for ws in self.ws) + (self.HF.numel() * 4 if self.mode == "rev" else 0)


#: net -> {(mode, shape, ...): 1 (seen once) | ModuleGraph}; weak on the net, so deepcopy / pickling / deletion of a net never
#: meets a hipGraph
_MODULE_GRAPHS: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()
#: SELFC_MODULE_GRAPH=0 keeps the module API on its eager single-stream path (debugging, profiling of single launches)
MODULE_GRAPH = os.environ.get("SELFC_MODULE_GRAPH", "1") != "0"
#: at most this many bytes of cached workspaces per net (least recently used instances are dropped first)
MODULE_GRAPH_BYTES = int(float(os.environ.get("SELFC_MODULE_GRAPH_GB", "8")) * 2 ** 30)


def module_graph(net, mode: str, n: int, h: int, w: int, device):
    """The cached ModuleGraph of this call, or None when the call has to run eagerly: fast path switched off, a capture by
    somebody else in progress on this stream (pipeline.*, RescaleTrainer.capture), injected STP noise (`stp_net.eps`, a
    host-side test hook), a clip length that does not divide the batch, or the FIRST call of a shape - a shape is captured
    on its second call, so one-off calls do not pay for a capture."""
    if not MODULE_GRAPH or rt.SHUT_DOWN or torch.cuda.is_current_stream_capturing():
        return None
    t = GlobalVar.get_Temporal_LEN()
    if not t or n % t:
        return None
    if mode == "rev" and getattr(net.stp_net, "eps", None) is not None:
        return None
    blocks = net._blocks()
    if not blocks or any(b.split_len1 > 3 for b in blocks) or len(blocks) != len(net.operations) - 1:
        return None
    cache = _MODULE_GRAPHS.get(net)
    if cache is None:
        cache = _MODULE_GRAPHS[net] = {}
    key = (mode, n, h, w, t, str(device), getattr(net.stp_net, "fh_loss", None) if mode == "rev" else None)
    ent = cache.get(key)
    if ent is None:
        cache[key] = 1                    # seen once: next time it is captured
        return None
    if ent == 1:
        clips = n // t
        ent = ModuleGraph(net, mode, n, h, w, device, 2 if clips % 2 == 0 else 1)
        cache[key] = ent
        live = [(k_, v) for k_, v in cache.items() if isinstance(v, ModuleGraph)]
        total = sum(v.nbytes() for _, v in live)
        for k_, v in sorted(live, key=lambda kv: kv[1].calls):         # over budget: drop the least used instances
            if total <= MODULE_GRAPH_BYTES or v is ent:
                continue
            total -= v.nbytes()
            del cache[k_]
    ent.calls += 1
    return ent
