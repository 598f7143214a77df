"""Weight-init helpers with the reference's semantics (codes/models/modules/module_util.py:7-44).

Only ``nn.Conv2d`` / ``nn.Linear`` / ``nn.BatchNorm2d`` are touched: ``nn.Conv3d``
layers keep PyTorch's default init (SURVEY trap 4), which is why D2DTInput's
"INN_init" leaves conv5 non-zero while DenseBlock's conv5 starts at zero.
"""
import torch
import torch.nn as nn
import torch.nn.init as init


def _apply(net_l, fn, scale):
    nets = net_l if isinstance(net_l, list) else [net_l]
    for net in nets:
        for m in net.modules():
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                fn(m.weight)
                m.weight.data *= scale
                if m.bias is not None:
                    m.bias.data.zero_()
            elif isinstance(m, nn.BatchNorm2d):
                init.constant_(m.weight, 1)
                init.constant_(m.bias.data, 0.0)


def initialize_weights(net_l, scale=1):
    _apply(net_l, lambda w: init.kaiming_normal_(w, a=0, mode="fan_in"), scale)


def initialize_weights_xavier(net_l, scale=1):
    _apply(net_l, init.xavier_normal_, scale)


_MODULE_BASE_KEYS = frozenset(nn.Module().__dict__)


def cache_free_state(self):
    """``__getstate__`` of the HIP-backed modules: copy.deepcopy(net), torch.save(net) and a spawn hand-off take the module's
    registered state (parameters, buffers, sub-modules, public attributes) and leave behind everything the runtime cached on
    it under a private name - packed weights, ctypes views of them, workspaces, gather plans (ctypes pointers and hipGraphs
    cannot be pickled, and the copies would alias the original's device buffers).  Every such cache is rebuilt on first use.
    ``InvBlockExp.s`` (public, reference-visible state) is materialised before its lazy view is dropped."""
    d = self.__dict__
    if d.get("_s_ws") is not None:
        _ = self.s                           # NCHW tensor from the kernel's buffer (InvBlockExp)
    keep = {k: v for k, v in d.items() if not k.startswith("_") or k in _MODULE_BASE_KEYS}
    if d.get("_s_nchw") is not None:
        keep["_s_nchw"] = d["_s_nchw"]
    return keep


class HeadOutput(torch.Tensor):
    """The tensor an STP net publishes as ``self.parameters`` (the reference assigns the raw head output to that name,
    SelfC_GMM_arch_inv.py:377 / SelfC_arch_inv.py:149,153, shadowing ``nn.Module.parameters`` on the instance).  Reading it
    behaves as the reference's tensor does (shape, indexing, arithmetic - results are plain tensors, autograd intact); CALLING
    it still yields the module's parameters, so ``Adam(stp_net.parameters())``, ``stp_net.zero_grad()`` or wrapping the
    sub-module in DistributedDataParallel keep working after the first forward, and ``copy.deepcopy(net)`` copies it detached
    instead of failing on a non-leaf tensor."""

    __torch_function__ = torch._C._disabled_torch_function_impl

    @staticmethod
    def wrap(t: torch.Tensor, owner: nn.Module) -> "HeadOutput":
        import weakref
        out = t.as_subclass(HeadOutput)
        out._owner = weakref.ref(owner)
        return out

    def __call__(self, recurse: bool = True):
        owner = self._owner() if getattr(self, "_owner", None) is not None else None
        if owner is None:
            raise TypeError("this head output is no longer attached to its STP module")
        return nn.Module.parameters(owner, recurse)

    def __deepcopy__(self, memo):
        return self.detach().clone().as_subclass(torch.Tensor)

    def __reduce_ex__(self, protocol):
        # pickling a module after its first forward (torch.save(net), a spawn-based DataLoader / multiprocessing hand-off): the
        # weak reference to the owner cannot be pickled and the non-leaf tensor carries its graph - hand over a detached plain
        # tensor instead (the next forward publishes a fresh head output)
        return self.detach().as_subclass(torch.Tensor).__reduce_ex__(protocol)
