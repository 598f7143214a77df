"""8-bit rounding between the forward and the reverse pass (codes/models/modules/Quantization.py:4-26).

``Quantization()(x)`` clamps to [0, 1] and rounds to the 1/255 grid on the device (``selfc_quantize_inplace``); the
gradient is the identity (straight-through), as in the reference.  ``quant_v`` / ``is_clip`` stay class-level settings
shared by every instance, because that is how the reference stores them."""
import torch
import torch.nn as nn

from .. import _lib, runtime as rt


def _round_to_grid(values: torch.Tensor, private: bool = False) -> torch.Tensor:
    """A quantised copy of `values` (any shape, fp32, on the GPU).  The kernel works on whole float4 groups, so a length
    that is not a multiple of 4 goes through a zero-padded staging buffer.  private: `values` already is a copy nobody else
    holds (the contiguous copy of a sliced view, e.g. out[:, :3]) and is rounded in place."""
    count = values.numel()
    padded = (count + 3) // 4 * 4
    if padded == count:
        work = values if private else values.clone()
    else:
        work = values.new_zeros(padded)
        work[:count].copy_(values.reshape(-1))
    rt.call("selfc_quantize_inplace_v", work.data_ptr(), padded, float(Quantization.quant_v), 1 if Quantization.is_clip else 0, _lib.stream_ptr())
    return work if padded == count else work[:count].reshape(values.shape)


class Quant(torch.autograd.Function):
    """Straight-through estimator: forward = the rounding kernel, backward passes the gradient unchanged (:15-17)."""

    @staticmethod
    def forward(ctx, input):
        dense = rt.as_input(input)
        return _round_to_grid(dense, private=dense is not input and dense.data_ptr() != input.data_ptr())

    @staticmethod
    def backward(ctx, grad_output):
        return grad_output


class Quantization(nn.Module):
    quant_v = 255.0
    is_clip = True

    def __init__(self, quant_v=255.0, is_clip=True):
        super().__init__()
        Quantization.quant_v, Quantization.is_clip = quant_v, is_clip

    def forward(self, input):
        return Quant.apply(input)
