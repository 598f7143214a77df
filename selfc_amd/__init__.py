"""selfc_amd - MI355X (gfx950) implementation of SelfC's invertible-rescaling hot path.

Host side mirrors the reference's module API (codes/models/modules/*.py,
codes/global_var.py); compute is hand-written HIP in ``selfc_amd/csrc`` behind
the C ABI of ``include/selfc_hip.h`` (``libselfc_hip.so``).  There is no CPU or
eager-PyTorch fallback: every forward raises if the library or a GPU is missing.
"""
from .global_var import GlobalVar  # noqa: F401

__all__ = ["GlobalVar"]
