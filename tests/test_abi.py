"""The C-ABI library builds, loads and exports every symbol include/selfc_hip.h declares (no GPU needed)."""
import ctypes
import os
import re

from conftest import ROOT


def test_exports_match_header():
    import __graft_entry__ as ge
    ge.build()
    from selfc_amd import _lib
    assert os.path.exists(_lib.LIB_PATH)
    hdr = open(os.path.join(ROOT, "include", "selfc_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(selfc_[a-z0-9_]+)\s*\(", hdr)))
    assert declared == sorted(_lib.SYMBOLS)
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name
    assert _lib.lib().selfc_abi_version() == _lib.ABI_VERSION == 13
    assert b"gfx950" in _lib.lib().selfc_version() and b"operands=f16" in _lib.lib().selfc_version()
    bf = ctypes.CDLL(os.path.join(os.path.dirname(_lib.LIB_PATH), "libselfc_hip_bf16.so"))     # the bf16-operand build
    bf.selfc_version.restype = ctypes.c_char_p
    assert b"operands=bf16" in bf.selfc_version()
    for name in declared:
        assert hasattr(bf, name), name


def test_struct_layout_matches_header():
    from selfc_amd import _lib
    p = ctypes.sizeof(ctypes.c_void_p)
    assert ctypes.sizeof(_lib.SubnetW) == 12 * p
    assert ctypes.sizeof(_lib.InvBlockW) == 36 * p + 8        # float + tail padding
    assert ctypes.sizeof(_lib.Latent) == 8 * 4 + 7 * p + 8 + 3 * p    # 7 ints padded to 8, 7 pointers, flags (padded), fd_next, x1_out, x2_out


def test_product_does_not_import_oracle():
    """The product package must never route through the CPU oracle."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "selfc_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("# oracle", ""), f
