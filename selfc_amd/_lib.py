"""ctypes binding of libselfc_hip.so (C ABI: include/selfc_hip.h).

This is the binding a reference maintainer would add: plain
pointers (``tensor.data_ptr()``), sizes and the current HIP stream.  Loading
fails loudly - there is no fallback implementation behind it.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
#: MFMA operand type of the loaded library: "f16" (default, parity-grade) or "bf16" (SELFC_OPERAND=bf16 selects
#: libselfc_hip_bf16.so, the same kernels built with bfloat16 operands; per process, fixed at import)
OPERAND = os.environ.get("SELFC_OPERAND", "f16").lower()
if OPERAND not in ("f16", "bf16"):
    raise RuntimeError(f"SELFC_OPERAND must be 'f16' or 'bf16', got {OPERAND!r}")
# SELFC_LIB: developer override to A/B two builds of the library inside one process launch script
LIB_PATH = os.environ.get("SELFC_LIB") or os.path.join(_HERE, "libselfc_hip.so" if OPERAND == "f16" else "libselfc_hip_bf16.so")


def operand_dtype():
    """torch dtype of packed weights and dense feature buffers (matches the loaded library)."""
    import torch
    return torch.float16 if OPERAND == "f16" else torch.bfloat16

SUBNET_D2DT = 0
SUBNET_DB2D = 1
LAT_KEEP_FEATURES = 1      # selfc_latent.flags (SELFC_LAT_KEEP_FEATURES)
ABI_VERSION = 13

#: every symbol include/selfc_hip.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "selfc_version", "selfc_abi_version",
    "selfc_haar_fwd_nchw", "selfc_haar_inv_nchw", "selfc_freq_fwd", "selfc_freq_inv",
    "selfc_nchw_to_latent", "selfc_latent_to_nchw", "selfc_quantize_inplace", "selfc_quantize_inplace_v",
    "selfc_invblock_run", "selfc_invstack_run", "selfc_subnet_run",
    "selfc_nchw_to_nhwc4", "selfc_nhwc4_to_nchw",
    "selfc_profile_enable", "selfc_profile_read", "selfc_profile_reset", "selfc_profile_calibrate", "selfc_profile_clock_sample",
    "selfc_globalagg_run", "selfc_globalagg_run_d", "selfc_gmm_sample_generic", "selfc_globalagg_partial_floats", "selfc_pwconv_run", "selfc_gmm_sample", "selfc_stp_head_gmm",
    "selfc_conv_planes_run", "selfc_nhwc_to_planes", "selfc_y_sse", "selfc_y_sse_blocks", "selfc_y_ssim", "selfc_gauss_down4",
    "selfc_subnet_bwd_scratch_bytes", "selfc_subnet_bwd", "selfc_subnet_bwd_phase", "selfc_coupling_fwd", "selfc_coupling_bwd", "selfc_freq_fwd_bwd", "selfc_freq_inv_bwd",
    "selfc_bwd_scale", "selfc_bwd_to_planes", "selfc_f16_rows_to_planes", "selfc_bwd_conv_planes",
    "selfc_bwd_wgrad_scratch_bytes", "selfc_bwd_wgrad", "selfc_gmm_sample_bwd", "selfc_gmm_sample_generic_bwd", "selfc_lrelu_bwd",
    "selfc_globalagg_bwd_scratch_bytes", "selfc_globalagg_bwd", "selfc_rowsum_accum",
    "selfc_coupling_bwd_x", "selfc_add_absmax", "selfc_subnet_bwd_phase_x", "selfc_stream_create", "selfc_stream_destroy", "selfc_set_pointers", "selfc_freq_fwd_ind", "selfc_freq_inv_ind", "selfc_nchw_to_latent_ind", "selfc_latent_to_nchw_ind",
    "selfc_nchw_to_nhwc4_ind", "selfc_nhwc4_to_nchw_ind", "selfc_graph_stats",
    "selfc_fin_job_bytes", "selfc_wgrad_finish_jobs", "selfc_subnet_bwd_phase_d", "selfc_gh_bwd_pair_scratch_bytes", "selfc_gh_bwd_pair", "selfc_recon_loss_blocks", "selfc_recon_loss", "selfc_wg_job_bytes", "selfc_wgrad_run_jobs", "selfc_clip_adam_blocks", "selfc_clip_adam", "selfc_globalagg_bwd_x",
]


class SubnetW(C.Structure):
    _fields_ = [("w3", C.c_void_p * 4), ("b3", C.c_void_p * 4), ("w5", C.c_void_p), ("b5", C.c_void_p),
                ("wfused", C.c_void_p), ("w5p", C.c_void_p)]


class SubnetBW(C.Structure):
    _fields_ = [("wt5", C.c_void_p), ("wtd", C.c_void_p * 3), ("wtx", C.c_void_p)]


class RowSum(C.Structure):
    _fields_ = [("src", C.c_void_p * 8), ("dst", C.c_void_p * 8), ("len", C.c_int * 8), ("rows", C.c_int * 8), ("beta", C.c_float * 8), ("n", C.c_int)]


class InvBlockW(C.Structure):
    _fields_ = [("F", SubnetW), ("G", SubnetW), ("H", SubnetW), ("clamp", C.c_float)]


class Latent(C.Structure):
    _fields_ = [("kind", C.c_int), ("N", C.c_int), ("T", C.c_int), ("H", C.c_int), ("W", C.c_int),
                ("c1", C.c_int), ("c2", C.c_int),
                ("x1", C.c_void_p), ("x2", C.c_void_p), ("fd", C.c_void_p), ("gd", C.c_void_p),
                ("hd", C.c_void_p), ("s_out", C.c_void_p), ("pf", C.c_void_p), ("flags", C.c_int), ("fd_next", C.c_void_p),
                ("x1_out", C.c_void_p), ("x2_out", C.c_void_p)]


_lib = None


def lib():
    """The loaded library; raises RuntimeError when it has not been built."""
    global _lib
    if _lib is None:
        # torch must be loaded first: it bundles its own libamdhip64, and libselfc_hip.so has to bind to
        # THAT runtime (same soname) - loading our library first pulls in /opt/rocm's copy and the process
        # ends up with two HIP runtimes (kernel launches then fail with hipErrorNoDevice).
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C selfc_amd/csrc`). selfc_amd has no CPU / eager fallback.")
        L = C.CDLL(LIB_PATH)
        vp, i, f, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
        L.selfc_version.restype = C.c_char_p
        L.selfc_version.argtypes = []
        L.selfc_abi_version.restype = i
        L.selfc_abi_version.argtypes = []
        sigs = {
            "selfc_haar_fwd_nchw": [vp, vp, i, i, i, i, vp],
            "selfc_haar_inv_nchw": [vp, vp, i, i, i, i, vp],
            "selfc_freq_fwd": [vp, vp, vp, vp, i, i, i, i, i, vp],
            "selfc_freq_inv": [vp, vp, vp, i, i, i, i, vp],
            "selfc_nchw_to_latent": [vp, vp, vp, vp, i, i, i, i, i, i, vp],
            "selfc_latent_to_nchw": [vp, vp, vp, i, i, i, i, i, vp],
            "selfc_quantize_inplace": [vp, sz, vp],
            "selfc_quantize_inplace_v": [vp, sz, f, i, vp],
            "selfc_invblock_run": [C.POINTER(InvBlockW), C.POINTER(Latent), i, vp],
            "selfc_invstack_run": [C.POINTER(InvBlockW), i, C.POINTER(Latent), i, vp],
            "selfc_subnet_run": [C.POINTER(SubnetW), i, vp, vp, vp, i, i, i, i, i, i, vp],
            "selfc_nchw_to_nhwc4": [vp, vp, i, i, i, i, vp],
            "selfc_nhwc4_to_nchw": [vp, vp, i, i, i, i, vp],
            "selfc_profile_enable": [i],
            "selfc_profile_read": [i, C.POINTER(C.c_double), C.POINTER(C.c_longlong)],
            "selfc_profile_reset": [],
            "selfc_profile_calibrate": [C.POINTER(C.c_double), C.POINTER(C.c_double), vp],
            "selfc_profile_clock_sample": [vp, i, vp],
            "selfc_globalagg_run": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i, i, i, vp],
            "selfc_globalagg_run_d": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i, i, i, i, vp],
            "selfc_gmm_sample_generic": [vp, vp, vp, sz, i, i, i, i, f, vp],
            "selfc_pwconv_run": [vp, i, vp, i, vp, vp, sz, i, i, i, i, i, vp],
            "selfc_gmm_sample": [vp, vp, vp, sz, i, i, vp],
            "selfc_stp_head_gmm": [vp, vp, vp, vp, vp, sz, i, i, i, i, vp],
            "selfc_conv_planes_run": [vp, i, i, vp, vp, i, i, vp, i, i, i, i, vp],
            "selfc_nhwc_to_planes": [vp, vp, sz, i, vp],
            "selfc_y_sse": [vp, vp, vp, i, i, vp],
            "selfc_y_sse_blocks": [i],
            "selfc_gauss_down4": [vp, vp, vp, i, i, i, vp],
            "selfc_y_ssim": [vp, vp, vp, vp, i, i, i, vp],
            "selfc_subnet_bwd": [C.POINTER(SubnetBW), i, vp, vp, vp, f, vp, i, C.POINTER(vp), C.POINTER(vp), f,
                                 vp, sz, i, i, i, i, i, i, vp],
            "selfc_subnet_bwd_phase": [i, C.POINTER(SubnetBW), i, vp, vp, vp, f, vp, i, C.POINTER(vp), C.POINTER(vp), f,
                                       vp, sz, i, i, i, i, i, i, vp],
            "selfc_subnet_bwd_phase_x": [i, C.POINTER(SubnetBW), i, vp, vp, vp, f, vp, i, C.POINTER(vp), C.POINTER(vp), f,
                                         vp, sz, i, i, i, i, i, i, vp, vp, vp],
            "selfc_subnet_bwd_phase_d": [i, C.POINTER(SubnetBW), i, vp, vp, vp, f, vp, i, C.POINTER(vp), C.POINTER(vp), f,
                                         vp, sz, i, i, i, i, i, i, vp, vp, vp, vp, vp],
            "selfc_gh_bwd_pair": [i, C.POINTER(SubnetBW), C.POINTER(SubnetBW), vp, vp, vp, vp, vp, f, f, vp, i,
                                  C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), f, vp, sz, i, i, i, i, i, i, vp, vp, vp, vp, vp, vp],
            "selfc_wgrad_finish_jobs": [vp, i, vp],
            "selfc_wgrad_run_jobs": [vp, i, vp],
            "selfc_recon_loss": [vp, sz, vp, sz, sz, sz, i, f, f, vp, vp, vp, vp],
            "selfc_recon_loss_blocks": [],
            "selfc_clip_adam_blocks": [],
            "selfc_clip_adam": [vp, vp, vp, vp, sz, vp, f, vp, f, C.c_double, C.c_double, f, f, vp, f, vp, vp],
            "selfc_coupling_bwd_x": [i, vp, vp, vp, vp, vp, f, sz, vp, vp, vp],
            "selfc_add_absmax": [vp, vp, sz, vp, vp],
            "selfc_coupling_fwd": [i, vp, vp, vp, vp, vp, f, sz, vp],
            "selfc_coupling_bwd": [i, vp, vp, vp, vp, vp, f, sz, vp],
            "selfc_freq_fwd_bwd": [vp, vp, vp, i, i, i, vp],
            "selfc_freq_inv_bwd": [vp, vp, vp, i, i, i, vp],
            "selfc_bwd_scale": [vp, sz, vp, vp],
            "selfc_bwd_to_planes": [vp, vp, sz, i, i, i, f, vp, vp],
            "selfc_f16_rows_to_planes": [vp, vp, sz, i, vp],
            "selfc_bwd_conv_planes": [vp, i, i, i, vp, i, vp, vp, vp, i, vp, i, i, vp, i, i, i, i, vp],
            "selfc_bwd_wgrad": [vp, i, vp, i, i, vp, i, i, vp, f, vp, vp, sz, i, i, i, i, vp],
            "selfc_gmm_sample_bwd": [vp, vp, vp, vp, sz, i, i, vp],
            "selfc_gmm_sample_generic_bwd": [vp, vp, vp, vp, sz, i, i, i, i, f, vp],
            "selfc_lrelu_bwd": [vp, vp, sz, vp],
            "selfc_rowsum_accum": [C.POINTER(RowSum), vp],
            "selfc_stream_create": [C.POINTER(vp)],
            "selfc_stream_destroy": [vp],
            "selfc_graph_stats": [vp, C.POINTER(C.c_longlong)],
            "selfc_set_pointers": [vp, i, vp, vp, vp, vp, vp],
            "selfc_freq_fwd_ind": [vp, sz, vp, vp, vp, i, i, i, i, i, vp],
            "selfc_freq_inv_ind": [vp, vp, vp, sz, i, i, i, i, vp],
            "selfc_nchw_to_latent_ind": [vp, sz, vp, vp, vp, i, i, i, i, i, i, vp],
            "selfc_latent_to_nchw_ind": [vp, vp, vp, sz, i, i, i, i, i, vp],
            "selfc_nchw_to_nhwc4_ind": [vp, sz, vp, i, i, i, i, vp],
            "selfc_nhwc4_to_nchw_ind": [vp, vp, sz, i, i, i, i, vp],
            "selfc_globalagg_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, i, i, i, i, vp],
            "selfc_globalagg_bwd_x": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, i, i, i, i, vp, vp, vp],
        }
        for name, args in sigs.items():
            fn = getattr(L, name)
            fn.restype = i
            fn.argtypes = args
        L.selfc_y_sse_blocks.restype = i
        L.selfc_bwd_wgrad_scratch_bytes.restype = sz
        L.selfc_bwd_wgrad_scratch_bytes.argtypes = [i, i, i, i, i, i]
        L.selfc_globalagg_bwd_scratch_bytes.restype = sz
        L.selfc_globalagg_bwd_scratch_bytes.argtypes = [i, i, i, i]
        L.selfc_gh_bwd_pair_scratch_bytes.restype = sz
        L.selfc_gh_bwd_pair_scratch_bytes.argtypes = [i, i, i, i, i]
        L.selfc_fin_job_bytes.restype = sz
        L.selfc_fin_job_bytes.argtypes = []
        L.selfc_wg_job_bytes.restype = sz
        L.selfc_wg_job_bytes.argtypes = []
        L.selfc_subnet_bwd_scratch_bytes.restype = sz
        L.selfc_subnet_bwd_scratch_bytes.argtypes = [i, i, i, i, i]
        L.selfc_globalagg_partial_floats.restype = sz
        L.selfc_globalagg_partial_floats.argtypes = [i, i]
        if L.selfc_abi_version() != ABI_VERSION:
            raise RuntimeError(f"{LIB_PATH} has ABI {L.selfc_abi_version()}, this binding needs {ABI_VERSION}: rebuild it (make -C selfc_amd/csrc)")
        if ("operands=" + OPERAND).encode() not in L.selfc_version():
            raise RuntimeError(f"{LIB_PATH} was not built for {OPERAND} operands: {L.selfc_version()!r}")
        _lib = L
    return _lib


def check(rc, what):
    if rc == 0:
        return
    if rc == -1:
        raise RuntimeError(f"{what}: SELFC_EINVAL - shape/argument not covered by the HIP kernels")
    raise RuntimeError(f"{what}: HIP runtime error {-(rc + 1000)}")


def stream_ptr():
    """hipStream_t of torch's current stream (kernels are enqueued there)."""
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu(t):
    if not t.is_cuda:
        raise RuntimeError("selfc_amd runs on MI355X only: tensor is on %s (no CPU fallback)" % t.device)
