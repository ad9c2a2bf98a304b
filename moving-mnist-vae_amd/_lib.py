"""ctypes binding of libmmvae_hip.so (the C ABI declared in include/mmvae.h).

There is no CPU fallback: if the library is missing the product path raises, loudly.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# (MMVAE_LIB_PATH: developer A/B runs of two builds on one box; the product always loads the in-tree library)
LIB_PATH = os.environ.get("MMVAE_LIB_PATH") or os.path.join(_HERE, "libmmvae_hip.so")
_lib = None
ABI_VERSION = 3            # MMVAE_ABI_VERSION of include/mmvae.h

P = c_void_p

# name -> (restype, argtypes); mirrors include/mmvae.h one to one (checked by tests/test_capi_cpu.py)
PROTOTYPES = {
    "mmvae_abi_version": (c_int, []),
    "mmvae_last_error": (c_char_p, []),
    "mmvae_net_create": (c_int, [POINTER(c_void_p), c_int, c_int, c_int, c_int, c_int, c_int]),
    "mmvae_net_create_ex": (c_int, [POINTER(c_void_p), c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "mmvae_net_destroy": (None, [P]),
    "mmvae_net_sizes": (c_int, [P, POINTER(c_int64), POINTER(c_int64), POINTER(c_int32), POINTER(c_int64), POINTER(c_int32)]),
    "mmvae_net_num_entries": (c_int, [P]),
    "mmvae_net_entry": (c_int, [P, c_int, c_char_p, c_int, POINTER(c_int32), POINTER(c_int32), POINTER(c_int32), POINTER(c_int64)]),
    "mmvae_net_workspace_bytes": (c_size_t, [P, c_int]),
    "mmvae_encoder_fwd": (c_int, [P, c_int, P, P, P, P, P, c_size_t, P, P, c_int, P]),
    "mmvae_net_stage_labels": (c_int, [P, c_int, P, c_int, c_float, c_float, P, P, c_size_t, P]),
    "mmvae_encoder_fwd_staged": (c_int, [P, c_int, P, P, P, P, P, c_size_t, P, P, c_int, P]),
    "mmvae_encoder_bwd": (c_int, [P, c_int, P, P, P, P, P, c_size_t, P]),
    "mmvae_decoder_fwd": (c_int, [P, c_int, P, P, P, P, P, c_size_t, P, c_int, P]),
    "mmvae_decoder_bwd": (c_int, [P, c_int, P, P, P, P, c_size_t, P, P]),
    "mmvae_decoder_bwd_gauss": (c_int, [P, c_int, P, c_float, c_float, P, P, P, P, c_size_t, P, P]),
    "mmvae_net_defer_join": (c_int, [P, c_int]),
    "mmvae_net_join": (c_int, [P, P]),
    "mmvae_net_fork": (P, [P, P]),
    "mmvae_net_side_stream": (P, [P]),
    "mmvae_net_set_join_grad": (c_int, [P, c_int]),
    "mmvae_net_set_sync_bn": (c_int, [P, P, P, c_int]),
    "mmvae_rsample_fwd": (c_int, [P, P, P, P, c_int64, P]),
    "mmvae_rsample_bwd": (c_int, [P, P, P, P, P, c_int64, P]),
    "mmvae_kl_fwd": (c_int, [P, P, c_int64, P, P]),
    "mmvae_kl_bwd": (c_int, [P, P, c_float, P, P, P, c_int64, P]),
    "mmvae_gauss_nll_fwd": (c_int, [P, P, c_int64, c_float, P, P]),
    "mmvae_gauss_nll_bwd": (c_int, [P, P, c_int64, c_float, c_float, P, P, P]),
    "mmvae_ce_fwd": (c_int, [P, P, P, c_int, c_int, c_int, P, P]),
    "mmvae_ce_bwd": (c_int, [P, P, P, c_int, c_int, c_int, c_float, P, P, P]),
    "mmvae_mmd_fwd": (c_int, [P, P, c_int, c_int, P, P, P]),
    "mmvae_mmd_bwd": (c_int, [P, P, c_int, c_int, c_float, P, P, P]),
    "mmvae_loss_finish": (c_int, [P, P, c_float, c_float, c_float, c_float, P]),
    "mmvae_normalise_labels": (c_int, [P, c_int64, c_float, c_float, P, P]),
    "mmvae_quantise_normalise": (c_int, [P, c_int64, P, c_int, c_float, c_float, P, P, P]),
    "mmvae_adam_step": (c_int, [P, P, P, P, c_int64, c_float, c_float, c_float, c_float, c_float, c_float, c_float, c_float, P]),
    "mmvae_adam_step_dev": (c_int, [P, P, P, P, c_int64, c_float, c_float, c_float, c_float, c_float, P, c_float, P]),
    "mmvae_conv2d_fwd": (c_int, [c_int, c_int, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, c_int, P, P, P]),
    "mmvae_conv2d_dgrad": (c_int, [c_int, c_int, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P]),
    "mmvae_conv2d_wgrad": (c_int, [c_int, c_int, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, c_int, P, P]),
    "mmvae_conv2d_wgrad_pair": (c_int, [c_int, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, P, c_int, P, P]),
    "mmvae_rbf_kernel": (c_int, [P, P, c_int, c_int, c_int, P, P]),
    "mmvae_comm_unique_id": (c_int, [P]),
    "mmvae_comm_init": (c_int, [POINTER(c_void_p), c_int, c_int, P]),
    "mmvae_comm_allreduce": (c_int, [P, P, c_int64, P]),
    "mmvae_comm_destroy": (c_int, [P]),
    "mmvae_net_set_sync_bn_comm": (c_int, [P, P]),
    "mmvae_net_set_sync_bn_comm2": (c_int, [P, P, P]),
    "mmvae_convT_bwd_fused": (c_int, [c_int, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, c_int, P, P, P, P, P, P]),
    "mmvae_batchnorm_fwd": (c_int, [c_int, P, c_int64, c_int, P, P, P, P, P, c_float, c_float, c_int, P, P, P, P, P]),
    "mmvae_batchnorm_bwd": (c_int, [c_int, P, P, P, c_int64, c_int, P, P, P, P, P, P, P, P]),
    "mmvae_stem_fwd": (c_int, [c_int, P, P, P, c_int, c_int, P, P, P]),
    "mmvae_stem_bwd": (c_int, [c_int, P, P, P, P, P, P, P, P, P, P, P, P, c_int, c_int, P, P]),
    "mmvae_tail_join_fwd": (c_int, [c_int, P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, P]),
    "mmvae_tail_join_fwd_stream": (c_int, [c_int, P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, P]),
    "mmvae_tail_join_bwd_reduce": (c_int, [c_int, P, P, c_int, P, P, P, P, P, P, P, P, c_int, c_int, c_int, P]),
    "mmvae_tail_join_bwd_apply": (c_int, [c_int, P, P, c_int, P, P, P, P, P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, P]),
    "mmvae_upblock_bwd_fused": (c_int, [P] * 27 + [c_int, P, P]),
    "mmvae_join_conv1x1_fwd": (c_int, [P, P, P, P, P, P, P, c_int, P, P, P, c_int64, P, P]),
    "mmvae_conv1x1_bwd_fused": (c_int, [P] * 13 + [c_int64, P, P]),
    "mmvae_upblock_tail_fwd": (c_int, [P] * 16 + [c_int, P, P]),
    "mmvae_pixelcnn_create": (c_int, [POINTER(c_void_p), c_int, c_int, c_int, c_int, c_int]),
    "mmvae_pixelcnn_destroy": (None, [P]),
    "mmvae_pixelcnn_num_params": (c_int64, [P]),
    "mmvae_pixelcnn_workspace_bytes": (c_size_t, [P, c_int, c_int]),
    "mmvae_pixelcnn_fwd": (c_int, [P, c_int, c_int, P, P, P, c_size_t, P, P]),
    "mmvae_pixelcnn_bwd": (c_int, [P, c_int, c_int, P, P, P, P, P, c_size_t, P, P]),
    "mmvae_convert": (c_int, [c_int, c_int, P, P, c_int64, P]),
}


class MmvaeError(RuntimeError):
    pass


def lib():
    """Load (once) and return the shared library.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MmvaeError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` (or `make -C moving-mnist-vae_amd/csrc`). "
            "There is no CPU fallback for the product path.")
    l = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(l, name)
        fn.restype = res
        fn.argtypes = args
    # include/mmvae.h: "bindings check mmvae_abi_version() at load time" -- a stale or foreign build would be called with shifted arguments
    got = l.mmvae_abi_version()
    if got != ABI_VERSION:
        raise MmvaeError(f"{LIB_PATH} has C ABI version {got}, this binding expects {ABI_VERSION}: rebuild the library "
                         "(make -C moving-mnist-vae_amd/csrc)")
    _lib = l
    return l


def build_hash() -> str:
    """Short hash of the kernel sources the library was built from (csrc/ + the public header); keys the committed PMC
    tables under profiles/ to the build they were measured on."""
    import hashlib
    h = hashlib.sha1()
    csrc = os.path.join(_HERE, "csrc")
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".inc", ".cpp", ".hpp")))
    files.append(os.path.join(os.path.dirname(_HERE), "include", "mmvae.h"))
    for f in files:
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode() + b"\0" + fh.read())
    return h.hexdigest()[:12]


def check(rc: int, what: str) -> int:
    if rc < 0:
        msg = lib().mmvae_last_error()
        raise MmvaeError(f"{what} failed ({rc}): {msg.decode() if msg else '?'}")
    return rc


def ptr(t):
    """Device/host pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()
