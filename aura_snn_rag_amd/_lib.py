"""ctypes binding of ``libaura_hip.so`` (the C ABI declared in ``include/aura_hip.h``).

The library is the product: there is no CPU or PyTorch fallback.  ``load()`` raises
``AuraHipUnavailable`` if the shared object is missing, and every op wrapper in ``ops.py`` raises
if it is handed tensors that are not on a HIP device.
"""
from __future__ import annotations

import ctypes
import os
import threading
from ctypes import c_char_p, c_float, c_int, c_int32, c_int64, c_uint32, c_void_p
from typing import Optional

LIB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib")
# AURA_HIP_LIB: another build of the same library (kernel tuning experiments, tools/variant_sweep.sh)
LIB_PATH = os.environ.get("AURA_HIP_LIB") or os.path.join(LIB_DIR, "libaura_hip.so")

AURA_OK = 0
ERRORS = {-1: "AURA_E_INVAL (bad argument)", -2: "AURA_E_LAUNCH (HIP launch error)",
          -3: "AURA_E_ALIGN (pointer alignment)"}

DTYPE_F32, DTYPE_BF16 = 0, 1
GIF_TIME_INVARIANT, GIF_MEAN_OUT = 1, 2
KNN_FORCE_DENSE = 1
KNN_FP32_SCAN = 2
KNN_FLAG_LISTS_STALE = 128     # bit of the overflow flag: the inverted lists dropped a row (re-pack and repeat)
KNN_FLAG_NO_CANDIDATES = 64   # bit of the overflow flag: a query without any candidate row (not an overflow)

# name -> (restype, argtypes); mirrors include/aura_hip.h one to one
P, I64, I32, F, I, U32 = c_void_p, c_int64, c_int32, c_float, c_int, c_uint32
SIGNATURES = {
    "aura_version": (c_char_p, []),
    "aura_izh_run_nt": (I, [P, P, P, P, F, F, F, F, F, I64, I64, P]),
    "aura_izh_run_btd": (I, [P, P, P, P, F, F, F, F, F, I64, I64, I64, P]),
    "aura_adex_run_nt": (I, [P, P, P, P, P, I64, I64, P]),
    "aura_adex_run_btd": (I, [P, P, P, P, P, I64, I64, I64, P]),
    "aura_lif_run": (I, [P, P, P, P, P, I64, I64, I64, P]),
    "aura_gif_run": (I, [P, P, P, P, F, I, F, F, I64, I64, I64, I, I, P]),
    "aura_bank_row_norms": (I, [P, P, I64, I64, I64, P]),
    "aura_bank_write": (I, [P, P, P, P, P, P, I, P, P, P, I, F, I64, I64, P]),
    "aura_bank_write_online_workspace_bytes": (I64, [I64]),
    "aura_bank_write_online": (I, [P, P, P, P, P, P, I, P, P, P, I, F, I64, I64, P, I64, P]),
    "aura_bank_decay": (I, [P, F, I64, P]),
    "aura_knn_workspace_bytes": (I64, [I64, I64, I]),
    "aura_knn_search": (I, [P, P, P, P, I, P, P, F, I64, I64, I64, I, I32, P, P, P, I64, P]),
    "aura_knn_search_ex": (I, [P, P, P, P, I, P, P, F, I64, I64, I64, I, I32, P, P, P, I64, I, P,
                               P, I, P]),
    "aura_knn_ivf_workspace_bytes": (I64, [I64, I, I]),
    "aura_knn_search_ivf": (I, [P, P, P, P, F, I64, I64, I64, I, P, I, P, P, P, I, I32, P, P, P, I64, P, P]),
    "aura_topk_merge": (I, [P, P, I, I64, I, P, P, P]),
    "aura_bank_gather": (I, [P, I64, P, P, I64, I64, P]),
    "aura_kmeans_assign": (I, [P, P, P, P, I64, I64, I, P]),
    "aura_kmeans_means_workspace_bytes": (I64, [I64, I64, I]),
    "aura_kmeans_segment_means": (I, [P, P, P, P, P, I64, I64, I64, I, I, P]),
    "aura_kmeans_commit": (I, [P, P, P, P, I64, I, P]),
    "aura_addition_linear": (I, [P, P, P, P, I64, I64, I64, P]),
    "aura_addition_linear_backward": (I, [P, P, P, P, P, I64, I64, I64, P]),
    "aura_gif_train_forward": (I, [P, P, P, P, P, P, F, I, F, F, I64, I64, I64, P]),
    "aura_gif_backward": (I, [P, P, P, P, P, P, F, I, F, F, I64, I64, I64, P]),
    "aura_gif_train_forward_bf16": (I, [P, P, P, P, P, P, F, I, F, F, I64, I64, I64, P]),
    "aura_gif_backward_bf16": (I, [P, P, P, P, P, P, F, I, F, F, I64, I64, I64, P]),
    "aura_lif_train_forward": (I, [P, P, P, P, P, P, P, I64, I64, P]),
    "aura_lif_backward": (I, [P, P, P, P, P, P, P, P, P, I64, I64, P]),
    "aura_gif_prosody_run": (I, [P, P, P, P, P, F, I, F, F, F, I64, I64, I64, P]),
    "aura_gif_prosody_train_forward": (I, [P, P, P, P, P, P, P, F, I, F, F, F, I64, I64, I64, P]),
    "aura_gif_prosody_backward": (I, [P, P, P, P, P, P, P, P, P, F, I, F, F, F, I64, I64, I64, P]),
    "aura_lif_run_bf16": (I, [P, P, P, P, P, I64, I64, I64, P]),
    "aura_lif_train_forward_bf16": (I, [P, P, P, P, P, P, P, I64, I64, P]),
    "aura_lif_backward_bf16": (I, [P, P, P, P, P, P, P, P, P, I64, I64, P]),
    "aura_gif_prosody_run_bf16": (I, [P, P, P, P, P, F, I, F, F, F, I64, I64, I64, P]),
    "aura_gif_prosody_train_forward_bf16": (I, [P, P, P, P, P, P, P, F, I, F, F, F, I64, I64, I64, P]),
    "aura_gif_prosody_backward_bf16": (I, [P, P, P, P, P, P, P, P, P, F, I, F, F, F, I64, I64, I64, P]),
    "aura_bank_shadow_update": (I, [P, P, P, P, P, I64, I64, I64, P]),
    "aura_knn_search_shadow": (I, [P, P, P, P, P, P, F, I64, I64, I64, I, I32, P, P, P, I64, I, P, P, I, P]),
    "aura_knn_ivf2_workspace_bytes": (I64, [I64, I64, I]),
    "aura_bank_shadow_sorted": (I, [P, P, P, P, P, P, I64, I64, P]),
    "aura_ivf2_append": (I, [P, P, P, P, I64, I64, P, P, P, P, P, P, P, P, F, P]),
    "aura_ivf2_row_constants": (I, [P, P, P, I64, I64, F, P, P]),
    "aura_knn_search_ivf2": (I, [P, P, P, P, P, P, P, P, P, P, I64, I64, P, F, I64, I64, I, P, I, I32, P, P, P, I64, P, P]),
    "aura_centroid_probe_workspace_bytes": (I64, [I64]),
    "aura_centroid_probe": (I, [P, P, I64, I64, I, P, P, I64, P]),
    "aura_knn_search_ivf2_probed": (I, [P, P, P, P, P, P, P, P, P, P, I64, I64, P, F, I64, I64, I, P, I, P, I32, P, P, P,
                                        I64, P, P]),
    "aura_knn_search_ivf2_staged": (I, [P, P, P, P, P, P, P, P, P, P, I64, I64, P, F, I64, I64, I, P, I, P, I32, P, P, P,
                                        I64, P, I, I, P, P]),
    "aura_host_word_alloc": (I, [P]),
    "aura_host_word_free": (I, [P]),
    "aura_signal_flag": (I, [P, P, U32, P]),
    "aura_knn_search_ivf2_signal": (I, [P, P, P, P, P, P, P, P, P, P, I64, I64, P, F, I64, I64, I, P, I, P, I32, P, P, P,
                                        I64, P, P, U32, P]),
    "aura_profile_begin": (I, [I]),
    "aura_debug_cs_flags": (I, [I]),
    "aura_debug_clock_mhz": (I, [P, I, P]),
    "aura_profile_end": (I, [P, I]),
    "aura_profile_last_scan": (I, [P, P]),
    "aura_profile_last_scan_kind": (I, []),
}


class AuraHipUnavailable(RuntimeError):
    """libaura_hip.so is not built / not loadable.  There is deliberately no fallback."""


class AuraHipError(RuntimeError):
    """A C-ABI call returned a negative status."""


_lock = threading.Lock()
_lib: Optional[ctypes.CDLL] = None


def load() -> ctypes.CDLL:
    """Load the shared library once and attach the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise AuraHipUnavailable(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
                f"g.build()'` or `make -C aura_snn_rag_amd/csrc`.  aura_snn_rag_amd has no CPU "
                f"or PyTorch fallback for the hot path.")
        # torch ships its own libamdhip64 (same SONAME); import it first so that our library
        # binds to the runtime that owns torch's streams and allocations.
        import torch  # noqa: F401
        try:
            lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        except OSError as e:  # pragma: no cover - depends on the box
            raise AuraHipUnavailable(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError here = header/library mismatch: be loud
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def check(rc: int, what: str) -> None:
    if rc != AURA_OK:
        raise AuraHipError(f"{what} failed: {ERRORS.get(rc, rc)}")


def loaded_path() -> Optional[str]:
    return LIB_PATH if _lib is not None else None
