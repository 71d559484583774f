"""ctypes loader of libphmm_amd.so (the HIP extension).  Fails loudly if it is missing:
there is no CPU fallback in the product path."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PHMM_AMD_LIB") or os.path.join(_HERE, "libphmm_amd.so")  # (PHMM_AMD_LIB: a variant build, dev only)
CSRC = os.path.join(_HERE, "csrc")

PHMM_OK, PHMM_EINVAL, PHMM_ENODEVICE, PHMM_ENOMEM, PHMM_ECAPACITY, PHMM_EINTERNAL = 0, -1, -2, -3, -4, -5
PHMM_READ_DEFERRED, PHMM_READ_WIDE_FRONTIER, PHMM_READ_FORCED_SWITCH = 1, 2, 4  # phmm_reads_last_call_info flags


class PhmmError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"phmm_amd error {code}: {msg}")
        self.code = code


def build(force: bool = False) -> str:
    """Compile every HIP source for gfx950 (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", CSRC, "-s", "clean"])
    subprocess.check_call(["make", "-C", CSRC, "-s", "-j4"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C dbgphmm_amd/csrc). The MI355X path has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, u32, u64, i32, i64, dbl = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int, C.c_int64, C.c_double
    P = C.POINTER
    sig = {
        "phmm_last_error": (C.c_char_p, []),
        "phmm_version": (C.c_char_p, []),
        "phmm_device_count": (i32, []),
        "phmm_set_device": (i32, [i32]),
        "phmm_set_stream": (i32, [vp]),
        "phmm_set_workspace_limit": (i32, [u64]),
        "phmm_release_workspace": (i32, []),
        "phmm_workspace_bytes": (u64, []),
        "phmm_reads_last_call_info": (i32, [vp, vp, vp]),
        "phmm_params_new": (i32, [dbl, dbl, dbl, dbl, i64, i64, vp]),
        "phmm_params_uniform": (i32, [dbl, vp]),
        "phmm_model_create": (i32, [u32, u32, vp, vp, vp, vp, vp, vp, P(vp)]),
        "phmm_model_set_probs": (i32, [vp, vp, vp]),
        "phmm_model_set_params": (i32, [vp, vp]),
        "phmm_model_n_nodes": (u32, [vp]),
        "phmm_model_n_edges": (u32, [vp]),
        "phmm_model_destroy": (None, [vp]),
        "phmm_reads_create": (i32, [vp, vp, u64, P(vp)]),
        "phmm_reads_count": (u64, [vp]),
        "phmm_reads_total_bases": (u64, [vp]),
        "phmm_reads_destroy": (None, [vp]),
        "phmm_run_dense": (i32, [vp, vp, vp, vp, vp]),
        "phmm_dense_tables": (i32, [vp, vp, u64, vp, vp, vp, vp, vp, vp, vp, vp]),
        "phmm_mappings_create": (i32, [vp, vp, vp, vp, P(vp)]),
        "phmm_mappings_total_positions": (u64, [vp]),
        "phmm_mappings_total_entries": (u64, [vp]),
        "phmm_mappings_export": (i32, [vp, vp, vp, vp]),
        "phmm_mappings_node_freqs": (i32, [vp, u32, vp]),
        "phmm_mappings_read_logp": (i32, [vp, vp, vp]),
        "phmm_mappings_destroy": (None, [vp]),
        "phmm_full_prob_reads": (i32, [vp, vp, vp, i32, vp, vp]),
        "phmm_run_dense_edges": (i32, [vp, vp, vp, vp, vp]),
        "phmm_full_prob_sparse_backward": (i32, [vp, vp, vp, vp]),
        "phmm_backward_sparse_tables": (i32, [vp, vp, u64, vp, vp, vp, vp, vp]),
        "phmm_run_sparse": (i32, [vp, vp, vp, vp, vp]),
        "phmm_q_score_exact": (i32, [vp, vp, vp, vp]),
        "phmm_mappings_map_nodes": (i32, [vp, vp, vp, vp, vp, u32, vp]),
        "phmm_full_prob_reads_candidates": (i32, [vp, vp, vp, u32, vp, vp, vp, vp]),
        "phmm_full_prob_reads_copy_nums": (i32, [vp, vp, vp, u32, vp, u32, vp, vp]),
        "phmm_generate_mappings": (i32, [vp, vp, vp, i32, P(vp), vp]),
        "phmm_last_call_stats": (i32, [i32, P(dbl), P(u64), P(u64)]),
        "phmm_enable_timing": (i32, [i32]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


DECLARED_SYMBOLS = [
    "phmm_last_error", "phmm_version", "phmm_device_count", "phmm_set_device", "phmm_set_stream",
    "phmm_set_workspace_limit", "phmm_release_workspace", "phmm_workspace_bytes", "phmm_reads_last_call_info", "phmm_params_new", "phmm_params_uniform", "phmm_model_create",
    "phmm_model_set_probs", "phmm_model_set_params", "phmm_model_n_nodes", "phmm_model_n_edges",
    "phmm_model_destroy", "phmm_reads_create", "phmm_reads_count", "phmm_reads_total_bases",
    "phmm_reads_destroy", "phmm_run_dense", "phmm_run_dense_edges", "phmm_q_score_exact", "phmm_run_sparse", "phmm_full_prob_sparse_backward", "phmm_backward_sparse_tables", "phmm_dense_tables", "phmm_mappings_create",
    "phmm_mappings_total_positions", "phmm_mappings_total_entries", "phmm_mappings_export",
    "phmm_mappings_node_freqs", "phmm_mappings_read_logp", "phmm_mappings_map_nodes", "phmm_mappings_destroy", "phmm_full_prob_reads",
    "phmm_full_prob_reads_candidates", "phmm_full_prob_reads_copy_nums", "phmm_generate_mappings", "phmm_last_call_stats", "phmm_enable_timing",
]


def check(rc: int) -> None:
    if rc != PHMM_OK:
        raise PhmmError(rc, lib().phmm_last_error().decode())
