"""ctypes binding of ``lib/libanemoi_graphs_hip.so`` (declared in ``include/anemoi_graphs_hip.h``).

The library is mandatory: there is no CPU or PyTorch fallback for any operation of this package.
"""
from __future__ import annotations

import ctypes
import os

import torch  # noqa: F401  -- must come first: the library then binds to the HIP runtime PyTorch loaded

from ctypes import c_char_p, c_double, c_int, c_int64, c_size_t, c_void_p
from pathlib import Path

LIB_PATH = Path(__file__).resolve().parent / "lib" / "libanemoi_graphs_hip.so"

_P = c_void_p  # every device pointer and the stream travel as void*

# name -> (restype, argtypes); mirrors include/anemoi_graphs_hip.h one to one
SIGNATURES = {
    "ag_last_error": (c_char_p, []),
    "ag_version": (c_int, []),
    "ag_status_take": (c_int, [ctypes.POINTER(c_int)]),
    "ag_debug_poison_scans": (c_int, [c_int]),
    "ag_radius_group_order": (c_int, [c_int]),
    "ag_profile_enable": (c_int, [c_int]),
    "ag_profile_filter": (c_int, [ctypes.c_char_p]),
    "ag_profile_collect": (c_int, [ctypes.c_char_p, c_size_t]),
    "ag_knn_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int, c_double]),
    "ag_knn": (c_int, [_P, c_int64, _P, c_int64, c_int, c_double, c_double, _P, _P, _P, _P, c_size_t, _P]),
    "ag_point_density_workspace_bytes": (c_size_t, [c_int64]),
    "ag_point_density": (c_int, [_P, c_int64, _P, _P, c_size_t, _P]),
    "ag_knn_within": (c_int, [_P, c_int64, _P, c_int64, c_int, c_double, c_double, c_double, _P, _P, _P, _P, c_size_t, _P]),
    "ag_nn_maxdist_workspace_bytes": (c_size_t, [c_int64, c_double]),
    "ag_nn_maxdist": (c_int, [_P, c_int64, c_double, _P, _P, c_size_t, _P]),
    "ag_nn_maxdist_static": (c_int, [_P, c_int64, _P, _P]),
    "ag_radius_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "ag_radius_count": (c_int, [_P, c_int64, _P, c_int64, c_double, _P, _P, c_size_t, _P]),
    "ag_radius_fill": (c_int, [_P, c_int64, _P, c_int64, c_double, _P, c_int64, c_int64, _P, _P, _P, c_size_t, _P]),
    "ag_binning_bytes": (c_size_t, [c_int64]),
    "ag_radius_cell": (c_double, [c_int64, c_double]),
    "ag_binning_build": (c_int, [_P, c_int64, c_double, _P, c_size_t, _P]),
    "ag_radius_binned_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "ag_radius_count_binned": (c_int, [_P, c_int64, _P, c_double, _P, c_int64, c_double, _P, _P, c_size_t, _P]),
    "ag_radius_fill_binned": (
        c_int, [_P, c_int64, _P, c_double, _P, c_int64, c_double, _P, c_int64, c_int64, _P, _P, _P, c_size_t, _P]),
    "ag_knn_binned_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int, c_double]),
    "ag_knn_binned": (c_int, [_P, c_int64, _P, c_int64, _P, c_double, c_int, c_double, _P, _P, _P, _P, c_size_t, _P]),
    "ag_knn_edges": (c_int, [_P, c_int64, _P, c_int64, _P, c_double, c_int, c_double, c_double, _P, _P, _P, _P, c_size_t, _P]),
    "ag_icosphere_sizes": (c_int, [c_int, ctypes.POINTER(c_int64), ctypes.POINTER(c_int64), ctypes.POINTER(c_int64)]),
    "ag_icosphere_workspace_bytes": (c_size_t, [c_int]),
    "ag_icosphere": (c_int, [c_int, _P, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "ag_multiscale_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "ag_multiscale_count": (
        c_int, [c_int, ctypes.POINTER(ctypes.c_int32), c_int, c_int, _P, _P, _P, _P, _P, c_size_t, _P]),
    "ag_multiscale_fill": (
        c_int, [c_int, ctypes.POINTER(ctypes.c_int32), c_int, c_int, c_int64, _P, _P, _P, c_size_t, _P]),
    "ag_multiscale_complete_edges": (c_int64, [ctypes.POINTER(ctypes.c_int32), c_int]),
    "ag_multiscale_complete": (
        c_int, [c_int, ctypes.POINTER(ctypes.c_int32), c_int, _P, _P, _P, _P, _P, c_size_t, _P]),
    "ag_edge_attrs": (c_int, [_P, c_int64, _P, c_int64, _P, _P, c_int64, c_int, _P, _P, _P, _P]),
    "ag_edge_attrs_range": (c_int, [_P, c_int64, _P, c_int64, _P, _P, c_int64, c_int, _P, _P, _P, _P, _P]),
    "ag_len_range_words": (c_size_t, []),
    "ag_normalise_range": (c_int, [_P, c_int64, c_int, c_int, _P, _P]),
    "ag_normalise_workspace_bytes": (c_size_t, [c_int64]),
    "ag_normalise": (c_int, [_P, c_int64, c_int, c_int, _P, c_size_t, _P]),
    "ag_normalise_f64": (c_int, [_P, c_int64, c_int, _P, c_int, _P, c_size_t, _P]),
    "ag_voronoi_areas_knn": (c_int, [_P, c_int64, _P, c_int, _P, _P, _P]),
    "ag_voronoi_far_workspace_bytes": (c_size_t, [c_int64]),
    "ag_voronoi_areas_far": (c_int, [_P, c_int64, _P, c_int64, _P, _P, _P, c_size_t, _P]),
    "ag_voronoi_areas_rows": (c_int, [_P, c_int64, _P, c_int64, _P, _P, c_double, _P, _P, _P]),
    "ag_sort_edges_workspace_bytes": (c_size_t, [c_int64]),
    "ag_sort_edges": (c_int, [_P, _P, c_int64, c_int, _P, c_size_t, _P]),
    "ag_unique_edges": (c_int, [_P, _P, c_int64, _P, _P, c_size_t, _P]),
    "ag_mask_compact_workspace_bytes": (c_size_t, [c_int64]),
    "ag_mask_compact": (c_int, [_P, c_int64, _P, _P, _P, c_size_t, _P]),
    "ag_gather_coords": (c_int, [_P, c_int64, _P, c_int64, _P, _P, _P]),
    "ag_remap_ids": (c_int, [_P, c_int64, _P, c_int64, _P, _P]),
    "ag_mesh_maxdist_workspace_bytes": (c_size_t, [c_int]),
    "ag_mesh_maxdist": (c_int, [c_int, _P, _P, _P, _P, _P, c_size_t, _P]),
    "ag_node_order_workspace_bytes": (c_size_t, [c_int64]),
    "ag_node_order": (c_int, [_P, c_int64, _P, _P, c_size_t, _P]),
    "ag_mark_ids": (c_int, [_P, c_int64, _P, c_int64, _P, _P]),
    "ag_invert_ids": (c_int, [_P, c_int64, _P, c_int64, _P, _P]),
    "ag_scan_i32_state_words": (c_size_t, [c_int64]),
    "ag_scan_i32": (c_int, [_P, _P, c_int64, _P, c_int, _P]),
    "ag_fill_row_ids": (c_int, [_P, c_int64, c_int, _P]),
}


class LibraryError(RuntimeError):
    """An entry point of libanemoi_graphs_hip.so reported failure."""


_lib = None


def load() -> ctypes.CDLL:
    """Load the shared library (once).  Raises if it has not been built: no fallback exists."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m anemoi_graphs_amd.build` "
            "(hipcc, gfx950).  This package has no CPU or PyTorch fallback."
        )
    lib = ctypes.CDLL(str(LIB_PATH))
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    if os.environ.get("AG_RADIUS_GROUPED") == "1":  # (for experiments: the cut-off query takes its targets in group order)
        lib.ag_radius_group_order(1)
    _lib = lib
    return lib


def check(status: int, what: str) -> None:
    if status != 0:
        msg = load().ag_last_error()
        raise LibraryError(f"{what} failed ({status}): {msg.decode() if msg else 'unknown error'}")
