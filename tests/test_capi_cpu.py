"""CPU: the shared library loads, exports every symbol the header declares, the ctypes table mirrors
the header, and argument validation (which runs before any HIP call) reports errors."""
import ctypes
import re
from pathlib import Path

import pytest

from anemoi_graphs_amd import _lib

ROOT = Path(__file__).resolve().parents[1]
HEADER = (ROOT / "include" / "anemoi_graphs_hip.h").read_text()


def declared_functions():
    body = re.sub(r"/\*.*?\*/", "", HEADER, flags=re.S)
    return sorted(set(re.findall(r"\b(ag_[a-z0-9_]+)\s*\(", body)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(str(_lib.LIB_PATH))
    names = declared_functions()
    assert len(names) >= 25
    for name in names:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"


def test_ctypes_table_matches_header():
    assert sorted(_lib.SIGNATURES) == declared_functions()


def test_argument_validation_without_gpu():
    lib = _lib.load()
    assert lib.ag_version() >= 100
    rc = lib.ag_knn(None, 10, None, 5, 0, 1.0, 1.0, None, None, None, None, 0, None)
    assert rc == -1 and b"k=0" in lib.ag_last_error()
    rc = lib.ag_knn(None, 2, None, 5, 3, 1.0, 1.0, None, None, None, None, 0, None)
    assert rc == -1 and b"exceeds" in lib.ag_last_error()
    assert lib.ag_nn_maxdist(None, 1, 1.0, None, None, 0, None) == -1
    assert lib.ag_normalise(None, 10, 9, 0, None, 0, None) == -1
    v, f, e = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
    assert lib.ag_icosphere_sizes(7, ctypes.byref(v), ctypes.byref(f), ctypes.byref(e)) == 0
    assert (v.value, f.value, e.value) == (163842, 327680, 10 * (4**8 - 1))
    assert lib.ag_icosphere_sizes(12, ctypes.byref(v), ctypes.byref(f), ctypes.byref(e)) == -1
    with pytest.raises(_lib.LibraryError):
        _lib.check(lib.ag_icosphere_sizes(-1, None, None, None), "ag_icosphere_sizes")


def test_id_table_sizes_are_validated_without_gpu():
    """The entry points that follow ids take the length of the table the ids index; sizes are validated before
    any HIP call (the per-id check itself runs in the kernels, tests/test_edges_utils_gpu.py)."""
    lib = _lib.load()
    assert lib.ag_remap_ids(None, 5, None, -1, None, None) == -1 and b"negative" in lib.ag_last_error()
    assert lib.ag_remap_ids(None, 5, None, 10, None, None) == -1 and b"null" in lib.ag_last_error()
    assert lib.ag_gather_coords(None, -1, None, 5, None, None, None) == -1
    assert lib.ag_invert_ids(None, 5, None, -3, None, None) == -1
    assert lib.ag_invert_ids(None, 2**31, None, 5, None, None) == -1
    assert lib.ag_mark_ids(None, 5, None, -1, None, None) == -1
    assert lib.ag_remap_ids(None, 0, None, 0, None, None) == 0  # nothing to do
    # edges between empty node sets cannot be valid
    assert lib.ag_edge_attrs(None, 0, None, 10, None, None, 4, 1, None, None, None, None) == -1
    assert b"4 edges" in lib.ag_last_error()
    assert lib.ag_edge_attrs(None, 2**31, None, 10, None, None, 4, 1, None, None, None, None) == -1


def test_builtin_scratch_holds_the_largest_input_it_promises():
    """ag_nn_maxdist_static checks the node count, then that its built-in scratch array holds the workspace of that
    many nodes, then the pointers -- all before any HIP call: the largest promised input must get as far as the
    pointer check (a layout change that outgrows the array shows up here, not on the GPU)."""
    lib = _lib.load()
    text = HEADER[HEADER.index("#define AG_NN_MAXDIST_STATIC_MAX_N"):]
    max_n = int(text.split()[2])
    assert lib.ag_nn_maxdist_static(None, max_n, None, None) == -1 and b"null pointer" in lib.ag_last_error()
    assert lib.ag_nn_maxdist_static(None, max_n + 1, None, None) == -1 and b"exceed" in lib.ag_last_error()
