"""TEST INFRASTRUCTURE ONLY -- tie-aware comparison of neighbour tables and edge sets."""
from __future__ import annotations

import numpy as np

# two float64 reduced distances closer than this (relative) are an exact tie for the purpose of
# neighbour selection: the reference breaks such ties by ball-tree traversal order
# (sklearn/utils/_heap.pyx: first pushed wins), which no formula reproduces
TIE_RTOL = 1e-13


def rdist64(tgt: np.ndarray, src: np.ndarray) -> np.ndarray:
    """sklearn's HaversineDistance.rdist on float64 copies of float32 coordinates (row-wise)."""
    t = tgt.astype(np.float64)
    s = src.astype(np.float64)
    s0 = np.sin(0.5 * (t[..., 0] - s[..., 0]))
    s1 = np.sin(0.5 * (t[..., 1] - s[..., 1]))
    return s0 * s0 + np.cos(t[..., 0]) * np.cos(s[..., 0]) * s1 * s1


def canonical(edge_index: np.ndarray) -> np.ndarray:
    e = np.asarray(edge_index)
    order = np.lexsort((e[0], e[1]))
    return e[:, order]


def knn_rows(edge_index: np.ndarray, nt: int, k: int) -> np.ndarray:
    """[nt,k] source ids from a target-major [2, nt*k] edge list."""
    e = np.asarray(edge_index)
    assert e.shape == (2, nt * k)
    assert np.array_equal(e[1], np.repeat(np.arange(nt), k))
    return e[0].reshape(nt, k)


# With at most 11 sources scikit-learn answers by brute force (sklearn/neighbors/_base.py:622-655: n_neighbors
# >= n_samples // 2 with the default n_neighbors = 5) and evaluates the haversine metric on the float32 inputs
# in float32: its ranking is only float32-accurate there, so members that differ by less than this count as tied.
BRUTE_MAX_SOURCES = 11
BRUTE_TIE_RTOL = 2e-6


def compare_knn(src, tgt, ids_a: np.ndarray, ids_b: np.ndarray, tie_rtol: float | None = None) -> dict:
    """Row-wise set comparison of two [nt,k] neighbour tables.  Rows whose sets differ must differ
    only in members whose float64 rdist ties with the k-th smallest (within TIE_RTOL; BRUTE_TIE_RTOL
    for source sets small enough for the reference's float32 brute-force path)."""
    if tie_rtol is None:
        tie_rtol = BRUTE_TIE_RTOL if len(src) <= BRUTE_MAX_SOURCES else TIE_RTOL
    nt, k = ids_a.shape
    sa, sb = np.sort(ids_a, axis=1), np.sort(ids_b, axis=1)
    diff_rows = np.where((sa != sb).any(axis=1))[0]
    bad = []
    for t in diff_rows:
        a, b = set(ids_a[t].tolist()), set(ids_b[t].tolist())
        ra = rdist64(tgt[t][None, :], src[ids_a[t]])
        kth = ra.max()
        odd = np.array(sorted(a ^ b))
        ro = rdist64(tgt[t][None, :], src[odd])
        if not np.all(np.abs(ro - kth) <= tie_rtol * max(kth, 1e-300)):
            bad.append(int(t))
    return {"rows": nt, "tie_rows": int(len(diff_rows) - len(bad)), "bad_rows": bad}


def compare_edge_sets(src, tgt, e_a: np.ndarray, e_b: np.ndarray, radius: float) -> dict:
    """Set comparison of two radius edge lists; differing edges must sit on the radius boundary."""
    ns = len(src)
    ka = np.unique(e_a[1].astype(np.int64) * ns + e_a[0])
    kb = np.unique(e_b[1].astype(np.int64) * ns + e_b[0])
    assert len(ka) == e_a.shape[1] and len(kb) == e_b.shape[1], "duplicate edges"
    odd = np.setxor1d(ka, kb)
    bad = 0
    if len(odd):
        t, s = odd // ns, odd % ns
        r = rdist64(tgt[t], src[s])
        thr = np.sin(0.5 * radius) ** 2
        bad = int(np.sum(np.abs(r - thr) > TIE_RTOL * thr))
    return {"edges_a": int(len(ka)), "edges_b": int(len(kb)), "boundary": int(len(odd) - bad), "bad": bad}
