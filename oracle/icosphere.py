"""TEST INFRASTRUCTURE ONLY -- numpy restatement of ``trimesh.creation.icosphere``.

PARITY UNPINNED for vertex coordinates and pre-sort vertex order: the reference
obtains its icosphere from the third-party library trimesh (pinned
``trimesh>=4.1`` in ``/root/reference/pyproject.toml:55``; call sites
``/root/reference/src/anemoi/graphs/generate/tri_icosahedron.py:121`` and
``:173-174``), which is not installed in the build container and cannot be
fetched.  This file restates trimesh's published algorithm:

* ``icosahedron()``: the 12 vertices ``(-1,t,0) (1,t,0) (-1,-t,0) (1,-t,0)
  (0,-1,t) (0,1,t) (0,-1,-t) (0,1,-t) (t,0,-1) (t,0,1) (-t,0,-1) (-t,0,1)``
  divided by ``sqrt(2+t)``, ``t=(1+sqrt 5)/2``, and its 20 faces.
* ``subdivide()``: one midpoint per unique undirected edge, appended *after*
  the existing vertices (so level-l vertices are a prefix of level-(l+1)
  vertices); unique edges are enumerated in increasing order of the packed key
  ``(max_vertex << 32) | min_vertex``; each face ``(a,b,c)`` with midpoints
  ``(ab,bc,ca)`` becomes ``(a,ab,ca) (ab,b,bc) (ca,bc,c) (ab,bc,ca)``.
* after every subdivision *all* vertices are pushed to radius 1 in float64:
  ``v += (v/|v|) * (1-|v|)``.

What pins it: vertex/face/edge counts per level
(``/root/reference/docs/graphs/node_coordinates/tri_nodes.csv:2-8``,
``/root/reference/docs/graphs/edges/tri_refined_edges.csv:2-8``), the (162, 2)
shape test (``/root/reference/tests/nodes/test_tri_nodes.py:27-32``) and the
edge counts of the documented O96 <-> TriNodes(5) example
(``/root/reference/docs/_static/hetero_data_graph.txt:12-29``).
"""
from __future__ import annotations

import numpy as np

_T = (1.0 + 5.0**0.5) / 2.0

_ICO_VERTICES = np.array(
    [
        [-1, _T, 0], [1, _T, 0], [-1, -_T, 0], [1, -_T, 0],
        [0, -1, _T], [0, 1, _T], [0, -1, -_T], [0, 1, -_T],
        [_T, 0, -1], [_T, 0, 1], [-_T, 0, -1], [-_T, 0, 1],
    ],
    dtype=np.float64,
)

_ICO_FACES = np.array(
    [
        [0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11],
        [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6], [7, 1, 8],
        [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9],
        [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1],
    ],
    dtype=np.int64,
)


def icosahedron() -> tuple[np.ndarray, np.ndarray]:
    return _ICO_VERTICES / np.sqrt(2.0 + _T), _ICO_FACES.copy()


def face_edges(faces: np.ndarray) -> np.ndarray:
    """The 3 undirected edges (min, max) of every face: (a,b) (b,c) (c,a)."""
    return np.sort(faces[:, [0, 1, 1, 2, 2, 0]].reshape(-1, 2), axis=1)


def unique_edges(faces: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """Unique undirected edges in (max, min) key order and the per-half-edge inverse."""
    e = face_edges(faces)
    key = (e[:, 1].astype(np.uint64) << np.uint64(32)) | e[:, 0].astype(np.uint64)
    _, first, inverse = np.unique(key, return_index=True, return_inverse=True)
    return e[first], inverse


def subdivide(vertices: np.ndarray, faces: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    edges, inverse = unique_edges(faces)
    mid = (vertices[edges[:, 0]] + vertices[edges[:, 1]]) / 2.0
    mid_idx = inverse.reshape(-1, 3) + len(vertices)
    f = np.column_stack(
        [
            faces[:, 0], mid_idx[:, 0], mid_idx[:, 2],
            mid_idx[:, 0], faces[:, 1], mid_idx[:, 1],
            mid_idx[:, 2], mid_idx[:, 1], faces[:, 2],
            mid_idx[:, 0], mid_idx[:, 1], mid_idx[:, 2],
        ]
    ).reshape(-1, 3)
    return np.vstack([vertices, mid]), f


def icosphere(subdivisions: int) -> tuple[np.ndarray, np.ndarray]:
    """Unit icosphere: float64 vertices [V,3], int64 faces [F,3]."""
    v, f = icosahedron()
    for _ in range(subdivisions):
        v, f = subdivide(v, f)
        scalar = np.sqrt(v[:, 0] ** 2 + v[:, 1] ** 2 + v[:, 2] ** 2)
        unit = v / scalar[:, None]
        v = v + unit * (1.0 - scalar)[:, None]
    return v, f


def latlon_rad(xyz: np.ndarray) -> np.ndarray:
    """``cartesian_to_latlon_rad`` (/root/reference/src/anemoi/graphs/generate/transforms.py:34-52)."""
    lat = np.arcsin(xyz[:, 2] / (xyz[:, 0] ** 2 + xyz[:, 1] ** 2 + xyz[:, 2] ** 2))
    lon = np.arctan2(xyz[:, 1], xyz[:, 0])
    return np.array((lat, lon), dtype=np.float32).transpose()
