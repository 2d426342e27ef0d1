"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's hot path.

This module is the checker, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  It restates, call for call, what the reference does on the CPU with the
same third-party libraries the reference uses (scikit-learn, scipy, numpy,
networkx -- all present in the image, so this file also runs on the GPU box,
where ``/root/reference`` does not exist).

Pinned against outputs of the unmodified reference classes run in the build
container (``oracle/make_golden.py`` -> ``tests/golden/*.npz``; checked by
``tests/test_oracle_golden.py``).  Exception: everything that goes through
``oracle.icosphere`` (TriNodes coordinates, MultiScaleEdges) is PARITY UNPINNED
with respect to trimesh's vertex coordinates, see that module's header.
"""
from __future__ import annotations

import numpy as np

EARTH_RADIUS = 6371.0  # /root/reference/src/anemoi/graphs/__init__.py:10
N_JOBS = 4  # hard-coded in the reference: edges/builder.py:259,364 ; utils.py:37


def _nearest_neighbors(coords: np.ndarray):
    from sklearn.neighbors import NearestNeighbors

    nn = NearestNeighbors(metric="haversine", n_jobs=N_JOBS)
    nn.fit(coords)
    return nn


def knn_edges(src: np.ndarray, tgt: np.ndarray, k: int, return_dist: bool = False):
    """KNNEdges.get_adjacency_matrix + BaseEdgeBuilder.get_edge_index.

    /root/reference/src/anemoi/graphs/edges/builder.py:259-265 (sklearn calls) and
    :86-87 (``edge_index = [adj.col (source), adj.row (target)]`` as int32).
    """
    adj = _nearest_neighbors(src).kneighbors_graph(tgt, n_neighbors=k, mode="distance").tocoo()
    edge_index = np.stack([adj.col, adj.row], axis=0).astype(np.int32)
    return (edge_index, adj.data) if return_dist else edge_index


def grid_reference_distance(coords: np.ndarray) -> float:
    """utils.get_grid_reference_distance, /root/reference/src/anemoi/graphs/utils.py:61-63."""
    dists, _ = _nearest_neighbors(coords).kneighbors(coords, n_neighbors=2, return_distance=True)
    return dists[dists > 0].max()


def cutoff_radius(tgt: np.ndarray, cutoff_factor: float) -> float:
    """CutOffEdges.get_cutoff_radius, edges/builder.py:330-334 (reference distance of the TARGET nodes)."""
    return grid_reference_distance(tgt) * cutoff_factor


def radius_edges(src: np.ndarray, tgt: np.ndarray, radius: float) -> np.ndarray:
    """CutOffEdges.get_adjacency_matrix, edges/builder.py:364-366, then :86-87."""
    adj = _nearest_neighbors(src).radius_neighbors_graph(tgt, radius=radius).tocoo()
    return np.stack([adj.col, adj.row], axis=0).astype(np.int32)


def cutoff_edges(src: np.ndarray, tgt: np.ndarray, cutoff_factor: float):
    radius = cutoff_radius(tgt, cutoff_factor)
    return radius_edges(src, tgt, radius), radius


def apply_mask(coords: np.ndarray, mask: np.ndarray | None):
    """NodeMaskingMixin.get_node_coordinates, edges/builder.py:162-174."""
    return coords if mask is None else coords[np.asarray(mask).squeeze().astype(bool)]


def undo_masking(edge_index: np.ndarray, src_mask, tgt_mask) -> np.ndarray:
    """NodeMaskingMixin.undo_masking, edges/builder.py:176-193 (compact ids -> full ids)."""
    out = edge_index.copy()
    if tgt_mask is not None:
        out[1] = np.where(np.asarray(tgt_mask).squeeze())[0][edge_index[1]]
    if src_mask is not None:
        out[0] = np.where(np.asarray(src_mask).squeeze())[0][edge_index[0]]
    return out


def concat_edges(e1: np.ndarray, e2: np.ndarray) -> np.ndarray:
    """utils.concat_edges, utils.py:81: unique columns in lexicographic (row0,row1) order."""
    return np.unique(np.concatenate([e1, e2], axis=1), axis=1)


def canonical_sort(edge_index: np.ndarray) -> np.ndarray:
    """Canonical edge order used for every parity comparison: by target id, then source id."""
    order = np.lexsort((edge_index[0], edge_index[1]))
    return edge_index[:, order]


# ----------------------------------------------------------------------------- attributes


def haversine_distance(source_coords: np.ndarray, target_coords: np.ndarray) -> np.ndarray:
    """utils.haversine_distance, utils.py:99-103 (float32 in -> float32 arithmetic)."""
    dlat = target_coords[:, 0] - source_coords[:, 0]
    dlon = target_coords[:, 1] - source_coords[:, 1]
    a = np.sin(dlat / 2) ** 2 + np.cos(source_coords[:, 0]) * np.cos(target_coords[:, 0]) * np.sin(dlon / 2) ** 2
    return 2 * np.arctan2(np.sqrt(a), np.sqrt(1 - a))


def _latlon_rad_to_cartesian(loc) -> np.ndarray:
    """generate/transforms.py:106-110."""
    latr, lonr = loc[0], loc[1]
    return np.array((np.cos(latr) * np.cos(lonr), np.cos(latr) * np.sin(lonr), np.sin(latr))).T


def _direction_vec(points: np.ndarray, reference: np.ndarray, epsilon: float = 10e-11) -> np.ndarray:
    """generate/transforms.py:133-140 (note: nudges ``points`` in place, as the reference does)."""
    v = np.cross(points, reference)
    vnorm1 = np.power(v, 2).sum(axis=-1)
    redo_idx = np.where(vnorm1 < epsilon)[0]
    if len(redo_idx) > 0:
        points[redo_idx] += epsilon
        v = np.cross(points, reference)
        vnorm1 = np.power(v, 2).sum(axis=-1)
    return v.T / np.sqrt(vnorm1)


def edge_directions(src_x: np.ndarray, tgt_x: np.ndarray, edge_index: np.ndarray, rotated: bool = True) -> np.ndarray:
    """EdgeDirection.get_raw_values (edges/attributes.py:100-104) -> directional_edge_features
    (edges/directional.py:89-94) -> compute_directions (:57-65) -> get_rotation_from_unit_vecs (:34-37)."""
    from scipy.spatial.transform import Rotation

    loc1 = src_x[edge_index[0]].T
    loc2 = tgt_x[edge_index[1]].T
    if not rotated:
        return (loc2 - loc1).T
    pole = np.array([0, 0, 1])
    loc1_xyz = _latlon_rad_to_cartesian(loc1)
    loc2_xyz = _latlon_rad_to_cartesian(loc2)
    v_unit = _direction_vec(loc2_xyz, pole)
    theta = np.arccos(np.dot(loc2_xyz, pole))
    rot = Rotation.from_rotvec(np.transpose(v_unit * theta))
    direction = _direction_vec(rot.apply(loc1_xyz), pole)
    direction = direction / np.sqrt(np.power(direction, 2).sum(axis=0))
    return direction[:2].T


def edge_lengths(src_x: np.ndarray, tgt_x: np.ndarray, edge_index: np.ndarray) -> np.ndarray:
    """EdgeLength.get_raw_values, edges/attributes.py:144-148."""
    return haversine_distance(src_x[edge_index[0]], tgt_x[edge_index[1]])


def normalise(values: np.ndarray, norm: str | None) -> np.ndarray:
    """NormaliserMixin.normalise, /root/reference/src/anemoi/graphs/normalise.py:36-55."""
    if norm is None:
        return values
    if norm == "l1":
        return values / np.sum(values)
    if norm == "l2":
        return values / np.linalg.norm(values)
    if norm == "unit-max":
        return values / np.amax(values)
    if norm == "unit-range":
        return (values - np.amin(values)) / (np.amax(values) - np.amin(values))
    if norm == "unit-std":
        std = np.std(values)
        return values if std == 0 else values / std
    raise ValueError(f'Attribute normalisation "{norm}" is not valid.')


def post_process(values: np.ndarray, norm: str | None, invert: bool = False) -> np.ndarray:
    """BaseEdgeAttribute.post_process (edges/attributes.py:33-40) and EdgeLength.post_process (:150-157)."""
    if values.ndim == 1:
        values = values[:, np.newaxis]
    out = np.asarray(normalise(values, norm)).astype(np.float32)
    return 1 - out if invert else out


# ----------------------------------------------------------------------------- node attributes


def spherical_area_weights(coords: np.ndarray, norm: str | None = None, fill_value: float = 0.0,
                           dtype: str = "float32", raw: bool = False) -> np.ndarray:
    """SphericalAreaWeights.get_raw_values (nodes/attributes.py:198-221) with radius 1 and centre 0, then
    BaseNodeAttribute.post_process (:45-52).  ``raw`` returns the float64 areas before post-processing."""
    from scipy.spatial import SphericalVoronoi

    latitudes, longitudes = coords[:, 0], coords[:, 1]
    points = _latlon_rad_to_cartesian((np.asarray(latitudes), np.asarray(longitudes)))
    sv = SphericalVoronoi(points, 1.0, np.array([0, 0, 0]))
    mask = np.array([bool(i) for i in sv.regions])
    sv.regions = [region for region in sv.regions if region]
    area_weights = sv.calculate_areas()
    result = np.ones(points.shape[0]) * fill_value
    result[mask] = area_weights
    if raw:
        return result
    return np.asarray(normalise(result[:, np.newaxis], norm)).astype(dtype)


def uniform_weights(num_nodes: int, norm: str | None = None, dtype: str = "float32") -> np.ndarray:
    """UniformWeights.get_raw_values (nodes/attributes.py:85-100) + post_process."""
    return np.asarray(normalise(np.ones(num_nodes)[:, np.newaxis], norm)).astype(dtype)


# ----------------------------------------------------------------------------- icosahedral nodes / edges


def coordinates_ordering_reference(coords: np.ndarray) -> np.ndarray:
    """generate/utils.py:30-32 verbatim semantics: two default (unstable) argsorts.

    The order among equal latitudes depends on numpy's sort implementation."""
    index_latitude = np.argsort(coords[:, 1])
    index_longitude = np.argsort(coords[index_latitude][:, 0])[::-1]
    return np.arange(coords.shape[0])[index_latitude][index_longitude]


def coordinates_ordering_canonical(coords: np.ndarray) -> np.ndarray:
    """What generate/utils.py:30-32 yields when both sorts are stable: latitude descending,
    ties by longitude descending, remaining ties by descending vertex id.  Deterministic on
    every platform; the product adopts this order."""
    n = coords.shape[0]
    first = np.argsort(coords[:, 1], kind="stable")
    second = np.argsort(coords[first][:, 0], kind="stable")[::-1]
    return np.arange(n)[first][second]


def tri_nodes(resolution: int, canonical: bool = True):
    """create_tri_nodes, generate/tri_icosahedron.py:47-58 -> (x float32 [V,2] ordered, node_ordering)."""
    from . import icosphere as ico

    v, _ = ico.icosphere(resolution)
    coords = ico.latlon_rad(v)
    order = coordinates_ordering_canonical(coords) if canonical else coordinates_ordering_reference(coords)
    return coords[order], order


EARTH_RADIUS = 6371.0  # km, anemoi/graphs/__init__.py


def area_mask(reference_coords: np.ndarray, coords: np.ndarray, margin_radius_km: float) -> np.ndarray:
    """KNNAreaMaskBuilder.fit_coords + get_mask (generate/masks.py:72-99)."""
    nn = _nearest_neighbors(np.asarray(reference_coords))
    dists, _ = nn.kneighbors(np.asarray(coords), n_neighbors=1)
    return dists[:, 0] * EARTH_RADIUS <= margin_radius_km


def limited_area_tri_nodes(resolution: int, reference_coords: np.ndarray, margin_radius_km: float):
    """create_tri_nodes with an area mask (generate/tri_icosahedron.py:47-58), canonical order
    -> (x float32 [n,2], node_ordering = vertex ids, all float64 vertex coordinates)."""
    from . import icosphere as ico

    coords = ico.latlon_rad(ico.icosphere(resolution)[0])
    order = coordinates_ordering_canonical(coords)
    mask = area_mask(reference_coords, coords, margin_radius_km)
    order = order[mask[order]]
    return coords[order].astype(np.float32), order, coords


def stretched_tri_nodes(base_resolution: int, lam_resolution: int, reference_coords: np.ndarray,
                        margin_radius_km: float):
    """create_stretched_tri_nodes (generate/tri_icosahedron.py:89-105), canonical order
    -> (x float32 [n,2], node_ordering = ids into the concatenated coordinates, those coordinates)."""
    from . import icosphere as ico

    base = ico.latlon_rad(ico.icosphere(base_resolution)[0])
    lam = ico.latlon_rad(ico.icosphere(lam_resolution)[0])
    base_out = ~area_mask(reference_coords, base, margin_radius_km)
    lam_in = area_mask(reference_coords, lam, margin_radius_km)
    coords = np.concatenate([base[base_out], lam[lam_in]])
    order = coordinates_ordering_canonical(coords)
    return coords[order].astype(np.float32), order, coords


def multiscale_edges_one_hop(resolutions: list[int], node_ordering: np.ndarray) -> np.ndarray:
    """The x_hops = 1 case of ``multiscale_edges`` on the complete icosphere (TriNodes), in numpy only, for
    meshes too large for the per-vertex networkx loop (r7: 163,842 vertices).

    With one hop the neighbours of a level-l vertex are the other ends of its level-l mesh edges
    (generate/tri_icosahedron.py:212-224: ego_graph(radius=1, center=False) of the graph built from
    ``edges_unique``), so a level contributes both directions of every unique mesh edge; the level's vertices
    map to the final mesh's nodes with the same vertex id (:185: nearest final-mesh node of a vertex that is
    itself a vertex of the final mesh), and node ids are positions in ``node_ordering`` (edges/builder.py:453).
    The DiGraph merges the levels (:187).  ``tests/test_oracle_golden.py`` proves this function equal to the
    networkx restatement on the meshes that one handles.  Returns int32 [2,E] sorted by (target, source)."""
    from . import icosphere as ico

    node_ordering = np.asarray(node_ordering)
    rank = np.empty(len(node_ordering), dtype=np.int64)
    rank[node_ordering] = np.arange(len(node_ordering))
    pairs = []
    for resolution in sorted(set(int(r) for r in resolutions)):
        _, f = ico.icosphere(resolution)
        e = np.unique(ico.face_edges(f), axis=0)
        pairs.append(np.concatenate([rank[e], rank[e[:, ::-1]]], axis=0))
    pairs = np.unique(np.concatenate(pairs, axis=0), axis=0)  # columns (source, target)
    return canonical_sort(pairs.T.astype(np.int32))


def multiscale_edges_hops(resolutions: list[int], x_hops: int, node_ordering: np.ndarray) -> np.ndarray:
    """``multiscale_edges`` on the complete icosphere (TriNodes) for any ``x_hops``, without the per-vertex
    networkx loop, for meshes too large for it (r6, r7).

    ``nx.ego_graph(mesh, i, radius=x_hops, center=False)`` (generate/tri_icosahedron.py:218-222) is the set of
    vertices at most ``x_hops`` mesh edges from i, without i: with A the boolean adjacency matrix of the level's
    unique edges, the non-zero pattern of (A + I)^x_hops with the diagonal removed.  The matrix power is taken
    by repeated sparse products (frontier expansion); a level's vertices map to the final mesh's nodes with the
    same vertex id (:185) and node ids are positions in ``node_ordering`` (edges/builder.py:453); the DiGraph
    merges the levels (:187).  ``tests/test_oracle_golden.py`` proves this function equal to the networkx
    restatement on the meshes that one handles.  Returns int32 [2,E] sorted by (target, source)."""
    import scipy.sparse as sp

    from . import icosphere as ico

    node_ordering = np.asarray(node_ordering)
    rank = np.empty(len(node_ordering), dtype=np.int64)
    rank[node_ordering] = np.arange(len(node_ordering))
    pairs = []
    for resolution in sorted(set(int(r) for r in resolutions)):
        v, f = ico.icosphere(resolution)
        e = np.unique(ico.face_edges(f), axis=0)
        n = len(v)
        ones = np.ones(len(e), dtype=np.int8)
        step = (sp.coo_matrix((ones, (e[:, 0], e[:, 1])), shape=(n, n))
                + sp.coo_matrix((ones, (e[:, 1], e[:, 0])), shape=(n, n)) + sp.identity(n, dtype=np.int8)).tocsr()
        reach = step.copy()
        for _ in range(x_hops - 1):
            reach = reach @ step
            reach.data[:] = 1  # (boolean: the counts of walks are of no interest and would overflow)
        reach = reach.tocoo()
        keep = reach.row != reach.col
        # (neighbour -> node), as create_node_neighbours_list lists them (:298-310); the relation is symmetric
        pairs.append(np.stack([rank[reach.col[keep]], rank[reach.row[keep]]], axis=1))
    pairs = np.unique(np.concatenate(pairs, axis=0), axis=0)  # columns (source, target)
    return canonical_sort(pairs.T.astype(np.int32))


def multiscale_edges(resolutions: list[int], x_hops: int, node_ordering: np.ndarray, node_coords: np.ndarray | None = None,
                     mask_reference: np.ndarray | None = None, margin_radius_km: float = 100.0) -> np.ndarray:
    """add_edges_to_nx_graph (generate/tri_icosahedron.py:165-188), get_neighbours_within_hops
    (:212-224), create_node_neighbours_list (:298-310) and MultiScaleEdges.get_adjacency_matrix
    (edges/builder.py:453) restated with networkx, as the reference uses it.

    ``node_coords``: float64 coordinates indexed by node id (default: the vertices of the finest
    icosphere, the TriNodes case).  ``mask_reference``/``margin_radius_km``: the area mask applied to
    the vertices of every level (LimitedAreaTriNodes: the builder's reference nodes and margin;
    StretchedTriNodes: the graph's own float32 node coordinates with 1 km, edges/builder.py:422-432).

    Returns int32 [2,E]; node ids are positions in ``node_ordering`` (the graph's insertion
    order, edges/builder.py:453 ``nx.to_scipy_sparse_array``)."""
    import networkx as nx
    from sklearn.neighbors import BallTree

    from . import icosphere as ico

    if node_coords is None:
        node_coords = ico.latlon_rad(ico.icosphere(max(resolutions))[0])
    graph = nx.DiGraph()
    for node_id in node_ordering:
        graph.add_node(int(node_id))
    sorted_nodes = sorted(graph.nodes)
    tree = BallTree(node_coords[sorted_nodes], metric="haversine")
    for resolution in resolutions:
        v, f = ico.icosphere(resolution)
        r_coords = ico.latlon_rad(v)
        edges = np.unique(ico.face_edges(f), axis=0)
        if mask_reference is not None:
            valid = np.where(area_mask(mask_reference, r_coords, margin_radius_km))[0]
            edges = edges[np.isin(edges, valid).all(axis=1)]
        else:
            valid = range(len(v))
        mesh = nx.from_edgelist(edges)
        neighbours = {
            int(i): set(nx.ego_graph(mesh, i, radius=x_hops, center=False) if i in mesh else []) for i in valid
        }
        _, mapping = tree.query(r_coords, k=1)
        pairs = [
            (sorted_nodes[mapping[nb][0]], sorted_nodes[mapping[node][0]])
            for node, nbs in neighbours.items()
            for nb in nbs
            if node != nb
        ]
        graph.add_edges_from(pairs)
    adj = nx.to_scipy_sparse_array(graph, format="coo")
    return np.stack([adj.col, adj.row], axis=0).astype(np.int32)
