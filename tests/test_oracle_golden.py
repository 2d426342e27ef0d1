"""CPU: the oracle (oracle/reference_path.py) reproduces the outputs the UNMODIFIED reference classes
produced in the build container (tests/golden/*.npz, written by oracle/make_golden.py)."""
import numpy as np
import pytest

from anemoi_graphs_amd import grids
from oracle import reference_path as rp
from tests import helpers as h

NORMS = [None, "l1", "l2", "unit-max", "unit-range", "unit-std"]


def _tag(norm):
    return "none" if norm is None else norm.replace("-", "_")


def test_knn_and_cutoff_random(golden):
    g = golden("random_sphere")
    for k in (1, 3, 8):
        assert np.array_equal(rp.knn_edges(g["src"], g["tgt"], k), g[f"knn{k}_edge_index"])
    for f, tag in ((0.6, "06"), (1.3, "13")):
        e, radius = rp.cutoff_edges(g["tgt"], g["src"], f)
        assert radius == float(g[f"cutoff{tag}_radius"])
        assert np.array_equal(h.canonical(e), h.canonical(g[f"cutoff{tag}_edge_index"]))
    assert rp.grid_reference_distance(g["src"]) == float(g["refdist_src"])


def test_masks(golden):
    g = golden("masks")
    s, t = rp.apply_mask(g["src"], g["src_mask"]), rp.apply_mask(g["tgt"], g["tgt_mask"])
    e = rp.undo_masking(rp.knn_edges(s, t, 4), g["src_mask"], g["tgt_mask"])
    assert np.array_equal(e, g["knn4_edge_index"])
    # CutOffEdges computes the radius on ALL target nodes (edges/builder.py:338, mask_attr never passed)
    radius = rp.cutoff_radius(g["src"], 0.8)
    assert radius == float(g["cutoff08_radius"])
    e = rp.undo_masking(rp.radius_edges(t, s, radius), g["tgt_mask"], g["src_mask"])
    assert np.array_equal(h.canonical(e), h.canonical(g["cutoff08_edge_index"]))


@pytest.mark.parametrize("case,src_key,tgt_key,prefix,ekey", [
    ("random_sphere", "src", "tgt", "knn3_", "knn3_edge_index"),
    ("random_sphere", "tgt", "src", "cutoff06_", "cutoff06_edge_index"),
    ("conftest_grid", "x", "x", "knn3_", "knn3_edge_index"),
    ("conftest_grid_f32", "x", "x", "knn3_", "knn3_edge_index"),
])
def test_attributes(golden, case, src_key, tgt_key, prefix, ekey):
    g = golden(case)
    src, tgt, e = g[src_key], g[tgt_key], g[ekey]
    for norm in NORMS:
        np.testing.assert_array_equal(rp.post_process(rp.edge_lengths(src, tgt, e), norm), g[f"{prefix}len_{_tag(norm)}"])
        np.testing.assert_array_equal(rp.post_process(rp.edge_directions(src, tgt, e), norm), g[f"{prefix}dir_{_tag(norm)}"])
    np.testing.assert_array_equal(rp.post_process(rp.edge_lengths(src, tgt, e), "unit-max", True), g[f"{prefix}len_inv_unit_max"])
    np.testing.assert_array_equal(rp.post_process(rp.edge_directions(src, tgt, e, False), None), g[f"{prefix}dir_plain_none"])


def test_concat_edges(golden):
    g = golden("concat_edges")
    assert np.array_equal(rp.concat_edges(g["e1"], g["e2"]), g["e12"])
    assert rp.concat_edges(g["e1"], g["e2"]).tolist() == [[0, 1, 2, 3, 4], [-1, -2, -3, -4, -5]]  # tests/test_utils.py:17
    assert np.array_equal(rp.concat_edges(g["r1"], g["r2"]), g["r12"])


def _relabel(x_ref, edge_ref, x_mine):
    lut = {(float(a), float(b)): i for i, (a, b) in enumerate(x_mine)}
    m = np.array([lut[(float(a), float(b))] for a, b in x_ref])
    return m[edge_ref]


@pytest.mark.parametrize("r,hops", [(0, 1), (1, 1), (2, 1), (2, 2), (3, 1), (3, 3)])
def test_tri_nodes_and_multiscale(golden, r, hops):
    """The reference orders equal-latitude nodes arbitrarily (unstable sorts); same node SET, and the
    same edges once the reference's nodes are matched to the canonical order by coordinates."""
    g = golden("tri_multiscale")
    x, order = rp.tri_nodes(r)
    ref_x = g[f"r{r}_x"]
    assert x.shape == ref_x.shape
    assert np.array_equal(x[np.lexsort((x[:, 1], x[:, 0]))], ref_x[np.lexsort((ref_x[:, 1], ref_x[:, 0]))])
    assert np.array_equal(ref_x[:, 0], x[:, 0])  # latitudes descend identically
    e = rp.multiscale_edges(list(range(r + 1)), hops, order)
    want = _relabel(ref_x, g[f"r{r}_h{hops}_edge_index"], x)
    assert np.array_equal(h.canonical(e), h.canonical(want))


@pytest.mark.parametrize("r,levels", [(0, [0]), (1, [0, 1]), (2, [0, 1, 2]), (3, [0, 1, 2, 3]), (3, [0, 2, 3]), (3, [3]),
                                      (4, [0, 1, 2, 3, 4]), (4, [1, 4])])
def test_one_hop_oracle_equals_networkx_restatement(golden, r, levels):
    """The numpy-only x_hops = 1 oracle (used for the r6 / r7 meshes the networkx loop cannot handle in test
    time) is the same function as the networkx restatement wherever that one runs, and through it equal to
    the reference's own output (golden fixtures, r <= 3)."""
    order = rp.tri_nodes(r)[1]
    fast = rp.multiscale_edges_one_hop(levels, order)
    slow = rp.multiscale_edges(levels, 1, order)
    assert np.array_equal(fast, h.canonical(slow))
    assert fast.shape[1] == sum(60 * 4**l for l in levels)
    if levels == list(range(r + 1)) and r <= 3:
        g = golden("tri_multiscale")
        want = _relabel(g[f"r{r}_x"], g[f"r{r}_h1_edge_index"], rp.tri_nodes(r)[0])
        assert np.array_equal(fast, h.canonical(want))


def _same_nodes(x, ref_x):
    assert x.shape == ref_x.shape
    assert np.array_equal(x[np.lexsort((x[:, 1], x[:, 0]))], ref_x[np.lexsort((ref_x[:, 1], ref_x[:, 0]))])
    assert np.array_equal(ref_x[:, 0], x[:, 0])  # latitudes descend identically


@pytest.mark.parametrize("r,hops", [(4, 1), (5, 1), (4, 2)])
def test_limited_area_tri_nodes(golden, r, hops):
    """LimitedAreaTriNodes + MultiScaleEdges of the reference (from_refined_icosahedron.py:119-131,
    generate/tri_icosahedron.py:47-58,165-188) against the restatement."""
    g = golden("lam_stretched")
    tag = f"lam_r{r}_h{hops}"
    margin = float(g[f"{tag}_margin"])
    x, order, coords = rp.limited_area_tri_nodes(r, g["area"], margin)
    ref_x = g[f"{tag}_x"]
    _same_nodes(x, ref_x)
    e = rp.multiscale_edges(list(range(r + 1)), hops, order, node_coords=coords, mask_reference=g["area"],
                            margin_radius_km=margin)
    want = _relabel(ref_x, g[f"{tag}_edge_index"], x)
    assert np.array_equal(h.canonical(e), h.canonical(want))


@pytest.mark.parametrize("rb,rl,hops", [(2, 4, 1), (1, 4, 1), (2, 4, 2)])
def test_stretched_tri_nodes(golden, rb, rl, hops):
    """StretchedTriNodes + MultiScaleEdges of the reference (from_refined_icosahedron.py:179-191,
    generate/tri_icosahedron.py:89-105, edges/builder.py:422-432) against the restatement."""
    g = golden("lam_stretched")
    tag = f"stretched_r{rb}_{rl}_h{hops}"
    margin = float(g[f"{tag}_margin"])
    x, order, coords = rp.stretched_tri_nodes(rb, rl, g["area"], margin)
    ref_x = g[f"{tag}_x"]
    _same_nodes(x, ref_x)
    e = rp.multiscale_edges(list(range(rl + 1)), hops, order, node_coords=coords, mask_reference=x,
                            margin_radius_km=1.0)
    want = _relabel(ref_x, g[f"{tag}_edge_index"], x)
    assert np.array_equal(h.canonical(e), h.canonical(want))


def test_reference_ordering_restated(golden):
    g = golden("tri_multiscale")
    from oracle import icosphere as ico

    for r in (0, 1, 2, 3):
        coords = ico.latlon_rad(ico.icosphere(r)[0])
        assert np.array_equal(rp.coordinates_ordering_reference(coords), g[f"r{r}_node_ordering"])


def test_documented_counts():
    """docs/_static/hetero_data_graph.txt:12-29 of the reference: O96 <-> TriNodes(5)."""
    data = grids.octahedral(96)
    assert data.shape == (40320, 2)
    hidden, order = rp.tri_nodes(5)
    assert hidden.shape == (10242, 2)
    e, _ = rp.cutoff_edges(data, hidden, 0.6)
    assert e.shape[1] == 62980
    assert rp.multiscale_edges(list(range(4)), 1, rp.tri_nodes(3)[1]).shape[1] == 5100


def test_grids():
    assert grids.octahedral(1280).shape == (6599680, 2)
    assert grids.regular(0.25).shape == (1038240, 2)
    x = grids.octahedral(8)
    assert np.all(np.diff(x[:, 0]) <= 0)


@pytest.mark.parametrize("name", ["o32", "tri3", "random", "aniso"])
def test_area_weights_oracle_matches_reference(golden, name):
    """SphericalAreaWeights of the reference (scipy's SphericalVoronoi) against the oracle's restatement."""
    gold = golden("area_weights")
    x = gold[f"{name}_x"]
    np.testing.assert_allclose(rp.spherical_area_weights(x, raw=True), gold[f"{name}_area_raw"], rtol=1e-12)
    for norm in [None, "l1", "l2", "unit-max", "unit-range", "unit-std"]:
        tag = "none" if norm is None else norm.replace("-", "_")
        want = gold[f"{name}_area_{tag}"]
        got = rp.spherical_area_weights(x, norm)
        assert got.dtype == np.float32 and got.shape == want.shape
        np.testing.assert_allclose(got, want, rtol=1.2e-7, atol=1e-10 if norm == "unit-range" else 0)


def test_uniform_weights_oracle_matches_reference(golden):
    gold = golden("area_weights")
    n = gold["o32_x"].shape[0]
    np.testing.assert_array_equal(rp.uniform_weights(n, "l1"), gold["o32_uniform_l1"])
    np.testing.assert_array_equal(rp.uniform_weights(n), gold["o32_uniform_none"])


@pytest.mark.parametrize("r,hops,levels", [(2, 1, None), (3, 2, None), (4, 2, None), (3, 3, None), (4, 3, [1, 3, 4]),
                                           (2, 4, None)])
def test_frontier_expansion_oracle_equals_networkx_restatement(r, hops, levels):
    """The sparse-matrix oracle for x_hops >= 2 (used on r6 / r7, where the per-vertex networkx loop of the
    restatement takes hours) against that restatement on the meshes it handles; x_hops = 1 also against the
    one-hop oracle."""
    levels = list(range(r + 1)) if levels is None else levels
    order = rp.tri_nodes(r)[1]
    got = rp.multiscale_edges_hops(levels, hops, order)
    want = rp.multiscale_edges(levels, hops, order)
    assert np.array_equal(got, h.canonical(want))
    if hops == 1:
        assert np.array_equal(got, rp.multiscale_edges_one_hop(levels, order))
