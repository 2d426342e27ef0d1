"""GPU parity of the k-NN / reference-distance / radius kernels against the CPU oracle
(scikit-learn called exactly as the reference calls it) and the golden fixtures."""
import numpy as np
import pytest
import torch

from anemoi_graphs_amd import grids, ops
from oracle import reference_path as rp
from tests import helpers as h

pytestmark = pytest.mark.gpu


def _knn_case(src, tgt, k, device):
    ids, dist, stats = ops.knn(torch.from_numpy(src).to(device), torch.from_numpy(tgt).to(device), k,
                               return_dist=True, return_stats=True)
    ids, dist = ids.cpu().numpy(), dist.cpu().numpy()
    ref, ref_d = rp.knn_edges(src, tgt, k, return_dist=True)
    ref_rows = h.knn_rows(ref, len(tgt), k)
    res = h.compare_knn(src, tgt, ids, ref_rows)
    assert res["bad_rows"] == [], res
    # distances: ascending, and equal to the reference's (2*asin(sqrt(rdist))) to float64 rounding
    assert np.all(np.diff(dist, axis=1) >= 0)
    # (with <= 11 sources scikit-learn switches to its brute-force path, which evaluates the metric on
    # float32 copies: its distances are only float32-accurate there)
    rtol = 1e-12 if len(src) > 11 else 1e-6
    np.testing.assert_allclose(dist, ref_d.reshape(len(tgt), k), rtol=rtol, atol=1e-15)
    return res, stats.cpu().numpy()


@pytest.mark.parametrize("k", [1, 3, 8, 12, 16, 20, 32, 40])
def test_knn_random(device, k):
    src, tgt = grids.random_sphere(3000, seed=11), grids.random_sphere(20000, seed=12)
    res, stats = _knn_case(src, tgt, k, device)
    assert res["tie_rows"] == 0
    print("knn random", k, res["tie_rows"], stats)


@pytest.mark.parametrize("ns,nt,k", [(300, 500, 100), (300, 500, 300), (3000, 2000, 65), (3000, 2000, 200),
                                      (400, 100, 300), (5000, 300, 1000)])
def test_knn_more_neighbours_than_the_first_stage_holds(device, ns, nt, k):
    """k beyond 64 -- the reference takes any positive integer up to the number of sources
    (edges/builder.py:222-233) -- up to every source there is: all targets go through the wave-per-target stage
    (k <= 256 ranked from LDS, more in k passes).  Rows and distances equal scikit-learn's."""
    src, tgt = grids.random_sphere(ns, seed=41), grids.random_sphere(nt, seed=42)
    res, stats = _knn_case(src, tgt, k, device)
    assert res["tie_rows"] == 0 and stats[0] == nt
    # and through the builder, with the edge list written by the search
    e = ops.knn_edges(torch.from_numpy(src).to(device), torch.from_numpy(tgt).to(device), k).cpu().numpy()
    assert np.array_equal(e, rp.knn_edges(src, tgt, k))


def test_knn_regular_grid_with_ties(device):
    # poles (coincident points), exact symmetric ties
    src, tgt = grids.fibonacci_sphere(642), grids.regular(2.0)
    res, stats = _knn_case(src, tgt, 3, device)
    print("knn regular", res["tie_rows"], stats)


def test_knn_regular_to_regular(device):
    # massive exact ties: every target equidistant from several sources
    src, tgt = grids.regular(5.0), grids.regular(3.0)
    res, stats = _knn_case(src, tgt, 4, device)
    print("knn reg-reg", res["tie_rows"], stats)


def test_knn_sparse_targets_dense_sources(device):
    src, tgt = grids.random_sphere(50000, seed=3), grids.random_sphere(500, seed=4)
    _knn_case(src, tgt, 5, device)


def test_knn_clustered_sources(device):
    # all sources in a small cap: most targets need the widening float64 path
    rng = np.random.default_rng(5)
    src = np.stack([0.3 + 0.05 * rng.standard_normal(2000), 1.0 + 0.05 * rng.standard_normal(2000)], 1).astype(np.float32)
    tgt = grids.random_sphere(3000, seed=6)
    res, stats = _knn_case(src, tgt, 3, device)
    assert stats[0] > 0


def test_knn_tiny(device):
    src = grids.random_sphere(5, seed=1)
    tgt = grids.random_sphere(7, seed=2)
    _knn_case(src, tgt, 5, device)
    with pytest.raises(ValueError):
        ops.knn(torch.from_numpy(src).to(device), torch.from_numpy(tgt).to(device), 6)


def test_knn_golden(device, golden):
    g = golden("random_sphere")
    for k in (1, 3, 8):
        ids = ops.knn(torch.from_numpy(g["src"]).to(device), torch.from_numpy(g["tgt"]).to(device), k).cpu().numpy()
        ref = h.knn_rows(g[f"knn{k}_edge_index"], len(g["tgt"]), k)
        assert np.array_equal(ids, ref)  # tie-free input: identical including the in-row order
    g = golden("o96_tri4")
    ids = ops.knn(torch.from_numpy(g["hidden"]).to(device), torch.from_numpy(g["data"]).to(device), 3).cpu().numpy()
    res = h.compare_knn(g["hidden"], g["data"], ids, h.knn_rows(g["dec_knn3_edge_index"], len(g["data"]), 3))
    assert res["bad_rows"] == [], res
    print("o96->tri4 tie rows", res["tie_rows"])


@pytest.mark.parametrize("name", ["rand5000", "fib642", "reg2.0", "o32"])
def test_reference_distance(device, name):
    x = grids.named(name)
    got = float(ops.nn_maxdist(torch.from_numpy(x).to(device)).item())
    want = rp.grid_reference_distance(x)
    assert got == pytest.approx(want, rel=1e-13, abs=0.0), (got, want)


def test_reference_distance_golden(device, golden):
    g = golden("random_sphere")
    assert float(ops.nn_maxdist(torch.from_numpy(g["src"]).to(device)).item()) == pytest.approx(float(g["refdist_src"]), rel=1e-13)
    g = golden("regular_2deg")
    assert float(ops.nn_maxdist(torch.from_numpy(g["data"]).to(device)).item()) == pytest.approx(float(g["refdist_data"]), rel=1e-13)


@pytest.fixture(params=["callers_order", "group_order"])
def radius_path(request):
    """The two orders in which the count pass of the cut-off query takes its targets: the caller's (the default), and
    sorted into groups of neighbours (``ag_radius_group_order``).  Both must give the same edge list."""
    from anemoi_graphs_amd import _lib

    previous = _lib.load().ag_radius_group_order(0 if request.param == "callers_order" else 1)
    yield request.param
    _lib.load().ag_radius_group_order(previous)


def _radius_case(src, tgt, radius, device):
    e = ops.radius_edges(torch.from_numpy(src).to(device), torch.from_numpy(tgt).to(device), radius).cpu().numpy()
    assert np.array_equal(e, h.canonical(e)), "rows must be target-major with ascending source ids"
    ref = rp.radius_edges(src, tgt, radius)
    res = h.compare_edge_sets(src, tgt, e, ref, radius)
    assert res["bad"] == 0, res
    return res


@pytest.mark.parametrize("radius", [0.0, 0.01, 0.05, 0.4, 2.0, 3.1])
def test_radius_random(device, radius, radius_path):
    src, tgt = grids.random_sphere(20000, seed=21), grids.random_sphere(1500, seed=22)
    if radius > 1.0:
        src, tgt = src[:2000], tgt[:300]
    res = _radius_case(src, tgt, radius, device)
    assert res["boundary"] == 0
    print("radius", radius, res)


def test_radius_whole_sphere(device, radius_path):
    # radius >= pi: every pair.  (scikit-learn's threshold sin^2(r/2) is not monotonic beyond pi and
    # drops near-antipodal pairs there; a cut-off radius beyond pi is meaningless, so the library
    # saturates instead.)
    src, tgt = grids.random_sphere(300, seed=1), grids.random_sphere(200, seed=2)
    e = ops.radius_edges(torch.from_numpy(src).to(device), torch.from_numpy(tgt).to(device), 3.2)
    assert e.shape[1] == 300 * 200


def test_radius_regular(device, radius_path):
    data, hidden = grids.regular(2.0), grids.fibonacci_sphere(642)
    radius = rp.cutoff_radius(hidden, 0.6)
    print(_radius_case(data, hidden, radius, device))
    # cutoff_factor 1.0: the farthest nearest-neighbour pair sits exactly on the boundary
    print(_radius_case(hidden, hidden, rp.cutoff_radius(hidden, 1.0), device))


def test_radius_golden(device, golden, radius_path):
    g = golden("regular_2deg")
    src, tgt, radius = g["data"], g["hidden"], float(g["enc_cutoff06_radius"])
    e = ops.radius_edges(torch.from_numpy(src).to(device), torch.from_numpy(tgt).to(device), radius).cpu().numpy()
    res = h.compare_edge_sets(src, tgt, e, g["enc_cutoff06_edge_index"], radius)
    assert res["bad"] == 0 and res["boundary"] == 0, res
    g = golden("o96_tri4")
    src, tgt, radius = g["data"], g["hidden"], float(g["enc_cutoff06_radius"])
    e = ops.radius_edges(torch.from_numpy(src).to(device), torch.from_numpy(tgt).to(device), radius).cpu().numpy()
    assert np.array_equal(e, h.canonical(g["enc_cutoff06_edge_index"]))


@pytest.mark.parametrize("ns,nt,radius", [(200000, 400, 0.08), (200000, 300, 0.5), (60000, 50, 2.5)])
def test_radius_long_rows(device, ns, nt, radius, radius_path):
    """Rows of hundreds to tens of thousands of sources: beyond what the count pass keeps, ordered by
    the segmented sort (mixed with short and empty rows: half the targets sit in a source-free cap)."""
    src = grids.random_sphere(ns, seed=31)
    src = src[src[:, 0] < 0.9]  # no sources near the north pole
    tgt = grids.random_sphere(nt, seed=32)
    tgt[: nt // 2, 0] = 1.5  # targets near the pole: short or empty rows for small radii
    res = _radius_case(src, tgt, radius, device)
    e = ops.radius_edges(torch.from_numpy(src).to(device), torch.from_numpy(tgt).to(device), radius)
    rows = np.bincount(e[1].cpu().numpy(), minlength=nt)
    assert rows.max() > 256
    print("long rows", radius, res, "max row", rows.max(), "empty rows", int((rows == 0).sum()))


def test_radius_rows_of_every_length_in_one_query(device, radius_path):
    """Sources clustered around a point: the rows of one query run from empty through a few hundred entries (kept by
    the count pass), a thousand (sorted in LDS by the fill pass) to tens of thousands (the segmented sort)."""
    src, tgt = _clustered(300_000, 61), _clustered(2_500, 62)
    src = src[src[:, 0] > -1.0]  # (no sources around the south pole: empty rows)
    radius = 0.03
    res = _radius_case(src, tgt, radius, device)
    e = ops.radius_edges(torch.from_numpy(src).to(device), torch.from_numpy(tgt).to(device), radius)
    rows = np.bincount(e[1].cpu().numpy(), minlength=len(tgt))
    assert (rows == 0).any() and ((rows > 0) & (rows <= 256)).any() and ((rows > 256) & (rows <= 1024)).any() and (rows > 1024).any()
    # and without any row for the segmented sort
    few = rows <= 1024
    e2 = ops.radius_edges(torch.from_numpy(src).to(device), torch.from_numpy(tgt[few]).to(device), radius).cpu().numpy()
    want = e.cpu().numpy()[:, few[e[1].cpu().numpy()]]
    assert np.array_equal(e2[0], want[0]) and np.array_equal(np.flatnonzero(few)[e2[1]], want[1])
    print("mixed rows", res, "max row", rows.max(), "rows in LDS", int(((rows > 256) & (rows <= 1024)).sum()))


def test_radius_o1280_to_r4(device, radius_path):
    """A coarse hidden mesh under a dense data grid: every row has thousands of entries."""
    data = grids.octahedral(320)
    hidden, _ = rp.tri_nodes(3)
    radius = rp.cutoff_radius(hidden, 0.6)
    res = _radius_case(data, hidden, radius, device)
    print(res)


def _clustered(n, seed):
    rng = np.random.default_rng(seed)
    half = n // 2
    lat = np.concatenate([np.arcsin(rng.uniform(-1, 1, half)), np.clip(rng.normal(1.2, 0.15, n - half), -1.57, 1.57)])
    lon = np.concatenate([rng.uniform(0, 2 * np.pi, half), rng.normal(0.1, 0.3, n - half) % (2 * np.pi)])
    return np.stack([lat, lon], axis=1).astype(np.float32)


@pytest.mark.parametrize("n_src,n_tgt,seed", [(200_000, 3_000, 0), (5_000, 5_000, 1), (300, 40, 2), (50_000, 1, 3)])
@pytest.mark.parametrize("scale", [1.0, 0.3, 4.0, 1e-6, 1e6])
def test_radius_query_on_a_kept_binning(device, n_src, n_tgt, seed, scale, radius_path):
    """The cut-off query gives the same edge list whether it bins its sources itself or is handed a binning the
    caller keeps, whatever the cell size of that binning (ag_radius_count_binned / ag_radius_fill_binned)."""
    src, tgt = _clustered(n_src, seed), grids.random_sphere(n_tgt, seed=100 + seed)
    radius = 2.5 * np.sqrt(4 * np.pi / n_src)
    s, t = torch.from_numpy(src).to(device), torch.from_numpy(tgt).to(device)
    want = ops.radius_edges(s, t, radius)
    kept = ops.bin_points(s, scale * ops.radius_cell(n_src, radius))
    got = ops.radius_edges(None, t, radius, src_binning=kept)
    assert torch.equal(got, want)
    again = ops.radius_edges(None, t[: max(1, n_tgt // 2)], 0.5 * radius, src_binning=kept)  # the binning is not consumed
    assert torch.equal(again, ops.radius_edges(s, t[: max(1, n_tgt // 2)], 0.5 * radius))


@pytest.mark.parametrize("n_src,n_tgt,k,seed", [(40_000, 900_000, 3, 0), (2_000, 50_000, 1, 1), (2_000, 50_000, 8, 2),
                                                (60_000, 60_000, 3, 3), (12, 5_000, 3, 4), (100_000, 700, 20, 5)])
@pytest.mark.parametrize("scale", [1.0, 0.25, 3.0, 1e-6, 50.0])
def test_knn_with_targets_from_a_kept_binning(device, n_src, n_tgt, k, seed, scale):
    """k-NN with the targets taken from a kept binning (their order, their unit vectors) -- any cell size, dense and
    sparse target sets -- returns the rows of the plain query (ag_knn_binned); both equal scikit-learn's."""
    src, tgt = grids.random_sphere(n_src, seed=seed), _clustered(n_tgt, 50 + seed)
    s, t = torch.from_numpy(src).to(device), torch.from_numpy(tgt).to(device)
    want = ops.knn_edges(s, t, k)
    cell = scale * np.sqrt(4 * np.pi / n_tgt) * 4.0
    kept = ops.bin_points(t, cell)
    got = ops.knn_edges(s, None, k, tgt_binning=kept)
    assert torch.equal(got, want)
    ids, dist = ops.knn(s, None, k, return_dist=True, tgt_binning=kept)
    ids2, dist2 = ops.knn(s, t, k, return_dist=True)
    assert torch.equal(ids, ids2) and torch.equal(dist, dist2)
    if n_tgt <= 60_000:
        rows = h.knn_rows(rp.knn_edges(src, tgt, k), n_tgt, k)
        assert h.compare_knn(src, tgt, ids.cpu().numpy(), rows)["bad_rows"] == []


def test_builders_share_one_binning_of_the_data_nodes(device):
    """CutOffEdges leaves its binning of the source nodes on the node storage; KNNEdges towards those nodes takes
    its targets from it; the graph is the one two independent builders make, and clean() drops the binning."""
    from anemoi_graphs_amd import settings
    from anemoi_graphs_amd.create import GraphCreator
    from anemoi_graphs_amd.edges import CutOffEdges, KNNEdges
    from anemoi_graphs_amd.hetero import HeteroData
    from anemoi_graphs_amd.nodes import TriNodes

    data = grids.octahedral(200)
    for resident in (True, False):
        with settings.resident(resident):
            graph = HeteroData()
            graph["data"].x = torch.from_numpy(data).to(device) if resident else torch.from_numpy(data)
            graph["data"].node_type = "LatLonNodes"
            graph = TriNodes(5, "hidden").update_graph(graph, {})
            graph = CutOffEdges("data", "hidden", 0.6).update_graph(graph)
            kept = graph["data"]["_binning"]
            assert isinstance(kept, ops.Binning) and kept.n == len(data)
            graph = KNNEdges("hidden", "data", 3).update_graph(graph)
            assert graph["data"]["_binning"] is kept
            hidden = graph["hidden"].x
            want_enc = ops.radius_edges(torch.from_numpy(data).to(device), hidden.to(device),
                                        0.6 * float(ops.nn_maxdist(hidden).item()))
            want_dec = ops.knn_edges(hidden.to(device), torch.from_numpy(data).to(device), 3)
            assert torch.equal(graph[("data", "to", "hidden")].edge_index.to(device), want_enc)
            assert torch.equal(graph[("hidden", "to", "data")].edge_index.to(device), want_dec)
            # new coordinates under the same name: the kept binning is not trusted
            graph["data"].x = graph["data"].x.clone()
            graph = KNNEdges("hidden", "data", 3).update_graph(graph)
            graph = GraphCreator({"nodes": {}, "edges": []}).clean(graph)
            assert "_binning" not in graph["data"]


def test_knn_edges_entry_point_writes_both_rows(device):
    """ag_knn_edges = ag_knn + the target of every entry, with or without a binning of the targets."""
    from anemoi_graphs_amd import ops

    src, tgt = grids.fibonacci_sphere(5000), grids.regular(3.0)
    k = 4
    s, t = torch.from_numpy(src).to(device), torch.from_numpy(tgt).to(device)
    ids = ops.knn(s, t, k)
    want = torch.stack([ids.reshape(-1), torch.arange(len(tgt), device=device, dtype=torch.int32).repeat_interleave(k)])
    assert torch.equal(ops.knn_edges(s, t, k), want)
    kept = ops.bin_points(t, 0.05)
    assert torch.equal(ops.knn_edges(s, None, k, tgt_binning=kept), want)


@pytest.mark.parametrize("k", [1, 3, 8, 20, 40])
@pytest.mark.parametrize("sources", ["random", "regular"])
def test_knn_targets_far_sparser_than_the_source_cells(device, sources, k):
    """Fewer than one target per source cell: the targets are not binned, every lane of stage 1 takes one -- random
    targets, the poles, targets on a source and targets midway between the nodes of a regular grid (four-way ties),
    in an order without any locality."""
    rng = np.random.default_rng(40 + k)
    src = grids.random_sphere(150_000, seed=41) if sources == "random" else grids.regular(0.5)
    picks = src[rng.integers(0, len(src), 60)].copy()
    half = np.deg2rad(0.25)
    between = grids.regular(0.5)[rng.integers(0, 259_200, 60)] + np.array([half, half], dtype=np.float32)
    tgt = np.concatenate([grids.random_sphere(400, seed=42), picks, between.astype(np.float32),
                          np.array([[np.pi / 2, 0.3], [-np.pi / 2, 5.0], [0.0, 0.0], [0.0, 2 * np.pi - 1e-6]], dtype=np.float32)])
    tgt = tgt[rng.permutation(len(tgt))]
    res, stats = _knn_case(src, tgt, k, device)
    edges = ops.knn_edges(torch.from_numpy(src).to(device), torch.from_numpy(tgt).to(device), k)
    assert torch.equal(edges[0], ops.knn(torch.from_numpy(src).to(device), torch.from_numpy(tgt).to(device), k).reshape(-1))
    assert torch.equal(edges[1], torch.arange(len(tgt), device=device, dtype=torch.int32).repeat_interleave(k))
    print("knn per target", sources, k, res["tie_rows"], stats)


@pytest.mark.parametrize("bound_km", [2.0, 40.0, 400.0, 3000.0])
def test_knn_within_a_bound_of_a_dense_regional_set(device, bound_km):
    """The area mask of a stretched mesh: the vertices of a global mesh looked up in a dense regional set, within a
    bound.  Most vertices have nothing in reach (one look at the widest window settles them); those near the
    region find few sources in it (one scan of that window) or many (the window grows by doubling)."""
    from sklearn.neighbors import NearestNeighbors

    rng = np.random.default_rng(int(bound_km))
    lat = np.deg2rad(rng.uniform(35.0, 60.0, 400_000))
    lon = np.deg2rad(rng.uniform(-15.0, 30.0, 400_000))
    ref = np.stack([lat, lon], axis=1).astype(np.float32)
    query = rp.tri_nodes(5)[0]
    bound = bound_km / 6371.0
    for k in (1, 2):
        ids, dist = ops.knn(torch.from_numpy(ref).to(device), torch.from_numpy(query).to(device), k, return_dist=True,
                            max_dist=bound)
        ids, dist = ids.cpu().numpy(), dist.cpu().numpy()
        want_d, want_i = NearestNeighbors(metric="haversine").fit(ref).kneighbors(query, n_neighbors=k)
        inside = want_d <= bound
        edge = np.abs(want_d - bound) <= 1e-9 * bound
        assert inside.any() and (~inside).any()
        keep = inside & ~edge
        assert np.all(ids[keep] >= 0) and np.all(ids[~inside & ~edge] == -1) and np.all(np.isinf(dist[~inside & ~edge]))
        np.testing.assert_allclose(dist[keep], want_d[keep], rtol=1e-12, atol=1e-15)
        differ = keep & (ids != want_i)  # (exact distance ties may list the tied sources in another order)
        np.testing.assert_allclose(dist[differ], want_d[differ], rtol=1e-12)


def _regional(n, seed, lat0=35.0, lat1=72.0, lon0=-25.0, lon1=45.0):
    rng = np.random.default_rng(seed)
    z = rng.uniform(np.sin(np.deg2rad(lat0)), np.sin(np.deg2rad(lat1)), n)
    lon = np.deg2rad(rng.uniform(lon0, lon1, n)) % (2 * np.pi)
    return np.stack([np.arcsin(z), lon], axis=1).astype(np.float32)


def test_point_density_of_global_regional_and_stretched_sets(device):
    """ag_point_density: 1 for sets spread over the sphere (random, lattice, regular grid, small), 1/f for a set
    confined to the fraction f of the sphere, and for a stretched set the mean over its nodes."""
    area = (np.sin(np.deg2rad(72.0)) - np.sin(np.deg2rad(35.0))) * np.deg2rad(70.0) / (4 * np.pi)
    for x, want, tol in [(grids.random_sphere(300_000, seed=1), 1.0, 0.05), (grids.fibonacci_sphere(200_000), 1.0, 0.05),
                         (grids.random_sphere(3_000, seed=2), 1.0, 0.3), (grids.regular(0.5), np.pi / 2, 0.1),
                         (_regional(400_000, 3), 1.0 / area, 0.1), (_regional(5_000, 4), 1.0 / area, 0.4)]:
        got = ops.point_density(torch.from_numpy(x).to(device))
        assert abs(got / want - 1.0) < tol, (len(x), got, want)
    # nine tenths of the nodes in the region, one tenth over the sphere
    both = np.concatenate([_regional(360_000, 5), grids.random_sphere(40_000, seed=6)])
    want = (0.9 + 0.1 * area) ** 2 / area + 0.1 ** 2 * (1 - area) / 1.0  # sum over the two parts of share^2 / area
    got = ops.point_density(torch.from_numpy(both).to(device))
    assert abs(got / want - 1.0) < 0.1, (got, want)
    assert ops.point_density(torch.from_numpy(grids.random_sphere(100, seed=7)).to(device)) == 1.0


@pytest.mark.parametrize("k", [1, 3, 12])
def test_knn_between_regional_sets_for_any_density_given(device, k):
    """The densities size the cells and choose the kernels; the neighbours are scikit-learn's whatever is given:
    measured (the default), 1 (a regional set binned as if it were global), far too large, and mixed."""
    src, tgt = _regional(60_000, 11), _regional(9_000, 12)
    s, t = torch.from_numpy(src).to(device), torch.from_numpy(tgt).to(device)
    rows = h.knn_rows(rp.knn_edges(src, tgt, k), len(tgt), k)
    measured = ops.knn(s, t, k)
    assert h.compare_knn(src, tgt, measured.cpu().numpy(), rows)["bad_rows"] == []
    for sd, td in [(1.0, 1.0), (25.0, 1.0), (1.0, 25.0), (5_000.0, 5_000.0), (0.0, float("nan")), (float("inf"), 1e300)]:
        assert torch.equal(ops.knn(s, t, k, src_density=sd, tgt_density=td), measured)
        assert torch.equal(ops.knn_edges(s, t, k, src_density=sd, tgt_density=td)[0], measured.reshape(-1))
    kept = ops.bin_points(t, 0.004)
    for sd in (1.0, 25.0, 5_000.0):
        assert torch.equal(ops.knn(s, None, k, tgt_binning=kept, src_density=sd), measured)
    # the other way round (dense targets), a bounded search, and the reference distance
    back = ops.knn(t, s, k)
    assert h.compare_knn(tgt, src, back.cpu().numpy(), h.knn_rows(rp.knn_edges(tgt, src, k), len(src), k))["bad_rows"] == []
    assert torch.equal(ops.knn(t, s, k, src_density=1.0, tgt_density=1.0), back)
    world = torch.from_numpy(grids.fibonacci_sphere(20_000)).to(device)
    for sd in (None, 1.0, 400.0):
        ids, dist = ops.knn(s, world, k, return_dist=True, max_dist=0.02, src_density=sd, tgt_density=1.0)
        if sd is None:
            want_ids, want_dist = ids, dist
        assert torch.equal(ids, want_ids) and torch.equal(dist, want_dist)
    assert float(ops.nn_maxdist(s).item()) == float(ops.nn_maxdist(s, density=1.0).item()) == float(ops.nn_maxdist(s, density=300.0).item())


@pytest.mark.parametrize("seed", range(10))
def test_knn_over_regions_of_any_size(device, seed):
    """Sources confined to regions from a speck to half the sphere (densities from 2 to far beyond what the cell
    arrays are allowed to follow), targets inside, around or all over the sphere, with and without a bound: the rows
    are scikit-learn's, with the measured densities and with arbitrary ones."""
    from sklearn.neighbors import NearestNeighbors

    rng = np.random.default_rng(900 + seed)
    ns = int(rng.choice([2_000, 30_000, 200_000]))
    nt = int(rng.choice([500, 20_000]))
    k = int(rng.choice([1, 2, 3, 8, 12]))
    half = 10.0 ** rng.uniform(-2.0, 1.7)  # half-width of the region in degrees: 0.01 .. 50
    lat_c, lon_c = rng.uniform(-80.0, 80.0), rng.uniform(0.0, 360.0)
    lat0, lat1 = max(lat_c - half, -89.9), min(lat_c + half, 89.9)
    src = _regional(ns, 1000 + seed, lat0, lat1, lon_c - half, lon_c + half)
    where = rng.integers(0, 3)
    if where == 0:
        tgt = _regional(nt, 2000 + seed, lat0, lat1, lon_c - half, lon_c + half)
    elif where == 1:  # a wider box around the region: many targets outside it
        tgt = _regional(nt, 2000 + seed, max(lat_c - 3 * half, -89.9), min(lat_c + 3 * half, 89.9), lon_c - 3 * half, lon_c + 3 * half)
    else:
        tgt = grids.random_sphere(nt, seed=2000 + seed)
    s, t = torch.from_numpy(src).to(device), torch.from_numpy(tgt).to(device)
    ids, dist = ops.knn(s, t, k, return_dist=True)
    want_d, want_i = NearestNeighbors(metric="haversine").fit(src).kneighbors(tgt, n_neighbors=k)
    np.testing.assert_allclose(dist.cpu().numpy(), want_d, rtol=1e-12, atol=1e-15)
    differ = ids.cpu().numpy() != want_i  # (exact distance ties may list the tied sources in another order)
    np.testing.assert_array_equal(dist.cpu().numpy()[differ], want_d[differ])
    for sd, td in [(1.0, 1.0), (float(10.0 ** rng.uniform(0, 6)), float(10.0 ** rng.uniform(0, 6)))]:
        assert torch.equal(ops.knn(s, t, k, src_density=sd, tgt_density=td), ids)
    bound = float(np.deg2rad(half) * 10.0 ** rng.uniform(-1.5, 0.5))
    ids_b, dist_b = ops.knn(s, t, k, return_dist=True, max_dist=bound)
    inside = want_d <= bound
    edge = np.abs(want_d - bound) <= 1e-9 * bound
    keep = inside & ~edge
    got_d = dist_b.cpu().numpy()
    np.testing.assert_allclose(got_d[keep], want_d[keep], rtol=1e-12, atol=1e-15)
    assert np.all(np.isinf(got_d[~inside & ~edge])) and np.all(ids_b.cpu().numpy()[~inside & ~edge] == -1)
    print("regions", seed, ns, nt, k, half, int(where), ops.point_density(s), int(inside.sum()))


@pytest.mark.parametrize("case", ["o320_r5", "era5like_r4", "clustered", "kept_fine", "kept_coarse", "polar_targets"])
def test_radius_encoder_shaped_queries(device, case, radius_path):
    """The cut-off query on configurations shaped like encoders, rows of 40 to 200 entries (the row sort's network
    for the first 64 entries, the placement of the entries beyond, the ranking of long rows): a reduced Gaussian grid
    under an icosphere, a regular grid with crowded polar cells, clustered node sets, kept binnings with cells finer
    and coarser than the radius asks for (windows of four bands), targets on and around the poles.  Canonical order,
    and equal to scikit-learn's rows on a sample of the targets."""
    if case == "o320_r5":
        src, tgt = grids.octahedral(320), rp.tri_nodes(5)[0]
        radius, scale = rp.cutoff_radius(tgt, 0.6), None
    elif case == "era5like_r4":
        src, tgt = grids.regular(0.5), rp.tri_nodes(4)[0]
        radius, scale = rp.cutoff_radius(tgt, 0.6), None
    elif case == "clustered":
        src, tgt = _clustered(400_000, 71), _clustered(30_000, 72)
        radius, scale = 0.012, None
    elif case in ("kept_fine", "kept_coarse"):
        src, tgt = grids.random_sphere(300_000, seed=73), grids.random_sphere(20_000, seed=74)
        radius, scale = 0.02, (0.55 if case == "kept_fine" else 2.5)
    else:
        src = grids.random_sphere(200_000, seed=75)
        rng = np.random.default_rng(76)
        lat = np.concatenate([np.full(50, np.pi / 2), np.full(50, -np.pi / 2), np.pi / 2 - rng.uniform(0, 0.05, 6000),
                              -np.pi / 2 + rng.uniform(0, 0.05, 6000)])
        tgt = np.stack([lat, rng.uniform(0, 2 * np.pi, lat.size)], axis=1).astype(np.float32)
        radius, scale = 0.03, None
    s, t = torch.from_numpy(src).to(device), torch.from_numpy(tgt).to(device)
    kept = None if scale is None else ops.bin_points(s, scale * ops.radius_cell(len(src), radius))
    e = ops.radius_edges(s if kept is None else None, t, radius, src_binning=kept).cpu().numpy()
    assert np.array_equal(e, h.canonical(e))
    rows = np.bincount(e[1], minlength=len(tgt))
    rng = np.random.default_rng(7)
    pick = np.sort(rng.choice(len(tgt), size=min(len(tgt), 400), replace=False))
    ref = rp.radius_edges(src, tgt[pick], radius)
    keep = np.isin(e[1], pick)
    got = np.stack([e[0][keep], np.searchsorted(pick, e[1][keep])]).astype(e.dtype)
    res = h.compare_edge_sets(src, tgt[pick], got, ref, radius)
    assert res["bad"] == 0, res
    print(case, "edges", e.shape[1], "rows: mean", rows.mean(), "over 64:", int((rows > 64).sum()), "over 128:",
          int((rows > 128).sum()), res)


@pytest.mark.parametrize("n", [1, 2, 3, 31, 63, 64, 65, 66, 90, 127, 128, 129, 200, 256])
def test_radius_rows_of_exact_lengths_are_sorted(device, n, radius_path):
    """Rows of exactly n entries, for the lengths around which the row sort changes method (one key per lane: 64;
    placement of the entries beyond: 65 ... 128; ranking: 129 ...), with source ids in descending, ascending and
    shuffled order of their position: every target sees the same n sources."""
    rng = np.random.default_rng(n)
    centre = np.array([0.4, 1.3])
    near = centre + 1.0e-3 * rng.uniform(-1, 1, (n, 2))
    far = grids.random_sphere(5000, seed=n)
    far = far[np.abs(far[:, 0] - 0.4) > 0.2]
    for order in ("shuffled", "ascending", "descending"):
        src = np.concatenate([far, near]).astype(np.float32)
        perm = rng.permutation(len(src)) if order == "shuffled" else np.arange(len(src))
        if order == "descending":
            perm = perm[::-1]
        src = np.ascontiguousarray(src[perm])
        tgt = (centre + 1.0e-3 * rng.uniform(-1, 1, (40, 2))).astype(np.float32)
        e = ops.radius_edges(torch.from_numpy(src).to(device), torch.from_numpy(tgt).to(device), 0.01).cpu().numpy()
        assert e.shape[1] == 40 * n
        assert np.array_equal(e, h.canonical(e)), (n, order)
        want = np.sort(np.flatnonzero(np.isin(perm, np.arange(len(far), len(far) + n))))
        assert np.array_equal(e[0].reshape(40, n), np.broadcast_to(want, (40, n)))


def test_searches_do_not_depend_on_the_order_of_the_nodes(device):
    """A grid stored row by row (neighbours next to each other: the binning counts runs of points and places them with
    a cursor per cell) and the same nodes shuffled (a scattered order: every point keeps the slot its counter update
    returned) give the same neighbours and the same cut-off edges, id for id."""
    data = grids.octahedral(160)  # 108,160 nodes, rows of constant latitude
    # (nudged: on the exact grid a node's left and right neighbours tie, and ties go to the lower id -- which the
    # shuffle changes)
    data = (data + np.random.default_rng(4).uniform(-2e-4, 2e-4, data.shape)).astype(np.float32)
    data[:, 0] = np.clip(data[:, 0], -np.pi / 2, np.pi / 2)
    data[:, 1] = np.mod(data[:, 1], np.float32(2 * np.pi))
    hidden = grids.fibonacci_sphere(2562)
    perm = np.random.default_rng(5).permutation(len(data))
    shuffled = data[perm]
    d0, d1 = torch.from_numpy(data).to(device), torch.from_numpy(shuffled).to(device)
    hd = torch.from_numpy(hidden).to(device)
    radius = rp.cutoff_radius(hidden, 0.6)
    # the grid as sources of a cut-off query
    e0 = ops.radius_edges(d0, hd, radius).cpu().numpy()
    e1 = ops.radius_edges(d1, hd, radius).cpu().numpy()
    e1 = np.stack([perm[e1[0]], e1[1]])
    assert np.array_equal(h.canonical(e0), h.canonical(e1)) and e0.shape[1] > 0
    # the grid as sources of a k-NN query, with itself as the targets
    n0 = ops.knn(d0, d0[:5000], 4).cpu().numpy()
    n1 = ops.knn(d1, d0[:5000], 4).cpu().numpy()
    assert np.array_equal(np.sort(n0, axis=1), np.sort(perm[n1], axis=1))
    # and against the oracle once
    from oracle import compare as cmp

    want = cmp.knn_rows(rp.knn_edges(data, data[:5000], 4), 5000, 4)
    assert np.array_equal(np.sort(n0, axis=1), np.sort(want, axis=1))
