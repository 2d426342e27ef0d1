"""GPU: degenerate and boundary inputs of the search entry points -- empty target sets, a single source,
k equal to the number of sources, coincident points, poles, longitudes outside [0, 2 pi), empty edge
lists -- against scikit-learn called as the reference calls it."""
import numpy as np
import pytest
import torch

from anemoi_graphs_amd import grids, ops
from oracle import reference_path as rp
from tests import helpers as h

pytestmark = pytest.mark.gpu


def _t(x, device):
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(device)


def test_empty_targets(device):
    src = _t(grids.random_sphere(100, seed=1), device)
    empty = torch.empty((0, 2), dtype=torch.float32, device=device)
    assert ops.knn(src, empty, 3).shape == (0, 3)
    assert ops.knn_edges(src, empty, 3).shape == (2, 0)
    assert ops.radius_edges(src, empty, 0.1).shape == (2, 0)
    e = ops.radius_edges(src, src[:5], 0.0)  # radius 0: only the coincident pairs
    assert np.array_equal(e.cpu().numpy(), np.stack([np.arange(5), np.arange(5)]))


def test_empty_edge_list_attributes(device):
    x = _t(grids.random_sphere(10, seed=2), device)
    e = torch.empty((2, 0), dtype=torch.int32, device=device)
    length, direction = ops.edge_attrs(x, x, e)
    assert length.shape == (0,) and direction.shape == (0, 2)
    assert ops.normalise_(length, "unit-max").shape == (0,)
    assert ops.unique_edges(e).shape == (2, 0)


def test_single_source_and_k_equal_ns(device):
    tgt = grids.random_sphere(500, seed=3)
    one = grids.random_sphere(1, seed=4)
    ids = ops.knn(_t(one, device), _t(tgt, device), 1).cpu().numpy()
    assert np.all(ids == 0)
    src = grids.random_sphere(12, seed=5)  # above scikit-learn's brute-force threshold of 11 sources
    got = ops.knn(_t(src, device), _t(tgt, device), 12).cpu().numpy()
    want = h.knn_rows(rp.knn_edges(src, tgt, 12), len(tgt), 12)
    assert h.compare_knn(src, tgt, got, want)["bad_rows"] == []
    with pytest.raises(ValueError):
        ops.knn(_t(src, device), _t(tgt, device), 13)
    with pytest.raises(ValueError):
        ops.nn_maxdist(_t(one, device))


def test_coincident_points(device):
    base = grids.random_sphere(50, seed=6)
    src = np.repeat(base, 4, axis=0)  # every location four times
    tgt = base.copy()
    got = ops.knn(_t(src, device), _t(tgt, device), 4).cpu().numpy()
    # the four nearest sources of target i are its four copies (distance 0), in ascending id order
    assert np.array_equal(got, np.arange(200).reshape(50, 4))
    # reference distance (utils.py:61-63: dists[dists > 0].max() over the 2 nearest, self included): when every
    # node has a twin there is no positive distance; the reference then fails inside numpy, and so does the
    # host function here, while the C entry point reports 0
    assert float(ops.nn_maxdist(_t(src, device)).item()) == 0.0
    with pytest.raises(ValueError):
        rp.grid_reference_distance(src)
    from anemoi_graphs_amd.utils import get_grid_reference_distance

    with pytest.raises(ValueError):
        get_grid_reference_distance(_t(src, device))
    # one twin pair among distinct nodes: the positive distances of the others count
    mixed = np.concatenate([base, base[:1]])
    assert float(ops.nn_maxdist(_t(mixed, device)).item()) == pytest.approx(rp.grid_reference_distance(mixed), rel=1e-12)


@pytest.mark.parametrize("shift", [0.0, -np.pi, 2 * np.pi, -7.0, 11.0])
def test_longitude_conventions(device, shift):
    """Longitudes in [-pi, pi), [0, 2 pi) or further out describe the same points."""
    src = grids.random_sphere(3000, seed=7)
    tgt = grids.random_sphere(800, seed=8)
    src_s, tgt_s = src.copy(), tgt.copy()
    src_s[:, 1] += np.float32(shift)
    tgt_s[::2, 1] -= np.float32(shift)
    got = ops.knn(_t(src_s, device), _t(tgt_s, device), 3).cpu().numpy()
    want = h.knn_rows(rp.knn_edges(src_s, tgt_s, 3), len(tgt), 3)
    assert h.compare_knn(src_s, tgt_s, got, want)["bad_rows"] == []
    e = ops.radius_edges(_t(src_s, device), _t(tgt_s, device), 0.12).cpu().numpy()
    ref = rp.radius_edges(src_s, tgt_s, 0.12)
    assert h.compare_edge_sets(src_s, tgt_s, e, ref, 0.12)["bad"] == 0


def test_poles_and_antimeridian(device):
    lat = np.array([np.pi / 2, -np.pi / 2, np.pi / 2 - 1e-4, -np.pi / 2 + 1e-4, 0.0, 0.0, 1e-3, -1e-3], dtype=np.float32)
    lon = np.array([0.0, 1.0, 2.0, 3.0, 1e-5, 2 * np.pi - 1e-5, np.pi, -np.pi], dtype=np.float32)
    special = np.stack([lat, lon], axis=1)
    src = np.concatenate([special, grids.random_sphere(4000, seed=9)])
    tgt = np.concatenate([special + np.float32(1e-5) * np.array([[-1, 1]], dtype=np.float32), grids.random_sphere(300, seed=10)])
    tgt[:, 0] = np.clip(tgt[:, 0], -np.pi / 2, np.pi / 2)
    got = ops.knn(_t(src, device), _t(tgt, device), 5).cpu().numpy()
    want = h.knn_rows(rp.knn_edges(src, tgt, 5), len(tgt), 5)
    assert h.compare_knn(src, tgt, got, want)["bad_rows"] == []
    for radius in (1e-3, 0.05, 0.6):
        e = ops.radius_edges(_t(src, device), _t(tgt, device), radius).cpu().numpy()
        ref = rp.radius_edges(src, tgt, radius)
        assert h.compare_edge_sets(src, tgt, e, ref, radius)["bad"] == 0


def test_mask_all_false_and_all_true(device):
    m0 = ops.mask_ids(torch.zeros(100, dtype=torch.bool), device)
    m1 = ops.mask_ids(torch.ones(100, dtype=torch.bool), device)
    assert m0.numel() == 0 and np.array_equal(m1.cpu().numpy(), np.arange(100))


@pytest.mark.parametrize("case", ["latlon_grid_with_poles", "one_band_multiple_of_64", "one_band_plus_one", "two_caps"])
def test_dense_targets_by_ranges(device, case):
    """Dense target sets are searched in ranges of 64 binned targets per latitude band: bands whose
    population is an exact multiple of 64, one more than that, thousands of coincident targets at a
    pole, and bands without any target between populated ones."""
    rng = np.random.default_rng(17)
    src = grids.fibonacci_sphere(500)
    if case == "latlon_grid_with_poles":
        tgt = grids.regular(1.0)  # 181 x 360, the two pole rows are 360 coincident points each
    elif case == "one_band_multiple_of_64":
        tgt = np.stack([np.full(64 * 300, 0.4), rng.uniform(0, 2 * np.pi, 64 * 300)], axis=1)
    elif case == "one_band_plus_one":
        tgt = np.stack([0.4 + rng.uniform(-1e-3, 1e-3, 64 * 300 + 1), rng.uniform(0, 2 * np.pi, 64 * 300 + 1)], axis=1)
    else:
        lat = np.concatenate([np.pi / 2 - np.abs(rng.normal(0, 0.05, 15000)), -np.pi / 2 + np.abs(rng.normal(0, 0.05, 15000))])
        tgt = np.stack([lat, rng.uniform(-np.pi, np.pi, 30000)], axis=1)
    tgt = tgt.astype(np.float32)
    for k in (1, 3, 7):
        got = ops.knn(_t(src, device), _t(tgt, device), k).cpu().numpy()
        want = h.knn_rows(rp.knn_edges(src, tgt, k), len(tgt), k)
        res = h.compare_knn(src, tgt, got, want)
        assert res["bad_rows"] == [], (case, k, res)
