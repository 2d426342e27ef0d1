"""GPU parity of EdgeLength / EdgeDirection / normalisation.  Tolerance: 1e-6 RELATIVE per element, as the
north-star states -- no floor scaled by the largest value; an absolute floor exists only for exact zeros of the
expected value and for direction components below 1e-3 (there: 1e-9 absolute, a millionth of the floor itself).
Every comparison prints the largest relative error it saw and the share of bit-identical float32 values."""
import numpy as np
import pytest
import torch

from anemoi_graphs_amd import grids, ops
from oracle import reference_path as rp

pytestmark = pytest.mark.gpu

RTOL = 1e-6
NORMS = [None, "l1", "l2", "unit-max", "unit-range", "unit-std"]


def _tag(norm):
    return "none" if norm is None else norm.replace("-", "_")


def _attrs(src, tgt, e, device, norm=None, rotated=True, invert=False):
    """What EdgeLength / EdgeDirection do with these arguments (anemoi_graphs_amd/edges/attributes.py)."""
    from anemoi_graphs_amd.edges.attributes import EdgeDirection

    dir_f64 = EdgeDirection(norm, rotated).wants_float64
    length, direction = ops.edge_attrs(torch.from_numpy(src).to(device), torch.from_numpy(tgt).to(device),
                                       torch.from_numpy(np.ascontiguousarray(e)).to(device), rotated=rotated,
                                       dir_f64=dir_f64)
    length = ops.normalise_(length.reshape(-1, 1), norm, invert)
    direction = ops.normalise_f64(direction, norm) if dir_f64 else ops.normalise_(direction, norm)
    return length.cpu().numpy(), direction.cpu().numpy()


def _close(got, want, what="", small=0.0):
    """|got - want| <= 1e-6 |want| element by element; where |want| <= ``small`` (0: exact zeros only) the bound
    is 1e-6 * max(small, 1e-6) absolute instead.  Reports what it measured."""
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, (got.shape, want.shape)
    err = np.abs(got.astype(np.float64) - want.astype(np.float64))
    mag = np.abs(want.astype(np.float64))
    big = mag > small
    rel = float((err[big] / mag[big]).max()) if big.any() else 0.0
    same = float((got.astype(np.float32).view(np.int32) == want.astype(np.float32).view(np.int32)).mean())
    print(f"{what}: max relative error {rel:.3g} on {int(big.sum())} values, {100.0 * same:.4f} % bit-identical")
    assert rel <= RTOL, (what, rel)
    if (~big).any():
        worst = float(err[~big].max())
        assert worst <= RTOL * max(small, 1e-6), (what, worst)


DIR_SMALL = 1e-3  # direction components below this are compared absolutely (1e-9)


@pytest.mark.parametrize("case,src_key,tgt_key,prefix,ekey", [
    ("random_sphere", "src", "tgt", "knn3_", "knn3_edge_index"),
    ("random_sphere", "tgt", "src", "cutoff06_", "cutoff06_edge_index"),
    ("conftest_grid_f32", "x", "x", "knn3_", "knn3_edge_index"),
])
def test_attrs_golden(device, golden, case, src_key, tgt_key, prefix, ekey):
    g = golden(case)
    src, tgt, e = g[src_key].astype(np.float32), g[tgt_key].astype(np.float32), g[ekey]
    raw_dir = g[f"{prefix}dir_none"]
    loops = (e[0] == e[1]) if src_key == tgt_key else np.zeros(e.shape[1], dtype=bool)
    for norm in NORMS:
        length, direction = _attrs(src, tgt, e, device, norm)
        _close(length, g[f"{prefix}len_{_tag(norm)}"], f"{case} {prefix}len {norm}")
        if norm == "l1" and abs(raw_dir.sum()) < 1e-3 * np.abs(raw_dir).sum():
            # "l1" divides by the plain sum (normalise.py:40); for direction components that cancel
            # (a symmetric grid) the divisor is rounding noise in the reference itself: not comparable
            continue
        want_dir = g[f"{prefix}dir_{_tag(norm)}"]
        scale = float(np.abs(want_dir).max())  # (the norms rescale the unit components)
        _close(direction[~loops], want_dir[~loops], f"{case} {prefix}dir {norm}", small=DIR_SMALL * scale)
        # an edge from a node to itself has no direction: the reference normalises the rounding noise of the
        # rotated source (1e-17) after nudging it by 1e-10 (generate/transforms.py:133-140), which leaves
        # (1, -1) / sqrt(2) with a relative noise of 1e-7 that no other implementation reproduces digit by digit
        if loops.any():
            np.testing.assert_allclose(direction[loops], want_dir[loops], rtol=0, atol=4e-6 * scale)
    length, _ = _attrs(src, tgt, e, device, "unit-max", invert=True)
    # stored value 1 - v, v = length / max: the tolerance is 1e-6 relative ON v (the subtraction cancels for the
    # longest edges) plus one float32 rounding of the stored value, which lies in [0, 1]
    want_inv = g[f"{prefix}len_inv_unit_max"].astype(np.float64)
    err = np.abs(length.astype(np.float64) - want_inv)
    print(f"{case} {prefix}len inverted: max error {err.max():.3g}")
    assert np.all(err <= RTOL * (1.0 - want_inv) + 6.0e-8)
    _, direction = _attrs(src, tgt, e, device, None, rotated=False)
    _close(direction, g[f"{prefix}dir_plain_none"], f"{case} {prefix}dir plain")


def test_attrs_raw_are_float32_exact_inputs(device, golden):
    """The float32 unit vectors are bit-identical to numpy's, so short edges (where one float32 ulp
    of a coordinate is 1e-4 of the edge) still agree to 1e-6."""
    g = golden("regular_2deg")
    length, direction = _attrs(g["data"], g["hidden"], g["enc_cutoff06_edge_index"], device)
    _close(length, g["enc_len_none"], "regular_2deg len")
    _close(direction, g["enc_dir_none"], "regular_2deg dir", small=DIR_SMALL)


def test_attrs_live_oracle_fine_grid(device):
    # 0.25-degree neighbours: edge length ~4e-3 rad; poles and the date line included
    data = grids.regular(0.25)
    hidden = grids.fibonacci_sphere(40962)
    rng = np.random.default_rng(0)
    t = rng.integers(0, len(hidden), 200000)
    s = rp.knn_edges(data, hidden[np.unique(t)], 4)[0][:200000]
    t = np.unique(t)[np.arange(len(s)) // 4][: len(s)]
    e = np.stack([s, t]).astype(np.int32)
    length, direction = _attrs(data, hidden, e, device)
    _close(length[:, 0], rp.edge_lengths(data, hidden, e), "fine grid len")
    _close(direction, rp.edge_directions(data, hidden, e), "fine grid dir", small=DIR_SMALL)


def test_attrs_degenerate_edges(device):
    # self edges (length 0), targets exactly at the poles
    x = np.array([[0.3, 1.0], [np.pi / 2, 0.0], [-np.pi / 2, 2.0], [0.0, 0.0], [1.2, -3.0]], dtype=np.float32)
    e = np.array([[0, 0, 3, 4, 1, 2, 3], [0, 1, 2, 1, 1, 2, 4]], dtype=np.int32)
    length, direction = _attrs(x, x, e, device)
    want_len = rp.edge_lengths(x, x, e)
    np.testing.assert_allclose(length[:, 0], want_len, rtol=RTOL, atol=1e-9)
    want_dir = rp.edge_directions(x, x, e)
    np.testing.assert_allclose(direction, want_dir, rtol=RTOL, atol=1e-6)


def test_normalise_errors_and_std_zero(device):
    v = torch.full((10, 1), 2.5, device=device)
    assert torch.equal(ops.normalise_(v.clone(), "unit-std"), v)  # std == 0: untouched (normalise.py:49-51)
    with pytest.raises(ValueError):
        ops.normalise_(v.clone(), "l3")


@pytest.mark.parametrize("offset", [0, 1, 2, 3])
@pytest.mark.parametrize("n", [1, 2, 3, 5, 64, 1001, 40003])
@pytest.mark.parametrize("norm", ["l1", "l2", "unit-max", "unit-range", "unit-std"])
def test_normalise_any_alignment_and_length(device, offset, n, norm):
    """The kernels read 16 bytes per lane: views that do not start on a 16-byte boundary and lengths that are
    not multiples of 4 must give the same values (normalise.py:36-55)."""
    rng = np.random.default_rng(n * 7 + offset)
    full = rng.uniform(0.5, 3.0, size=n + offset).astype(np.float32)
    t = torch.from_numpy(full).to(device)[offset:]
    assert t.is_contiguous()
    got = ops.normalise_(t, norm).cpu().numpy()
    want = rp.post_process(full[offset:].reshape(-1, 1), norm)[:, 0]
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize("n_edges", [1, 3, 127, 128, 129, 5000, 300001])
def test_length_range_from_the_attribute_kernel(device, n_edges):
    """The range of the lengths reported by the attribute kernel gives the unit-max / unit-range
    normalisations (normalise.py:45-48) the same bits as the reduction pass over the values."""
    rng = np.random.default_rng(n_edges)
    x = np.stack([np.arcsin(rng.uniform(-1, 1, 4000)), rng.uniform(0, 2 * np.pi, 4000)], axis=1).astype(np.float32)
    e = rng.integers(0, len(x), size=(2, n_edges)).astype(np.int32)
    if n_edges > 3:
        e[0, 2] = e[1, 2]  # a self edge: the minimum is exactly 0
    xd, ed = torch.from_numpy(x).to(device), torch.from_numpy(e).to(device)
    length, _, value_range = ops.edge_attrs(xd, xd, ed, len_range=True)
    plain, _ = ops.edge_attrs(xd, xd, ed)
    assert torch.equal(length, plain)
    bits = value_range.cpu().numpy().view(np.uint32)
    assert np.uint32(bits[0]) == length.cpu().numpy().max().view(np.uint32)
    assert np.uint32(~bits[1]) == length.cpu().numpy().min().view(np.uint32)
    for norm in (None, "unit-max", "unit-range"):
        for invert in (False, True):
            want = ops.normalise_(plain.clone().reshape(-1, 1), norm, invert)
            got = ops.normalise_(length.clone().reshape(-1, 1), norm, invert, value_range=value_range)
            assert torch.equal(got.view(torch.int32), want.view(torch.int32)), (norm, invert)  # (0/0 = NaN for one edge)
    # a normalisation that does not follow from the range takes the ordinary route
    want = ops.normalise_(plain.clone().reshape(-1, 1), "l2")
    got = ops.normalise_(length.clone().reshape(-1, 1), "l2", value_range=value_range)
    assert torch.equal(got, want)


def test_out_of_range_edge_ids_raise_index_error(device):
    """numpy raises IndexError for an edge that refers to a node that does not exist
    (edges/attributes.py:144-148); the kernel never follows such an id (every launch ends normally) and the
    host raises the same error: right after the call when graph tensors live on the host, at the end of the
    resident context when they stay on the device."""
    from anemoi_graphs_amd import settings

    x = torch.from_numpy(grids.fibonacci_sphere(1000)).to(device)
    good = torch.tensor([[0, 5, 999], [1, 2, 3]], dtype=torch.int32, device=device)
    want_len, want_dir = ops.edge_attrs(x, x, good)
    for bad_value, row in ((1000, 0), (-1, 0), (2**31 - 1, 1), (-2**31, 1)):
        e = good.clone()
        e[row, 1] = bad_value
        with pytest.raises(IndexError):
            ops.edge_attrs(x, x, e)
        # the error word is cleared by the raise, the other edges were evaluated as usual
        length, direction = ops.edge_attrs(x, x, good)
        assert torch.equal(length, want_len) and torch.equal(direction, want_dir)
    e = good.clone()
    e[0, 2] = 5000
    with pytest.raises(IndexError):
        with settings.resident(True):
            length, _ = ops.edge_attrs(x, x, e)  # deferred: no synchronisation inside a resident build
            assert torch.equal(length[:2], want_len[:2])
    with settings.resident(True):
        ops.edge_attrs(x, x, good)
    with pytest.raises(IndexError):
        ops.edge_attrs(x[:0], x, good)
    ops.check_ids()
