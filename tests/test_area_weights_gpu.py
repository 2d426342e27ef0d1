"""GPU: SphericalAreaWeights / UniformWeights against the reference's own classes (golden fixture written by
oracle/make_golden.py from scipy's SphericalVoronoi as the reference calls it), against the oracle on larger
sets, and through size-independent properties at full size."""
import math

import numpy as np
import pytest
import torch

from anemoi_graphs_amd import grids, ops
from anemoi_graphs_amd.hetero import HeteroData
from anemoi_graphs_amd.nodes import TriNodes
from anemoi_graphs_amd.nodes.attributes import AreaWeights, SphericalAreaWeights, UniformWeights
from oracle import reference_path as rp

pytestmark = pytest.mark.gpu

NORMS = [None, "l1", "l2", "unit-max", "unit-range", "unit-std"]
SETS = ["o32", "tri3", "random", "aniso"]


def _graph(x, device):
    g = HeteroData()
    g["nodes"].x = torch.as_tensor(x).to(device)
    return g


@pytest.mark.parametrize("name", SETS)
def test_raw_areas_match_reference(device, golden, name):
    """The float64 cell areas themselves: scipy's values carry the rounding of its cumulative sum over all
    triangles (about 1e-12 absolute here), so 1e-9 relative is the precision of the reference itself."""
    gold = golden("area_weights")
    x = gold[f"{name}_x"]
    area, stats = ops.voronoi_areas(torch.from_numpy(x).to(device), return_stats=True)
    area = area.cpu().numpy()
    np.testing.assert_allclose(area, gold[f"{name}_area_raw"], rtol=1e-9, atol=0)
    # (the generators are float32 unit vectors, so the solid angles of the reference's formula tile
    # the sphere to about 1e-6 only; the sum is compared with the reference's own)
    assert abs(area.sum() - gold[f"{name}_area_raw"].sum()) < 1e-9
    assert abs(area.sum() - 4 * math.pi) < 1e-4
    if name == "aniso":  # elongated cells near the poles need more than the nearest 16 nodes
        assert stats["first_pass_open"] > 0 and stats["row_passes"] >= 1
    if name == "tri3":
        assert stats["first_pass_open"] == 0


@pytest.mark.parametrize("name", SETS)
@pytest.mark.parametrize("norm", NORMS)
def test_spherical_area_weights_match_reference(device, golden, name, norm):
    gold = golden("area_weights")
    tag = "none" if norm is None else norm.replace("-", "_")
    g = _graph(gold[f"{name}_x"], device)
    got = SphericalAreaWeights(norm=norm).compute(g, "nodes")
    want = gold[f"{name}_area_{tag}"]
    assert got.dtype == torch.float32 and tuple(got.shape) == want.shape
    # float32 values: one rounding apart at most (the float64 values agree to 1e-9); unit-range subtracts
    # the minimum, which leaves the smallest cells with few significant digits in the reference itself
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1.2e-7, atol=1e-10 if norm == "unit-range" else 1e-12)


def test_uniform_weights_match_reference(device, golden):
    gold = golden("area_weights")
    g = _graph(gold["o32_x"], device)
    np.testing.assert_array_equal(UniformWeights(norm="l1").compute(g, "nodes").cpu().numpy(), gold["o32_uniform_l1"])
    np.testing.assert_array_equal(UniformWeights().compute(g, "nodes").cpu().numpy(), gold["o32_uniform_none"])


@pytest.mark.parametrize("make", [lambda: grids.named("o96"), lambda: rp.tri_nodes(5)[0].astype(np.float32),
                                  lambda: grids.random_sphere(50000, seed=11)], ids=["o96", "tri5", "random50k"])
def test_areas_match_oracle(device, make):
    x = make()
    want = rp.spherical_area_weights(x, raw=True)
    got = ops.voronoi_areas(torch.from_numpy(x).to(device)).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=1e-9, atol=0)


def test_builder_registers_area_weights(device):
    """Through the node builder, as the orchestrator passes the attribute configuration (nodes/builders/base.py:66-79)."""
    g = HeteroData()
    cfg = {"area_weight": {"_target_": "anemoi.graphs.nodes.attributes.AreaWeights", "norm": "unit-max"},
           "ones": {"_target_": "anemoi.graphs.nodes.attributes.UniformWeights"}}
    g = TriNodes(3, "hidden").update_graph(g, cfg)
    w = g["hidden"]["area_weight"]
    assert w.dtype == torch.float32 and tuple(w.shape) == (642, 1)
    assert float(w.max()) == 1.0 and float(w.min()) > 0.7  # icosphere cells are of similar size
    want = rp.spherical_area_weights(g["hidden"].x.cpu().numpy(), norm="unit-max")
    np.testing.assert_allclose(w.cpu().numpy(), want, rtol=1.2e-7)
    assert torch.equal(g["hidden"]["ones"].cpu(), torch.ones(642, 1))


def test_full_size_properties(device):
    """O1280 (6.6 M nodes), beyond what scipy finishes in test time: the cells tile the sphere, every area is
    positive, and rows of equal latitude (equal cells by symmetry) have equal areas."""
    x = grids.named("o1280")
    area, stats = ops.voronoi_areas(torch.from_numpy(x).to(device), return_stats=True)
    total = float(area.sum())
    assert abs(total - 4 * math.pi) < 1e-5, total  # float32 unit vectors: see test_raw_areas_match_reference
    assert float(area.min()) > 0.0
    a = area.cpu().numpy()
    row = a[20 + 24:20 + 24 + 28]  # third row from the north pole: 28 nodes
    assert np.allclose(row, row[0], rtol=1e-6)
    mean = 4 * math.pi / len(a)
    assert 0.1 * mean < a.min() and a.max() < 3.0 * mean  # the rows next to the poles hold the smallest cells


def test_errors_follow_scipy(device):
    x = grids.named("o16").copy()
    g = _graph(x, device)
    with pytest.raises(ValueError, match="Radius inconsistent with generators"):
        SphericalAreaWeights(radius=2.0).compute(g, "nodes")
    x[5] = x[4]
    with pytest.raises(ValueError, match="Duplicate generators present"):
        SphericalAreaWeights().compute(_graph(x, device), "nodes")
    # scipy's threshold: generators within 1e-6 of each other count as duplicates -- the 1440 nodes of a
    # regular grid's pole row are such a set (cos(pi/2) is 4e-8 in float32, not 0)
    with pytest.raises(ValueError, match="Duplicate generators present"):
        SphericalAreaWeights().compute(_graph(grids.named("reg2.0"), device), "nodes")
    with pytest.raises(ValueError, match="Duplicate generators present"):
        rp.spherical_area_weights(grids.named("reg2.0"))
    with pytest.raises(ValueError, match="not valid"):
        SphericalAreaWeights(norm="l3").compute(g, "nodes")
    with pytest.raises(NotImplementedError):
        AreaWeights(flat=True)


@pytest.mark.parametrize("dtype,torch_dtype", [("float64", torch.float64), ("f8", torch.float64), ("float", torch.float64),
                                                ("float16", torch.float16), ("float32", torch.float32)])
def test_node_attribute_dtype_keeps_float64_precision(device, dtype, torch_dtype):
    """nodes/attributes.py:45-52 of the reference: normalise(values).astype(dtype) on the float64 values --
    a float64 attribute is not rounded through float32, and dtype takes whatever numpy's astype takes."""
    from anemoi_graphs_amd.hetero import HeteroData
    from anemoi_graphs_amd.nodes.attributes import SphericalAreaWeights, UniformWeights
    from oracle import reference_path as rp

    x = rp.tri_nodes(3)[0]
    graph = HeteroData()
    graph["n"].x = torch.from_numpy(x)
    got = SphericalAreaWeights(norm="unit-max", dtype=dtype).compute(graph, "n")
    want = rp.spherical_area_weights(x, "unit-max", dtype=dtype)
    assert got.dtype == torch_dtype and tuple(got.shape) == want.shape
    rtol = {torch.float64: 1e-9, torch.float32: 1.2e-7, torch.float16: 1e-3}[torch_dtype]
    np.testing.assert_allclose(got.numpy().astype(np.float64), want.astype(np.float64), rtol=rtol)
    if torch_dtype == torch.float64:  # more than float32 precision survives
        assert np.abs(got.numpy() - got.numpy().astype(np.float32)).max() > 0
    u = UniformWeights(norm="l1", dtype=dtype).compute(graph, "n")
    assert u.dtype == torch_dtype and np.array_equal(u.numpy(), rp.uniform_weights(len(x), "l1", dtype=dtype))
    with pytest.raises(TypeError):
        UniformWeights(dtype="not-a-dtype").compute(graph, "n")
