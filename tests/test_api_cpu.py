"""CPU: host-side behaviour of the plugin classes, mirroring the reference's own tests
(/root/reference/tests/edges/test_knn.py, test_cutoff.py, test_multiscale_edges.py,
tests/test_normaliser.py, tests/test_create.py) where they do not need the device."""
import numpy as np
import pytest
import torch

from anemoi_graphs_amd import ops
from anemoi_graphs_amd.create import DotDict, GraphCreator
from anemoi_graphs_amd.edges import CutOffEdges, KNNEdges, MultiScaleEdges
from anemoi_graphs_amd.edges.attributes import EdgeDirection, EdgeLength
from anemoi_graphs_amd.hetero import HeteroData
from anemoi_graphs_amd.instantiate import instantiate, resolve
from anemoi_graphs_amd.nodes import LatLonNodes, TriNodes
from anemoi_graphs_amd.nodes.attributes import (AreaWeights, BooleanAndMask, BooleanNot, BooleanOrMask,
                                                SphericalAreaWeights, UniformWeights)


def test_knn_init():
    assert isinstance(KNNEdges("test_nodes1", "test_nodes2", 3), KNNEdges)
    assert KNNEdges("a", "b", 3).name == ("a", "to", "b")


@pytest.mark.parametrize("num_nearest_neighbours", [-1, 2.6, "hello", None])
def test_knn_fail_init(num_nearest_neighbours):
    with pytest.raises(AssertionError):
        KNNEdges("test_nodes1", "test_nodes2", num_nearest_neighbours)


def test_cutoff_init():
    assert isinstance(CutOffEdges("test_nodes1", "test_nodes2", 0.5), CutOffEdges)


@pytest.mark.parametrize("cutoff_factor", [-0.5, "hello", None])
def test_cutoff_fail_init(cutoff_factor):
    with pytest.raises(AssertionError):
        CutOffEdges("test_nodes1", "test_nodes2", cutoff_factor)


def test_multiscale_init():
    assert isinstance(MultiScaleEdges("test_nodes", "test_nodes", 1), MultiScaleEdges)


@pytest.mark.parametrize("x_hops", [-0.5, "hello", None, -4])
def test_multiscale_fail_init(x_hops):
    with pytest.raises(AssertionError):
        MultiScaleEdges("test_nodes", "test_nodes", x_hops)


def test_multiscale_fail_init_diff_nodes():
    with pytest.raises(AssertionError):
        MultiScaleEdges("test_nodes", "test_nodes2", 0)


def test_multiscale_rejects_foreign_node_type():
    graph = HeteroData()
    graph["fail_nodes"].x = [1, 2, 3]
    graph["fail_nodes"].node_type = "FailNodes"
    with pytest.raises(AssertionError):
        MultiScaleEdges("fail_nodes", "fail_nodes", 1).update_graph(graph)


def test_attribute_requires_known_nodes():
    graph = HeteroData()
    graph["test_nodes"].x = torch.zeros((4, 2))
    with pytest.raises(AssertionError):
        EdgeLength().compute(graph, ("test_nodes", "to", "unknown_nodes"))
    with pytest.raises(AssertionError):
        EdgeDirection().compute(graph, ("unknown_nodes", "to", "test_nodes"))


def test_invalid_norm_is_a_value_error():
    with pytest.raises(ValueError):
        ops.normalise_(torch.zeros(3), "invalid")


def test_targets_of_the_reference_package_resolve_here():
    assert resolve("anemoi.graphs.edges.KNNEdges") is KNNEdges
    assert resolve("anemoi.graphs.nodes.TriNodes") is TriNodes
    assert resolve("anemoi.graphs.edges.attributes.EdgeLength") is EdgeLength
    b = instantiate({"_target_": "anemoi.graphs.edges.CutOffEdges", "cutoff_factor": 0.6}, source_name="a", target_name="b")
    assert isinstance(b, CutOffEdges) and b.cutoff_factor == 0.6
    with pytest.raises(ImportError):
        resolve("anemoi.graphs.nodes.HexNodes")


def test_resolution_list_and_hidden_attributes():
    assert TriNodes(3, "h").resolutions == [0, 1, 2, 3]
    assert TriNodes([0, 2], "h").resolutions == [0, 2]
    assert {"resolutions", "nx_graph", "node_ordering", "area_mask_builder"} <= TriNodes(1, "h").hidden_attributes


def test_latlon_nodes_shape_checks():
    with pytest.raises(AssertionError):
        LatLonNodes([0.0, 1.0], [0.0], "n")
    x = LatLonNodes([0.0, 90.0], [0.0, 180.0], "n").reshape_coords(
        torch.tensor([0.0, 90.0]).numpy(), torch.tensor([0.0, 180.0]).numpy())
    assert x.dtype == torch.float32 and x.shape == (2, 2)
    assert x[1, 0].item() == pytest.approx(torch.pi / 2)


def test_graph_creator_legacy_edge_builder_key():
    cfg = {
        "nodes": {},
        "edges": [{
            "source_name": "a", "target_name": "b",
            "edge_builder": {"_target_": "anemoi.graphs.edges.KNNEdges", "num_nearest_neighbours": 3},
            "source_mask_attr_name": "m",
        }],
    }
    with pytest.warns(DeprecationWarning):
        creator = GraphCreator(cfg)
    eb = creator.config.edges[0].edge_builders
    assert len(eb) == 1 and eb[0].source_mask_attr_name == "m" and eb[0].target_mask_attr_name is None


def test_clean_removes_private_attributes():
    graph = HeteroData()
    graph["n"].x = torch.zeros((2, 2))
    graph["n"]["_hidden"] = 1
    graph[("n", "to", "n")].edge_index = torch.zeros((2, 1), dtype=torch.int32)
    graph[("n", "to", "n")]["_tmp"] = 2
    graph = GraphCreator({"nodes": {}, "edges": []}).clean(graph)
    assert "_hidden" not in graph["n"] and "_tmp" not in graph[("n", "to", "n")] and "x" in graph["n"]


def test_dotdict():
    d = DotDict({"a": {"b": [{"c": 1}]}})
    assert d.a.b[0].c == 1


def test_no_silent_fallback_without_a_gpu():
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.knn(torch.zeros((4, 2)), torch.zeros((4, 2)), 1)
    with pytest.raises(RuntimeError):
        TriNodes(1, "h").update_graph(HeteroData(), {})


# ---- /root/reference/tests/generate/test_masks.py:17-39 -------------------------------------------------


def test_area_mask_builder_init():
    from anemoi_graphs_amd.generate import KNNAreaMaskBuilder

    for kwargs in ({}, {"margin_radius_km": 120}, {"mask_attr_name": "mask"},
                   {"margin_radius_km": 120, "mask_attr_name": "mask"}):
        assert isinstance(KNNAreaMaskBuilder("nodes", **kwargs), KNNAreaMaskBuilder)


@pytest.mark.parametrize("margin", [-1, "120", None])
def test_area_mask_builder_fail_init_wrong_margin(margin):
    from anemoi_graphs_amd.generate import KNNAreaMaskBuilder

    with pytest.raises(AssertionError):
        KNNAreaMaskBuilder("nodes", margin_radius_km=margin)


def test_limited_area_and_stretched_builders_construct_like_the_reference():
    """from_refined_icosahedron.py:81-92,159-172: constructor arguments, resolutions, mask builder."""
    from anemoi_graphs_amd.nodes import LimitedAreaTriNodes, StretchedTriNodes

    lam = LimitedAreaTriNodes(3, "ref", "hidden", margin_radius_km=50.0)
    assert lam.resolutions == [0, 1, 2, 3] and lam.area_mask_builder.margin_radius_km == 50.0
    assert lam.area_mask_builder.reference_node_name == "ref"
    st = StretchedTriNodes(2, 5, "hidden", "ref", "cutout")
    assert st.global_resolution == 2 and st.resolutions == list(range(6))
    assert st.area_mask_builder.mask_attr_name == "cutout"
    assert {"resolutions", "nx_graph", "node_ordering", "area_mask_builder"} <= set(st.hidden_attributes)


def test_boolean_node_masks():
    """BooleanNot / BooleanAndMask / BooleanOrMask of nodes/attributes.py:271-316 (host logic)."""
    g = HeteroData()
    g = LatLonNodes(np.array([0.0, 10.0, 20.0, 30.0]), np.array([0.0, 1.0, 2.0, 3.0]), "n").update_graph(g, {})
    g["n"]["a"] = torch.tensor([True, True, False, False])
    g["n"]["b"] = torch.tensor([True, False, True, False])
    assert BooleanNot("a").compute(g, "n").reshape(-1).tolist() == [False, False, True, True]
    assert BooleanAndMask(["a", "b"]).compute(g, "n").reshape(-1).tolist() == [True, False, False, False]
    assert BooleanOrMask(["a", BooleanNot("b")]).compute(g, "n").reshape(-1).tolist() == [True, True, False, True]
    out = BooleanOrMask(["a", "b"]).compute(g, "n")
    assert out.dtype == torch.bool and tuple(out.shape) == (4, 1)
    g["n"]["c"] = torch.tensor([1.0, 0.0, 1.0, 0.0])
    with pytest.raises(AssertionError):
        BooleanNot("c").compute(g, "n")


def test_node_attribute_constructors_and_aliases():
    """Constructor defaults of nodes/attributes.py:38-40,185-196 and the recipe targets."""
    a = SphericalAreaWeights()
    assert (a.norm, a.dtype, a.radius, a.fill_value) == (None, "float32", 1.0, 0.0)
    assert isinstance(AreaWeights(norm="l1"), SphericalAreaWeights)
    with pytest.raises(NotImplementedError):
        AreaWeights(flat=True)
    assert resolve("anemoi.graphs.nodes.attributes.SphericalAreaWeights") is SphericalAreaWeights
    assert resolve("anemoi.graphs.nodes.attributes.UniformWeights") is UniformWeights
    assert isinstance(instantiate({"_target_": "anemoi.graphs.nodes.attributes.AreaWeights", "norm": "unit-max"}),
                      SphericalAreaWeights)


def test_graph_descriptor_summaries(tmp_path, capsys):
    """describe.py:40-106 of the reference: node and edge tables with the isolated-node counts."""
    import math

    import torch

    from anemoi_graphs_amd.describe import GraphDescriptor
    from anemoi_graphs_amd.hetero import HeteroData

    g = HeteroData()
    g["a"].x = torch.tensor([[0.1, 0.0], [0.2, 1.0], [0.3, 2.0], [-0.4, 3.0], [0.0, 6.0]])
    g["a"].node_type = "LatLonNodes"
    g["a"].weights = torch.ones(5, 1)
    g["b"].x = torch.tensor([[0.5, 0.5], [0.6, 0.7], [-0.6, 0.9]])
    g["b"].node_type = "LatLonNodes"
    e = g[("a", "to", "b")]
    e.edge_index = torch.tensor([[0, 1, 1], [0, 0, 2]], dtype=torch.int32)
    e.edge_type = "KNNEdges"
    e.edge_length = torch.tensor([[0.5], [1.0], [0.25]])
    e.edge_dirs = torch.zeros(3, 2)
    d = GraphDescriptor(g)
    nodes = d.get_node_summary()
    assert nodes[0][:4] == ["a", 5, "weights", 1] and nodes[1][:4] == ["b", 3, "", 0]
    assert nodes[0][4] == pytest.approx(-0.4 / 2 / math.pi * 360) and nodes[0][7] == pytest.approx(6.0 / 2 / math.pi * 360)
    edges = d.get_edge_summary()
    assert edges == [["a", "b", 3, 3, 1, 3, "edge_length(1D), edge_dirs(2D)"]]
    assert d.total_size == (10 + 5 + 6 + 3 + 6) * 4 + 6 * 4
    table = d.get_attribute_table()
    assert [r[:3] for r in table] == [["Node", "a", "weights"], ["Edge", "a-->b", "edge_length"], ["Edge", "a-->b", "edge_dirs"]]
    assert table[1][4:7] == [0.25, pytest.approx(1.75 / 3), 1.0]
    from anemoi_graphs_amd import hetero

    hetero.save_graph(g, tmp_path / "g.pt")
    if not hetero.HAVE_TORCH_GEOMETRIC:  # plain containers: nothing in the file is executed on load
        assert torch.load(tmp_path / "g.pt", weights_only=True)["format"] == hetero.PORTABLE_FORMAT
    back = hetero.load_graph(tmp_path / "g.pt")
    assert back.node_types == ["a", "b"] and back.edge_types == [("a", "to", "b")]
    assert torch.equal(back[("a", "to", "b")].edge_index, e.edge_index) and back["a"].node_type == "LatLonNodes"
    GraphDescriptor(tmp_path / "g.pt").describe()
    out = capsys.readouterr().out
    assert "Isolated Source" in out and "Graph ready." in out


def test_stores_of_the_stand_in_container_can_be_deleted():
    """`del graph[key]`, as torch_geometric's HeteroData allows it (a recipe step may replace an edge set)."""
    from anemoi_graphs_amd import hetero

    if hetero.HAVE_TORCH_GEOMETRIC:
        pytest.skip("torch_geometric's own container is in use")
    graph = HeteroData()
    graph["a"].x = torch.zeros(3, 2)
    graph[("a", "to", "a")].edge_index = torch.zeros((2, 0), dtype=torch.int32)
    assert graph.edge_types == [("a", "to", "a")]
    del graph[["a", "to", "a"]]
    assert graph.edge_types == [] and graph.node_types == ["a"] and ("a", "to", "a") not in graph
    with pytest.raises(KeyError):
        del graph[("a", "to", "a")]


def test_a_host_tensor_rewritten_behind_torchs_back_is_noticed():
    """Device mirrors, kept binnings and cached distances are tied to the host tensor they came from: the object, its
    version counter and a fingerprint of sampled elements -- a grid rewritten through the numpy array that shares
    the tensor's memory leaves the first two alone."""
    import numpy as np
    import torch

    from anemoi_graphs_amd import ops

    arr = np.random.default_rng(0).random((5000, 2), dtype=np.float32)
    x = torch.from_numpy(arr)
    stamp = ops.tensor_key(x)
    assert stamp.matches(x) and not stamp.matches(x.clone())
    version = x._version
    arr[:] = arr[::-1].copy()  # torch does not see this write
    assert x._version == version
    assert not stamp.matches(x)
    assert ops.tensor_key(x).matches(x)
    assert ops.host_fingerprint(x[:, 0]) is None  # (not contiguous: never mirrored)
    assert ops.host_fingerprint(torch.empty(0)) == b""
