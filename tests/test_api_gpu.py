"""GPU: the plugin classes and the orchestrator end to end, mirroring the reference's own tests
(/root/reference/tests/test_create.py, tests/nodes/test_tri_nodes.py, tests/edges/test_knn.py,
test_cutoff.py, test_multiscale_edges.py, test_edge_attributes.py) and checking values against the
oracle / golden fixtures."""
import numpy as np
import pytest
import torch

from anemoi_graphs_amd import grids, settings
from anemoi_graphs_amd.create import GraphCreator
from anemoi_graphs_amd.edges import CutOffEdges, KNNEdges, MultiScaleEdges
from anemoi_graphs_amd.edges.attributes import EdgeDirection, EdgeLength
from anemoi_graphs_amd.hetero import HeteroData
from anemoi_graphs_amd.nodes import TriNodes
from oracle import reference_path as rp
from tests import helpers as h

pytestmark = pytest.mark.gpu


def _recipe():
    lat, lon = grids.octahedral_grid_deg(16)
    return {
        "nodes": {
            "data": {"node_builder": {"_target_": "anemoi.graphs.nodes.LatLonNodes", "latitudes": lat, "longitudes": lon}},
            "hidden": {"node_builder": {"_target_": "anemoi.graphs.nodes.TriNodes", "resolution": 2}},
        },
        "edges": [
            {"source_name": "data", "target_name": "hidden",
             "edge_builders": [{"_target_": "anemoi.graphs.edges.CutOffEdges", "cutoff_factor": 0.6}],
             "attributes": {
                 "edge_length": {"_target_": "anemoi.graphs.edges.attributes.EdgeLength", "norm": "unit-max"},
                 "edge_dirs": {"_target_": "anemoi.graphs.edges.attributes.EdgeDirection", "norm": "unit-std"}}},
            {"source_name": "hidden", "target_name": "hidden",
             "edge_builders": [{"_target_": "anemoi.graphs.edges.MultiScaleEdges", "x_hops": 1}],
             "attributes": {"edge_length": {"_target_": "anemoi.graphs.edges.attributes.EdgeLength", "norm": "unit-std"}}},
            {"source_name": "hidden", "target_name": "data",
             "edge_builders": [{"_target_": "anemoi.graphs.edges.KNNEdges", "num_nearest_neighbours": 3}],
             "attributes": {
                 "edge_length": {"_target_": "anemoi.graphs.edges.attributes.EdgeLength", "norm": "l2", "invert": True},
                 "edge_dirs": {"_target_": "anemoi.graphs.edges.attributes.EdgeDirection", "luse_rotated_features": False}}},
        ],
    }


@pytest.mark.parametrize("resident", [False, True])
def test_graph_creator_end_to_end(device, tmp_path, resident):
    path = tmp_path / "graph.pt"
    with settings.resident(resident):
        graph = GraphCreator(_recipe()).create(save_path=path)
    # tests/test_create.py:35-50 of the reference: tensors of int32 / float32, no private attributes
    for name in list(graph.node_types) + list(graph.edge_types):
        for key, value in graph[name].items():
            assert not key.startswith("_")
            if isinstance(value, torch.Tensor):
                assert value.dtype in (torch.int32, torch.float32)
                assert value.is_cuda == resident or path.exists()
    assert path.exists()
    from anemoi_graphs_amd.hetero import load_graph

    loaded = load_graph(path)
    assert set(loaded.node_types) == {"data", "hidden"} and len(loaded.edge_types) == 3
    assert all(not v.is_cuda for n in loaded.edge_types for v in loaded[n].values() if isinstance(v, torch.Tensor))

    data, hidden = grids.octahedral(16), graph["hidden"].x.cpu().numpy()
    want_hidden, order = rp.tri_nodes(2)
    assert hidden.shape == (162, 2)  # tests/nodes/test_tri_nodes.py:27-32
    assert np.abs(hidden.view(np.int32).astype(np.int64) - want_hidden.view(np.int32)).max() <= 1
    assert graph["hidden"].node_type == "TriNodes" and graph["data"].node_type == "LatLonNodes"
    assert np.array_equal(graph["data"].x.cpu().numpy(), data)

    enc = graph[("data", "to", "hidden")]
    e = enc.edge_index.cpu().numpy()
    want, _ = rp.cutoff_edges(data, hidden, 0.6)
    assert np.array_equal(e, h.canonical(want)) and enc.edge_type == "CutOffEdges"
    np.testing.assert_allclose(enc["edge_length"].cpu().numpy(),
                               rp.post_process(rp.edge_lengths(data, hidden, e), "unit-max"), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(enc["edge_dirs"].cpu().numpy(),
                               rp.post_process(rp.edge_directions(data, hidden, e), "unit-std"), rtol=1e-6, atol=2e-6)
    proc = graph[("hidden", "to", "hidden")].edge_index.cpu().numpy()
    assert np.array_equal(proc, h.canonical(rp.multiscale_edges([0, 1, 2], 1, order)))
    dec = graph[("hidden", "to", "data")]
    d = dec.edge_index.cpu().numpy()
    res = h.compare_knn(hidden, data, h.knn_rows(d, len(data), 3), h.knn_rows(rp.knn_edges(hidden, data, 3), len(data), 3))
    assert res["bad_rows"] == []
    np.testing.assert_allclose(dec["edge_length"].cpu().numpy(),
                               rp.post_process(rp.edge_lengths(hidden, data, d), "l2", True), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(dec["edge_dirs"].cpu().numpy(), rp.edge_directions(hidden, data, d, False), rtol=1e-6)


def test_tri_nodes_hidden_attributes(device):
    """tests/nodes/test_tri_nodes.py:35-43 of the reference."""
    graph = TriNodes(2, "test_nodes").update_graph(HeteroData(), {})
    nodes = graph["test_nodes"]
    assert nodes.x.shape == (162, 2) and nodes.x.dtype == torch.float32
    for key in ("_resolutions", "_nx_graph", "_node_ordering", "_area_mask_builder"):
        assert key in nodes
    assert nodes["_resolutions"] == [0, 1, 2]
    assert len(nodes["_node_ordering"]) == nodes.x.shape[0] == nodes["_nx_graph"].number_of_nodes()
    # without an attribute config nothing hidden is stored (nodes/builders/base.py:120-123) ...
    bare = TriNodes(1, "n").update_graph(HeteroData())
    assert "_nx_graph" not in bare["n"]
    # ... and MultiScaleEdges then fails, as in the reference
    with pytest.raises(KeyError):
        MultiScaleEdges("n", "n", 1).update_graph(bare)


def test_masked_builders_golden(device, golden):
    g = golden("masks")
    graph = HeteroData()
    graph["a"].x = torch.from_numpy(g["src"])
    graph["b"].x = torch.from_numpy(g["tgt"])
    graph["a"]["m"] = torch.from_numpy(g["src_mask"])
    graph["b"]["m"] = torch.from_numpy(g["tgt_mask"])
    graph = KNNEdges("a", "b", 4, source_mask_attr_name="m", target_mask_attr_name="m").update_graph(graph)
    e = graph[("a", "to", "b")].edge_index
    assert e.dtype == torch.int32 and not e.is_cuda
    assert np.array_equal(e.numpy(), g["knn4_edge_index"])
    c = CutOffEdges("b", "a", 0.8, source_mask_attr_name="m", target_mask_attr_name="m")
    graph = c.update_graph(graph)
    assert c.radius == pytest.approx(float(g["cutoff08_radius"]), rel=1e-13)
    assert np.array_equal(h.canonical(graph[("b", "to", "a")].edge_index.numpy()), h.canonical(g["cutoff08_edge_index"]))


def test_two_builders_on_one_edge_set_are_merged(device):
    """edges/builder.py:105-110 of the reference: union via concat_edges, edge_type joined by a comma."""
    src, tgt = grids.random_sphere(400, seed=8), grids.random_sphere(900, seed=9)
    graph = HeteroData()
    graph["s"].x, graph["t"].x = torch.from_numpy(src), torch.from_numpy(tgt)
    graph = KNNEdges("s", "t", 2).update_graph(graph)
    graph = CutOffEdges("s", "t", 0.7).update_graph(graph)
    store = graph[("s", "to", "t")]
    assert store.edge_type == "KNNEdges,CutOffEdges"
    want = rp.concat_edges(rp.knn_edges(src, tgt, 2), rp.cutoff_edges(src, tgt, 0.7)[0])
    assert np.array_equal(store.edge_index.numpy(), want)
    graph = KNNEdges("s", "t", 2).update_graph(graph)  # same type again: not repeated
    assert graph[("s", "to", "t")].edge_type == "KNNEdges,CutOffEdges"


@pytest.mark.parametrize("norm", ["l1", "l2", "unit-max", "unit-std"])
@pytest.mark.parametrize("rotated", [True, False])
def test_edge_attribute_types(device, norm, rotated):
    """tests/edges/test_edge_attributes.py:17-31 of the reference."""
    x = np.array([[lat, lon] for lat in (-0.15, 0, 0.15) for lon in (0, 0.25, 0.5, 0.75)]) * 2 * np.pi
    graph = HeteroData()
    graph["test_nodes"].x = torch.tensor(x, dtype=torch.float32)
    graph[("test_nodes", "to", "test_nodes")].edge_index = torch.tensor([[0, 1, 2, 3], [1, 2, 3, 0]], dtype=torch.int32)
    d = EdgeDirection(norm=norm, luse_rotated_features=rotated).compute(graph, ("test_nodes", "to", "test_nodes"))
    l = EdgeLength(norm=norm).compute(graph, ("test_nodes", "to", "test_nodes"))
    assert isinstance(d, torch.Tensor) and d.shape == (4, 2) and d.dtype == torch.float32
    assert isinstance(l, torch.Tensor) and l.shape == (4, 1) and l.dtype == torch.float32


def test_graph_descriptor_on_device_graph(device):
    """Isolated-node counts of GraphDescriptor.get_edge_summary (describe.py:100-101 of the reference) when the
    edge lists live on the GPU: ag_mark_ids, equal to torch.unique on the host."""
    from anemoi_graphs_amd import grids, settings
    from anemoi_graphs_amd.describe import GraphDescriptor
    from anemoi_graphs_amd.distributed import ShardedGraphBuilder

    data = torch.from_numpy(grids.octahedral(32)).to(device)
    with settings.resident(True):
        graph = ShardedGraphBuilder(data, 3, cutoff_factor=0.6, num_nearest_neighbours=3, x_hops=1).build()
    rows = {(r[0], r[1]): r for r in GraphDescriptor(graph).get_edge_summary()}
    for (src, _, dst), store in graph.edge_items():
        e = store.edge_index.cpu()
        want_src = graph[src].x.shape[0] - len(torch.unique(e[0]))
        want_dst = graph[dst].x.shape[0] - len(torch.unique(e[1]))
        assert rows[(src, dst)][2:5] == [e.shape[1], want_src, want_dst]
    assert rows[("hidden", "data")][4] == 0  # every data node receives its k nearest hidden nodes
