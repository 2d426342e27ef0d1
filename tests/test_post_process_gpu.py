"""GPU: RemoveUnconnectedNodes, mirroring /root/reference/tests/processors/test_post_process.py:18-79
(same fixture, /root/reference/tests/conftest.py:67-72, same expected values) plus a random case
against a numpy restatement of processors/post_process.py:41-61,133-149."""
import numpy as np
import pytest
import torch

from anemoi_graphs_amd.hetero import HeteroData
from anemoi_graphs_amd.processors import RemoveUnconnectedNodes

pytestmark = pytest.mark.gpu


@pytest.fixture
def graph_with_isolated_nodes() -> HeteroData:
    graph = HeteroData()
    graph["test_nodes"].x = torch.tensor([[1], [2], [3], [4], [5], [6]])
    graph["test_nodes"]["mask_attr"] = torch.tensor([[1], [1], [1], [0], [0], [0]], dtype=torch.bool)
    graph["test_nodes", "to", "test_nodes"].edge_index = torch.tensor([[2, 3, 4], [1, 2, 3]])
    return graph


def test_remove_unconnected_nodes(device, graph_with_isolated_nodes):
    processor = RemoveUnconnectedNodes(nodes_name="test_nodes", ignore=None, save_mask_indices_to_attr=None)
    graph = processor.update_graph(graph_with_isolated_nodes)
    assert graph["test_nodes"].num_nodes == 4
    assert torch.equal(graph["test_nodes"].x, torch.tensor([[2], [3], [4], [5]]))
    assert "original_indices" not in graph["test_nodes"]


def test_remove_unconnected_nodes_with_indices_attr(device, graph_with_isolated_nodes):
    processor = RemoveUnconnectedNodes(nodes_name="test_nodes", ignore=None, save_mask_indices_to_attr="original_indices")
    graph = processor.update_graph(graph_with_isolated_nodes)
    assert graph["test_nodes"].num_nodes == 4
    assert torch.equal(graph["test_nodes"].x, torch.tensor([[2], [3], [4], [5]]))
    assert torch.equal(graph["test_nodes", "to", "test_nodes"].edge_index, torch.tensor([[1, 2, 3], [0, 1, 2]]))
    assert torch.equal(graph["test_nodes"].original_indices, torch.tensor([[1], [2], [3], [4]]))


def test_remove_unconnected_nodes_with_ignore(device, graph_with_isolated_nodes):
    processor = RemoveUnconnectedNodes(nodes_name="test_nodes", ignore="mask_attr", save_mask_indices_to_attr=None)
    graph = processor.update_graph(graph_with_isolated_nodes)
    assert graph["test_nodes"].num_nodes == 5
    assert torch.equal(graph["test_nodes"].x, torch.tensor([[1], [2], [3], [4], [5]]))
    assert torch.equal(graph["test_nodes", "to", "test_nodes"].edge_index, torch.tensor([[2, 3, 4], [1, 2, 3]]))


@pytest.mark.parametrize(
    "nodes_name,ignore,save_mask_indices_to_attr",
    [("test_nodes", None, "original_indices"), ("test_nodes", "mask_attr", None), ("test_nodes", None, None)],
)
def test_remove_unconnected_nodes_parametrized(device, graph_with_isolated_nodes, nodes_name, ignore,
                                               save_mask_indices_to_attr):
    before = list(graph_with_isolated_nodes[nodes_name].node_attrs())
    processor = RemoveUnconnectedNodes(nodes_name=nodes_name, ignore=ignore,
                                       save_mask_indices_to_attr=save_mask_indices_to_attr)
    graph = processor.update_graph(graph_with_isolated_nodes)
    assert isinstance(graph, HeteroData)
    assert graph[nodes_name].num_nodes == (4 if ignore is None else 5)
    if save_mask_indices_to_attr:
        assert save_mask_indices_to_attr in graph[nodes_name]
        assert graph[nodes_name][save_mask_indices_to_attr].ndim == 2
    else:
        assert graph[nodes_name].node_attrs() == before


def test_random_two_node_sets(device):
    rng = np.random.default_rng(3)
    na, nb = 5000, 3000
    e_ab = np.stack([rng.integers(0, na, 4000), rng.integers(0, nb, 4000)]).astype(np.int32)
    e_aa = np.stack([rng.integers(0, na // 2, 1500), rng.integers(0, na // 2, 1500)]).astype(np.int32)
    g = HeteroData()
    g["a"].x = torch.arange(na * 2, dtype=torch.float32).reshape(na, 2)
    g["b"].x = torch.arange(nb * 2, dtype=torch.float32).reshape(nb, 2)
    g["a", "to", "b"].edge_index = torch.as_tensor(e_ab)
    g["a", "to", "a"].edge_index = torch.as_tensor(e_aa)
    g = RemoveUnconnectedNodes("a", save_mask_indices_to_attr="orig").update_graph(g)
    mask = np.zeros(na, bool)
    mask[e_ab[0]] = True
    mask[e_aa[0]] = True
    mask[e_aa[1]] = True
    new_id = np.cumsum(mask) - 1
    assert g["a"].x.shape[0] == mask.sum()
    assert np.array_equal(g["a"].x.numpy()[:, 0], np.arange(na)[mask] * 2.0)
    assert np.array_equal(g["a"].orig.numpy()[:, 0], np.where(mask)[0])
    assert np.array_equal(g["a", "to", "b"].edge_index.numpy(), np.stack([new_id[e_ab[0]], e_ab[1]]))
    assert np.array_equal(g["a", "to", "a"].edge_index.numpy(), np.stack([new_id[e_aa[0]], new_id[e_aa[1]]]))
    assert g["b"].x.shape[0] == nb
