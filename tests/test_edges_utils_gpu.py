"""GPU parity of the edge-list utilities (canonical sort, concat/unique, node masking)."""
import numpy as np
import pytest
import torch

from anemoi_graphs_amd import ops
from oracle import reference_path as rp
from tests import helpers as h

pytestmark = pytest.mark.gpu


def test_concat_edges_reference_vector(device, golden):
    """/root/reference/tests/test_utils.py:7-18."""
    g = golden("concat_edges")
    for a, b, want in (("e1", "e2", "e12"), ("r1", "r2", "r12")):
        cat = torch.from_numpy(np.concatenate([g[a], g[b]], axis=1).astype(np.int32)).to(device)
        got = ops.unique_edges(cat).cpu().numpy()
        assert np.array_equal(got, g[want])
    e12 = ops.unique_edges(torch.from_numpy(np.concatenate([g["e1"], g["e2"]], 1).astype(np.int32)).to(device))
    assert e12.cpu().tolist() == [[0, 1, 2, 3, 4], [-1, -2, -3, -4, -5]]


@pytest.mark.parametrize("n", [0, 1, 1000, 3_000_000])
def test_sort_and_unique_random(device, n):
    rng = np.random.default_rng(n)
    e = rng.integers(0, 5000, size=(2, n)).astype(np.int32)
    t = torch.from_numpy(e).to(device)
    assert np.array_equal(ops.sort_edges_(t.clone(), by_target=True).cpu().numpy(), h.canonical(e))
    order = np.lexsort((e[1], e[0]))
    assert np.array_equal(ops.sort_edges_(t.clone(), by_target=False).cpu().numpy(), e[:, order])
    if n:
        assert np.array_equal(ops.unique_edges(t).cpu().numpy(), np.unique(e, axis=1))


def test_mask_compaction(device):
    rng = np.random.default_rng(0)
    mask = rng.uniform(size=(100000, 1)) < 0.3
    ids = ops.mask_ids(torch.from_numpy(mask), device).cpu().numpy()
    assert np.array_equal(ids, np.where(mask[:, 0])[0])
    assert ops.mask_ids(torch.zeros(5, dtype=torch.bool), device).numel() == 0
