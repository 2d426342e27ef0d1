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


def test_ids_outside_their_table_raise_index_error(device):
    """Ids are followed only inside their table: numpy / torch indexing raises IndexError in the reference
    (edges/builder.py:176-193, processors/post_process.py:51-61,142-147); here the kernels skip such an id, end
    normally, and the host raises."""
    from anemoi_graphs_amd import settings

    x = torch.rand((100, 2), device=device)
    ids = torch.tensor([0, 7, 99], dtype=torch.int32, device=device)
    assert torch.equal(ops.gather_coords(x, ids), x[ids.long()])
    id_map = torch.arange(100, 200, dtype=torch.int32, device=device)
    assert ops.remap_ids_(ids.clone(), id_map).tolist() == [100, 107, 199]
    assert ops.invert_ids(ids, 100)[ids.long()].tolist() == [0, 1, 2]
    mask = torch.zeros(100, dtype=torch.uint8, device=device)
    assert int(ops.mark_ids_(mask, ids).sum()) == 3
    for bad in (100, -1, 2**31 - 1):
        wrong = torch.tensor([0, bad, 99], dtype=torch.int32, device=device)
        with pytest.raises(IndexError):
            ops.gather_coords(x, wrong)
        with pytest.raises(IndexError):
            ops.remap_ids_(wrong.clone(), id_map)
        with pytest.raises(IndexError):
            ops.invert_ids(wrong, 100)
        with pytest.raises(IndexError):
            ops.mark_ids_(torch.zeros(100, dtype=torch.uint8, device=device), wrong)
    # resident graphs: reported when the context ends, the valid entries were processed
    wrong = torch.tensor([3, 100, 4], dtype=torch.int32, device=device)
    with pytest.raises(IndexError):
        with settings.resident(True):
            out = ops.remap_ids_(wrong.clone(), id_map)
            assert out.tolist() == [103, -1, 104]
    ops.check_ids()  # nothing pending


def test_post_processor_rejects_edges_to_missing_nodes(device):
    from anemoi_graphs_amd.hetero import HeteroData
    from anemoi_graphs_amd.processors import RemoveUnconnectedNodes

    graph = HeteroData()
    graph["a"].x = torch.rand((10, 2))
    graph["a"].node_type = "LatLonNodes"
    graph[("a", "to", "a")].edge_index = torch.tensor([[0, 1, 12], [1, 2, 3]], dtype=torch.int32)
    with pytest.raises(IndexError):
        RemoveUnconnectedNodes("a").update_graph(graph)


def test_single_launch_scan_and_its_bounded_wait(device):
    """The chained-tile scan (cell starts, row offsets): correct on zeroed state words; on STALE state words -- the
    ticket counter not at zero, so that a tile waits for a predecessor that never runs -- its look-back gives up
    after the bound the host passes, every block ends, and the total reads -1.  (The wait is bounded by the kernel
    itself: nothing here hangs or faults the GPU.)"""
    import ctypes

    from anemoi_graphs_amd import _lib

    lib = _lib.load()
    stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    for n in (1, 4096, 4097, 1_000_003):
        values = torch.randint(0, 7, (n,), dtype=torch.int32, device=device)
        out = torch.empty(n + 1, dtype=torch.int32, device=device)
        state = torch.zeros(lib.ag_scan_i32_state_words(n), dtype=torch.int64, device=device)
        _lib.check(lib.ag_scan_i32(values.data_ptr(), out.data_ptr(), n, state.data_ptr(), 1 << 22, stream), "ag_scan_i32")
        want = torch.cumsum(values.long(), 0)
        assert int(out[n]) == int(want[-1]) and torch.equal(out[1:n].long(), want[:-1]) and int(out[0]) == 0
    n = 40_000  # ten tiles
    values = torch.ones(n, dtype=torch.int32, device=device)
    out = torch.zeros(n + 1, dtype=torch.int32, device=device)
    state = torch.zeros(lib.ag_scan_i32_state_words(n), dtype=torch.int64, device=device)
    state[0] = 1  # stale ticket counter: tile 0 never runs, tile 1 waits for it in vain
    _lib.check(lib.ag_scan_i32(values.data_ptr(), out.data_ptr(), n, state.data_ptr(), 16, stream), "ag_scan_i32")
    torch.cuda.synchronize(device)
    assert int(state[3]) == 1  # the error word
    # ... and the status word of the device, which the host reads (and clears) where it synchronises
    status = ctypes.c_int(0)
    _lib.check(lib.ag_status_take(ctypes.byref(status)), "ag_status_take")
    assert status.value == 1


def test_large_results_reach_the_host_whole(device):
    """Results of tens of megabytes go to page-locked host memory with a copy queued behind the work that makes the
    tensor: the host tensor holds every element when the call returns."""
    from anemoi_graphs_amd.utils import to_host

    torch.cuda.set_device(device)
    for shape, dtype in (((2, 3_000_001), torch.int64), ((5_000_003, 2), torch.float32), ((4_194_304,), torch.float32)):
        n = int(np.prod(shape))
        for _ in range(3):
            base = torch.arange(n, device=device, dtype=torch.float32)
            t = (base * 3.0 + 1.0).to(dtype).reshape(shape)  # queued, not waited for
            host = to_host(t)
            assert host.device.type == "cpu" and host.shape == t.shape and host.dtype == dtype
            assert torch.equal(host, t.cpu())
    view = torch.arange(12_000_000, device=device, dtype=torch.float32).reshape(2, -1).t()  # not contiguous
    assert torch.equal(to_host(view), view.cpu())


def test_device_mirror_follows_the_host_tensor(device):
    """Coordinates given as a host tensor are uploaded once; a write torch counts, or one through the numpy array
    behind the tensor, brings a fresh upload."""
    arr = np.random.default_rng(1).random((4000, 2)).astype(np.float32)
    x = torch.from_numpy(arr)
    first = ops.coords(x, device)
    assert ops.coords(x, device) is first and torch.equal(first.cpu(), x)
    arr[:] = arr[::-1].copy()
    second = ops.coords(x, device)
    assert second is not first and torch.equal(second.cpu(), x)
    x.mul_(0.5)
    third = ops.coords(x, device)
    assert third is not second and torch.equal(third.cpu(), x)
    ops.forget_mirrors()
    assert ops.coords(x, device) is not third
