"""GPU: a prefix sum that gives up inside a binning does not pass unnoticed.  The condition is forced from the host
(a test hook makes the binning leave the state words of its prefix sum uncleared: every tile of the scan then draws a
ticket beyond the last tile and ends at once), so every kernel still ends normally: the binning kernel that finds
the total wrong leaves an empty, fully initialised binning and sets the status word of the device, and from then
until the host has taken the status every kernel of the searches returns at its first instruction."""
import numpy as np
import pytest
import torch

from anemoi_graphs_amd import _lib, grids, ops, settings

pytestmark = pytest.mark.gpu


def test_a_scan_that_gives_up_inside_a_binning_surfaces_as_an_error(device):
    """A binning whose prefix sum gave up: the call raises LibraryError (graph tensors on the host: right away;
    resident: when the context ends), the binning it leaves is empty but sound -- a cut-off query on it finds
    nothing and stays inside its arrays -- and with the hook off everything is as before."""
    lib = _lib.load()
    src = torch.from_numpy(grids.random_sphere(200_000, seed=1)).to(device)
    tgt = torch.from_numpy(grids.random_sphere(500, seed=2)).to(device)
    want_edges = ops.radius_edges(src, tgt, 0.05)
    want_ids = ops.knn(src, tgt, 3)
    ops.check_status()
    kept = None
    _lib.check(lib.ag_debug_poison_scans(1), "ag_debug_poison_scans")
    try:
        with pytest.raises(_lib.LibraryError, match="binning was left empty|prefix sum gave up"):
            ops.bin_points(src, 0.05)
        ops.check_status()  # read and cleared by the failure above
        with pytest.raises(_lib.LibraryError):
            with settings.resident(True):  # no synchronisation per call: raised when the context ends
                kept = ops.bin_points(src, 0.05)
        # a failure of the caller's own inside a resident build: the error words of that build are dropped with it
        with pytest.raises(ZeroDivisionError):
            with settings.resident(True):
                ops.bin_points(src, 0.05)
                1 / 0
        ops.check_status()  # nothing left over
    finally:
        _lib.check(lib.ag_debug_poison_scans(0), "ag_debug_poison_scans")
    # the binning the failed call left: every cell empty, every record initialised -- nothing within any radius
    assert kept is not None
    assert ops.radius_edges(None, tgt, 0.05, src_binning=kept).shape[1] == 0
    # with the hook off everything is as before, and the status stays clear
    assert torch.equal(ops.radius_edges(src, tgt, 0.05), want_edges)
    assert torch.equal(ops.knn(src, tgt, 3), want_ids)
    good = ops.bin_points(src, 0.05)
    assert torch.equal(ops.radius_edges(None, tgt, 0.05, src_binning=good), want_edges)
    ops.check_status()


def test_the_scan_reports_a_give_up_through_the_status_word(device):
    """The single-launch prefix sum on its own, with state words that are not clear and a wait of one poll: the
    total comes back as -1 and the status word of the device is set (ag_status_take reads and clears it)."""
    import ctypes

    lib = _lib.load()
    n = 100_000
    values = torch.ones(n, dtype=torch.int32, device=device)
    out = torch.zeros(n + 1, dtype=torch.int32, device=device)
    words = lib.ag_scan_i32_state_words(n)
    state = torch.zeros(words, dtype=torch.int64, device=device)
    _lib.check(lib.ag_scan_i32(ops._ptr(values), ops._ptr(out), n, ops._ptr(state), 1 << 20, ops._stream(device)), "scan")
    assert int(out[n]) == n and np.array_equal(out[:n].cpu().numpy(), np.arange(n))
    ops.check_status()
    state.fill_(0x40000000)  # tickets beyond the last tile
    _lib.check(lib.ag_scan_i32(ops._ptr(values), ops._ptr(out), n, ops._ptr(state), 1, ops._stream(device)), "scan")
    status = ctypes.c_int(0)
    _lib.check(lib.ag_status_take(ctypes.byref(status)), "ag_status_take")
    assert status.value == 1
    _lib.check(lib.ag_status_take(ctypes.byref(status)), "ag_status_take")
    assert status.value == 0
