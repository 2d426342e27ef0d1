"""Tensor-level entry to the HIP kernels: every function here launches on the current HIP stream
of the tensors' device and returns device tensors.  PyTorch only provides memory and streams.
"""
from __future__ import annotations

import contextlib
import ctypes
import math
import weakref

import torch

from . import _lib

NORMS = {None: 0, "l1": 1, "l2": 2, "unit-max": 3, "unit-range": 4, "unit-std": 5}


def device_of(*tensors: torch.Tensor) -> torch.device:
    """The HIP device the work runs on: that of the first device tensor, else the current one."""
    for t in tensors:
        if isinstance(t, torch.Tensor) and t.is_cuda:
            return t.device
    if not torch.cuda.is_available():
        raise RuntimeError(
            "anemoi_graphs_amd needs an AMD GPU (MI355X / gfx950): no HIP device is visible and "
            "there is no CPU fallback."
        )
    return torch.device("cuda", torch.cuda.current_device())


# ---------------------------------------------------------------------------------------------------
# Device mirrors of host tensors.  With graph tensors on the host (the package's default, as in the reference) the
# same tensors cross the link again and again: the data coordinates once per builder and attribute that reads them,
# every edge list down to the host and up again for its attributes.  A host tensor that was uploaded, or that is the
# host copy of a device result (``utils.to_host``), keeps its device twin here -- found again by the identity of the
# tensor object, its version counter (any write torch knows of invalidates it) and a fingerprint of its contents
# (below), dropped when the host tensor dies, and evicted oldest-first beyond a budget of device memory.
# ---------------------------------------------------------------------------------------------------
_MIRROR_BUDGET = 4 << 30  # bytes of device memory the mirrors may hold
_MIRRORS: dict = {}       # (id(host tensor), device index, dtype) -> (weak reference, version, device tensor, fingerprint)


def host_fingerprint(t: torch.Tensor):
    """The bytes of about 500 evenly spaced elements of a contiguous host tensor (None for any other tensor).
    torch counts the writes it makes itself; a write through a numpy array that shares the memory, or through
    ``.data``, is invisible to that counter.  Such a write is noticed here when it changes one of the sampled
    elements -- a grid rewritten in place does, an edit of a few nodes may not: ``forget_mirrors()`` after one."""
    if t.device.type != "cpu" or not t.is_contiguous():
        return None
    flat = t.reshape(-1)
    if flat.numel() == 0:
        return b""
    return flat[:: max(1, flat.numel() // 509)].contiguous().view(torch.uint8).numpy().tobytes()


def forget_mirrors() -> None:
    """Drop every device mirror of a host tensor (after writing to host tensors behind torch's back)."""
    _MIRRORS.clear()


def _mirror_drop(key) -> None:
    _MIRRORS.pop(key, None)


def mirror_put(host: torch.Tensor, twin: torch.Tensor) -> None:
    """Remember ``twin`` (a device tensor with the contents of ``host``) as the device side of ``host``."""
    nbytes = twin.numel() * twin.element_size()
    if nbytes > _MIRROR_BUDGET // 2 or twin.dtype != host.dtype or tuple(twin.shape) != tuple(host.shape):
        return
    print_of = host_fingerprint(host)
    if print_of is None:
        return
    index = twin.device.index if twin.device.index is not None else torch.cuda.current_device()
    key = (id(host), index, twin.dtype)
    held = sum(v[2].numel() * v[2].element_size() for v in _MIRRORS.values())
    for old in list(_MIRRORS):  # (insertion order: oldest first)
        if held + nbytes <= _MIRROR_BUDGET and len(_MIRRORS) < 256:
            break
        held -= _MIRRORS[old][2].numel() * _MIRRORS[old][2].element_size()
        del _MIRRORS[old]
    _MIRRORS[key] = (weakref.ref(host, lambda _ref, key=key: _mirror_drop(key)), host._version, twin, print_of)


def _mirror_get(host: torch.Tensor, device: torch.device, dtype: torch.dtype):
    index = device.index if device.index is not None else torch.cuda.current_device()
    found = _MIRRORS.get((id(host), index, dtype))
    if found is not None and found[0]() is host and found[1] == host._version and found[3] == host_fingerprint(host):
        return found[2]
    return None


def as_device(x, device: torch.device, dtype: torch.dtype) -> torch.Tensor:
    """Contiguous tensor of ``dtype`` on ``device`` with the contents of ``x``; a host tensor's upload is kept as
    its mirror, and found again the next time."""
    if isinstance(x, torch.Tensor) and x.device.type == "cpu" and x.dtype == dtype:
        twin = _mirror_get(x, device, dtype)
        if twin is None:
            twin = x.to(device=device).contiguous()
            mirror_put(x, twin)
        return twin
    return torch.as_tensor(x).to(device=device, dtype=dtype).contiguous()


def coords(x, device: torch.device) -> torch.Tensor:
    """float32 contiguous [N,2] copy/view of node coordinates on ``device``."""
    if (isinstance(x, torch.Tensor) and x.dtype is torch.float32 and x.device == device and x.ndim == 2
            and x.shape[1] == 2 and x.is_contiguous()):
        return x  # (already what is wanted: the conversions below cost the host microseconds even when they do nothing)
    t = torch.as_tensor(x)
    if t.ndim != 2 or t.shape[1] != 2:
        raise ValueError(f"coordinates must have shape [N, 2], got {tuple(t.shape)}")
    return as_device(t, device, torch.float32)


_NO_SWITCH = contextlib.nullcontext()


def _on(device: torch.device):
    """Context that makes ``device`` current; nothing to enter when it already is (the usual case: entering
    and leaving torch's device guard costs the host several microseconds per call)."""
    if device.index is None or device.index == torch.cuda.current_device():
        return _NO_SWITCH
    return torch.cuda.device(device)


def _ptr(t: torch.Tensor | None):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)  # (the handle without a Stream object around it)


def _stream(device: torch.device):
    index = device.index if device.index is not None else torch.cuda.current_device()
    _USED_DEVICES.add(index)
    if _RAW_STREAM is not None:
        return ctypes.c_void_p(_RAW_STREAM(index))
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _workspace(nbytes: int, device: torch.device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


# ---------------------------------------------------------------------------------------------------
# Ids supplied by the caller (an edge_index, a list of node ids) are bounds-checked inside the kernels that
# follow them: an id outside its table is never dereferenced and sets a per-device error word.  numpy /
# torch raise IndexError for such an id in the reference (edges/attributes.py:144-148,
# processors/post_process.py:51-61); here the word is read -- and IndexError raised -- right after the call
# when graph tensors live on the host (the results are copied there anyway), and at the next
# ``check_ids()`` when they stay on the device (``settings.resident``: leaving that context, and
# ``GraphCreator.create``, call it), so that a resident build takes no synchronisation for it.
_BAD_IDS: dict[int, torch.Tensor] = {}


def _bad_word(device: torch.device) -> torch.Tensor:
    index = device.index if device.index is not None else torch.cuda.current_device()
    word = _BAD_IDS.get(index)
    if word is None:
        word = _BAD_IDS[index] = torch.zeros(1, dtype=torch.int32, device=torch.device("cuda", index))
    return word


_USED_DEVICES: set[int] = set()  # devices a call of this module has launched work on


def check_status(device: torch.device | None = None) -> None:
    """Raise LibraryError if a kernel of the library had to give up since the last check (``ag_status_take``:
    a prefix sum that met state words that were not clear, a binning that was left empty for that reason).
    Synchronises with the device."""
    lib = _lib.load()
    indices = sorted(_USED_DEVICES) if device is None else [
        device.index if device.index is not None else torch.cuda.current_device()]
    failed = []
    for index in indices:
        with torch.cuda.device(index):
            status = ctypes.c_int(0)
            _lib.check(lib.ag_status_take(ctypes.byref(status)), "ag_status_take")
        if status.value:
            failed.append((index, status.value))
    if failed:
        what = {1: "a prefix sum gave up waiting for a tile (state words not cleared?)",
                2: "a binning was left empty because the prefix sum of its cell counts did not add up"}
        raise _lib.LibraryError("; ".join(f"cuda:{i}: {what.get(v, f'status {v}')}" for i, v in failed)
                                + " -- the results of the calls since the last check are not valid")


def check_ids(device: torch.device | None = None) -> None:
    """Raise IndexError if a kernel met an id outside its table since the last check, LibraryError if one had to
    give up (``check_status``).  Synchronises."""
    words = list(_BAD_IDS.values()) if device is None else [_bad_word(device)]
    bad = False
    for word in words:
        if int(word.item()):
            word.zero_()
            bad = True
    check_status(device)
    if bad:
        raise IndexError("index out of bounds: an edge_index or id array refers to a node that does not exist")


def discard_checks() -> None:
    """Read and drop whatever the error words hold (the caller is already reporting a failure of its own)."""
    for word in _BAD_IDS.values():
        word.zero_()
    try:
        check_status()
    except _lib.LibraryError:
        pass


def _ids_checked(device: torch.device) -> None:
    from . import settings

    if not settings.device_resident:
        check_ids(device)


def _status_checked(device: torch.device) -> None:
    """After a search or a binning: with graph tensors on the host the status is read at once (the results are
    about to be copied anyway); resident builds read it when their context ends."""
    from . import settings

    if not settings.device_resident:
        check_status(device)


class Binning:
    """A node set binned once for several searches (``ag_binning_build``): the device coordinates, the buffer
    that holds the binned records and cell tables, and the cell size they were built with."""

    def __init__(self, x: torch.Tensor, buffer: torch.Tensor, cell: float, key) -> None:
        self.x, self.buffer, self.cell, self.key = x, buffer, float(cell), key
        # the stream the binning was built on, and the point of it at which the buffer is complete
        self.stream = torch.cuda.current_stream(buffer.device)
        self.ready = torch.cuda.Event()
        self.ready.record(self.stream)

    @property
    def n(self) -> int:
        return int(self.x.shape[0])

    def wait(self) -> None:
        """Make the current stream wait for the binning (nothing to do on the stream that built it)."""
        current = torch.cuda.current_stream(self.buffer.device)
        if current != self.stream:
            current.wait_event(self.ready)


class TensorStamp:
    """What a value derived from a tensor (a binning, a reference distance, a density) remembers of it: the tensor
    OBJECT -- held, so that its memory cannot be handed to another tensor while the value is kept -- and the
    version counter it had.  ``matches`` is true for that very object as long as no operation of torch has written
    to it; for a host tensor also while its fingerprint stands (``host_fingerprint``: what torch cannot count, a
    write through a numpy view or ``.data``, is noticed when it changes one of the sampled elements)."""

    __slots__ = ("ref", "version", "print_of")

    def __init__(self, t: torch.Tensor):
        self.ref, self.version, self.print_of = t, t._version, host_fingerprint(t)

    def matches(self, t) -> bool:
        return self.ref is t and self.version == t._version and self.print_of == host_fingerprint(t)


def tensor_key(t: torch.Tensor) -> TensorStamp:
    return TensorStamp(t)


_DENSITIES: dict = {}  # id(tensor) -> (weak reference, version, density): valid while that very tensor is alive and unchanged


def point_density(x: torch.Tensor) -> float:
    """How many times more nodes per steradian the set has, where it has nodes, than as many nodes spread over the
    sphere (``ag_point_density``): 1 for a global grid, 1/f for a regional one covering the fraction f of the
    sphere.  What the k-NN searches size their cells with.  Measured on the device and read back (one
    synchronisation), once per tensor object."""
    src = torch.as_tensor(x)
    if src.shape[0] < 1024:
        return 1.0
    found = _DENSITIES.get(id(src))
    if found is not None and found[0]() is src and found[1] == src._version:
        return found[2]
    lib = _lib.load()
    dev = device_of(src)
    xd = coords(src, dev)
    n = xd.shape[0]
    with _on(dev):
        out = torch.empty((), dtype=torch.float64, device=dev)
        ws = _workspace(lib.ag_point_density_workspace_bytes(n), dev)
        _lib.check(lib.ag_point_density(_ptr(xd), n, _ptr(out), _ptr(ws), ws.numel(), _stream(dev)), "ag_point_density")
        value = float(out.item())
    if len(_DENSITIES) >= 64:
        for dead in [k for k, v in _DENSITIES.items() if v[0]() is None]:
            del _DENSITIES[dead]
        if len(_DENSITIES) >= 64:
            _DENSITIES.clear()
    _DENSITIES[id(src)] = (weakref.ref(src), src._version, value)
    return value


def _density(given: float | None, x: torch.Tensor) -> float:
    return float(given) if given is not None else point_density(x)


def radius_cell(ns: int, radius: float) -> float:
    """Cell size a cut-off search of this radius over ``ns`` sources works best with."""
    return float(_lib.load().ag_radius_cell(int(ns), float(radius)))


def bin_points(x: torch.Tensor, cell: float, device: torch.device | None = None) -> Binning:
    """Bin a node set (any tensor-like [N,2]) into cells of about ``cell`` radians, to be handed to
    ``radius_edges(src_binning=...)`` as its sources and ``knn`` / ``knn_edges(tgt_binning=...)`` as its targets."""
    lib = _lib.load()
    src = torch.as_tensor(x)
    dev = device_of(src) if device is None else device
    xd = coords(src, dev)
    n = xd.shape[0]
    if n < 1:
        raise ValueError("cannot bin an empty node set")
    with _on(dev):
        buf = torch.empty(lib.ag_binning_bytes(n), dtype=torch.uint8, device=dev)
        _lib.check(lib.ag_binning_build(_ptr(xd), n, float(cell), _ptr(buf), buf.numel(), _stream(dev)), "ag_binning_build")
    _status_checked(dev)
    return Binning(xd, buf, cell, tensor_key(src))


def knn(src: torch.Tensor, tgt: torch.Tensor, k: int, return_dist: bool = False, return_stats: bool = False,
        max_dist: float | None = None, tgt_binning: Binning | None = None, src_density: float | None = None,
        tgt_density: float | None = None):
    """ids [Nt,k] int32 of the k nearest sources of every target, ascending distance.  ``max_dist`` (radians):
    neighbours farther than that are not searched for; their ids are -1 and their distances infinity.
    ``src_density`` / ``tgt_density``: see ``point_density`` (measured when not given; they decide how fast the
    search runs, not what it returns)."""
    lib = _lib.load()
    dev = device_of(src, tgt) if tgt_binning is None else tgt_binning.x.device
    src_density = _density(src_density, src)
    if tgt_binning is None:
        tgt_density = _density(tgt_density, tgt)
    src = coords(src, dev)
    tgt = coords(tgt, dev) if tgt_binning is None else tgt_binning.x
    ns, nt = src.shape[0], tgt.shape[0]
    if k > ns:
        # what scikit-learn raises in the reference (sklearn/neighbors/_base.py, kneighbors)
        raise ValueError(f"Expected n_neighbors <= n_samples_fit, but n_neighbors = {k}, n_samples_fit = {ns}")
    with _on(dev):
        out = torch.empty((nt, k), dtype=torch.int32, device=dev)
        dist = torch.empty((nt, k), dtype=torch.float64, device=dev) if return_dist else None
        stats = torch.zeros(4, dtype=torch.int64, device=dev)
        if tgt_binning is not None and max_dist is None and nt > 0:
            tgt_binning.wait()
            ws = _workspace(lib.ag_knn_binned_workspace_bytes(ns, nt, k, src_density), dev)
            _lib.check(
                lib.ag_knn_binned(_ptr(src), ns, _ptr(tgt), nt, _ptr(tgt_binning.buffer), tgt_binning.cell, k, src_density,
                                  _ptr(out), _ptr(dist), _ptr(stats), _ptr(ws), ws.numel(), _stream(dev)),
                "ag_knn_binned",
            )
            _status_checked(dev)
            res = (out,) + ((dist,) if return_dist else ()) + ((stats,) if return_stats else ())
            return res[0] if len(res) == 1 else res
        if tgt_density is None:  # (a bounded search with a binning of the targets: they are searched unbinned)
            tgt_density = point_density(tgt)
        ws = _workspace(lib.ag_knn_workspace_bytes(ns, nt, k, src_density), dev)
        if max_dist is None:
            _lib.check(
                lib.ag_knn(_ptr(src), ns, _ptr(tgt), nt, k, src_density, tgt_density, _ptr(out), _ptr(dist), _ptr(stats),
                           _ptr(ws), ws.numel(), _stream(dev)),
                "ag_knn",
            )
        else:
            _lib.check(
                lib.ag_knn_within(_ptr(src), ns, _ptr(tgt), nt, k, float(max_dist), src_density, tgt_density, _ptr(out),
                                  _ptr(dist), _ptr(stats), _ptr(ws), ws.numel(), _stream(dev)),
                "ag_knn_within",
            )
    _status_checked(dev)
    res = (out,)
    if return_dist:
        res += (dist,)
    if return_stats:
        res += (stats,)
    return res[0] if len(res) == 1 else res


def knn_edges(src: torch.Tensor, tgt: torch.Tensor, k: int, tgt_binning: Binning | None = None,
              src_density: float | None = None, tgt_density: float | None = None) -> torch.Tensor:
    """edge_index [2, Nt*k] int32 of the k-NN graph: row 0 the k nearest sources of every target
    (ascending distance), row 1 the target id.  ``tgt_binning``: a kept binning of the targets.
    ``src_density`` / ``tgt_density``: as for ``knn``."""
    lib = _lib.load()
    dev = device_of(src, tgt) if tgt_binning is None else tgt_binning.x.device
    src_density = _density(src_density, src)
    tgt_density = 1.0 if tgt_binning is not None else _density(tgt_density, tgt)
    src = coords(src, dev)
    tgt = coords(tgt, dev) if tgt_binning is None else tgt_binning.x
    ns, nt = src.shape[0], tgt.shape[0]
    if k > ns:
        raise ValueError(f"Expected n_neighbors <= n_samples_fit, but n_neighbors = {k}, n_samples_fit = {ns}")
    with _on(dev):
        edge_index = torch.empty((2, nt * k), dtype=torch.int32, device=dev)
        binned = tgt_binning is not None and nt > 0
        if binned:
            tgt_binning.wait()
        ws = _workspace(lib.ag_knn_binned_workspace_bytes(ns, nt, k, src_density) if binned
                        else lib.ag_knn_workspace_bytes(ns, nt, k, src_density), dev)
        # (both rows come from the search kernels: the target row needs no pass of its own)
        _lib.check(
            lib.ag_knn_edges(_ptr(src), ns, _ptr(tgt), nt, _ptr(tgt_binning.buffer) if binned else None,
                             tgt_binning.cell if binned else 0.0, k, src_density, tgt_density, _ptr(edge_index[0]),
                             _ptr(edge_index[1]), None, _ptr(ws), ws.numel(), _stream(dev)),
            "ag_knn_edges",
        )
    _status_checked(dev)
    return edge_index


def row_ids(rows: int, k: int, device: torch.device, out: torch.Tensor | None = None) -> torch.Tensor:
    """int32 [rows*k]: 0 (k times), 1 (k times), ... -- the target row of a k-NN edge list
    (written into ``out``, a contiguous int32 device tensor of that length, when given)."""
    lib = _lib.load()
    dev = torch.device(device)
    with _on(dev):
        if out is None:
            out = torch.empty(rows * k, dtype=torch.int32, device=dev)
        assert out.dtype == torch.int32 and out.is_contiguous() and out.numel() == rows * k
        _lib.check(lib.ag_fill_row_ids(_ptr(out), rows, k, _stream(dev)), "ag_fill_row_ids")
    return out


def nn_maxdist(x: torch.Tensor, density: float | None = None) -> torch.Tensor:
    """0-dim float64 device tensor: max over the nodes of the distance to the nearest other node.
    ``density``: of the node set, see ``point_density`` (measured when not given)."""
    lib = _lib.load()
    dev = device_of(x)
    density = _density(density, x)
    x = coords(x, dev)
    n = x.shape[0]
    if n < 2:
        raise ValueError(f"Expected n_neighbors <= n_samples_fit, but n_neighbors = 2, n_samples_fit = {n}")
    with _on(dev):
        out = torch.zeros((), dtype=torch.float64, device=dev)
        ws = _workspace(lib.ag_nn_maxdist_workspace_bytes(n, density), dev)
        _lib.check(lib.ag_nn_maxdist(_ptr(x), n, density, _ptr(out), _ptr(ws), ws.numel(), _stream(dev)), "ag_nn_maxdist")
    _status_checked(dev)
    return out


def radius_edges(src: torch.Tensor, tgt: torch.Tensor, radius: float, src_binning: Binning | None = None) -> torch.Tensor:
    """edge_index [2,E] int32 (row 0 source id, row 1 target id), sorted by (target, source).
    ``src_binning``: a kept binning of the sources (``bin_points``), used instead of binning them here."""
    lib = _lib.load()
    dev = device_of(src, tgt) if src_binning is None else src_binning.x.device
    src = coords(src, dev) if src_binning is None else src_binning.x
    tgt = coords(tgt, dev)
    ns, nt = src.shape[0], tgt.shape[0]
    radius = float(radius)
    with _on(dev):
        offsets = torch.empty(nt + 2, dtype=torch.int64, device=dev)
        if src_binning is None:
            ws = _workspace(lib.ag_radius_workspace_bytes(ns, nt), dev)
            _lib.check(
                lib.ag_radius_count(_ptr(src), ns, _ptr(tgt), nt, radius, _ptr(offsets), _ptr(ws), ws.numel(), _stream(dev)),
                "ag_radius_count",
            )
        else:
            src_binning.wait()
            ws = _workspace(lib.ag_radius_binned_workspace_bytes(ns, nt), dev)
            _lib.check(
                lib.ag_radius_count_binned(_ptr(src), ns, _ptr(src_binning.buffer), src_binning.cell, _ptr(tgt), nt, radius,
                                           _ptr(offsets), _ptr(ws), ws.numel(), _stream(dev)),
                "ag_radius_count_binned",
            )
        num_edges, max_row = offsets[nt:].tolist()  # the one host synchronisation: output size, longest row
        if num_edges < 0:
            discard_checks()
            raise _lib.LibraryError("ag_radius_count: the scan of the row lengths gave up waiting for a tile "
                                    "(workspace state not cleared?)")
        _status_checked(dev)
        edge_index = torch.empty((2, num_edges), dtype=torch.int32, device=dev)
        if num_edges and src_binning is None:
            _lib.check(
                lib.ag_radius_fill(_ptr(src), ns, _ptr(tgt), nt, radius, _ptr(offsets), num_edges, max_row,
                                   _ptr(edge_index[0]), _ptr(edge_index[1]), _ptr(ws), ws.numel(), _stream(dev)),
                "ag_radius_fill",
            )
        elif num_edges:
            _lib.check(
                lib.ag_radius_fill_binned(_ptr(src), ns, _ptr(src_binning.buffer), src_binning.cell, _ptr(tgt), nt, radius,
                                          _ptr(offsets), num_edges, max_row, _ptr(edge_index[0]), _ptr(edge_index[1]),
                                          _ptr(ws), ws.numel(), _stream(dev)),
                "ag_radius_fill_binned",
            )
    return edge_index


def sort_edges_(edge_index: torch.Tensor, by_target: bool = True) -> torch.Tensor:
    """Canonical in-place sort of a [2,E] int32 device edge list."""
    lib = _lib.load()
    dev = edge_index.device
    assert edge_index.dtype == torch.int32 and edge_index.is_contiguous() and edge_index.shape[0] == 2
    num_edges = edge_index.shape[1]
    with _on(dev):
        ws = _workspace(lib.ag_sort_edges_workspace_bytes(num_edges), dev)
        _lib.check(
            lib.ag_sort_edges(_ptr(edge_index[0]), _ptr(edge_index[1]), num_edges, int(by_target), _ptr(ws), ws.numel(),
                              _stream(dev)),
            "ag_sort_edges",
        )
    return edge_index


def unique_edges(edge_index: torch.Tensor) -> torch.Tensor:
    """Columns sorted by (source, target) with duplicates removed (``torch.unique(dim=1)`` order)."""
    lib = _lib.load()
    dev = device_of(edge_index)
    e = edge_index.to(device=dev, dtype=torch.int32).contiguous().clone()
    num_edges = e.shape[1]
    with _on(dev):
        count = torch.zeros((), dtype=torch.int64, device=dev)
        ws = _workspace(lib.ag_sort_edges_workspace_bytes(num_edges), dev)
        _lib.check(
            lib.ag_unique_edges(_ptr(e[0]), _ptr(e[1]), num_edges, _ptr(count), _ptr(ws), ws.numel(), _stream(dev)),
            "ag_unique_edges",
        )
        n = int(count.item())
    return e[:, :n].contiguous()


def mask_ids(mask: torch.Tensor, device: torch.device) -> torch.Tensor:
    """int32 ids of the true entries of a boolean node mask."""
    lib = _lib.load()
    m = torch.as_tensor(mask).reshape(-1).to(device=device, dtype=torch.uint8).contiguous()
    n = m.numel()
    with _on(device):
        ids = torch.empty(n, dtype=torch.int32, device=device)
        count = torch.zeros((), dtype=torch.int64, device=device)
        ws = _workspace(lib.ag_mask_compact_workspace_bytes(n), device)
        _lib.check(lib.ag_mask_compact(_ptr(m), n, _ptr(ids), _ptr(count), _ptr(ws), ws.numel(), _stream(device)),
                   "ag_mask_compact")
        c = int(count.item())
    return ids[:c]


def gather_coords(x: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
    """out[i] = x[ids[i]] (float32 [N,2] device coordinates, int32 device ids); IndexError for an id outside x."""
    lib = _lib.load()
    dev = x.device
    assert x.dtype == torch.float32 and x.is_contiguous() and ids.dtype == torch.int32 and ids.is_contiguous()
    out = torch.empty((ids.numel(), 2), dtype=torch.float32, device=dev)
    with _on(dev):
        _lib.check(lib.ag_gather_coords(_ptr(x), x.shape[0], _ptr(ids), ids.numel(), _ptr(out), _ptr(_bad_word(dev)),
                                        _stream(dev)), "ag_gather_coords")
    _ids_checked(dev)
    return out


def remap_ids_(values: torch.Tensor, id_map: torch.Tensor) -> torch.Tensor:
    """values[i] = id_map[values[i]] in place; IndexError for a value outside the map."""
    lib = _lib.load()
    dev = values.device
    assert values.dtype == torch.int32 and values.is_contiguous() and id_map.dtype == torch.int32
    assert id_map.is_contiguous() and id_map.device == dev
    with _on(dev):
        _lib.check(lib.ag_remap_ids(_ptr(values), values.numel(), _ptr(id_map), id_map.numel(), _ptr(_bad_word(dev)),
                                    _stream(dev)), "ag_remap_ids")
    _ids_checked(dev)
    return values


def mark_ids_(mask: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
    """mask[ids] = 1 in place (uint8 device mask, int32 device ids); IndexError for an id outside the mask."""
    lib = _lib.load()
    dev = mask.device
    assert mask.dtype == torch.uint8 and mask.is_contiguous()
    ids = ids.to(device=dev, dtype=torch.int32).contiguous()
    with _on(dev):
        _lib.check(lib.ag_mark_ids(_ptr(ids), ids.numel(), _ptr(mask), mask.numel(), _ptr(_bad_word(dev)),
                                   _stream(dev)), "ag_mark_ids")
    _ids_checked(dev)
    return mask


def invert_ids(ids: torch.Tensor, size: int) -> torch.Tensor:
    """int32 [size] map with map[ids[j]] = j and -1 elsewhere; IndexError for an id outside [0, size)."""
    lib = _lib.load()
    dev = ids.device
    assert ids.dtype == torch.int32 and ids.is_contiguous()
    with _on(dev):
        out = torch.full((size,), -1, dtype=torch.int32, device=dev)
        _lib.check(lib.ag_invert_ids(_ptr(ids), ids.numel(), _ptr(out), size, _ptr(_bad_word(dev)), _stream(dev)),
                   "ag_invert_ids")
    _ids_checked(dev)
    return out


def icosphere(resolution: int, device: torch.device | None = None, ordered: bool = True):
    """(xyz float64 [V,3], faces int32 [F,3], latlon float32 [V,2] in node order, order int32 [V],
    rank int32 [V], level_edges int64 [E_all] = unique mesh edges of every level 0..resolution).
    ``ordered=False``: no node order is computed -- latlon comes in vertex order, order and rank are None
    (for the builders that keep a part of the vertices and order that part)."""
    lib = _lib.load()
    dev = device_of() if device is None else device
    v, f, e = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
    _lib.check(lib.ag_icosphere_sizes(resolution, ctypes.byref(v), ctypes.byref(f), ctypes.byref(e)),
               "ag_icosphere_sizes")
    nv, nf, ne = v.value, f.value, e.value
    with _on(dev):
        xyz = torch.empty((nv, 3), dtype=torch.float64, device=dev)
        faces = torch.empty((nf, 3), dtype=torch.int32, device=dev)
        latlon = torch.empty((nv, 2), dtype=torch.float32, device=dev)
        order = torch.empty(nv, dtype=torch.int32, device=dev) if ordered else None
        rank = torch.empty(nv, dtype=torch.int32, device=dev) if ordered else None
        level_edges = torch.empty(ne, dtype=torch.int64, device=dev)
        ws = _workspace(lib.ag_icosphere_workspace_bytes(resolution), dev)
        _lib.check(
            lib.ag_icosphere(resolution, _ptr(xyz), _ptr(faces), _ptr(latlon), _ptr(order), _ptr(rank),
                             _ptr(level_edges), _ptr(ws), ws.numel(), _stream(dev)),
            "ag_icosphere",
        )
    return xyz, faces, latlon, order, rank, level_edges


def mesh_maxdist(resolution: int, level_edges: torch.Tensor, latlon: torch.Tensor, rank: torch.Tensor) -> torch.Tensor:
    """0-dim float64 device tensor: the grid reference distance of the COMPLETE icosphere of this resolution -- what
    ``nn_maxdist(latlon)`` finds by a search -- from its mesh edges (``ag_mesh_maxdist``).  ``level_edges``, ``latlon``
    (node order) and ``rank`` as ``icosphere(resolution)`` returns them."""
    lib = _lib.load()
    dev = level_edges.device
    latlon = coords(latlon, dev)
    with _on(dev):
        out = torch.empty((), dtype=torch.float64, device=dev)
        ws = _workspace(lib.ag_mesh_maxdist_workspace_bytes(resolution), dev)
        _lib.check(lib.ag_mesh_maxdist(resolution, _ptr(level_edges), _ptr(latlon), _ptr(rank), _ptr(out), _ptr(ws),
                                       ws.numel(), _stream(dev)), "ag_mesh_maxdist")
    return out


def node_order(latlon: torch.Tensor) -> torch.Tensor:
    """int32 [N] permutation that orders nodes by latitude descending (ties: longitude descending,
    then id descending)."""
    lib = _lib.load()
    dev = device_of(latlon)
    latlon = coords(latlon, dev)
    n = latlon.shape[0]
    with _on(dev):
        order = torch.empty(n, dtype=torch.int32, device=dev)
        ws = _workspace(lib.ag_node_order_workspace_bytes(n), dev)
        _lib.check(lib.ag_node_order(_ptr(latlon), n, _ptr(order), _ptr(ws), ws.numel(), _stream(dev)), "ag_node_order")
    return order


def multiscale_edges(max_resolution: int, levels: list[int], x_hops: int, rank: torch.Tensor,
                     level_edges: torch.Tensor, active: torch.Tensor | None = None) -> torch.Tensor:
    """edge_index [2,E] int32 of the multi-scale icosahedral edges, sorted by (target, source)."""
    lib = _lib.load()
    dev = rank.device
    lv = (ctypes.c_int32 * len(levels))(*[int(v) for v in levels])
    act = None if active is None else active.to(device=dev, dtype=torch.uint8).contiguous()
    with _on(dev):
        if act is None and x_hops == 1 and len(set(int(v) for v in levels)) == len(levels):
            # The complete icosphere: the edge count is a closed form (level l has 30 * 4^l mesh edges, each listed
            # in both directions, and edges of different levels never coincide), so nothing is asked of the device
            # before the output is allocated, and one call builds the rows without a counting pass.
            num_edges = int(lib.ag_multiscale_complete_edges(lv, len(levels)))
            edge_index = torch.empty((2, num_edges), dtype=torch.int32, device=dev)
            ws = _workspace(lib.ag_multiscale_workspace_bytes(max_resolution, len(levels), 1), dev)
            _lib.check(
                lib.ag_multiscale_complete(max_resolution, lv, len(levels), _ptr(rank), _ptr(level_edges),
                                           _ptr(edge_index[0]), _ptr(edge_index[1]), _ptr(ws), ws.numel(), _stream(dev)),
                "ag_multiscale_complete",
            )
            return edge_index
        count = torch.zeros((), dtype=torch.int64, device=dev)
        ws = _workspace(lib.ag_multiscale_workspace_bytes(max_resolution, len(levels), x_hops), dev)
        _lib.check(
            lib.ag_multiscale_count(max_resolution, lv, len(levels), x_hops, _ptr(rank), _ptr(act), _ptr(level_edges),
                                    _ptr(count), _ptr(ws), ws.numel(), _stream(dev)),
            "ag_multiscale_count",
        )
        num_edges = int(count.item())
        edge_index = torch.empty((2, num_edges), dtype=torch.int32, device=dev)
        if num_edges:
            _lib.check(
                lib.ag_multiscale_fill(max_resolution, lv, len(levels), x_hops, num_edges, _ptr(edge_index[0]),
                                       _ptr(edge_index[1]), _ptr(ws), ws.numel(), _stream(dev)),
                "ag_multiscale_fill",
            )
    return edge_index


def edge_attrs(src_x: torch.Tensor, tgt_x: torch.Tensor, edge_index: torch.Tensor, want_len: bool = True,
               want_dir: bool = True, rotated: bool = True, len_range: bool = False, dir_f64: bool = False):
    """Raw EdgeLength [E] and EdgeDirection [E,2] float32 values of an edge list.  With ``len_range``
    a third result: the device-side range of the lengths (2 words, see ag_edge_attrs_range) that
    ``normalise_`` takes as ``value_range`` to skip its reduction pass.  ``dir_f64``: the directions as
    float64, before the rounding to float32 (for a normalisation applied in float64, ``normalise_f64``).
    IndexError for an edge that refers to a node outside its node set (see ``check_ids``)."""
    lib = _lib.load()
    dev = device_of(edge_index, src_x, tgt_x)
    src_x, tgt_x = coords(src_x, dev), coords(tgt_x, dev)
    e = as_device(edge_index, dev, torch.int32)
    if e.ndim != 2 or e.shape[0] != 2:
        raise ValueError(f"edge_index must have shape [2, E], got {tuple(e.shape)}")
    num_edges = e.shape[1]
    if num_edges and (src_x.shape[0] == 0 or tgt_x.shape[0] == 0):
        raise IndexError("index out of bounds: edges between node sets of "
                         f"{src_x.shape[0]} and {tgt_x.shape[0]} nodes")
    flags = (1 if rotated else 0) | (2 if (dir_f64 and want_dir) else 0)
    with _on(dev):
        length = torch.empty(num_edges, dtype=torch.float32, device=dev) if want_len else None
        direction = (torch.empty((num_edges, 2), dtype=torch.float64 if dir_f64 else torch.float32, device=dev)
                     if want_dir else None)
        rng = (torch.empty(lib.ag_len_range_words(), dtype=torch.int32, device=dev)
               if (len_range and want_len) else None)
        _lib.check(
            lib.ag_edge_attrs_range(_ptr(src_x), src_x.shape[0], _ptr(tgt_x), tgt_x.shape[0], _ptr(e[0]), _ptr(e[1]),
                                    num_edges, flags, _ptr(length), _ptr(direction), _ptr(rng), _ptr(_bad_word(dev)),
                                    _stream(dev)),
            "ag_edge_attrs_range",
        )
    _ids_checked(dev)
    return (length, direction, rng) if len_range else (length, direction)


def normalise_(values: torch.Tensor, norm: str | None, invert: bool = False,
               value_range: torch.Tensor | None = None) -> torch.Tensor:
    """In-place global normalisation of a float32 device tensor (all elements reduced together).
    ``value_range``: the range of exactly these values as reported by ``edge_attrs(len_range=True)``;
    the normalisations that follow from a range (unit-max, unit-range) then take one pass."""
    if norm not in NORMS:
        raise ValueError(
            f"Attribute normalisation \"{norm}\" is not valid. Options are: 'l1', 'l2', 'unit-max' or 'unit-std'."
        )
    if NORMS[norm] == 0 and not invert:
        return values
    lib = _lib.load()
    dev = values.device
    assert values.dtype == torch.float32 and values.is_contiguous()
    n = values.numel()
    if value_range is not None and NORMS[norm] in (0, 3, 4):
        with _on(dev):
            _lib.check(lib.ag_normalise_range(_ptr(values), n, NORMS[norm], int(invert), _ptr(value_range), _stream(dev)),
                       "ag_normalise_range")
        return values
    with _on(dev):
        ws = _workspace(lib.ag_normalise_workspace_bytes(n), dev)
        _lib.check(lib.ag_normalise(_ptr(values), n, NORMS[norm], int(invert), _ptr(ws), ws.numel(), _stream(dev)),
                   "ag_normalise")
    return values


def normalise_f64(values: torch.Tensor, norm: str | None, out_dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """Global normalisation of a float64 device tensor, evaluated in float64 and rounded to float32 once
    (``out_dtype`` float64: not rounded at all)."""
    if norm not in NORMS:
        raise ValueError(
            f"Attribute normalisation \"{norm}\" is not valid. Options are: 'l1', 'l2', 'unit-max' or 'unit-std'."
        )
    lib = _lib.load()
    dev = values.device
    assert values.dtype == torch.float64 and values.is_contiguous()
    n = values.numel()
    assert out_dtype in (torch.float32, torch.float64)
    with _on(dev):
        out = torch.empty(values.shape, dtype=out_dtype, device=dev)
        ws = _workspace(lib.ag_normalise_workspace_bytes(n), dev)
        _lib.check(lib.ag_normalise_f64(_ptr(values), n, NORMS[norm], _ptr(out), int(out_dtype == torch.float64),
                                        _ptr(ws), ws.numel(), _stream(dev)),
                   "ag_normalise_f64")
    return out


def radius_rows(src: torch.Tensor, tgt: torch.Tensor, radius: float):
    """CSR form of the radius query: (offsets int64 [Nt+1], source ids int32 [E]), rows sorted by source id."""
    lib = _lib.load()
    dev = device_of(src, tgt)
    src, tgt = coords(src, dev), coords(tgt, dev)
    ns, nt = src.shape[0], tgt.shape[0]
    with _on(dev):
        offsets = torch.empty(nt + 2, dtype=torch.int64, device=dev)
        ws = _workspace(lib.ag_radius_workspace_bytes(ns, nt), dev)
        _lib.check(
            lib.ag_radius_count(_ptr(src), ns, _ptr(tgt), nt, float(radius), _ptr(offsets), _ptr(ws), ws.numel(),
                                _stream(dev)),
            "ag_radius_count",
        )
        num_edges, max_row = offsets[nt:].tolist()
        ids = torch.empty(max(num_edges, 1), dtype=torch.int32, device=dev)
        rows = torch.empty(max(num_edges, 1), dtype=torch.int32, device=dev)
        if num_edges:
            _lib.check(
                lib.ag_radius_fill(_ptr(src), ns, _ptr(tgt), nt, float(radius), _ptr(offsets), num_edges, max_row,
                                   _ptr(ids), _ptr(rows), _ptr(ws), ws.numel(), _stream(dev)),
                "ag_radius_fill",
            )
    return offsets[: nt + 1], ids[:num_edges]


# Candidates of the first pass: each node's nearest nodes, itself included.  Ten close the cells of
# quasi-uniform grids (eight surrounding nodes and one beyond them) and keep the k-NN query on its fast path.
VORONOI_NEIGHBOURS = 10


def voronoi_areas(x: torch.Tensor, neighbours: int = VORONOI_NEIGHBOURS, return_stats: bool = False):
    """float64 [N] areas of the spherical Voronoi cells of the nodes (unit sphere).

    First pass: every cell is clipped by the node's nearest ``neighbours`` nodes.  Cells that are not
    provably final after that (elongated cells of anisotropic grids) are finished from the rows of
    radius queries around them, the radius growing until every cell's security radius fits."""
    lib = _lib.load()
    dev = device_of(x)
    x = coords(x, dev)
    n = x.shape[0]
    if n < 4:
        # what scipy raises in the reference for a degenerate point set (scipy/spatial/_spherical_voronoi.py)
        raise ValueError("Rank of input points must be at least 3")
    k = min(int(neighbours), n, 64)
    with _on(dev):
        nbr = knn(x, x, k)
        check_status(dev)  # (the cells follow these ids: a search that failed must not get that far; this path
        #                     synchronises with the device after every pass anyway)
        area = torch.empty(n, dtype=torch.float64, device=dev)
        reach = torch.empty(n, dtype=torch.float64, device=dev)
        _lib.check(lib.ag_voronoi_areas_knn(_ptr(x), n, _ptr(nbr), k, _ptr(area), _ptr(reach), _stream(dev)),
                   "ag_voronoi_areas_knn")
        del nbr
        stats = {"first_pass_open": 0, "row_passes": 0}
        passes = 0
        while True:
            if bool((reach == -1.0).any()):
                raise ValueError("Duplicate generators present.")  # scipy's message for coincident points
            if bool((reach == -2.0).any()):
                raise RuntimeError("voronoi_areas: a cell has more vertices than the kernel holds")
            todo = torch.nonzero(reach > 0.0).squeeze(1)
            if passes == 0:
                stats["first_pass_open"] = int(todo.numel())
            if todo.numel() == 0:
                break
            need = reach[todo]
            # an unbounded cell has no radius to ask for yet: it starts from a few mean spacings and doubles
            floor = 4.0 * math.sqrt(4.0 * math.pi / n) * (2.0 ** passes)
            need = torch.where(torch.isfinite(need), need, torch.full_like(need, floor)).clamp(max=math.pi)
            # one radius query per octave of requested radii, so that a few wide cells do not make
            # every open cell search wide
            base = float(need.min())
            octave = torch.ceil(torch.log2(need / base)).clamp(min=0).to(torch.int64)
            for o in torch.unique(octave).tolist():
                sel = todo[octave == o]
                radius = min(base * (2.0 ** o) * (1.0 + 1.0e-12), math.pi)
                ids = sel.to(torch.int32).contiguous()
                offsets, rows = radius_rows(x, x[sel].contiguous(), radius)
                check_status(dev)
                _lib.check(
                    lib.ag_voronoi_areas_rows(_ptr(x), n, _ptr(ids), ids.numel(), _ptr(offsets), _ptr(rows), radius,
                                              _ptr(area), _ptr(reach), _stream(dev)),
                    "ag_voronoi_areas_rows",
                )
                if radius >= math.pi:  # every node was offered: the cell is final ...
                    left = reach[sel]
                    far = sel[torch.isinf(left)]
                    reach[sel] = torch.where(left > 0.0, torch.zeros_like(left), left)
                    if far.numel():  # ... unless it does not fit the chart of its node: charts centred inside the cell
                        far_ids = far.to(torch.int32).contiguous()
                        ws = _workspace(lib.ag_voronoi_far_workspace_bytes(n), dev)
                        _lib.check(
                            lib.ag_voronoi_areas_far(_ptr(x), n, _ptr(far_ids), far_ids.numel(), _ptr(area), _ptr(reach),
                                                     _ptr(ws), ws.numel(), _stream(dev)),
                            "ag_voronoi_areas_far",
                        )
                        stats["far_cells"] = stats.get("far_cells", 0) + int(far.numel())
                        if bool(torch.isinf(reach[far]).any()):
                            raise ValueError(
                                "voronoi_areas: a Voronoi cell spans a hemisphere (fewer than three effective "
                                "neighbours); such cells are not supported"
                            )
            passes += 1
            stats["row_passes"] = passes
    return (area, stats) if return_stats else area


def rdist_to_dist(r: float) -> float:
    return 2.0 * math.asin(math.sqrt(r))
