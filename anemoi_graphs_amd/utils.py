"""Counterparts of ``anemoi.graphs.utils`` (/root/reference/src/anemoi/graphs/utils.py) on the GPU."""
from __future__ import annotations

import torch

from . import ops, settings


_PINNED_FROM = 1 << 20  # bytes: smaller results are not worth a page-locked buffer


def to_host(t: torch.Tensor) -> torch.Tensor:
    """Host copy of a device tensor.  Large results go through page-locked memory (torch keeps such buffers for
    reuse): the copy then runs at the speed of the link instead of a quarter of it -- a graph of 31 million edges is
    630 MB of edge lists and attributes, 41 GB/s once the buffers exist.  (Cutting a large result into pieces that
    travel on streams of their own was measured too: 16.7 ms per O1280 build with one piece, 16.7 with four, 17.0 with
    eight -- one copy command already fills the link.)  The tensor returned is an ordinary host tensor (in page-locked
    memory), complete when this function returns."""
    if not t.is_cuda:
        return t
    if t.numel() * t.element_size() < _PINNED_FROM:
        out = t.cpu()
    else:
        out = torch.empty(t.shape, dtype=t.dtype, device="cpu", pin_memory=True)
        out.copy_(t, non_blocking=True)
        torch.cuda.current_stream(t.device).synchronize()
    # the device tensor stays on as the mirror of its host copy: an edge list that goes to the host here is read on
    # the device again by the attributes of its edge set
    if t.is_contiguous():
        ops.mirror_put(out, t)
    return out


def place(t: torch.Tensor, like: torch.Tensor | None = None) -> torch.Tensor:
    """Put a result tensor where graph tensors live (host memory unless ``settings.device_resident``)."""
    if settings.device_resident:
        return t
    return to_host(t)


def get_grid_reference_distance(coords_rad: torch.Tensor, mask: torch.Tensor | None = None,
                                density: float | None = None) -> float:
    """Maximum over the nodes of the distance to the nearest other node (utils.py:44-63).

    As in the reference the mask is only checked for shape, never applied (utils.py:32-39).  ``density``: of the
    node set when the caller knows it (``ops.point_density``), else it is measured."""
    assert mask is None or mask.shape == (coords_rad.shape[0], 1), (
        "Mask must have the same shape as the number of nodes."
    )
    value = float(ops.nn_maxdist(coords_rad, density=density).item())
    if value == 0.0:
        # every node coincides with another one: the reference takes dists[dists > 0].max() of an empty
        # selection, which numpy refuses with exactly this error (utils.py:61-63)
        raise ValueError("zero-size array to reduction operation maximum which has no identity")
    return value


def concat_edges(edge_indices1: torch.Tensor, edge_indices2: torch.Tensor) -> torch.Tensor:
    """Union of two edge lists, columns unique and sorted by (source, target) (utils.py:66-81)."""
    dev = ops.device_of(edge_indices1, edge_indices2)
    cat = torch.cat([edge_indices1.to(dev), edge_indices2.to(dev)], dim=1)
    out = ops.unique_edges(cat.to(torch.int32))
    out = out.to(edge_indices1.dtype)
    return out if edge_indices1.is_cuda else out.cpu()


def haversine_distance(source_coords: torch.Tensor, target_coords: torch.Tensor) -> torch.Tensor:
    """Row-wise great-circle angle in radians, float32 (utils.py:84-103)."""
    dev = ops.device_of(source_coords, target_coords)
    n = source_coords.shape[0]
    idx = torch.arange(n, dtype=torch.int32, device=dev)
    length, _ = ops.edge_attrs(source_coords, target_coords, torch.stack([idx, idx]), want_dir=False)
    return length if torch.as_tensor(source_coords).is_cuda else length.cpu()
