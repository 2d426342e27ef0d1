"""The operations interface of ``anemoi_graphs_amd.ops`` implemented with the CPU oracle, so that the
host-side logic (sharding, gathering, re-indexing) can be tested without a GPU.  Test infrastructure."""
from __future__ import annotations

import numpy as np
import torch

from oracle import icosphere as ico
from oracle import reference_path as rp


def _np(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def icosphere(resolution, device=None):
    v, f = ico.icosphere(resolution)
    coords = ico.latlon_rad(v)
    order = rp.coordinates_ordering_canonical(coords)
    rank = np.empty_like(order)
    rank[order] = np.arange(len(order))
    return (torch.from_numpy(v), torch.from_numpy(f.astype(np.int32)), torch.from_numpy(coords[order].copy()),
            torch.from_numpy(order.astype(np.int32)), torch.from_numpy(rank.astype(np.int32)), None)


def nn_maxdist(x):
    return torch.tensor(rp.grid_reference_distance(_np(x)), dtype=torch.float64)


def radius_edges(src, tgt, radius):
    e = rp.radius_edges(_np(src), _np(tgt), radius)
    return torch.from_numpy(np.ascontiguousarray(rp.canonical_sort(e)))


def knn(src, tgt, k):
    e = rp.knn_edges(_np(src), _np(tgt), k)
    return torch.from_numpy(e[0].reshape(len(_np(tgt)), k).copy())


def row_ids(rows, k, device=None, out=None):
    ids = torch.arange(rows, dtype=torch.int32).repeat_interleave(k)
    if out is None:
        return ids
    out.copy_(ids)
    return out


def multiscale_edges(max_resolution, levels, x_hops, rank, level_edges=None, active=None):
    rank = _np(rank)
    order = np.empty_like(rank)
    order[rank] = np.arange(len(rank))
    e = rp.multiscale_edges(list(levels), x_hops, order)
    return torch.from_numpy(np.ascontiguousarray(rp.canonical_sort(e)))


def edge_attrs(src_x, tgt_x, edge_index, want_len=True, want_dir=True, rotated=True):
    s, t, e = _np(src_x), _np(tgt_x), _np(edge_index)
    length = torch.from_numpy(rp.edge_lengths(s, t, e).astype(np.float32)) if want_len else None
    direction = torch.from_numpy(rp.edge_directions(s, t, e, rotated).astype(np.float32)) if want_dir else None
    return length, direction


def normalise_(values, norm, invert=False):
    out = rp.post_process(_np(values), norm, invert)
    return torch.from_numpy(np.ascontiguousarray(out))


def normalise_band_(values, norm, comm=None):
    """Normalisation of one rank's share of an attribute by the range over ALL ranks (the CPU counterpart of the
    range words the HIP attribute kernel reports): numpy's float32 arithmetic, normalise.py:45-48."""
    v = _np(values).astype(np.float32)
    if norm is None:
        return torch.from_numpy(v)
    lo = torch.tensor([float(v.min()) if v.size else np.inf], dtype=torch.float32)
    hi = torch.tensor([float(v.max()) if v.size else -np.inf], dtype=torch.float32)
    if comm is not None and comm.world > 1:
        comm.dist.all_reduce(lo, op=comm.dist.ReduceOp.MIN)
        comm.dist.all_reduce(hi, op=comm.dist.ReduceOp.MAX)
    lo, hi = np.float32(lo.item()), np.float32(hi.item())
    if norm == "unit-max":
        return torch.from_numpy(v / hi)
    if norm == "unit-range":
        return torch.from_numpy((v - lo) / (hi - lo))
    raise ValueError(norm)
