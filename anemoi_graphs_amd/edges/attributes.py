"""Edge attributes with the reference's interface
(/root/reference/src/anemoi/graphs/edges/attributes.py:24-157, normalise.py:17-55)."""
from __future__ import annotations

from abc import ABC, abstractmethod

import torch

from .. import ops
from ..hetero import HeteroData
from ..utils import place


class NormaliserMixin:
    """Global normalisation of an attribute (normalise.py:20-55) on the device."""

    def normalise(self, values: torch.Tensor) -> torch.Tensor:
        return ops.normalise_(values, self.norm)


class BaseEdgeAttribute(ABC, NormaliserMixin):
    def __init__(self, norm: str | None = None) -> None:
        self.norm = norm

    @abstractmethod
    def get_raw_values(self, graph: HeteroData, source_name: str, target_name: str, *args, **kwargs) -> torch.Tensor: ...

    def post_process(self, values: torch.Tensor) -> torch.Tensor:
        if values.ndim == 1:
            values = values[:, None]
        if values.dtype == torch.float64:
            # normalised in float64 and rounded to float32 afterwards, as the reference treats a float64 array
            # (edges/attributes.py:33-40: normalise(values) then astype(float32))
            return place(ops.normalise_f64(values.contiguous(), self.norm))
        return place(self.normalise(values.contiguous()))

    def check_nodes(self, graph: HeteroData, edges_name: tuple[str, str, str]) -> None:
        source_name, _, target_name = edges_name
        assert (
            source_name in graph.node_types
        ), f"Node \"{source_name}\" not found in graph. Optional nodes are {', '.join(graph.node_types)}."
        assert (
            target_name in graph.node_types
        ), f"Node \"{target_name}\" not found in graph. Optional nodes are {', '.join(graph.node_types)}."

    def compute(self, graph: HeteroData, edges_name: tuple[str, str, str], *args, **kwargs) -> torch.Tensor:
        self.check_nodes(graph, edges_name)
        source_name, _, target_name = edges_name
        values = self.get_raw_values(graph, source_name, target_name, *args, **kwargs)
        return self.post_process(values)


def compute_fused(graph: HeteroData, edges_name: tuple[str, str, str], attributes: dict) -> dict:
    """Evaluate several edge attributes of one edge set.  An EdgeLength and a rotated-or-not
    EdgeDirection of the same edge set share one kernel launch (the endpoints are gathered once);
    the results are identical to calling ``compute`` on each attribute."""
    source_name, _, target_name = edges_name
    lengths = [a for a in attributes.values() if type(a) is EdgeLength]
    directions = [a for a in attributes.values() if type(a) is EdgeDirection]
    raw_len = raw_dir = len_range = None
    if lengths and directions and len({(d.luse_rotated_features, d.wants_float64) for d in directions}) == 1:
        for a in attributes.values():
            a.check_nodes(graph, edges_name)
        edge_index = graph[edges_name].edge_index
        # (the kernel also reports the range of the lengths: unit-max / unit-range need no reduction pass)
        raw_len, raw_dir, len_range = ops.edge_attrs(graph[source_name].x, graph[target_name].x, edge_index,
                                                     rotated=directions[0].luse_rotated_features, len_range=True,
                                                     dir_f64=directions[0].wants_float64)
    out = {}
    n_len, n_dir = len(lengths), len(directions)
    for name, a in attributes.items():
        if raw_len is not None and type(a) is EdgeLength:
            n_len -= 1  # normalisation is in place: all but the last user work on a copy of the raw values
            out[name] = a.post_process(raw_len.clone() if n_len else raw_len, value_range=len_range)
        elif raw_dir is not None and type(a) is EdgeDirection:
            n_dir -= 1
            out[name] = a.post_process(raw_dir.clone() if n_dir else raw_dir)
        else:
            out[name] = a.compute(graph, edges_name)
    return out


class EdgeDirection(BaseEdgeAttribute):
    """Unit (x, y) direction of every edge, by default after rotating the target node to the north
    pole (edges/attributes.py:56-104)."""

    def __init__(self, norm: str | None = None, luse_rotated_features: bool = True) -> None:
        super().__init__(norm)
        self.luse_rotated_features = luse_rotated_features

    @property
    def wants_float64(self) -> bool:
        """The reference's rotated directions are a float64 array that is normalised before it is rounded to
        float32 (edges/directional.py:57-65, edges/attributes.py:33-40).  For the normalisations that only
        scale, rounding first changes a result by float32 roundings; `unit-range` subtracts the minimum, which
        cancels, so it is applied to the float64 values here too."""
        return self.luse_rotated_features and self.norm == "unit-range"

    def get_raw_values(self, graph: HeteroData, source_name: str, target_name: str) -> torch.Tensor:
        edge_index = graph[(source_name, "to", target_name)].edge_index
        _, direction = ops.edge_attrs(graph[source_name].x, graph[target_name].x, edge_index, want_len=False,
                                      rotated=self.luse_rotated_features, dir_f64=self.wants_float64)
        return direction


class EdgeLength(BaseEdgeAttribute):
    """Great-circle length of every edge in radians (edges/attributes.py:107-157)."""

    def __init__(self, norm: str | None = None, invert: bool = False) -> None:
        super().__init__(norm)
        self.invert = invert

    def get_raw_values(self, graph: HeteroData, source_name: str, target_name: str) -> torch.Tensor:
        edge_index = graph[(source_name, "to", target_name)].edge_index
        length, _ = ops.edge_attrs(graph[source_name].x, graph[target_name].x, edge_index, want_dir=False)
        return length

    def post_process(self, values: torch.Tensor, value_range: torch.Tensor | None = None) -> torch.Tensor:
        """``value_range``: the range of ``values`` when the kernel that produced them reported it."""
        if values.ndim == 1:
            values = values[:, None]
        return place(ops.normalise_(values.contiguous(), self.norm, invert=self.invert, value_range=value_range))
