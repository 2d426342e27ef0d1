"""Text summary of a finished graph with the reference's interface
(/root/reference/src/anemoi/graphs/describe.py:21-225): node table, edge table with the numbers of
isolated source / target nodes, attribute statistics.  Works on a graph object or on a file written
by ``GraphCreator.save``; the isolated-node counts run on the GPU when the edge lists live there
(``ag_mark_ids``), otherwise in plain torch on the host -- a summary is not part of the build path.
"""
from __future__ import annotations

import math
from pathlib import Path

import torch

from .hetero import HeteroData


def _is_tensor(v) -> bool:
    return isinstance(v, torch.Tensor)


def _table(rows: list[list], header: list[str]) -> str:
    cells = [[str(c) for c in header]] + [[f"{c:.6g}" if isinstance(c, float) else str(c) for c in r] for r in rows]
    widths = [max(len(r[i]) for r in cells) for i in range(len(header))]
    lines = ["   ".join(c.ljust(w) for c, w in zip(r, widths)) for r in cells]
    lines.insert(1, "   ".join("-" * w for w in widths))
    return "\n".join(lines)


class GraphDescriptor:
    """``GraphDescriptor(path_or_graph).describe()`` (describe.py:24-26, 156-225)."""

    def __init__(self, path: str | Path | HeteroData, trusted: bool = False, **kwargs):
        if isinstance(path, (str, Path)):
            self.path = path
            from .hetero import load_graph

            # (a pickled graph file is only read when the caller vouches for it: see hetero.load_graph)
            self.graph = load_graph(path, trusted=trusted)
        else:
            self.path = None
            self.graph = path

    def _stores(self):
        return [s for _, s in self.graph.node_items()] + [s for _, s in self.graph.edge_items()]

    @property
    def total_size(self) -> int:
        """Bytes of all tensors of the graph (describe.py:28-38)."""
        return sum(v.numel() * v.element_size() for store in self._stores() for v in store.values() if _is_tensor(v))

    @staticmethod
    def _attrs(store, skip: str) -> list[str]:
        return [k for k, v in store.items() if _is_tensor(v) and k != skip and not k.startswith("_")]

    def get_node_summary(self) -> list[list]:
        """Name, node count, attribute names, attribute dimension, latitude / longitude range in degrees
        (describe.py:40-73)."""
        rows = []
        for name, nodes in self.graph.node_items():
            attrs = self._attrs(nodes, "x")
            x = nodes["x"]
            rows.append([
                name, int(x.shape[0]), ", ".join(attrs),
                sum(int(nodes[a].shape[1]) for a in attrs if nodes[a].ndim > 1),
                x[:, 0].min().item() / 2 / math.pi * 360, x[:, 0].max().item() / 2 / math.pi * 360,
                x[:, 1].min().item() / 2 / math.pi * 360, x[:, 1].max().item() / 2 / math.pi * 360,
            ])
        return rows

    @staticmethod
    def _touched(ids: torch.Tensor, size: int) -> int:
        """Number of distinct node ids in a row of an edge_index."""
        if ids.is_cuda:
            from . import ops

            mask = torch.zeros(size, dtype=torch.uint8, device=ids.device)
            return int(ops.mark_ids_(mask, ids).sum().item())
        return int(torch.unique(ids).numel())

    def get_edge_summary(self) -> list[list]:
        """Source, target, edge count, isolated source / target nodes, attribute dimension, attribute
        names (describe.py:75-106)."""
        rows = []
        for (src, _, dst), edges in self.graph.edge_items():
            attrs = self._attrs(edges, "edge_index")
            e = edges["edge_index"]
            n_src, n_dst = int(self.graph[src]["x"].shape[0]), int(self.graph[dst]["x"].shape[0])
            rows.append([
                src, dst, int(e.shape[1]), n_src - self._touched(e[0], n_src), n_dst - self._touched(e[1], n_dst),
                sum(int(edges[a].shape[1]) for a in attrs),
                ", ".join(f"{a}({int(edges[a].shape[1])}D)" for a in attrs),
            ])
        return rows

    def get_attribute_table(self) -> list[list]:
        """Type, owner, name, dtype, min, mean, max, standard deviation of every attribute
        (describe.py:108-154)."""
        rows = []
        for name, nodes in self.graph.node_items():
            for a in self._attrs(nodes, "x"):
                v = nodes[a].float()
                rows.append(["Node", name, a, nodes[a].dtype, v.min().item(), v.mean().item(), v.max().item(),
                             v.std().item()])
        for (src, _, dst), edges in self.graph.edge_items():
            for a in self._attrs(edges, "edge_index"):
                v = edges[a].float()
                rows.append(["Edge", f"{src}-->{dst}", a, edges[a].dtype, v.min().item(), v.mean().item(),
                             v.max().item(), v.std().item()])
        return rows

    def describe(self, show_attribute_distributions: bool | None = True) -> None:
        print()
        print(f"Path : {self.path}")
        print(f"Size : {self.total_size} bytes")
        print()
        print("Nodes summary")
        print(_table(self.get_node_summary(), ["Nodes name", "Num. nodes", "Attributes", "Attribute dim",
                                               "Min. latitude", "Max. latitude", "Min. longitude", "Max. longitude"]))
        print()
        print("Edges summary")
        print(_table(self.get_edge_summary(), ["Source", "Target", "Num. edges", "Isolated Source", "Isolated Target",
                                               "Attribute dim", "Attributes"]))
        print()
        if show_attribute_distributions:
            print("Attribute distributions")
            print(_table(self.get_attribute_table(), ["Type", "Source", "Name", "Dtype", "Min.", "Mean", "Max.",
                                                      "Std. dev."]))
            print()
        print("Graph ready.")
