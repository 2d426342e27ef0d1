"""Post-processors of a finished graph (/root/reference/src/anemoi/graphs/processors/post_process.py:22-149).

The masks and the edge-index remapping run on the GPU: marking the nodes that occur in an edge list,
compacting the mask into the list of kept ids, inverting it into an old->new map and applying the map to
every edge list are four small kernels instead of a Python dictionary lookup per edge."""
from __future__ import annotations

import logging
from abc import ABC, abstractmethod

import torch

from .. import ops
from ..hetero import HeteroData

LOGGER = logging.getLogger(__name__)


class PostProcessor(ABC):
    @abstractmethod
    def update_graph(self, graph: HeteroData) -> HeteroData:
        raise NotImplementedError(f"The {self.__class__.__name__} class does not implement the method update_graph().")


class BaseMaskingProcessor(PostProcessor, ABC):
    """Base class for mask based processors (post_process.py:29-95)."""

    def __init__(self, nodes_name: str, save_mask_indices_to_attr: str | None = None) -> None:
        self.nodes_name = nodes_name
        self.save_mask_indices_to_attr = save_mask_indices_to_attr
        self.mask: torch.Tensor = None

    def _kept_ids(self) -> torch.Tensor:
        dev = ops.device_of(self.mask)
        return ops.mask_ids(self.mask, dev)

    def removing_nodes(self, graph: HeteroData) -> HeteroData:
        """Remove nodes based on the mask (post_process.py:41-46)."""
        kept = self._kept_ids().long()
        for attr_name in graph[self.nodes_name].node_attrs():
            value = graph[self.nodes_name][attr_name]
            graph[self.nodes_name][attr_name] = value[kept.to(value.device)]
        return graph

    def update_edge_indices(self, graph: HeteroData) -> HeteroData:
        """Move the edge indices to the new positions of the nodes (post_process.py:51-61)."""
        kept = self._kept_ids()
        id_map = ops.invert_ids(kept, int(self.mask.numel()))
        for edges_name in graph.edge_types:
            edge_index = graph[edges_name].edge_index
            rows = [r for r, end in ((0, edges_name[0]), (1, edges_name[2])) if end == self.nodes_name]
            if not rows:
                continue
            work = edge_index.to(device=id_map.device, dtype=torch.int32).contiguous()
            for r in rows:
                ops.remap_ids_(work[r], id_map)
            graph[edges_name].edge_index = work.to(device=edge_index.device, dtype=edge_index.dtype)
        return graph

    @abstractmethod
    def compute_mask(self, graph: HeteroData) -> torch.Tensor: ...

    def add_attribute(self, graph: HeteroData) -> HeteroData:
        """Add the kept indices as a node attribute (post_process.py:66-75)."""
        if self.save_mask_indices_to_attr is not None:
            LOGGER.info(
                f"An attribute {self.save_mask_indices_to_attr} has been added with the indices to mask the nodes "
                "from the original graph."
            )
            x = graph[self.nodes_name].x
            mask_indices = self._kept_ids().to(device=x.device, dtype=torch.int64).reshape((x.shape[0], -1))
            graph[self.nodes_name][self.save_mask_indices_to_attr] = mask_indices
        return graph

    def update_graph(self, graph: HeteroData) -> HeteroData:
        self.mask = self.compute_mask(graph)
        LOGGER.info(f"Removing {(~self.mask.bool()).sum()} nodes from {self.nodes_name}.")
        graph = self.removing_nodes(graph)
        graph = self.update_edge_indices(graph)
        graph = self.add_attribute(graph)
        return graph


class RemoveUnconnectedNodes(BaseMaskingProcessor):
    """Remove the nodes that no edge touches (post_process.py:98-149)."""

    def __init__(self, nodes_name: str, save_mask_indices_to_attr: str | None = None, ignore: str | None = None) -> None:
        super().__init__(nodes_name, save_mask_indices_to_attr)
        self.ignore = ignore

    def compute_mask(self, graph: HeteroData) -> torch.Tensor:
        """uint8 device mask of the connected nodes (post_process.py:133-149)."""
        nodes = graph[self.nodes_name]
        dev = ops.device_of(nodes.x)
        num_nodes = nodes.x.shape[0]
        if self.ignore is not None:
            LOGGER.info(f"The nodes with {self.ignore}=True will not be removed.")
            connected = torch.as_tensor(nodes[self.ignore]).bool().reshape(-1).to(device=dev, dtype=torch.uint8).contiguous()
            assert connected.numel() == num_nodes
        else:
            connected = torch.zeros(num_nodes, dtype=torch.uint8, device=dev)
        for (source_name, _, target_name), edges in graph.edge_items():
            if source_name == self.nodes_name:
                ops.mark_ids_(connected, edges.edge_index[0])
            if target_name == self.nodes_name:
                ops.mark_ids_(connected, edges.edge_index[1])
        return connected
