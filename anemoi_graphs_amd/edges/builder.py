"""Edge builders with the reference's plugin interface
(/root/reference/src/anemoi/graphs/edges/builder.py:42-462), running on the HIP kernels."""
from __future__ import annotations

import logging
import math
from abc import ABC, abstractmethod

import torch

from .. import ops
from ..hetero import HeteroData, NodeStorage
from ..instantiate import instantiate
from ..utils import concat_edges, place

LOGGER = logging.getLogger(__name__)
EARTH_RADIUS = 6371.0  # km (/root/reference/src/anemoi/graphs/__init__.py:10)


def kept_binning(nodes: NodeStorage, cell: float | None, accept: tuple[float, float] = (0.8, 2.0)):
    """The binning of a node set that the edge builders of a graph share, kept on the node storage as the private
    attribute ``_binning`` (removed with the other underscore attributes when the graph is finished).

    The reference fits one ball tree per builder (edges/builder.py:259-260,364-365); here a node set that serves
    several searches -- the data nodes are the sources of the cut-off encoder and the targets of the k-NN decoder
    -- is binned once.  ``cell`` (radians): the cell size the caller wants; a kept binning is reused when it is of
    these coordinates and its cells are within ``accept`` times that size, else a new one replaces it.  ``cell``
    None: whatever is kept, or None."""
    x = torch.as_tensor(nodes.x)
    kept = nodes["_binning"] if "_binning" in nodes else None
    if isinstance(kept, ops.Binning) and kept.key.matches(x):
        if cell is None or accept[0] * cell <= kept.cell <= accept[1] * cell:
            return kept
    if cell is None:
        return None
    kept = ops.bin_points(x, cell)
    nodes["_binning"] = kept
    return kept


class BaseEdgeBuilder(ABC):
    """Base class for edge builders (edges/builder.py:42-156)."""

    def __init__(
        self,
        source_name: str,
        target_name: str,
        source_mask_attr_name: str | None = None,
        target_mask_attr_name: str | None = None,
    ):
        self.source_name = source_name
        self.target_name = target_name
        self.source_mask_attr_name = source_mask_attr_name
        self.target_mask_attr_name = target_mask_attr_name

    @property
    def name(self) -> tuple[str, str, str]:
        return self.source_name, "to", self.target_name

    @abstractmethod
    def get_adjacency_matrix(self, source_nodes: NodeStorage, target_nodes: NodeStorage) -> torch.Tensor:
        """The edges as a device int32 tensor [2, E]: row 0 source ids, row 1 target ids
        (the reference returns a scipy COO matrix of shape (n_target, n_source) whose (col, row)
        are exactly these two rows: edges/builder.py:84-87)."""

    def prepare_node_data(self, graph: HeteroData) -> tuple[NodeStorage, NodeStorage]:
        return graph[self.source_name], graph[self.target_name]

    def get_edge_index(self, graph: HeteroData) -> torch.Tensor:
        source_nodes, target_nodes = self.prepare_node_data(graph)
        edge_index = self.get_adjacency_matrix(source_nodes, target_nodes)
        return place(edge_index.to(torch.int32))

    def register_edges(self, graph: HeteroData) -> HeteroData:
        edge_index = self.get_edge_index(graph)
        edge_type = type(self).__name__
        if "edge_index" in graph[self.name]:
            graph[self.name].edge_index = concat_edges(graph[self.name].edge_index, edge_index)
            if edge_type not in graph[self.name].edge_type:
                graph[self.name].edge_type = graph[self.name].edge_type + "," + edge_type
            return graph
        graph[self.name].edge_index = edge_index
        graph[self.name].edge_type = edge_type
        return graph

    def register_attributes(self, graph: HeteroData, config: dict) -> HeteroData:
        from .attributes import compute_fused

        attributes = {attr_name: instantiate(attr_config) for attr_name, attr_config in config.items()}
        for attr_name, values in compute_fused(graph, self.name, attributes).items():
            graph[self.name][attr_name] = values
        return graph

    def update_graph(self, graph: HeteroData, attrs_config: dict | None = None) -> HeteroData:
        graph = self.register_edges(graph)
        if attrs_config is not None:
            graph = self.register_attributes(graph, attrs_config)
        return graph


def node_density(nodes: NodeStorage) -> float:
    """Density of a node set (``ops.point_density``): the value its builder left with it -- a mesh knows the share
    of the sphere it covers -- else measured once and kept with the set (a private attribute)."""
    x = torch.as_tensor(nodes.x)
    kept = nodes["_density"] if "_density" in nodes else None
    if isinstance(kept, tuple) and kept[0].matches(x):
        return kept[1]
    value = ops.point_density(x)
    nodes["_density"] = (ops.tensor_key(x), value)
    return value


class NodeMaskingMixin:
    """Masking of source/target nodes (edges/builder.py:159-193): the search runs on the compacted
    node sets and the ids are mapped back through the compaction maps on the device."""

    def get_node_coordinates(self, source_nodes: NodeStorage, target_nodes: NodeStorage):
        dev = ops.device_of(source_nodes.x, target_nodes.x)
        source_coords, target_coords = ops.coords(source_nodes.x, dev), ops.coords(target_nodes.x, dev)
        self._source_ids = self._target_ids = None
        if self.source_mask_attr_name is not None:
            self._source_ids = ops.mask_ids(source_nodes[self.source_mask_attr_name], dev)
            source_coords = ops.gather_coords(source_coords, self._source_ids)
        if self.target_mask_attr_name is not None:
            self._target_ids = ops.mask_ids(target_nodes[self.target_mask_attr_name], dev)
            target_coords = ops.gather_coords(target_coords, self._target_ids)
        return source_coords, target_coords

    def undo_masking(self, edge_index: torch.Tensor, source_nodes: NodeStorage, target_nodes: NodeStorage):
        if self._target_ids is not None:
            ops.remap_ids_(edge_index[1], self._target_ids)
        if self._source_ids is not None:
            ops.remap_ids_(edge_index[0], self._source_ids)
        return edge_index


class KNNEdges(BaseEdgeBuilder, NodeMaskingMixin):
    """k nearest source nodes of every target node (edges/builder.py:196-270)."""

    def __init__(
        self,
        source_name: str,
        target_name: str,
        num_nearest_neighbours: int,
        source_mask_attr_name: str | None = None,
        target_mask_attr_name: str | None = None,
    ) -> None:
        super().__init__(source_name, target_name, source_mask_attr_name, target_mask_attr_name)
        assert isinstance(num_nearest_neighbours, int), "Number of nearest neighbours must be an integer"
        assert num_nearest_neighbours > 0, "Number of nearest neighbours must be positive"
        self.num_nearest_neighbours = num_nearest_neighbours

    def get_adjacency_matrix(self, source_nodes: NodeStorage, target_nodes: NodeStorage) -> torch.Tensor:
        assert self.num_nearest_neighbours is not None, "number of neighbors required for knn encoder"
        LOGGER.info(
            "Using KNN-Edges (with %d nearest neighbours) between %s and %s.",
            self.num_nearest_neighbours, self.source_name, self.target_name,
        )
        k = self.num_nearest_neighbours
        if self.source_mask_attr_name is None and self.target_mask_attr_name is None:
            # targets that another builder of this graph already binned (the data nodes, as the sources of a
            # cut-off encoder) are taken in that order, if its cells are not coarser than this search's own
            kept = kept_binning(target_nodes, None)
            n_src = int(torch.as_tensor(source_nodes.x).shape[0])
            src_density = node_density(source_nodes)
            own_cell = math.sqrt((2.0 * (k + 1) + 2.0) * 4.0 / (max(n_src, 1) * src_density))
            if kept is not None and kept.cell <= 1.5 * own_cell:
                return ops.knn_edges(ops.coords(source_nodes.x, kept.x.device), None, k, tgt_binning=kept,
                                     src_density=src_density)
        source_coords, target_coords = self.get_node_coordinates(source_nodes, target_nodes)
        # (masked node sets are measured: the mask may pick any part of the set)
        edge_index = ops.knn_edges(  # per target: ascending distance
            source_coords, target_coords, k,
            src_density=node_density(source_nodes) if self.source_mask_attr_name is None else None,
            tgt_density=node_density(target_nodes) if self.target_mask_attr_name is None else None)
        return self.undo_masking(edge_index, source_nodes, target_nodes)


class CutOffEdges(BaseEdgeBuilder, NodeMaskingMixin):
    """All source nodes within ``cutoff_factor`` x the target grid's reference distance of every
    target node (edges/builder.py:273-371)."""

    def __init__(
        self,
        source_name: str,
        target_name: str,
        cutoff_factor: float,
        source_mask_attr_name: str | None = None,
        target_mask_attr_name: str | None = None,
    ):
        super().__init__(source_name, target_name, source_mask_attr_name, target_mask_attr_name)
        assert isinstance(cutoff_factor, (int, float)), "Cutoff factor must be a float"
        assert cutoff_factor > 0, "Cutoff factor must be positive"
        self.cutoff_factor = cutoff_factor

    def get_cutoff_radius(self, graph: HeteroData, mask_attr: torch.Tensor | None = None) -> float:
        from ..utils import get_grid_reference_distance

        target_nodes = graph[self.target_name]
        mask = target_nodes[mask_attr] if mask_attr is not None else None
        # the value stays with the node set (a private attribute, removed when the graph is finished) for the other
        # cut-off builders of the graph that have these targets
        x = torch.as_tensor(target_nodes.x)
        kept = target_nodes["_reference_distance"] if "_reference_distance" in target_nodes else None
        if mask is None and isinstance(kept, tuple) and kept[0].matches(x):
            return kept[1] * self.cutoff_factor
        mesh = target_nodes["_nx_graph"] if "_nx_graph" in target_nodes else None
        node_type = target_nodes["node_type"] if "node_type" in target_nodes else None
        if (node_type == "TriNodes" and mesh is not None and getattr(mesh, "active", True) is None
                and int(target_nodes.x.shape[0]) == 10 * 4 ** mesh.resolution + 2 and mask is None):
            # the complete icosphere: the nearest neighbour of a vertex is a mesh neighbour, no search is needed
            target_grid_reference_distance = float(
                ops.mesh_maxdist(mesh.resolution, mesh.level_edges, target_nodes.x, mesh.rank).item())
        else:
            target_grid_reference_distance = get_grid_reference_distance(target_nodes.x, mask,
                                                                         density=node_density(target_nodes))
        if mask is None:
            target_nodes["_reference_distance"] = (ops.tensor_key(x), target_grid_reference_distance)
        return target_grid_reference_distance * self.cutoff_factor

    def prepare_node_data(self, graph: HeteroData) -> tuple[NodeStorage, NodeStorage]:
        self.radius = self.get_cutoff_radius(graph)
        return super().prepare_node_data(graph)

    def get_adjacency_matrix(self, source_nodes: NodeStorage, target_nodes: NodeStorage) -> torch.Tensor:
        LOGGER.info(
            "Using CutOff-Edges (with radius = %.1f km) between %s and %s.",
            self.radius * EARTH_RADIUS, self.source_name, self.target_name,
        )
        if self.source_mask_attr_name is None and self.target_mask_attr_name is None:
            # the sources are binned into a buffer that stays with the node set, for the builders that follow
            n_src = int(torch.as_tensor(source_nodes.x).shape[0])
            if n_src > 0:
                kept = kept_binning(source_nodes, ops.radius_cell(n_src, self.radius))
                return ops.radius_edges(None, ops.coords(target_nodes.x, kept.x.device), self.radius, src_binning=kept)
        source_coords, target_coords = self.get_node_coordinates(source_nodes, target_nodes)
        edge_index = ops.radius_edges(source_coords, target_coords, self.radius)
        return self.undo_masking(edge_index, source_nodes, target_nodes)


class MultiScaleEdges(BaseEdgeBuilder):
    """Multi-scale edges between the nodes of a refined icosahedron (edges/builder.py:374-462)."""

    VALID_NODES = ["TriNodes", "LimitedAreaTriNodes", "StretchedTriNodes"]

    def __init__(self, source_name: str, target_name: str, x_hops: int, **kwargs):
        super().__init__(source_name, target_name)
        assert source_name == target_name, f"{self.__class__.__name__} requires source and target nodes to be the same."
        assert isinstance(x_hops, int), "Number of x_hops must be an integer"
        assert x_hops > 0, "Number of x_hops must be positive"
        self.x_hops = x_hops

    def get_adjacency_matrix(self, source_nodes: NodeStorage, target_nodes: NodeStorage) -> torch.Tensor:
        if source_nodes.node_type not in self.VALID_NODES:
            raise ValueError(f"Invalid node type {source_nodes.node_type}")
        mesh = source_nodes["_nx_graph"]
        resolutions = list(source_nodes["_resolutions"])
        # LimitedAreaTriNodes / StretchedTriNodes: only mesh vertices that are nodes take part, at every
        # level (generate/tri_icosahedron.py:176-183 with the builder's area mask, respectively the
        # 1 km "is a node" mask of edges/builder.py:422-432)
        return ops.multiscale_edges(mesh.resolution, resolutions, self.x_hops, mesh.rank, mesh.level_edges,
                                    getattr(mesh, "active", None))

    def update_graph(self, graph: HeteroData, attrs_config: dict | None = None) -> HeteroData:
        node_type = graph[self.source_name].node_type
        assert node_type in self.VALID_NODES, f"{self.__class__.__name__} requires {','.join(self.VALID_NODES)} nodes."
        return super().update_graph(graph, attrs_config)
