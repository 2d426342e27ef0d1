"""Node builders with the reference's plugin interface
(/root/reference/src/anemoi/graphs/nodes/builders/base.py:22-125,
 nodes/builders/from_refined_icosahedron.py:30-106, nodes/builders/from_vectors.py:22-67)."""
from __future__ import annotations

import logging
from abc import ABC, abstractmethod

import numpy as np
import torch

from .. import ops, settings
from ..generate.masks import KNNAreaMaskBuilder
from ..hetero import HeteroData
from ..instantiate import instantiate

LOGGER = logging.getLogger(__name__)


class BaseNodeBuilder(ABC):
    """Base class for node builders: coordinates go to ``graph[name].x`` in radians (base.py:22-33)."""

    hidden_attributes: set[str] = set()

    def __init__(self, name: str) -> None:
        self.name = name
        self.area_mask_builder = None

    def register_nodes(self, graph: HeteroData) -> HeteroData:
        graph[self.name].x = self.get_coordinates()
        graph[self.name].node_type = type(self).__name__
        return graph

    def register_attributes(self, graph: HeteroData, config: dict | None = None) -> HeteroData:
        for hidden_attr in self.hidden_attributes:
            graph[self.name][f"_{hidden_attr}"] = getattr(self, hidden_attr)
        for attr_name, attr_config in (config or {}).items():
            graph[self.name][attr_name] = instantiate(attr_config).compute(graph, self.name)
        return graph

    @abstractmethod
    def get_coordinates(self) -> torch.Tensor: ...

    def reshape_coords(self, latitudes: np.ndarray, longitudes: np.ndarray) -> torch.Tensor:
        """Degrees -> float32 radians [N, 2] (base.py:84-101)."""
        coords = np.stack([latitudes, longitudes], axis=-1).reshape((-1, 2))
        coords = np.deg2rad(coords)
        return torch.tensor(coords, dtype=torch.float32)

    def update_graph(self, graph: HeteroData, attrs_config: dict | None = None) -> HeteroData:
        graph = self.register_nodes(graph)
        if attrs_config is None:  # as the reference: hidden attributes only with a config (base.py:120-123)
            return graph
        return self.register_attributes(graph, attrs_config)


class LatLonNodes(BaseNodeBuilder):
    """Nodes from latitude/longitude vectors in degrees (nodes/builders/from_vectors.py:22-67)."""

    def __init__(self, latitudes, longitudes, name: str) -> None:
        super().__init__(name)
        self.latitudes = np.asarray(latitudes)
        self.longitudes = np.asarray(longitudes)
        assert len(self.latitudes) == len(self.longitudes), "Lenght of latitudes and longitudes must match."
        assert self.latitudes.ndim == 1 or self.latitudes.shape[1] == 1, "latitudes must have shape (N, ) or (N, 1)."
        assert self.longitudes.ndim == 1 or self.longitudes.shape[1] == 1, "longitudes must have shape (N, ) or (N, 1)."

    def get_coordinates(self) -> torch.Tensor:
        x = self.reshape_coords(self.latitudes, self.longitudes)
        return x.to(ops.device_of()) if settings.device_resident else x


class IcosahedralMesh:
    """What ``_nx_graph`` is in the reference (the node set of the mesh, kept for MultiScaleEdges):
    here the device-side description of the icosphere -- node<->vertex permutations -- instead of a
    networkx graph.  ``number_of_nodes`` and ``nodes`` keep the two accessors callers use."""

    def __init__(self, resolution: int, order: torch.Tensor, rank: torch.Tensor, level_edges: torch.Tensor,
                 active: torch.Tensor | None = None):
        self.resolution = resolution
        self.order = order  # node -> id of the node in the builder's unsorted coordinate array (int32, device)
        self.rank = rank    # mesh vertex -> node position, -1 for vertices that are not nodes (int32, device)
        self.level_edges = level_edges  # unique mesh edges of every refinement level (int64 keys, device)
        self.active = active  # uint8 per mesh vertex: 1 where the vertex is a node (None: all)

    def number_of_nodes(self) -> int:
        return int(self.order.numel())

    @property
    def nodes(self) -> list[int]:
        return self.order.cpu().tolist()


class IcosahedralNodes(BaseNodeBuilder, ABC):
    """Nodes of a refined icosahedron (from_refined_icosahedron.py:30-69)."""

    def __init__(self, resolution: int | list[int], name: str) -> None:
        if isinstance(resolution, int):
            self.resolutions = list(range(resolution + 1))
        else:
            self.resolutions = resolution
        super().__init__(name)
        self.hidden_attributes = BaseNodeBuilder.hidden_attributes | {
            "resolutions", "nx_graph", "node_ordering", "area_mask_builder",
        }

    # nodes per steradian where the mesh has nodes, relative to as many nodes spread over the sphere
    # (ops.point_density): a mesh knows what share of the sphere it covers, so nobody has to measure it
    density: float = 1.0

    def get_coordinates(self) -> torch.Tensor:
        self.nx_graph, coords_rad, self.node_ordering = self.create_nodes()
        return coords_rad

    def register_nodes(self, graph: HeteroData) -> HeteroData:
        graph = super().register_nodes(graph)
        graph[self.name]["_density"] = (ops.tensor_key(torch.as_tensor(graph[self.name].x)), float(self.density))
        return graph

    @abstractmethod
    def create_nodes(self): ...


class TriNodes(IcosahedralNodes):
    """Icosphere nodes (from_refined_icosahedron.py:99-106 -> generate/tri_icosahedron.py:24-58),
    built by the HIP subdivision kernels.  Node order: latitude descending, ties longitude
    descending (what generate/utils.py:30-32 produces when its two sorts are stable)."""

    def create_nodes(self):
        resolution = max(self.resolutions)
        _xyz, _faces, latlon, order, rank, level_edges = ops.icosphere(resolution)
        mesh = IcosahedralMesh(resolution, order, rank, level_edges)
        coords = latlon if settings.device_resident else latlon.cpu()
        return mesh, coords, order.cpu().tolist() if not settings.device_resident else order


def _finish(mesh, coords, order):
    if settings.device_resident:
        return mesh, coords, order
    return mesh, coords.cpu(), order.cpu().tolist()


class LimitedAreaIcosahedralNodes(IcosahedralNodes, ABC):
    """Icosahedral nodes restricted to an area of interest (from_refined_icosahedron.py:72-96)."""

    def __init__(self, resolution: int | list[int], reference_node_name: str, name: str,
                 mask_attr_name: str | None = None, margin_radius_km: float = 100.0) -> None:
        super().__init__(resolution, name)
        self.area_mask_builder = KNNAreaMaskBuilder(reference_node_name, margin_radius_km, mask_attr_name)

    def register_nodes(self, graph: HeteroData) -> HeteroData:
        self.area_mask_builder.fit(graph)
        return super().register_nodes(graph)


class LimitedAreaTriNodes(LimitedAreaIcosahedralNodes):
    """Icosphere nodes within ``margin_radius_km`` of the reference nodes
    (from_refined_icosahedron.py:119-131 -> generate/tri_icosahedron.py:47-58): the order is the
    global TriNodes order restricted to the masked vertices."""

    def create_nodes(self):
        resolution = max(self.resolutions)
        # (the order of the whole sphere's vertices is not needed: the vertices kept are ordered among themselves,
        # with the same comparison -- latitude, longitude, vertex id, all descending)
        _xyz, _faces, vertices, _order, _rank, level_edges = ops.icosphere(resolution, ordered=False)
        kept = torch.nonzero(self.area_mask_builder.get_mask(vertices, density=1.0)).squeeze(1)  # vertex ids, ascending
        unordered = vertices[kept].contiguous()
        among_kept = ops.node_order(unordered).long()
        coords = unordered[among_kept].contiguous()
        node_ordering = kept[among_kept].to(torch.int32).contiguous()
        self.density = vertices.shape[0] / max(int(coords.shape[0]), 1)  # (the vertices of the sphere over those kept)
        rank = torch.full((vertices.shape[0],), -1, dtype=torch.int32, device=vertices.device)
        rank[node_ordering.long()] = torch.arange(node_ordering.numel(), dtype=torch.int32, device=vertices.device)
        mesh = IcosahedralMesh(resolution, node_ordering, rank, level_edges, (rank >= 0).to(torch.uint8))
        return _finish(mesh, coords, node_ordering)


class StretchedIcosahedronNodes(IcosahedralNodes, ABC):
    """Icosahedral nodes at two resolutions (from_refined_icosahedron.py:149-176)."""

    def __init__(self, global_resolution: int, lam_resolution: int, name: str, reference_node_name: str,
                 mask_attr_name: str, margin_radius_km: float = 100.0) -> None:
        super().__init__(lam_resolution, name)
        self.global_resolution = global_resolution
        self.area_mask_builder = KNNAreaMaskBuilder(reference_node_name, margin_radius_km, mask_attr_name)

    def register_nodes(self, graph: HeteroData) -> HeteroData:
        self.area_mask_builder.fit(graph)
        return super().register_nodes(graph)


class StretchedTriNodes(StretchedIcosahedronNodes):
    """Icosphere nodes of ``global_resolution`` outside the area of interest plus those of
    ``lam_resolution`` inside it (from_refined_icosahedron.py:179-191 ->
    generate/tri_icosahedron.py:61-105).  Node ids index the concatenation [coarse outside, fine
    inside]; the node order is latitude descending over that concatenation."""

    def create_nodes(self):
        lam_resolution = max(self.resolutions)
        if self.global_resolution > lam_resolution:
            raise ValueError(f"StretchedTriNodes: the global resolution ({self.global_resolution}) must not exceed "
                             f"the resolution inside the area ({lam_resolution})")
        _x, _f, lam, _o, _r, level_edges = ops.icosphere(lam_resolution, ordered=False)  # vertex order
        # Refinement keeps the vertices it starts from, at the head of the vertex list: the coarse mesh is the first
        # 10 * 4**r + 2 vertices of the fine one, and one look-up of the fine vertices answers for both meshes.
        base = lam[: 10 * 4 ** self.global_resolution + 2]
        lam_in = self.area_mask_builder.get_mask(lam, density=1.0)
        base_out = ~lam_in[: base.shape[0]]
        coords_all = torch.cat([base[base_out], lam[lam_in]]).contiguous()
        node_ordering = ops.node_order(coords_all)
        coords = coords_all[node_ordering.long()].contiguous()
        dev = lam.device
        n_base = int(base_out.sum().item())
        # two densities, the coarse mesh outside the area (the share `outside` of the sphere) and the fine one inside:
        # the mean over the nodes of the density around them
        n_all, n_fine = int(coords_all.shape[0]), int(coords_all.shape[0]) - n_base
        outside = n_base / base.shape[0]
        if 0.0 < outside < 1.0 and n_all > 0:
            self.density = max(1.0, (n_base ** 2 / outside + n_fine ** 2 / (1.0 - outside)) / n_all ** 2)
        # mesh vertex -> id in the concatenation (coarse vertices are the first vertices of the fine mesh)
        cid = torch.full((lam.shape[0],), -1, dtype=torch.int64, device=dev)
        cid[torch.nonzero(base_out).squeeze(1)] = torch.arange(n_base, device=dev)
        idx_l = torch.nonzero(lam_in).squeeze(1)
        cid[idx_l] = n_base + torch.arange(idx_l.numel(), device=dev)
        position = torch.empty(coords_all.shape[0], dtype=torch.int32, device=dev)
        position[node_ordering.long()] = torch.arange(coords_all.shape[0], dtype=torch.int32, device=dev)
        rank = torch.where(cid >= 0, position[cid.clamp(min=0)], torch.full_like(position[:1], -1).expand(cid.numel()))
        rank = rank.to(torch.int32).contiguous()
        mesh = IcosahedralMesh(lam_resolution, node_ordering, rank, level_edges, (rank >= 0).to(torch.uint8))
        return _finish(mesh, coords, node_ordering)
