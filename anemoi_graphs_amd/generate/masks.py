"""Area masks from the distance to a cloud of reference nodes
(/root/reference/src/anemoi/graphs/generate/masks.py:23-99): the nearest-neighbour query runs on
the HIP k-NN kernels instead of scikit-learn's ball tree."""
from __future__ import annotations

import logging

import numpy as np
import torch

from .. import ops

LOGGER = logging.getLogger(__name__)
EARTH_RADIUS = 6371.0  # km, /root/reference/src/anemoi/graphs/__init__.py


class KNNAreaMaskBuilder:
    """Mask of the points closer than ``margin_radius_km`` to the (optionally masked) reference
    nodes (masks.py:23-99; same constructor, ``fit``, ``fit_coords``, ``get_mask``)."""

    def __init__(self, reference_node_name: str, margin_radius_km: float = 100, mask_attr_name: str | None = None):
        if not isinstance(margin_radius_km, (int, float)):
            raise AssertionError("The margin radius must be a number.")
        if not margin_radius_km > 0:
            raise AssertionError("The margin radius must be positive.")
        self.reference_node_name, self.mask_attr_name = reference_node_name, mask_attr_name
        self.margin_radius_km = margin_radius_km
        self._reference: torch.Tensor | None = None  # device coordinates of the fitted nodes

    @property
    def _described(self) -> str:
        """The reference nodes as the log names them: ``name`` or ``name (mask attribute)``."""
        return self.reference_node_name + (f" ({self.mask_attr_name})" if self.mask_attr_name is not None else "")

    def get_reference_coords(self, graph) -> torch.Tensor:
        """[n, 2] coordinates (radians) of the reference nodes, only those the mask attribute marks when one was
        named (masks.py:56-70; a tensor on whichever device the graph keeps it, where the reference returns numpy)."""
        if self.reference_node_name not in graph.node_types:
            raise AssertionError(f'Reference node "{self.reference_node_name}" not found in the graph.')
        nodes = graph[self.reference_node_name]
        coords = torch.as_tensor(nodes.x)
        if self.mask_attr_name is None:
            return coords
        if self.mask_attr_name not in nodes.node_attrs():
            raise AssertionError(f'Mask attribute "{self.mask_attr_name}" not found in the reference nodes.')
        keep = torch.as_tensor(nodes[self.mask_attr_name]).reshape(-1).to(device=coords.device, dtype=torch.bool)
        return coords[keep]

    def fit_coords(self, coords_rad) -> None:
        """Take the reference nodes (masks.py:72-74): here they only move to the device; the search structure is
        built by the query."""
        points = coords_rad if isinstance(coords_rad, torch.Tensor) else np.asarray(coords_rad)
        self._reference = ops.coords(points, ops.device_of(torch.as_tensor(points)))

    def fit(self, graph) -> None:
        """Take the reference nodes from a graph (masks.py:76-92)."""
        self.fit_coords(self.get_reference_coords(graph))
        LOGGER.info('Fitting %s with %d reference nodes from "%s".', type(self).__name__, self._reference.shape[0],
                    self._described)

    def get_mask(self, coords_rad, density: float | None = None) -> torch.Tensor:
        """bool [N] device tensor: distance to the nearest reference node, in km, within the margin
        (masks.py:94-99).  ``density``: of the queried points when the caller knows it (``ops.point_density``; 1 for
        the vertices of a global mesh), else it is measured."""
        assert self._reference is not None, "fit() or fit_coords() must be called first."
        query = ops.coords(coords_rad if isinstance(coords_rad, torch.Tensor) else np.asarray(coords_rad),
                           self._reference.device)
        # The mask asks whether a reference node lies within the margin, not for the nearest one wherever it is:
        # the search stops at the margin (a vertex with nothing that close comes back with an infinite distance),
        # so vertices far from a regional reference set cost a few empty window look-ups instead of a scan.
        bound = float(self.margin_radius_km) / EARTH_RADIUS * (1.0 + 1.0e-9)
        _ids, dists = ops.knn(self._reference, query, 1, return_dist=True, max_dist=bound, tgt_density=density)
        return dists[:, 0] * EARTH_RADIUS <= self.margin_radius_km
