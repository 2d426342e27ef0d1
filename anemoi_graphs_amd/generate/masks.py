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
        assert isinstance(margin_radius_km, (int, float)), "The margin radius must be a number."
        assert margin_radius_km > 0, "The margin radius must be positive."
        self.margin_radius_km = margin_radius_km
        self.reference_node_name = reference_node_name
        self.mask_attr_name = mask_attr_name
        self._reference: torch.Tensor | None = None

    def get_reference_coords(self, graph) -> torch.Tensor:
        """Coordinates of the reference nodes (masks.py:56-70)."""
        assert (
            self.reference_node_name in graph.node_types
        ), f'Reference node "{self.reference_node_name}" not found in the graph.'
        coords_rad = torch.as_tensor(graph[self.reference_node_name].x)
        if self.mask_attr_name is not None:
            assert (
                self.mask_attr_name in graph[self.reference_node_name].node_attrs()
            ), f'Mask attribute "{self.mask_attr_name}" not found in the reference nodes.'
            mask = torch.as_tensor(graph[self.reference_node_name][self.mask_attr_name]).squeeze().bool()
            coords_rad = coords_rad[mask.to(coords_rad.device)]
        return coords_rad

    def fit_coords(self, coords_rad) -> None:
        dev = ops.device_of(torch.as_tensor(coords_rad))
        self._reference = ops.coords(np.asarray(coords_rad) if not isinstance(coords_rad, torch.Tensor) else coords_rad, dev)

    def fit(self, graph) -> None:
        reference_mask_str = self.reference_node_name
        if self.mask_attr_name is not None:
            reference_mask_str += f" ({self.mask_attr_name})"
        coords_rad = self.get_reference_coords(graph)
        self.fit_coords(coords_rad)
        LOGGER.info(
            'Fitting %s with %d reference nodes from "%s".', self.__class__.__name__, len(coords_rad), reference_mask_str
        )

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
