"""Node attributes with the interface of ``anemoi.graphs.nodes.attributes``
(/root/reference/src/anemoi/graphs/nodes/attributes.py).

``SphericalAreaWeights`` -- the expensive one: the reference builds scipy's ``SphericalVoronoi`` of the
nodes on the CPU -- is evaluated by the HIP kernels (one Voronoi cell per thread, clipped by the
node's nearest nodes).  ``UniformWeights`` and the boolean mask combinators are host logic.  The
attributes that read datasets (``NonmissingZarrVariable``, ``CutOutMask``) and ``PlanarAreaWeights``
(qhull with joggled input, ``QJ``: random by construction) are not provided.
"""
from __future__ import annotations

import logging
from abc import ABC, abstractmethod

import numpy as np
import torch

from .. import ops
from ..utils import place

LOGGER = logging.getLogger(__name__)



def _torch_dtype(dtype) -> torch.dtype:
    """The torch dtype of anything numpy's ``astype`` accepts ("float32", "f8", "float", np.float16, ...)."""
    name = np.dtype(dtype).name  # TypeError "data type ... not understood" for an unknown name, as astype raises
    if not hasattr(torch, name):
        raise TypeError(f"data type {dtype!r} has no torch counterpart")
    return getattr(torch, name)



class BaseNodeAttribute(ABC):
    """Base class for the weights of the nodes (nodes/attributes.py:35-73)."""

    def __init__(self, norm: str | None = None, dtype: str = "float32") -> None:
        self.norm = norm
        self.dtype = dtype

    @abstractmethod
    def get_raw_values(self, nodes, **kwargs) -> torch.Tensor:
        """float64 (or bool) values, one per node, on the device of the coordinates."""

    def post_process(self, values: torch.Tensor) -> torch.Tensor:
        """normalise(values).astype(dtype) as a [N,1] tensor (nodes/attributes.py:45-52)."""
        if values.ndim == 1:
            values = values[:, None]
        if values.dtype == torch.bool:
            return place(values)
        if self.norm not in ops.NORMS:
            # the message of NormaliserMixin.normalise (normalise.py:53-55)
            raise ValueError(
                f"Attribute normalisation \"{self.norm}\" is not valid. Options are: 'l1', 'l2', 'unit-max' or 'unit-std'."
            )
        values = values.to(torch.float64).contiguous()
        if not values.is_cuda:
            values = values.to(ops.device_of(values))
        wanted = _torch_dtype(self.dtype)
        if wanted == torch.float32:
            # float64 statistics and division, one rounding to float32
            return place(ops.normalise_f64(values, self.norm))
        # any other dtype: the float64 result converted once (float64 itself: untouched), as astype does
        return place(ops.normalise_f64(values, self.norm, out_dtype=torch.float64).to(wanted))

    def compute(self, graph, nodes_name: str, **kwargs) -> torch.Tensor:
        nodes = graph[nodes_name]
        return self.post_process(self.get_raw_values(nodes, **kwargs))


class UniformWeights(BaseNodeAttribute):
    """A uniform weight for the nodes (nodes/attributes.py:76-100)."""

    def get_raw_values(self, nodes, **kwargs) -> torch.Tensor:
        x = nodes.x
        return torch.ones(x.shape[0], dtype=torch.float64, device=ops.device_of(x))


class SphericalAreaWeights(BaseNodeAttribute):
    """Area of the spherical Voronoi cell of every node (nodes/attributes.py:165-221)."""

    def __init__(self, norm: str | None = None, radius: float = 1.0, centre=np.array([0, 0, 0]),
                 fill_value: float = 0.0, dtype: str = "float32") -> None:
        super().__init__(norm, dtype)
        self.radius = radius
        self.centre = centre
        self.fill_value = fill_value

    def get_raw_values(self, nodes, **kwargs) -> torch.Tensor:
        # The nodes are unit vectors, so scipy accepts them only for a unit sphere at the origin
        # (SphericalVoronoi checks that the generators lie on the given sphere); its errors are kept.
        if float(self.radius) <= 0.0 or abs(float(self.radius) - 1.0) > 1.0e-6:
            raise ValueError("Radius inconsistent with generators.")
        if np.any(np.abs(np.asarray(self.centre, dtype=np.float64)) > 1.0e-6):
            raise ValueError("Radius inconsistent with generators.")
        areas = ops.voronoi_areas(nodes.x)
        # every node of a set without duplicates owns a region, so fill_value is never needed here
        # (the reference fills regions that scipy reports empty, nodes/attributes.py:202-215)
        LOGGER.debug("There are %d of weights, which (unscaled) add up a total weight of %.2f.", areas.numel(),
                     float(areas.sum()))
        return areas * float(self.radius) ** 2


class AreaWeights(BaseNodeAttribute):
    """Deprecated front of the area weights (nodes/attributes.py:103-129)."""

    def __new__(cls, flat: bool = False, **kwargs):
        logging.warning(
            "Creating %s with flat=%s and kwargs=%s. In a future release, AreaWeights will be deprecated: please use directly PlanarAreaWeights or SphericalAreaWeights.",
            cls.__name__, flat, kwargs,
        )
        if flat:
            raise NotImplementedError("PlanarAreaWeights (qhull with joggled input) is not provided by this package")
        return SphericalAreaWeights(**kwargs)

    def get_raw_values(self, nodes, **kwargs):  # pragma: no cover - never instantiated
        raise NotImplementedError


class BooleanBaseNodeAttribute(BaseNodeAttribute, ABC):
    """Base class for boolean node attributes (nodes/attributes.py:224-228)."""

    def __init__(self) -> None:
        super().__init__(norm=None, dtype="bool")


class BooleanOperation(BooleanBaseNodeAttribute, ABC):
    """Base class for boolean operations on masks (nodes/attributes.py:271-294)."""

    def __init__(self, masks) -> None:
        super().__init__()
        self.masks = masks if isinstance(masks, list) else [masks]

    @staticmethod
    def get_mask_values(mask, nodes, **kwargs) -> torch.Tensor:
        if isinstance(mask, str):
            attributes = nodes[mask]
            assert attributes.dtype in ("bool", torch.bool), (
                f"The mask attribute '{mask}' must be a boolean but is {attributes.dtype}."
            )
            return torch.as_tensor(attributes).reshape(-1)
        return torch.as_tensor(mask.get_raw_values(nodes, **kwargs)).reshape(-1)

    @abstractmethod
    def reduce_op(self, masks: list[torch.Tensor]) -> torch.Tensor: ...

    def get_raw_values(self, nodes, **kwargs) -> torch.Tensor:
        mask_values = [BooleanOperation.get_mask_values(mask, nodes, **kwargs) for mask in self.masks]
        return self.reduce_op(mask_values)


class BooleanNot(BooleanOperation):
    """Boolean NOT mask (nodes/attributes.py:297-302)."""

    def reduce_op(self, masks):
        assert len(self.masks) == 1, f"The {self.__class__.__name__} can only be aplied to one mask."
        return ~masks[0]


class BooleanAndMask(BooleanOperation):
    """Boolean AND mask (nodes/attributes.py:305-309)."""

    def reduce_op(self, masks):
        out = masks[0]
        for m in masks[1:]:
            out = out & m
        return out


class BooleanOrMask(BooleanOperation):
    """Boolean OR mask (nodes/attributes.py:312-316)."""

    def reduce_op(self, masks):
        out = masks[0]
        for m in masks[1:]:
            out = out | m
        return out
