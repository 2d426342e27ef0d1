"""``_target_`` instantiation for recipe entries (the reference uses ``hydra.utils.instantiate``:
/root/reference/src/anemoi/graphs/create.py:79,85,138; nodes/builders/base.py:77; edges/builder.py:133).

Targets of the reference package (``anemoi.graphs.<...>``) are mapped onto this package so existing
recipes run unchanged."""
from __future__ import annotations

import importlib

_ALIASES = {
    "anemoi.graphs.nodes.TriNodes": "anemoi_graphs_amd.nodes.TriNodes",
    "anemoi.graphs.nodes.LimitedAreaTriNodes": "anemoi_graphs_amd.nodes.LimitedAreaTriNodes",
    "anemoi.graphs.nodes.StretchedTriNodes": "anemoi_graphs_amd.nodes.StretchedTriNodes",
    "anemoi.graphs.nodes.LatLonNodes": "anemoi_graphs_amd.nodes.LatLonNodes",
    "anemoi.graphs.edges.KNNEdges": "anemoi_graphs_amd.edges.KNNEdges",
    "anemoi.graphs.edges.CutOffEdges": "anemoi_graphs_amd.edges.CutOffEdges",
    "anemoi.graphs.edges.MultiScaleEdges": "anemoi_graphs_amd.edges.MultiScaleEdges",
    "anemoi.graphs.processors.RemoveUnconnectedNodes": "anemoi_graphs_amd.processors.RemoveUnconnectedNodes",
    "anemoi.graphs.nodes.attributes.SphericalAreaWeights": "anemoi_graphs_amd.nodes.attributes.SphericalAreaWeights",
    "anemoi.graphs.nodes.attributes.AreaWeights": "anemoi_graphs_amd.nodes.attributes.AreaWeights",
    "anemoi.graphs.nodes.attributes.UniformWeights": "anemoi_graphs_amd.nodes.attributes.UniformWeights",
    "anemoi.graphs.nodes.attributes.BooleanNot": "anemoi_graphs_amd.nodes.attributes.BooleanNot",
    "anemoi.graphs.nodes.attributes.BooleanAndMask": "anemoi_graphs_amd.nodes.attributes.BooleanAndMask",
    "anemoi.graphs.nodes.attributes.BooleanOrMask": "anemoi_graphs_amd.nodes.attributes.BooleanOrMask",
    "anemoi.graphs.edges.attributes.EdgeLength": "anemoi_graphs_amd.edges.attributes.EdgeLength",
    "anemoi.graphs.edges.attributes.EdgeDirection": "anemoi_graphs_amd.edges.attributes.EdgeDirection",
}


_RESOLVED: dict[str, object] = {}  # (the import machinery costs tens of microseconds per look-up, a build makes a dozen)


def resolve(target: str):
    found = _RESOLVED.get(target)
    if found is not None:
        return found
    aliased = _ALIASES.get(target, target)
    module, _, name = aliased.rpartition(".")
    try:
        found = getattr(importlib.import_module(module), name)
    except (ImportError, AttributeError) as e:
        raise ImportError(
            f"cannot instantiate {aliased!r}: only the graph-construction hot path is provided by this package "
            f"({', '.join(sorted(_ALIASES))})"
        ) from e
    _RESOLVED[target] = found
    return found


def instantiate(config, **kwargs):
    cfg = dict(config)
    target = cfg.pop("_target_")
    cfg.update(kwargs)
    return resolve(target)(**cfg)
