"""TEST INFRASTRUCTURE ONLY -- loader for the *unmodified* reference package.

Runs only in the build container, where ``/root/reference`` exists.  It is used
by ``oracle/make_golden.py`` to write the committed fixtures under
``tests/golden/`` and by the oracle self-checks; it never travels to the GPU box
and nothing in the product package imports it.

The reference (``/root/reference/src/anemoi/graphs``) delegates all arithmetic
on the hot path to numpy / scipy / scikit-learn / networkx, which are installed.
What is *not* installed are the container libraries it only uses for plumbing
(``torch_geometric`` as a dict-of-storages, ``hydra.utils.instantiate``,
``anemoi.utils.config.DotDict``, ``omegaconf``, ``typeguard``) and libraries of
out-of-scope builders (``h3``, ``healpy``, ``netCDF4``, ``anemoi.datasets``).
Those are replaced by inert duck-typed modules so the reference's own
``KNNEdges`` / ``CutOffEdges`` / ``EdgeLength`` / ``EdgeDirection`` classes run
unmodified.  ``trimesh`` is absent too: the module placed here exposes
``creation.icosphere`` backed by ``oracle.icosphere`` (our restatement of
trimesh's published algorithm), so ``TriNodes``/``MultiScaleEdges`` results
obtained through this harness are "parity unpinned" with respect to trimesh's
vertex coordinates; only the reference's own hop / mapping / ordering logic is
exercised for them.
"""
from __future__ import annotations

import importlib
import sys
import types
from pathlib import Path

REFERENCE_SRC = Path("/root/reference/src")


class _Storage(dict):
    """dict with attribute access, the subset of torch_geometric's storages used."""

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError as e:
            raise AttributeError(key) from e

    def __setattr__(self, key, value):
        self[key] = value

    def __delattr__(self, key):
        del self[key]

    def node_attrs(self):
        return [k for k in self.keys()]

    @property
    def num_nodes(self):
        return self["x"].shape[0]


class HeteroData:
    """Duck-typed stand-in: ``graph[name]`` / ``graph[(src, rel, dst)]`` storages."""

    def __init__(self):
        self._stores: dict = {}

    def __getitem__(self, key):
        if isinstance(key, list):
            key = tuple(key)
        if key not in self._stores:
            self._stores[key] = _Storage()
        return self._stores[key]

    def __contains__(self, key):
        return key in self._stores

    @property
    def node_types(self):
        return [k for k in self._stores if isinstance(k, str)]

    @property
    def edge_types(self):
        return [k for k in self._stores if isinstance(k, tuple)]


class DotDict(dict):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        for k, v in list(self.items()):
            if isinstance(v, dict) and not isinstance(v, DotDict):
                self[k] = DotDict(v)

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError as e:
            raise AttributeError(key) from e

    def __setattr__(self, key, value):
        self[key] = value


def _instantiate(cfg, **kwargs):
    cfg = dict(cfg)
    target = cfg.pop("_target_")
    mod, _, name = target.rpartition(".")
    cls = getattr(importlib.import_module(mod), name)
    cfg.update(kwargs)
    return cls(**cfg)


def _module(name: str, **attrs) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _IcoMesh:
    def __init__(self, vertices, faces):
        import numpy as np

        self.vertices = vertices
        self.faces = faces
        e = np.sort(faces[:, [0, 1, 1, 2, 2, 0]].reshape(-1, 2), axis=1)
        self.edges_unique = np.unique(e, axis=0)


def _icosphere(subdivisions=3, radius=1.0, **_):
    from oracle import icosphere as ico

    v, f = ico.icosphere(int(subdivisions))
    return _IcoMesh(v * float(radius), f)


_installed = False


def install() -> None:
    """Put the reference on ``sys.path`` with inert plumbing modules."""
    global _installed
    if _installed:
        return
    if not REFERENCE_SRC.exists():
        raise RuntimeError("reference sources are not available on this machine")

    def identity_decorator(f=None, **_):
        return f if f is not None else (lambda g: g)

    _module("typeguard", typechecked=identity_decorator)
    _module("hydra")
    _module("hydra.utils", instantiate=_instantiate)
    _module("omegaconf", DictConfig=type("DictConfig", (dict,), {}), OmegaConf=type("OmegaConf", (), {}))
    tg = _module("torch_geometric")
    tgd = _module("torch_geometric.data", HeteroData=HeteroData)
    tgs = _module("torch_geometric.data.storage", NodeStorage=_Storage)
    tg.data, tgd.storage = tgd, tgs
    for name in ("h3", "healpy", "netCDF4"):
        _module(name)
    tm = _module("trimesh")
    tm.creation = _module("trimesh.creation", icosphere=_icosphere)
    tm.Trimesh = _IcoMesh

    # the namespace package "anemoi" with the reference's "graphs" plus plumbing stand-ins
    anemoi = types.ModuleType("anemoi")
    anemoi.__path__ = [str(REFERENCE_SRC / "anemoi")]
    sys.modules["anemoi"] = anemoi
    au = _module("anemoi.utils")
    au.config = _module("anemoi.utils.config", DotDict=DotDict)
    ad = _module("anemoi.datasets", open_dataset=None)
    ad.__path__ = []
    _installed = True


def reference():
    """Return a namespace with the reference classes on the hot path."""
    install()
    ns = types.SimpleNamespace()
    ns.HeteroData = HeteroData
    ns.DotDict = DotDict
    from anemoi.graphs.edges.builder import CutOffEdges, KNNEdges, MultiScaleEdges
    from anemoi.graphs.edges.attributes import EdgeDirection, EdgeLength
    from anemoi.graphs.nodes.builders.from_refined_icosahedron import LimitedAreaTriNodes, StretchedTriNodes, TriNodes
    from anemoi.graphs.utils import concat_edges, get_grid_reference_distance, haversine_distance
    from anemoi.graphs.generate.utils import get_coordinates_ordering

    from anemoi.graphs.nodes.attributes import (BooleanAndMask, BooleanNot, BooleanOrMask, SphericalAreaWeights,
                                                UniformWeights)

    ns.SphericalAreaWeights, ns.UniformWeights = SphericalAreaWeights, UniformWeights
    ns.BooleanNot, ns.BooleanAndMask, ns.BooleanOrMask = BooleanNot, BooleanAndMask, BooleanOrMask
    ns.KNNEdges, ns.CutOffEdges, ns.MultiScaleEdges = KNNEdges, CutOffEdges, MultiScaleEdges
    ns.EdgeLength, ns.EdgeDirection, ns.TriNodes = EdgeLength, EdgeDirection, TriNodes
    ns.LimitedAreaTriNodes, ns.StretchedTriNodes = LimitedAreaTriNodes, StretchedTriNodes
    ns.concat_edges = concat_edges
    ns.get_grid_reference_distance = get_grid_reference_distance
    ns.haversine_distance = haversine_distance
    ns.get_coordinates_ordering = get_coordinates_ordering
    return ns
