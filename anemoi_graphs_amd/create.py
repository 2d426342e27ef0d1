"""Recipe -> graph: the caller of the plugin classes, with the public surface of
``anemoi.graphs.create.GraphCreator`` (/root/reference/src/anemoi/graphs/create.py:26-188) --
``GraphCreator(config).create(save_path, overwrite)`` and the five steps it is made of.

The recipe is normalised once, at construction, into ``EdgeSet`` records (one per ``edges`` entry, whichever
of the two spellings of the recipe it uses); every later step walks those records or the stores of the graph.
"""
from __future__ import annotations

import logging
import warnings
from pathlib import Path
from typing import Iterator, NamedTuple

from . import hetero, ops
from .hetero import HeteroData
from .instantiate import instantiate

LOGGER = logging.getLogger(__name__)

LEGACY_KEY_NOTICE = (
    "This format will be deprecated. The key 'edge_builder' is renamed to 'edge_builders' and takes a list of edge "
    "builders. In addition, the source_mask_attr_name & target_mask_attr_name fields are moved under the each edge "
    "builder."
)


class DotDict(dict):
    """Mapping whose keys also read as attributes, nested mappings included (the role of
    ``anemoi.utils.config.DotDict`` in the reference)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        for key in list(self):
            super().__setitem__(key, _dotted(self[key]))

    def __getattr__(self, key):
        if key in self:
            return self[key]
        raise AttributeError(key)

    def __setattr__(self, key, value):
        self[key] = _dotted(value)

    @classmethod
    def from_file(cls, path) -> "DotDict":
        import yaml

        with open(path) as stream:
            return cls(yaml.safe_load(stream))


def _dotted(value):
    if isinstance(value, dict):
        return value if isinstance(value, DotDict) else DotDict(value)
    if isinstance(value, list):
        return [_dotted(item) for item in value]
    return value


class EdgeSet(NamedTuple):
    """One entry of the recipe's ``edges`` list: the builders that contribute to the edge set
    (source -> target) and the attributes evaluated on their union."""

    source_name: str
    target_name: str
    builders: list
    attributes: dict


def _modern_entry(entry: DotDict) -> DotDict:
    """The entry with an ``edge_builders`` list.  The older spelling -- a single ``edge_builder`` with the two
    mask names beside it (create.py:42-60 of the reference) -- is rewritten, with the reference's warning."""
    if "edge_builder" not in entry:
        return entry
    warnings.warn(LEGACY_KEY_NOTICE, DeprecationWarning, stacklevel=4)
    single = entry.get("edge_builder")
    if single is not None:
        masks = {key: entry.get(key) for key in ("source_mask_attr_name", "target_mask_attr_name")}
        entry["edge_builders"] = [DotDict({**single, **masks})]
    return entry


class GraphCreator:
    """Builds the graph a recipe describes (create.py:26-188 of the reference)."""

    def __init__(self, config) -> None:
        loaded = DotDict.from_file(config) if isinstance(config, (str, Path)) else DotDict(config)
        loaded.edges = [_modern_entry(entry) for entry in loaded.get("edges") or []]
        self.config = loaded

    # ------------------------------------------------------------------ the recipe, normalised
    def _node_sets(self) -> Iterator[tuple[str, DotDict, dict]]:
        for name, entry in (self.config.get("nodes") or {}).items():
            yield name, entry.node_builder, entry.get("attributes", {})

    def _edge_sets(self) -> Iterator[EdgeSet]:
        for entry in self.config.edges:
            yield EdgeSet(entry.source_name, entry.target_name, list(entry.edge_builders), entry.get("attributes", {}))

    # ------------------------------------------------------------------ the steps of create()
    def update_graph(self, graph: HeteroData) -> HeteroData:
        """Node sets first, in recipe order; then every edge set: all its builders (their edges merge), and the
        attributes once, on the merged list, through the builder that ran last (create.py:78-90)."""
        for name, builder_cfg, attributes in self._node_sets():
            graph = instantiate(builder_cfg, name=name).update_graph(graph, attrs_config=attributes)
        for edge_set in self._edge_sets():
            builder = None
            for builder_cfg in edge_set.builders:
                builder = instantiate(builder_cfg, source_name=edge_set.source_name, target_name=edge_set.target_name)
                graph = builder.update_graph(graph, attrs_config=None)
            if builder is None:
                raise ValueError(f"no edge builder for {edge_set.source_name} -> {edge_set.target_name}")
            graph = builder.register_attributes(graph, edge_set.attributes)
        return graph

    def clean(self, graph: HeteroData) -> HeteroData:
        """Drop what the builders left for each other: every attribute whose name starts with an underscore
        (create.py:94-114)."""
        for store_name in [*graph.node_types, *graph.edge_types]:
            store = graph[store_name]
            for private in [name for name in store.keys() if name.startswith("_")]:
                del store[private]
                LOGGER.info("%s deleted from graph.", private)
        return graph

    def post_process(self, graph: HeteroData) -> HeteroData:
        """Run the recipe's ``post_processors`` in order (create.py:116-140)."""
        for processor_cfg in self.config.get("post_processors") or []:
            graph = instantiate(processor_cfg).update_graph(graph)
        return graph

    def save(self, graph: HeteroData, save_path: Path, overwrite: bool = False) -> None:
        """Write the graph with host tensors, so that the file loads on a machine without a GPU; an existing
        file is kept unless ``overwrite`` (create.py:142-161)."""
        save_path = Path(save_path)
        if save_path.exists() and not overwrite:
            LOGGER.info("Graph already exists. Use overwrite=True to overwrite.")
            return
        save_path.parent.mkdir(parents=True, exist_ok=True)
        hetero.save_graph(hetero.to_host(graph), save_path)
        LOGGER.info("Graph saved at %s.", save_path)

    def create(self, save_path: Path | None = None, overwrite: bool = False) -> HeteroData:
        """update_graph -> clean -> post_process -> save (create.py:163-188); returns the graph."""
        graph = self.post_process(self.clean(self.update_graph(HeteroData())))
        ops.check_ids()  # an id outside its node set met by a kernel of this build is an IndexError here at the latest
        if save_path is None:
            LOGGER.warning("No output path specified. The graph will not be saved.")
        else:
            self.save(graph, save_path, overwrite)
        return graph
