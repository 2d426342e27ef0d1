"""Recipe -> graph orchestration, the counterpart of ``anemoi.graphs.create.GraphCreator``
(/root/reference/src/anemoi/graphs/create.py:26-188) for the builders this package provides."""
from __future__ import annotations

import logging
from itertools import chain
from pathlib import Path
from warnings import warn

import torch

from .hetero import HeteroData
from .instantiate import instantiate

LOGGER = logging.getLogger(__name__)


class DotDict(dict):
    """dict with attribute access, recursively (stand-in for ``anemoi.utils.config.DotDict``)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        for k, v in list(self.items()):
            self[k] = self._wrap(v)

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, DotDict):
            return cls(v)
        if isinstance(v, list):
            return [cls._wrap(i) for i in v]
        return v

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError as e:
            raise AttributeError(key) from e

    def __setattr__(self, key, value):
        self[key] = self._wrap(value)

    @classmethod
    def from_file(cls, path):
        import yaml

        with open(path) as f:
            return cls(yaml.safe_load(f))


class GraphCreator:
    """Graph creator (create.py:26-188)."""

    def __init__(self, config):
        if isinstance(config, (Path, str)):
            self.config = DotDict.from_file(config)
        else:
            self.config = DotDict(config)
        # legacy key 'edge_builder' -> 'edge_builders' (create.py:42-60)
        edges = []
        for edges_cfg in self.config.get("edges", []):
            if "edge_builder" in edges_cfg:
                warn(
                    "This format will be deprecated. The key 'edge_builder' is renamed to 'edge_builders' and takes "
                    "a list of edge builders. In addition, the source_mask_attr_name & target_mask_attr_name fields "
                    "are moved under the each edge builder.",
                    DeprecationWarning,
                    stacklevel=2,
                )
                edge_builder_cfg = edges_cfg.get("edge_builder")
                if edge_builder_cfg is not None:
                    edge_builder_cfg = DotDict(edge_builder_cfg)
                    edge_builder_cfg.source_mask_attr_name = edges_cfg.get("source_mask_attr_name", None)
                    edge_builder_cfg.target_mask_attr_name = edges_cfg.get("target_mask_attr_name", None)
                    edges_cfg["edge_builders"] = [edge_builder_cfg]
            edges.append(edges_cfg)
        self.config.edges = edges

    def update_graph(self, graph: HeteroData) -> HeteroData:
        for nodes_name, nodes_cfg in self.config.get("nodes", {}).items():
            graph = instantiate(nodes_cfg.node_builder, name=nodes_name).update_graph(
                graph, attrs_config=nodes_cfg.get("attributes", {})
            )
        for edges_cfg in self.config.get("edges", {}):
            for edge_builder_cfg in edges_cfg.edge_builders:
                edge_builder = instantiate(
                    edge_builder_cfg, source_name=edges_cfg.source_name, target_name=edges_cfg.target_name
                )
                graph = edge_builder.update_graph(graph, attrs_config=None)
            graph = edge_builder.register_attributes(graph, edges_cfg.get("attributes", {}))
        return graph

    def clean(self, graph: HeteroData) -> HeteroData:
        """Remove the private (underscore) attributes used during creation (create.py:94-114)."""
        for type_name in chain(graph.node_types, graph.edge_types):
            attr_names_to_remove = [attr_name for attr_name in graph[type_name] if attr_name.startswith("_")]
            for attr_name in attr_names_to_remove:
                del graph[type_name][attr_name]
                LOGGER.info(f"{attr_name} deleted from graph.")
        return graph

    def post_process(self, graph: HeteroData) -> HeteroData:
        for processor in self.config.get("post_processors", []):
            graph = instantiate(processor).update_graph(graph)
        return graph

    def save(self, graph: HeteroData, save_path: Path, overwrite: bool = False) -> None:
        save_path = Path(save_path)
        if not save_path.exists() or overwrite:
            save_path.parent.mkdir(parents=True, exist_ok=True)
            # the file must load on a machine without a GPU
            for type_name in chain(graph.node_types, graph.edge_types):
                for attr_name in list(graph[type_name].keys()):
                    value = graph[type_name][attr_name]
                    if isinstance(value, torch.Tensor) and value.is_cuda:
                        graph[type_name][attr_name] = value.cpu()
            torch.save(graph, save_path)
            LOGGER.info(f"Graph saved at {save_path}.")
        else:
            LOGGER.info("Graph already exists. Use overwrite=True to overwrite.")

    def create(self, save_path: Path | None = None, overwrite: bool = False) -> HeteroData:
        graph = HeteroData()
        graph = self.update_graph(graph)
        graph = self.clean(graph)
        graph = self.post_process(graph)
        if save_path is None:
            LOGGER.warning("No output path specified. The graph will not be saved.")
        else:
            self.save(graph, save_path, overwrite)
        return graph
