"""``HeteroData`` container: torch_geometric's when it is installed, else a minimal stand-in with
the subset of its interface the graph builders and ``torch.save``/``torch.load`` need."""
from __future__ import annotations

try:  # pragma: no cover - depends on the environment
    from torch_geometric.data import HeteroData  # type: ignore
    from torch_geometric.data.storage import NodeStorage  # type: ignore

    HAVE_TORCH_GEOMETRIC = True
except ImportError:
    HAVE_TORCH_GEOMETRIC = False

    class NodeStorage(dict):
        """Attribute-style dict, like torch_geometric's storages."""

        def __getattr__(self, key):
            if key.startswith("__"):
                raise AttributeError(key)
            try:
                return self[key]
            except KeyError as e:
                raise AttributeError(key) from e

        def __setattr__(self, key, value):
            self[key] = value

        def __delattr__(self, key):
            del self[key]

        def node_attrs(self):
            """Names of the per-node tensors (first dimension = number of nodes)."""
            if "x" not in self:
                return [k for k, v in self.items() if hasattr(v, "shape")]
            n = self["x"].shape[0]
            return [k for k, v in self.items() if hasattr(v, "shape") and len(v.shape) > 0 and v.shape[0] == n]

        @property
        def num_nodes(self):
            return self["x"].shape[0]

    class HeteroData:
        """``graph[name]`` node storages and ``graph[(src, rel, dst)]`` edge storages."""

        def __init__(self):
            self._stores: dict = {}

        def __getitem__(self, key):
            if isinstance(key, list):
                key = tuple(key)
            if key not in self._stores:
                self._stores[key] = NodeStorage()
            return self._stores[key]

        def __delitem__(self, key):
            if isinstance(key, list):
                key = tuple(key)
            del self._stores[key]

        def __contains__(self, key):
            return key in self._stores

        @property
        def node_types(self):
            return [k for k, v in self._stores.items() if isinstance(k, str)]

        @property
        def edge_types(self):
            return [k for k, v in self._stores.items() if isinstance(k, tuple)]

        def edge_items(self):
            return [(k, v) for k, v in self._stores.items() if isinstance(k, tuple)]

        def node_items(self):
            return [(k, v) for k, v in self._stores.items() if isinstance(k, str)]

        def __repr__(self):
            def shape(v):
                return list(v.shape) if hasattr(v, "shape") else repr(v)

            lines = ["HeteroData("]
            for key, store in self._stores.items():
                lines.append(f"  {key}={{")
                for name, value in store.items():
                    lines.append(f"    {name}={shape(value)},")
                lines.append("  },")
            lines.append(")")
            return "\n".join(lines)


# ---------------------------------------------------------------------------------------------------
# graph files
# ---------------------------------------------------------------------------------------------------
PORTABLE_FORMAT = "anemoi_graphs_amd.graph/1"


def to_host(graph):
    """Move every device tensor of the graph to host memory, in place (a graph file must load without a GPU)."""
    import torch

    from .utils import to_host as tensor_to_host

    for name in [*graph.node_types, *graph.edge_types]:
        store = graph[name]
        for key in list(store.keys()):
            value = store[key]
            if isinstance(value, torch.Tensor) and value.is_cuda:
                store[key] = tensor_to_host(value)
    return graph


def to_portable(graph) -> dict:
    """The graph as plain containers -- dicts, lists, strings, numbers and tensors only -- which ``torch.load``
    reads back with ``weights_only=True`` and without this package."""
    return {
        "format": PORTABLE_FORMAT,
        "nodes": {name: dict(graph[name].items()) for name in graph.node_types},
        "edges": [{"source": s, "relation": r, "target": t, "attributes": dict(graph[(s, r, t)].items())}
                  for (s, r, t) in graph.edge_types],
    }


def from_portable(content: dict):
    """``HeteroData`` (torch_geometric's when it is installed, so this is also the converter to the class the
    reference's consumers expect) from the output of ``to_portable``."""
    if not isinstance(content, dict) or content.get("format") != PORTABLE_FORMAT:
        raise ValueError("not a graph written by anemoi_graphs_amd.hetero.save_graph")
    graph = HeteroData()
    for name, attributes in content["nodes"].items():
        for key, value in attributes.items():
            graph[name][key] = value
    for edge_set in content["edges"]:
        store = graph[(edge_set["source"], edge_set["relation"], edge_set["target"])]
        for key, value in edge_set["attributes"].items():
            store[key] = value
    return graph


def save_graph(graph, path) -> None:
    """With torch_geometric installed the graph is a ``torch_geometric.data.HeteroData`` and is written as the
    reference writes it, ``torch.save(graph, path)`` (create.py:158): the file anemoi-training loads.  Without
    torch_geometric the container is this module's stand-in, which no other program could unpickle: the graph is
    then written as plain containers (``to_portable``), and ``load_graph`` -- on a machine that has
    torch_geometric -- turns it into the real class."""
    import torch

    torch.save(graph if HAVE_TORCH_GEOMETRIC else to_portable(graph), path)


def load_graph(path, trusted: bool = False):
    """Read a graph file.  The portable form is read with a loader that executes nothing from the file.  A pickled
    container object (the reference's format, ``torch.save(HeteroData)``) can only be read by running its pickle,
    as torch_geometric's own users do for graph.pt: that happens only when the caller says the file comes from a
    source it trusts (``trusted=True``); otherwise such a file is refused."""
    import pickle

    import torch

    try:
        content = torch.load(path, weights_only=True)
    except pickle.UnpicklingError as e:
        if not trusted:
            raise ValueError(
                f"{path} is not in the portable format and can only be read by executing its pickle; "
                "pass trusted=True if the file comes from a source you trust") from e
        return torch.load(path, weights_only=False)
    return from_portable(content)
