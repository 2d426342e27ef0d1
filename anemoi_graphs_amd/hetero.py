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
