"""MI355X-native graph-construction hot path of anemoi-graphs: TriNodes, KNNEdges, CutOffEdges,
MultiScaleEdges, EdgeLength and EdgeDirection behind the reference's NodeBuilder / EdgeBuilder
plugin interface, running on hand-written HIP kernels (``lib/libanemoi_graphs_hip.so``)."""

EARTH_RADIUS = 6371.0  # km

__all__ = ["EARTH_RADIUS"]
