from .builder import BaseEdgeBuilder, CutOffEdges, KNNEdges, MultiScaleEdges, NodeMaskingMixin

__all__ = ["BaseEdgeBuilder", "NodeMaskingMixin", "KNNEdges", "CutOffEdges", "MultiScaleEdges"]
