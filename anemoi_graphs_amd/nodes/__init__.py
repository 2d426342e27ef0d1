from .builders import BaseNodeBuilder, IcosahedralNodes, LatLonNodes, TriNodes

__all__ = ["BaseNodeBuilder", "IcosahedralNodes", "TriNodes", "LatLonNodes"]
