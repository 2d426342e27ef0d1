from .builders import (
    BaseNodeBuilder,
    IcosahedralNodes,
    LatLonNodes,
    LimitedAreaTriNodes,
    StretchedTriNodes,
    TriNodes,
)

__all__ = ["BaseNodeBuilder", "IcosahedralNodes", "TriNodes", "LimitedAreaTriNodes", "StretchedTriNodes", "LatLonNodes"]
