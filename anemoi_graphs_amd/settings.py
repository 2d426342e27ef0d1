"""Process-wide switches of the package."""
from __future__ import annotations

import contextlib

# When False (default) every tensor placed in the graph lives where the reference puts it: in host
# memory (``graph.pt`` must load without a GPU).  When True, node coordinates, edge indices and
# attributes stay in HBM between builders, and only ``GraphCreator.save`` copies them to the host.
device_resident: bool = False


@contextlib.contextmanager
def resident(flag: bool = True):
    """Context manager that keeps graph tensors on the GPU between builders."""
    global device_resident
    old = device_resident
    device_resident = flag
    try:
        yield
    finally:
        device_resident = old
