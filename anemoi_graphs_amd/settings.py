"""Process-wide switches of the package."""
from __future__ import annotations

import contextlib

# When False (default) every tensor placed in the graph lives where the reference puts it: in host
# memory (``graph.pt`` must load without a GPU).  When True, node coordinates, edge indices and
# attributes stay in HBM between builders, and only ``GraphCreator.save`` copies them to the host.
device_resident: bool = False


@contextlib.contextmanager
def resident(flag: bool = True):
    """Context manager that keeps graph tensors on the GPU between builders.

    Kernels that follow caller-supplied ids report an id outside its table through a device-side error word
    instead of a synchronisation per call (``ops.check_ids``); leaving the outermost resident context reads the
    word, so an IndexError of the reference surfaces here at the latest."""
    global device_resident
    old = device_resident
    device_resident = flag
    try:
        yield
    except BaseException:
        device_resident = old
        if flag and not old:
            # the body failed for a reason of its own: what its kernels left in the error words belongs to this
            # build and must not surface in a later, unrelated one -- read and dropped
            from . import ops

            ops.discard_checks()
        raise
    device_resident = old
    if flag and not old:
        from . import ops

        ops.check_ids()
