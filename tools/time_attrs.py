"""Times ag_edge_attrs variants on the O1280 -> r7 decoder / encoder edge sets (GPU)."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from anemoi_graphs_amd import grids, ops  # noqa: E402

dev = torch.device("cuda", 0)
data = torch.from_numpy(grids.named("o1280")).to(dev)
_x, _f, hidden, _o, _r, _e = ops.icosphere(7)
dec = ops.knn_edges(hidden, data, 3)
radius = 0.6 * float(ops.nn_maxdist(hidden).item())
enc = ops.radius_edges(data, hidden, radius)


def timeit(fn, n=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for name, sx, tx, e in (("decoder", hidden, data, dec), ("encoder", data, hidden, enc)):
    for wl, wd, rot in ((True, True, True), (True, False, True), (False, True, True), (False, True, False)):
        ms = timeit(lambda: ops.edge_attrs(sx, tx, e, want_len=wl, want_dir=wd, rotated=rot))
        print(f"{name} E={e.shape[1]} len={wl} dir={wd} rotated={rot}: {ms:.3f} ms", flush=True)
