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


def timeit(fn, n=40, batches=7):
    """Median and minimum over batches of the mean time of n back-to-back calls (ms)."""
    fn()
    torch.cuda.synchronize()
    res = []
    for _ in range(batches):
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / n * 1e3)
    res.sort()
    return res[len(res) // 2], res[0]


for name, sx, tx, e in (("decoder", hidden, data, dec), ("encoder", data, hidden, enc)):
    for wl, wd, rot in ((True, True, True), (True, False, True), (False, True, True), (False, True, False)):
        med, best = timeit(lambda: ops.edge_attrs(sx, tx, e, want_len=wl, want_dir=wd, rotated=rot))
        print(f"{name} E={e.shape[1]} len={wl} dir={wd} rotated={rot}: median {med:.4f} ms, best {best:.4f} ms", flush=True)
