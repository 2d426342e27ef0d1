"""Developer tool: device time of the searches in a dozen regimes (orders of the nodes, densities, row lengths, k),
one line each -- run it in two checkouts to see what a change did outside the benchmark workload.
python tools/regime_sweep.py [repeats]"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from anemoi_graphs_amd import grids, ops  # noqa: E402

dev = torch.device("cuda", 0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5


def timed(name, fn):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    size = tuple(out.shape) if isinstance(out, torch.Tensor) else ""
    print(f"{name:58s} {ms:9.3f} ms  {size}", flush=True)


def dev_t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


rng = np.random.default_rng(0)
o1280 = grids.named("o1280")
hidden7 = ops.icosphere(7, dev)[2]
hidden7_np = hidden7.cpu().numpy()
r7 = float(ops.nn_maxdist(hidden7).item()) * 0.6
data, data_shuf = dev_t(o1280), dev_t(o1280[rng.permutation(len(o1280))])
hid_shuf = dev_t(hidden7_np[rng.permutation(len(hidden7_np))])
rand6 = dev_t(grids.random_sphere(1_000_000, seed=1))
rand5 = dev_t(grids.random_sphere(100_000, seed=2))
fib = dev_t(grids.fibonacci_sphere(2_000_000))

timed("radius: o1280 -> r7 (rows of 63)", lambda: ops.radius_edges(data, hidden7, r7))
timed("radius: o1280 shuffled -> r7", lambda: ops.radius_edges(data_shuf, hidden7, r7))
timed("radius: o1280 -> r7 shuffled", lambda: ops.radius_edges(data, hid_shuf, r7))
timed("radius: 1e6 random -> 1e5 random, rows of ~20", lambda: ops.radius_edges(rand6, rand5, 0.0045 * 2))
timed("radius: 1e6 random -> 1e5 random, rows of ~300", lambda: ops.radius_edges(rand6, rand5, 0.035))
timed("radius: 1e6 random -> 2000 random, rows of ~5000", lambda: ops.radius_edges(rand6, rand5[:2000], 0.14))
timed("radius: spiral 2e6 -> r7", lambda: ops.radius_edges(fib, hidden7, r7))
timed("knn: r7 -> o1280, k=3", lambda: ops.knn(hidden7, data, 3))
timed("knn: r7 -> o1280 shuffled, k=3", lambda: ops.knn(hidden7, data_shuf, 3))
timed("knn: r7 -> o1280, k=8", lambda: ops.knn(hidden7, data, 8))
timed("knn: 1e6 random -> 1e5 random, k=1", lambda: ops.knn(rand6, rand5, 1))
timed("knn: 1e6 random -> 1e5 random, k=16", lambda: ops.knn(rand6, rand5, 16))
timed("knn: 1e6 random self, k=10", lambda: ops.knn(rand6, rand6, 10))
timed("nn_maxdist: 1e6 random", lambda: ops.nn_maxdist(rand6))
timed("nn_maxdist: spiral 2e6", lambda: ops.nn_maxdist(fib))
timed("bin_points: o1280, cell 0.006", lambda: ops.bin_points(data, 0.006).buffer)
timed("bin_points: o1280 shuffled, cell 0.006", lambda: ops.bin_points(data_shuf, 0.006).buffer)
timed("bin_points: o1280, cell 0.0005", lambda: ops.bin_points(data, 0.0005).buffer)
