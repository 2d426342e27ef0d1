"""Developer tool: the cut-off query of the capacity check (spiral data nodes against an icosphere), repeated, with
the device time of every kernel.  python tools/time_capacity_radius.py [nodes] [resolution] [repeats]"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from anemoi_graphs_amd import ops, profiling  # noqa: E402
from tools.capacity_check import fibonacci_on_device  # noqa: E402

dev = torch.device("cuda", 0)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 510_000_000
res = int(sys.argv[2]) if len(sys.argv) > 2 else 9
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
data = fibonacci_on_device(n, dev)
hidden = ops.icosphere(res, dev)[2]
radius = float(ops.nn_maxdist(hidden).item()) * 0.6
for i in range(reps + 1):
    if i == 1:
        profiling.enable(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e = ops.radius_edges(data, hidden, radius)
    torch.cuda.synchronize()
    print(f"call {i}: {e.shape[1]} edges, {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
    del e
k = profiling.collect()
profiling.enable(False)
for a, b in sorted(k.items(), key=lambda kv: -kv[1][1]):
    print(f"  {a:36s} {b[1] / reps:.3f} ms")
