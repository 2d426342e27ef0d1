"""Developer tool: the cut-off encoder of a workload alone, on a kept binning of the data nodes, with the device
time of every kernel (HIP events).  python tools/time_radius.py [grid] [resolution] [repeats]
Under rocprofv3 --pmc this is the short program whose counters tools/pmc_kernels.py summarises."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from anemoi_graphs_amd import grids, ops, profiling  # noqa: E402

dev = torch.device("cuda", 0)
name = sys.argv[1] if len(sys.argv) > 1 else "o1280"
res = int(sys.argv[2]) if len(sys.argv) > 2 else 7
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
data = torch.from_numpy(grids.named(name)).to(dev)
_, _, hidden, order, rank, level_edges = ops.icosphere(res, dev)
radius = float(ops.nn_maxdist(hidden).item()) * 0.6
kept = ops.bin_points(data, ops.radius_cell(data.shape[0], radius))
e = ops.radius_edges(None, hidden, radius, src_binning=kept)
torch.cuda.synchronize()
profiling.enable(True)
t0 = time.perf_counter()
for _ in range(reps):
    e = ops.radius_edges(None, hidden, radius, src_binning=kept)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps * 1e3
k = profiling.collect()
profiling.enable(False)
print(f"{name} -> r{res}: {e.shape[1]} edges, {dt:.3f} ms per query (host clock, events on)")
for a, b in sorted(k.items(), key=lambda kv: -kv[1][1]):
    print(f"  {a:36s} {b[1] / reps:.4f} ms")
