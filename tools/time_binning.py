"""Developer tool: the binning of a node set alone, with the device time of every kernel (HIP events).
python tools/time_binning.py [grid | stretched:<nodes> | spiral:<nodes> | random:<nodes>] [mesh resolution, for the cell
size] [repeats]        (spiral: a Fibonacci spiral, consecutive nodes far apart; random: uniform, in random order)"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from anemoi_graphs_amd import grids, ops, profiling  # noqa: E402

dev = torch.device("cuda", 0)
name = sys.argv[1] if len(sys.argv) > 1 else "o1280"
res = int(sys.argv[2]) if len(sys.argv) > 2 else 7
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
if name.startswith("stretched:"):
    n = float(name.split(":")[1])
    fine = 1.0 * (5.0e8 / n) ** 0.5
    data = grids.stretched_for_nodes(n, 9.0 * fine, fine).generate(device=dev)
elif name.startswith("spiral:"):
    from tools.capacity_check import fibonacci_on_device

    data = fibonacci_on_device(int(float(name.split(":")[1])), dev)
elif name.startswith("random:"):
    n = int(float(name.split(":")[1]))
    gen = torch.Generator(device=dev).manual_seed(1)
    data = torch.empty((n, 2), dtype=torch.float32, device=dev)
    data[:, 0] = torch.asin(torch.rand(n, generator=gen, device=dev, dtype=torch.float64) * 2.0 - 1.0).to(torch.float32)
    data[:, 1] = (torch.rand(n, generator=gen, device=dev, dtype=torch.float64) * 6.283185307179586).to(torch.float32)
else:
    data = torch.from_numpy(grids.named(name)).to(dev)
hidden = ops.icosphere(res, dev)[2]
radius = float(ops.nn_maxdist(hidden).item()) * 0.6
cell = ops.radius_cell(data.shape[0], radius)
print(f"{name}: {data.shape[0]} nodes, cell {cell:.3e} rad", flush=True)
kept = ops.bin_points(data, cell)
torch.cuda.synchronize()
profiling.enable(True)
t0 = time.perf_counter()
for _ in range(reps):
    kept = ops.bin_points(data, cell)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps * 1e3
k = profiling.collect()
profiling.enable(False)
print(f"{dt:.3f} ms per binning (host clock, events on)")
for a, b in sorted(k.items(), key=lambda kv: -kv[1][1]):
    print(f"  {a:36s} {b[1] / reps:.4f} ms")
