"""Times the spherical Voronoi cell areas (SphericalAreaWeights) on the bench grids (GPU)."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from anemoi_graphs_amd import grids, ops, profiling  # noqa: E402

dev = torch.device("cuda", 0)
for name in sys.argv[1:] or ["o96", "o1280", "tri7"]:
    if name.startswith("tri"):
        x = ops.icosphere(int(name[3:]))[2]
    else:
        x = torch.from_numpy(grids.named(name)).to(dev)
    ops.voronoi_areas(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        area, stats = ops.voronoi_areas(x, return_stats=True)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    profiling.enable(True)
    ops.voronoi_areas(x)
    torch.cuda.synchronize()
    kernels = profiling.collect()
    profiling.enable(False)
    top = sorted(kernels.items(), key=lambda kv: -kv[1][1])[:6]
    print(f"{name}: {x.shape[0]} nodes, {ms:.3f} ms, {stats}, sum/4pi - 1 = {float(area.sum()) / (4 * 3.141592653589793) - 1:.2e}")
    for k, (calls, t) in top:
        print(f"    {k:32s} {calls:3d} launches {t:.3f} ms")
