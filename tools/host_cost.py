"""Wall time per build of a tiny graph (the device work is negligible: what remains is the host's)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from anemoi_graphs_amd import grids, settings  # noqa: E402
from anemoi_graphs_amd.distributed import ShardedGraphBuilder  # noqa: E402

dev = torch.device("cuda", 0)
data = torch.from_numpy(grids.octahedral(int(sys.argv[1]) if len(sys.argv) > 1 else 32)).to(dev)
res = int(sys.argv[2]) if len(sys.argv) > 2 else 3
for overlap in (False, True, False, True):
    b = ShardedGraphBuilder(data, res, overlap_processor=overlap)
    with settings.resident(True):
        for _ in range(20):
            b.build()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            b.build()
        torch.cuda.synchronize()
    print(f"overlap={overlap}: {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms per build", flush=True)
