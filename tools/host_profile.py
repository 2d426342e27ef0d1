"""Where the host's time of a build goes: cProfile over builds of a tiny graph (negligible device work)."""
import cProfile
import pstats
import sys

import torch

sys.path.insert(0, ".")
from anemoi_graphs_amd import grids, settings  # noqa: E402
from anemoi_graphs_amd.distributed import ShardedGraphBuilder  # noqa: E402

dev = torch.device("cuda", 0)
data = torch.from_numpy(grids.octahedral(32)).to(dev)
b = ShardedGraphBuilder(data, 3, overlap_processor=False)
with settings.resident(True):
    for _ in range(20):
        b.build()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(200):
        b.build()
    torch.cuda.synchronize()
    pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
st.sort_stats("tottime").print_stats(30)
