"""Where the host's time of a build goes: cProfile over builds of a tiny graph (the device work is negligible).
python tools/host_profile.py [octahedral N] [resolution] [overlap 0/1]"""
import cProfile
import pstats
import sys

import torch

sys.path.insert(0, ".")
from anemoi_graphs_amd import grids, settings  # noqa: E402
from anemoi_graphs_amd.distributed import ShardedGraphBuilder  # noqa: E402

dev = torch.device("cuda", 0)
data = torch.from_numpy(grids.octahedral(int(sys.argv[1]) if len(sys.argv) > 1 else 32)).to(dev)
res = int(sys.argv[2]) if len(sys.argv) > 2 else 3
overlap = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
b = ShardedGraphBuilder(data, res, overlap_processor=overlap)
with settings.resident(True):
    for _ in range(20):
        b.build()
    torch.cuda.synchronize()
    prof = cProfile.Profile()
    prof.enable()
    for _ in range(200):
        b.build()
    torch.cuda.synchronize()
    prof.disable()
stats = pstats.Stats(prof)
stats.sort_stats("tottime").print_stats(45)
stats.sort_stats("cumulative").print_stats(130)
