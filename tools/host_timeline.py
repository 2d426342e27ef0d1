"""Host clock at the stations of a two-stream build of O1280 -> r7 (the sequence of ShardedGraphBuilder._build_single):
how long the host takes to queue every part and how long it waits at the two synchronisations.  A wait of a few
microseconds means that the device had finished before the host asked: the host is what the build waits for there."""
import math
import sys
import time

import torch

sys.path.insert(0, ".")
from anemoi_graphs_amd import grids, ops, settings  # noqa: E402
from anemoi_graphs_amd.distributed import ShardedGraphBuilder  # noqa: E402
from anemoi_graphs_amd.edges import CutOffEdges, KNNEdges, MultiScaleEdges  # noqa: E402
from anemoi_graphs_amd.edges.builder import kept_binning  # noqa: E402
from anemoi_graphs_amd.hetero import HeteroData  # noqa: E402
from anemoi_graphs_amd.nodes import TriNodes  # noqa: E402

dev = torch.device("cuda", 0)
name = sys.argv[1] if len(sys.argv) > 1 else "o1280"
res = int(sys.argv[2]) if len(sys.argv) > 2 else 7
data = torch.from_numpy(grids.named(name)).to(dev)
sb = ShardedGraphBuilder(data, res, overlap_processor=True)
attrs = {
    "edge_length": {"_target_": "anemoi.graphs.edges.attributes.EdgeLength", "norm": "unit-max"},
    "edge_dirs": {"_target_": "anemoi.graphs.edges.attributes.EdgeDirection", "norm": None},
}
stations = {}
waits = {"item": 0.0, "tolist": 0.0, "n": 0}
_item, _tolist = torch.Tensor.item, torch.Tensor.tolist


def _timed(name, fn):
    def call(self, *a, **k):
        if not self.is_cuda:
            return fn(self, *a, **k)
        t0 = time.perf_counter()
        out = fn(self, *a, **k)
        waits[name] += time.perf_counter() - t0
        waits["n"] += 1
        return out
    return call


torch.Tensor.item = _timed("item", _item)
torch.Tensor.tolist = _timed("tolist", _tolist)


def build():
    t = [time.perf_counter()]
    names = []

    def mark(n):
        t.append(time.perf_counter())
        names.append(n)

    graph = HeteroData()
    graph["data"].x = data
    graph["data"].node_type = "LatLonNodes"
    main, side = torch.cuda.current_stream(dev), sb._side_stream(dev)
    guess = 0.6 * 1.185 * math.sqrt(4.0 * math.pi / (10 * 4 ** res + 2))
    side.wait_stream(main)
    with torch.cuda.stream(side):
        kept_binning(graph["data"], ops.radius_cell(data.shape[0], guess))
    mark("queue early binning")
    graph = TriNodes(res, "hidden").update_graph(graph, {})
    mark("queue mesh")
    enc, proc, dec = CutOffEdges("data", "hidden", 0.6), MultiScaleEdges("hidden", "hidden", 1), KNNEdges("hidden", "data", 3)
    for b in (enc, proc, dec):
        graph[(b.source_name, "to", b.target_name)]

    def run(b):
        b.update_graph(graph, attrs_config=None)
        b.register_attributes(graph, attrs)

    side.wait_stream(main)
    with torch.cuda.stream(side):
        run(proc)
    mark("queue processor")
    radius = enc.get_cutoff_radius(graph)
    mark("reference distance (queue + WAIT 1)")
    kept_binning(graph["data"], ops.radius_cell(data.shape[0], radius))
    with torch.cuda.stream(side):
        run(dec)
    mark("queue decoder")
    enc.update_graph(graph, attrs_config=None)
    mark("encoder search (queue + WAIT 2)")
    enc.register_attributes(graph, attrs)
    main.wait_stream(side)
    mark("queue encoder attributes + join")
    for n, a, b in zip(names, t[:-1], t[1:]):
        stations[n] = stations.get(n, 0.0) + (b - a)
    return graph


with settings.resident(True):
    for _ in range(10):
        build()
    torch.cuda.synchronize()
    stations.clear()
    waits.update(item=0.0, tolist=0.0, n=0)
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps):
        build()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
print(f"{name} -> r{res}: {wall * 1e3:.3f} ms per build")
for n, v in stations.items():
    print(f"  {n:44s} {v / reps * 1e6:8.1f} us")
print(f"  {'sum of the stations':44s} {sum(stations.values()) / reps * 1e6:8.1f} us")
print(f"  of which inside .item() / .tolist() of device tensors ({waits['n'] / reps:.1f} per build): "
      f"{(waits['item'] + waits['tolist']) / reps * 1e6:.1f} us")
