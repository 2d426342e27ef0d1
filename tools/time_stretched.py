"""Device time of every step of a graph over a stretched grid with a StretchedTriNodes hidden mesh (the
configuration of tests/test_stretched_gpu.py): where a realistic stretched build spends its time."""
import sys
import time

import torch

sys.path.insert(0, ".")
from anemoi_graphs_amd import grids, settings  # noqa: E402
from anemoi_graphs_amd.edges import CutOffEdges, KNNEdges, MultiScaleEdges  # noqa: E402
from anemoi_graphs_amd.hetero import HeteroData  # noqa: E402
from anemoi_graphs_amd.nodes import StretchedTriNodes  # noqa: E402

ATTRS = {
    "edge_length": {"_target_": "anemoi.graphs.edges.attributes.EdgeLength", "norm": "unit-max"},
    "edge_dirs": {"_target_": "anemoi.graphs.edges.attributes.EdgeDirection", "norm": None},
}
n = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0e7
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (8, 9)
device = torch.device("cuda", 0)
grid = grids.stretched_for_nodes(n)
data, fine = grid.generate(device=device, return_fine=True)
for trial in range(2):
    graph = HeteroData()
    graph["data"].x = data
    graph["data"].node_type = "LatLonNodes"
    graph["data"]["fine"] = fine[:, None]
    steps = []

    def timed(name, fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        steps.append((name, (time.perf_counter() - t0) * 1e3))
        return out

    with settings.resident(True):
        timed("hidden mesh", lambda: StretchedTriNodes(lo, hi, "hidden", "data", "fine", margin_radius_km=50.0).update_graph(graph, {}))
        enc = CutOffEdges("data", "hidden", 0.6)
        timed("encoder edges", lambda: enc.update_graph(graph))
        timed("encoder attributes", lambda: enc.register_attributes(graph, ATTRS))
        dec = KNNEdges("hidden", "data", 3)
        timed("decoder edges", lambda: dec.update_graph(graph))
        timed("decoder attributes", lambda: dec.register_attributes(graph, ATTRS))
        timed("processor edges", lambda: MultiScaleEdges("hidden", "hidden", 1).update_graph(graph))
    print(f"trial {trial}: data {data.shape[0]:,}, hidden {graph['hidden'].x.shape[0]:,}, encoder "
          f"{graph[('data', 'to', 'hidden')].edge_index.shape[1]:,}, decoder {graph[('hidden', 'to', 'data')].edge_index.shape[1]:,}, "
          f"processor {graph[('hidden', 'to', 'hidden')].edge_index.shape[1]:,} edges", flush=True)
    for name, ms in steps:
        print(f"   {name:22s} {ms:9.2f} ms", flush=True)
