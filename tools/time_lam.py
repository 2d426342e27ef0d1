"""Device time of every step of a limited-area recipe: a regional 2.5 km grid inside a global 0.25-degree grid
(one data node set with a `cutout` mask), a LimitedAreaTriNodes hidden mesh, cut-off and k-NN encoders with a
source mask, a k-NN decoder, multi-scale processor edges, their attributes, and the post-processor that drops the
data nodes no edge touches.  Usage: python tools/time_lam.py [mesh level]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from anemoi_graphs_amd import settings  # noqa: E402
from anemoi_graphs_amd.edges import CutOffEdges, KNNEdges, MultiScaleEdges  # noqa: E402
from anemoi_graphs_amd.hetero import HeteroData  # noqa: E402
from anemoi_graphs_amd.nodes import LimitedAreaTriNodes  # noqa: E402
from anemoi_graphs_amd.processors import RemoveUnconnectedNodes  # noqa: E402

ATTRS = {
    "edge_length": {"_target_": "anemoi.graphs.edges.attributes.EdgeLength", "norm": "unit-max"},
    "edge_dirs": {"_target_": "anemoi.graphs.edges.attributes.EdgeDirection", "norm": None},
}
level = int(sys.argv[1]) if len(sys.argv) > 1 else 10
device = torch.device("cuda", 0)


def regular(lat0, lat1, lon0, lon1, step):
    lat = torch.arange(lat1, lat0 - 1e-9, -step, dtype=torch.float64, device=device)
    lon = torch.arange(lon0, lon1 + 1e-9, step, dtype=torch.float64, device=device)
    la, lo = torch.meshgrid(lat, lon, indexing="ij")
    return torch.deg2rad(torch.stack([la.reshape(-1), lo.reshape(-1)], 1)).float()


regional = regular(35.0, 72.0, -25.0, 45.0, 0.0225)
world = regular(-90.0, 90.0, 0.0, 359.75, 0.25)
data = torch.cat([regional, world]).contiguous()
cutout = torch.zeros(data.shape[0], 1, dtype=torch.bool, device=device)
cutout[: regional.shape[0]] = True
for trial in range(2):
    graph = HeteroData()
    graph["data"].x = data
    graph["data"].node_type = "LatLonNodes"
    graph["data"]["cutout"] = cutout
    steps = []

    def timed(name, fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        steps.append((name, (time.perf_counter() - t0) * 1e3))
        return out

    with settings.resident(True):
        timed("hidden mesh", lambda: LimitedAreaTriNodes(level, "data", "hidden", mask_attr_name="cutout", margin_radius_km=100.0).update_graph(graph, {}))
        enc = CutOffEdges("data", "hidden", 0.6, source_mask_attr_name="cutout")
        timed("cut-off encoder edges", lambda: enc.update_graph(graph))
        timed("cut-off encoder attributes", lambda: enc.register_attributes(graph, ATTRS))
        n_enc = graph[("data", "to", "hidden")].edge_index.shape[1]
        del graph[("data", "to", "hidden")]
        knn_enc = KNNEdges("data", "hidden", 12, source_mask_attr_name="cutout")
        timed("k-NN encoder edges", lambda: knn_enc.update_graph(graph))
        timed("k-NN encoder attributes", lambda: knn_enc.register_attributes(graph, ATTRS))
        dec = KNNEdges("hidden", "data", 3, target_mask_attr_name="cutout")
        timed("decoder edges", lambda: dec.update_graph(graph))
        timed("decoder attributes", lambda: dec.register_attributes(graph, ATTRS))
        timed("processor edges", lambda: MultiScaleEdges("hidden", "hidden", 1).update_graph(graph))
        timed("remove unconnected", lambda: RemoveUnconnectedNodes("data", save_mask_indices_to_attr="kept").update_graph(graph))
    print(f"trial {trial}: data {data.shape[0]:,} ({regional.shape[0]:,} regional), hidden {graph['hidden'].x.shape[0]:,}, cut-off encoder "
          f"{n_enc:,}, k-NN encoder {graph[('data', 'to', 'hidden')].edge_index.shape[1]:,}, decoder "
          f"{graph[('hidden', 'to', 'data')].edge_index.shape[1]:,}, processor {graph[('hidden', 'to', 'hidden')].edge_index.shape[1]:,} edges, "
          f"data nodes kept {graph['data'].x.shape[0]:,}", flush=True)
    for name, ms in steps:
        print(f"   {name:28s} {ms:9.2f} ms", flush=True)
