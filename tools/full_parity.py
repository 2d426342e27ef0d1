"""Every row of a benchmark configuration against scikit-learn called as the reference calls it
(edges/builder.py:259-265, 364-366 of the reference): the complete decoder (k-NN) and encoder (cut-off) edge lists
of one build, not samples.  Outside pytest because of its run time (minutes of CPU for O1280 -> r7).

    python tools/full_parity.py o1280_r7 > profiles/r03_full_parity_o1280_r7.txt

Prints, per edge set: rows / edges compared, rows that differ only by exact float64 ties (the reference breaks
those by ball-tree traversal order), edges on the radius boundary, and anything else -- which is a failure."""
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from anemoi_graphs_amd import grids, settings  # noqa: E402
from anemoi_graphs_amd.distributed import ShardedGraphBuilder  # noqa: E402
from oracle import compare as h  # noqa: E402
from oracle import reference_path as rp  # noqa: E402

CASES = {"o1280_r7": ("o1280", 7), "era5_r6": ("era5", 6), "o320_r5": ("o320", 5)}
name = sys.argv[1] if len(sys.argv) > 1 else "o1280_r7"
grid, resolution = CASES[name]
rp.N_JOBS = max(4, min(16, os.cpu_count() or 4))
data = grids.named(grid)
device = torch.device("cuda", 0)
builder = ShardedGraphBuilder(torch.from_numpy(data).to(device), resolution, cutoff_factor=0.6, num_nearest_neighbours=3,
                              x_hops=1, attributes=False)
with settings.resident(True):
    graph = builder.build()
torch.cuda.synchronize()
hidden = graph["hidden"].x.cpu().numpy()
print(f"{name}: {len(data)} data nodes, {len(hidden)} hidden nodes; scikit-learn with n_jobs={rp.N_JOBS}", flush=True)
assert np.array_equal(hidden, rp.tri_nodes(resolution)[0]), "hidden nodes differ from the oracle's"
failed = False

t0 = time.perf_counter()
dec = graph[("hidden", "to", "data")].edge_index.cpu().numpy()
rows = h.knn_rows(dec, len(data), 3)
want = h.knn_rows(rp.knn_edges(hidden, data, 3), len(data), 3)
res = h.compare_knn(hidden, data, rows, want)
identical = int((rows == want).all(axis=1).sum())
print(f"decoder (KNNEdges k=3): {res['rows']} rows compared in {time.perf_counter() - t0:.0f} s; {identical} identical "
      f"entry for entry; {res['tie_rows']} differ by exact float64 ties only; {len(res['bad_rows'])} bad", flush=True)
failed |= bool(res["bad_rows"])

t0 = time.perf_counter()
enc = graph[("data", "to", "hidden")].edge_index.cpu().numpy()
radius = rp.cutoff_radius(hidden, 0.6)
want_enc = rp.radius_edges(data, hidden, radius)
res = h.compare_edge_sets(data, hidden, enc, want_enc, radius)
print(f"encoder (CutOffEdges 0.6, radius {radius:.9f}): {res['edges_a']} edges here, {res['edges_b']} in the reference, "
      f"compared in {time.perf_counter() - t0:.0f} s; {res['boundary']} on the radius boundary; {res['bad']} bad",
      flush=True)
failed |= res["bad"] != 0
same_order = np.array_equal(enc, h.canonical(enc))
print(f"encoder rows in canonical order (target, then source): {same_order}")
failed |= not same_order

proc = graph[("hidden", "to", "hidden")].edge_index.cpu().numpy()
order = rp.tri_nodes(resolution)[1]
want_proc = rp.multiscale_edges_one_hop(list(range(resolution + 1)), order)
ok = np.array_equal(proc, want_proc)
print(f"processor (MultiScaleEdges x_hops=1): {proc.shape[1]} edges, equal to the oracle's: {ok}")
failed |= not ok
print("FAILED" if failed else "ALL ROWS AGREE")
sys.exit(1 if failed else 0)
