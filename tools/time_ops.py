"""Developer tool: per-operation device timings of the O1280 -> r7 workload."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from anemoi_graphs_amd import grids, ops, profiling

dev = torch.device("cuda", 0)
name = sys.argv[1] if len(sys.argv) > 1 else "o1280"
res = int(sys.argv[2]) if len(sys.argv) > 2 else 7
data = torch.from_numpy(grids.named(name)).to(dev)
_, _, hidden, order, rank, level_edges = ops.icosphere(res, dev)

def timed(label, fn, n=5):
    fn(); torch.cuda.synchronize()
    profiling.enable(True)
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n * 1e3
    k = profiling.collect(); profiling.enable(False)
    top = sorted(k.items(), key=lambda kv: -kv[1][1])[:6]
    print(f"{label:28s} {dt:8.3f} ms/call  " + "  ".join(f"{a}:{b[1]/n:.3f}" for a, b in top), flush=True)
    return out

ids, stats = timed("knn dec k=3", lambda: ops.knn(hidden, data, 3, return_stats=True))
print("   knn stats (exact-path targets, extra rounds):", stats.tolist())
timed("nn_maxdist hidden", lambda: ops.nn_maxdist(hidden))
r = float(ops.nn_maxdist(hidden).item()) * 0.6
enc = timed("radius enc", lambda: ops.radius_edges(data, hidden, r))
proc = timed("multiscale", lambda: ops.multiscale_edges(res, list(range(res + 1)), 1, rank, level_edges))
timed("icosphere", lambda: ops.icosphere(res, dev))
dec = torch.stack([ids.reshape(-1), torch.arange(data.shape[0], dtype=torch.int32, device=dev).repeat_interleave(3)])
timed("attrs dec", lambda: ops.edge_attrs(hidden, data, dec))
timed("attrs dec len only", lambda: ops.edge_attrs(hidden, data, dec, want_dir=False))
timed("attrs enc", lambda: ops.edge_attrs(data, hidden, enc))
timed("knn enc-dir k=3 (src=data)", lambda: ops.knn(data, hidden, 3))
