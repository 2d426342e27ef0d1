"""Single-GPU capacity check: a ~1 km global data grid (5.1e8 nodes, generated on the device) against an
icosphere, cut-off encoder + k-NN decoder + attributes.  Not a test and not the headline benchmark;
prints sizes and times.

    python tools/capacity_check.py [--nodes 510000000] [--resolution 9]
"""
import argparse
import math
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from anemoi_graphs_amd import ops  # noqa: E402


def fibonacci_on_device(n: int, device, chunk: int = 1 << 26) -> torch.Tensor:
    """Spiral points, latitude descending (north -> south), float64 arithmetic, float32 result."""
    out = torch.empty((n, 2), dtype=torch.float32, device=device)
    golden = math.pi * (3.0 - math.sqrt(5.0))
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        i = torch.arange(lo, hi, dtype=torch.float64, device=device)
        out[lo:hi, 0] = torch.asin(1.0 - 2.0 * (i + 0.5) / n).to(torch.float32)
        out[lo:hi, 1] = torch.remainder(i * golden, 2.0 * math.pi).to(torch.float32)
    return out


def timed(label, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    print(f"{label}: {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=510_000_000)
    ap.add_argument("--resolution", type=int, default=9)
    ap.add_argument("--cutoff", type=float, default=0.6)
    ap.add_argument("--check", action="store_true", help="verify sampled rows (brute force / scikit-learn)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    data = timed("generate data nodes", lambda: fibonacci_on_device(args.nodes, dev))
    print(f"data nodes {data.shape[0]:,} ({data.numel() * 4 / 1e9:.1f} GB)", flush=True)
    hidden = timed("icosphere", lambda: ops.icosphere(args.resolution)[2])
    print(f"hidden nodes {hidden.shape[0]:,}", flush=True)
    radius = args.cutoff * float(timed("reference distance", lambda: ops.nn_maxdist(hidden)).item())
    print(f"cut-off radius {radius:.6f} rad = {radius * 6371:.2f} km", flush=True)
    enc = timed("cut-off encoder (radius query)", lambda: ops.radius_edges(data, hidden, radius))
    print(f"encoder edges {enc.shape[1]:,} ({enc.shape[1] / hidden.shape[0]:.1f} per hidden node)", flush=True)
    assert bool((enc[1][1:] >= enc[1][:-1]).all()), "rows must be target-major"
    la, da = timed("encoder attributes", lambda: ops.edge_attrs(data, hidden, enc))
    print("max encoder edge length / radius:", float(la.max()) / radius, flush=True)
    del la, da
    dec = timed("k-NN decoder (k=3)", lambda: ops.knn_edges(hidden, data, 3))
    print(f"decoder edges {dec.shape[1]:,}", flush=True)
    lb, db = timed("decoder attributes", lambda: ops.edge_attrs(hidden, data, dec))
    print("max decoder edge length (rad):", float(lb.max()), flush=True)
    print(f"peak device memory {torch.cuda.max_memory_allocated() / 1e9:.1f} GB", flush=True)
    if args.check:
        check(data, hidden, radius, enc, dec)


def check(data, hidden, radius, enc, dec):
    """Encoder rows of a few hidden nodes against a float64 brute force over ALL data nodes (torch, on the
    device); decoder rows of sampled data nodes against scikit-learn on the hidden nodes."""
    import numpy as np

    sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
    from oracle import compare as cmp
    from oracle import reference_path as rp

    rthr = math.sin(0.5 * radius) ** 2
    picks = [0, 1, hidden.shape[0] // 3, hidden.shape[0] // 2, hidden.shape[0] - 1]
    starts = torch.searchsorted(enc[1].contiguous(), torch.tensor(picks, dtype=torch.int32, device=enc.device))
    ends = torch.searchsorted(enc[1].contiguous(), torch.tensor(picks, dtype=torch.int32, device=enc.device), right=True)
    lat = data[:, 0].double()
    lon = data[:, 1].double()
    coslat = torch.cos(lat)
    for t, a, b in zip(picks, starts.tolist(), ends.tolist()):
        tl, to = float(hidden[t, 0]), float(hidden[t, 1])
        rd = torch.sin(0.5 * (tl - lat)) ** 2 + math.cos(tl) * coslat * torch.sin(0.5 * (to - lon)) ** 2
        want = torch.nonzero(rd <= rthr).squeeze(1).to(torch.int32)
        got = enc[0, a:b]
        ok = want.numel() == got.numel() and bool((want == got).all())
        if not ok:  # only pairs on the radius boundary may differ
            odd = np.setxor1d(want.cpu().numpy(), got.cpu().numpy())
            ok = bool(np.all(np.abs(rd[torch.as_tensor(odd, device=rd.device).long()].cpu().numpy() - rthr) <= 1e-12 * rthr))
        print(f"encoder row of hidden node {t}: {got.numel()} sources, brute force {'agrees' if ok else 'DIFFERS'}", flush=True)
        assert ok
    del lat, lon, coslat
    rng = np.random.default_rng(0)
    sample = np.sort(rng.choice(data.shape[0], 20000, replace=False))
    h_np = hidden.cpu().numpy()
    t_np = data[torch.as_tensor(sample, device=data.device)].cpu().numpy()
    want = cmp.knn_rows(rp.knn_edges(h_np, t_np, 3), len(sample), 3)
    rows = dec[0].reshape(-1, 3)[torch.as_tensor(sample, device=dec.device)].cpu().numpy()
    res = cmp.compare_knn(h_np, t_np, rows, want)
    print("decoder rows of 20000 sampled data nodes vs scikit-learn:", res, flush=True)
    assert res["bad_rows"] == []


if __name__ == "__main__":
    main()
