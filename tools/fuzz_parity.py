"""Randomised parity sweep of the search kernels against the oracle (GPU): many small and medium
cases with uneven node distributions.  python tools/fuzz_parity.py [cases] [first_seed]"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from anemoi_graphs_amd import ops  # noqa: E402
from oracle import compare as cmp  # noqa: E402
from oracle import reference_path as rp  # noqa: E402

dev = torch.device("cuda", 0)


def points(rng, n):
    """uniform, clustered, banded or polar node sets (float32 radians)"""
    kind = rng.integers(0, 5)
    if kind == 0:
        lat = np.arcsin(rng.uniform(-1, 1, n))
        lon = rng.uniform(0, 2 * np.pi, n)
    elif kind == 1:  # clusters
        c = rng.integers(1, 6)
        centres = np.stack([np.arcsin(rng.uniform(-1, 1, c)), rng.uniform(0, 2 * np.pi, c)], 1)
        which = rng.integers(0, c, n)
        spread = 10.0 ** rng.uniform(-4, -0.5)
        lat = np.clip(centres[which, 0] + rng.normal(0, spread, n), -np.pi / 2, np.pi / 2)
        lon = (centres[which, 1] + rng.normal(0, spread, n)) % (2 * np.pi)
    elif kind == 2:  # a latitude band
        lat0 = rng.uniform(-1.4, 1.4)
        lat = np.clip(lat0 + rng.uniform(-0.05, 0.05, n), -np.pi / 2, np.pi / 2)
        lon = rng.uniform(-np.pi, np.pi, n)
    elif kind == 3:  # around a pole, longitudes in [-pi, pi]
        sign = rng.choice([-1.0, 1.0])
        lat = sign * (np.pi / 2 - np.abs(rng.normal(0, 0.02, n)))
        lon = rng.uniform(-np.pi, np.pi, n)
    else:  # across the date line
        lat = rng.uniform(-0.3, 0.3, n)
        lon = (rng.normal(0, 0.05, n)) % (2 * np.pi)
    return np.stack([lat, lon], 1).astype(np.float32)


def run(cases: int, seed0: int = 0) -> int:
    """Number of failing cases (each is printed)."""
    bad = 0
    t0 = time.time()
    for seed in range(seed0, seed0 + cases):
        rng = np.random.default_rng(seed)
        ns = int(rng.choice([1, 2, 5, 12, 40, 300, 3000, 20000]))
        nt = int(rng.choice([1, 3, 50, 700, 5000]))
        src, tgt = points(rng, ns), points(rng, nt)
        if rng.random() < 0.3:
            tgt = points(np.random.default_rng(seed + 10**6), nt)  # unrelated distribution
        k = int(min(ns, rng.choice([1, 2, 3, 5, 8, 11, 16, 30])))
        ds, dt = torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev)
        tag = f"seed {seed} ns {ns} nt {nt} k {k}"
        try:
            got = ops.knn(ds, dt, k).cpu().numpy()
            want = cmp.knn_rows(rp.knn_edges(src, tgt, k), nt, k)
            res = cmp.compare_knn(src, tgt, got, want)
            if res["bad_rows"]:
                bad += 1
                print("KNN MISMATCH", tag, res, flush=True)
            # a radius around the typical spacing of the sources, sometimes much larger
            scale = float(rng.choice([0.5, 2.0, 8.0])) * np.sqrt(4 * np.pi / max(ns, 1))
            # (beyond pi scikit-learn's threshold sin^2(r/2) is no longer monotonic; the library returns all pairs there)
            radius = float(min(scale, 3.1)) if rng.random() < 0.9 else float(rng.uniform(0, 3.1))
            e = ops.radius_edges(ds, dt, radius).cpu().numpy()
            w = rp.canonical_sort(rp.radius_edges(src, tgt, radius))
            if e.shape != w.shape or not (e == w).all():
                r2 = cmp.compare_edge_sets(src, tgt, e, w, radius)
                if r2["bad"]:
                    bad += 1
                    print("RADIUS MISMATCH", tag, "radius", radius, r2, flush=True)
            if ns >= 2:
                m = float(ops.nn_maxdist(ds).item())
                try:
                    wm = rp.grid_reference_distance(src)
                except ValueError:
                    wm = 0.0
                # (up to 11 nodes scikit-learn answers by brute force in float32: its value is float32-accurate)
                if abs(m - wm) > (1e-12 if ns > 11 else 1e-5) * max(wm, 1e-30):
                    bad += 1
                    print("MAXDIST MISMATCH", tag, m, wm, flush=True)
        except Exception as exc:  # noqa: BLE001
            bad += 1
            print("EXCEPTION", tag, type(exc).__name__, exc, flush=True)
        if (seed - seed0) % 25 == 24:
            print(f"... {seed - seed0 + 1} cases, {bad} bad, {time.time() - t0:.0f} s", flush=True)
    print(f"done: {cases} cases, {bad} bad", flush=True)
    return bad


if __name__ == "__main__":
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    sys.exit(1 if run(n_cases, first) else 0)
