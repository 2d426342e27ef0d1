"""Randomised parity sweep of the k-NN search on DENSE target sets (tens of targets per source cell, the
decoder's regime: stage 1 by ranges of 64 binned targets) at medium size, every row compared with the oracle --
uneven target densities, bands, polar caps, regular latitude-longitude grids with coincident pole rows (GPU).
python tools/fuzz_dense.py [radius] [cases] [first_seed]      (radius: the cut-off search with dense sources instead)"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from anemoi_graphs_amd import grids, ops  # noqa: E402
from oracle import compare as cmp  # noqa: E402
from oracle import reference_path as rp  # noqa: E402

dev = torch.device("cuda", 0)


def targets(rng, n):
    kind = rng.integers(0, 5)
    if kind == 0:  # uniform
        lat, lon = np.arcsin(rng.uniform(-1, 1, n)), rng.uniform(0, 2 * np.pi, n)
    elif kind == 1:  # density growing towards one pole
        lat = np.pi / 2 - np.abs(rng.normal(0, rng.uniform(0.05, 0.8), n)) % np.pi
        lon = rng.uniform(-np.pi, np.pi, n)
    elif kind == 2:  # a few latitude bands of very different populations
        centres = rng.uniform(-1.5, 1.5, 4)
        which = rng.choice(4, n, p=[0.7, 0.2, 0.09, 0.01])
        lat = np.clip(centres[which] + rng.uniform(-0.02, 0.02, n), -np.pi / 2, np.pi / 2)
        lon = rng.uniform(0, 2 * np.pi, n)
    elif kind == 3:  # regular latitude-longitude grid, pole rows of coincident points
        step = float(rng.choice([0.5, 0.75, 1.0]))
        g = grids.regular(step)
        return g[rng.permutation(len(g))[: min(n, len(g))]] if rng.random() < 0.5 else g
    else:  # regional patch across the date line plus a sparse background
        m = n // 10
        lat = np.concatenate([rng.uniform(0.2, 0.9, n - m), np.arcsin(rng.uniform(-1, 1, m))])
        lon = np.concatenate([rng.uniform(-0.4, 0.4, n - m) % (2 * np.pi), rng.uniform(0, 2 * np.pi, m)])
    return np.stack([np.clip(lat, -np.pi / 2, np.pi / 2), lon], 1).astype(np.float32)


def run(cases: int, seed0: int = 0) -> int:
    bad = 0
    t0 = time.time()
    for seed in range(seed0, seed0 + cases):
        rng = np.random.default_rng(seed)
        ns = int(rng.choice([200, 642, 2562, 10242]))
        nt = int(rng.choice([60_000, 150_000, 300_000]))
        src = grids.fibonacci_sphere(ns) if rng.random() < 0.5 else grids.random_sphere(ns, seed=seed)
        tgt = targets(rng, nt)
        k = int(rng.choice([1, 3, 3, 5, 10]))
        tag = f"seed {seed} ns {ns} nt {len(tgt)} k {k}"
        try:
            got = ops.knn(torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev), k).cpu().numpy()
            want = cmp.knn_rows(rp.knn_edges(src, tgt, k), len(tgt), k)
            res = cmp.compare_knn(src, tgt, got, want)
            if res["bad_rows"]:
                bad += 1
                print("KNN MISMATCH", tag, {**res, "bad_rows": res["bad_rows"][:5]}, flush=True)
        except Exception as exc:  # noqa: BLE001
            bad += 1
            print("EXCEPTION", tag, type(exc).__name__, exc, flush=True)
        if (seed - seed0) % 5 == 4:
            print(f"... {seed - seed0 + 1} cases, {bad} bad, {time.time() - t0:.0f} s", flush=True)
    print(f"done: {cases} cases, {bad} bad", flush=True)
    return bad


def run_radius(cases: int, seed0: int = 0) -> int:
    """The encoder's regime: dense, uneven SOURCES around a sparse quasi-uniform target set, radius a multiple of
    the targets' reference distance -- rows from empty to thousands of entries (stash, long-row path)."""
    bad = 0
    t0 = time.time()
    for seed in range(seed0, seed0 + cases):
        rng = np.random.default_rng(seed)
        nt = int(rng.choice([162, 642, 2562]))
        ns = int(rng.choice([60_000, 150_000, 300_000]))
        tgt = grids.fibonacci_sphere(nt) if rng.random() < 0.5 else grids.random_sphere(nt, seed=seed)
        src = targets(rng, ns)
        factor = float(rng.choice([0.3, 0.6, 0.6, 1.0, 2.0]))
        tag = f"seed {seed} ns {len(src)} nt {nt} factor {factor}"
        try:
            ds, dt = torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev)
            radius = float(ops.nn_maxdist(dt).item()) * factor
            want_radius = rp.cutoff_radius(tgt, factor)
            if abs(radius - want_radius) > 1e-12 * want_radius:
                bad += 1
                print("RADIUS VALUE MISMATCH", tag, radius, want_radius, flush=True)
                continue
            e = ops.radius_edges(ds, dt, want_radius).cpu().numpy()
            w = rp.canonical_sort(rp.radius_edges(src, tgt, want_radius))
            if e.shape != w.shape or not (e == w).all():
                r2 = cmp.compare_edge_sets(src, tgt, e, w, want_radius)
                if r2["bad"]:
                    bad += 1
                    print("RADIUS MISMATCH", tag, r2, flush=True)
            rows = np.bincount(e[1], minlength=nt)
            order_ok = np.all(np.diff(e[1]) >= 0) and np.all((np.diff(e[0]) > 0) | (np.diff(e[1]) > 0))
            if not order_ok:
                bad += 1
                print("ORDER MISMATCH", tag, flush=True)
            if (seed - seed0) % 5 == 4:
                print(f"... {seed - seed0 + 1} cases, {bad} bad, longest row {rows.max()}, {time.time() - t0:.0f} s", flush=True)
        except Exception as exc:  # noqa: BLE001
            bad += 1
            print("EXCEPTION", tag, type(exc).__name__, exc, flush=True)
    print(f"done: {cases} cases, {bad} bad", flush=True)
    return bad


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "radius"]
    n_cases = int(args[0]) if len(args) > 0 else 20
    first = int(args[1]) if len(args) > 1 else 0
    sweep = run_radius if "radius" in sys.argv[1:] else run
    sys.exit(1 if sweep(n_cases, first) else 0)
