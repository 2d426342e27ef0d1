"""Randomised parity sweep of the k-NN search on DENSE target sets (tens of targets per source cell, the
decoder's regime: stage 1 by ranges of 64 binned targets) at medium size, every row compared with the oracle --
uneven target densities, bands, polar caps, regular latitude-longitude grids with coincident pole rows (GPU).
python tools/fuzz_dense.py [cases] [first_seed]"""
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


if __name__ == "__main__":
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    sys.exit(1 if run(n_cases, first) else 0)
