"""Randomised comparison of EdgeLength / EdgeDirection with the oracle on uneven node sets and arbitrary
edge lists (unsorted targets, long and short runs, poles, the date line, long edges) -- GPU.
python tools/fuzz_attrs.py [cases] [first_seed]"""
import importlib.util
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from anemoi_graphs_amd import ops  # noqa: E402
from oracle import reference_path as rp  # noqa: E402

_spec = importlib.util.spec_from_file_location("fuzz_parity", Path(__file__).resolve().parent / "fuzz_parity.py")
_fz = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_fz)
dev = torch.device("cuda", 0)
RTOL = 1e-6


def run(cases: int, seed0: int = 0) -> int:
    bad = 0
    t0 = time.time()
    for seed in range(seed0, seed0 + cases):
        rng = np.random.default_rng(seed)
        ns = int(rng.choice([1, 3, 40, 900, 20000]))
        nt = int(rng.choice([1, 2, 60, 1500, 30000]))
        src, tgt = _fz.points(rng, ns), _fz.points(rng, nt)
        ne = int(rng.choice([1, 5, 127, 128, 129, 1000, 50001]))
        kind = rng.integers(0, 3)
        if kind == 0:  # arbitrary pairs, arbitrary order
            e = np.stack([rng.integers(0, ns, ne), rng.integers(0, nt, ne)])
        elif kind == 1:  # grouped by target, as the builders emit them; runs of random length
            t = np.sort(rng.integers(0, nt, ne))
            e = np.stack([rng.integers(0, ns, ne), t])
        else:  # nearest sources (short edges), when there are enough nodes
            k = int(min(ns, rng.choice([1, 3, 7])))
            e = rp.knn_edges(src, tgt, k)[:, :ne]
        e = np.ascontiguousarray(e.astype(np.int32))
        rotated = bool(rng.random() < 0.8)
        tag = f"seed {seed} ns {ns} nt {nt} E {e.shape[1]} kind {kind} rotated {rotated}"
        try:
            length, direction = ops.edge_attrs(torch.from_numpy(src).to(dev), torch.from_numpy(tgt).to(dev),
                                               torch.from_numpy(e).to(dev), rotated=rotated)
            length, direction = length.cpu().numpy(), direction.cpu().numpy()
            want_len = rp.edge_lengths(src, tgt, e)
            want_dir = rp.edge_directions(src, tgt, e, rotated)
            # a self edge (or coincident end points) has no direction: the reference's value there is the
            # rounding noise of its float64 rotation; edges to an exactly antipodal node likewise
            ok_dir = want_len > 1e-6
            ok_dir &= want_len < np.pi - 1e-3
            # the relative-to-target rotation is ill-conditioned when the target sits on a pole
            ok_dir &= np.abs(np.abs(tgt[e[1], 0]) - np.pi / 2) > 1e-6
            err_l = np.abs(length - want_len) - RTOL * np.abs(want_len) - 1e-9
            scale = 1.0 if rotated else max(1.0, float(np.abs(want_dir).max()))
            err_d = (np.abs(direction - want_dir) - RTOL * np.abs(want_dir) - 2e-6 * scale)[ok_dir]
            if not (np.all(err_l <= 0) and np.all(err_d <= 0)):
                bad += 1
                i = int(np.argmax(err_l))
                print("ATTR MISMATCH", tag, "len worst", float(err_l.max()), length[i], want_len[i], "dir worst",
                      float(err_d.max()) if err_d.size else None, flush=True)
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
