"""Randomised comparison of the spherical Voronoi cell areas with scipy (the reference's path) on uneven
node sets (GPU).  python tools/fuzz_voronoi.py [cases] [first_seed]"""
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


def run(cases: int, seed0: int = 0) -> int:
    bad = 0
    unsupported = 0
    t0 = time.time()
    for seed in range(seed0, seed0 + cases):
        rng = np.random.default_rng(seed)
        n = int(rng.choice([4, 6, 9, 20, 100, 800, 5000]))
        x = _fz.points(rng, n)
        if rng.random() < 0.5:  # a background of uniform nodes so that every cell is bounded by near nodes
            x = np.concatenate([x, _fz.points(np.random.default_rng(seed + 7), max(8, n // 4))[:max(8, n // 4)]])
        tag = f"seed {seed} n {len(x)}"
        want = err_w = None
        try:
            want = rp.spherical_area_weights(x, raw=True)
        except Exception as exc:  # noqa: BLE001
            err_w = f"{type(exc).__name__}: {exc}"
        got = err_g = None
        try:
            got = ops.voronoi_areas(torch.from_numpy(x).to(dev)).cpu().numpy()
        except Exception as exc:  # noqa: BLE001
            err_g = f"{type(exc).__name__}: {exc}"
        if err_g and "not supported" in err_g:
            # a cell spanning (nearly) a hemisphere: refused with a clear error (the sets drawn here are
            # mostly regional, so their rim cells run to the antipode)
            unsupported += 1
            continue
        unreliable = (err_w is not None and not err_w.startswith("ValueError")) or (
            want is not None and (abs(want.sum() - 4 * np.pi) > 1e-5 or (want == 0).any()))
        if unreliable:
            # the reference itself breaks down (qhull drops generators closer than the float32 noise of the
            # unit vectors allows, regions come back empty or mis-sorted, or scipy raises an IndexError):
            # nothing to compare with; the cells here must still roughly tile the sphere if they were built
            if got is not None and abs(got.sum() - 4 * np.pi) > 1e-2:
                bad += 1
                print("SUM MISMATCH (reference unreliable)", tag, got.sum(), flush=True)
            continue
        if err_w or err_g:
            # both must refuse, with the same kind of error (duplicates, degenerate sets)
            same = err_w is not None and err_g is not None and err_w.split(":")[0] == err_g.split(":")[0]
            if not same:
                bad += 1
                print("ERROR MISMATCH", tag, "| reference:", err_w, "| here:", err_g, flush=True)
            continue
        rel = np.abs(got - want) / np.maximum(np.abs(want), 1e-300)
        if not np.all(rel < 1e-7):
            bad += 1
            i = int(np.argmax(rel))
            print("AREA MISMATCH", tag, "max rel", rel.max(), "at", i, got[i], want[i], x[i], flush=True)
        if (seed - seed0) % 20 == 19:
            print(f"... {seed - seed0 + 1} cases, {bad} bad, {time.time() - t0:.0f} s", flush=True)
    print(f"done: {cases} cases, {bad} bad, {unsupported} refused as unsupported", flush=True)
    return bad


if __name__ == "__main__":
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    sys.exit(1 if run(n_cases, first) else 0)
