"""Randomised comparison of the limited-area / stretched meshes and their multi-scale edges, and of the
complete icosphere's edges, with the oracle (GPU).  python tools/fuzz_mesh.py [cases] [first_seed]"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from anemoi_graphs_amd.edges import MultiScaleEdges  # noqa: E402
from anemoi_graphs_amd.hetero import HeteroData  # noqa: E402
from anemoi_graphs_amd.nodes import LimitedAreaTriNodes, StretchedTriNodes, TriNodes  # noqa: E402
from oracle import reference_path as rp  # noqa: E402


def canonical(e):
    e = np.asarray(e)
    return e[:, np.lexsort((e[0], e[1]))]


def area(rng):
    """reference nodes of a region: a disc of random size somewhere on the sphere"""
    n = int(rng.choice([1, 5, 60, 400]))
    lat0, lon0 = np.arcsin(rng.uniform(-1, 1)), rng.uniform(-np.pi, np.pi)
    spread = 10.0 ** rng.uniform(-2, -0.3)
    lat = np.clip(lat0 + rng.normal(0, spread, n), -np.pi / 2, np.pi / 2)
    lon = lon0 + rng.normal(0, spread, n)
    return np.stack([lat, lon], 1).astype(np.float32)


def run(cases: int, seed0: int = 0) -> int:
    bad = 0
    t0 = time.time()
    for seed in range(seed0, seed0 + cases):
        rng = np.random.default_rng(seed)
        kind = int(rng.integers(0, 3))
        hops = int(rng.choice([1, 1, 2, 3]))
        ref = area(rng)
        margin = float(10.0 ** rng.uniform(1.5, 3.7))  # 30 km .. 5000 km
        g = HeteroData()
        g["area"].x = torch.as_tensor(ref)
        g["area"].node_type = "LatLonNodes"
        tag = f"seed {seed} kind {kind} hops {hops} area {len(ref)} margin {margin:.0f} km"
        try:
            if kind == 0:
                r = int(rng.integers(0, 5))
                tag += f" tri r{r}"
                g = TriNodes(r, "h").update_graph(g, {})
                want_x, order = rp.tri_nodes(r)
                want_x = want_x.astype(np.float32)
                want_e = rp.multiscale_edges(list(range(r + 1)), hops, order)
            elif kind == 1:
                r = int(rng.integers(1, 6))
                tag += f" lam r{r}"
                want_x, order, coords = rp.limited_area_tri_nodes(r, ref, margin)
                if len(want_x) == 0:
                    continue  # no mesh node within the margin: nothing to build
                g = LimitedAreaTriNodes(r, "area", "h", margin_radius_km=margin).update_graph(g, {})
                want_e = rp.multiscale_edges(list(range(r + 1)), hops, order, node_coords=coords, mask_reference=ref,
                                             margin_radius_km=margin)
            else:
                rb, rl = int(rng.integers(0, 3)), int(rng.integers(3, 6))
                tag += f" stretched r{rb}/{rl}"
                want_x, order, coords = rp.stretched_tri_nodes(rb, rl, ref, margin)
                g = StretchedTriNodes(rb, rl, "h", "area", None, margin_radius_km=margin).update_graph(g, {})
                # (the reference keeps the level vertices within 1 km of a node of the stretched mesh,
                # edges/builder.py:422-432)
                want_e = rp.multiscale_edges(list(range(rl + 1)), hops, order, node_coords=coords, mask_reference=want_x,
                                             margin_radius_km=1.0)
            x = g["h"].x.cpu().numpy()
            if x.shape != want_x.shape or not np.array_equal(x, want_x):
                bad += 1
                print("NODE MISMATCH", tag, x.shape, want_x.shape, flush=True)
                continue
            g = MultiScaleEdges("h", "h", hops).update_graph(g)
            e = g[("h", "to", "h")].edge_index.cpu().numpy()
            if e.shape != want_e.shape or not np.array_equal(canonical(e), canonical(want_e)):
                bad += 1
                print("EDGE MISMATCH", tag, e.shape, np.asarray(want_e).shape, flush=True)
        except Exception as exc:  # noqa: BLE001
            bad += 1
            print("EXCEPTION", tag, type(exc).__name__, exc, flush=True)
        if (seed - seed0) % 20 == 19:
            print(f"... {seed - seed0 + 1} cases, {bad} bad, {time.time() - t0:.0f} s", flush=True)
    print(f"done: {cases} cases, {bad} bad", flush=True)
    return bad


if __name__ == "__main__":
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    sys.exit(1 if run(n_cases, first) else 0)
