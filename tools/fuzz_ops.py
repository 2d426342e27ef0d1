"""Randomised comparison of the edge-list and mask primitives (unique / sort / compact / gather / remap /
mark / invert, the masked builders and the post-processor) with numpy (GPU).
python tools/fuzz_ops.py [cases] [first_seed]"""
import importlib.util
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from anemoi_graphs_amd import ops  # noqa: E402
from anemoi_graphs_amd.edges import CutOffEdges, KNNEdges  # noqa: E402
from anemoi_graphs_amd.hetero import HeteroData  # noqa: E402
from anemoi_graphs_amd.processors import RemoveUnconnectedNodes  # noqa: E402
from oracle import reference_path as rp  # noqa: E402

_spec = importlib.util.spec_from_file_location("fuzz_parity", Path(__file__).resolve().parent / "fuzz_parity.py")
_fz = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_fz)
dev = torch.device("cuda", 0)


def run(cases: int, seed0: int = 0) -> int:
    bad = 0
    t0 = time.time()

    def check(ok, what, tag):
        nonlocal bad
        if not ok:
            bad += 1
            print("MISMATCH", what, tag, flush=True)

    for seed in range(seed0, seed0 + cases):
        rng = np.random.default_rng(seed)
        ns = int(rng.choice([1, 4, 70, 3000]))
        nt = int(rng.choice([1, 5, 90, 2500]))
        ne = int(rng.choice([0, 1, 17, 1000, 60000]))
        tag = f"seed {seed} ns {ns} nt {nt} E {ne}"
        try:
            e = np.stack([rng.integers(0, ns, ne), rng.integers(0, nt, ne)]).astype(np.int32)
            et = torch.from_numpy(e).to(dev)
            # concat_edges' unique: torch.unique(dim=1) order = sorted by (row 0, row 1)
            got = ops.unique_edges(et).cpu().numpy()
            want = np.unique(e, axis=1) if ne else e
            check(got.shape == want.shape and np.array_equal(got, want), "unique_edges", tag)
            for by_target in (True, False):
                s = ops.sort_edges_(et.clone().contiguous(), by_target).cpu().numpy()
                key = (e[0], e[1]) if by_target else (e[1], e[0])
                w = e[:, np.lexsort(key)]
                check(np.array_equal(s, w), f"sort_edges by_target={by_target}", tag)
            mask = rng.random(ns) < rng.choice([0.0, 0.3, 1.0])
            ids = ops.mask_ids(torch.from_numpy(mask), dev)
            check(np.array_equal(ids.cpu().numpy(), np.nonzero(mask)[0]), "mask_ids", tag)
            x = _fz.points(rng, ns)
            xt = torch.from_numpy(x).to(dev)
            check(np.array_equal(ops.gather_coords(xt, ids).cpu().numpy(), x[mask]), "gather_coords", tag)
            inv = ops.invert_ids(ids, ns).cpu().numpy()
            w = np.full(ns, -1, np.int32)
            w[np.nonzero(mask)[0]] = np.arange(mask.sum())
            check(np.array_equal(inv, w), "invert_ids", tag)
            marks = torch.zeros(ns, dtype=torch.uint8, device=dev)
            ops.mark_ids_(marks, et[0])
            w = np.zeros(ns, np.uint8)
            w[e[0]] = 1
            check(np.array_equal(marks.cpu().numpy(), w), "mark_ids", tag)
            if ids.numel():
                vals = torch.from_numpy(rng.integers(0, ids.numel(), 500).astype(np.int32)).to(dev)
                r = ops.remap_ids_(vals.clone(), ids).cpu().numpy()
                check(np.array_equal(r, np.nonzero(mask)[0][vals.cpu().numpy()]), "remap_ids", tag)

            # masked builders + post-processor against the oracle
            y = _fz.points(rng, nt)
            smask = rng.random(ns) < 0.6
            tmask = rng.random(nt) < 0.7
            # (a node set of one node is left out: the reference squeezes its mask to 0-d and numpy refuses np.where on it)
            if ns > 1 and nt > 1 and smask.sum() >= 3 and tmask.sum() >= 1:
                g = HeteroData()
                g["a"].x = torch.from_numpy(x)
                g["b"].x = torch.from_numpy(y)
                g["a"]["m"] = torch.from_numpy(smask[:, None])
                g["b"]["m"] = torch.from_numpy(tmask[:, None])
                k = int(min(3, smask.sum()))
                g = KNNEdges("a", "b", k, source_mask_attr_name="m", target_mask_attr_name="m").update_graph(g)
                got = g[("a", "to", "b")].edge_index.cpu().numpy()
                sub = rp.knn_edges(x[smask], y[tmask], k)
                want = rp.undo_masking(sub, smask, tmask)
                rows_g = got[0].reshape(-1, k)
                rows_w = want[0].reshape(-1, k)
                same = np.array_equal(got[1], want[1]) and np.array_equal(np.sort(rows_g, 1), np.sort(rows_w, 1))
                check(same, "masked KNNEdges", tag)
                g2 = RemoveUnconnectedNodes("a", save_mask_indices_to_attr="kept").update_graph(g)
                used = np.unique(got[0])
                check(g2["a"].x.shape[0] == len(used) and np.array_equal(g2["a"]["kept"].cpu().numpy().reshape(-1), used),
                      "RemoveUnconnectedNodes nodes", tag)
                e2 = g2[("a", "to", "b")].edge_index.cpu().numpy()
                check(np.array_equal(used[e2[0]], got[0]) and np.array_equal(e2[1], got[1]), "RemoveUnconnectedNodes edges", tag)
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
