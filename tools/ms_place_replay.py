"""Host replay of the placement scheme of ag_multiscale_complete on the oracle's icospheres: every row slot is
written exactly once and the rows hold the symmetric closure of the mesh edges, for several level subsets.
python tools/ms_place_replay.py"""
import sys
import numpy as np
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[1]))
from oracle import icosphere as ico

def level_edge_arrays(R):
    """per-level (min,max) arrays in the order the GPU library stores them"""
    v, f = ico.icosahedron()
    faces = [f]
    for l in range(R):
        v, fn = ico.subdivide(v, faces[-1]); faces.append(fn)
    out = []
    for l in range(R + 1):
        e, inv = ico.unique_edges(faces[l])  # sorted by (max,min); e[:,0]=min, e[:,1]=max
        if l < R or R == 0:
            out.append(e.copy())
        else:
            ep, invp = ico.unique_edges(faces[l - 1])
            Vp = 10 * 4 ** (l - 1) + 2
            split = []
            for i, (a, b) in enumerate(ep):
                m = Vp + i
                split.append((a, m)); split.append((b, m))
            mid = invp.reshape(-1, 3) + Vp
            face = []
            for (m0, m1, m2) in mid:
                for x, y in ((m0, m1), (m1, m2), (m2, m0)):
                    face.append((min(x, y), max(x, y)))
            arr = np.array(split + face, dtype=np.int64)
            # same set as the sorted unique edges
            assert set(map(tuple, arr)) == set(map(tuple, e)) and len(arr) == len(e)
            out.append(arr)
    return out

def emulate(R, levels):
    E = level_edge_arrays(R)
    V = 10 * 4 ** R + 2
    size = np.zeros(V, dtype=np.int64)
    for v in range(V):
        alive = sum(1 for l in levels if 10 * 4 ** l + 2 > v)
        size[v] = (5 if v < 12 else 6) * alive
    start = np.concatenate([[0], np.cumsum(size)])
    out_src = -np.ones(start[-1], dtype=np.int64); out_tgt = -np.ones(start[-1], dtype=np.int64)
    front = np.zeros(V, dtype=np.int64); back = np.zeros(V, dtype=np.int64)
    def put(pos, s, t):
        assert start[t] <= pos < start[t + 1], (pos, t)
        assert out_src[pos] == -1, ("slot written twice", pos, s, t)
        out_src[pos] = s; out_tgt[pos] = t
    for l in levels:
        e = E[l]
        final = (l == R and R > 0)
        n_split = 2 * 30 * 4 ** (R - 1) if final else 0
        for i, (vmin, vmax) in enumerate(e):
            if not final:
                before = 0
                for d in range(1, 6):
                    if i - d >= 0 and e[i - d][1] == vmax and before == d - 1: before += 1
                    else: break
                fslot = before
            elif i < n_split:
                fslot = i & 1
            else:
                fslot = 2 + front[vmax]; front[vmax] += 1
            put(start[vmax] + fslot, vmin, vmax)
            put(start[vmin + 1] - 1 - back[vmin], vmax, vmin); back[vmin] += 1
    assert (out_src >= 0).all(), "unwritten slots"
    # compare with the symmetric closure
    want = set()
    for l in levels:
        for a, b in E[l]:
            want.add((a, b)); want.add((b, a))
    got = set(zip(out_src.tolist(), out_tgt.tolist()))
    assert got == want and len(want) == len(out_src)
    return len(out_src)

def level_edge_arrays_vec(R):
    """level_edge_arrays without Python loops (the r6 / r7 meshes)"""
    v, f = ico.icosahedron()
    faces = [f]
    for l in range(R):
        v, fn = ico.subdivide(v, faces[-1]); faces.append(fn)
    out = []
    for l in range(R + 1):
        e, inv = ico.unique_edges(faces[l])
        if l < R or R == 0:
            out.append(e.astype(np.int64))
            continue
        ep, invp = ico.unique_edges(faces[l - 1])
        Vp = 10 * 4 ** (l - 1) + 2
        m = Vp + np.arange(len(ep), dtype=np.int64)
        split = np.empty((2 * len(ep), 2), dtype=np.int64)
        split[0::2, 0] = ep[:, 0]; split[0::2, 1] = m
        split[1::2, 0] = ep[:, 1]; split[1::2, 1] = m
        mid = invp.reshape(-1, 3).astype(np.int64) + Vp
        sides = np.stack([mid[:, [0, 1]], mid[:, [1, 2]], mid[:, [2, 0]]], axis=1).reshape(-1, 2)
        arr = np.concatenate([split, np.sort(sides, axis=1)])
        assert np.array_equal(np.unique(arr, axis=0), np.unique(e.astype(np.int64), axis=0)) and len(arr) == len(e)
        out.append(arr)
    return out

def _cumcount(keys):
    """for every element, the number of earlier elements with the same key"""
    order = np.argsort(keys, kind="stable")
    ks = keys[order]
    first = np.concatenate([[True], ks[1:] != ks[:-1]])
    run_start = np.maximum.accumulate(np.where(first, np.arange(len(ks)), 0))
    out = np.empty(len(keys), dtype=np.int64)
    out[order] = np.arange(len(ks)) - run_start
    return out

def emulate_vec(R, levels, rank=None):
    """emulate() in numpy, rows in node order (``rank``: vertex -> node, default identity): positions of all
    entries, each inside its row and written once, every slot filled."""
    E = level_edge_arrays_vec(R)
    V = 10 * 4 ** R + 2
    rank = np.arange(V) if rank is None else np.asarray(rank, dtype=np.int64)
    vid = np.arange(V)
    alive = np.zeros(V, dtype=np.int64)
    for l in levels:
        alive += (10 * 4 ** l + 2 > vid)
    size = np.empty(V, dtype=np.int64)
    size[rank] = np.where(vid < 12, 5, 6) * alive
    start = np.concatenate([[0], np.cumsum(size)])
    pos_all, back_keys, back_rows = [], [], []
    for l in levels:
        e = E[l]
        vmin, vmax = e[:, 0], e[:, 1]
        assert (vmin < vmax).all()
        final = (l == R and R > 0)
        idx = np.arange(len(e))
        if not final:
            first = np.concatenate([[True], vmax[1:] != vmax[:-1]])
            fslot = idx - np.maximum.accumulate(np.where(first, idx, 0))
            assert fslot.max() <= 5
        else:
            n_split = 2 * 30 * 4 ** (R - 1)
            fslot = np.where(idx < n_split, idx & 1, 0)
            fslot[n_split:] = 2 + _cumcount(vmax[n_split:])
        front = start[rank[vmax]] + fslot
        assert (front < start[rank[vmax] + 1]).all(), "front slot beyond its row"
        pos_all.append(front)
        back_keys.append(vmin)
    vmin_all = np.concatenate(back_keys)
    back = start[rank[vmin_all] + 1] - 1 - _cumcount(vmin_all)
    assert (back >= start[rank[vmin_all]]).all(), "back slot before its row"
    pos = np.concatenate(pos_all + [back])
    assert len(pos) == start[-1], "entries != slots"
    assert len(np.unique(pos)) == len(pos), "slot written twice"
    return len(pos)

CASES = [(0, [0]), (1, [0, 1]), (1, [1]), (2, [0, 1, 2]), (2, [0, 2]), (3, [0, 1, 2, 3]), (3, [1, 3]), (3, [2]),
         (4, [0, 1, 2, 3, 4]), (4, [4]), (4, [0, 3])]

LARGE_CASES = [(5, [0, 1, 2, 3, 4, 5]), (6, [0, 1, 2, 3, 4, 5, 6]), (6, [2, 6]), (7, list(range(8))), (7, [7]),
               (7, [0, 3, 7]), (7, [1, 2, 3, 4, 5, 6])]

if __name__ == "__main__":
    for R, levels in CASES:
        print(R, levels, emulate(R, levels), emulate_vec(R, levels))
    for R, levels in LARGE_CASES:
        print(R, levels, emulate_vec(R, levels))
