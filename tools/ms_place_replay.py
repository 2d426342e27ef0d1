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

CASES = [(0, [0]), (1, [0, 1]), (1, [1]), (2, [0, 1, 2]), (2, [0, 2]), (3, [0, 1, 2, 3]), (3, [1, 3]), (3, [2]),
         (4, [0, 1, 2, 3, 4]), (4, [4]), (4, [0, 3])]

if __name__ == "__main__":
    for R, levels in CASES:
        print(R, levels, emulate(R, levels))
