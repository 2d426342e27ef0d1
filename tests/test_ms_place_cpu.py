"""CPU: the placement scheme of ag_multiscale_complete (front block read off the edge order, back block by a
counter, the finest level stored as ag_icosphere leaves it) replayed on the oracle's icospheres -- every row slot
is written exactly once and the rows hold both directions of every mesh edge (tri_icosahedron.py:165-188)."""
import importlib.util
from pathlib import Path

import pytest

_spec = importlib.util.spec_from_file_location("ms_place_replay", Path(__file__).resolve().parents[1] / "tools" / "ms_place_replay.py")
replay = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(replay)


@pytest.mark.parametrize("resolution,levels", [c for c in replay.CASES if c[0] <= 3])
def test_placement_fills_every_row_slot_once(resolution, levels):
    assert replay.emulate(resolution, levels) == sum(60 * 4 ** l for l in levels)


@pytest.mark.parametrize("resolution,levels", replay.CASES)
def test_vectorised_replay_agrees_with_the_loop_replay(resolution, levels):
    want = sum(60 * 4 ** l for l in levels)
    assert replay.emulate_vec(resolution, levels) == want
    if resolution <= 3:
        assert replay.emulate(resolution, levels) == want


@pytest.mark.parametrize("resolution,levels", replay.LARGE_CASES)
def test_placement_on_the_benchmark_meshes(resolution, levels):
    """r6 and r7 (the headline mesh), all levels and subsets, rows in the node order of the oracle's mesh."""
    import numpy as np

    from oracle import reference_path as rp

    order = rp.tri_nodes(resolution)[1]
    rank = np.empty(len(order), dtype=np.int64)
    rank[order] = np.arange(len(order))
    assert replay.emulate_vec(resolution, levels, rank) == sum(60 * 4 ** l for l in levels)
