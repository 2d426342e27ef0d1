"""GPU: randomised parity sweeps against the oracle on uneven node sets (clusters, latitude bands, polar caps, the
date line): k-NN, radius query and reference distance (tools/fuzz_parity.py), spherical Voronoi cell areas
(tools/fuzz_voronoi.py), edge attributes (tools/fuzz_attrs.py), meshes and multi-scale edges (tools/fuzz_mesh.py), k-NN on dense target sets (tools/fuzz_dense.py).  The tools also run longer sweeps from the command line."""
import importlib.util
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu


def _tool(name):
    spec = importlib.util.spec_from_file_location(name, Path(__file__).resolve().parents[1] / "tools" / f"{name}.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_randomised_parity_sweep(device):
    assert _tool("fuzz_parity").run(80, seed0=1000) == 0


def test_randomised_voronoi_sweep(device):
    """Cell areas against scipy on uneven node sets: agreement to 1e-7 wherever scipy's own result is sound,
    the same kind of error where both refuse, a clear refusal for cells that span a hemisphere."""
    assert _tool("fuzz_voronoi").run(120, seed0=2000) == 0


def test_randomised_attribute_sweep(device):
    """EdgeLength / EdgeDirection on arbitrary edge lists (unsorted targets, runs of any length, poles, long edges)."""
    assert _tool("fuzz_attrs").run(60, seed0=3000) == 0


def test_randomised_mesh_sweep(device):
    """TriNodes / LimitedAreaTriNodes / StretchedTriNodes and their multi-scale edges for random regions,
    margins, resolutions and hop counts."""
    assert _tool("fuzz_mesh").run(40, seed0=4000) == 0


def test_randomised_primitive_sweep(device):
    """unique / sort / compact / gather / remap / mark / invert against numpy; masked KNNEdges and
    RemoveUnconnectedNodes against the oracle."""
    assert _tool("fuzz_ops").run(60, seed0=5000) == 0


def test_randomised_dense_target_sweep(device):
    """k-NN on dense target sets (the decoder's regime, stage 1 by ranges of 64 binned targets): uneven
    densities, bands, polar caps, regular grids with coincident pole rows -- every row against the oracle."""
    assert _tool("fuzz_dense").run(6, seed0=6000) == 0


def test_randomised_dense_source_radius_sweep(device):
    """Cut-off search in the encoder's regime: dense uneven sources around a sparse target set, rows from empty
    to tens of thousands of entries, exact edge lists and (target, source) order against the oracle."""
    assert _tool("fuzz_dense").run_radius(8, seed0=7000) == 0
