"""GPU: randomised parity sweep of k-NN, radius query and reference distance against the oracle on uneven node
sets (clusters, latitude bands, polar caps, the date line; 1 to 20,000 sources) -- tools/fuzz_parity.py, which
also runs longer sweeps from the command line."""
import importlib.util
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu


def test_randomised_parity_sweep(device):
    spec = importlib.util.spec_from_file_location("fuzz_parity", Path(__file__).resolve().parents[1] / "tools" / "fuzz_parity.py")
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    assert fuzz.run(80, seed0=1000) == 0
