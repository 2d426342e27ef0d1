"""CPU: the host logic of bench.py that needs no GPU -- the launcher of a multi-GPU run and the bookkeeping of
the roofline object."""
import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

import bench  # noqa: E402


def test_importing_bench_loads_no_gpu_library():
    """The process that starts the ranks of ``--gpus N`` must not have touched a GPU: importing bench.py (and
    parsing its arguments) loads neither torch nor the HIP library."""
    code = ("import sys; sys.argv=['bench.py']; import bench; bench.parse_args(['--gpus','4']); "
            "assert 'torch' not in sys.modules and 'anemoi_graphs_amd' not in sys.modules")
    proc = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True)
    assert proc.returncode == 0, proc.stderr


def test_launcher_fails_loudly_without_gpus(tmp_path):
    """``bench.py --gpus 2`` with no launcher around it starts its own ranks; on a machine without GPUs the ranks
    refuse (one rank per GPU) and the parent reports their failure: non-zero exit code, no result line."""
    import os

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "AG_BENCH_ONE_DEVICE")}
    proc = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--workload", "o96_r4", "--steps", "1",
                           "--warmup", "0"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    import torch

    if torch.cuda.device_count() >= 2:
        return  # a real multi-GPU machine: covered by the GPU tests
    assert proc.returncode != 0
    assert proc.stdout.strip() == ""
    assert "no GPU of its own" in proc.stderr or "ended with code" in proc.stderr


def test_rank_count_must_match():
    """A rank started by a launcher with another world size than ``--gpus`` stops before it does anything."""
    import os

    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    proc = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2"], cwd=ROOT, env=env, capture_output=True,
                          text=True, timeout=600)
    assert proc.returncode != 0 and "--gpus 2" in proc.stderr


def test_traffic_table_takes_the_newest_round_only(tmp_path, monkeypatch):
    prof = tmp_path / "profiles"
    prof.mkdir()
    entry = {"kernels": {"ag_edge_attrs_kernel": {"hbm_bytes_per_launch": 10}}}
    (prof / "r01_hbm_traffic.json").write_text(json.dumps({"o1280_r7": entry}))
    (prof / "r02_hbm_traffic.json").write_text(json.dumps(
        {"o1280_r7_one_stream": {"kernels": {"ag_edge_attrs_kernel": {"hbm_bytes_per_launch": 20}}}}))
    monkeypatch.setattr(bench, "ROOT", tmp_path)
    table, src = bench.traffic_table("o1280_r7")
    assert table == {"ag_edge_attrs_kernel": 20.0} and src == "r02_hbm_traffic.json:o1280_r7_one_stream"
    # a newer summary that does not cover the workload: no traffic is quoted, the older file is not consulted
    (prof / "r03_hbm_traffic.json").write_text(json.dumps({"era5_r6": entry}))
    assert bench.traffic_table("o1280_r7") == ({}, None)
    # round numbers order numerically
    (prof / "r10_hbm_traffic.json").write_text(json.dumps({"o1280_r7": entry}))
    assert bench.traffic_table("o1280_r7")[1] == "r10_hbm_traffic.json:o1280_r7"


def test_algorithmic_bytes_follow_the_stated_units():
    sizes = {"n_data": 1000, "n_hidden": 100, "e_enc": 5000, "e_dec": 3000, "e_proc": 600, "e_total": 8600}
    assert bench.algorithmic_bytes("ag_knn_cell_kernel", sizes) == 8 * 100 + 8 * 1000 + 8 * 3000
    assert bench.algorithmic_bytes("ag_radius_grouped_kernel", sizes) == 8 * 1000 + 8 * 100 + 8 * 100
    assert bench.algorithmic_bytes("ag_edge_attrs_kernel", sizes) == 36 * 8600 / 3
    rows = bench.kernel_table({"ag_edge_attrs_kernel": (3, 0.3), "ag_x": (1, 0.5), "rocprim": (1, 9.0)}, sizes,
                              {"ag_edge_attrs_kernel": 2.0 * 36 * 8600 / 3})
    assert list(rows) == ["ag_x", "ag_edge_attrs_kernel"]
    assert rows["ag_edge_attrs_kernel"]["traffic_ratio"] == 2.0 and rows["ag_x"]["algorithmic_bytes"] is None
