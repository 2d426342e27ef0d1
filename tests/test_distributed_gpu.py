"""GPU, two, three or eight processes sharing the one GPU over gloo (tensors staged through host memory), and one
process over RCCL with device tensors: the latitude-band sharded build with the HIP kernels yields exactly the
single-process graph."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from anemoi_graphs_amd import grids

pytestmark = pytest.mark.gpu

NAMES = (("data", "to", "hidden"), ("hidden", "to", "data"), ("hidden", "to", "hidden"))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _dump(g, path):
    out = {}
    for name in NAMES:
        key = "_".join(name)
        out[key] = g[name].edge_index.cpu().numpy()
        out[key + "_len"] = g[name]["edge_length"].cpu().numpy()
        out[key + "_dir"] = g[name]["edge_dirs"].cpu().numpy()
    np.savez(path, **out)


def _worker(rank, world, port, grid, resolution, factor, out_dir, backend="gloo", exchange=False):
    from anemoi_graphs_amd import settings
    from anemoi_graphs_amd.distributed import ShardedGraphBuilder

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if backend == "nccl":
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        data = torch.from_numpy(grids.named(grid)).to("cuda:0")
        b = ShardedGraphBuilder(data, resolution, cutoff_factor=factor, num_nearest_neighbours=3, x_hops=1, rank=rank,
                                world_size=world, dist=dist, edge_length_norm="unit-max", edge_direction_norm="unit-std",
                                exchange_halo=exchange)
        with settings.resident(True):
            g = b.build()
        _dump(g, os.path.join(out_dir, f"rank{rank}.npz"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("grid,resolution,factor,world,backend,exchange", [
    ("o96", 4, 0.6, 2, "gloo", False), ("o320", 5, 0.6, 3, "gloo", False), ("o96", 3, 3.0, 2, "gloo", False),
    ("o320", 5, 0.6, 8, "gloo", False),  # as many ranks as a node has GPUs
    # RCCL with device tensors: one rank only, two ranks cannot share the one GPU of the test machine
    ("o320", 5, 0.6, 1, "nccl", False),
    # coordinates treated as SHARDED (every rank searches its own band plus what its neighbours send): a
    # radius narrower than a band -> point-to-point halo rows from the two adjacent ranks; wider than a band
    # (factor 3.0 on a coarse mesh) -> all-gather of the bands; eight bands of O320 are one to two radii tall
    ("o96", 4, 0.6, 2, "gloo", True), ("o320", 5, 0.6, 3, "gloo", True), ("o96", 3, 3.0, 2, "gloo", True),
    ("o320", 5, 0.6, 8, "gloo", True), ("o96", 3, 3.0, 3, "gloo", True),
    ("o320", 5, 0.6, 1, "nccl", True),
])
def test_sharded_hip_build_equals_single(device, tmp_path, grid, resolution, factor, world, backend, exchange):
    from anemoi_graphs_amd import settings
    from anemoi_graphs_amd.distributed import ShardedGraphBuilder

    port = _free_port()
    mp.spawn(_worker, args=(world, port, grid, resolution, factor, str(tmp_path), backend, exchange), nprocs=world,
             join=True)
    data = torch.from_numpy(grids.named(grid)).to(device)
    single = ShardedGraphBuilder(data, resolution, cutoff_factor=factor, num_nearest_neighbours=3, x_hops=1,
                                 edge_length_norm="unit-max", edge_direction_norm="unit-std")
    with settings.resident(True):
        g = single.build()
    _dump(g, tmp_path / "single.npz")
    want = np.load(tmp_path / "single.npz")
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npz")
        for key in want.files:
            assert np.array_equal(got[key], want[key]), f"rank {r} {key}"


# ---------------------------------------------------------------------------------------------------
# band build with the HIP kernels: every rank generates and holds only its band of a stretched grid
# ---------------------------------------------------------------------------------------------------
def _stretched_grid():
    return grids.StretchedGrid(60.0, 7.0, ((45.0, 10.0, 2500.0),), seed=1)


def _band_dump(g, path):
    out = {}
    for name in NAMES:
        key = "_".join(name)
        out[key] = g[name].edge_index.cpu().numpy()
        out[key + "_len"] = g[name]["edge_length"].cpu().numpy()
        out[key + "_dir"] = g[name]["edge_dirs"].cpu().numpy()
    np.savez(path, **out)


def _band_worker(rank, world, port, resolution, factor, out_dir, backend="gloo"):
    from anemoi_graphs_amd import settings
    from anemoi_graphs_amd.distributed import BandGraphBuilder, band_bounds

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        grid = _stretched_grid()
        bounds = band_bounds(len(grid), world)
        lo, hi = bounds[rank], bounds[rank + 1]
        band = grid.generate(lo, hi, device=torch.device("cuda", 0))  # only this rank's rows ever exist here
        b = BandGraphBuilder(band, lo, len(grid), resolution, cutoff_factor=factor, num_nearest_neighbours=3, x_hops=1,
                             rank=rank, world_size=world, dist=dist, edge_length_norm="unit-max")
        with settings.resident(True):
            g = b.build()
        _band_dump(g, os.path.join(out_dir, f"rank{rank}.npz"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("resolution,factor,world,backend", [(5, 0.6, 2, "gloo"), (5, 0.6, 5, "gloo"), (4, 0.6, 8, "gloo"),
                                                             (3, 2.5, 3, "gloo"), (5, 0.6, 1, "nccl")])
def test_band_build_of_a_stretched_grid_equals_single(device, tmp_path, resolution, factor, world, backend):
    """BandGraphBuilder with the HIP kernels on a stretched grid (7 km disc in a 60 km background, ~430,000 nodes):
    ranks sharing the GPU over gloo (host-staged halo rows), one rank over RCCL.  Concatenated in rank order the
    ranks' encoder / decoder lists and attributes are those of the plain plugin sequence on the whole grid, bit for
    bit (factor 2.5 on a coarse mesh: reach wider than a band, all bands gathered)."""
    from anemoi_graphs_amd import settings
    from anemoi_graphs_amd.distributed import ShardedGraphBuilder

    port = _free_port()
    mp.spawn(_band_worker, args=(world, port, resolution, factor, str(tmp_path), backend), nprocs=world, join=True)
    data = _stretched_grid().generate(device=device)
    single = ShardedGraphBuilder(data, resolution, cutoff_factor=factor, num_nearest_neighbours=3, x_hops=1,
                                 edge_length_norm="unit-max", edge_direction_norm=None)
    with settings.resident(True):
        g = single.build()
    _band_dump(g, tmp_path / "single.npz")
    want = np.load(tmp_path / "single.npz")
    got = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    for key in ("data_to_hidden", "hidden_to_data"):
        for suffix in ("", "_len", "_dir"):
            cat = np.concatenate([p[key + suffix] for p in got], axis=1 if suffix == "" else 0)
            assert np.array_equal(cat, want[key + suffix]), key + suffix
    for p in got:
        for suffix in ("", "_len", "_dir"):
            assert np.array_equal(p["hidden_to_hidden" + suffix], want["hidden_to_hidden" + suffix])


# ---------------------------------------------------------------------------------------------------
# bench.py --gpus N starts its own ranks
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("workload,extra", [("stretched_small", []), ("o96_r4", []), ("o96_r4", ["--exchange-halo"])])
def test_bench_starts_its_own_ranks(device, workload, extra):
    """The command shape the multi-GPU measurement uses, ``python bench.py --gpus 2 ...`` with no launcher around
    it: the parent starts two fresh rank processes (before it has touched a GPU) and hands on rank 0's line.  On
    this one-GPU machine both ranks share GPU 0 and exchange through host memory (gloo)."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parents[1]
    env = dict(os.environ, AG_BENCH_ONE_DEVICE="1", AG_BENCH_BACKEND="gloo")
    for key in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(key, None)
    proc = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--workload", workload, "--steps", "1",
                           "--warmup", "1", *extra], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0, proc.stderr[-4000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, proc.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["backend"] == "gloo"
    assert line["value"] > 0 and line["steps"] == 1
    assert "phases_ms_rank0" in line
