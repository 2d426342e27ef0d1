#!/usr/bin/env python
"""Benchmark of the graph-construction hot path on MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python bench.py --gpus N ...                         (starts its own N ranks, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" builds the whole graph of the workload from node coordinates that are already
resident in HBM: TriNodes(r) hidden mesh, CutOffEdges(0.6) encoder data->hidden,
MultiScaleEdges(x_hops=1) processor hidden->hidden, KNNEdges(3) decoder hidden->data, and
EdgeLength + EdgeDirection for the three edge sets.  Default workload: O1280 (6,599,680 nodes)
-> icosphere r7 (163,842 nodes), the configuration the headline metric is quoted on.

``--workload stretched_1km``: a synthetic stretched grid (9 km global background, 1 km refinement disc,
``--nodes`` nodes in all, default 5e8) against icosphere r10; every rank generates and holds only its own
latitude band and keeps only its band's edges (``BandGraphBuilder``).

Prints ONE JSON line (rank 0).  ``value`` = edges of the finished graph / wall time, in Medges/s.
"""
from __future__ import annotations

import argparse
import json
import os
import re
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

# numpy, torch and the package are imported by run(): a process that only starts the ranks of a multi-GPU run
# (launch_ranks) loads nothing that could touch a GPU.

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak
WORKLOADS = {
    # name: (data grid, hidden resolution)
    "o1280_r7": ("o1280", 7),
    "era5_r6": ("era5", 6),
    "o96_r4": ("o96", 4),
    "o3600_r9": ("o3600", 9),  # 52 M data nodes, 2.6 M hidden nodes: a size check, not a headline
    # stretched synthetic grid, band-sharded (every rank makes and keeps only its band): 1 km disc in a 9 km
    # background against r10 (7 km mesh), and a small one of the same shape for rehearsals
    "stretched_1km": ("stretched", 10),
    "stretched_small": ("stretched", 7),
}
STRETCHED_DEFAULT_NODES = {"stretched_1km": 5.0e8, "stretched_small": 3.0e6}
STRETCHED_SPACING_KM = {"stretched_1km": (9.0, 1.0), "stretched_small": (90.0, 10.0)}
CUTOFF_FACTOR = 0.6
KNN_K = 3
X_HOPS = 1


def algorithmic_bytes(kernel: str, sizes: dict) -> float:
    """Compulsory HBM traffic per launch.  Node coordinates are float32 pairs (8 B/node, each node
    set read once), edge ids are int32 pairs (8 B/edge), attributes are float32 length + float32
    direction pair (12 B/edge)."""
    nd, nh = sizes["n_data"], sizes["n_hidden"]
    if kernel == "ag_knn_cell_kernel":  # decoder: hidden sources, data targets, both rows of the edge list written
        return 8.0 * nh + 8.0 * nd + 8.0 * sizes["e_dec"]
    if kernel in ("ag_radius_count_kernel", "ag_radius_grouped_kernel"):
        # the search of the cut-off query: both node sets read, a count per target written
        return 8.0 * nd + 8.0 * nh + 8.0 * nh
    if kernel == "ag_radius_copy_kernel":  # the rows the search kept, to their place in the edge list
        return 4.0 * sizes["e_enc"] + 8.0 * sizes["e_enc"]
    if kernel == "ag_edge_attrs_kernel":
        # one fused launch per edge set (3 per build), averaged.  Per edge: 8 B of ids read, 2 x 8 B of
        # endpoint coordinates gathered, 4 B of length and 8 B of direction written = 36 B.
        return 36.0 * sizes["e_total"] / 3.0
    if kernel in ("ag_bin_count_kernel", "ag_bin_scatter_kernel", "ag_bin_records_kernel"):
        # binning calls of a build, averaged (the data nodes once -- the binning is kept and serves the encoder as
        # sources and the decoder as targets -- and the hidden nodes once, as the decoder's sources).  Per point:
        # count reads the coordinates and writes the cell (12 B); scatter reads the cell and writes the id to its
        # place (8 B); records reads the id and its coordinates and writes (unit vector, id) (28 B)
        n = (nd + nh) / 2.0
        return {"ag_bin_count_kernel": 12.0, "ag_bin_scatter_kernel": 8.0, "ag_bin_records_kernel": 28.0}[kernel] * n
    if kernel == "ag_normalise_apply_kernel":  # the lengths of an edge set, read and written once (3 launches per build)
        return 8.0 * sizes["e_total"] / 3.0
    if kernel == "ag_ms_place_kernel":
        return 4.0 * nh + 8.0 * sizes["e_proc"]
    return 0.0


def build_bytes(sizes: dict) -> float:
    """Compulsory HBM traffic of one whole build: every node set read once per search that uses it, every edge
    id pair written once, 36 B per edge for the attributes (see ``algorithmic_bytes``)."""
    nd, nh = sizes["n_data"], sizes["n_hidden"]
    decoder = 8.0 * nh + 8.0 * nd + 8.0 * sizes["e_dec"]
    encoder = 8.0 * nh + (8.0 * nd + 8.0 * nh) + 8.0 * sizes["e_enc"]  # reference distance, then the radius query
    nodes = 12.0 * nh  # coordinates + node order
    processor = 4.0 * nh + 8.0 * sizes["e_proc"]
    return decoder + encoder + nodes + processor + 36.0 * sizes["e_total"]


def unique_bytes(kernel: str, sizes: dict):
    """A stricter count for kernels that gather: every node's coordinates counted once per launch however many
    edges touch it (what an ideal cache would leave of the gathers); None where it equals ``algorithmic_bytes``."""
    if kernel == "ag_edge_attrs_kernel":
        nd, nh = sizes["n_data"], sizes["n_hidden"]
        nodes = (nd + nh) + (nh + nh) + (nh + nd)
        return (20.0 * sizes["e_total"] + 8.0 * nodes) / 3.0
    return None


def traffic_table(workload: str):
    """{kernel: HBM bytes per launch} from the NEWEST committed rocprofv3 counter summary
    (profiles/rNN_hbm_traffic.json, written by tools/pmc_summary.py from separate FETCH_SIZE and WRITE_SIZE
    passes over this same command) and the name of that file.  The passes are labelled with the workload,
    optionally followed by how the build was run (``o1280_r7_one_stream``).  An older round's file is never
    consulted: when the newest summary does not cover the workload there is no measured traffic to quote."""
    def round_of(path: Path) -> int:
        m = re.match(r"r(\d+)_", path.name)
        return int(m.group(1)) if m else -1

    files = sorted((ROOT / "profiles").glob("r*_hbm_traffic.json"), key=round_of)
    if not files:
        return {}, None
    newest = files[-1]
    try:
        doc = json.loads(newest.read_text())
    except (OSError, ValueError):
        return {}, None
    labels = [k for k in doc if k == workload or k.startswith(workload + "_")]
    if not labels:
        return {}, None
    label = workload if workload in labels else sorted(labels)[0]
    table = {k: float(v["hbm_bytes_per_launch"]) for k, v in doc[label].get("kernels", {}).items()}
    return table, f"{newest.name}:{label}"


def cpu_baseline(data, hidden) -> dict:
    """The reference's CPU path (scikit-learn / numpy / scipy, called as the reference calls them) on a bounded
    sample of the same workload, timed on this machine's host cores: the two searches and the two edge
    attributes of the sampled edges.  The mesh (TriNodes, MultiScaleEdges) is not in it: the reference builds it
    with trimesh and networkx, and trimesh is not installed."""
    from oracle import reference_path as rp

    n_dec = min(len(data), 150_000)
    n_enc = min(len(hidden), 16_384)
    lo = (len(data) - n_dec) // 2
    t0 = time.perf_counter()
    e_dec = rp.knn_edges(hidden, data[lo:lo + n_dec], KNN_K)
    t1 = time.perf_counter()
    radius = rp.cutoff_radius(hidden, CUTOFF_FACTOR)
    lo_h = (len(hidden) - n_enc) // 2
    e_enc = rp.radius_edges(data, hidden[lo_h:lo_h + n_enc], radius)
    t2 = time.perf_counter()
    # EdgeLength + EdgeDirection of the sampled edges (utils.py:99-103, edges/directional.py:40-94)
    for e, sx, tx in ((e_dec, hidden, data[lo:lo + n_dec]), (e_enc, data, hidden[lo_h:lo_h + n_enc])):
        rp.post_process(rp.edge_lengths(sx, tx, e), "unit-max")
        rp.post_process(rp.edge_directions(sx, tx, e), None)
    t3 = time.perf_counter()
    edges = e_dec.shape[1] + e_enc.shape[1]
    return {
        "value": round(edges / (t3 - t0) / 1e6, 4),
        "unit": "Medges/s",
        "cores": rp.N_JOBS,
        "host_cpus": os.cpu_count(),
        "kind": "port",
        "kind_note": "oracle/reference_path.py: the reference's own third-party calls (scikit-learn NearestNeighbors, "
                     "numpy, scipy Rotation) restated call for call; the reference package itself is not on this machine",
        "sample": (
            f"scikit-learn NearestNeighbors(haversine, n_jobs={rp.N_JOBS}) as the reference calls it: "
            f"KNN k={KNN_K} decoder for {n_dec} of {len(data)} data nodes ({t1 - t0:.1f} s) + reference distance "
            f"and CutOff encoder for {n_enc} of {len(hidden)} hidden nodes over all data nodes ({t2 - t1:.1f} s) + "
            f"EdgeLength (numpy haversine) and EdgeDirection (scipy Rotation) of those {edges} edges ({t3 - t2:.1f} s, "
            f"one thread: numpy); tree builds included"
        ),
        "excluded": "TriNodes and MultiScaleEdges (trimesh / networkx path of the reference: trimesh is not installed), "
                    "so the CPU number covers less work than `value` does",
        "search_medges_per_s": round(edges / (t2 - t0) / 1e6, 4),
    }


def host_tensor_step(data_np, resolution: int, steps: int = 3):
    """One build in the package's DEFAULT mode, as the reference hands graphs around: node coordinates come as host
    tensors, every edge list and attribute goes back to host memory (PCIe both ways inside the step).  Returns
    (milliseconds per build, bytes that crossed the link per build)."""
    import torch

    from anemoi_graphs_amd import settings
    from anemoi_graphs_amd.distributed import ShardedGraphBuilder

    data_host = torch.from_numpy(data_np)
    builder = ShardedGraphBuilder(data_host, resolution, cutoff_factor=CUTOFF_FACTOR, num_nearest_neighbours=KNN_K,
                                  x_hops=X_HOPS)
    with settings.resident(False):
        for _ in range(2):  # (two builds before the clock starts: the page-locked buffers of a build are free again
            graph = builder.build()  # only when the graph before it has been dropped, so two sets of them exist)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            graph = builder.build()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    moved = data_host.numel() * data_host.element_size()
    for name in [*graph.node_types, *graph.edge_types]:
        for key, value in graph[name].items():
            if isinstance(value, torch.Tensor) and not (key == "x" and value is data_host):
                moved += value.numel() * value.element_size()
    return elapsed / steps * 1e3, moved


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="o1280_r7", choices=sorted(WORKLOADS))
    ap.add_argument("--nodes", type=float, default=None, help="data nodes of a stretched workload")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-tensors", action="store_true", help="skip the default-mode (host tensors) timing")
    ap.add_argument("--no-overlap", action="store_true",
                    help="single GPU: queue the processor edges behind the encoder instead of on a second stream")
    ap.add_argument("--exchange-halo", action="store_true",
                    help="o1280-style workloads on N > 1 ranks: treat the coordinates as sharded (halo rows travel)")
    ap.add_argument("--profile", default="dominant", choices=["dominant", "all", "off"],
                    help="which kernel launches carry HIP events inside the timed region")
    return ap.parse_args(argv)


def launch_ranks(args, argv) -> int:
    """``--gpus N`` without a launcher: start N fresh processes -- ``python -m torch.distributed.run``, one rank
    per GPU -- and hand on rank 0's JSON line.  This process has not touched a GPU (it has not even imported torch)
    and never will: it only waits for the children and returns their exit code, or 1 when the line is missing or
    does not report N ranks."""
    with socket.socket() as s:  # a free rendezvous port
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)  # (stderr passes through)
    line = None
    for raw in proc.stdout.splitlines():
        raw = raw.strip()
        if raw.startswith("{") and '"metric"' in raw:
            line = raw
    if proc.returncode != 0:
        sys.stderr.write(f"bench.py: the launcher of the {args.gpus} ranks ended with code {proc.returncode}\n")
        return proc.returncode
    if line is None:
        sys.stderr.write("bench.py: the ranks printed no result line\n")
        return 1
    try:
        ranks = int(json.loads(line).get("n_gpus", -1))
    except ValueError:
        ranks = -1
    if ranks != args.gpus:
        sys.stderr.write(f"bench.py: the result line reports {ranks} ranks, --gpus asked for {args.gpus}\n")
        return 1
    print(line, flush=True)
    return 0


def kernel_table(warm_ms: dict, sizes: dict, traffic: dict, top: int = 5) -> dict:
    """The longest kernels of one fully event-timed build: average launch duration, compulsory bytes, the fraction
    of the HBM peak those bytes make, and (where the committed counter summary covers the kernel) measured traffic
    per launch and its ratio to the compulsory bytes."""
    rows = {}
    ranked = sorted(((k, v) for k, v in warm_ms.items() if k.startswith("ag_") and v[0] > 0),
                    key=lambda kv: kv[1][1], reverse=True)
    for name, (launches, total_ms) in ranked[:top]:
        avg_ms = total_ms / launches
        nbytes = algorithmic_bytes(name, sizes)
        row = {"launches": launches, "avg_ms": round(avg_ms, 4), "algorithmic_bytes": nbytes or None}
        if nbytes and avg_ms > 0:
            row["frac"] = round(nbytes / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        if name in traffic:
            row["traffic"] = traffic[name]
            if nbytes:
                row["traffic_ratio"] = round(traffic[name] / nbytes, 2)
        rows[name] = row
    return rows


def run(args) -> int:
    import torch

    from anemoi_graphs_amd import distributed, grids, profiling, settings
    from anemoi_graphs_amd.distributed import BandGraphBuilder, ShardedGraphBuilder, band_bounds

    # stdout carries the one JSON line and nothing else: whatever libraries write to file descriptor 1
    # (RCCL prints a version banner there when the first communicator is created) goes to stderr
    line_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    # rehearsal knobs for a one-GPU machine: AG_BENCH_ONE_DEVICE=1 puts every rank on GPU 0 and
    # AG_BENCH_BACKEND=gloo exchanges through host memory (RCCL refuses two ranks on one GPU)
    if os.environ.get("AG_BENCH_ONE_DEVICE"):
        local_rank = 0
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} has no GPU of its own ({torch.cuda.device_count()} visible); "
                         "one rank per GPU is the rule (AG_BENCH_ONE_DEVICE=1 AG_BENCH_BACKEND=gloo rehearses on one)")
    backend = os.environ.get("AG_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # AG_BENCH_FORCE_DIST=1: a process group even for one rank, so that a one-GPU machine runs the
    # sharded route (collectives of one rank) instead of the plain single-GPU sequence
    if world > 1 or os.environ.get("AG_BENCH_FORCE_DIST"):
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if os.environ["MASTER_ADDR"] in ("127.0.0.1", "localhost"):
            # one node: the host-side (gloo) group that carries the edge counts talks over the loopback interface,
            # whatever the machine's host name resolves to (or fails to)
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")

        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
        group_ranks = dist.get_world_size()
        if group_ranks != args.gpus:
            raise SystemExit(f"bench.py: the process group has {group_ranks} ranks, --gpus asked for {args.gpus}")
    else:
        dist = None
        group_ranks = 1

    grid_name, resolution = WORKLOADS[args.workload]
    stretched = grid_name == "stretched"
    if stretched:
        nodes = args.nodes or STRETCHED_DEFAULT_NODES[args.workload]
        coarse_km, fine_km = STRETCHED_SPACING_KM[args.workload]
        grid = grids.stretched_for_nodes(nodes, coarse_km, fine_km)
        n_data = len(grid)
        bounds = band_bounds(n_data, world)
        band = grid.generate(bounds[rank], bounds[rank + 1], device=device)  # this rank's rows only
        data_np = None
        builder = BandGraphBuilder(band, bounds[rank], n_data, resolution, cutoff_factor=CUTOFF_FACTOR,
                                   num_nearest_neighbours=KNN_K, x_hops=X_HOPS, rank=rank, world_size=world, dist=dist)
        workload_name = (f"stretched synthetic grid ({n_data} nodes: {coarse_km:g} km global background, {fine_km:g} km "
                         f"inside a disc of {grid.discs[0][2] * grids.EARTH_RADIUS_KM:.0f} km radius)")
    else:
        data_np = grids.named(grid_name)
        n_data = len(data_np)
        data = torch.from_numpy(data_np).to(device)
        builder = ShardedGraphBuilder(
            data, resolution, cutoff_factor=CUTOFF_FACTOR, num_nearest_neighbours=KNN_K, x_hops=X_HOPS,
            rank=rank, world_size=world, dist=dist, exchange_halo=args.exchange_halo,
            overlap_processor=False if args.no_overlap else None,
        )
        workload_name = f"{grid_name.upper()} ({n_data} nodes)"

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)

    with settings.resident(True):
        graph = None
        # every kernel is event-timed during one untimed build (per-kernel table, choice of the dominant
        # kernel); in the timed region only the dominant kernel's launches carry events
        for _ in range(args.warmup):
            graph = builder.build()
        barrier()
        profiling.enable(True)
        builder.timer.enabled = world > 1 or stretched  # phases of a sharded build (one untimed step)
        builder.timer.reset()
        graph = builder.build()  # one more untimed step, with events on every kernel launch
        barrier()
        phase_ms = builder.timer.milliseconds() if builder.timer.enabled else None
        builder.timer.enabled = False
        warm_ms = profiling.collect()
        # the dominant hand-written kernel among those with a stated algorithmic byte count
        probe = {"n_data": 1, "n_hidden": 1, "e_enc": 1, "e_dec": 1, "e_proc": 1, "e_total": 1}
        own = {k: v for k, v in warm_ms.items() if k.startswith("ag_") and algorithmic_bytes(k, probe) > 0.0}
        dominant = max(own.items(), key=lambda kv: kv[1][1])[0] if own else None
        profiling.enable(args.profile != "off", only=None if args.profile == "all" else dominant)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            graph = builder.build()
        barrier()
        elapsed = time.perf_counter() - t0
        kernel_ms = profiling.collect()  # {kernel: (launches, total ms)} over the timed region
        profiling.enable(False)
        # A single-GPU build runs its edge sets on two streams: in the timed region the dominant kernel shares the
        # GPU with the kernels of the other stream, and its launches take longer than they do alone.  Its duration
        # alone is measured afterwards, over builds on one stream, and reported beside the timed-region figure.
        isolated_ms = None
        iso_steps = 0
        if world == 1 and not stretched and getattr(builder, "overlap_processor", False) and dominant and args.profile != "off":
            builder.overlap_processor = False
            graph = builder.build()
            barrier()
            profiling.enable(True)  # every launch: the per-kernel table comes from these builds too
            iso_steps = max(1, min(args.steps, 10))
            for _ in range(iso_steps):
                graph = builder.build()
            barrier()
            isolated_ms = profiling.collect()
            profiling.enable(False)
            builder.overlap_processor = True

    e_enc = int(graph[("data", "to", "hidden")].edge_index.shape[1])
    e_proc = int(graph[("hidden", "to", "hidden")].edge_index.shape[1])
    e_dec = int(graph[("hidden", "to", "data")].edge_index.shape[1])
    if dist is not None:
        stage = device if backend == "nccl" else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=stage)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        if stretched:  # every rank holds its band's edges: the graph is their sum (the processor is replicated)
            counts = torch.tensor([e_enc, e_dec], dtype=torch.int64, device=stage)
            dist.all_reduce(counts, op=dist.ReduceOp.SUM)
            e_enc, e_dec = (int(v) for v in counts.tolist())
    e_total = e_enc + e_proc + e_dec
    sizes = {"n_data": n_data, "n_hidden": int(graph["hidden"].x.shape[0]), "e_enc": e_enc, "e_dec": e_dec,
             "e_proc": e_proc, "e_total": e_total}

    if rank != 0:
        if dist is not None:
            distributed.shutdown(dist)
            dist.destroy_process_group()
        return 0

    ms_per_step = elapsed / args.steps * 1e3
    per_rank = dict(sizes)
    if stretched and world > 1:  # a launch of a band build works on one rank's band
        for key in ("n_data", "e_enc", "e_dec"):
            per_rank[key] = sizes[key] / world
        per_rank["e_total"] = per_rank["e_enc"] + per_rank["e_dec"] + e_proc
    traffic_by_kernel, traffic_src = traffic_table(args.workload) if world == 1 else ({}, None)
    # dominant kernel of the timed region, by total device time (HIP events on the launch stream)
    roofline = None
    if kernel_ms and dominant in kernel_ms:
        # (rocPRIM's radix sort is a multi-kernel library composite timed as a whole; the roofline
        # object is about the dominant hand-written kernel)
        name = dominant
        launches, total_ms = kernel_ms[name]
        avg_ms = total_ms / max(launches, 1)
        nbytes = algorithmic_bytes(name, per_rank)
        achieved = nbytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        strict = unique_bytes(name, per_rank)
        total_bytes = build_bytes(sizes)
        roofline = {
            "bound": "hbm", "kernel": name, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic_by_kernel.get(name),
            "traffic_source": traffic_src if name in traffic_by_kernel else None,
            "avg_launch_ms": round(avg_ms, 4), "launches": launches, "algorithmic_bytes_per_launch": nbytes,
            "measured": "HIP events on the launch stream around every launch of the kernel inside the timed region"
                        + (" (two streams: the kernel shares the GPU with the kernels of the other stream)"
                           if isolated_ms else ""),
            # the whole build against the same roofline: its compulsory bytes over the wall time of a step
            "build_bytes": total_bytes,
            "build_frac": round(total_bytes / (ms_per_step * 1e-3) / 1e9 / (HBM_PEAK_GBS * world), 5),
            "launches_per_build": int(sum(v[0] for v in warm_ms.values())),
        }
        if strict is not None and avg_ms > 0:
            # gathered coordinates counted once per node instead of once per edge
            roofline["unique_bytes_per_launch"] = strict
            roofline["frac_unique_bytes"] = round(strict / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
        if isolated_ms and name in isolated_ms and isolated_ms[name][1] > 0:
            iso_launches, iso_total = isolated_ms[name]
            iso_avg = iso_total / max(iso_launches, 1)
            roofline["isolated"] = {
                "what": "the kernel's launches alone: the same builds on one stream (--no-overlap), run after the "
                        "timed region",
                "avg_launch_ms": round(iso_avg, 4), "launches": iso_launches,
                "achieved": round(nbytes / (iso_avg * 1e-3) / 1e9, 2),
                "frac": round(nbytes / (iso_avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
            }
        # the longest kernels, every launch bracketed by HIP events: of the one-stream builds after the timed region
        # when there are any (a kernel's duration alone), else of the one fully event-timed build before it
        table_ms = isolated_ms if isolated_ms else warm_ms
        roofline["kernels"] = kernel_table(table_ms, per_rank, traffic_by_kernel)
        roofline["kernels_source"] = (
            (f"{iso_steps} builds on one stream after the timed region" if isolated_ms else "one untimed build")
            + " with HIP events on every launch; traffic: "
            + (traffic_src or "no committed counter summary covers this workload"))
        roofline["kernels_ms_untimed_profiled_step"] = {k: round(v[1], 4) for k, v in sorted(warm_ms.items())}

    if stretched:
        parallelism = "single GPU" if world == 1 else f"latitude bands x{world}, band-local input and output, halo rows only"
    elif world == 1:
        parallelism = "single GPU"
    else:
        parallelism = (f"latitude bands x{world}, " + ("sharded coordinates + halo rows" if args.exchange_halo
                                                        else "replicated coordinates") + ", edge lists all-gathered")
    out = {
        "metric": "Medges/s, full graph build (TriNodes + CutOff encoder + MultiScale processor + KNN decoder + edge attributes)",
        "value": round(e_total / (elapsed / args.steps) / 1e6, 3),
        "unit": "Medges/s",
        "n_gpus": world,
        "rccl_ranks": group_ranks,
        "backend": (backend if dist is not None else "none (one process)"),
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "dtype_note": "float64 haversine ranking behind a float32 filter, float32 attributes, int32 ids",
        "data": "synthetic",
        "config": {
            "workload": f"{workload_name} -> icosphere r{resolution} "
                        f"({sizes['n_hidden']} nodes): CutOff {CUTOFF_FACTOR} encoder {e_enc} edges, MultiScale "
                        f"x_hops={X_HOPS} processor {e_proc} edges, KNN k={KNN_K} decoder {e_dec} edges, "
                        f"EdgeLength+EdgeDirection on all",
            "edges": e_total,
            "parallelism": parallelism,
        },
        "roofline": roofline,
    }
    if phase_ms is not None:
        # device time of the phases of ONE untimed, fully event-timed build on rank 0 (waits for collectives included
        # in the phase that waits): where the time of a sharded build goes
        out["phases_ms_rank0"] = {k: round(v, 4) for k, v in phase_ms.items()}
    if world == 1 and not stretched:
        if not args.no_host_tensors:
            host_ms, moved = host_tensor_step(data_np, resolution)
            out["ms_per_step_host_tensors"] = round(host_ms, 3)
            out["pcie_gb_s"] = round(moved / (host_ms * 1e-3) / 1e9, 2)
            out["ms_per_step_host_tensors_note"] = (
                "the package's default mode, PCIe inside the step: coordinates from host tensors, edge lists and "
                f"attributes back to host memory ({moved / 1e6:.0f} MB over the link per build); never `value`")
        if not args.no_cpu_baseline:
            hidden_np = graph["hidden"].x.cpu().numpy()
            out["cpu_baseline"] = cpu_baseline(data_np, hidden_np)
    sys.stdout.flush()
    os.write(line_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        distributed.shutdown(dist)
        dist.destroy_process_group()
    return 0


def main() -> int:
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args, argv)
    return run(args)


if __name__ == "__main__":
    sys.exit(main())
