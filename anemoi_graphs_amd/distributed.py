"""Latitude-band sharding of one graph build over the GPUs of a node (one process per GPU).

The data nodes are split into ``world_size`` contiguous latitude bands of equal node count (the
data grid must be ordered north -> south, as every generated grid and the icosphere nodes are).
Rank r owns band r:

* decoder (KNNEdges hidden -> data): targets = the rank's own data band, sources = the hidden
  mesh, which is tiny and replicated -> no exchange;
* encoder (CutOffEdges data -> hidden): targets = the hidden nodes whose latitude falls in the
  band (a contiguous id range, hidden nodes are latitude-sorted), sources = the rank's data band
  widened by the cut-off radius;
* processor (MultiScaleEdges) and TriNodes: replicated, they are a few MB;
* the edge lists are all-gathered (variable length, rank order = target order, so the
  concatenation is already the single-GPU edge list); the all-gathers run asynchronously while the
  next stage computes.  Edge attributes and their global normalisations are then evaluated on the
  gathered edge lists by every rank: 8 B/edge of edge ids travel over xGMI instead of 20 B/edge, and
  the values are bit-identical to the single-GPU build by construction.

Every rank is given the complete coordinate array (53 MB for O1280), so the widened source band is a
slice of it and no coordinates travel (``exchange_halo=True`` treats them as sharded instead: the
missing rows come from the two adjacent ranks by point-to-point send/recv, or from an all-gather of the
bands when the radius is wider than a band).  Collectives of a build: the edge counts of the encoder
bands (host integers, over a host-side gloo group so that they neither wait for queued device work
nor occupy the stream), and the all-gathers of the edge lists into their final place.  With the
``gloo`` backend (CPU tests) tensors are staged through host memory.
"""
from __future__ import annotations

import math

import torch

from . import ops as _hip_ops


def _global_sources(backend) -> dict:
    """What the HIP searches are told about a hidden mesh that covers the sphere (``ops.point_density`` = 1), so that
    they need not measure it; other backends take no such argument."""
    return {"src_density": 1.0} if backend is _hip_ops else {}


def _reference_distance(backend, resolution: int, hidden, rank_map, level_edges):
    """Grid reference distance of the complete hidden mesh: from its mesh edges with the HIP backend (the nearest
    neighbour of a vertex of the icosphere is a mesh neighbour), by the search of any other backend."""
    if backend is _hip_ops:
        return backend.mesh_maxdist(resolution, level_edges, hidden, rank_map)
    return backend.nn_maxdist(hidden)
from . import settings
from .hetero import HeteroData


# One host-side (gloo) group per process, shared by every builder: creating a group is a collective, so it is
# done once, by all ranks together, and whether it is usable is decided by all ranks together.
_HOST_GROUP: dict = {}


def host_group(dist):
    """The process-wide gloo group for host integers next to a device (RCCL) process group: (group, usable).
    Created on first use by a call every rank makes; ``usable`` is the same on every rank (a rank where the
    creation failed makes all ranks fall back to exchanging the integers on the device)."""
    if "group" not in _HOST_GROUP:
        group, ok = None, 1
        try:
            group = dist.new_group(backend="gloo")
        except Exception:  # noqa: BLE001 - no usable host transport on this rank
            ok = 0
        # agree: the minimum over the ranks, through the process group that is known to work
        flag = torch.tensor([ok], dtype=torch.int32, device=torch.device("cuda", torch.cuda.current_device()))
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        _HOST_GROUP["group"], _HOST_GROUP["usable"] = group, bool(flag.item())
    return _HOST_GROUP["group"], _HOST_GROUP["usable"]


def shutdown(dist) -> None:
    """Release the host-side group (call before ``dist.destroy_process_group()``)."""
    group = _HOST_GROUP.pop("group", None)
    _HOST_GROUP.pop("usable", None)
    if group is not None:
        dist.destroy_process_group(group)


class Communicator:
    """The three exchanges the sharded build needs, over ``torch.distributed`` (RCCL or gloo)."""

    def __init__(self, dist, rank: int, world_size: int):
        self.dist, self.rank, self.world = dist, rank, world_size
        self.host_staged = dist is not None and dist.get_backend() == "gloo"
        # Edge counts and the like are host integers: they travel over a host-side (gloo) group, so that the
        # exchange neither waits for the device work queued before it nor puts a kernel into the stream.
        self.host_group = None
        self.sizes_on_device = False
        if dist is not None and not self.host_staged:
            self.host_group, usable = host_group(dist)
            self.sizes_on_device = not usable
            if not usable:
                self.host_group = None

    def _out(self, t: torch.Tensor) -> torch.Tensor:
        return t.cpu() if self.host_staged else t

    def _back(self, t: torch.Tensor, device) -> torch.Tensor:
        return t.to(device) if self.host_staged else t

    def all_gather_sizes(self, values: list[int], device) -> torch.Tensor:
        return self.all_gather_sizes_async(values, device)()

    def all_gather_sizes_async(self, values: list[int], device):
        """Start the exchange of a few host integers per rank; the returned callable waits and gives the
        [world, len(values)] table on the host."""
        where = device if self.sizes_on_device else "cpu"
        mine = torch.tensor(values, dtype=torch.int64, device=where)
        out = torch.empty((self.world, len(values)), dtype=torch.int64, device=where)
        work = self.dist.all_gather_into_tensor(out.reshape(-1), mine, group=self.host_group, async_op=True)

        def finish() -> torch.Tensor:
            work.wait()
            return out.cpu()

        return finish

    def all_gather_rows(self, t: torch.Tensor, counts: list[int] | None = None) -> torch.Tensor:
        """Concatenate variable-length chunks [n_r, ...] of every rank along dim 0 in rank order."""
        return self.all_gather_rows_async([t], counts)()[0]

    def all_gather_rows_async(self, parts: list[torch.Tensor], counts: list[int] | None = None,
                              out: list[torch.Tensor] | None = None):
        """Start one all-gather per tensor of ``parts`` (all with the same, per-rank variable, length
        along dim 0); the returned callable waits for them and gives the list of concatenations in rank
        order.  ``counts`` (the chunk length of every rank) saves the size exchange when the caller knows
        them.  When every rank holds the same number of rows the chunks land directly in their final
        place (``out[i]`` if given, e.g. a row of an edge_index): no padding, no second copy."""
        device = parts[0].device
        n = parts[0].shape[0]
        if counts is None:
            counts = self.all_gather_sizes([n], device)[:, 0].tolist()
        width, total = max(counts), sum(counts)
        equal = total == width * self.world
        stage = "cpu" if self.host_staged else device
        works, gathered, keep = [], [], []
        for i, t in enumerate(parts):
            t = t.contiguous()
            if n != width:
                padded = torch.zeros((width,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
                padded[:n] = t
                t = padded
            t = self._out(t)
            direct = equal and not self.host_staged and out is not None
            buf = out[i] if direct else torch.empty((self.world * width,) + tuple(t.shape[1:]), dtype=t.dtype,
                                                    device=stage)
            works.append(self.dist.all_gather_into_tensor(buf, t, async_op=True))
            gathered.append(buf)
            keep.append(t)

        def finish() -> list[torch.Tensor]:
            for w in works:
                w.wait()
            res = []
            for i, buf in enumerate(gathered):
                dst = out[i] if out is not None else None
                if dst is None and equal:
                    dst = self._back(buf, device)
                elif dst is None or dst.data_ptr() != buf.data_ptr():
                    if dst is None:
                        dst = torch.empty((total,) + tuple(buf.shape[1:]), dtype=buf.dtype, device=device)
                    off = 0
                    for r, c in enumerate(counts):  # drop the padding of the shorter chunks
                        dst[off: off + c].copy_(buf[r * width: r * width + c])
                        off += c
                res.append(dst)
            keep.clear()
            return res

        return finish

    def exchange_halo(self, to_north: torch.Tensor, to_south: torch.Tensor, n_from_north: int, n_from_south: int):
        """Send rows to the two neighbours, receive theirs (``n_from_*`` rows): returns (from_north, from_south)."""
        device = to_north.device
        recv_dev = "cpu" if self.host_staged else device
        from_north = torch.empty((n_from_north, 2), dtype=to_north.dtype, device=recv_dev)
        from_south = torch.empty((n_from_south, 2), dtype=to_north.dtype, device=recv_dev)
        ops = []
        P2POp = self.dist.P2POp
        if self.rank > 0:
            if to_north.shape[0]:
                ops.append(P2POp(self.dist.isend, self._out(to_north.contiguous()), self.rank - 1))
            if n_from_north:
                ops.append(P2POp(self.dist.irecv, from_north, self.rank - 1))
        if self.rank < self.world - 1:
            if to_south.shape[0]:
                ops.append(P2POp(self.dist.isend, self._out(to_south.contiguous()), self.rank + 1))
            if n_from_south:
                ops.append(P2POp(self.dist.irecv, from_south, self.rank + 1))
        if ops:
            for req in self.dist.batch_isend_irecv(ops):
                req.wait()
        return self._back(from_north, device), self._back(from_south, device)


def band_bounds(n_total: int, world_size: int) -> list[int]:
    """Equal-count contiguous bands: band r = ids [b[r], b[r+1])."""
    return [(n_total * r) // world_size for r in range(world_size + 1)]


class ShardedGraphBuilder:
    """Builds the encoder / processor / decoder graph of a data grid and an icosphere of the given
    resolution; on ``world_size`` > 1 ranks each builds its latitude band and all end with the
    complete graph.  ``backend`` provides the search / mesh / attribute operations (default: the HIP
    kernels in ``anemoi_graphs_amd.ops``; the CPU tests inject the oracle)."""

    def __init__(self, data_x: torch.Tensor, resolution: int, cutoff_factor: float = 0.6,
                 num_nearest_neighbours: int = 3, x_hops: int = 1, rank: int = 0, world_size: int = 1, dist=None,
                 backend=None, attributes: bool = True, edge_length_norm: str | None = "unit-max",
                 edge_direction_norm: str | None = None, exchange_halo: bool = False, overlap_processor: bool | None = None):
        self.backend = _hip_ops if backend is None else backend
        # Two streams for a single-GPU build (see _build_single).  Handing work from one stream to the other costs
        # the device about 30 us each time, 0.13 ms per build: worth it when the searches run for longer than that
        # (measured: O1280 -> r7 1.20 -> 1.04 ms per build, ERA5 0.25 deg -> r6 0.73 -> 0.89 ms), hence by default
        # only for data grids of a few million nodes.
        self.overlap_processor = int(data_x.shape[0]) >= 2_500_000 if overlap_processor is None else overlap_processor
        self._side = {}
        self.resolution = resolution
        self.cutoff_factor = cutoff_factor
        self.k = num_nearest_neighbours
        self.x_hops = x_hops
        self.rank, self.world = rank, world_size
        self.attributes = attributes
        self.len_norm, self.dir_norm = edge_length_norm, edge_direction_norm
        self.n_data = int(data_x.shape[0])
        self.bounds = band_bounds(self.n_data, world_size)
        lo, hi = self.bounds[rank], self.bounds[rank + 1]
        self.d_lo, self.d_hi = lo, hi
        self.data_full = data_x if world_size == 1 else None
        self.data_all = data_x  # every rank is given the complete coordinate array (53 MB for O1280)
        self.data_local = data_x[lo:hi].contiguous()
        self.exchange_halo = exchange_halo
        self.timer = PhaseTimer()
        if world_size > 1 or dist is not None:  # (a process group of one rank still takes the sharded route)
            lat = data_x[:, 0]
            if lat.numel() > 1 and bool((lat[1:] > lat[:-1]).any()):
                raise ValueError("latitude-band sharding needs data nodes ordered north -> south")
            self.comm = Communicator(dist, rank, world_size)
            # cut latitudes: the first latitude of every band (band r holds lat <= first_lat[r]); every rank
            # holds the complete coordinate array, so they are read, not exchanged
            firsts = torch.as_tensor([min(b, self.n_data - 1) for b in self.bounds[:-1]], device=data_x.device)
            self.first_lat = lat[firsts].cpu().tolist()
            # search keys of the band limits, kept on the device (ascending = negated latitudes)
            dev = self.data_local.device
            self.neg_lat = (-self.data_local[:, 0]).contiguous()
            self.neg_lat_all = (-lat).contiguous()
            self.neg_upper = torch.tensor(-self.first_lat[rank], dtype=lat.dtype, device=dev)
            self.neg_lower = torch.tensor(-self.first_lat[min(rank + 1, world_size - 1)], dtype=lat.dtype, device=dev)
        else:
            self.comm = None

    def _side_stream(self, device):
        """The second stream of a single-GPU build (default priority), made once per device."""
        key = device.index if device.index is not None else torch.cuda.current_device()
        if key not in self._side:
            self._side[key] = torch.cuda.Stream(device=device)
        return self._side[key]

    # ---------------------------------------------------------------------------------- single GPU
    def _build_single(self) -> HeteroData:
        """The plain plugin sequence the orchestrator runs (create.py:78-90 of the reference)."""
        from .edges import CutOffEdges, KNNEdges, MultiScaleEdges
        from .nodes import TriNodes

        graph = HeteroData()
        graph["data"].x = self.data_full
        graph["data"].node_type = "LatLonNodes"
        overlap = self.overlap_processor and torch.as_tensor(self.data_full).is_cuda and self.backend is _hip_ops
        if overlap:
            # The binning of the data nodes that the encoder and the decoder share needs the cut-off radius only for
            # the size of its cells, and that within a factor: it is started now, on the second stream, beside the
            # generation of the mesh, with the radius a quasi-uniform mesh of this many nodes will turn out to have
            # (reference distance = 1.18 x the mean node spacing for icospheres of level 5 and above, less for coarser
            # ones).  The encoder takes the binning if its cells are within its accepted range of the ones it would
            # choose for the radius it then computes, and bins again otherwise.
            from .edges.builder import kept_binning

            dev = self.data_full.device
            main, side = torch.cuda.current_stream(dev), self._side_stream(dev)
            n_hidden = 10 * 4 ** self.resolution + 2
            guess = self.cutoff_factor * 1.185 * math.sqrt(4.0 * math.pi / n_hidden)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                kept_binning(graph["data"], _hip_ops.radius_cell(self.n_data, guess))
        graph = TriNodes(self.resolution, "hidden").update_graph(graph, {})
        builders = [
            CutOffEdges("data", "hidden", self.cutoff_factor),
            MultiScaleEdges("hidden", "hidden", self.x_hops),
            KNNEdges("hidden", "data", self.k),
        ]
        attrs = {
            "edge_length": {"_target_": "anemoi.graphs.edges.attributes.EdgeLength", "norm": self.len_norm},
            "edge_dirs": {"_target_": "anemoi.graphs.edges.attributes.EdgeDirection", "norm": self.dir_norm},
        }
        def run(b):  # as the orchestrator does: edges first, then the attributes of the edge set
            b.update_graph(graph, attrs_config=None)
            if self.attributes:
                b.register_attributes(graph, attrs)

        def finished(graph):
            # the working memory of the build -- the kept binning of the data nodes (190 MB for O1280), the cached
            # reference distance and densities -- does not leave with the graph; what MultiScaleEdges and the tests
            # of the node builders read (the mesh, the node order) stays, as after ``TriNodes.update_graph``
            for name in graph.node_types:
                for key in ("_binning", "_reference_distance", "_density"):
                    if key in graph[name]:
                        del graph[name][key]
            return graph

        x = graph["hidden"].x
        if not (overlap and x.is_cuda):
            for b in builders:
                run(b)
            return finished(graph)
        # Two streams.  The processor edges need the hidden mesh and nothing else, and their kernels are many and
        # small (they wait for memory, not for compute units): they are queued on a second stream, behind the mesh,
        # and run beside the reference distance of the first.  The decoder search needs the binned data nodes -- which
        # the encoder makes -- and no answer from the device: queued on the second stream once the binning is, it
        # fills the GPU while the host waits for the size of the encoder's edge list.  The edge sets are registered
        # first, so that the graph lists them in the order a sequential build does.
        encoder, processor, decoder = builders
        for b in builders:
            graph[(b.source_name, "to", b.target_name)]
        side.wait_stream(main)
        with torch.cuda.stream(side):
            run(processor)
        radius = encoder.get_cutoff_radius(graph)  # (kept with the hidden nodes: the encoder does not compute it again)
        kept_binning(graph["data"], _hip_ops.radius_cell(self.n_data, radius))
        with torch.cuda.stream(side):
            run(decoder)
        run(encoder)
        main.wait_stream(side)
        return finished(graph)

    # ---------------------------------------------------------------------------------- sharded
    def _attrs(self, src_x, tgt_x, edge_index):
        if self.backend is _hip_ops:  # the kernel reports the range of the lengths for their normalisation
            length, direction, len_range = self.backend.edge_attrs(src_x, tgt_x, edge_index, len_range=True)
            return length.reshape(-1, 1), direction, len_range
        length, direction = self.backend.edge_attrs(src_x, tgt_x, edge_index)
        return length.reshape(-1, 1), direction, None

    def _finish_attrs(self, store, length, direction, len_range=None):
        extra = {} if len_range is None else {"value_range": len_range}
        store["edge_length"] = self.backend.normalise_(length.contiguous(), self.len_norm, **extra)
        store["edge_dirs"] = self.backend.normalise_(direction.contiguous(), self.dir_norm)

    def _build_sharded(self) -> HeteroData:
        be, comm, r, W = self.backend, self.comm, self.rank, self.world
        dev = self.data_local.device
        mark = self.timer.mark
        mark("start")
        graph = HeteroData()
        _xyz, _faces, hidden, order, rank_map, level_edges = be.icosphere(self.resolution)
        hidden = hidden.to(dev)
        graph["hidden"].x = hidden
        graph["hidden"].node_type = "TriNodes"
        nh = hidden.shape[0]
        mark("hidden_mesh")

        # ---- decoder first: my data band as targets, replicated hidden sources; only the source ids travel
        # (the target row of a k-NN edge list is 0,0,0,1,1,1,... whatever the band).  It needs nothing from
        # the other ranks and its gather is the largest transfer of the build, so it is started before the
        # reference distance and the encoder search, which then run while the ids travel.
        ids = be.knn(hidden, self.data_local, self.k, **_global_sources(be)).reshape(-1)
        band_sizes = [(self.bounds[i + 1] - self.bounds[i]) * self.k for i in range(W)]
        dec_all = torch.empty((2, self.n_data * self.k), dtype=torch.int32, device=dev)
        dec_gather = comm.all_gather_rows_async([ids], band_sizes, out=[dec_all[0]])
        be.row_ids(self.n_data, self.k, dev, out=dec_all[1])
        mark("decoder_search")

        # ---- encoder: hidden targets of my band, data sources = my band + halo
        # (band limits by binary search on the device; one host synchronisation for all of them)
        radius_t = _reference_distance(be, self.resolution, hidden, rank_map, level_edges).to(dev).double() * self.cutoff_factor
        reach_t = radius_t * (1.0 + 1.0e-6) + 1.0e-6  # (the pad also covers the float32 rounding of the search keys)
        nd = self.data_local.shape[0]
        neg_hlat = (-hidden[:, 0]).contiguous()
        zero = torch.zeros((), dtype=torch.int64, device=dev)
        # hidden nodes are latitude-descending: band r = lat in (first_lat[r+1], first_lat[r]]
        h_lo = torch.searchsorted(neg_hlat, self.neg_upper.to(neg_hlat.dtype)) if r > 0 else zero
        h_hi = torch.searchsorted(neg_hlat, self.neg_lower.to(neg_hlat.dtype)) if r < W - 1 else zero + nh
        if not self.exchange_halo:
            # the sources of my targets: every data node within the reach of my latitude band -- a slice of
            # the coordinate array every rank holds, however many bands it spans
            keys = self.neg_lat_all
            s_lo = torch.searchsorted(keys, (-(self.first_lat[r] + reach_t)).to(keys.dtype)) if r > 0 else zero
            s_hi = (torch.searchsorted(keys, (-(self.first_lat[r + 1] - reach_t)).to(keys.dtype), right=True)
                    if r < W - 1 else zero + self.n_data)
            vals = torch.stack([radius_t] + [v.reshape(()).double() for v in (h_lo, h_hi, s_lo, s_hi)]).tolist()
            radius = vals[0]
            h_lo, h_hi, s_lo, s_hi = (int(v) for v in vals[1:])
            sources = self.data_all[s_lo:s_hi]
            g_lo = s_lo
        else:
            # coordinates treated as sharded: the rows beyond my band come from the neighbouring ranks
            sources, g_lo, vals = exchange_band_sources(comm, self.data_local, self.neg_lat, self.first_lat, reach_t,
                                                        self.d_lo, extra=[radius_t, h_lo, h_hi])
            radius, h_lo, h_hi = vals[0], int(vals[1]), int(vals[2])
        targets = hidden[h_lo:h_hi].contiguous()
        mark("reference_distance_and_halo")
        enc = be.radius_edges(sources, targets, radius)
        if g_lo:
            enc[0] += g_lo
        if h_lo:
            enc[1] += h_lo
        mark("encoder_search")
        enc_counts_ready = comm.all_gather_sizes_async([enc.shape[1]], dev)  # read once the processor is queued

        # ---- processor: replicated, while the edge lists travel
        proc = be.multiscale_edges(self.resolution, list(range(self.resolution + 1)), self.x_hops, rank_map,
                                   level_edges)
        data_all = self.data_all
        graph["data"].x = data_all
        graph["data"].node_type = "LatLonNodes"
        enc_name, proc_name, dec_name = ("data", "to", "hidden"), ("hidden", "to", "hidden"), ("hidden", "to", "data")
        for name in (enc_name, proc_name, dec_name):  # edge sets in the order a single-GPU build registers them
            graph[name]
        graph[proc_name].edge_index = proc
        graph[proc_name].edge_type = "MultiScaleEdges"
        if self.attributes:
            self._finish_attrs(graph[proc_name], *self._attrs(hidden, hidden, proc))
        mark("processor")

        enc_counts = enc_counts_ready()[:, 0].tolist()
        enc_all = torch.empty((2, sum(enc_counts)), dtype=enc.dtype, device=dev)
        enc_gather = comm.all_gather_rows_async([enc[0], enc[1]], enc_counts, out=[enc_all[0], enc_all[1]])

        # ---- the complete edge lists; attributes on them (same kernels and inputs as a single-GPU build);
        # the decoder's ids left first, so its attributes are evaluated while the encoder's lists still travel
        for name, gather, edge_index, src_x, tgt_x, etype in (
            (dec_name, dec_gather, dec_all, hidden, data_all, "KNNEdges"),
            (enc_name, enc_gather, enc_all, data_all, hidden, "CutOffEdges"),
        ):
            gather()
            mark("gather_wait_" + ("decoder" if name == dec_name else "encoder"))
            graph[name].edge_index = edge_index
            graph[name].edge_type = etype
            if self.attributes:
                self._finish_attrs(graph[name], *self._attrs(src_x, tgt_x, edge_index))
                mark("attributes_" + ("decoder" if name == dec_name else "encoder"))
        return graph

    def build(self) -> HeteroData:
        if self.comm is None:
            with settings.resident(settings.device_resident):
                return self._build_single()
        return self._build_sharded()


class PhaseTimer:
    """Device-side durations of the phases of a build: an event is recorded on the current stream at every
    ``mark``; ``milliseconds()`` (after a synchronisation) gives the time between consecutive marks, summed per
    name over all builds since ``reset``.  Waiting for a collective is stream work too, so it shows up in the
    phase that waits.  Off (no events) unless ``enabled``."""

    def __init__(self) -> None:
        self.enabled = False
        self._marks: list[tuple[str, torch.cuda.Event]] = []

    def reset(self) -> None:
        self._marks = []

    def mark(self, name: str) -> None:
        if self.enabled:
            event = torch.cuda.Event(enable_timing=True)
            event.record()
            self._marks.append((name, event))

    def milliseconds(self) -> dict[str, float]:
        out: dict[str, float] = {}
        for (_, before), (name, after) in zip(self._marks, self._marks[1:]):
            if name != "start":
                out[name] = out.get(name, 0.0) + before.elapsed_time(after)
        return out


def exchange_band_sources(comm: Communicator, data_local: torch.Tensor, neg_lat: torch.Tensor, first_lat: list[float],
                          reach_t: torch.Tensor, d_lo: int, extra: list[torch.Tensor] = ()):
    """The data rows a rank's cut-off search needs when every rank holds only its own latitude band: the band
    itself plus the rows of the adjacent bands within ``reach_t`` (0-dim device tensor, radians) of it, sent
    point-to-point by the two neighbours; when the reach is wider than a neighbouring band, all bands (all-gather).
    ``extra``: 0-dim device tensors whose values the caller wants on the host (they ride along with the one
    synchronisation this function needs).  Returns (sources, global id of sources[0], host values of ``extra``)."""
    r, W = comm.rank, comm.world
    nd = data_local.shape[0]
    dev = data_local.device
    zero = torch.zeros((), dtype=torch.int64, device=dev)
    if r > 0:  # the north neighbour's targets reach down to first_lat[r] - reach
        n_north = torch.searchsorted(neg_lat, (-(first_lat[r] - reach_t)).to(neg_lat.dtype), right=True)
    else:
        n_north = zero
    if r < W - 1:  # the south neighbour's targets reach up to first_lat[r+1] + reach (exclusive band top)
        s_start = torch.searchsorted(neg_lat, (-(first_lat[r + 1] + reach_t)).to(neg_lat.dtype), right=False)
    else:
        s_start = zero + nd
    vals = torch.stack([v.reshape(()).double() for v in (n_north, s_start, *extra)]).tolist()
    n_north, s_start = int(vals[0]), int(vals[1])
    one_hop = (r == 0 or n_north < nd) and (r == W - 1 or s_start > 0)
    sizes = comm.all_gather_sizes([int(one_hop), n_north, nd - s_start, nd], dev)
    if bool(sizes[:, 0].min()):
        from_north, from_south = comm.exchange_halo(
            data_local[:n_north], data_local[s_start:],
            int(sizes[r - 1, 2]) if r > 0 else 0, int(sizes[r + 1, 1]) if r < W - 1 else 0)
        sources = torch.cat([from_north, data_local, from_south], dim=0) if W > 1 else data_local
        return sources, d_lo - from_north.shape[0], vals[2:]
    # reach wider than a band: every rank takes all data rows
    return comm.all_gather_rows(data_local, sizes[:, 3].tolist()), 0, vals[2:]


class BandGraphBuilder:
    """The graph of a data grid too large to replicate: every rank holds ONE latitude band of the data nodes
    (``data_band``: rows ``first_id`` .. of a grid of ``n_data`` nodes ordered north -> south; the bands of the ranks
    are consecutive) and ends with the edges of that band -- nothing but halo rows and a few scalars travels.

    * hidden mesh (TriNodes of ``resolution``), its reference distance and the processor edges: replicated;
    * encoder (CutOffEdges data -> hidden): targets = the hidden nodes of the band's latitude range, sources =
      the band plus the rows its two neighbours send (``exchange_band_sources``);
    * decoder (KNNEdges hidden -> data): targets = the band;
    * edge ids are GLOBAL node ids; concatenating the ranks' encoder (decoder) lists in rank order gives the
      list of a single-GPU build, edge for edge;
    * attributes: evaluated per band; the global normalisation of EdgeLength (``unit-max``, ``unit-range`` or
      none) takes the maximum / minimum over the ranks (one all-reduce of two words), so the values equal a
      single-GPU build's bit for bit.  Directions are not normalised in this mode.

    With ``dist=None`` (one rank) the band is the whole grid."""

    def __init__(self, data_band: torch.Tensor, first_id: int, n_data: int, resolution: int, cutoff_factor: float = 0.6,
                 num_nearest_neighbours: int = 3, x_hops: int = 1, rank: int = 0, world_size: int = 1, dist=None,
                 backend=None, attributes: bool = True, edge_length_norm: str | None = "unit-max",
                 decoder: bool = True, processor: bool = True, overlap: bool = True):
        if edge_length_norm not in (None, "unit-max", "unit-range"):
            raise ValueError("a band build normalises edge lengths by their range only (None, unit-max, unit-range)")
        self.backend = _hip_ops if backend is None else backend
        self.data_band = data_band.contiguous()
        self.first_id, self.n_data = int(first_id), int(n_data)
        self.resolution, self.cutoff_factor, self.k, self.x_hops = resolution, cutoff_factor, num_nearest_neighbours, x_hops
        self.rank, self.world = rank, world_size
        self.attributes, self.len_norm = attributes, edge_length_norm
        self.decoder, self.processor = decoder, processor
        self.overlap = overlap  # second stream for the processor edges (and, on one rank, an early binning of the band)
        self._side = None
        self.timer = PhaseTimer()
        lat = self.data_band[:, 0]
        if lat.numel() > 1 and bool((lat[1:] > lat[:-1]).any()):
            raise ValueError("latitude-band sharding needs data nodes ordered north -> south")
        self.comm = Communicator(dist, rank, world_size) if (dist is not None and world_size >= 1) else None
        self.neg_lat = (-lat).contiguous()
        if self.comm is not None and world_size > 1:
            if lat.numel() == 0:
                raise ValueError("every rank needs at least one data node")
            # first latitude of every band: one float per rank, exchanged once (as the bits of the float32)
            mine = int(lat[0].cpu().view(torch.int32).item())
            bits = self.comm.all_gather_sizes([mine], lat.device)[:, 0].to(torch.int32)
            self.first_lat = bits.view(torch.float32).double().tolist()
            if any(a < b for a, b in zip(self.first_lat, self.first_lat[1:])):
                raise ValueError("the bands of the ranks must follow each other north -> south")
        else:
            self.first_lat = [float(lat[0])] if lat.numel() else [0.0]

    def _range_over_ranks(self, len_range: torch.Tensor) -> torch.Tensor:
        """Largest and smallest length over all ranks: both words of the range are ordered like the values
        (the second one holds the complement of the smallest length's bits), so one MAX all-reduce does it."""
        if self.comm is None or self.world == 1:
            return len_range
        head = len_range[:2].clone()
        if self.comm.host_staged:
            host = head.cpu()
            self.comm.dist.all_reduce(host, op=self.comm.dist.ReduceOp.MAX)
            head = host.to(len_range.device)
        else:
            self.comm.dist.all_reduce(head, op=self.comm.dist.ReduceOp.MAX)
        out = len_range.clone()
        out[:2] = head
        return out

    def _edge_attributes(self, store, src_x, tgt_x, local_index):
        be = self.backend
        if be is _hip_ops:
            length, direction, len_range = be.edge_attrs(src_x, tgt_x, local_index, len_range=True)
            len_range = self._range_over_ranks(len_range)
            store["edge_length"] = be.normalise_(length.reshape(-1, 1), self.len_norm, value_range=len_range)
        else:  # (a backend without the range: the CPU tests' oracle, one rank's worth of edges at a time)
            length, direction = be.edge_attrs(src_x, tgt_x, local_index)
            store["edge_length"] = be.normalise_band_(length.reshape(-1, 1), self.len_norm, self.comm)
        store["edge_dirs"] = direction

    def _processor_edges(self, graph, proc_name, hidden, rank_map, level_edges, mark) -> None:
        be = self.backend
        proc = be.multiscale_edges(self.resolution, list(range(self.resolution + 1)), self.x_hops, rank_map, level_edges)
        graph[proc_name].edge_index = proc
        graph[proc_name].edge_type = "MultiScaleEdges"
        mark("processor_search")
        if self.attributes:
            length, direction, len_range = (be.edge_attrs(hidden, hidden, proc, len_range=True) if be is _hip_ops
                                            else (*be.edge_attrs(hidden, hidden, proc), None))
            extra = {} if len_range is None else {"value_range": len_range}
            graph[proc_name]["edge_length"] = be.normalise_(length.reshape(-1, 1).contiguous(), self.len_norm, **extra)
            graph[proc_name]["edge_dirs"] = direction
            mark("processor_attributes")

    def build(self) -> HeteroData:
        be, r, W = self.backend, self.rank, self.world
        dev = self.data_band.device
        mark = self.timer.mark
        mark("start")
        graph = HeteroData()
        nd = self.data_band.shape[0]
        # Second stream (not while the phases are being timed: they are phases of one stream).  On one rank the band
        # is all there is to bin, and the binning needs the cut-off radius only for the size of its cells: it is
        # started now, beside the generation of the mesh, for the radius a quasi-uniform mesh of this size will have
        # (see ShardedGraphBuilder._build_single), and taken later if its cells are within the accepted range.
        two_streams = self.overlap and be is _hip_ops and dev.type == "cuda" and not self.timer.enabled
        main = side = early = None
        if two_streams:
            main = torch.cuda.current_stream(dev)
            if self._side is None:
                self._side = torch.cuda.Stream(device=dev)
            side = self._side
            side.wait_stream(main)
            if W == 1 and nd > 0:
                guess = self.cutoff_factor * 1.185 * math.sqrt(4.0 * math.pi / (10 * 4 ** self.resolution + 2))
                with torch.cuda.stream(side):
                    early = be.bin_points(self.data_band, be.radius_cell(nd, guess))
        _xyz, _faces, hidden, _order, rank_map, level_edges = be.icosphere(self.resolution)
        hidden = hidden.to(dev)
        nh = hidden.shape[0]
        graph["hidden"].x = hidden
        graph["hidden"].node_type = "TriNodes"
        graph["data"].x = self.data_band
        graph["data"].node_type = "LatLonNodes"
        graph["data"]["_first_id"] = self.first_id
        mark("hidden_mesh")
        enc_name, proc_name, dec_name = ("data", "to", "hidden"), ("hidden", "to", "hidden"), ("hidden", "to", "data")
        for name in (enc_name, proc_name, dec_name):
            graph[name]
        if two_streams and self.processor:  # the processor edges need the mesh and nothing else
            side.wait_stream(main)
            with torch.cuda.stream(side):
                self._processor_edges(graph, proc_name, hidden, rank_map, level_edges, mark)

        # ---- sources of the encoder: my band plus the rows my neighbours send
        radius_t = _reference_distance(be, self.resolution, hidden, rank_map, level_edges).to(dev).double() * self.cutoff_factor
        reach_t = radius_t * (1.0 + 1.0e-6) + 1.0e-6  # (the pad also covers the float32 rounding of the search keys)
        neg_hlat = (-hidden[:, 0]).contiguous()
        zero = torch.zeros((), dtype=torch.int64, device=dev)
        nd = self.data_band.shape[0]
        if W > 1:
            upper = torch.tensor(-self.first_lat[r], dtype=neg_hlat.dtype, device=dev)
            lower = torch.tensor(-self.first_lat[min(r + 1, W - 1)], dtype=neg_hlat.dtype, device=dev)
            h_lo = torch.searchsorted(neg_hlat, upper) if r > 0 else zero
            h_hi = torch.searchsorted(neg_hlat, lower) if r < W - 1 else zero + nh
            sources, g_lo, vals = exchange_band_sources(self.comm, self.data_band, self.neg_lat, self.first_lat, reach_t,
                                                        self.first_id, extra=[radius_t, h_lo, h_hi])
            radius, h_lo, h_hi = vals[0], int(vals[1]), int(vals[2])
        else:
            sources, g_lo, radius, h_lo, h_hi = self.data_band, self.first_id, float(radius_t.item()), 0, nh
        mark("halo_exchange")
        # one binning of those rows serves the encoder (as its sources) and, when they are little more than the band
        # itself, the decoder (as its targets: the few halo rows are searched too and dropped)
        kept = None
        if be is _hip_ops and sources.shape[0] > 0:
            cell = be.radius_cell(sources.shape[0], radius)
            if early is not None and sources is self.data_band and 0.8 * cell <= early.cell <= 2.0 * cell:
                kept = early
            else:
                kept = be.bin_points(sources, cell)
            mark("binning")
        del early
        before = self.first_id - g_lo  # halo rows in front of my band

        # ---- decoder: hidden sources, my band as targets
        if self.decoder:
            own_cell = ((2.0 * (self.k + 1) + 2.0) * 4.0 / max(nh, 1)) ** 0.5
            binned = kept is not None and sources.shape[0] <= nd + nd // 8 and kept.cell <= 1.5 * own_cell
            if binned and sources.shape[0] == nd:  # no halo rows: the search writes the edge list itself
                dec = be.knn_edges(hidden, None, self.k, tgt_binning=kept, **_global_sources(be))
            else:
                dec = torch.empty((2, nd * self.k), dtype=torch.int32, device=dev)
                if binned:
                    ids = be.knn(hidden, None, self.k, tgt_binning=kept, **_global_sources(be))
                    dec[0] = ids[before:before + nd].reshape(-1)
                else:
                    dec[0] = be.knn(hidden, self.data_band, self.k, **_global_sources(be)).reshape(-1)
                be.row_ids(nd, self.k, dev, out=dec[1])
            mark("decoder_search")
            if self.attributes:
                self._edge_attributes(graph[dec_name], hidden, self.data_band, dec)
                mark("decoder_attributes")
            if self.first_id:
                dec[1] += self.first_id
            graph[dec_name].edge_index = dec
            graph[dec_name].edge_type = "KNNEdges"
        else:
            del graph._stores[dec_name]

        # ---- encoder: hidden targets of my latitude range
        targets = hidden[h_lo:h_hi].contiguous()
        enc = (be.radius_edges(None, targets, radius, src_binning=kept) if kept is not None
               else be.radius_edges(sources, targets, radius))
        mark("encoder_search")
        if h_lo:  # (a pass over the list each: not made for a band that starts at the first node)
            enc[1] += h_lo
        if self.attributes:
            self._edge_attributes(graph[enc_name], sources, hidden, enc)
            mark("encoder_attributes")
        if g_lo:
            enc[0] += g_lo
        graph[enc_name].edge_index = enc
        graph[enc_name].edge_type = "CutOffEdges"
        graph[enc_name]["_radius"] = radius
        del kept

        # ---- processor: replicated
        if not self.processor:
            del graph._stores[proc_name]
        elif two_streams:
            main.wait_stream(side)  # (queued on the second stream, behind the mesh)
        else:
            self._processor_edges(graph, proc_name, hidden, rank_map, level_edges, mark)
        return graph
