// Spatial binning on the sphere, k-nearest-neighbour, nearest-neighbour reference distance and
// radius (cut-off) queries for gfx950.
//
// Design
//  * Points are binned into the latitude-band grid of ag_common.hpp by a counting sort: one pass
//    takes a slot in the point's cell (atomic counter), a prefix sum turns the counts into cell
//    starts, one pass scatters (unit vector, id) records.  The order inside a cell is arbitrary;
//    no result depends on it (ties are broken on ids, radius rows are sorted before they are
//    written).
//  * k-NN, stage 1 (ag_knn_cell_kernel): one wave per target cell.  The sources of the 3 bands
//    around the cell are staged once into an LDS tile as float32 unit vectors; every lane owns
//    one target of the cell and scans the tile (broadcast LDS reads) keeping a sorted register
//    list of its CAP >= k+2 best float32 chord^2 values.  If the float32 gaps around the k-th entry
//    exceed the float32 error bound the list is final; otherwise the lane re-ranks the entries
//    up to the end of the ambiguous cluster with scikit-learn's float64 haversine formula.  Only
//    targets whose k-th chord is not provably inside the tile's coverage, or whose ambiguous
//    cluster runs past the list, are queued for stage 2.
//  * k-NN, stage 2 (ag_knn_exact_kernel): one wave per queued target, float64 rdist over a window
//    that doubles until it provably holds the k nearest; the k smallest (rdist, id) pairs are
//    extracted by k wavefront min-reductions.
//  * Radius query: one wave per target; lanes stride over the contiguous runs of binned sources
//    of the covering cells (coalesced 16-byte loads), float32 chord test with a float64
//    fallback inside the error band, ballot-compacted rows, sorted by source id in LDS.
#include <algorithm>
#include <mutex>
#include <stdlib.h>

#include <atomic>

#include "ag_common.hpp"

// ---------------------------------------------------------------------------------------------
// grid construction
// ---------------------------------------------------------------------------------------------
// What a binning wants to know about the order of its points before it counts them (ag_bin_points): one block of
// the band-table launch looks at 1024 pairs of consecutive points spread over the array and writes, to *order,
// whether most pairs lie within a cell of each other (AG_ORDER_NEIGHBOURS: a grid stored row by row, a mesh sorted
// by latitude) or not (AG_ORDER_SCATTERED: a spiral, a random order).
#define AG_ORDER_NEIGHBOURS 0
#define AG_ORDER_SCATTERED 1
struct AgBinProbe {
  const float2* pts = nullptr;  // null: nothing to probe (a grid that only counts)
  int64_t n = 0;
  int* order = nullptr;
};

// Block 0 computes the band tables; every block clears its share of the cell counters and of the
// state words of the scan that turns the counters into cell starts.
__global__ __launch_bounds__(1024) void ag_band_table_kernel(int nb, double dlat, int mult, int* __restrict__ ncols,
                                                             int* __restrict__ off, double* __restrict__ win,
                                                             int* __restrict__ cell_count, int64_t ncount,
                                                             unsigned long long* __restrict__ scan_state,
                                                             int64_t nstate, unsigned long long state_fill,
                                                             AgBinProbe probe) {
  const int64_t gtid = blockIdx.x * int64_t(blockDim.x) + threadIdx.x, gsize = int64_t(gridDim.x) * blockDim.x;
  for (int64_t i = gtid; i < ncount; i += gsize) cell_count[i] = 0;
  // state of the scan of the counters: zero (anything else only from the test hook ag_debug_poison_scans)
  for (int64_t i = gtid; i < nstate; i += gsize) scan_state[i] = state_fill;
  __shared__ int partial[1024];
  const int tid = threadIdx.x;
  if (probe.pts != nullptr && blockIdx.x == (gridDim.x > 1 ? 1u : 0u)) {
    // a pair (p, p + 1) per thread, at equal distances over the array; "within a cell": both differences below the
    // band height
    const int64_t pairs = probe.n - 1;
    const int64_t samples = pairs < 1024 ? pairs : 1024;
    bool close = false;
    if (tid < samples) {
      const int64_t p = (pairs / samples) * tid;
      const float2 a = probe.pts[p], b = probe.pts[p + 1];
      const float reach = static_cast<float>(dlat);
      float dlon = fabsf(a.y - b.y);
      dlon = fminf(dlon, fabsf(static_cast<float>(AG_TWO_PI) - dlon));
      close = fabsf(a.x - b.x) <= reach && dlon * __cosf(a.x) <= reach;
    }
    if (tid == 0) partial[0] = 0;
    __syncthreads();
    const int in_wave = __popcll(__ballot(close));
    if ((tid & 63) == 0 && in_wave) atomicAdd(&partial[0], in_wave);
    __syncthreads();
    if (tid == 0) *probe.order = (2 * int64_t(partial[0]) >= samples) ? AG_ORDER_NEIGHBOURS : AG_ORDER_SCATTERED;
    __syncthreads();
  }
  if (blockIdx.x != 0) return;
  const int per = (nb + 1023) / 1024;
  const int b_begin = tid * per, b_end = min(nb, b_begin + per);
  const double rg = 0.98 * dlat;
  const double pad = 1.0e-7 + 1.0e-6 * rg;
  int sum = 0;
  for (int b = b_begin; b < b_end; ++b) {
    const double lat_hi = AG_HALF_PI - b * dlat, lat_lo = lat_hi - dlat;
    const double mid = AG_HALF_PI - (b + 0.5) * dlat;
    int nc = static_cast<int>(floor(AG_TWO_PI * ag_cos_f64(mid) / dlat));
    nc = max(nc, 1);
    nc = mult >= 0 ? nc * mult : max(1, nc / (-mult));  // negative: merge -mult cells into one
    ncols[b] = nc;
    sum += nc;
    // longitude half-width of a cap of radius rg centred anywhere in the band (see ag_for_each_range)
    const double phit = fmax(fabs(lat_lo), fabs(lat_hi));
    double w = -1.0;
    if (phit + rg + pad < AG_HALF_PI && rg < AG_HALF_PI) {
      const double q = ag_sin_f64(rg) / ag_cos_f64(phit);
      if (q < 1.0) w = asin(q) * (1.0 + 1.0e-6) + pad;
    }
    win[b] = w;
  }
  partial[tid] = sum;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    int v = (tid >= d) ? partial[tid - d] : 0;
    __syncthreads();
    partial[tid] += v;
    __syncthreads();
  }
  int run = partial[tid] - sum;
  for (int b = b_begin; b < b_end; ++b) {
    off[b] = run;
    run += ncols[b];
  }
  if (tid == 1023) off[nb] = partial[1023];
}

// band of a cell: binary search in the band offsets
__device__ __forceinline__ int ag_band_of_cell(const AgGrid& g, int c) {
  int lo = 0, hi = g.nb - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (g.off[mid] <= c) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// Which points a block takes: tiles of 256 * U consecutive points, U per thread (the thread's points are 256 apart,
// every wavefront reads and writes whole lines).  U = 4 for sets of 2^22 points and more, 8 from 2^25: the U loads
// of a thread, its U counter updates and its U stores are in flight together, and a kernel that waits for memory
// most of its time gets U times the bytes per wait.
// Blocks are handed to the eight dies in turn (block b to die b % 8), and every die has a second-level cache of
// its own.  Points that follow each other in the caller's order (the rows of a grid) fall into the same cells, a
// few points each: the record lines of those cells are completed by several blocks, and they are completed in ONE
// cache, and leave it whole, when those blocks run on the same die.  So the tiles are dealt out in chunks: `chunk`
// consecutive tiles to one die, the next chunk to the next die.  The launch has a multiple of eight blocks, and
// every block walks b, b + gridDim, ... (the same die every time).
struct AgTileDeal {
  int64_t ntiles_padded;  // a multiple of 8 * chunk; tiles beyond the last point hold nothing
  int chunk;
  int unroll;             // U: 1, 4 or 8
};
static AgTileDeal ag_tile_deal(int64_t n) {
  const int unroll = n >= (int64_t(1) << 25) ? 8 : (n >= (int64_t(1) << 22) ? 4 : 1);
  const int64_t per_tile = 256 * unroll, ntiles = (n + per_tile - 1) / per_tile;
  int chunk = 1;
  while (chunk < 1024 && int64_t(chunk) * 2 * 64 <= ntiles) chunk *= 2;  // (at least 8 rounds over the dies)
  const int64_t round = 8 * int64_t(chunk);
  return AgTileDeal{(ntiles + round - 1) / round * round, chunk, unroll};
}
static int ag_tile_blocks(const AgTileDeal& d) { return static_cast<int>(d.ntiles_padded < 16384 ? d.ntiles_padded : 16384); }
__device__ __forceinline__ int64_t ag_dealt_tile(int64_t vb, int chunk) {
  const int64_t round = 8 * int64_t(chunk), r = vb % round;
  return ((vb / round) * 8 + (r & 7)) * chunk + (r >> 3);
}

// Runs of a wavefront: consecutive lanes with the same cell.  Neighbouring points usually share a cell, so the first
// lane of every run speaks for the run (all runs of the wave in the same instruction).  Lanes past the end of the
// array (the highest lanes of the last wave) carry the cell -1: a run of their own that takes part in nothing.
struct AgLaneRun {
  bool head;    // this lane is the first of its run
  int first;    // lane of the head of this lane's run
  int length;   // (heads only) lanes in the run
};
__device__ __forceinline__ AgLaneRun ag_lane_run(int cell, int lane) {
  const int prev = __shfl_up(cell, 1, 64);
  AgLaneRun r;
  r.head = (lane == 0) || (prev != cell);
  const unsigned long long heads = __ballot(r.head);
  r.first = 63 - __clzll(static_cast<long long>(heads & (~0ull >> (63 - lane))));
  const unsigned long long above = (lane == 63) ? 0ull : ((heads >> (lane + 1)) << (lane + 1));
  r.length = (above ? (__ffsll(static_cast<long long>(above)) - 1) : 64) - lane;
  return r;
}

// The cell of every point, and how many points every cell holds.  Points stored next to their neighbours: one
// counter update per run, and nobody waits for the old value.  Points stored in a scattered order (every lane a run of
// its own): the old value IS the point's slot in its cell, kept in slot_of, so that the scatter pass needs no second
// round of updates -- with a point per update the updates are what a binning costs (510 million points of a spiral:
// 23 ms per round).
template <int U>
__global__ __launch_bounds__(256) void ag_bin_count_kernel(AgGrid g, const float2* __restrict__ pts, int64_t n,
                                                           int* __restrict__ cell_of, int* __restrict__ slot_of,
                                                           int* __restrict__ counter, const int* __restrict__ order,
                                                           AgTileDeal deal) {
  const int lane = ag_lane();
  const bool scattered = *order == AG_ORDER_SCATTERED;
  for (int64_t vb = blockIdx.x; vb < deal.ntiles_padded; vb += gridDim.x) {
    const int64_t first = ag_dealt_tile(vb, deal.chunk) * (256 * U) + threadIdx.x;
    if (first - threadIdx.x >= n) continue;  // (the whole block)
    float2 p[U];
#pragma unroll
    for (int u = 0; u < U; ++u) p[u] = pts[min(first + u * 256, n - 1)];
    if (!scattered) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t i = first + u * 256;
        const bool on = i < n;
        const int c = on ? ag_cell_of(g, static_cast<double>(p[u].x), static_cast<double>(p[u].y)) : -1;
        const AgLaneRun run = ag_lane_run(c, lane);
        if (run.head && on) atomicAdd(&counter[c], run.length);  // (nobody waits for the old value)
        if (on) cell_of[i] = c;
      }
    } else {
      int c[U], base[U], head_lane[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool on = first + u * 256 < n;
        c[u] = on ? ag_cell_of(g, static_cast<double>(p[u].x), static_cast<double>(p[u].y)) : -1;
        const AgLaneRun run = ag_lane_run(c[u], lane);
        head_lane[u] = run.first;
        base[u] = 0;
        if (run.head && on) base[u] = atomicAdd(&counter[c[u]], run.length);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int start = __shfl(base[u], head_lane[u], 64);
        const int64_t i = first + u * 256;
        if (i < n) {
          cell_of[i] = c[u];
          slot_of[i] = start + (lane - head_lane[u]);
        }
      }
    }
  }
}

// The scan of the counters leaves the start of cell c in cell_start[c] and their total in cell_start[ncells]: every
// point was counted once, so it is n.  The points then take their places.  Stored next to their neighbours: the head
// of a run takes the run's places off the END of what is left of its cell -- the counters, which the scan read but
// did not change, are counted down, so nothing has to be cleared for this pass -- and the run's ids go to the cell's
// start plus what was left minus the run; only the ids are scattered (4 bytes a point: the lines
// of a cell are completed by the blocks of several rows of the caller's grid, and lines that leave the cache half
// written are what a scatter pays for), the 16-byte records are then written in their own order, whole lines at a
// time, by ag_bin_records_kernel.  Stored in a scattered order: every point has its slot from the counting pass and
// puts its id and its record there itself (a gather pass would meet the same scattered lines again).
// A total other than n means that the scan gave up (state words that were not clear) and the starts cannot be
// trusted -- some places would never be written, and the searches follow the ids they hold.  The binning is then
// left EMPTY but sound: every cell without points, the ids (and with them the records) in the points' own order
// (all initialised, all ids valid), and the library's status word set, which the host turns into an error at its
// next synchronisation.
#define AG_STATUS_BINNING_FAILED 2
template <int U>
__global__ __launch_bounds__(256) void ag_bin_scatter_kernel(const float2* __restrict__ pts, int64_t n,
                                                             const int* __restrict__ cell_of,
                                                             const int* __restrict__ slot_of, int* __restrict__ left,
                                                             int* __restrict__ cell_start, int64_t ncells,
                                                             float4* __restrict__ out_pts, int* __restrict__ out_ids,
                                                             int* __restrict__ order, int* __restrict__ status,
                                                             AgTileDeal deal) {
  if (cell_start[ncells] != static_cast<int>(n)) {  // (the same for every thread of the launch: nobody writes that word)
    const int64_t gtid = blockIdx.x * int64_t(blockDim.x) + threadIdx.x, gsize = int64_t(gridDim.x) * blockDim.x;
    for (int64_t c = gtid; c < ncells; c += gsize) cell_start[c] = 0;
    for (int64_t i = gtid; i < n; i += gsize) out_ids[i] = static_cast<int>(i);
    if (gtid == 0 && status != nullptr) *status = AG_STATUS_BINNING_FAILED;
    return;  // (the records follow the ids: ag_bin_records_kernel, whatever the order of the points)
  }
  const int lane = ag_lane();
  const bool scattered = *order == AG_ORDER_SCATTERED;
  for (int64_t vb = blockIdx.x; vb < deal.ntiles_padded; vb += gridDim.x) {
    const int64_t first = ag_dealt_tile(vb, deal.chunk) * (256 * U) + threadIdx.x;
    if (first - threadIdx.x >= n) continue;  // (the whole block)
    int c[U], place[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = first + u * 256;
      c[u] = i < n ? cell_of[i] : -1;
    }
    if (!scattered) {
      int head_lane[U], start[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const AgLaneRun run = ag_lane_run(c[u], lane);
        head_lane[u] = run.first;
        place[u] = 0;
        start[u] = c[u] >= 0 ? cell_start[c[u]] : 0;
        if (run.head && c[u] >= 0) place[u] = atomicSub(&left[c[u]], run.length) - run.length;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        place[u] = start[u] + __shfl(place[u], head_lane[u], 64) + (lane - head_lane[u]);
        const int64_t i = first + u * 256;
        if (i < n) out_ids[place[u]] = static_cast<int>(i);
      }
    } else {
      float2 p[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {  // (lanes past the end read the last point again and write nothing)
        const int64_t i = min(first + u * 256, n - 1);
        place[u] = slot_of[i];
        if (out_pts != nullptr) p[u] = pts[i];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) place[u] += c[u] >= 0 ? cell_start[c[u]] : 0;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t i = first + u * 256;
        if (i >= n) continue;
        out_ids[place[u]] = static_cast<int>(i);
        if (out_pts != nullptr) {
          const float3 v = ag_unit_vec(p[u].x, p[u].y);
          out_pts[place[u]] = make_float4(v.x, v.y, v.z, __int_as_float(static_cast<int>(i)));
        }
      }
    }
  }
}

// The records of a binning, in the order of the binning: the unit vector of the point every id names, and the id.
// (The points of a cell come from a few rows of the caller's grid, a short run of neighbours from each: the reads
// are served by the caches, the writes are whole lines.)  After a binning that failed the ids are the identity.
template <int U>
__global__ __launch_bounds__(256) void ag_bin_records_kernel(const float2* __restrict__ pts, int64_t n,
                                                             const int* __restrict__ ids, float4* __restrict__ out_pts,
                                                             const int* __restrict__ order,
                                                             const int* __restrict__ cell_start, int64_t ncells,
                                                             AgTileDeal deal) {
  // (points in a scattered order: the scatter pass wrote the records too -- unless the binning failed, and the ids
  // it left are the identity)
  if (*order == AG_ORDER_SCATTERED && cell_start[ncells] == static_cast<int>(n)) return;
  for (int64_t vb = blockIdx.x; vb < deal.ntiles_padded; vb += gridDim.x) {
    const int64_t first = ag_dealt_tile(vb, deal.chunk) * (256 * U) + threadIdx.x;
    if (first - threadIdx.x >= n) continue;  // (the whole block)
    int id[U];
    float2 p[U];
#pragma unroll
    for (int u = 0; u < U; ++u) id[u] = ids[min(first + u * 256, n - 1)];
#pragma unroll
    for (int u = 0; u < U; ++u) p[u] = pts[min(max(id[u], 0), static_cast<int>(n - 1))];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t pos = first + u * 256;
      if (pos >= n) continue;
      const float3 v = ag_unit_vec(p[u].x, p[u].y);
      out_pts[pos] = make_float4(v.x, v.y, v.z, __int_as_float(id[u]));
    }
  }
}

// How many points every cell of a (coarse) grid holds -- the occupancy a bounded search consults before it looks at
// the sources themselves.  (Consecutive points mostly share a cell: the first lane of a run counts for the run.)
__global__ void ag_occupancy_count_kernel(AgGrid g, const float2* __restrict__ pts, int64_t n, int* __restrict__ count) {
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
    const float2 p = pts[i];
    const int c = ag_cell_of(g, static_cast<double>(p.x), static_cast<double>(p.y));
    const int lane = ag_lane();
    const int prev = __shfl_up(c, 1, 64);
    const bool head = (lane == 0) || (prev != c);
    const unsigned long long heads = __ballot(head);
    const unsigned long long active = __ballot(1);
    const unsigned long long above = (lane == 63) ? 0ull : ((heads >> (lane + 1)) << (lane + 1));
    const int run_end = above ? (__ffsll(static_cast<long long>(above)) - 1) : (64 - __clzll(static_cast<long long>(active)));
    if (head) atomicAdd(&count[c], run_end - lane);
  }
}

int ag_grid_bands(double delta) {
  if (!(delta > 0.0)) delta = AG_PI;
  double nb = floor(AG_PI / delta);
  if (nb < 1.0) nb = 1.0;
  if (nb > 32768.0) nb = 32768.0;
  return static_cast<int>(nb);
}

static int64_t ag_grid_cells_ub(int nb, int mult = 1) {
  const double dlat = AG_PI / nb;
  return (static_cast<int64_t>(4.0 * AG_PI / (dlat * dlat) * 1.02) + nb + 64) * (mult > 1 ? mult : 1);
}

struct AgBinCarve {
  int *ncols, *off, *count, *cell_start, *order, *cell_of, *slot_of, *ids;
  double* win;
  float4* pts;
  unsigned long long* scan_state;
  int64_t scan_words;
};

static void ag_bin_carve(AgCarver& ws, int64_t n, int nb_reserve, int mult, bool want_pts, AgBinCarve* c) {
  const int64_t cells = ag_grid_cells_ub(nb_reserve, mult) + 1;
  c->ncols = ws.take<int>(nb_reserve);
  c->off = ws.take<int>(nb_reserve + 1);
  c->win = ws.take<double>(nb_reserve);
  c->count = ws.take<int>(cells);
  c->cell_start = ws.take<int>(cells);
  c->order = ws.take<int>(4);
  c->cell_of = ws.take<int>(n);
  c->slot_of = ws.take<int>(n);
  c->ids = ws.take<int>(n);
  c->scan_words = static_cast<int64_t>(ag_chained_scan_state_words_i32(cells));
  c->scan_state = ws.take<unsigned long long>(c->scan_words);
  c->pts = want_pts ? ws.take<float4>(n) : nullptr;
}

static void ag_bin_describe(const AgBinCarve& c, int64_t n, int nb, int mult, AgBinned* out) {
  out->grid.nb = nb;
  out->grid.dlat = AG_PI / nb;
  out->grid.inv_dlat = nb / AG_PI;
  out->grid.ncols = c.ncols;
  out->grid.off = c.off;
  out->grid.win = c.win;
  out->n = n;
  out->ncells_ub = ag_grid_cells_ub(nb, mult);
  out->pts = c.pts;
  out->ids = c.ids;
  out->cell_of = c.cell_of;
  out->cell_start = c.cell_start;
}

int ag_bin_points(AgCarver& ws, const float* d_pts, int64_t n, int nb, int nb_reserve, int mult, bool want_pts,
                  hipStream_t stream, AgBinned* out) {
  if (nb_reserve < nb) nb_reserve = nb;
  AgBinCarve c;
  ag_bin_carve(ws, n, nb_reserve, mult, want_pts, &c);
  if (out == nullptr) return AG_OK;  // sizing pass
  ag_bin_describe(c, n, nb, mult, out);
  const int64_t ncells_ub = out->ncells_ub;
  AgBinProbe probe;
  probe.pts = reinterpret_cast<const float2*>(d_pts);
  probe.n = n;
  probe.order = c.order;
  AG_KERNEL("ag_band_table_kernel", ag_band_table_kernel, dim3(ag_blocks(ncells_ub + 1, 4096, 256)), dim3(1024), stream,
            nb, out->grid.dlat, mult, c.ncols, c.off, c.win, c.count, ncells_ub + 1, c.scan_state, c.scan_words,
            ag_debug_scans_poisoned() ? 0x40000000ull : 0ull,  // (poisoned: every tile draws a ticket beyond the last tile)
            probe);
  AG_LAUNCH_CHECK();
  int rc = AG_OK;
  const AgTileDeal deal = ag_tile_deal(n);
  const int blocks = ag_tile_blocks(deal);  // (a multiple of 8, blocks of 256 threads: see ag_dealt_tile)
#define AG_BIN_LAUNCH(name, kernel, ...)                                                         \
  do {                                                                                          \
    if (deal.unroll == 8) {                                                                     \
      AG_KERNEL(name, kernel<8>, dim3(blocks), dim3(256), stream, __VA_ARGS__);                 \
    } else if (deal.unroll == 4) {                                                              \
      AG_KERNEL(name, kernel<4>, dim3(blocks), dim3(256), stream, __VA_ARGS__);                 \
    } else {                                                                                    \
      AG_KERNEL(name, kernel<1>, dim3(blocks), dim3(256), stream, __VA_ARGS__);                 \
    }                                                                                           \
  } while (0)
  AG_BIN_LAUNCH("ag_bin_count_kernel", ag_bin_count_kernel, out->grid, reinterpret_cast<const float2*>(d_pts), n,
                c.cell_of, c.slot_of, c.count, c.order, deal);
  AG_LAUNCH_CHECK();
  rc = ag_chained_scan_i32(c.count, c.cell_start, ncells_ub, c.scan_state, stream);  // total -> cell_start[ncells_ub]
  if (rc) return rc;
  AG_BIN_LAUNCH("ag_bin_scatter_kernel", ag_bin_scatter_kernel, reinterpret_cast<const float2*>(d_pts), n, c.cell_of,
                c.slot_of, c.count, c.cell_start, ncells_ub, c.pts, c.ids, c.order, ag_status_word(), deal);
  AG_LAUNCH_CHECK();
  if (c.pts != nullptr) {
    AG_BIN_LAUNCH("ag_bin_records_kernel", ag_bin_records_kernel, reinterpret_cast<const float2*>(d_pts), n, c.ids, c.pts,
                  c.order, c.cell_start, ncells_ub, deal);
    AG_LAUNCH_CHECK();
  }
#undef AG_BIN_LAUNCH
  return AG_OK;
}

// ---------------------------------------------------------------------------------------------
// A binning the caller keeps: the same layout ag_bin_points carves from a workspace, in a buffer of its
// own, so that several searches over one node set (the cut-off query that reads it as sources, the k-NN
// query that takes it as targets) bin it once.  The buffer is described again from (n, cell) by
// whoever uses it; nothing but device data lives in it.
// ---------------------------------------------------------------------------------------------
struct AgKeptBinning {
  const void* buffer = nullptr;  // null: none
  double cell = 0.0;
};

static int ag_kept_bands_reserve(int64_t n) { return ag_grid_bands(sqrt(4.0 * AG_PI / static_cast<double>(n > 0 ? n : 1))); }

// bands of a kept binning with cells of about `cell` radians: never finer than the buffer is laid out for
static int ag_kept_bands(int64_t n, double cell) {
  const int nb = ag_grid_bands(cell), cap = ag_kept_bands_reserve(n);
  return nb < cap ? nb : cap;
}

static void ag_kept_describe(const AgKeptBinning& kept, int64_t n, AgBinned* out) {
  AgCarver c(const_cast<void*>(kept.buffer));
  AgBinCarve bc;
  ag_bin_carve(c, n, ag_kept_bands_reserve(n), 1, true, &bc);
  ag_bin_describe(bc, n, ag_kept_bands(n, kept.cell), 1, out);
}

extern "C" size_t ag_binning_bytes(int64_t n) {
  if (n < 1) return 0;
  AgCarver c(nullptr);
  AgBinCarve bc;
  ag_bin_carve(c, n, ag_kept_bands_reserve(n), 1, true, &bc);
  return c.total();
}

extern "C" int ag_binning_build(const float* d_pts, int64_t n, double cell, void* d_binning, size_t binning_bytes,
                                void* stream) {
  AG_REQUIRE(n >= 1 && n < (int64_t(1) << 31), "ag_binning_build: node count %lld out of range", static_cast<long long>(n));
  AG_REQUIRE(cell > 0.0, "ag_binning_build: the cell size must be positive (got %g)", cell);
  AG_REQUIRE(d_pts && d_binning, "ag_binning_build: null pointer");
  if (binning_bytes < ag_binning_bytes(n)) return ag_fail(AG_ERR_WORKSPACE, "ag_binning_build: buffer too small");
  AgCarver c(d_binning);
  AgBinned out;
  const int reserve = ag_kept_bands_reserve(n);
  return ag_bin_points(c, d_pts, n, ag_kept_bands(n, cell), reserve, 1, true, static_cast<hipStream_t>(stream), &out);
}

// ---------------------------------------------------------------------------------------------
// k-NN stage 1: target sub-cell tiles in LDS, float32 chord^2, register top-CAP (float32 only)
// ---------------------------------------------------------------------------------------------
#define AG_TILE 256        // staged sources per wave and pass
#define AG_WAVES_PER_BLOCK 4
// The targets stage 1 hands to the float64 re-ranking are appended to queues, one atomic per wave.  Atomics on ONE
// word are served one after the other by the memory side, about 11 ns each: with a single queue the 7.8 million
// waves of a 5e8-target search waited 47 ms for their slots (the arithmetic of the kernel takes 9 ms).  Hence many
// queues, taken in turn by consecutive waves, with their fill words a cache line apart; every queue has room for
// the targets of all the waves that can use it.
#define AG_KNN_RQ 256
#define AG_KNN_RQ_STRIDE 32

// A search reads what a binning and prefix sums upstream of it wrote: cell starts, candidate lists, queues.  When
// one of those gave up (see ag_bin_scatter_kernel, ag_chained_scan_kernel) the status word of the device is set,
// and until the host has taken it (ag_status_take, which turns it into an error) the kernels of the searches do
// nothing at all: whatever lies in their intermediate buffers is never followed.  Their outputs are then
// unwritten; the host discards them with the error.
__device__ __forceinline__ bool ag_aborted(const int* status) {
  return status != nullptr && __builtin_nontemporal_load(status) != 0;
}

struct AgKnnArgs {
  AgGrid sgrid;              // source grid
  AgGrid tgrid;              // target grid: same bands, `tmult` times the longitude cells
  const float4* spts;        // binned sources
  const int* sstart;         // source cell starts
  const int* tids;           // binned target ids
  const float4* tpts;        // binned target records (unit vector, id) when the targets come from a kept binning, else null
  int foreign_targets;       // 1: the target grid is a kept binning's own grid (bands and cells unrelated to the sources')
  const int* tstart;         // target cell starts
  const float2* tgt;         // target coordinates, original order
  const float2* src;         // source coordinates, original order
  int64_t ns, nt;
  int k;
  double rg;                 // angular radius a tile is guaranteed to cover
  float rg_chord2;           // conservative float32 chord^2 of rg
  int32_t* out_src;          // [nt*k]
  int32_t* out_tgt;          // [nt*k] or null: the target of every entry of out_src (the second row of an edge list)
  double* out_dist;          // [nt*k] or null
  int* cand_n;               // [cand_slots] per re-rank queue slot: number of candidates to re-rank in float64
  int* cand_t;               // [cand_slots] per re-rank queue slot: the binned target
  int* cand_ids;             // [CAP][cand_slots] candidate ids by queue slot, coalesced per rank
  int64_t cand_slots;        // slots of the re-rank queues together
  int* rq_count;             // [rq * AG_KNN_RQ_STRIDE] fill of every re-rank queue
  int rq;                    // re-rank queues (a power of two); queue q owns the slots [q * rq_cap, (q + 1) * rq_cap)
  int rq_cap;
  int* queue;                // [nt] targets left to stage 2
  int* queue_count;
  const struct AgWindow* windows;  // [nwin] one descriptor per wave of stage 1: its targets and their source runs
  int nwin;                  // number of descriptors (an upper bound known to the host; unused ones are empty)
  int by_range;              // 1: a descriptor is `range_len` consecutive binned targets of one band; 0: a target cell
  int range_len;             // 64 for dense target sets; fewer for a set queried against itself (sparse: the window grows with the span)
  int per_target;            // 1: no descriptors and no binned targets -- every lane of stage 1 takes one target, in the caller's order
  const int* tcell_of;       // cell of every target, by original id
  int* range_off;            // [bands + 1] by_range: first descriptor of every band; range_off[bands] = descriptors in use
  AgGrid occ_grid;           // bounded per-target searches: a coarse grid (cells of about the bound) ...
  const int* occ_count;      // ... and the number of sources in each of its cells
  int bounded;               // 1: neighbours farther than rmax are not wanted (their entries become -1)
  double rmax_window;        // bounded: the widest window stage 2 scans (rmax and a rounding pad)
  double rmax_rdist;         // bounded: sin^2(rmax / 2), the acceptance threshold on the reduced distance
  int self_query;            // 1: reference-distance mode (sources == targets, k == 2)
  unsigned long long* max_bits;  // reference-distance mode: running max of rdist (uint64 bits)
  long long* stats;          // [4]
  AgKnnArgs* device_copy;    // this structure in device memory, written by the window kernel (see AgKnnHot)
  const int* abort;          // the status word of the device (see ag_aborted)
};

// What a wave of stage 1 reads on its usual path.  The arguments of a kernel sit in scalar registers for its whole
// run, and the two grids of AgKnnArgs pushed those of the candidate loop out into lanes of vector registers (a
// hundred lane writes per wave before the first useful instruction); the grids are only needed by a window with
// more runs than its descriptor holds, which reads them from `full`, the copy the window kernel left in device memory.
struct AgKnnHot {
  int32_t* out_tgt;
  int* rq_count;
  int64_t cand_slots;
  int rq, rq_cap;
  const struct AgWindow* windows;
  const float4* spts;
  const float4* tpts;
  const int* tids;
  const float2* tgt;
  int32_t* out_src;
  int* cand_n;
  int* cand_t;
  int* cand_ids;
  int* queue;
  int* queue_count;
  const AgKnnArgs* full;
  int64_t nt;
  int nwin;
  int k;
  int self_query;
  float rg_chord2;
  const int* abort;
};

// Largest candidate list that is re-ranked in float64 by one thread (ag_knn_resolve_kernel, registers
// only); longer lists (k > 32) go to the wave-per-target stage 2.
#define AG_KNN_RESOLVE_CAP 34

template <int CAP>
__device__ __forceinline__ void ag_topk_insert(float (&bd)[CAP], int (&bi)[CAP], float d, int id) {
  if (d < bd[CAP - 1]) {
    bd[CAP - 1] = d;
    bi[CAP - 1] = id;
#pragma unroll
    for (int j = CAP - 1; j > 0; --j) {
      const bool sw = bd[j] < bd[j - 1];
      const float d0 = sw ? bd[j] : bd[j - 1], d1 = sw ? bd[j - 1] : bd[j];
      const int i0 = sw ? bi[j] : bi[j - 1], i1 = sw ? bi[j - 1] : bi[j];
      bd[j - 1] = d0;
      bd[j] = d1;
      bi[j - 1] = i0;
      bi[j] = i1;
    }
  }
}

__device__ __forceinline__ bool ag_pair_less(double r1, int i1, double r2, int i2) {
  return (r1 < r2) || (r1 == r2 && i1 < i2);
}

__device__ __forceinline__ double ag_wave_max_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}

// Enumerates the runs of binned sources within the guaranteed radius of any point of the
// longitude interval [lon_lo, lon_hi] of band `band`, using the per-band half-width table.
template <class F>
__device__ __forceinline__ void ag_cell_window(const AgGrid& g, const int* __restrict__ cell_start, int band,
                                               double lon_lo, double lon_hi, F&& f) {
  const double delta = g.win[band];
  AgPointWindow w;
  w.b0 = max(band - 1, 0);
  w.b1 = min(band + 1, g.nb - 1);
  w.lo = lon_lo - delta;
  w.hi = lon_hi + delta;
  w.full = delta < 0.0 || w.hi - w.lo >= AG_TWO_PI;
  for (int b = w.b0; b <= w.b1; ++b) {
    const int nc = g.ncols[b];
    const int base = g.off[b];
    int c_first, c_last;
    ag_window_cells(w, nc, false, &c_first, &c_last);
    f(cell_start[base + c_first], cell_start[base + c_last]);
    ag_window_cells(w, nc, true, &c_first, &c_last);
    if (c_last > c_first) f(cell_start[base + c_first], cell_start[base + c_last]);
  }
}

// The work of one wave of stage 1: the binned targets [t0, t1) -- all in latitude band `band`, with
// longitudes inside [lon_lo, lon_hi] -- and up to AG_WIN_RANGES half-open runs of binned sources that
// hold every source within the guaranteed radius of any of them.
// The runs laid end to end are the wave's tile: slot s of the tile is binned source s + d[1 + 2i] for the run i
// with d[2 + 2(i-1)] <= s < d[2 + 2i]  (d[1 + 2i] = first source of run i minus the slots before it, d[2 + 2i] =
// slots up to and including run i), so that every lane finds the sources of its slots without a loop over the
// runs and all loads of a tile are in flight together.  d[0] = number of runs (-1: too many, the wave enumerates
// them itself); total = slots of all runs (-1 when the runs are not listed).
#define AG_WIN_RANGES 9
struct __attribute__((aligned(32))) AgWindow {
  int d[20];
  int t0, t1, band, total;
  double lon_lo, lon_hi;
};

// Dense target sets: descriptors are ranges of 64 consecutive binned targets that never cross a band,
// so every lane of a wave has a target whatever the cell populations are (a regular latitude-longitude
// grid puts thousands of targets into a polar cell and a handful into its neighbours).  One block:
// range_off[b] = number of ranges of the bands before b.
__global__ __launch_bounds__(1024) void ag_knn_band_ranges_kernel(AgKnnArgs a) {
  if (ag_aborted(a.abort)) return;  // (a binning or a prefix sum upstream failed: see ag_aborted)
  __shared__ int partial[1024];
  const int nb = a.tgrid.nb, tid = threadIdx.x;
  const int per = (nb + 1023) / 1024;
  const int b_begin = tid * per, b_end = min(nb, b_begin + per);
  int sum = 0;
  for (int b = b_begin; b < b_end; ++b) {
    const int cnt = a.tstart[a.tgrid.off[b + 1]] - a.tstart[a.tgrid.off[b]];
    sum += (cnt + a.range_len - 1) / a.range_len;
  }
  partial[tid] = sum;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const int v = (tid >= d) ? partial[tid - d] : 0;
    __syncthreads();
    partial[tid] += v;
    __syncthreads();
  }
  int run = partial[tid] - sum;
  for (int b = b_begin; b < b_end; ++b) {
    a.range_off[b] = run;
    const int cnt = a.tstart[a.tgrid.off[b + 1]] - a.tstart[a.tgrid.off[b]];
    run += (cnt + a.range_len - 1) / a.range_len;
  }
  if (tid == 1023) a.range_off[nb] = partial[1023];
}

// one thread per descriptor: the targets of the wave and the runs of sources that cover them (stage 1
// then starts with one load instead of a chain of dependent table look-ups per wave)
__global__ void ag_knn_window_kernel(AgKnnArgs a, AgWindow* __restrict__ out) {
  if (ag_aborted(a.abort)) return;  // (a binning or a prefix sum upstream failed: see ag_aborted)
  if (blockIdx.x == 0 && threadIdx.x == 0) {  // counters of the stages that follow
    a.queue_count[0] = 0;  // stage-2 queue
    a.queue_count[1] = 0;
    *a.device_copy = a;
    if (a.max_bits) *a.max_bits = 0ull;
    if (a.stats) a.stats[0] = a.stats[1] = a.stats[2] = a.stats[3] = 0;
  }
  if (blockIdx.x == 0)  // fill of the float64 re-rank queues
    for (int q = threadIdx.x; q < a.rq; q += blockDim.x) a.rq_count[q * AG_KNN_RQ_STRIDE] = 0;
  if (a.per_target) return;  // (no descriptors, and no target grid to read)
  const int nb = a.tgrid.nb;
  const int* range_off = a.range_off;
  const int used = a.by_range ? range_off[nb] : a.tgrid.off[nb];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < a.nwin; i += gridDim.x * blockDim.x) {
    // (the descriptor is written in place: a local copy indexed by the run counter would live in scratch memory)
    AgWindow* w = out + i;
    int t0 = 0, t1 = 0;
    int band = 0, col_first = 0, col_last = 0;
    if (i < used) {
      if (a.by_range) {
        int lo = 0, hi = nb - 1;  // last band whose first range is <= i (bands without targets share an offset)
        while (lo < hi) {
          const int mid = (lo + hi + 1) >> 1;
          if (range_off[mid] <= i) lo = mid; else hi = mid - 1;
        }
        band = lo;
        const int b0 = a.tstart[a.tgrid.off[band]], b1 = a.tstart[a.tgrid.off[band + 1]];
        t0 = b0 + a.range_len * (i - range_off[band]);
        t1 = min(t0 + a.range_len, b1);
        // binned targets are ordered by cell, so the cells of the first and the last target bound the rest
        col_first = a.tcell_of[a.tids[t0]] - a.tgrid.off[band];
        col_last = a.tcell_of[a.tids[t1 - 1]] - a.tgrid.off[band];
      } else {
        t0 = a.tstart[i];
        t1 = a.tstart[i + 1];
        if (t0 < t1) {
          band = ag_band_of_cell(a.tgrid, i);
          col_first = col_last = i - a.tgrid.off[band];
        }
      }
    }
    int nruns = 0, total = 0;
    double lon_lo = 0.0, lon_hi = 0.0;
    if (t0 < t1) {
      const double wlon = AG_TWO_PI / a.tgrid.ncols[band];
      lon_lo = col_first * wlon;
      lon_hi = (col_last + 1) * wlon;
      // the window of the range of targets: for targets from a kept binning -- whose bands belong to another grid --
      // every source cell within the guaranteed radius of the box the range lies in (its band's latitudes, the
      // longitudes of its cells); otherwise the three bands around the targets' own, with the half-width of the table
      AgPointWindow pw;
      if (a.foreign_targets) {
        const double lat_hi = AG_HALF_PI - band * a.tgrid.dlat, lat_lo = lat_hi - a.tgrid.dlat;
        pw = ag_box_window(a.sgrid, lat_lo, lat_hi, lon_lo, lon_hi, a.rg);
      } else {
        const double delta = a.sgrid.win[band];
        pw.b0 = max(band - 1, 0);
        pw.b1 = min(band + 1, a.sgrid.nb - 1);
        pw.lo = lon_lo - delta;
        pw.hi = lon_hi + delta;
        pw.full = delta < 0.0 || pw.hi - pw.lo >= AG_TWO_PI;
      }
      // the table entries of all bands are fetched together, then the cell starts of all runs: two waits for
      // memory per descriptor instead of two per band
      constexpr int BANDS = 5;
      const int nbands = pw.b1 - pw.b0 + 1;
      int n = 0;
      if (nbands <= BANDS) {
        int nc[BANDS], base[BANDS];
#pragma unroll
        for (int i = 0; i < BANDS; ++i) {
          const int b = min(pw.b0 + i, pw.b1);
          nc[i] = a.sgrid.ncols[b];
          base[i] = a.sgrid.off[b];
        }
        int first[2 * BANDS], last[2 * BANDS];
#pragma unroll
        for (int j = 0; j < 2 * BANDS; ++j) {
          int c_first = 0, c_last = 0;
          if ((j >> 1) < nbands) ag_window_cells(pw, nc[j >> 1], (j & 1) != 0, &c_first, &c_last);
          const bool some = c_last > c_first;
          first[j] = a.sstart[base[j >> 1] + (some ? c_first : 0)];
          last[j] = a.sstart[base[j >> 1] + (some ? c_last : 0)];
        }
#pragma unroll
        for (int j = 0; j < 2 * BANDS; ++j) {
          if (first[j] < last[j]) {
            if (n < AG_WIN_RANGES) {
              w->d[1 + 2 * n] = first[j] - total;
              w->d[2 + 2 * n] = total + (last[j] - first[j]);
            }
            ++n;
            total += last[j] - first[j];
          }
        }
      }
      if (nbands > BANDS || n > AG_WIN_RANGES) {
        if (a.foreign_targets) {
          // more runs than a descriptor holds: one run over all covered bands (their cells are contiguous
          // in the binned array) -- a superset of the window
          const int first = a.sstart[a.sgrid.off[pw.b0]], last = a.sstart[a.sgrid.off[pw.b1 + 1]];
          w->d[1] = first;
          w->d[2] = last - first;
          n = last > first ? 1 : 0;
          total = n ? last - first : 0;
        } else {
          n = -1;
        }
      }
      nruns = n;
    } else {
      t0 = t1 = 0;
    }
    w->d[0] = nruns;
    w->t0 = t0;
    w->t1 = t1;
    w->band = (t0 < t1) ? band : 0;
    w->total = nruns >= 0 ? total : -1;
    w->lon_lo = lon_lo;
    w->lon_hi = lon_hi;
  }
}

// The window of ONE target in the source grid -- the three bands around it, with the half-width of the band table --
// as six runs of binned sources [first[j], last[j]) (two per band, the second one empty unless the window wraps);
// returns the number of sources in it.  The table entries of the three bands are fetched together, then the cell
// starts of the six runs: two waits for memory.
__device__ __forceinline__ int ag_knn_own_window(const AgGrid& g, const int* __restrict__ sstart, float2 tp,
                                                 int (&first)[6], int (&last)[6]) {
  const double lonw = ag_wrap_lon(static_cast<double>(tp.y));
  const int band = ag_band_of(g, static_cast<double>(tp.x));
  const double delta = g.win[band];
  AgPointWindow pw;
  pw.b0 = max(band - 1, 0);
  pw.b1 = min(band + 1, g.nb - 1);
  pw.lo = lonw - delta;
  pw.hi = lonw + delta;
  pw.full = delta < 0.0 || pw.hi - pw.lo >= AG_TWO_PI;
  const int nbands = pw.b1 - pw.b0 + 1;
  int nc[3], base[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int b = min(pw.b0 + i, pw.b1);
    nc[i] = g.ncols[b];
    base[i] = g.off[b];
  }
  int total = 0;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    int c_first = 0, c_last = 0;
    if ((j >> 1) < nbands) ag_window_cells(pw, nc[j >> 1], (j & 1) != 0, &c_first, &c_last);
    const bool some = c_last > c_first;
    first[j] = sstart[base[j >> 1] + (some ? c_first : 0)];
    last[j] = sstart[base[j >> 1] + (some ? c_last : 0)];
    total += last[j] - first[j];
  }
  return total;
}

// CAP best of the six runs of a lane's own window (float32 chord^2 to the unit vector t, sorted insert with ids)
template <int CAP>
__device__ __forceinline__ void ag_knn_scan_own(const float4* __restrict__ spts, const int (&first)[6], const int (&last)[6],
                                                float tx, float ty, float tz, float (&bd)[CAP], int (&bi)[CAP]) {
#pragma nounroll
  for (int j = 0; j < 6; ++j) {
    int p = first[0], end = last[0];
#pragma unroll
    for (int r = 1; r < 6; ++r)
      if (j == r) {
        p = first[r];
        end = last[r];
      }
    for (; p < end; ++p) {
      const float4 s = spts[p];
      const float dx = s.x - tx, dy = s.y - ty, dz = s.z - tz;
      ag_topk_insert<CAP>(bd, bi, dx * dx + dy * dy + dz * dz, __float_as_int(s.w));
    }
  }
}

// Stage 1 for a window that does not fit one tile, or whose runs its descriptor does not list (a handful of
// windows of a search, if any): tile after tile, sorted inserts with ids.  Kept out of line, behind a call: inlined, its
// loops and the grid it reads cost the usual path scalar registers for the whole run of the kernel.  Every value
// read from the descriptor is the same in all lanes; `out_d` / `out_i` receive the CAP best (float32 chord^2, id).
//
// Where the 64 targets of a descriptor lie far apart -- the coarse background of a stretched grid, binned in cells
// sized for its fine part: a run of 64 binned targets spans hundreds of cells -- their common window holds a thousand
// sources of which a target's own neighbourhood holds thirty.  Every lane then scans the window of its own target
// (`tp`, the three bands around it) straight from the binned array, as ag_knn_point_kernel does; taken when the
// longest own window is an eighth of the common one or less (a lane's candidate costs about eight times a tile's).
template <int CAP>
__device__ __attribute__((noinline)) void ag_knn_scan_general(const AgKnnArgs* __restrict__ full,
                                                             const AgWindow* __restrict__ wp, float4* tile, float tx,
                                                             float ty, float tz, bool active, float2 tp, float* out_d,
                                                             int* out_i) {
  const int lane = ag_lane();
  const float4* __restrict__ spts = full->spts;
  float bd[CAP];
  int bi[CAP];
#pragma unroll
  for (int j = 0; j < CAP; ++j) {
    bd[j] = 3.0e38f;
    bi[j] = -1;
  }
  const int common = __builtin_amdgcn_readfirstlane(wp->total);
  if (common > 0) {
    const AgGrid& g = full->sgrid;
    const int* __restrict__ sstart = full->sstart;
    int first[6], last[6];
    int own = 0;
    if (active) {
      own = ag_knn_own_window(g, sstart, tp, first, last);
    } else {
#pragma unroll
      for (int j = 0; j < 6; ++j) first[j] = last[j] = 0;
    }
    int longest = own;
#pragma unroll
    for (int ofs = 32; ofs > 0; ofs >>= 1) longest = max(longest, __shfl_xor(longest, ofs, 64));
    if (8 * longest <= common) {
      ag_knn_scan_own<CAP>(spts, first, last, tx, ty, tz, bd, bi);
#pragma unroll
      for (int j = 0; j < CAP; ++j) {
        out_d[j] = bd[j];
        out_i[j] = bi[j];
      }
      return;
    }
  }
  int fill = 0;
  auto scan_tile = [&](int count) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int j = 0; j < count; ++j) {
      const float4 s = tile[j];
      const float dx = s.x - tx, dy = s.y - ty, dz = s.z - tz;
      const float d = dx * dx + dy * dy + dz * dz;
      ag_topk_insert<CAP>(bd, bi, d, __float_as_int(s.w));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  auto stage_run = [&](int first_pos, int last_pos) {
    int pos = first_pos;
    while (pos < last_pos) {
      const int take = min(last_pos - pos, AG_TILE - fill);
      for (int i = lane; i < take; i += 64) tile[fill + i] = spts[pos + i];
      fill += take;
      pos += take;
      if (fill == AG_TILE) {
        scan_tile(fill);
        fill = 0;
      }
    }
  };
  const int nruns = __builtin_amdgcn_readfirstlane(wp->d[0]);
  if (nruns >= 0) {
    int before = 0;
    for (int r = 0; r < nruns; ++r) {
      const int off = __builtin_amdgcn_readfirstlane(wp->d[1 + 2 * r]);
      const int upto = __builtin_amdgcn_readfirstlane(wp->d[2 + 2 * r]);
      stage_run(off + before, off + upto);
      before = upto;
    }
  } else {
    ag_cell_window(full->sgrid, full->sstart, __builtin_amdgcn_readfirstlane(wp->band), wp->lon_lo, wp->lon_hi, stage_run);
  }
  if (fill > 0) scan_tile(fill);
#pragma unroll
  for (int j = 0; j < CAP; ++j) {
    out_d[j] = bd[j];
    out_i[j] = bi[j];
  }
}

// K = the number of neighbours when it is one of the small values instantiated below (the comparisons with it then
// cost nothing), 0 = read it from the arguments.
// (Short lists are held to 80 vector registers -- six waves per SIMD instead of five: O1280 -> r7 decoder 0.120 ->
// 0.114 ms; at seven waves the registers spill and nothing is gained.)
template <int CAP, int K>
__global__ __launch_bounds__(64 * AG_WAVES_PER_BLOCK, CAP <= 5 ? 6 : 1) void ag_knn_cell_kernel(AgKnnHot a) {
  if (ag_aborted(a.abort)) return;  // (a binning or a prefix sum upstream failed: see ag_aborted)
  __shared__ float4 tile_all[AG_WAVES_PER_BLOCK][AG_TILE];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = ag_lane();
  float4* tile = tile_all[wave];
  int wi = __builtin_amdgcn_readfirstlane(blockIdx.x * AG_WAVES_PER_BLOCK + wave);
  if (wi >= a.nwin) return;
  const int k = K > 0 ? K : a.k;

  for (int pass = 0;; ++pass) {
    // A descriptor nearly always holds one pass of targets.  It is read inside the loop, behind an index made
    // opaque: hoisted out, everything derived from it stays in scalar registers for the whole pass, and those
    // run out.  (Read through the constant address space: descriptors are written by an earlier launch, and only
    // loads the compiler knows to be unclobbered become scalar loads -- inside this loop, after the stores of a
    // previous pass, it cannot know.)
    asm volatile("" : "+v"(wi));
    wi = __builtin_amdgcn_readfirstlane(wi);
    typedef const int __attribute__((address_space(4))) * AgScalarInts;
    const AgScalarInts wq = (AgScalarInts)(reinterpret_cast<uintptr_t>(a.windows + wi));
    struct {
      int d[2 * AG_WIN_RANGES + 1];
      int total;
    } win;
#pragma unroll
    for (int i = 0; i < 2 * AG_WIN_RANGES + 1; ++i) win.d[i] = wq[i];
    const int t1 = wq[offsetof(AgWindow, t1) / sizeof(int)];
    const int tb = wq[offsetof(AgWindow, t0) / sizeof(int)] + 64 * pass;
    win.total = wq[offsetof(AgWindow, total) / sizeof(int)];
    if (tb >= t1) break;
    const int ti = tb + lane;
    const bool active = ti < t1;
    int tid = -1;
    float3 tu = make_float3(0.f, 0.f, 0.f);
    if (active) {
      if (a.tpts != nullptr) {  // a kept binning holds the unit vector (rounded from float64) next to the id
        const float4 rec = a.tpts[ti];
        tid = __float_as_int(rec.w);
        tu = make_float3(rec.x, rec.y, rec.z);
      } else {
        tid = a.tids[ti];
        const float2 tp = a.tgt[tid];
        tu = ag_unit_vec_f32(tp.x, tp.y);
      }
    }
    float bd[CAP];
    int bi[CAP];
#pragma unroll
    for (int j = 0; j < CAP; ++j) {
      bd[j] = 3.0e38f;
      bi[j] = -1;
    }
    // Windows that fit one tile (the usual case) take the tagged path: the chord^2 bits with the
    // low 8 mantissa bits replaced by the tile slot form a 32-bit key, and a sorted insert of a
    // key is one v_min_u32 plus CAP-1 v_med3_u32, with no branch and no id bookkeeping; the ids are
    // read back from the tile at the end.  The tag costs 255 ulp of resolution, which the
    // classification below accounts for (ag_chord2_err_tagged).
    const int win_total = win.total;
    const bool tagged = win_total >= 0 && win_total <= AG_TILE;
    if (tagged) {
      // every lane loads the sources of its slots (lane, lane + 64, ...): the loads of the whole tile -- and the
      // one of the target above -- are in flight together, one wait for all of them
      static_assert(AG_TILE == 256, "four slots per lane");
      auto load_slot = [&](int slot) {
        float4 rec = make_float4(0.f, 0.f, 0.f, 0.f);
        if (slot < win_total) {
          int off = win.d[1];
#pragma unroll
          for (int r = 1; r < AG_WIN_RANGES; ++r)
            if (r < win.d[0]) off = (slot >= win.d[2 * r]) ? win.d[1 + 2 * r] : off;
          rec = a.spts[slot + off];
        }
        return rec;
      };
      const float4 rec0 = load_slot(lane);
      float4 rec1 = make_float4(0.f, 0.f, 0.f, 0.f), rec2 = rec1, rec3 = rec1;
      if (win_total > 64) {
        rec1 = load_slot(lane + 64);
        rec2 = load_slot(lane + 128);
        rec3 = load_slot(lane + 192);
      }
      if (lane < win_total) tile[lane] = rec0;
      if (win_total > 64) {
        if (lane + 64 < win_total) tile[lane + 64] = rec1;
        if (lane + 128 < win_total) tile[lane + 128] = rec2;
        if (lane + 192 < win_total) tile[lane + 192] = rec3;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      unsigned kb[CAP];
#pragma unroll
      for (int j = 0; j < CAP; ++j) kb[j] = 0xffffffffu;
      auto chord2 = [&](const float4& s) {
        const float dx = s.x - tu.x, dy = s.y - tu.y, dz = s.z - tu.z;
        return __float_as_uint(__fmaf_rn(dz, dz, __fmaf_rn(dy, dy, dx * dx)));
      };
      auto insert = [&](unsigned x) {
        unsigned prev = kb[0];
        kb[0] = min(prev, x);
#pragma unroll
        for (int i = 1; i < CAP; ++i) {
          const unsigned cur = kb[i];
          kb[i] = max(min(prev, cur), min(max(prev, cur), x));  // median of three
          prev = cur;
        }
      };
      // four candidates per turn: their slots travel as the four bytes of one register, and one byte permute
      // per candidate puts the slot under the upper three bytes of the chord^2 (one add per four candidates
      // instead of an and + or per candidate)
      unsigned slots = 0x03020100u;
      int j = 0;
      for (; j + 4 <= win_total; j += 4) {
        const float4 s0 = tile[j], s1 = tile[j + 1], s2 = tile[j + 2], s3 = tile[j + 3];
        insert(__builtin_amdgcn_perm(chord2(s0), slots, 0x07060500u));
        insert(__builtin_amdgcn_perm(chord2(s1), slots, 0x07060501u));
        insert(__builtin_amdgcn_perm(chord2(s2), slots, 0x07060502u));
        insert(__builtin_amdgcn_perm(chord2(s3), slots, 0x07060503u));
        slots += 0x04040404u;
      }
      for (; j < win_total; ++j) insert((chord2(tile[j]) & 0xffffff00u) | static_cast<unsigned>(j));
#pragma unroll
      for (int j2 = 0; j2 < CAP; ++j2) {
        if (kb[j2] != 0xffffffffu) {
          bd[j2] = __uint_as_float(kb[j2] & 0xffffff00u);
          bi[j2] = __float_as_int(tile[kb[j2] & 0xffu].w);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    if (!tagged) {
      float far_d[CAP];
      int far_i[CAP];
      ag_knn_scan_general<CAP>(a.full, a.windows + wi, tile, tu.x, tu.y, tu.z, active,
                               active ? a.tgt[tid] : make_float2(0.f, 0.f), far_d, far_i);
#pragma unroll
      for (int j = 0; j < CAP; ++j) {
        bd[j] = far_d[j];
        bi[j] = far_i[j];
      }
    }

#include "ag_knn_classify.inc"
    if (tb + 64 >= t1) break;
  }
}

// Stage 1 for target sets far sparser than the source cells (the vertices of a coarse mesh looked up in a fine grid).
// A wave per target cell would run with a lane or two in use, and merging cells until the lanes fill up makes every
// lane compare against the sources of the whole merged window.  Here every lane takes one target -- in the caller's
// order, the targets are not binned at all -- and reads the sources of its own window (the three bands around it,
// with the half-width of the table: a few dozen sources) straight from the binned array.
template <int CAP, int K>
__global__ __launch_bounds__(256) void ag_knn_point_kernel(AgKnnHot a, AgGrid g, const int* __restrict__ sstart,
                                                           const double rmax_window, AgGrid occ,
                                                           const int* __restrict__ occ_count) {
  if (ag_aborted(a.abort)) return;  // (a binning or a prefix sum upstream failed: see ag_aborted)
  const int lane = ag_lane();
  const int k = K > 0 ? K : a.k;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const int pieces = static_cast<int>((a.nt + 63) >> 6);
  for (int wi = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6); wi < pieces; wi += nwaves) {
    const int64_t t = int64_t(wi) * 64 + lane;
    const bool has_target = t < a.nt;
    const int tid = has_target ? static_cast<int>(t) : -1;
    float bd[CAP];
    int bi[CAP];
#pragma unroll
    for (int j = 0; j < CAP; ++j) {
      bd[j] = 3.0e38f;
      bi[j] = -1;
    }
    if (has_target) {
      const float2 tp = a.tgt[tid];
      const float3 tu = ag_unit_vec_f32(tp.x, tp.y);
      int first[6], last[6];
      (void)ag_knn_own_window(g, sstart, tp, first, last);
      ag_knn_scan_own<CAP>(a.spts, first, last, tu.x, tu.y, tu.z, bd, bi);
    }
    // A search with a bound (rmax_window > 0): a target whose window is empty consults the occupancy of a coarse
    // grid, with cells of about the bound -- a few cells around it -- and is done if no source lies anywhere within
    // the bound (of the vertices of a global mesh looked up in a regional grid, nearly all).
    bool settled = false;
    if (rmax_window > 0.0 && has_target && bi[0] < 0) {
      const float2 tp = a.tgt[tid];
      const AgPointWindow pw = ag_point_window(occ, static_cast<double>(tp.x), ag_wrap_lon(static_cast<double>(tp.y)), rmax_window);
      int near = 0;
      for (int b = pw.b0; b <= pw.b1 && near == 0; ++b) {
        const int nc = occ.ncols[b], base = occ.off[b];
        for (int part = 0; part < 2; ++part) {
          int c_first, c_last;
          ag_window_cells(pw, nc, part != 0, &c_first, &c_last);
          for (int c = c_first; c < c_last && near == 0; ++c) near += occ_count[base + c];
        }
      }
      settled = near == 0;
      if (settled) {
        for (int j = 0; j < k; ++j) {
          a.out_src[static_cast<int64_t>(tid) * k + j] = -1;
          if (a.out_tgt != nullptr) a.out_tgt[static_cast<int64_t>(tid) * k + j] = tid;
        }
      }
    }
    // what the shared end of a pass works on (the targets are not binned: the re-ranking finds them by their ids)
    const bool active = has_target && !settled;
    const bool tagged = false;
    const int ti = tid;
#include "ag_knn_classify.inc"
  }
}

// ---------------------------------------------------------------------------------------------
// k-NN stage 1b: float64 re-ranking of the candidate lists stage 1 could not order in float32
// (one thread per binned target, scikit-learn's rdist, ties on ids)
// ---------------------------------------------------------------------------------------------
template <int CAP>
__global__ __launch_bounds__(256) void ag_knn_resolve_kernel(AgKnnArgs a) {
  if (ag_aborted(a.abort)) return;  // (a binning or a prefix sum upstream failed: see ag_aborted)
  __shared__ double sh[4];
  double my_max = 0.0;
  const int k = a.k;
  // Consecutive blocks take consecutive queues (a set queried against itself has one queue, with a slot per
  // target): the entries of a queue sit in its first blocks, and blocks go to the dies of the GPU in turn -- a grid
  // with the queue as its second dimension put every block that has work on one die.
  const int queues = a.self_query ? 1 : a.rq;
  const int q = blockIdx.x % queues, part = blockIdx.x / queues, parts = gridDim.x / queues;
  const int64_t first_slot = a.self_query ? 0 : int64_t(q) * a.rq_cap;
  const int64_t nslots = a.self_query ? a.nt : a.rq_count[q * AG_KNN_RQ_STRIDE];
  for (int64_t at = part * int64_t(blockDim.x) + threadIdx.x; at < nslots; at += int64_t(parts) * blockDim.x) {
    const int64_t slot = first_slot + at;
    const int n = a.cand_n[slot];
    if (n == 0) continue;
    const int ti = a.cand_t[slot];
    const int tid = a.per_target ? ti : a.tids[ti];
    const float2 tp = a.tgt[tid];
    double rr[CAP];
    int bi[CAP];
#pragma unroll
    for (int j = 0; j < CAP; ++j) {
      rr[j] = 1.0e300;
      bi[j] = 0x7fffffff;
      if (j < n) {
        bi[j] = a.cand_ids[static_cast<int64_t>(j) * a.cand_slots + slot];
        const float2 p = a.src[bi[j]];
        rr[j] = ag_rdist(tp.x, tp.y, p.x, p.y);
      }
    }
    // selection sort of the (rdist, id) pairs, fully unrolled so everything stays in registers
#pragma unroll
    for (int i = 0; i < CAP - 1; ++i) {
#pragma unroll
      for (int j = i + 1; j < CAP; ++j) {
        const bool sw = ag_pair_less(rr[j], bi[j], rr[i], bi[i]);
        const double r0 = sw ? rr[j] : rr[i], r1 = sw ? rr[i] : rr[j];
        const int i0 = sw ? bi[j] : bi[i], i1 = sw ? bi[i] : bi[j];
        rr[i] = r0;
        rr[j] = r1;
        bi[i] = i0;
        bi[j] = i1;
      }
    }
    if (a.self_query) {
      my_max = fmax(my_max, rr[1]);
    } else {
      int32_t* o = a.out_src + static_cast<int64_t>(tid) * k;
#pragma unroll
      for (int j = 0; j < CAP; ++j)
        if (j < k) o[j] = bi[j];
      if (a.out_tgt != nullptr) {
        int32_t* ot = a.out_tgt + static_cast<int64_t>(tid) * k;
#pragma unroll
        for (int j = 0; j < CAP; ++j)
          if (j < k) ot[j] = tid;
      }
    }
  }
  if (a.self_query) {
    my_max = ag_wave_max_f64(my_max);
    if (ag_lane() == 0) sh[threadIdx.x >> 6] = my_max;
    __syncthreads();
    if (threadIdx.x == 0) {
      const double m = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
      if (m > 0.0) atomicMax(a.max_bits, static_cast<unsigned long long>(__double_as_longlong(m)));
    }
  }
}

// distances of the final neighbour table (when the caller asks for them) and, for a bounded search, the removal
// of the neighbours beyond the bound: their ids become -1, their distances +infinity
__global__ void ag_knn_dist_kernel(AgKnnArgs a) {
  if (ag_aborted(a.abort)) return;  // (a binning or a prefix sum upstream failed: see ag_aborted)
  const int64_t total = a.nt * a.k;
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < total; i += int64_t(gridDim.x) * blockDim.x) {
    const int64_t t = i / a.k;
    const int id = a.out_src[i];
    double r = 2.0;  // (no neighbour: beyond every bound)
    if (id >= 0) {
      const float2 tp = a.tgt[t];
      const float2 sp = a.src[id];
      r = ag_rdist(tp.x, tp.y, sp.x, sp.y);
    }
    const bool keep = id >= 0 && (!a.bounded || r <= a.rmax_rdist);
    if (!keep && id >= 0) a.out_src[i] = -1;
    if (a.out_dist != nullptr) a.out_dist[i] = keep ? ag_rdist_to_dist(r) : __longlong_as_double(0x7ff0000000000000ll);
  }
}

// ---------------------------------------------------------------------------------------------
// k-NN stage 2: float64, one wave per queued target, k wavefront min-reductions
// ---------------------------------------------------------------------------------------------
struct AgPair {
  double r;
  int id;
};

__device__ __forceinline__ AgPair ag_wave_min_pair(double r, int id) {
#pragma unroll
  for (int ofs = 32; ofs > 0; ofs >>= 1) {
    const double r2 = __shfl_xor(r, ofs, 64);
    const int i2 = __shfl_xor(id, ofs, 64);
    if (ag_pair_less(r2, i2, r, id)) {
      r = r2;
      id = i2;
    }
  }
  return {r, id};
}

// Stage 2 works on one target per wave, in windows that grow until they provably hold the k nearest.  A window can
// hold thousands of sources (a target in the margin around a regional grid, a vertex near the rim of an area mask),
// and ranking all of them in float64 -- k times over -- is what this path used to cost.  Now a window is first
// looked at in float32: the candidates that can be among the k nearest are picked by their float32 chord^2
// against a threshold found by bisection on counts, and only that set (at most AG_EXACT_SET ids, held in LDS) is
// evaluated in float64 -- once -- and ranked from registers.
#define AG_EXACT_SET 256

// Outer threshold for an inner one: with A = {chord^2 <= t_in} and S = {chord^2 <= ag_exact_outer(t_in)} (float32
// values), every source outside S is farther from the target, in exact arithmetic, than every member of A --
// t_out - err(t_out) >= t_in + err(t_in) for the error bound of the float32 value.  So if A has k members, S holds
// the k nearest.
__device__ __forceinline__ float ag_exact_outer(float t_in) { return t_in + 4.0f * ag_chord2_err(t_in) + 1.0e-11f; }

__global__ __launch_bounds__(256) void ag_knn_exact_kernel(AgKnnArgs a) {
  if (ag_aborted(a.abort)) return;  // (a binning or a prefix sum upstream failed: see ag_aborted)
  __shared__ int set_all[4][AG_EXACT_SET];
  __shared__ double rdist_all[4][AG_EXACT_SET];
  const int lane = ag_lane();
  int* set = set_all[__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)];
  double* set_r = rdist_all[__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)];
  const int wave_global = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const int nq = *a.queue_count;
  const int k = a.k;
  auto wave_sum = [](int v) {
#pragma unroll
    for (int ofs = 32; ofs > 0; ofs >>= 1) v += __shfl_xor(v, ofs, 64);
    return v;
  };
  // (counted per wave and added up once at the end: atomics on one word are served one after the other)
  unsigned long long stat_targets = 0, stat_rounds = 0;
  for (int q = wave_global; q < nq; q += nwaves) {
    const int tid = a.queue[q];
    const float2 tpf = a.tgt[tid];
    const double tlat = tpf.x, tlon = tpf.y;
    const double tlonw = ag_wrap_lon(tlon);
    const float3 tu = ag_unit_vec(tpf.x, tpf.y);
    // sources of a window, a run per lane: the dependent table look-ups of all bands are in flight together
    auto count_window = [&](const AgPointWindow& w) {
      const int nruns = 2 * (w.b1 - w.b0 + 1);
      int total = 0;
      for (int j = lane; j < nruns; j += 64) {
        int first, last;
        ag_window_run(a.sgrid, a.sstart, w, j, &first, &last);
        total += last - first;
      }
      return wave_sum(total);
    };
    auto write_missing = [&](int from) {
      if (lane == 0 && !a.self_query)
        for (int j = from; j < k; ++j) {
          a.out_src[static_cast<int64_t>(tid) * k + j] = -1;
          if (a.out_tgt != nullptr) a.out_tgt[static_cast<int64_t>(tid) * k + j] = tid;
        }
    };
    double R = a.rg;
    int rounds = 0;
    if (a.bounded) {
      // A bounded search first counts the sources of the widest window it may look at.  Most targets that get here
      // have no source anywhere near (the vertices of a global mesh outside the area of the reference grid) and are
      // done with this one look; those with up to a few thousand sources in reach take that window at once instead of
      // working up to it (taking it at once whatever it holds was measured: no faster).
      const int total = count_window(ag_point_window(a.sgrid, tlat, tlonw, a.rmax_window));
      if (total == 0) {
        write_missing(0);
        ++stat_targets;
        continue;
      }
      if (total <= 16 * AG_EXACT_SET) R = a.rmax_window;
    }
    for (;;) {
      // a bounded search never looks farther than the bound: the pass at the bound is the last one, and what it
      // does not find is reported as missing
      const bool last = a.bounded && R >= a.rmax_window;
      if (last) R = a.rmax_window;
      const bool whole = R >= AG_PI;
      const double Rw = whole ? AG_PI : R;
      const double sh = ag_sin_f64(0.5 * Rw);
      const double rlimit = sh * sh * (1.0 - 1.0e-9);
      const AgPointWindow w = ag_point_window(a.sgrid, tlat, tlonw, Rw);
      const int nruns = 2 * (w.b1 - w.b0 + 1);
      const int total = count_window(w);
      if (total < k && !last && !whole) {  // (no need to look at the sources of a window that holds fewer than k)
        R *= 2.0;
        ++rounds;
        continue;
      }
      // f(id, chord2): every source of the window with its float32 chord^2 to the target.  All lanes make every
      // call together (f may count with ballots): a lane without a source of its own offers an infinite chord^2.
      auto for_each_source = [&](auto&& f) {
        for (int j0 = 0; j0 < nruns; j0 += 64) {
          int first_l, last_l;
          ag_window_run(a.sgrid, a.sstart, w, j0 + lane, &first_l, &last_l);
          const int nr = min(64, nruns - j0);
          for (int r = 0; r < nr; ++r) {
            const int first = __shfl(first_l, r, 64), last_p = __shfl(last_l, r, 64);
            for (int p0 = first; p0 < last_p; p0 += 64) {
              int id = 0x7fffffff;
              float c = __builtin_inff();
              if (p0 + lane < last_p) {
                const float4 s = a.spts[p0 + lane];
                const float dx = s.x - tu.x, dy = s.y - tu.y, dz = s.z - tu.z;
                id = __float_as_int(s.w);
                c = dx * dx + dy * dy + dz * dz;
              }
              f(id, c);
            }
          }
        }
      };
      // ---- the float32 threshold below which the candidates are collected, and the set
      // every source below t goes to the set (LDS) while it has room; returns how many there are
      auto collect = [&](float t) {
        int fill = 0;
        for_each_source([&](int id, float c) {
          const bool in = c <= t;
          const unsigned long long m = __ballot(in);
          const int slot = fill + __popcll(m & ((1ull << lane) - 1ull));
          if (in && slot < AG_EXACT_SET) set[slot] = id;
          fill += __popcll(m);
        });
        return fill;
      };
      float t_out = 3.0e38f;  // (a window that fits the set: everything)
      float t_lo = 0.0f, t_hi = 0.0f;
      int n_set = -1;
      if (total > AG_EXACT_SET) {
        // first bracket: the k-th smallest of the lanes' own minima is the chord^2 of a k-th candidate at least
        // (every lane also keeps the source of its minimum and its second smallest value: see below)
        float lane_min = 3.0e38f, lane_second = 3.0e38f;
        int lane_min_id = 0x7fffffff;
        for_each_source([&](int id, float c) {
          const bool lower = c < lane_min;
          lane_second = lower ? lane_min : fminf(lane_second, c);
          lane_min_id = lower ? id : lane_min_id;
          lane_min = lower ? c : lane_min;
        });
        int rank = 0;
        for (int o = 0; o < 64; ++o) {
          const float v = __shfl(lane_min, o, 64);
          rank += (v < lane_min || (v == lane_min && o < lane)) ? 1 : 0;
        }
        // (more neighbours wanted than there are lanes: no lane holds a k-th minimum, the bracket is the window's own)
        const unsigned long long at = __ballot(rank == k - 1);
        t_hi = at != 0ull ? __shfl(lane_min, __ffsll(static_cast<long long>(at)) - 1, 64) : 3.0e38f;
        // (fewer than k lanes met a source: every source the window is sure to hold completely)
        const float c_window = static_cast<float>(4.0 * sh * sh) * 1.000001f + 1.0e-11f;
        if (!(t_hi < c_window)) t_hi = c_window;
        // (below the smallest chord^2 of the window: nothing)
        t_lo = lane_min;
#pragma unroll
        for (int ofs = 32; ofs > 0; ofs >>= 1) t_lo = fminf(t_lo, __shfl_xor(t_lo, ofs, 64));
        t_lo = t_lo * 0.999999f - 1.0e-30f;
        t_out = ag_exact_outer(t_hi);
        if (k == 1) {
          // one neighbour wanted, and one candidate only below the threshold (one lane's minimum, that lane's next
          // value above it): that is the set -- the usual case, settled without a second pass
          const unsigned long long near = __ballot(lane_min <= t_out);
          if (__popcll(near) == 1) {
            const int owner = __ffsll(static_cast<long long>(near)) - 1;
            if (__shfl(lane_second, owner, 64) > t_out) {
              if (lane == owner) set[0] = lane_min_id;
              n_set = 1;
            }
          }
        }
      }
      // the set of the first bracket nearly always fits: collected at once, without counting first
      if (n_set < 0) n_set = collect(t_out);
      if (n_set > AG_EXACT_SET && total > AG_EXACT_SET) {
        auto counts = [&](float t_in, int* n_in, int* n_out) {
          const float t_o = ag_exact_outer(t_in);
          int ci = 0, co = 0;
          for_each_source([&](int, float c) {
            ci += c <= t_in ? 1 : 0;
            co += c <= t_o ? 1 : 0;
          });
          *n_in = wave_sum(ci);
          *n_out = wave_sum(co);
        };
        int n_in;
        counts(t_hi, &n_in, &n_set);
        // bisection: the inner set keeps k members (or all there are), the outer one shrinks until it fits
        for (int it = 0; it < 24 && n_set > AG_EXACT_SET && n_in > k; ++it) {
          const float mid = 0.5f * (t_lo + t_hi);
          if (!(mid > t_lo && mid < t_hi)) break;
          int m_in, m_out;
          counts(mid, &m_in, &m_out);
          if (m_in >= k) {
            t_hi = mid;
            n_in = m_in;
            n_set = m_out;
          } else {
            t_lo = mid;
          }
        }
        t_out = ag_exact_outer(t_hi);
        if (n_set <= AG_EXACT_SET) n_set = collect(t_out);
      }
      double prev_r = -1.0;
      int prev_i = -1;
      bool covered = true;
      int found = 0;
      auto accept = [&](const AgPair& m, int j) {  // false: the window does not hold the j-th neighbour for sure
        if (m.id == 0x7fffffff) return false;
        if (!whole && !(m.r <= rlimit)) return false;
        prev_r = m.r;
        prev_i = m.id;
        found = j + 1;
        if (a.self_query) {
          if (j == 1 && lane == 0 && m.r > 0.0)
            atomicMax(a.max_bits, static_cast<unsigned long long>(__double_as_longlong(m.r)));
        } else if (lane == 0) {
          a.out_src[static_cast<int64_t>(tid) * k + j] = m.id;
          if (a.out_tgt != nullptr) a.out_tgt[static_cast<int64_t>(tid) * k + j] = tid;
        }
        return true;
      };
      if (n_set <= AG_EXACT_SET) {
        // ---- the usual end: the set goes to LDS, its members are evaluated in float64, once, and the k smallest
        // are taken by k wavefront minima
        const int fill = n_set;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // (one member at a time, the distances parked in LDS: four evaluations side by side took 170 registers)
#pragma nounroll
        for (int slot = lane; slot < fill; slot += 64) {
          const float2 sp = a.src[set[slot]];
          set_r[slot] = ag_rdist(tlat, tlon, sp.x, sp.y);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int j = 0; j < k; ++j) {
          double best_r = 1.0e300;
          int best_i = 0x7fffffff;
          for (int slot = lane; slot < fill; slot += 64) {
            const double r = set_r[slot];
            const int id = set[slot];
            if (ag_pair_less(prev_r, prev_i, r, id) && ag_pair_less(r, id, best_r, best_i)) {
              best_r = r;
              best_i = id;
            }
          }
          if (!accept(ag_wave_min_pair(best_r, best_i), j)) {
            covered = false;
            break;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();  // (the set is reused by the next window)
      } else {
        // ---- more candidates than the set holds that float32 cannot tell apart (hundreds of sources at one
        // distance): k passes over the window, float64 for the candidates below the threshold only
        for (int j = 0; j < k; ++j) {
          double best_r = 1.0e300;
          int best_i = 0x7fffffff;
          for_each_source([&](int id, float c) {
            if (c <= t_out) {
              const float2 sp = a.src[id];
              const double r = ag_rdist(tlat, tlon, sp.x, sp.y);
              if (ag_pair_less(prev_r, prev_i, r, id) && ag_pair_less(r, id, best_r, best_i)) {
                best_r = r;
                best_i = id;
              }
            }
          });
          if (!accept(ag_wave_min_pair(best_r, best_i), j)) {
            covered = false;
            break;
          }
        }
      }
      if (covered || whole) break;
      if (last) {
        write_missing(found);
        break;
      }
      R *= 2.0;
      ++rounds;
    }
    ++stat_targets;
    stat_rounds += rounds;
  }
  if (lane == 0 && a.stats != nullptr) {
    if (stat_targets) atomicAdd(reinterpret_cast<unsigned long long*>(a.stats + 0), stat_targets);
    if (stat_rounds) atomicAdd(reinterpret_cast<unsigned long long*>(a.stats + 1), stat_rounds);
  }
}

// k beyond what the lists of stage 1 hold (AG_KNN_MAX_K): every target goes straight to the wave-per-target stage,
// whose windows double until they provably hold the k nearest (up to 256 candidates are ranked from LDS; more
// neighbours than that are taken in k passes over the window).  The reference accepts any k up to the number of
// sources (edges/builder.py:222-233).
__global__ void ag_knn_queue_all_kernel(AgKnnArgs a) {
  if (ag_aborted(a.abort)) return;  // (a binning or a prefix sum upstream failed: see ag_aborted)
  const int64_t gtid = blockIdx.x * int64_t(blockDim.x) + threadIdx.x, gsize = int64_t(gridDim.x) * blockDim.x;
  if (gtid == 0) {
    a.queue_count[0] = static_cast<int>(a.nt);
    a.queue_count[1] = 0;
    if (a.max_bits) *a.max_bits = 0ull;
    if (a.stats) a.stats[0] = a.stats[1] = a.stats[2] = a.stats[3] = 0;
  }
  for (int64_t t = gtid; t < a.nt; t += gsize) a.queue[t] = static_cast<int>(t);
}

// (Folding this into the kernel above -- the block that finishes last writes the result -- was measured: 2048 blocks
// counting themselves in on one word take 36 us, this launch 6.)
__global__ void ag_maxdist_finish_kernel(const unsigned long long* max_bits, double* out) {
  const double r = __longlong_as_double(static_cast<long long>(*max_bits));
  *out = (r > 0.0) ? ag_rdist_to_dist(r) : 0.0;
}

// cell size for a k-NN search over ns quasi-uniform sources: the disc of radius delta holds
// about 2(k+1)+2 sources, so the first window nearly always contains the k nearest
static double ag_knn_delta(int64_t ns, int k) {
  const double want = 2.0 * (k + 1) + 2.0;
  double d = sqrt(want * 4.0 / static_cast<double>(ns));  // pi d^2 * ns / (4 pi) = want
  if (d > AG_PI) d = AG_PI;
  return d;
}

// A density as the callers state it: how many times more nodes per steradian the set has, where it has nodes, than
// the same number of nodes spread over the whole sphere (1 for a global set, 1/f for one that covers the fraction f
// of the sphere; anything below 1, or not a number, counts as 1, anything beyond 1e12 as that).
static double ag_density(double d) { return d >= 1.0 ? (d < 1.0e12 ? d : 1.0e12) : 1.0; }

// Bands of the source grid of a k-NN search: cells sized for the density of the sources where they are.  A grid for
// the density d has d times the cells of a grid for as many sources spread over the sphere, all but one in d of them
// empty, and every cell is paid for when the cell counts are scanned -- about what a lane pays for a candidate.
// With few targets the candidates saved do not pay for the cells added: the sum of the two is smallest at
// 3.5 sqrt(nt ns d) cells, which is the grid chosen where it is the coarser one (the area mask of a stretched mesh:
// 2.6e6 vertices looked up in 4.6e7 grid nodes of density 10 -- 2.3 ms went into scanning 2.4e8 cells).  Never more
// than 32 cells per source, or 2^24 for a small set (a set confined to a speck of the sphere would otherwise ask for
// billions).
// `active`: the share of the targets that have sources around them at all (a search with a bound, whose other
// targets are settled from the occupancy grid: the density of the targets over that of the sources).
static int ag_knn_source_bands(int64_t ns, int64_t nt, int k, double src_density, double active = 1.0) {
  const double want = 2.0 * (k + 1) + 2.0;
  double d = ag_density(src_density);
  const double nt_active = static_cast<double>(nt > 0 ? nt : 1) * (active < 1.0 ? active : 1.0);
  const double balanced = 1.1 * want * sqrt(d * (nt_active > 1.0 ? nt_active : 1.0) / static_cast<double>(ns));
  if (d > balanced) d = balanced > 1.0 ? balanced : 1.0;
  int nb = ag_grid_bands(ag_knn_delta(ns, k) / sqrt(d));
  const double cells_max = fmax(32.0 * static_cast<double>(ns), 16777216.0) + 4096.0;
  const int nb_max = static_cast<int>(sqrt(cells_max * AG_PI / 4.0));  // the grid of nb bands has about 4 nb^2 / pi cells
  return nb < nb_max ? nb : (nb_max < 1 ? 1 : nb_max);
}

// list capacity of stage 1: k entries, the (k+1)-th as the boundary witness, room for an ambiguous cluster
static int ag_knn_cap(int k, bool self_query) {
  if (self_query) return 8;  // self + the 6 near-equidistant neighbours of a mesh node + one witness beyond them
  const int need = k + 2;
  const int caps[] = {3, 5, 8, 12, 18, 34, AG_KNN_MAX_K + 2};
  for (int c : caps)
    if (need <= c) return c;
  return AG_KNN_MAX_K + 2;
}

// Longitude cells of the target grid relative to the source grid, aiming at about 48 targets per
// target cell (one wave pass holds 64): m > 0 cuts every source cell into m target cells (dense
// targets), m < 0 merges -m source cells into one target cell (sparse targets: the window grows by
// the merged cells, but the lanes of a wave fill up).
// targets per source cell, where there are targets
static double ag_knn_targets_per_cell(int64_t nt, int nb, double tgt_density) {
  const double dlat = AG_PI / nb;
  return static_cast<double>(nt) * ag_density(tgt_density) / (4.0 * AG_PI / (dlat * dlat));
}

static int ag_knn_tmult(int64_t nt, int nb, double tgt_density) {
  const double per_cell = ag_knn_targets_per_cell(nt, nb, tgt_density);
  if (per_cell < 24.0) {
    int g = static_cast<int>(floor(48.0 / (per_cell > 1.0e-9 ? per_cell : 1.0e-9)));
    if (g > 16) g = 16;
    return g >= 2 ? -g : 1;
  }
  const double cut = ceil(per_cell / 48.0);
  return cut < 1.0 ? 1 : (cut > 64.0 ? 64 : static_cast<int>(cut));
}

// Dense target sets (what ag_knn_tmult answers with a multiplier >= 1) take their stage-1 descriptors by
// ranges of 64 binned targets; sparse ones by (merged) target cells; the sparsest are not binned at all, every
// lane of stage 1 takes a target (ag_knn_point_kernel).
static bool ag_knn_by_range(int tmult) { return tmult >= 1; }
// Below one target per source cell the per-target kernel wins for every k (4e6 sources, 2.5e5 targets, k = 3: 0.39 ms
// against 0.81 ms by merged cells; at k = 1, 1.6e4 targets: 0.33 against 0.75 -- what is left is the binning of the
// sources); from two to three targets per cell on it loses (k = 16, 1e6 targets: 3.4 against 2.9 ms), its lanes
// reading the same sources one by one that a wave of the cell kernel shares through its tile.
#define AG_KNN_PER_TARGET_BELOW 1.0
// Also where merged target cells would be chosen but the window of a merged cell -- its -tmult source cells, one more
// on either side, three bands -- holds more sources than a tile: every wave would then take the tile-after-tile
// path with its handful of targets (10^6 regional sources, 6e4 targets among them, k = 12: 14.7 ms against 1.5).
static bool ag_knn_per_target(int64_t ns, int64_t nt, int nb, int tmult, double src_density, double tgt_density,
                              bool foreign, bool same_set, bool self_query) {
  if (foreign || same_set || self_query) return false;
  if (ag_knn_targets_per_cell(nt, nb, tgt_density) < AG_KNN_PER_TARGET_BELOW) return true;
  if (tmult < 0) {
    const double dlat = AG_PI / nb;
    const double sources_per_cell = static_cast<double>(ns) * ag_density(src_density) / (4.0 * AG_PI / (dlat * dlat));
    if ((2 - tmult) * 3 * sources_per_cell > AG_TILE) return true;
  }
  return false;
}
// descriptors of a search: one per target cell, or one per range (every band ends with a partial one)
#define AG_KNN_SELF_RANGE 40  // targets per wave when a sparse set is its own target set (about 16 source cells)
static int64_t ag_knn_windows_ub(int64_t nt, int nb, int tmult, int range_len = 64) {
  return ag_knn_by_range(tmult) ? (nt + range_len - 1) / range_len + nb : ag_grid_cells_ub(nb, tmult);
}

struct AgKnnWs {
  AgBinned src, tgt;
  double rg;  // angular radius around a target that its window is guaranteed to cover
  int tmult;
  bool per_target;
  int range_len;
  int* range_off;
  int* queue;
  int* queue_count;
  unsigned long long* max_bits;
  int* cand_n;
  int* cand_t;
  int* cand_ids;
  int64_t cand_slots;
  int rq_cap;
  int* rq_count;
  struct AgWindow* windows;
  struct AgKnnArgs* args_copy;
  int *occ_ncols, *occ_off, *occ_count;  // coarse occupancy grid of a bounded search (never finer than the source grid)
  double* occ_win;
};

static int ag_knn_prepare(AgCarver& c, const float* d_src, int64_t ns, const float* d_tgt, int64_t nt, int k,
                          bool self_query, hipStream_t stream, AgKnnWs* w, double src_density, double tgt_density,
                          const AgKeptBinning& kept_tgt = AgKeptBinning(), bool bounded = false) {
  const bool foreign = kept_tgt.buffer != nullptr;
  // (Source cells of half the size for targets from a kept binning -- whose boxes do not line up with the source
  // cells, so that more of a window is cells the box only touches -- were measured: fewer candidates, but more runs
  // to stage per wave, and slower in the end: O1280 decoder 0.178 -> 0.210 ms.)
  const int nb_std = ag_knn_source_bands(ns, nt, k, src_density,
                                         bounded ? ag_density(tgt_density) / ag_density(src_density) : 1.0);
  const int nb = nb_std;
  const int tmult = ag_knn_tmult(nt, nb, tgt_density);
  int rc = ag_bin_points(c, d_src, ns, nb, nb, 1, true, stream, w ? &w->src : nullptr);
  if (rc) return rc;
  if (w) w->rg = 0.98 * (AG_PI / nb_std);
  // A set queried against itself (the reference distance, the neighbour lists of the Voronoi cells) is binned once:
  // the targets are the binned sources, taken in ranges of 64.  (The carve is the same either way.)
  const bool same_set = w != nullptr && d_src == d_tgt && ns == nt;
  const bool per_target = ag_knn_per_target(ns, nt, nb, tmult, src_density, tgt_density, foreign,
                                            d_src == d_tgt && ns == nt, self_query);
  int nb_t = nb;
  if (foreign) {  // the caller's binning of the targets: nothing to bin, nothing to carve
    nb_t = ag_kept_bands(nt, kept_tgt.cell);
    if (w) ag_kept_describe(kept_tgt, nt, &w->tgt);
  } else if (per_target) {
    if (w) w->tgt = AgBinned();  // (stage 1 reads the targets where the caller has them)
  } else {
    rc = ag_bin_points(c, d_tgt, nt, nb, nb, tmult, false, stream, (w && !same_set) ? &w->tgt : nullptr);
    if (rc) return rc;
    if (same_set) w->tgt = w->src;
  }
  const int cap = ag_knn_cap(k, self_query);
  int* queue = c.take<int>(nt);
  int* queue_count = c.take<int>(4);
  unsigned long long* max_bits = c.take<unsigned long long>(2);
  const int64_t win_a = ag_knn_windows_ub(nt, nb_t, foreign ? 1 : tmult), win_b = ag_knn_windows_ub(nt, nb, 1, AG_KNN_SELF_RANGE);
  const int64_t win_ub = (win_a > win_b ? win_a : win_b) + 1;
  // slots of the re-rank queues: by ranges (at most 64 targets per descriptor) every queue holds the targets of its
  // share of the descriptors; otherwise there is one queue, of all targets
  const int64_t by_range_a = ag_knn_windows_ub(nt, nb_t, 1), by_range_b = ag_knn_windows_ub(nt, nb, 1, AG_KNN_SELF_RANGE);
  const int64_t by_range_ub = (by_range_a > by_range_b ? by_range_a : by_range_b) + 1;
  const int64_t per_queue = ((by_range_ub + AG_KNN_RQ - 1) / AG_KNN_RQ) * 64;
  const int64_t cand_slots = per_queue * AG_KNN_RQ > nt ? per_queue * AG_KNN_RQ : nt;
  int* cand_n = cap <= AG_KNN_RESOLVE_CAP ? c.take<int>(cand_slots) : nullptr;
  int* cand_t = cap <= AG_KNN_RESOLVE_CAP ? c.take<int>(cand_slots) : nullptr;
  int* cand_ids = cap <= AG_KNN_RESOLVE_CAP ? c.take<int>(cand_slots * cap) : nullptr;
  int* rq_count = c.take<int>(AG_KNN_RQ * AG_KNN_RQ_STRIDE);
  AgWindow* windows = c.take<AgWindow>(win_ub);
  int* range_off = c.take<int>((nb_t > nb ? nb_t : nb) + 1);
  AgKnnArgs* args_copy = c.take<AgKnnArgs>(1);
  int* occ_ncols = c.take<int>(nb);
  int* occ_off = c.take<int>(nb + 1);
  double* occ_win = c.take<double>(nb);
  int* occ_count = c.take<int>(ag_grid_cells_ub(nb) + 1);
  if (w) {
    w->occ_ncols = occ_ncols;
    w->occ_off = occ_off;
    w->occ_win = occ_win;
    w->occ_count = occ_count;
    w->args_copy = args_copy;
    w->windows = windows;
    w->range_off = range_off;
    w->tmult = (same_set || foreign) ? 1 : tmult;  // (by ranges of the source binning itself / of the kept binning)
    w->per_target = per_target;
    w->range_len = (same_set && !foreign && !ag_knn_by_range(tmult)) ? AG_KNN_SELF_RANGE : 64;
    w->queue = queue;
    w->queue_count = queue_count;
    w->max_bits = max_bits;
    w->cand_n = cand_n;
    w->cand_t = cand_t;
    w->cand_ids = cand_ids;
    w->cand_slots = cand_slots;
    w->rq_cap = static_cast<int>(per_queue);
    w->rq_count = rq_count;
  }
  return AG_OK;
}

template <int CAP, int K>
static void ag_launch_stage1_k(const AgKnnArgs& a, hipStream_t stream) {
  const int blocks = ag_blocks(a.nwin, AG_WAVES_PER_BLOCK);
  AgKnnHot h;
  h.windows = a.windows;
  h.spts = a.spts;
  h.tpts = a.tpts;
  h.tids = a.tids;
  h.tgt = a.tgt;
  h.out_src = a.out_src;
  h.out_tgt = a.out_tgt;
  h.cand_n = a.cand_n;
  h.cand_t = a.cand_t;
  h.cand_ids = a.cand_ids;
  h.cand_slots = a.cand_slots;
  h.rq_count = a.rq_count;
  h.rq = a.rq;
  h.rq_cap = a.rq_cap;
  h.queue = a.queue;
  h.queue_count = a.queue_count;
  h.full = a.device_copy;
  h.nt = a.nt;
  h.nwin = a.nwin;
  h.k = a.k;
  h.self_query = a.self_query;
  h.rg_chord2 = a.rg_chord2;
  h.abort = a.abort;
  if (a.per_target) {
    AG_KERNEL("ag_knn_point_kernel", (ag_knn_point_kernel<CAP, K>), dim3(ag_blocks((a.nt + 63) / 64, 4, 8192)), dim3(256), stream, h, a.sgrid, a.sstart,
              a.bounded ? a.rmax_window : 0.0, a.occ_grid, a.occ_count);
  } else {
    AG_KERNEL(a.self_query ? "ag_knn_cell_kernel_self" : "ag_knn_cell_kernel", (ag_knn_cell_kernel<CAP, K>), dim3(blocks), dim3(64 * AG_WAVES_PER_BLOCK), stream, h);
  }
  if constexpr (CAP <= AG_KNN_RESOLVE_CAP) {
    const int64_t per_queue = a.self_query ? a.nt : a.rq_cap;
    const int queues = a.self_query ? 1 : a.rq;
    AG_KERNEL("ag_knn_resolve_kernel", ag_knn_resolve_kernel<CAP>, dim3(ag_blocks(per_queue, 256, 8192 / queues) * queues), dim3(256), stream, a);
  }
}

// the instance of stage 1 with the number of neighbours compiled in, where there is one
template <int CAP>
static void ag_launch_stage1(const AgKnnArgs& a, hipStream_t stream) {
  if constexpr (CAP == 3) {
    if (a.k == 1) return ag_launch_stage1_k<CAP, 1>(a, stream);
  } else if constexpr (CAP == 5) {
    if (a.k == 2) return ag_launch_stage1_k<CAP, 2>(a, stream);
    if (a.k == 3) return ag_launch_stage1_k<CAP, 3>(a, stream);
  } else if constexpr (CAP == 8) {
    if (a.k == 2) return ag_launch_stage1_k<CAP, 2>(a, stream);
    if (a.k == 4) return ag_launch_stage1_k<CAP, 4>(a, stream);
  }
  ag_launch_stage1_k<CAP, 0>(a, stream);
}

static int ag_knn_run(const float* d_src, int64_t ns, const float* d_tgt, int64_t nt, int k, bool self_query,
                      int32_t* d_out_src, double* d_out_dist, double* d_out_max, int64_t* d_stats, void* d_ws,
                      size_t ws_bytes, hipStream_t stream, double src_density, double tgt_density, double max_dist = 0.0,
                      const AgKeptBinning& kept_tgt = AgKeptBinning(), int32_t* d_out_tgt = nullptr) {
  AgCarver sizing(nullptr);
  const bool bounded = !self_query && max_dist > 0.0 && max_dist * (1.0 + 1.0e-6) + 1.0e-9 < AG_PI;
  ag_knn_prepare(sizing, d_src, ns, d_tgt, nt, k, self_query, stream, nullptr, src_density, tgt_density, kept_tgt, bounded);
  if (ws_bytes < sizing.total()) return ag_fail(AG_ERR_WORKSPACE, "workspace too small: %zu < %zu", ws_bytes, sizing.total());
  AgCarver c(d_ws);
  AgKnnWs w;
  int rc = ag_knn_prepare(c, d_src, ns, d_tgt, nt, k, self_query, stream, &w, src_density, tgt_density, kept_tgt, bounded);
  if (rc) return rc;
  AgKnnArgs a;
  a.sgrid = w.src.grid;
  a.tgrid = w.tgt.grid;
  a.spts = w.src.pts;
  a.sstart = w.src.cell_start;
  a.tids = w.tgt.ids;
  a.foreign_targets = kept_tgt.buffer != nullptr ? 1 : 0;
  // binned records of the targets (unit vector, id): a kept binning has them, and so has a set queried against
  // itself, whose targets are the binned sources
  a.tpts = (a.foreign_targets || (d_src == d_tgt && ns == nt)) ? w.tgt.pts : nullptr;
  a.tstart = w.tgt.cell_start;
  a.tgt = reinterpret_cast<const float2*>(d_tgt);
  a.src = reinterpret_cast<const float2*>(d_src);
  a.ns = ns;
  a.nt = nt;
  a.k = k;
  a.rg = w.rg;
  {
    const double s = sin(0.5 * a.rg);
    a.rg_chord2 = static_cast<float>(4.0 * s * s * (1.0 - 1.0e-4));
  }
  a.out_src = d_out_src;
  a.out_tgt = d_out_tgt;
  a.out_dist = d_out_dist;
  a.cand_n = w.cand_n;
  a.cand_t = w.cand_t;
  a.cand_ids = w.cand_ids;
  a.cand_slots = w.cand_slots;
  a.rq_count = w.rq_count;
  a.queue = w.queue;
  a.queue_count = w.queue_count;
  a.windows = w.windows;
  a.per_target = w.per_target ? 1 : 0;
  a.by_range = (!a.per_target && ag_knn_by_range(w.tmult)) ? 1 : 0;
  a.range_len = w.range_len;
  // (every 64 targets of a per-target search are a piece of work that takes the next queue, as the descriptors do)
  a.rq = (a.by_range || a.per_target) ? AG_KNN_RQ : 1;
  a.rq_cap = (a.by_range || a.per_target) ? w.rq_cap : static_cast<int>(nt);
  a.nwin = a.per_target ? 0 : static_cast<int>(ag_knn_windows_ub(nt, w.tgt.grid.nb, w.tmult, w.range_len));
  a.tcell_of = w.tgt.cell_of;
  a.range_off = w.range_off;
  a.self_query = self_query ? 1 : 0;
  a.rmax_window = max_dist * (1.0 + 1.0e-6) + 1.0e-9;
  a.bounded = (!self_query && max_dist > 0.0 && a.rmax_window < AG_PI) ? 1 : 0;
  {
    const double sh = sin(0.5 * max_dist);
    a.rmax_rdist = sh * sh;
  }
  a.max_bits = w.max_bits;
  a.stats = reinterpret_cast<long long*>(d_stats);
  a.device_copy = w.args_copy;
  a.abort = ag_status_word();
  a.occ_grid = AgGrid();
  a.occ_count = nullptr;
  if (a.bounded && a.per_target) {
    // occupancy of a grid with cells of about the bound (never finer than the source grid, whose cell array it is
    // laid out like): what lets stage 1 tell a target with nothing in reach from one look at a few cells
    int nb_occ = ag_grid_bands(a.rmax_window);
    if (nb_occ > w.src.grid.nb) nb_occ = w.src.grid.nb;
    a.occ_grid.nb = nb_occ;
    a.occ_grid.dlat = AG_PI / nb_occ;
    a.occ_grid.inv_dlat = nb_occ / AG_PI;
    a.occ_grid.ncols = w.occ_ncols;
    a.occ_grid.off = w.occ_off;
    a.occ_grid.win = w.occ_win;
    a.occ_count = w.occ_count;
    const int64_t occ_cells = ag_grid_cells_ub(nb_occ);
    AG_KERNEL("ag_band_table_kernel", ag_band_table_kernel, dim3(ag_blocks(occ_cells + 1, 4096, 256)), dim3(1024), stream,
              nb_occ, a.occ_grid.dlat, 1, w.occ_ncols, w.occ_off, w.occ_win, w.occ_count, occ_cells + 1,
              static_cast<unsigned long long*>(nullptr), int64_t(0), 0ull, AgBinProbe());
    AG_LAUNCH_CHECK();
    AG_KERNEL("ag_occupancy_count_kernel", ag_occupancy_count_kernel, dim3(ag_blocks(ns, 256, 16384)), dim3(256), stream,
              a.occ_grid, a.src, ns, w.occ_count);
    AG_LAUNCH_CHECK();
  }

  if (k > AG_KNN_MAX_K) {
    AG_KERNEL("ag_knn_queue_all_kernel", ag_knn_queue_all_kernel, dim3(ag_blocks(nt, 256, 4096)), dim3(256), stream, a);
    AG_LAUNCH_CHECK();
    AG_KERNEL("ag_knn_exact_kernel", ag_knn_exact_kernel, dim3(2048), dim3(256), stream, a);
    AG_LAUNCH_CHECK();
    if (d_out_dist != nullptr || a.bounded) {
      AG_KERNEL("ag_knn_dist_kernel", ag_knn_dist_kernel, dim3(ag_blocks(nt * k, 256, 16384)), dim3(256), stream, a);
      AG_LAUNCH_CHECK();
    }
    return AG_OK;
  }
  // (every block of the window kernel rebuilding this table in LDS, to save the one-block kernel, was measured
  // slower: 404 blocks x a prologue of dependent loads, 31.6 -> 57.6 us per build for the two searches)
  if (a.by_range) {
    AG_KERNEL("ag_knn_band_ranges_kernel", ag_knn_band_ranges_kernel, dim3(1), dim3(1024), stream, a);
    AG_LAUNCH_CHECK();
  }
  AG_KERNEL("ag_knn_window_kernel", ag_knn_window_kernel, dim3(ag_blocks(a.nwin, 256, 8192)), dim3(256), stream, a,
            w.windows);
  AG_LAUNCH_CHECK();
  switch (ag_knn_cap(k, self_query)) {
    case 3: ag_launch_stage1<3>(a, stream); break;
    case 5: ag_launch_stage1<5>(a, stream); break;
    case 8: ag_launch_stage1<8>(a, stream); break;
    case 12: ag_launch_stage1<12>(a, stream); break;
    case 18: ag_launch_stage1<18>(a, stream); break;
    case 34: ag_launch_stage1<34>(a, stream); break;
    default: ag_launch_stage1<AG_KNN_MAX_K + 2>(a, stream); break;
  }
  AG_LAUNCH_CHECK();
  // (a grid of 512 blocks instead of 2048 was measured slower: the queue of the O1280 decoder holds thousands of targets)
  AG_KERNEL("ag_knn_exact_kernel", ag_knn_exact_kernel, dim3(2048), dim3(256), stream, a);
  AG_LAUNCH_CHECK();
  if (self_query) {
    AG_KERNEL("ag_maxdist_finish_kernel", ag_maxdist_finish_kernel, dim3(1), dim3(1), stream, w.max_bits, d_out_max);
    AG_LAUNCH_CHECK();
  }
  if (!self_query && (d_out_dist != nullptr || a.bounded)) {
    AG_KERNEL("ag_knn_dist_kernel", ag_knn_dist_kernel, dim3(ag_blocks(nt * k, 256, 16384)), dim3(256), stream, a);
    AG_LAUNCH_CHECK();
  }
  return AG_OK;
}

extern "C" size_t ag_knn_workspace_bytes(int64_t ns, int64_t nt, int k, double src_density) {
  if (ns < 1 || nt < 0 || k < 1) return 0;
  // The carve depends on how the targets are handed to the lanes, which their density decides: by (merged) cells
  // for the sparsest the set can be, by ranges over ever finer target cells as it gets denser.  The two ends bound
  // everything between.
  AgCarver sparse(nullptr), dense(nullptr);
  ag_knn_prepare(sparse, nullptr, ns, nullptr, nt > 0 ? nt : 1, k, false, nullptr, nullptr, src_density, 1.0);
  ag_knn_prepare(dense, nullptr, ns, nullptr, nt > 0 ? nt : 1, k, false, nullptr, nullptr, src_density, 1.0e30);
  return sparse.total() > dense.total() ? sparse.total() : dense.total();
}

extern "C" int ag_knn(const float* d_src, int64_t ns, const float* d_tgt, int64_t nt, int k, double src_density,
                      double tgt_density, int32_t* d_out_src, double* d_out_dist, int64_t* d_stats, void* d_ws,
                      size_t ws_bytes, void* stream) {
  AG_REQUIRE(k >= 1, "ag_knn: k=%d must be positive", k);
  AG_REQUIRE(ns >= k, "ag_knn: k=%d exceeds the number of sources %lld", k, static_cast<long long>(ns));
  AG_REQUIRE(ns < (int64_t(1) << 31) && nt < (int64_t(1) << 31) && nt >= 0, "ag_knn: node counts must fit int32");
  AG_REQUIRE(nt * k < (int64_t(1) << 40), "ag_knn: output too large");
  if (nt == 0) return AG_OK;
  AG_REQUIRE(d_src && d_tgt && d_out_src && d_ws, "ag_knn: null pointer");
  return ag_knn_run(d_src, ns, d_tgt, nt, k, false, d_out_src, d_out_dist, nullptr, d_stats, d_ws, ws_bytes,
                    static_cast<hipStream_t>(stream), src_density, tgt_density);
}

extern "C" size_t ag_knn_binned_workspace_bytes(int64_t ns, int64_t nt, int k, double src_density) {
  if (ns < 1 || nt < 1 || k < 1) return 0;
  AgCarver c(nullptr);
  AgKeptBinning sizing;
  sizing.buffer = reinterpret_cast<const void*>(uintptr_t(256));  // (sizing only: never dereferenced)
  sizing.cell = 1.0e-9;                                           // the finest grid the buffer allows: the largest carve
  ag_knn_prepare(c, nullptr, ns, nullptr, nt, k, false, nullptr, nullptr, src_density, 1.0, sizing);
  return c.total();
}

extern "C" int ag_knn_binned(const float* d_src, int64_t ns, const float* d_tgt, int64_t nt, const void* d_tgt_binning,
                             double tgt_cell, int k, double src_density, int32_t* d_out_src, double* d_out_dist,
                             int64_t* d_stats, void* d_ws, size_t ws_bytes, void* stream) {
  AG_REQUIRE(k >= 1, "ag_knn_binned: k=%d must be positive", k);
  AG_REQUIRE(ns >= k, "ag_knn_binned: k=%d exceeds the number of sources %lld", k, static_cast<long long>(ns));
  AG_REQUIRE(ns < (int64_t(1) << 31) && nt < (int64_t(1) << 31) && nt >= 1, "ag_knn_binned: node counts out of range");
  AG_REQUIRE(nt * k < (int64_t(1) << 40), "ag_knn_binned: output too large");
  AG_REQUIRE(tgt_cell > 0.0, "ag_knn_binned: the cell size of the binning must be the one it was built with");
  AG_REQUIRE(d_src && d_tgt && d_tgt_binning && d_out_src && d_ws, "ag_knn_binned: null pointer");
  AgKeptBinning kept;
  kept.buffer = d_tgt_binning;
  kept.cell = tgt_cell;
  return ag_knn_run(d_src, ns, d_tgt, nt, k, false, d_out_src, d_out_dist, nullptr, d_stats, d_ws, ws_bytes,
                    static_cast<hipStream_t>(stream), src_density, 1.0, 0.0, kept);
}

// The k-NN graph as an edge list: what ag_knn (ag_knn_binned with a binning of the targets) finds, with the target of
// every entry written beside it by the same kernels
extern "C" int ag_knn_edges(const float* d_src, int64_t ns, const float* d_tgt, int64_t nt, const void* d_tgt_binning,
                            double tgt_cell, int k, double src_density, double tgt_density, int32_t* d_edge_src,
                            int32_t* d_edge_tgt, int64_t* d_stats, void* d_ws, size_t ws_bytes, void* stream) {
  AG_REQUIRE(k >= 1, "ag_knn_edges: k=%d must be positive", k);
  AG_REQUIRE(ns >= k, "ag_knn_edges: k=%d exceeds the number of sources %lld", k, static_cast<long long>(ns));
  AG_REQUIRE(ns < (int64_t(1) << 31) && nt < (int64_t(1) << 31) && nt >= 0, "ag_knn_edges: node counts must fit int32");
  AG_REQUIRE(nt * k < (int64_t(1) << 40), "ag_knn_edges: output too large");
  if (nt == 0) return AG_OK;
  AG_REQUIRE(d_src && d_tgt && d_edge_src && d_edge_tgt && d_ws, "ag_knn_edges: null pointer");
  AgKeptBinning kept;
  if (d_tgt_binning != nullptr) {
    AG_REQUIRE(tgt_cell > 0.0, "ag_knn_edges: the cell size of the binning must be the one it was built with");
    kept.buffer = d_tgt_binning;
    kept.cell = tgt_cell;
  }
  return ag_knn_run(d_src, ns, d_tgt, nt, k, false, d_edge_src, nullptr, nullptr, d_stats, d_ws, ws_bytes,
                    static_cast<hipStream_t>(stream), src_density, tgt_density, 0.0, kept, d_edge_tgt);
}

extern "C" int ag_knn_within(const float* d_src, int64_t ns, const float* d_tgt, int64_t nt, int k, double max_dist,
                             double src_density, double tgt_density, int32_t* d_out_src, double* d_out_dist,
                             int64_t* d_stats, void* d_ws, size_t ws_bytes, void* stream) {
  AG_REQUIRE(max_dist > 0.0, "ag_knn_within: the distance bound must be positive (got %g)", max_dist);
  AG_REQUIRE(k >= 1, "ag_knn_within: k=%d must be positive", k);
  AG_REQUIRE(ns >= k, "ag_knn_within: k=%d exceeds the number of sources %lld", k, static_cast<long long>(ns));
  AG_REQUIRE(ns < (int64_t(1) << 31) && nt < (int64_t(1) << 31) && nt >= 0, "ag_knn_within: node counts must fit int32");
  AG_REQUIRE(nt * k < (int64_t(1) << 40), "ag_knn_within: output too large");
  if (nt == 0) return AG_OK;
  AG_REQUIRE(d_src && d_tgt && d_out_src && d_ws, "ag_knn_within: null pointer");
  return ag_knn_run(d_src, ns, d_tgt, nt, k, false, d_out_src, d_out_dist, nullptr, d_stats, d_ws, ws_bytes,
                    static_cast<hipStream_t>(stream), src_density, tgt_density, max_dist);
}

extern "C" size_t ag_nn_maxdist_workspace_bytes(int64_t n, double density) {
  if (n < 2) return 0;
  AgCarver c(nullptr);
  ag_knn_prepare(c, nullptr, n, nullptr, n, 2, true, nullptr, nullptr, density, density);
  return c.total();
}

extern "C" int ag_nn_maxdist(const float* d_pts, int64_t n, double density, double* d_out_max, void* d_ws,
                             size_t ws_bytes, void* stream) {
  AG_REQUIRE(n >= 2, "ag_nn_maxdist: needs at least 2 points (n=%lld)", static_cast<long long>(n));
  AG_REQUIRE(n < (int64_t(1) << 31), "ag_nn_maxdist: node count must fit int32");
  AG_REQUIRE(d_pts && d_out_max && d_ws, "ag_nn_maxdist: null pointer");
  return ag_knn_run(d_pts, n, d_pts, n, 2, true, nullptr, nullptr, d_out_max, nullptr, d_ws, ws_bytes,
                    static_cast<hipStream_t>(stream), density, density);
}

// ---------------------------------------------------------------------------------------------
// ag_nn_maxdist for callers that cannot allocate: the workspace is a file-scope device array.
// ---------------------------------------------------------------------------------------------
// Every GPU the library runs on holds its own copy of the array, at its own address, until the process ends.
// In host code the array's name is only a handle for hipGetSymbolAddress: the address that kernels, copies and
// fills may use is the one that call returns for the current device, looked up once per device and kept.
#define AG_STATIC_SCRATCH_BYTES (size_t(48) << 20)
static __device__ __align__(256) unsigned char ag_static_scratch[AG_STATIC_SCRATCH_BYTES];

#define AG_STATIC_MAX_DEVICES 64
static int ag_static_scratch_address(void** out) {
  static std::mutex lock;
  static void* address[AG_STATIC_MAX_DEVICES] = {nullptr};
  int device = -1;
  AG_HIP(hipGetDevice(&device));
  AG_REQUIRE(device >= 0 && device < AG_STATIC_MAX_DEVICES, "ag_nn_maxdist_static: device %d is not supported", device);
  std::lock_guard<std::mutex> guard(lock);
  if (address[device] == nullptr) {
    void* p = nullptr;
    AG_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(ag_static_scratch)));
    AG_REQUIRE(p != nullptr, "ag_nn_maxdist_static: no address for the scratch array on device %d", device);
    address[device] = p;
  }
  *out = address[device];
  return AG_OK;
}

extern "C" int ag_nn_maxdist_static(const float* d_pts, int64_t n, double* d_out_max, void* stream) {
  AG_REQUIRE(n >= 2, "ag_nn_maxdist_static: needs at least 2 points (n=%lld)", static_cast<long long>(n));
  AG_REQUIRE(n <= AG_NN_MAXDIST_STATIC_MAX_N,
             "ag_nn_maxdist_static: %lld points exceed the %d the built-in scratch array holds; use ag_nn_maxdist with a "
             "workspace", static_cast<long long>(n), AG_NN_MAXDIST_STATIC_MAX_N);
  const size_t need = ag_nn_maxdist_workspace_bytes(n, 1.0);  // (the grid of a set spread over the sphere)
  AG_REQUIRE(need <= AG_STATIC_SCRATCH_BYTES, "ag_nn_maxdist_static: %zu bytes of scratch needed, %zu built in", need,
             AG_STATIC_SCRATCH_BYTES);
  AG_REQUIRE(d_pts && d_out_max, "ag_nn_maxdist_static: null pointer");
  void* scratch = nullptr;
  const int rc = ag_static_scratch_address(&scratch);
  if (rc) return rc;
  // (nothing in the scratch has to be zero on entry: the kernels of the query clear the counters, queue heads and
  // scan states they use, whatever an earlier call left there)
  return ag_knn_run(d_pts, n, d_pts, n, 2, true, nullptr, nullptr, d_out_max, nullptr, scratch, AG_STATIC_SCRATCH_BYTES,
                    static_cast<hipStream_t>(stream), 1.0, 1.0);
}

// ---------------------------------------------------------------------------------------------
// density of a node set, for the callers that do not know it
// ---------------------------------------------------------------------------------------------
// The sphere is cut into `cells` cells of equal area (rows of equal height in sin(latitude), columns of equal
// width in longitude), every node is counted into its cell, and the answer is the mean over the NODES of the
// density around them, relative to the density of as many nodes spread over the sphere:
//   cells * sum_c n_c (n_c - 1) / (n (n - 1))
// (n_c - 1 instead of n_c: an unbiased estimate of the squared cell densities, so that a small uniform set, most of
// whose cells hold one node or none, still comes out at 1).  1 for a global grid, 1/f for a grid that covers the
// fraction f of the sphere, about 1/f_dense for a stretched grid with most of its nodes in the fraction f_dense.
#define AG_DENSITY_MAX_CELLS 32768

static void ag_density_grid(int64_t n, int* rows, int* cols) {
  int r = 8;  // rows x 2 rows cells, 16 to 64 nodes per cell of a uniform set
  while (2 * (2 * r) * (2 * r) <= n / 16 && 2 * (2 * r) * (2 * r) <= AG_DENSITY_MAX_CELLS) r *= 2;
  *rows = r;
  *cols = 2 * r;
}

__global__ void ag_density_count_kernel(const float2* __restrict__ pts, int64_t n, int rows, int cols,
                                        int* __restrict__ count) {
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
    const float2 p = pts[i];
    const double z = ag_sin_f64(static_cast<double>(p.x));
    const int r = min(max(static_cast<int>((1.0 - z) * 0.5 * rows), 0), rows - 1);
    const int c = min(max(static_cast<int>(ag_wrap_lon(static_cast<double>(p.y)) * AG_INV_TWO_PI * cols), 0), cols - 1);
    const int cell = r * cols + c;
    // consecutive nodes mostly share a cell: the first lane of every run of equal cells counts for the run
    // (lanes past the end of the array are the highest lanes of the last wave)
    const int lane = ag_lane();
    const int prev = __shfl_up(cell, 1, 64);
    const bool head = (lane == 0) || (prev != cell);
    const unsigned long long heads = __ballot(head);
    const unsigned long long active = __ballot(1);
    const unsigned long long above = (lane == 63) ? 0ull : ((heads >> (lane + 1)) << (lane + 1));
    const int run_end = above ? (__ffsll(static_cast<long long>(above)) - 1) : (64 - __clzll(static_cast<long long>(active)));
    if (head) atomicAdd(&count[cell], run_end - lane);
  }
}

__global__ __launch_bounds__(1024) void ag_density_finish_kernel(const int* __restrict__ count, int cells, int64_t n,
                                                                 double* __restrict__ out) {
  __shared__ double partial[1024];
  double sum = 0.0;
  for (int c = threadIdx.x; c < cells; c += blockDim.x) {
    const double m = static_cast<double>(count[c]);
    sum += m * (m - 1.0);
  }
  partial[threadIdx.x] = sum;
  __syncthreads();
  for (int d = 512; d > 0; d >>= 1) {
    if (threadIdx.x < d) partial[threadIdx.x] += partial[threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double nn = static_cast<double>(n) * (static_cast<double>(n) - 1.0);
    double density = nn > 0.0 ? cells * partial[0] / nn : 1.0;
    if (!(density >= 1.0)) density = 1.0;
    if (density > cells) density = cells;
    *out = density;
  }
}

extern "C" size_t ag_point_density_workspace_bytes(int64_t n) {
  (void)n;
  return sizeof(int) * AG_DENSITY_MAX_CELLS + 256;
}

extern "C" int ag_point_density(const float* d_pts, int64_t n, double* d_out, void* d_ws, size_t ws_bytes, void* stream) {
  AG_REQUIRE(n >= 1, "ag_point_density: no points");
  AG_REQUIRE(d_pts && d_out && d_ws, "ag_point_density: null pointer");
  if (ws_bytes < ag_point_density_workspace_bytes(n)) return ag_fail(AG_ERR_WORKSPACE, "ag_point_density: workspace too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rows, cols;
  ag_density_grid(n, &rows, &cols);
  AgCarver c(d_ws);
  int* count = c.take<int>(AG_DENSITY_MAX_CELLS);
  int rc = ag_zero_i32(count, rows * cols, s);
  if (rc) return rc;
  AG_KERNEL("ag_density_count_kernel", ag_density_count_kernel, dim3(ag_blocks(n, 256, 8192)), dim3(256), s,
            reinterpret_cast<const float2*>(d_pts), n, rows, cols, count);
  AG_LAUNCH_CHECK();
  AG_KERNEL("ag_density_finish_kernel", ag_density_finish_kernel, dim3(1), dim3(1024), s, count, rows * cols, n, d_out);
  AG_LAUNCH_CHECK();
  return AG_OK;
}

// ---------------------------------------------------------------------------------------------
// radius query
// ---------------------------------------------------------------------------------------------
#define AG_ROW_LDS 1024  // row entries a wave can sort in LDS
#define AG_STASH_MAX 256  // longest row the count pass keeps for the fill pass
// Entries of the stash over all targets: 8 GiB of a 288 GB card.  (With 1 GiB the ten million targets of an
// icosphere of level 10 kept rows of 16 entries at most, and nearly every row of a 1 km encoder -- 66 sources on
// average -- went the way of the long rows: searched twice and sorted by a segmented radix sort, 15 ms of 92.)
#define AG_STASH_BUDGET (int64_t(1) << 31)
#define AG_RADIUS_RUNS 8  // runs of binned sources in a target's window (3 bands x 2 + spare)

// What the search needs to know about a target, computed once per target by one thread
struct __attribute__((aligned(16))) AgRadiusDesc {
  int first[AG_RADIUS_RUNS];  // start of every run in the binned source array
  int incl[AG_RADIUS_RUNS];   // inclusive prefix sum of the run lengths; incl[last] = candidates
  float4 unit;                // float32 unit vector of the target (rounded from float64)
};

struct AgRadiusArgs {
  AgGrid grid;
  const float4* spts;
  const int* sstart;
  const float2* src;
  const float2* tgt;
  int64_t nt;
  double radius;
  double rthr;       // sin^2(radius/2): the float64 acceptance threshold on rdist
  float c_in, c_out; // float32 chord^2: <= c_in surely inside, > c_out surely outside
  int64_t* offsets;  // count pass: counts are written here; fill pass: exclusive offsets
  int32_t* out_src;
  int32_t* out_tgt;
  int32_t* stash;    // [nt * stash_cap] sorted rows of at most stash_cap entries, written by the count pass
  int stash_cap;
  AgRadiusDesc* desc;  // [nt]
  unsigned long long* scan_state;  // state words of the scan of the counts, cleared by the window kernel
  int64_t scan_words;
  const int* abort;  // the status word of the device (see ag_aborted)
};

// rows the count pass can keep: as long as possible within a fixed budget of entries
static int ag_radius_stash_cap(int64_t nt) {
  const int64_t cap = (AG_STASH_BUDGET / (nt > 0 ? nt : 1)) & ~int64_t(15);
  return static_cast<int>(cap > AG_STASH_MAX ? AG_STASH_MAX : (cap < 16 ? 0 : cap));
}

// One thread per target: float32 unit vector and the runs of binned sources its window covers.
// The float64 window arithmetic is paid once per target here, not once per wave instruction in
// the search.
__global__ __launch_bounds__(256) void ag_radius_window_kernel(AgRadiusArgs a) {
  if (a.scan_state != nullptr)  // (count pass only) the scan that follows the count kernel starts from zeroed state
    for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < a.scan_words; i += int64_t(gridDim.x) * blockDim.x)
      a.scan_state[i] = 0ull;
  if (ag_aborted(a.abort)) {
    // (see ag_aborted)  Count pass: the counts become 0, 0, ... with -1 in front, so that their total -- which the
    // host reads for the size of the edge list -- is negative, its sign of a failed query.
    if (a.scan_state != nullptr)
      for (int64_t t = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; t < a.nt; t += int64_t(gridDim.x) * blockDim.x)
        a.offsets[t] = t == 0 ? -1 : 0;
    return;
  }
  for (int64_t t = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; t < a.nt; t += int64_t(gridDim.x) * blockDim.x) {
    const float2 tpf = a.tgt[t];
    const float3 tu = ag_unit_vec(tpf.x, tpf.y);
    const double tlat = tpf.x, tlonw = ag_wrap_lon(static_cast<double>(tpf.y));
    // the descriptor is written in place: only the runs that hold sources are kept (a band's second run
    // is empty unless the longitude window wraps), so the search's run look-up is as short as the window
    AgRadiusDesc* d = a.desc + t;
    int run = 0, kept = 0;
    const AgPointWindow w = ag_point_window(a.grid, tlat, tlonw, a.radius);
    const int nbands = w.b1 - w.b0 + 1;
    if (2 * nbands <= AG_RADIUS_RUNS) {
      // the table entries of all bands are fetched together, then the cell starts of all runs: two waits for
      // memory per target instead of two per run
      constexpr int BANDS = AG_RADIUS_RUNS / 2;
      int nc[BANDS], base[BANDS];
#pragma unroll
      for (int i = 0; i < BANDS; ++i) {
        const int b = min(w.b0 + i, w.b1);
        nc[i] = a.grid.ncols[b];
        base[i] = a.grid.off[b];
      }
      int first[AG_RADIUS_RUNS], last[AG_RADIUS_RUNS];
#pragma unroll
      for (int j = 0; j < AG_RADIUS_RUNS; ++j) {
        int c_first = 0, c_last = 0;
        if ((j >> 1) < nbands) ag_window_cells(w, nc[j >> 1], (j & 1) != 0, &c_first, &c_last);
        const bool some = c_last > c_first;
        first[j] = a.sstart[base[j >> 1] + (some ? c_first : 0)];
        last[j] = a.sstart[base[j >> 1] + (some ? c_last : 0)];
      }
#pragma unroll
      for (int j = 0; j < AG_RADIUS_RUNS; ++j) {
        if (last[j] > first[j]) {
          run += last[j] - first[j];
          d->first[kept] = first[j];
          d->incl[kept] = run;
          ++kept;
        }
      }
    } else {
      // more runs than the descriptor holds: one run over all covered bands (their cells are
      // contiguous in the binned array) -- a superset of the window
      const int first = a.sstart[a.grid.off[w.b0]];
      run = a.sstart[a.grid.off[w.b1 + 1]] - first;
      d->first[0] = first;
      d->incl[0] = run;
      kept = 1;
    }
    for (int j = kept; j < AG_RADIUS_RUNS; ++j) {
      d->first[j] = 0;
      d->incl[j] = run;
    }
    d->unit = make_float4(tu.x, tu.y, tu.z, __int_as_float(kept));  // (.w: the number of runs kept)
  }
}

__device__ __forceinline__ void ag_wave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Sources within the radius of target t, found by the whole wave from the target's descriptor:
// returns the count; matching ids are stored (unsorted) at row[0 .. min(count, row_cap)) when row
// is not null, and at spill[0 .. count) when spill is not null (rows too long for LDS).
__device__ __forceinline__ int ag_radius_scan(const AgRadiusArgs& a, int64_t t, int* row, int row_cap,
                                              int32_t* spill) {
  const int lane = ag_lane();
  const AgRadiusDesc* d = a.desc + t;
  const int4 fa = *reinterpret_cast<const int4*>(&d->first[0]), fb = *reinterpret_cast<const int4*>(&d->first[4]);
  const int4 ia = *reinterpret_cast<const int4*>(&d->incl[0]), ib = *reinterpret_cast<const int4*>(&d->incl[4]);
  const float4 tu = d->unit;
  const int nkept = __builtin_amdgcn_readfirstlane(__float_as_int(tu.w));
  const int total = ib.w;
  int cnt = 0;
  for (int q0 = 0; q0 < total; q0 += 128) {
    // two candidates per lane and round: both loads are in flight before the first test
    int q[2], id[2];
    bool live[2], in[2];
    float4 s[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      q[u] = q0 + 64 * u + lane;
      live[u] = q[u] < total;
      // run of candidate q: only as many steps as the window has runs (wave-uniform branches)
      int first = fa.x, excl = 0;
      if (nkept > 1) {
        if (q[u] >= ia.x) { first = fa.y; excl = ia.x; }
        if (nkept > 2) {
          if (q[u] >= ia.y) { first = fa.z; excl = ia.y; }
          if (nkept > 3) {
            if (q[u] >= ia.z) { first = fa.w; excl = ia.z; }
            if (nkept > 4) {
              if (q[u] >= ia.w) { first = fb.x; excl = ia.w; }
              if (q[u] >= ib.x) { first = fb.y; excl = ib.x; }
              if (q[u] >= ib.y) { first = fb.z; excl = ib.y; }
              if (q[u] >= ib.z) { first = fb.w; excl = ib.z; }
            }
          }
        }
      }
      s[u] = live[u] ? a.spts[first + (q[u] - excl)] : make_float4(4.f, 4.f, 4.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      id[u] = __float_as_int(s[u].w);
      const float dx = s[u].x - tu.x, dy = s[u].y - tu.y, dz = s[u].z - tu.z;
      const float dd = dx * dx + dy * dy + dz * dz;
      in[u] = live[u] && dd <= a.c_out;
      if (in[u] && dd > a.c_in) {
        const float2 tp = a.tgt[t];
        const float2 sp = a.src[id[u]];
        in[u] = ag_rdist(static_cast<double>(tp.x), static_cast<double>(tp.y), sp.x, sp.y) <= a.rthr;
      }
      const unsigned long long m = __ballot(in[u]);
      if (in[u]) {
        const int o = cnt + __popcll(m & ((1ull << lane) - 1ull));
        if (spill != nullptr) spill[o] = id[u];
        else if (row != nullptr && o < row_cap) row[o] = id[u];
      }
      cnt += __popcll(m);
    }
  }
  return cnt;
}

// The key of lane ^ D for D = 1, 2, 4, 8 (data-parallel primitives: the permutation happens in the operand fetch of
// the instruction, no trip through the LDS crossbar), 16 (swizzle) and 32 (permute).
template <int D>
__device__ __forceinline__ int ag_lane_xor(int v) {
  if constexpr (D == 1) return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);   // quad_perm:[1,0,3,2]
  if constexpr (D == 2) return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);   // quad_perm:[2,3,0,1]
  if constexpr (D == 4) {
    // lanes 0-3 and 8-11 of a row of 16 read four lanes up, lanes 4-7 and 12-15 four lanes down
    const int up = __builtin_amdgcn_update_dpp(v, v, 0x104, 0xF, 0x5, false);             // row_shl:4, banks 0 and 2
    return __builtin_amdgcn_update_dpp(up, v, 0x114, 0xF, 0xA, false);                    // row_shr:4, banks 1 and 3
  }
  if constexpr (D == 8) return __builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, true);  // row_ror:8
  if constexpr (D == 16) return __builtin_amdgcn_ds_swizzle(v, 0x401F);                   // xor 16 within 32 lanes
  return __builtin_amdgcn_ds_bpermute((ag_lane() ^ 32) << 2, v);
}

// The lanes that keep the smaller key in the step of distance D of a round of blocks of K keys (ascending where
// (lane & K) == 0, all of them for K = 64): a constant of the network.
template <int K, int D>
constexpr unsigned long long ag_bitonic_mask() {
  unsigned long long m = 0ull;
  for (int i = 0; i < 64; ++i) {
    const bool up = K == 64 ? true : (i & K) == 0;
    const bool lower = (i & D) == 0;
    if (lower == up) m |= 1ull << i;
  }
  return m;
}

// One compare-exchange step.  The select takes its lane mask as a scalar constant (left to the compiler, the 21
// masks of a sort are computed once, kept in scalar registers the kernels do not have, spilled to lanes of a
// vector register and read back two instructions a step).
template <int K, int D>
__device__ __forceinline__ int ag_bitonic_step(int v) {
  const int other = ag_lane_xor<D>(v);
  const int mn = min(v, other), mx = max(v, other);
  constexpr unsigned long long take_min = ag_bitonic_mask<K, D>();
  int r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(mx), "v"(mn), "s"(take_min));
  return r;
}

template <int K>
__device__ __forceinline__ int ag_bitonic_round(int v) {
  if constexpr (K >= 64) v = ag_bitonic_step<K, 32>(v);
  if constexpr (K >= 32) v = ag_bitonic_step<K, 16>(v);
  if constexpr (K >= 16) v = ag_bitonic_step<K, 8>(v);
  if constexpr (K >= 8) v = ag_bitonic_step<K, 4>(v);
  if constexpr (K >= 4) v = ag_bitonic_step<K, 2>(v);
  v = ag_bitonic_step<K, 1>(v);
  return v;
}

// Ascending sort of one key per lane, n <= 64 of them real (in lanes 0 ... n - 1), the others a value that ranks
// last: 21 compare-exchange steps, 18 of them without the LDS crossbar.  Short rows stop early: after the round of
// blocks of 16 (or 32) keys the first block is sorted ascending, and the row is inside it.
__device__ __forceinline__ int ag_wave_sort64(int v, int n) {
  v = ag_bitonic_round<2>(v);
  v = ag_bitonic_round<4>(v);
  v = ag_bitonic_round<8>(v);
  v = ag_bitonic_round<16>(v);
  if (n <= 16) return v;
  v = ag_bitonic_round<32>(v);
  if (n <= 32) return v;
  v = ag_bitonic_round<64>(v);
  return v;
}

// Sort of the n <= AG_ROW_LDS distinct ids in `row` (LDS, room for n rounded up to 4) into out[0..n).
//  * The first 64 entries are sorted in registers by a bitonic network: about 70 instructions, where counting the
//    smaller entries of every entry takes two per pair (130 for a row of 64) -- and rows of 60 to 70 entries are
//    what the targets of an encoder have, their sort more than half of the search kernel's instructions.
//  * Entries 64 ... 127 (a row a little over 64 has a handful) are placed one at a time: an entry's position is
//    the number of smaller entries, two ballots; every sorted entry that is larger moves up by one.
//  * Longer rows by rank, two entries per lane sharing every broadcast read of four row entries.
__device__ __forceinline__ void ag_row_rank_sort(int* row, int n, int32_t* __restrict__ out) {
  const int lane = ag_lane();
  if (n <= 128) {
    const int sa = ag_wave_sort64(lane < n ? row[lane] : 0x7fffffff, n);
    if (n <= 64) {
      if (lane < n) out[lane] = sa;
      return;
    }
    const int m = n - 64;
    const int vb = lane < m ? row[64 + lane] : 0x7fffffff;
    int shift = 0, rb = 0;
    for (int j = 0; j < m; ++j) {
      const int s = __builtin_amdgcn_readlane(vb, j);
      const unsigned long long below_a = __ballot(sa < s), below_b = __ballot(vb < s);
      shift += sa > s ? 1 : 0;
      rb = lane == j ? __popcll(below_a) + __popcll(below_b) : rb;
    }
    out[lane + shift] = sa;  // (n > 64: all 64 sorted entries are entries of the row)
    if (lane < m) out[rb] = vb;
    return;
  }
  const int n4 = (n + 3) & ~3;
  if (lane < n4 - n) row[n + lane] = 0x7fffffff;  // pad to a multiple of 4 with a value that ranks last
  ag_wave_sync_lds();
  const int4* row4 = reinterpret_cast<const int4*>(row);
  for (int e0 = 0; e0 < n; e0 += 128) {
    const int ea = e0 + lane, eb = e0 + 64 + lane;
    const int va = ea < n ? row[ea] : 0x7fffffff, vb = eb < n ? row[eb] : 0x7fffffff;
    int ra = 0, rb = 0;
    for (int j = 0; j < n4 / 4; ++j) {
      const int4 q = row4[j];
      ra += (q.x < va) + (q.y < va) + (q.z < va) + (q.w < va);
      rb += (q.x < vb) + (q.y < vb) + (q.z < vb) + (q.w < vb);
    }
    if (ea < n) out[ra] = va;
    if (eb < n) out[rb] = vb;
  }
}

// ---------------------------------------------------------------------------------------------
// The order in which the count pass takes its targets.
//
// In the caller's order every source record is fetched once per target whose window covers it -- eight times
// the bytes of the sources for an encoder (a mesh node's window holds 160 data nodes, 60 of them within the
// radius, and the windows of neighbouring mesh nodes overlap) -- because nothing ties the targets in flight
// together: mesh nodes in latitude order are neighbours in latitude only.  Once the count kernel itself no longer
// waited for its own instructions, that traffic was what it waited for (427 MB in 0.11 ms).  So the targets are
// sorted into groups first -- B latitude bands of the source grid by a stretch of longitude W cells wide
// (counting sort: count, scan, place) -- their window descriptors are written in that order, and the count pass
// walks them in that order, the blocks that share an L2 (every eighth) taking runs of consecutive positions: what
// one wave fetched, the waves around it find in the L2.  What a target's row holds does not depend on the order:
// every target has its own window, computed from its own coordinates.
// ---------------------------------------------------------------------------------------------
struct AgRadiusGroups {
  int B, W;                  // bands per group; cells per group in the band of the group with the most cells
  int nbg, nch_max;          // groups of bands; groups per group of bands in the index (those past a row's own count stay empty)
  int64_t ngroups;           // nbg * nch_max
  double rpad;               // radius + the pad of ag_box_window
  double* rwin;              // [nb] longitude half-width of the window of any point of band b for this radius; < 0: all longitudes
  int* gcount;               // [ngroups + 1] targets per group, then their exclusive prefix sum; [ngroups] = nt
  int* gof;                  // [nt] group of every target
  int* gslot;                // [nt] its place among the targets of the group
  int* gorder;               // [nt] the target at every position of the group order
  unsigned long long* gscan_state;
  int64_t gscan_words;
};

// groups along the longitude for the group of bands `beta`: as many as the band nearest the equator needs
__device__ __forceinline__ int ag_rg_chunks(const AgGrid& g, int B, int W, int nch_max, int beta) {
  const int b_lo = beta * B, b_hi = min(b_lo + B, g.nb) - 1;
  const int bref = min(max((g.nb - 1) >> 1, b_lo), b_hi);
  const int nch = (g.ncols[bref] + W - 1) / W;
  return min(max(nch, 1), nch_max);
}

// Clears the group counters and the state words of the two scans, and fills the table of window half-widths.
__global__ __launch_bounds__(256) void ag_radius_groups_init_kernel(AgRadiusArgs a, AgRadiusGroups gr) {
  const int64_t gtid = blockIdx.x * int64_t(blockDim.x) + threadIdx.x, gsize = int64_t(gridDim.x) * blockDim.x;
  for (int64_t i = gtid; i <= gr.ngroups; i += gsize) gr.gcount[i] = 0;
  for (int64_t i = gtid; i < gr.gscan_words; i += gsize) gr.gscan_state[i] = 0ull;
  if (a.scan_state != nullptr)
    for (int64_t i = gtid; i < a.scan_words; i += gsize) a.scan_state[i] = 0ull;
  const double R = a.radius;
  const double pad = 1.0e-7 + 1.0e-6 * R;
  for (int64_t b = gtid; b < a.grid.nb; b += gsize) {
    // (the window of the box of the band, as ag_box_window computes it: it holds the window of every point of the band)
    const double lat_hi = AG_HALF_PI - b * a.grid.dlat, lat_lo = lat_hi - a.grid.dlat;
    const double phit = fmax(fabs(lat_lo), fabs(lat_hi));
    double w = -1.0;
    if (phit + R + pad < AG_HALF_PI && R < AG_HALF_PI) {
      const double q = ag_sin_f64(R) / ag_cos_f64(phit);
      if (q < 1.0) w = asin(q) * (1.0 + 1.0e-6) + pad;
    }
    gr.rwin[b] = w;
  }
}

__global__ void ag_radius_group_count_kernel(AgRadiusArgs a, AgRadiusGroups gr) {
  if (ag_aborted(a.abort)) return;  // (a binning or a prefix sum upstream failed: see ag_aborted)
  for (int64_t t = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; t < a.nt; t += int64_t(gridDim.x) * blockDim.x) {
    const float2 p = a.tgt[t];
    const int band = ag_band_of(a.grid, static_cast<double>(p.x));
    const int beta = band / gr.B;
    const int nch = ag_rg_chunks(a.grid, gr.B, gr.W, gr.nch_max, beta);
    const double lonw = ag_wrap_lon(static_cast<double>(p.y));
    const int j = min(nch - 1, max(0, static_cast<int>(lonw * (nch * AG_INV_TWO_PI))));
    const int c = beta * gr.nch_max + j;
    gr.gof[t] = c;
    // consecutive lanes with the same group take their places with one atomic (see ag_bin_count_kernel)
    const int lane = ag_lane();
    const int prev = __shfl_up(c, 1, 64);
    const bool head = (lane == 0) || (prev != c);
    const unsigned long long heads = __ballot(head);
    const unsigned long long active = __ballot(1);
    const int my_head = 63 - __clzll(static_cast<long long>(heads & (~0ull >> (63 - lane))));
    const unsigned long long above = (lane == 63) ? 0ull : ((heads >> (lane + 1)) << (lane + 1));
    const int run_end = above ? (__ffsll(static_cast<long long>(above)) - 1)
                              : (64 - __clzll(static_cast<long long>(active)));
    int base = 0;
    if (head) base = atomicAdd(&gr.gcount[c], run_end - lane);
    base = __shfl(base, my_head, 64);
    gr.gslot[t] = base + (lane - my_head);
  }
}

// One thread per target: its window descriptor, written to the target's place in group order (the window's
// half-width from the table of the bands: no arcsine per target).  A scan of the group counts that did not come
// out at nt (it gave up: state words not cleared) leaves nothing to trust: no descriptor is written, the counts
// of the query become 0, 0, ... with -1 in front, and their total -- negative -- is what the caller's check of the
// edge count reports.
__global__ __launch_bounds__(256) void ag_radius_group_desc_kernel(AgRadiusArgs a, AgRadiusGroups gr) {
  const bool failed = ag_aborted(a.abort) || gr.gcount[gr.ngroups] != static_cast<int>(a.nt);
  const AgGrid& grid = a.grid;
  for (int64_t t = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; t < a.nt; t += int64_t(gridDim.x) * blockDim.x) {
    if (failed) {
      a.offsets[t] = t == 0 ? -1 : 0;
      continue;
    }
    const int pos = gr.gcount[gr.gof[t]] + gr.gslot[t];
    gr.gorder[pos] = static_cast<int>(t);
    const float2 p = a.tgt[t];
    const float3 u = ag_unit_vec(p.x, p.y);
    const double tlat = p.x, tlonw = ag_wrap_lon(static_cast<double>(p.y));
    AgRadiusDesc* d = a.desc + pos;
    AgPointWindow w;
    w.b0 = ag_band_of(grid, tlat + gr.rpad);
    w.b1 = ag_band_of(grid, tlat - gr.rpad);
    const double dw = gr.rwin[ag_band_of(grid, tlat)];
    w.lo = tlonw - dw;
    w.hi = tlonw + dw;
    w.full = dw < 0.0 || w.hi - w.lo >= AG_TWO_PI;
    const int nbands = w.b1 - w.b0 + 1;
    int run = 0, kept = 0;
    if (2 * nbands <= AG_RADIUS_RUNS) {
      constexpr int BANDS = AG_RADIUS_RUNS / 2;
      int nc[BANDS], base[BANDS];
#pragma unroll
      for (int i = 0; i < BANDS; ++i) {
        const int b = min(w.b0 + i, w.b1);
        nc[i] = grid.ncols[b];
        base[i] = grid.off[b];
      }
      int first[AG_RADIUS_RUNS], last[AG_RADIUS_RUNS];
#pragma unroll
      for (int j = 0; j < AG_RADIUS_RUNS; ++j) {
        int c_first = 0, c_last = 0;
        if ((j >> 1) < nbands) ag_window_cells(w, nc[j >> 1], (j & 1) != 0, &c_first, &c_last);
        const bool some = c_last > c_first;
        first[j] = a.sstart[base[j >> 1] + (some ? c_first : 0)];
        last[j] = a.sstart[base[j >> 1] + (some ? c_last : 0)];
      }
#pragma unroll
      for (int j = 0; j < AG_RADIUS_RUNS; ++j) {
        if (last[j] > first[j]) {
          run += last[j] - first[j];
          d->first[kept] = first[j];
          d->incl[kept] = run;
          ++kept;
        }
      }
    } else {
      // more runs than a descriptor holds: one run over all covered bands, a superset of the window
      const int first = a.sstart[grid.off[w.b0]];
      run = a.sstart[grid.off[w.b1 + 1]] - first;
      d->first[0] = first;
      d->incl[0] = run;
      kept = 1;
    }
    for (int j = kept; j < AG_RADIUS_RUNS; ++j) {
      d->first[j] = 0;
      d->incl[j] = run;
    }
    d->unit = make_float4(u.x, u.y, u.z, __int_as_float(kept));
  }
}

// Count pass: one wave per target counts its row; rows of at most stash_cap entries are sorted and
// kept in the stash so that the fill pass does not search again.
//
// A wave's targets follow each other through three dependent trips to memory -- descriptor, candidates, the
// row written -- and five waves per SIMD did not cover them: the kernel waited more than it computed.  Hence:
//  * the descriptor of the wave's NEXT target is fetched while the current one is searched;
//  * the runs of a window are walked in pieces of 64 candidates, a candidate per lane, FOUR pieces in flight
//    (a mesh node's window of 160 data nodes is three runs of fifty to sixty: one trip);
//  * the float64 test of a candidate inside the error band of the float32 one is not made where it is met:
//    such candidates (one in ten thousand) are set aside in LDS and tested after the loop, which keeps the
//    registers of the float64 sine and cosine out of the loop (95 -> 64: eight waves per SIMD instead of five).
// set bits of a wave mask below the calling lane (two instructions)
__device__ __forceinline__ int ag_lanes_below(unsigned long long m) {
  return __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(m), 0u));
}

#define AG_RC_RUN 16       // consecutive blocks' worth of positions a die takes at a time (64 positions)
#define AG_RC_PENDING 320  // candidates inside the error band a wave can set aside (they are decided when 64 have gathered)
// (What the count kernel reads of AgRadiusArgs: kernel arguments sit in scalar registers for the whole run, and the
// full set -- the grid tables among them -- pushed the loop's own scalars out into lanes of a vector register.)
struct AgRadiusHot {
  const int* order;     // the target at every position (its descriptor is desc[position]); null: position = target
  const AgRadiusDesc* desc;
  const float4* spts;
  const float2* src;
  const float2* tgt;
  int64_t* offsets;
  int32_t* stash;
  int64_t nt;
  double rthr;
  float c_in, c_out;
  int stash_cap;
  int run;              // consecutive blocks' worth of positions a die takes at a time
  const int* abort;
};

__global__ __launch_bounds__(256) void ag_radius_count_kernel(AgRadiusHot a) {
  if (ag_aborted(a.abort)) return;  // (a binning or a prefix sum upstream failed: see ag_aborted)
  __shared__ __attribute__((aligned(16))) int row_all[4][AG_STASH_MAX + 4];
  __shared__ int pending_all[4][AG_RC_PENDING];
  const int lane = ag_lane();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int* row = row_all[wave];
  int* pending = pending_all[wave];
  // Positions: the grid walks them in sweeps of one position per wave.  Within a sweep the blocks that share an L2
  // (b, b + 8, b + 16, ...) take runs of `run` consecutive blocks' worth of positions, the eight dies taking
  // the runs in turn: neighbours in the order are in flight on one die, and no die is handed a region of its own
  // (the work per target follows the density of the sources).  The grid is a multiple of 8 x run blocks.
  const int64_t nwaves = (int64_t(gridDim.x) * blockDim.x) >> 6;
  const int die = blockIdx.x & 7, on_die = blockIdx.x >> 3;
  const int64_t block = (int64_t(on_die / a.run) * 8 + die) * a.run + on_die % a.run;
  const int64_t first_pos = block * 4 + wave;
  // lanes 0 ... 7 hold the runs of a descriptor: (first, inclusive prefix of the lengths); every lane its unit vector
  auto fetch = [&](int64_t pos, int* vfirst, int* vincl, float4* unit, int* target) {
    const int64_t p = pos < a.nt ? pos : a.nt - 1;
    const AgRadiusDesc* d = a.desc + p;
    const int r = lane & (AG_RADIUS_RUNS - 1);
    *vfirst = d->first[r];
    *vincl = d->incl[r];
    *unit = d->unit;
    *target = a.order != nullptr ? a.order[p] : static_cast<int>(p);
  };
  int nfirst, nincl, ntarget;
  float4 nunit;
  if (first_pos < a.nt) fetch(first_pos, &nfirst, &nincl, &nunit, &ntarget);
  for (int64_t pos = first_pos; pos < a.nt; pos += nwaves) {
    const int vfirst = nfirst, vincl = nincl;
    const float4 tu = nunit;
    const int64_t t = __builtin_amdgcn_readfirstlane(ntarget);
    fetch(pos + nwaves, &nfirst, &nincl, &nunit, &ntarget);  // (past the end: the last descriptor again, unused)
    if (t < 0 || t >= a.nt) continue;
    const int nkept = __builtin_amdgcn_readfirstlane(__float_as_int(tu.w));
    int cnt = 0, npend = 0;
    // The window is walked as ONE sequence of candidates, the runs laid end to end, 64 candidates a piece: lane l
    // of piece p looks at candidate q = 64 p + l, which lies in the run j = number of runs that end at or before q
    // (compares against the scalar ends of the runs: two for the usual window of three runs), at the place
    // first[j] + q - (candidates before run j) of the binned array -- the second term comes from the lane that holds
    // run j.  Windows of many short runs (a sparse corner of a stretched grid: six runs of a dozen) fill their
    // lanes as well as windows of three runs of sixty.
    const int total = nkept > 0 ? __builtin_amdgcn_readlane(vincl, AG_RADIUS_RUNS - 1) : 0;
    const int before = __builtin_amdgcn_update_dpp(0, vincl, 0x111, 0xF, 0xF, true);  // row_shr:1: the end of the run before (lane 0: 0)
    const int voffset = vfirst - before;
    const int end0 = __builtin_amdgcn_readlane(vincl, 0), end1 = __builtin_amdgcn_readlane(vincl, 1);
    // the candidates set aside, decided by the float64 criterion (a candidate per lane)
    auto flush = [&]() {
      ag_wave_sync_lds();
      for (int p0 = 0; p0 < npend; p0 += 64) {
        bool ok = false;
        int id = 0;
        if (p0 + lane < npend) {
          id = pending[p0 + lane];
          const float2 tp = a.tgt[t];
          const float2 sp = a.src[id];
          ok = ag_rdist(static_cast<double>(tp.x), static_cast<double>(tp.y), sp.x, sp.y) <= a.rthr;
        }
        const unsigned long long mo = __builtin_amdgcn_ballot_w64(ok);
        if (ok) {
          const int oo = cnt + ag_lanes_below(mo);
          if (oo < a.stash_cap) row[oo] = id;
        }
        cnt += __popcll(mo);
      }
      ag_wave_sync_lds();
      npend = 0;
    };
    for (int q0 = 0; q0 < total; q0 += 256) {
      if (npend > AG_RC_PENDING - 256) flush();  // (room for every candidate of the four pieces that follow)
      float4 s[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        // (never predicated: a lane past the end reads the last candidate of the window and drops it)
        const int q = min(q0 + 64 * u + lane, total - 1);
        int j = (q >= end0 ? 1 : 0) + (q >= end1 ? 1 : 0);
        if (nkept > 3) {  // (wave-uniform)
#pragma unroll
          for (int i = 2; i < AG_RADIUS_RUNS - 1; ++i) j += q >= __builtin_amdgcn_readlane(vincl, i) ? 1 : 0;
        }
        const int idx = __builtin_amdgcn_ds_bpermute(j << 2, voffset) + q;
        s[u] = a.spts[idx];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int left = total - (q0 + 64 * u);  // candidates of this piece and the ones after it
        if (left <= 0) break;                    // (wave-uniform)
        const int id = __float_as_int(s[u].w);
        const float dx = s[u].x - tu.x, dy = s[u].y - tu.y, dz = s[u].z - tu.z;
        const float dd = dx * dx + dy * dy + dz * dz;
        const bool live = lane < left;
        const bool in = live && dd <= a.c_in;
        const bool maybe = live && dd <= a.c_out && !in;
        const unsigned long long m_in = __builtin_amdgcn_ballot_w64(in);
        if (in) {
          const int o = cnt + ag_lanes_below(m_in);
          if (o < a.stash_cap) row[o] = id;
        }
        cnt += __popcll(m_in);
        const unsigned long long mm = __builtin_amdgcn_ballot_w64(maybe);
        if (mm != 0ull) {  // (rare) inside the error band of the float32 test: set aside, decided in float64 later
          if (maybe) pending[npend + ag_lanes_below(mm)] = id;
          npend += __popcll(mm);
        }
      }
    }
    if (npend > 0) flush();
    if (lane == 0) a.offsets[t] = cnt;
    if (cnt > 0 && cnt <= a.stash_cap) {
      ag_wave_sync_lds();
      ag_row_rank_sort(row, cnt, a.stash + t * a.stash_cap);
    }
    ag_wave_sync_lds();  // the row is reused by the next target
  }
}

#define AG_COPY_GROUP 16  // targets per block of the copy kernel

// Fill pass, rows kept by the count pass: a block copies the rows of AG_COPY_GROUP consecutive
// targets; its threads stride over the block's contiguous piece of the output and find their row
// by binary search among the group's offsets (LDS).
__global__ __launch_bounds__(256) void ag_radius_copy_kernel(AgRadiusArgs a) {
  if (ag_aborted(a.abort)) return;  // (a binning or a prefix sum upstream failed: see ag_aborted)
  __shared__ int64_t off[AG_COPY_GROUP + 1];
  for (int64_t t0 = int64_t(blockIdx.x) * AG_COPY_GROUP; t0 < a.nt; t0 += int64_t(gridDim.x) * AG_COPY_GROUP) {
    const int n = static_cast<int>(min(int64_t(AG_COPY_GROUP), a.nt - t0));
    if (threadIdx.x <= n) off[threadIdx.x] = a.offsets[t0 + threadIdx.x];
    __syncthreads();
    const int64_t e_begin = off[0], e_end = off[n];
    for (int64_t e = e_begin + threadIdx.x; e < e_end; e += blockDim.x) {
      int lo = 0, hi = n - 1;  // last row whose offset is <= e
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (off[mid] <= e) lo = mid; else hi = mid - 1;
      }
      const int64_t t = t0 + lo;
      if (off[lo + 1] - off[lo] <= a.stash_cap) {
        a.out_src[e] = a.stash[t * a.stash_cap + (e - off[lo])];
        a.out_tgt[e] = static_cast<int32_t>(t);
      }
    }
    __syncthreads();
  }
}

// Fill pass, rows longer than the stash: searched again, the unsorted ids go to the row's piece of
// the TARGET column (scratch until the row is finished); every row also gets its segment
// [begin, end) for the segmented sort -- empty for the rows the copy kernel already wrote.
__global__ __launch_bounds__(256) void ag_radius_long_kernel(AgRadiusArgs a, uint32_t* __restrict__ seg_begin,
                                                             uint32_t* __restrict__ seg_end) {
  const bool aborted = ag_aborted(a.abort);  // (see ag_aborted: no row is searched, and the sort is left nothing to do)
  const int lane = ag_lane();
  const int64_t wave_global = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int64_t nwaves = (int64_t(gridDim.x) * blockDim.x) >> 6;
  for (int64_t t = wave_global; t < a.nt; t += nwaves) {
    if (aborted) {
      if (lane == 0) seg_begin[t] = seg_end[t] = 0u;
      continue;
    }
    const int64_t base_out = a.offsets[t];
    const int64_t n_row = a.offsets[t + 1] - base_out;
    const bool is_long = n_row > a.stash_cap;
    if (lane == 0) {
      seg_begin[t] = is_long ? static_cast<uint32_t>(base_out) : 0u;
      seg_end[t] = is_long ? static_cast<uint32_t>(base_out + n_row) : 0u;
    }
    if (is_long) (void)ag_radius_scan(a, t, nullptr, 0, a.out_tgt + base_out);
  }
}

// after the segmented sort: long rows get their sorted sources (when the sort left them in the
// target column) and their target ids
__global__ __launch_bounds__(256) void ag_radius_long_finish_kernel(AgRadiusArgs a, int sorted_in_tgt) {
  if (ag_aborted(a.abort)) return;  // (a binning or a prefix sum upstream failed: see ag_aborted)
  const int lane = ag_lane();
  const int64_t wave_global = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int64_t nwaves = (int64_t(gridDim.x) * blockDim.x) >> 6;
  for (int64_t t = wave_global; t < a.nt; t += nwaves) {
    const int64_t base_out = a.offsets[t];
    const int64_t n_row = a.offsets[t + 1] - base_out;
    if (n_row <= a.stash_cap) continue;
    for (int64_t e = lane; e < n_row; e += 64) {
      if (sorted_in_tgt) a.out_src[base_out + e] = a.out_tgt[base_out + e];
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int64_t e = lane; e < n_row; e += 64) a.out_tgt[base_out + e] = static_cast<int32_t>(t);
  }
}

struct AgRadiusWs {
  AgBinned src;
  unsigned long long* scan_state;
  int64_t scan_words;
  int32_t* stash;
  int stash_cap;
  AgRadiusDesc* desc;
  uint32_t *seg_begin, *seg_end;  // [nt] segments of the rows longer than the stash
  void* seg_temp;
  size_t seg_temp_bytes;
  // the order of the targets in the count pass (ag_radius_group_*)
  double* rwin;
  int *gcount, *gof, *gslot, *gorder;
  unsigned long long* gscan_state;
  int64_t gscan_words;
};

// groups the index of an ordered query may hold (a third of the index lies beyond the ends of the shorter rows)
static int64_t ag_radius_group_cap(int64_t nt) { return nt / 2 + 4096; }

// Whether the count pass takes its targets in group order, and the shape of a group: only when the caller has
// asked for it (ag_radius_group_order).  Measured on the O1280 -> r7 encoder, the order cuts the fabric traffic of the count
// kernel from 427 MB to 174 MB (7.7 x -> 3.1 x the bytes of the sources) and leaves its duration where it is
// (0.099 -> 0.101 ms: the kernel waits for instruction issue, not for memory), while the three launches of the
// counting sort of the targets add 0.04 ms to the query; 5e8 sources, 1e7 targets: 8.9 -> 8.2 ms for the kernel,
// 1.7 ms for the sorting.  So the caller's order is what the library uses.
static std::atomic<int> ag_radius_group_order_flag{0};

extern "C" int ag_radius_group_order(int on) { return ag_radius_group_order_flag.exchange(on != 0 ? 1 : 0); }

struct AgRadiusPlan {
  bool on = false;
  int B = 1, W = 1, nbg = 1, nch_max = 1;
  int64_t ngroups = 1;
  double rpad = 0.0;
};

static AgRadiusPlan ag_radius_plan(int64_t nt, int nb, double radius) {
  AgRadiusPlan p;
  if (ag_radius_group_order_flag.load() == 0 || nt < 1 || !(radius < AG_HALF_PI)) return p;
  const double dlat = AG_PI / nb;
  p.rpad = radius + 1.0e-7 + 1.0e-6 * radius;
  const double ncells = 4.0 * nb * double(nb) / AG_PI;
  const double tpc = static_cast<double>(nt) / ncells;  // targets per cell of the source grid
  // A group: a dozen bands tall, a few cells wide, two dozen targets or so; wider where the targets are sparse.
  p.B = nb < 12 ? nb : 12;
  const double w = 24.0 / (tpc * p.B > 1.0e-9 ? tpc * p.B : 1.0e-9);
  p.W = w < 2.0 ? 2 : (w > 64.0 ? 64 : static_cast<int>(w));
  p.nbg = (nb + p.B - 1) / p.B;
  const int ncols_max = static_cast<int>(AG_TWO_PI / dlat) + 2;
  p.nch_max = (ncols_max + p.W - 1) / p.W;
  p.ngroups = int64_t(p.nbg) * p.nch_max;
  while (p.ngroups > ag_radius_group_cap(nt) && p.W < (1 << 20)) {  // (sparse targets: coarser groups)
    p.W *= 2;
    p.nch_max = (ncols_max + p.W - 1) / p.W;
    p.ngroups = int64_t(p.nbg) * p.nch_max;
  }
  p.on = p.ngroups <= ag_radius_group_cap(nt);
  return p;
}

// cell size for a radius query: the query radius itself, but never so small that the grid has
// many more cells than there are sources
static double ag_radius_delta(int64_t ns, double radius) {
  const double spacing = sqrt(4.0 * AG_PI / static_cast<double>(ns));
  const double r = radius * 1.001 + 4.0e-7;  // > radius + pad: a window never spans more than 3 bands
  double d = r > spacing ? r : spacing;
  if (d > AG_PI) d = AG_PI;
  return d;
}

static int ag_radius_prepare(AgCarver& c, const float* d_src, int64_t ns, int64_t nt, double radius, bool build,
                             hipStream_t stream, AgRadiusWs* w, const AgKeptBinning& kept_src = AgKeptBinning()) {
  // The carve must not depend on the radius (the workspace is sized before the radius is
  // known): the cell arrays are reserved for the finest grid any radius can ask for.
  const int nb_reserve = ag_grid_bands(sqrt(4.0 * AG_PI / static_cast<double>(ns)));
  const int nb = ag_grid_bands(ag_radius_delta(ns, radius));
  if (kept_src.buffer != nullptr) {
    // the caller's binning of the sources (any cell size: the windows adapt): nothing to bin or carve here
    if (w) ag_kept_describe(kept_src, ns, &w->src);
  } else if (w == nullptr) {
    ag_bin_points(c, nullptr, ns, nb_reserve, nb_reserve, 1, true, stream, nullptr);
  } else if (build) {
    int rc = ag_bin_points(c, d_src, ns, nb, nb_reserve, 1, true, stream, &w->src);
    if (rc) return rc;
  } else {
    // descriptor of the grid the count pass built in this workspace (same carve, no kernels)
    AgBinCarve bc;
    ag_bin_carve(c, ns, nb_reserve, 1, true, &bc);
    ag_bin_describe(bc, ns, nb, 1, &w->src);
  }
  const int64_t scan_words = static_cast<int64_t>(ag_chained_scan_state_words_i64(nt));
  unsigned long long* scan_state = c.take<unsigned long long>(scan_words);
  const int cap = ag_radius_stash_cap(nt);
  int32_t* stash = c.take<int32_t>(nt * cap + 4);
  AgRadiusDesc* desc = c.take<AgRadiusDesc>(nt);
  uint32_t* seg_begin = c.take<uint32_t>(nt);
  uint32_t* seg_end = c.take<uint32_t>(nt);
  const size_t seg_temp_bytes = ag_segmented_sort_i32_temp_bytes(nt);
  void* seg_temp = c.take<char>(seg_temp_bytes);
  // (the order of the targets: sized for any radius, like everything above)
  const int64_t gcap = ag_radius_group_cap(nt);
  double* rwin = c.take<double>(nb_reserve);
  int* gcount = c.take<int>(gcap + 1);
  int* gof = c.take<int>(nt);
  int* gslot = c.take<int>(nt);
  int* gorder = c.take<int>(nt);
  const int64_t gscan_words = static_cast<int64_t>(ag_chained_scan_state_words_i32(gcap));
  unsigned long long* gscan_state = c.take<unsigned long long>(gscan_words);
  if (w) {
    w->rwin = rwin;
    w->gcount = gcount;
    w->gof = gof;
    w->gslot = gslot;
    w->gorder = gorder;
    w->gscan_state = gscan_state;
    w->gscan_words = gscan_words;
    w->seg_begin = seg_begin;
    w->seg_end = seg_end;
    w->seg_temp = seg_temp;
    w->seg_temp_bytes = seg_temp_bytes;
    w->desc = desc;
    w->scan_state = scan_state;
    w->scan_words = scan_words;
    w->stash = stash;
    w->stash_cap = cap;
  }
  return AG_OK;
}

extern "C" size_t ag_radius_workspace_bytes(int64_t ns, int64_t nt) {
  if (ns < 1 || nt < 0) return 0;
  AgCarver c(nullptr);
  ag_radius_prepare(c, nullptr, ns, nt > 0 ? nt : 1, 0.0, false, nullptr, nullptr);
  return c.total();
}

static int ag_radius_args(const float* d_src, const float* d_tgt, int64_t nt, double radius, const AgRadiusWs& w,
                          AgRadiusArgs* a) {
  a->grid = w.src.grid;
  a->spts = w.src.pts;
  a->sstart = w.src.cell_start;
  a->src = reinterpret_cast<const float2*>(d_src);
  a->tgt = reinterpret_cast<const float2*>(d_tgt);
  a->nt = nt;
  a->stash = w.stash;
  a->stash_cap = w.stash_cap;
  a->desc = w.desc;
  a->scan_state = nullptr;
  a->scan_words = 0;
  a->abort = ag_status_word();
  a->radius = radius < AG_PI ? radius : AG_PI;
  const double s = sin(0.5 * radius);
  a->rthr = s * s;
  const double c2 = 4.0 * s * s;
  // both unit vectors are rounded from float64 here: coordinate error <= 2^-24
  const double err = 6.0e-7 * sqrt(c2) + 4.0e-7 * c2 + 2.0e-13;
  a->c_in = static_cast<float>((c2 - 2.0 * err) * (1.0 - 1.0e-6));
  a->c_out = static_cast<float>((c2 + 2.0 * err) * (1.0 + 1.0e-6));
  if (radius >= AG_PI) {  // everything matches: rdist <= 1 always
    a->rthr = 2.0;
    a->c_in = 8.0f;
    a->c_out = 8.0f;
  }
  return AG_OK;
}

static size_t ag_radius_ws_bytes(int64_t ns, int64_t nt, const AgKeptBinning& kept) {
  AgCarver c(nullptr);
  ag_radius_prepare(c, nullptr, ns, nt > 0 ? nt : 1, 0.0, false, nullptr, nullptr, kept);
  return c.total();
}

extern "C" size_t ag_radius_binned_workspace_bytes(int64_t ns, int64_t nt) {
  if (ns < 1 || nt < 0) return 0;
  AgKeptBinning sizing;
  sizing.buffer = reinterpret_cast<const void*>(uintptr_t(256));  // (sizing only: never dereferenced)
  sizing.cell = 1.0;
  return ag_radius_ws_bytes(ns, nt, sizing);
}

static int ag_radius_count_impl(const float* d_src, int64_t ns, const AgKeptBinning& kept, const float* d_tgt, int64_t nt,
                                double radius, int64_t* d_offsets, void* d_ws, size_t ws_bytes, void* stream) {
  AG_REQUIRE(ns >= 1 && nt >= 0, "ag_radius_count: bad sizes");
  AG_REQUIRE(ns < (int64_t(1) << 31) && nt < (int64_t(1) << 31), "ag_radius_count: node counts must fit int32");
  AG_REQUIRE(radius >= 0.0, "ag_radius_count: negative radius");
  AG_REQUIRE(d_src && d_offsets && d_ws, "ag_radius_count: null pointer");
  if (ws_bytes < ag_radius_ws_bytes(ns, nt, kept)) return ag_fail(AG_ERR_WORKSPACE, "ag_radius_count: workspace too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  AgCarver c(d_ws);
  AgRadiusWs w;
  int rc = ag_radius_prepare(c, d_src, ns, nt > 0 ? nt : 1, radius, true, s, &w, kept);
  if (rc) return rc;
  AgRadiusArgs a;
  ag_radius_args(d_src, d_tgt, nt, radius, w, &a);
  a.offsets = d_offsets;
  a.out_src = nullptr;
  a.out_tgt = nullptr;
  a.scan_state = w.scan_state;
  a.scan_words = w.scan_words;
  if (nt > 0) {
    const AgRadiusPlan plan = ag_radius_plan(nt, w.src.grid.nb, radius);
    AgRadiusHot hot;
    hot.order = nullptr;
    if (plan.on) {
      AgRadiusGroups gr;
      gr.B = plan.B;
      gr.W = plan.W;
      gr.nbg = plan.nbg;
      gr.nch_max = plan.nch_max;
      gr.ngroups = plan.ngroups;
      gr.rpad = plan.rpad;
      gr.rwin = w.rwin;
      gr.gcount = w.gcount;
      gr.gof = w.gof;
      gr.gslot = w.gslot;
      gr.gorder = w.gorder;
      gr.gscan_state = w.gscan_state;
      gr.gscan_words = w.gscan_words;
      const int64_t init_items = std::max<int64_t>(std::max<int64_t>(gr.ngroups + 1, gr.gscan_words),
                                                   std::max<int64_t>(a.scan_words, a.grid.nb));
      AG_KERNEL("ag_radius_groups_init_kernel", ag_radius_groups_init_kernel, dim3(ag_blocks(init_items, 256, 4096)),
                dim3(256), s, a, gr);
      AG_LAUNCH_CHECK();
      AG_KERNEL("ag_radius_group_count_kernel", ag_radius_group_count_kernel, dim3(ag_blocks(nt, 256, 16384)), dim3(256),
                s, a, gr);
      AG_LAUNCH_CHECK();
      rc = ag_chained_scan_i32(gr.gcount, gr.gcount, gr.ngroups, gr.gscan_state, s);  // total -> [ngroups]
      if (rc) return rc;
      AG_KERNEL("ag_radius_group_desc_kernel", ag_radius_group_desc_kernel, dim3(ag_blocks(nt, 256, 16384)), dim3(256), s,
                a, gr);
      AG_LAUNCH_CHECK();
      hot.order = w.gorder;
    } else {
      AG_KERNEL("ag_radius_window_kernel", ag_radius_window_kernel, dim3(ag_blocks(nt, 256, 16384)), dim3(256), s, a);
      AG_LAUNCH_CHECK();
    }
    hot.desc = a.desc;
    hot.spts = a.spts;
    hot.src = a.src;
    hot.tgt = a.tgt;
    hot.offsets = a.offsets;
    hot.stash = a.stash;
    hot.nt = a.nt;
    hot.rthr = a.rthr;
    hot.c_in = a.c_in;
    hot.c_out = a.c_out;
    hot.stash_cap = a.stash_cap;
    hot.abort = a.abort;
    // A wave per position and sweep, a few sweeps; the grid a multiple of 8 x AG_RC_RUN blocks (see the kernel).
    hot.run = AG_RC_RUN;
    const int unit = 8 * AG_RC_RUN;
    const int blocks = (ag_blocks(nt, 4, 7168) + unit - 1) / unit * unit;
    AG_KERNEL("ag_radius_count_kernel", ag_radius_count_kernel, dim3(blocks), dim3(256), s, hot);
    AG_LAUNCH_CHECK();
  }
  // counts -> exclusive offsets, offsets[nt] = E, offsets[nt + 1] = the longest row, in one launch
  if (nt == 0) return ag_zero_i32(reinterpret_cast<int*>(d_offsets), 4, s);
  return ag_chained_scan_i64(d_offsets, d_offsets, nt, w.scan_state, 1, s);
}

extern "C" int ag_radius_count(const float* d_src, int64_t ns, const float* d_tgt, int64_t nt, double radius,
                               int64_t* d_offsets, void* d_ws, size_t ws_bytes, void* stream) {
  return ag_radius_count_impl(d_src, ns, AgKeptBinning(), d_tgt, nt, radius, d_offsets, d_ws, ws_bytes, stream);
}

extern "C" int ag_radius_count_binned(const float* d_src, int64_t ns, const void* d_src_binning, double src_cell,
                                      const float* d_tgt, int64_t nt, double radius, int64_t* d_offsets, void* d_ws,
                                      size_t ws_bytes, void* stream) {
  AG_REQUIRE(d_src_binning != nullptr && src_cell > 0.0, "ag_radius_count_binned: a binning and the cell size it was built with");
  AgKeptBinning kept;
  kept.buffer = d_src_binning;
  kept.cell = src_cell;
  return ag_radius_count_impl(d_src, ns, kept, d_tgt, nt, radius, d_offsets, d_ws, ws_bytes, stream);
}

static int ag_radius_fill_impl(const float* d_src, int64_t ns, const AgKeptBinning& kept, const float* d_tgt, int64_t nt,
                               double radius, const int64_t* d_offsets, int64_t num_edges, int64_t max_row,
                               int32_t* d_out_src, int32_t* d_out_tgt, void* d_ws, size_t ws_bytes, void* stream) {
  AG_REQUIRE(ns >= 1 && nt >= 0 && num_edges >= 0 && max_row >= 0, "ag_radius_fill: bad sizes");
  if (nt == 0 || num_edges == 0) return AG_OK;
  AG_REQUIRE(d_src && d_tgt && d_offsets && d_out_src && d_out_tgt && d_ws, "ag_radius_fill: null pointer");
  if (ws_bytes < ag_radius_ws_bytes(ns, nt, kept)) return ag_fail(AG_ERR_WORKSPACE, "ag_radius_fill: workspace too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  AgCarver c(d_ws);
  AgRadiusWs w;
  int rc = ag_radius_prepare(c, d_src, ns, nt, radius, false, s, &w, kept);
  if (rc) return rc;
  AgRadiusArgs a;
  ag_radius_args(d_src, d_tgt, nt, radius, w, &a);
  a.offsets = const_cast<int64_t*>(d_offsets);
  a.out_src = d_out_src;
  a.out_tgt = d_out_tgt;
  if (a.stash_cap > 0) {
    AG_KERNEL("ag_radius_copy_kernel", ag_radius_copy_kernel, dim3(ag_blocks(nt, AG_COPY_GROUP, 16384)), dim3(256), s, a);
    AG_LAUNCH_CHECK();
  }
  if (max_row <= a.stash_cap) return AG_OK;
  // rows longer than the stash: search again into the target column, sort every such row by source
  // id with one segmented radix sort, then write the target ids
  AG_REQUIRE(num_edges < (int64_t(1) << 32), "ag_radius_fill: rows longer than %d entries need fewer than 2^32 edges",
             a.stash_cap);
  if (ag_radius_plan(nt, w.src.grid.nb, radius).on) {
    // (the count pass left its descriptors in group order: written again by target for the rows searched again)
    AG_KERNEL("ag_radius_window_kernel", ag_radius_window_kernel, dim3(ag_blocks(nt, 256, 16384)), dim3(256), s, a);
    AG_LAUNCH_CHECK();
  }
  AG_KERNEL("ag_radius_long_kernel", ag_radius_long_kernel, dim3(ag_blocks(nt, 4, 16384)), dim3(256), s, a, w.seg_begin,
            w.seg_end);
  AG_LAUNCH_CHECK();
  int id_bits = 1;
  while ((int64_t(1) << id_bits) < ns) ++id_bits;
  bool sorted_in_src = false;
  rc = ag_segmented_sort_i32(w.seg_temp, w.seg_temp_bytes, d_out_tgt, d_out_src, num_edges, nt, w.seg_begin, w.seg_end,
                             id_bits, &sorted_in_src, s);
  if (rc) return rc;
  AG_KERNEL("ag_radius_long_finish_kernel", ag_radius_long_finish_kernel, dim3(ag_blocks(nt, 4, 16384)), dim3(256), s, a,
            sorted_in_src ? 0 : 1);
  AG_LAUNCH_CHECK();
  return AG_OK;
}

extern "C" int ag_radius_fill(const float* d_src, int64_t ns, const float* d_tgt, int64_t nt, double radius,
                              const int64_t* d_offsets, int64_t num_edges, int64_t max_row, int32_t* d_out_src,
                              int32_t* d_out_tgt, void* d_ws, size_t ws_bytes, void* stream) {
  return ag_radius_fill_impl(d_src, ns, AgKeptBinning(), d_tgt, nt, radius, d_offsets, num_edges, max_row, d_out_src,
                             d_out_tgt, d_ws, ws_bytes, stream);
}

extern "C" int ag_radius_fill_binned(const float* d_src, int64_t ns, const void* d_src_binning, double src_cell,
                                     const float* d_tgt, int64_t nt, double radius, const int64_t* d_offsets,
                                     int64_t num_edges, int64_t max_row, int32_t* d_out_src, int32_t* d_out_tgt, void* d_ws,
                                     size_t ws_bytes, void* stream) {
  AG_REQUIRE(d_src_binning != nullptr && src_cell > 0.0, "ag_radius_fill_binned: a binning and the cell size it was built with");
  AgKeptBinning kept;
  kept.buffer = d_src_binning;
  kept.cell = src_cell;
  return ag_radius_fill_impl(d_src, ns, kept, d_tgt, nt, radius, d_offsets, num_edges, max_row, d_out_src, d_out_tgt, d_ws,
                             ws_bytes, stream);
}

// the cell size ag_radius_count chooses for its own binning of ns sources and this radius: what a caller who
// keeps the binning (ag_binning_build) passes as `cell`
extern "C" double ag_radius_cell(int64_t ns, double radius) { return ag_radius_delta(ns > 0 ? ns : 1, radius); }
