// Spatial binning on the sphere, k-nearest-neighbour, nearest-neighbour reference distance and
// radius (cut-off) queries for gfx950.
//
// Design
//  * Points are binned into the latitude-band grid of ag_common.hpp by a stable radix sort of
//    (cell << 32 | id) keys, so the binned order -- and with it every result -- is deterministic.
//  * k-NN, stage 1 (ag_knn_cell_kernel): one wave per target cell.  The sources of the 3 bands
//    around the cell are staged once into an LDS tile as float32 unit vectors; every lane owns
//    one target of the cell and scans the tile (broadcast LDS reads) keeping a sorted
//    register list of its CAP best float32 chord^2 values.  A target is final when (a) its
//    k-th chord provably lies inside the region the tile covers and (b) the float32 gaps around
//    the k-th entry exceed the float32 error bound; the others are queued.
//  * k-NN, stage 2 (ag_knn_exact_kernel): one wave per queued target, float64 haversine rdist
//    (scikit-learn's formula) over a window that doubles until it provably holds the k nearest;
//    the k smallest (rdist, id) pairs are extracted by k wavefront min-reductions.
//  * Radius query: one wave per target; lanes stride over the contiguous runs of binned sources
//    of the covering cells (coalesced 16-byte loads), float32 chord test with a float64
//    fallback inside the error band, ballot-compacted writes.
#include "ag_common.hpp"

// ---------------------------------------------------------------------------------------------
// grid construction
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void ag_band_table_kernel(int nb, double dlat, int* __restrict__ ncols,
                                                             int* __restrict__ off) {
  __shared__ int partial[1024];
  const int tid = threadIdx.x;
  const int per = (nb + 1023) / 1024;
  const int b_begin = tid * per, b_end = min(nb, b_begin + per);
  int sum = 0;
  for (int b = b_begin; b < b_end; ++b) {
    const double mid = AG_HALF_PI - (b + 0.5) * dlat;
    int nc = static_cast<int>(floor(AG_TWO_PI * cos(mid) / dlat));
    nc = max(nc, 1);
    ncols[b] = nc;
    sum += nc;
  }
  partial[tid] = sum;
  __syncthreads();
  // exclusive scan of the 1024 partial sums (Hillis-Steele)
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

__global__ void ag_cell_key_kernel(AgGrid g, const float2* __restrict__ pts, int64_t n, uint64_t* __restrict__ keys) {
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
    const float2 p = pts[i];
    const int c = ag_cell_of(g, static_cast<double>(p.x), static_cast<double>(p.y));
    keys[i] = (static_cast<uint64_t>(c) << 32) | static_cast<uint32_t>(i);
  }
}

// cell_start[c] = first binned position whose cell id is >= c  (c in [0, ncells_ub])
__global__ void ag_cell_start_kernel(const uint64_t* __restrict__ keys, int64_t n, int64_t ncells_ub,
                                     int* __restrict__ cell_start) {
  for (int64_t c = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; c <= ncells_ub; c += int64_t(gridDim.x) * blockDim.x) {
    const uint64_t want = static_cast<uint64_t>(c) << 32;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (keys[mid] < want) lo = mid + 1; else hi = mid;
    }
    cell_start[c] = static_cast<int>(lo);
  }
}

__global__ void ag_binned_points_kernel(const uint64_t* __restrict__ keys, const float2* __restrict__ pts, int64_t n,
                                        float4* __restrict__ out) {
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
    const uint32_t id = static_cast<uint32_t>(keys[i]);
    const float2 p = pts[id];
    const float3 u = ag_unit_vec(p.x, p.y);
    out[i] = make_float4(u.x, u.y, u.z, __int_as_float(static_cast<int>(id)));
  }
}

int ag_grid_bands(double delta) {
  if (!(delta > 0.0)) delta = AG_PI;
  double nb = floor(AG_PI / delta);
  if (nb < 1.0) nb = 1.0;
  if (nb > 32768.0) nb = 32768.0;
  return static_cast<int>(nb);
}

static int64_t ag_grid_cells_ub(int nb) {
  const double dlat = AG_PI / nb;
  return static_cast<int64_t>(4.0 * AG_PI / (dlat * dlat) * 1.02) + nb + 64;
}

int ag_bin_points(AgCarver& ws, const float* d_pts, int64_t n, int nb, int nb_reserve, bool want_pts,
                  hipStream_t stream, AgBinned* out) {
  if (nb_reserve < nb) nb_reserve = nb;
  const int64_t ncells_ub = ag_grid_cells_ub(nb);
  int* ncols = ws.take<int>(nb_reserve);
  int* off = ws.take<int>(nb_reserve + 1);
  int* cell_start = ws.take<int>(ag_grid_cells_ub(nb_reserve) + 1);
  uint64_t* keys_a = ws.take<uint64_t>(n);
  uint64_t* keys_b = ws.take<uint64_t>(n);
  const size_t temp_bytes = ag_sort_u64_temp_bytes(n);
  void* temp = ws.take<char>(temp_bytes);
  float4* pts = want_pts ? ws.take<float4>(n) : nullptr;
  if (out == nullptr) return AG_OK;  // sizing pass

  AgGrid g;
  g.nb = nb;
  g.dlat = AG_PI / nb;
  g.inv_dlat = nb / AG_PI;
  g.ncols = ncols;
  g.off = off;
  AG_KERNEL("ag_band_table_kernel", ag_band_table_kernel, dim3(1), dim3(1024), stream, nb, g.dlat, ncols, off);
  AG_LAUNCH_CHECK();
  const int blocks = ag_blocks(n, 256, 16384);
  AG_KERNEL("ag_cell_key_kernel", ag_cell_key_kernel, dim3(blocks), dim3(256), stream, g, reinterpret_cast<const float2*>(d_pts), n, keys_a);
  AG_LAUNCH_CHECK();
  int cell_bits = 1;
  while ((int64_t(1) << cell_bits) <= ncells_ub) ++cell_bits;
  int rc = ag_sort_u64(temp, temp_bytes, keys_a, keys_b, n, 32, 32 + cell_bits, stream);
  if (rc) return rc;
  AG_KERNEL("ag_cell_start_kernel", ag_cell_start_kernel, dim3(ag_blocks(ncells_ub + 1, 256, 16384)), dim3(256), stream, keys_b, n, ncells_ub, cell_start);
  AG_LAUNCH_CHECK();
  if (want_pts) {
    AG_KERNEL("ag_binned_points_kernel", ag_binned_points_kernel, dim3(blocks), dim3(256), stream, keys_b, reinterpret_cast<const float2*>(d_pts), n, pts);
    AG_LAUNCH_CHECK();
  }
  out->grid = g;
  out->n = n;
  out->ncells_ub = ncells_ub;
  out->pts = pts;
  out->keys = keys_b;
  out->cell_start = cell_start;
  return AG_OK;
}

// ---------------------------------------------------------------------------------------------
// k-NN stage 1: cell tiles in LDS, float32 chord^2, register top-CAP
// ---------------------------------------------------------------------------------------------
#define AG_TILE 256        // staged sources per wave and pass
#define AG_WAVES_PER_BLOCK 4

struct AgKnnArgs {
  AgGrid grid;
  const float4* spts;        // binned sources
  const int* sstart;         // source cell starts
  const uint64_t* tkeys;     // binned target keys (cell << 32 | id)
  const int* tstart;         // target cell starts (same grid)
  const float2* tgt;         // target coordinates, original order
  const float2* src;         // source coordinates, original order
  int64_t ns, nt;
  int k;
  double rg;                 // angular radius the cell tile is guaranteed to cover
  float rg_chord2;           // conservative float32 chord^2 of rg
  int32_t* out_src;          // [nt*k]
  double* out_dist;          // [nt*k] or null
  int* queue;                // [nt] targets left to stage 2
  int* queue_count;
  int self_query;            // 1: reference-distance mode (sources == targets, k == 2)
  unsigned long long* max_bits;  // reference-distance mode: running max of rdist (as uint64 bits)
  long long* stats;          // [4]
};

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

template <int CAP>
__global__ __launch_bounds__(64 * AG_WAVES_PER_BLOCK) void ag_knn_cell_kernel(AgKnnArgs a) {
  __shared__ float4 tile_all[AG_WAVES_PER_BLOCK][AG_TILE];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = ag_lane();
  float4* tile = tile_all[wave];
  const int ncells = a.grid.off[a.grid.nb];
  const int cell = __builtin_amdgcn_readfirstlane(blockIdx.x * AG_WAVES_PER_BLOCK + wave);
  if (cell >= ncells) return;
  const int t0 = a.tstart[cell], t1 = a.tstart[cell + 1];
  if (t0 >= t1) return;

  // band and column of this cell (binary search in the band offsets)
  int blo = 0, bhi = a.grid.nb - 1;
  while (blo < bhi) {
    const int mid = (blo + bhi + 1) >> 1;
    if (a.grid.off[mid] <= cell) blo = mid; else bhi = mid - 1;
  }
  const int band = blo;
  const int col = cell - a.grid.off[band];
  const int nc = a.grid.ncols[band];
  const double lat_hi = AG_HALF_PI - band * a.grid.dlat;
  const double lat_lo = lat_hi - a.grid.dlat;
  const double wlon = AG_TWO_PI / nc;
  const double lon_lo = col * wlon, lon_hi = (col + 1) * wlon;
  const int k = a.k;

  for (int tb = t0; tb < t1; tb += 64) {
    const int ti = tb + lane;
    const bool active = ti < t1;
    int tid = -1;
    float3 tu = make_float3(0.f, 0.f, 0.f);
    float2 tp = make_float2(0.f, 0.f);
    if (active) {
      tid = static_cast<int>(static_cast<uint32_t>(a.tkeys[ti]));
      tp = a.tgt[tid];
      tu = ag_unit_vec(tp.x, tp.y);
    }
    float bd[CAP];
    int bi[CAP];
#pragma unroll
    for (int j = 0; j < CAP; ++j) {
      bd[j] = 3.0e38f;
      bi[j] = -1;
    }
    int fill = 0;
    auto scan_tile = [&](int count) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      for (int j = 0; j < count; ++j) {
        const float4 s = tile[j];
        const float dx = s.x - tu.x, dy = s.y - tu.y, dz = s.z - tu.z;
        const float d = dx * dx + dy * dy + dz * dz;
        ag_topk_insert<CAP>(bd, bi, d, __float_as_int(s.w));
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    };
    ag_for_each_range(a.grid, a.sstart, lat_lo, lat_hi, lon_lo, lon_hi, a.rg, [&](int first, int last) {
      int pos = first;
      while (pos < last) {
        const int take = min(last - pos, AG_TILE - fill);
        for (int i = lane; i < take; i += 64) tile[fill + i] = a.spts[pos + i];
        fill += take;
        pos += take;
        if (fill == AG_TILE) {
          scan_tile(fill);
          fill = 0;
        }
      }
    });
    if (fill > 0) scan_tile(fill);

    if (!active) continue;
    // (a) coverage: the k-th chord must lie inside the guaranteed radius;
    // (b) the float32 order around the k-th entry must be beyond the rounding error
    const float dk = bd[k - 1];
    bool settled = (bi[k - 1] >= 0) && (dk + ag_chord2_err(dk) <= a.rg_chord2);
    if (settled) {
      const float dn = bd[k];  // CAP >= k + 1
      settled = (dn - dk) > (ag_chord2_err(dn) + ag_chord2_err(dk));
    }
    bool order_ok = true;
    if (settled) {
#pragma unroll
      for (int j = 0; j + 1 < CAP; ++j)
        if (j + 1 < k) order_ok = order_ok && ((bd[j + 1] - bd[j]) > (ag_chord2_err(bd[j + 1]) + ag_chord2_err(bd[j])));
    }
    if (a.self_query) {
      // reference distance: the 2nd smallest float64 rdist of this point's neighbourhood
      if (settled) {
        double r0 = 0.0, r1 = 0.0;
        {
          const float2 p = a.src[bi[0]];
          r0 = ag_rdist(tp.x, tp.y, p.x, p.y);
        }
        {
          const float2 p = a.src[bi[1]];
          r1 = ag_rdist(tp.x, tp.y, p.x, p.y);
        }
        const double r = fmax(r0, r1);
        if (r > 0.0) atomicMax(a.max_bits, static_cast<unsigned long long>(__double_as_longlong(r)));
      } else {
        const int slot = atomicAdd(a.queue_count, 1);
        a.queue[slot] = tid;
      }
      continue;
    }
    if (settled && order_ok) {
      int32_t* o = a.out_src + static_cast<int64_t>(tid) * k;
#pragma unroll
      for (int j = 0; j < CAP; ++j)
        if (j < k) o[j] = bi[j];
      if (a.out_dist != nullptr) {
        double* od = a.out_dist + static_cast<int64_t>(tid) * k;
#pragma unroll
        for (int j = 0; j < CAP; ++j)
          if (j < k) {
            const float2 p = a.src[bi[j]];
            od[j] = ag_rdist_to_dist(ag_rdist(tp.x, tp.y, p.x, p.y));
          }
      }
    } else {
      const int slot = atomicAdd(a.queue_count, 1);
      a.queue[slot] = tid;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// k-NN stage 2: float64, one wave per queued target, k wavefront min-reductions
// ---------------------------------------------------------------------------------------------
struct AgPair {
  double r;
  int id;
};

__device__ __forceinline__ bool ag_pair_less(double r1, int i1, double r2, int i2) {
  return (r1 < r2) || (r1 == r2 && i1 < i2);
}

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

__global__ __launch_bounds__(256) void ag_knn_exact_kernel(AgKnnArgs a) {
  const int lane = ag_lane();
  const int wave_global = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const int nq = *a.queue_count;
  const int k = a.k;
  for (int q = wave_global; q < nq; q += nwaves) {
    const int tid = a.queue[q];
    const float2 tpf = a.tgt[tid];
    const double tlat = tpf.x, tlon = tpf.y;
    const double tlonw = ag_wrap_lon(tlon);
    double R = a.rg;
    int rounds = 0;
    for (;;) {
      const bool whole = R >= AG_PI;
      const double Rw = whole ? AG_PI : R;
      const double s = sin(0.5 * Rw);
      const double rlimit = s * s * (1.0 - 1.0e-9);
      double prev_r = -1.0;
      int prev_i = -1;
      int found = 0;
      bool covered = true;
      for (int j = 0; j < k; ++j) {
        double best_r = 1.0e300;
        int best_i = 0x7fffffff;
        ag_for_each_range(a.grid, a.sstart, tlat, tlat, tlonw, tlonw, Rw, [&](int first, int last) {
          for (int p = first + lane; p < last; p += 64) {
            const int id = __float_as_int(a.spts[p].w);
            const float2 sp = a.src[id];
            const double r = ag_rdist(tlat, tlon, sp.x, sp.y);
            if (ag_pair_less(prev_r, prev_i, r, id) && ag_pair_less(r, id, best_r, best_i)) {
              best_r = r;
              best_i = id;
            }
          }
        });
        const AgPair m = ag_wave_min_pair(best_r, best_i);
        if (m.id == 0x7fffffff) {
          covered = false;
          break;
        }
        if (!whole && !(m.r <= rlimit)) {
          covered = false;
          break;
        }
        prev_r = m.r;
        prev_i = m.id;
        ++found;
        if (a.self_query) {
          if (j == 1 && lane == 0 && m.r > 0.0)
            atomicMax(a.max_bits, static_cast<unsigned long long>(__double_as_longlong(m.r)));
        } else if (lane == 0) {
          a.out_src[static_cast<int64_t>(tid) * k + j] = m.id;
          if (a.out_dist != nullptr) a.out_dist[static_cast<int64_t>(tid) * k + j] = ag_rdist_to_dist(m.r);
        }
      }
      if (covered || whole) break;
      R *= 2.0;
      ++rounds;
    }
    if (lane == 0 && a.stats != nullptr) {
      atomicAdd(reinterpret_cast<unsigned long long*>(a.stats + 0), 1ull);
      if (rounds) atomicAdd(reinterpret_cast<unsigned long long*>(a.stats + 1), static_cast<unsigned long long>(rounds));
    }
  }
}

__global__ void ag_knn_init_kernel(int* queue_count, unsigned long long* max_bits, long long* stats) {
  if (threadIdx.x == 0) {
    *queue_count = 0;
    if (max_bits) *max_bits = 0ull;
    if (stats) {
      stats[0] = stats[1] = stats[2] = stats[3] = 0;
    }
  }
}

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

struct AgKnnWs {
  AgBinned src, tgt;
  int* queue;
  int* queue_count;
  unsigned long long* max_bits;
};

static int ag_knn_prepare(AgCarver& c, const float* d_src, int64_t ns, const float* d_tgt, int64_t nt, int k,
                          bool self_query, hipStream_t stream, AgKnnWs* w) {
  const int nb = ag_grid_bands(ag_knn_delta(ns, k));
  int rc = ag_bin_points(c, d_src, ns, nb, nb, true, stream, w ? &w->src : nullptr);
  if (rc) return rc;
  if (!self_query) {
    rc = ag_bin_points(c, d_tgt, nt, nb, nb, false, stream, w ? &w->tgt : nullptr);
    if (rc) return rc;
  } else if (w) {
    w->tgt = w->src;
  }
  int* queue = c.take<int>(nt);
  int* queue_count = c.take<int>(4);
  unsigned long long* max_bits = c.take<unsigned long long>(2);
  if (w) {
    w->queue = queue;
    w->queue_count = queue_count;
    w->max_bits = max_bits;
  }
  return AG_OK;
}

template <int CAP>
static void ag_launch_cell_kernel(const AgKnnArgs& a, int64_t ncells_ub, hipStream_t stream) {
  const int blocks = ag_blocks(ncells_ub, AG_WAVES_PER_BLOCK);
  AG_KERNEL("ag_knn_cell_kernel", ag_knn_cell_kernel<CAP>, dim3(blocks), dim3(64 * AG_WAVES_PER_BLOCK), stream, a);
}

static int ag_knn_run(const float* d_src, int64_t ns, const float* d_tgt, int64_t nt, int k, bool self_query,
                      int32_t* d_out_src, double* d_out_dist, double* d_out_max, int64_t* d_stats, void* d_ws,
                      size_t ws_bytes, hipStream_t stream) {
  AgCarver sizing(nullptr);
  ag_knn_prepare(sizing, d_src, ns, d_tgt, nt, k, self_query, stream, nullptr);
  if (ws_bytes < sizing.total()) return ag_fail(AG_ERR_WORKSPACE, "workspace too small: %zu < %zu", ws_bytes, sizing.total());
  AgCarver c(d_ws);
  AgKnnWs w;
  int rc = ag_knn_prepare(c, d_src, ns, d_tgt, nt, k, self_query, stream, &w);
  if (rc) return rc;
  AG_KERNEL("ag_knn_init_kernel", ag_knn_init_kernel, dim3(1), dim3(64), stream, w.queue_count, w.max_bits, reinterpret_cast<long long*>(d_stats));
  AG_LAUNCH_CHECK();

  AgKnnArgs a;
  a.grid = w.src.grid;
  a.spts = w.src.pts;
  a.sstart = w.src.cell_start;
  a.tkeys = w.tgt.keys;
  a.tstart = w.tgt.cell_start;
  a.tgt = reinterpret_cast<const float2*>(d_tgt);
  a.src = reinterpret_cast<const float2*>(d_src);
  a.ns = ns;
  a.nt = nt;
  a.k = k;
  a.rg = 0.98 * w.src.grid.dlat;
  {
    const double s = sin(0.5 * a.rg);
    a.rg_chord2 = static_cast<float>(4.0 * s * s * (1.0 - 1.0e-4));
  }
  a.out_src = d_out_src;
  a.out_dist = d_out_dist;
  a.queue = w.queue;
  a.queue_count = w.queue_count;
  a.self_query = self_query ? 1 : 0;
  a.max_bits = w.max_bits;
  a.stats = reinterpret_cast<long long*>(d_stats);

  const int64_t ncu = w.src.ncells_ub;
  if (k + 1 <= 2) ag_launch_cell_kernel<2>(a, ncu, stream);
  else if (k + 1 <= 4) ag_launch_cell_kernel<4>(a, ncu, stream);
  else if (k + 1 <= 8) ag_launch_cell_kernel<8>(a, ncu, stream);
  else if (k + 1 <= 16) ag_launch_cell_kernel<16>(a, ncu, stream);
  else if (k + 1 <= 32) ag_launch_cell_kernel<32>(a, ncu, stream);
  else ag_launch_cell_kernel<AG_KNN_MAX_K + 1>(a, ncu, stream);
  AG_LAUNCH_CHECK();
  AG_KERNEL("ag_knn_exact_kernel", ag_knn_exact_kernel, dim3(2048), dim3(256), stream, a);
  AG_LAUNCH_CHECK();
  if (self_query) {
    AG_KERNEL("ag_maxdist_finish_kernel", ag_maxdist_finish_kernel, dim3(1), dim3(1), stream, w.max_bits, d_out_max);
    AG_LAUNCH_CHECK();
  }
  return AG_OK;
}

extern "C" size_t ag_knn_workspace_bytes(int64_t ns, int64_t nt, int k) {
  if (ns < 1 || nt < 0 || k < 1) return 0;
  AgCarver c(nullptr);
  ag_knn_prepare(c, nullptr, ns, nullptr, nt > 0 ? nt : 1, k, false, nullptr, nullptr);
  return c.total();
}

extern "C" int ag_knn(const float* d_src, int64_t ns, const float* d_tgt, int64_t nt, int k, int32_t* d_out_src,
                      double* d_out_dist, int64_t* d_stats, void* d_ws, size_t ws_bytes, void* stream) {
  AG_REQUIRE(k >= 1 && k <= AG_KNN_MAX_K, "ag_knn: k=%d outside [1, %d]", k, AG_KNN_MAX_K);
  AG_REQUIRE(ns >= k, "ag_knn: k=%d exceeds the number of sources %lld", k, static_cast<long long>(ns));
  AG_REQUIRE(ns < (int64_t(1) << 31) && nt < (int64_t(1) << 31) && nt >= 0, "ag_knn: node counts must fit int32");
  AG_REQUIRE(nt * k < (int64_t(1) << 40), "ag_knn: output too large");
  if (nt == 0) return AG_OK;
  AG_REQUIRE(d_src && d_tgt && d_out_src && d_ws, "ag_knn: null pointer");
  return ag_knn_run(d_src, ns, d_tgt, nt, k, false, d_out_src, d_out_dist, nullptr, d_stats, d_ws, ws_bytes,
                    static_cast<hipStream_t>(stream));
}

extern "C" size_t ag_nn_maxdist_workspace_bytes(int64_t n) {
  if (n < 2) return 0;
  AgCarver c(nullptr);
  ag_knn_prepare(c, nullptr, n, nullptr, n, 2, true, nullptr, nullptr);
  return c.total();
}

extern "C" int ag_nn_maxdist(const float* d_pts, int64_t n, double* d_out_max, void* d_ws, size_t ws_bytes,
                             void* stream) {
  AG_REQUIRE(n >= 2, "ag_nn_maxdist: needs at least 2 points (n=%lld)", static_cast<long long>(n));
  AG_REQUIRE(n < (int64_t(1) << 31), "ag_nn_maxdist: node count must fit int32");
  AG_REQUIRE(d_pts && d_out_max && d_ws, "ag_nn_maxdist: null pointer");
  return ag_knn_run(d_pts, n, d_pts, n, 2, true, nullptr, nullptr, d_out_max, nullptr, d_ws, ws_bytes,
                    static_cast<hipStream_t>(stream));
}

// ---------------------------------------------------------------------------------------------
// radius query
// ---------------------------------------------------------------------------------------------
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
};

template <bool FILL>
__global__ __launch_bounds__(256) void ag_radius_kernel(AgRadiusArgs a) {
  const int lane = ag_lane();
  const int64_t wave_global = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int64_t nwaves = (int64_t(gridDim.x) * blockDim.x) >> 6;
  for (int64_t t = wave_global; t < a.nt; t += nwaves) {
    const float2 tpf = a.tgt[t];
    const double tlat = tpf.x, tlon = tpf.y;
    const double tlonw = ag_wrap_lon(tlon);
    const float3 tu = ag_unit_vec(tpf.x, tpf.y);
    int64_t cnt = 0;
    const int64_t base_out = FILL ? a.offsets[t] : 0;
    ag_for_each_range(a.grid, a.sstart, tlat, tlat, tlonw, tlonw, a.radius, [&](int first, int last) {
      for (int p0 = first; p0 < last; p0 += 64) {
        const int p = p0 + lane;
        bool in = false;
        int id = 0;
        if (p < last) {
          const float4 s = a.spts[p];
          id = __float_as_int(s.w);
          const float dx = s.x - tu.x, dy = s.y - tu.y, dz = s.z - tu.z;
          const float d = dx * dx + dy * dy + dz * dz;
          in = d <= a.c_out;
          if (in && d > a.c_in) {
            const float2 sp = a.src[id];
            in = ag_rdist(tlat, tlon, sp.x, sp.y) <= a.rthr;
          }
        }
        const unsigned long long m = __ballot(in);
        if (FILL && in) {
          const int64_t o = base_out + cnt + __popcll(m & ((1ull << lane) - 1ull));
          a.out_src[o] = id;
          if (a.out_tgt) a.out_tgt[o] = static_cast<int32_t>(t);
        }
        cnt += __popcll(m);
      }
    });
    if (!FILL && lane == 0) a.offsets[t] = cnt;
  }
}

struct AgRadiusWs {
  AgBinned src;
  void* scan_temp;
  size_t scan_temp_bytes;
};

// cell size for a radius query: the query radius itself, but never so small that the grid has
// many more cells than there are sources
static double ag_radius_delta(int64_t ns, double radius) {
  const double spacing = sqrt(4.0 * AG_PI / static_cast<double>(ns));
  double d = radius > spacing ? radius : spacing;
  if (d > AG_PI) d = AG_PI;
  return d;
}

static int ag_radius_prepare(AgCarver& c, const float* d_src, int64_t ns, int64_t nt, double radius, bool build,
                             hipStream_t stream, AgRadiusWs* w) {
  // The carve must not depend on the radius (the workspace is sized before the radius is
  // known): the cell arrays are reserved for the finest grid any radius can ask for.
  const int nb_reserve = ag_grid_bands(sqrt(4.0 * AG_PI / static_cast<double>(ns)));
  const int nb = ag_grid_bands(ag_radius_delta(ns, radius));
  if (w == nullptr) {
    ag_bin_points(c, nullptr, ns, nb_reserve, nb_reserve, true, stream, nullptr);
  } else if (build) {
    int rc = ag_bin_points(c, d_src, ns, nb, nb_reserve, true, stream, &w->src);
    if (rc) return rc;
  } else {
    // descriptor of the grid the count pass built in this workspace (same carve, no kernels)
    int* ncols = c.take<int>(nb_reserve);
    int* off = c.take<int>(nb_reserve + 1);
    int* cell_start = c.take<int>(ag_grid_cells_ub(nb_reserve) + 1);
    (void)c.take<uint64_t>(ns);
    uint64_t* keys_b = c.take<uint64_t>(ns);
    (void)c.take<char>(ag_sort_u64_temp_bytes(ns));
    float4* pts = c.take<float4>(ns);
    w->src.grid.nb = nb;
    w->src.grid.dlat = AG_PI / nb;
    w->src.grid.inv_dlat = nb / AG_PI;
    w->src.grid.ncols = ncols;
    w->src.grid.off = off;
    w->src.n = ns;
    w->src.ncells_ub = ag_grid_cells_ub(nb);
    w->src.pts = pts;
    w->src.keys = keys_b;
    w->src.cell_start = cell_start;
  }
  const size_t stb = ag_scan_i64_temp_bytes(nt);
  void* st = c.take<char>(stb);
  if (w) {
    w->scan_temp = st;
    w->scan_temp_bytes = stb;
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
  a->radius = radius < AG_PI ? radius : AG_PI;
  const double s = sin(0.5 * radius);
  a->rthr = s * s;
  const double c2 = 4.0 * s * s;
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

extern "C" int ag_radius_count(const float* d_src, int64_t ns, const float* d_tgt, int64_t nt, double radius,
                               int64_t* d_offsets, void* d_ws, size_t ws_bytes, void* stream) {
  AG_REQUIRE(ns >= 1 && nt >= 0, "ag_radius_count: bad sizes");
  AG_REQUIRE(ns < (int64_t(1) << 31) && nt < (int64_t(1) << 31), "ag_radius_count: node counts must fit int32");
  AG_REQUIRE(radius >= 0.0, "ag_radius_count: negative radius");
  AG_REQUIRE(d_src && d_offsets && d_ws, "ag_radius_count: null pointer");
  if (ws_bytes < ag_radius_workspace_bytes(ns, nt)) return ag_fail(AG_ERR_WORKSPACE, "ag_radius_count: workspace too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  AgCarver c(d_ws);
  AgRadiusWs w;
  int rc = ag_radius_prepare(c, d_src, ns, nt > 0 ? nt : 1, radius, true, s, &w);
  if (rc) return rc;
  AgRadiusArgs a;
  ag_radius_args(d_src, d_tgt, nt, radius, w, &a);
  a.offsets = d_offsets;
  a.out_src = nullptr;
  a.out_tgt = nullptr;
  if (nt > 0) {
    AG_KERNEL("ag_radius_kernel", ag_radius_kernel<false>, dim3(ag_blocks(nt, 4, 8192)), dim3(256), s, a);
    AG_LAUNCH_CHECK();
  }
  return ag_exclusive_scan_i64(w.scan_temp, w.scan_temp_bytes, d_offsets, d_offsets, nt, s);
}

extern "C" int ag_radius_fill(const float* d_src, int64_t ns, const float* d_tgt, int64_t nt, double radius,
                              const int64_t* d_offsets, int32_t* d_out_src, int32_t* d_out_tgt, void* d_ws,
                              size_t ws_bytes, void* stream) {
  AG_REQUIRE(ns >= 1 && nt >= 0, "ag_radius_fill: bad sizes");
  if (nt == 0) return AG_OK;
  AG_REQUIRE(d_src && d_tgt && d_offsets && d_out_src && d_ws, "ag_radius_fill: null pointer");
  if (ws_bytes < ag_radius_workspace_bytes(ns, nt)) return ag_fail(AG_ERR_WORKSPACE, "ag_radius_fill: workspace too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  AgCarver c(d_ws);
  AgRadiusWs w;
  int rc = ag_radius_prepare(c, d_src, ns, nt, radius, false, s, &w);
  if (rc) return rc;
  AgRadiusArgs a;
  ag_radius_args(d_src, d_tgt, nt, radius, w, &a);
  a.offsets = const_cast<int64_t*>(d_offsets);
  a.out_src = d_out_src;
  a.out_tgt = d_out_tgt;
  AG_KERNEL("ag_radius_kernel", ag_radius_kernel<true>, dim3(ag_blocks(nt, 4, 8192)), dim3(256), s, a);
  AG_LAUNCH_CHECK();
  return AG_OK;
}
