// Icosphere subdivision (TriNodes) and multi-scale mesh edges (MultiScaleEdges) for gfx950.
//
// The mesh follows trimesh.creation.icosphere: every subdivision appends one midpoint per
// unique undirected edge after the existing vertices -- unique edges enumerated in increasing
// order of (max vertex, min vertex) -- splits each face (a,b,c) into (a,ab,ca) (ab,b,bc)
// (ca,bc,c) (ab,bc,ca), and then pushes ALL vertices back to the unit sphere in float64.
// On the GPU one level needs no sort.  In the closed, consistently oriented mesh every edge
// appears once as a half-edge (a, b) with a > b; those half-edges are collected per larger vertex
// a (at most 6 per vertex, one atomic slot counter each), a prefix sum over the counters gives
// every vertex the position of its first edge, and sorting the <= 6 smaller neighbours of a
// vertex in registers yields exactly the (max vertex, min vertex) order.  One thread per vertex
// writes its unique edges and their midpoints; one thread per face finds its three midpoints by
// looking the smaller endpoint up among the <= 6 neighbours of the larger one and writes the
// four child faces.
#include <string.h>

#include "ag_common.hpp"

#define AG_MAX_DEG 6

__constant__ double ag_ico_vertices[12][3] = {
    {-1, 1.6180339887498949, 0}, {1, 1.6180339887498949, 0},  {-1, -1.6180339887498949, 0},
    {1, -1.6180339887498949, 0}, {0, -1, 1.6180339887498949}, {0, 1, 1.6180339887498949},
    {0, -1, -1.6180339887498949}, {0, 1, -1.6180339887498949}, {1.6180339887498949, 0, -1},
    {1.6180339887498949, 0, 1},  {-1.6180339887498949, 0, -1}, {-1.6180339887498949, 0, 1}};

__constant__ int ag_ico_faces[20][3] = {{0, 11, 5}, {0, 5, 1},  {0, 1, 7},   {0, 7, 10}, {0, 10, 11},
                                        {1, 5, 9},  {5, 11, 4}, {11, 10, 2}, {10, 7, 6}, {7, 1, 8},
                                        {3, 9, 4},  {3, 4, 2},  {3, 2, 6},   {3, 6, 8},  {3, 8, 9},
                                        {4, 9, 5},  {2, 4, 11}, {6, 2, 10},  {8, 6, 7},  {9, 8, 1}};

static inline int64_t ag_ico_V(int l) { return 10 * (int64_t(1) << (2 * l)) + 2; }
static inline int64_t ag_ico_F(int l) { return 20 * (int64_t(1) << (2 * l)); }
static inline int64_t ag_ico_E(int l) { return 30 * (int64_t(1) << (2 * l)); }
// offset of level l in the concatenated per-level unique-edge array: sum of E_j for j < l
static inline int64_t ag_ico_edge_offset(int l) { return 10 * ((int64_t(1) << (2 * l)) - 1); }
// bits needed for a vertex id of level l (1 << bits > V_l)
static inline int ag_vertex_bits(int l) {
  int bits = 1;
  while ((int64_t(1) << bits) <= ag_ico_V(l)) ++bits;
  return bits;
}

__device__ __forceinline__ uint64_t ag_edge_key(int a, int b) {
  const uint32_t lo = static_cast<uint32_t>(min(a, b)), hi = static_cast<uint32_t>(max(a, b));
  return (static_cast<uint64_t>(hi) << 32) | lo;
}

// half-edges (a, b) with a > b: slot of b among the smaller neighbours of a
// (the *_range functions take the first item and the stride of the calling thread, so that the same
// code serves a grid-wide launch and the single-block kernel of the small levels; cnt / first / nb
// may point to global memory or to LDS)
__device__ __forceinline__ void ag_ico_lower_range(int64_t i0, int64_t step, const int32_t* faces, int64_t F, int* cnt,
                                                   int32_t* nb, int* error_flag) {
  for (int64_t i = i0; i < 3 * F; i += step) {
    const int64_t f = i / 3;
    const int j = static_cast<int>(i - 3 * f);
    const int a = faces[3 * f + j], b = faces[3 * f + (j == 2 ? 0 : j + 1)];
    if (a > b) {
      const int slot = atomicAdd(&cnt[a], 1);
      if (slot < AG_MAX_DEG) nb[AG_MAX_DEG * int64_t(a) + slot] = b;
      else *error_flag = 1;
    }
  }
}

__device__ __forceinline__ void ag_cswap(int& x, int& y) {
  const int lo = min(x, y), hi = max(x, y);
  x = lo;
  y = hi;
}

// One thread per vertex a: sort its smaller neighbours, write them back, emit the unique edges
// (a, b_j) at first[a] + j and (optionally) their midpoints as vertices V + first[a] + j.
__device__ __forceinline__ void ag_ico_emit_range(int64_t i0, int64_t step, int64_t V, int64_t Eu, const int* first,
                                                  int32_t* nb, uint64_t* uniq, double* xyz, int write_midpoints,
                                                  int* error_flag) {
  for (int64_t a = i0; a < V; a += step) {
    if (a == 0 && first[V] != Eu) *error_flag = 1;  // not a closed, consistently oriented mesh
    const int base = first[a];
    const int n = min(first[a + 1] - base, AG_MAX_DEG);
    int32_t* mine = nb + AG_MAX_DEG * a;
    int v0 = n > 0 ? mine[0] : 0x7fffffff, v1 = n > 1 ? mine[1] : 0x7fffffff, v2 = n > 2 ? mine[2] : 0x7fffffff;
    int v3 = n > 3 ? mine[3] : 0x7fffffff, v4 = n > 4 ? mine[4] : 0x7fffffff, v5 = n > 5 ? mine[5] : 0x7fffffff;
    // 12-comparator sorting network for 6 values
    ag_cswap(v0, v5); ag_cswap(v1, v3); ag_cswap(v2, v4);
    ag_cswap(v1, v2); ag_cswap(v3, v4);
    ag_cswap(v0, v3); ag_cswap(v2, v5);
    ag_cswap(v0, v1); ag_cswap(v2, v3); ag_cswap(v4, v5);
    ag_cswap(v1, v2); ag_cswap(v3, v4);
    const int v[AG_MAX_DEG] = {v0, v1, v2, v3, v4, v5};
    double ax = 0.0, ay = 0.0, az = 0.0;
    if (write_midpoints) {
      ax = xyz[3 * a];
      ay = xyz[3 * a + 1];
      az = xyz[3 * a + 2];
    }
#pragma unroll
    for (int j = 0; j < AG_MAX_DEG; ++j) {
      if (j < n) {
        const int b = v[j];
        mine[j] = b;
        const int64_t i = base + j;
        uniq[i] = (static_cast<uint64_t>(static_cast<uint32_t>(a)) << 32) | static_cast<uint32_t>(b);
        if (write_midpoints) {
          xyz[3 * (V + i)] = (ax + xyz[3 * int64_t(b)]) / 2.0;
          xyz[3 * (V + i) + 1] = (ay + xyz[3 * int64_t(b) + 1]) / 2.0;
          xyz[3 * (V + i) + 2] = (az + xyz[3 * int64_t(b) + 2]) / 2.0;
        }
      }
    }
  }
}

// index of the unique edge {x, y}: position of the smaller endpoint among the sorted smaller
// neighbours of the larger one
__device__ __forceinline__ int ag_edge_index(const int* first, const int32_t* nb, int x, int y) {
  const int hi = max(x, y), lo = min(x, y);
  const int32_t* mine = nb + AG_MAX_DEG * int64_t(hi);
  const int n = first[hi + 1] - first[hi];
  int pos = 0;
#pragma unroll
  for (int j = 0; j < AG_MAX_DEG; ++j) pos += (j < n && mine[j] < lo) ? 1 : 0;
  return first[hi] + pos;
}

__device__ __forceinline__ void ag_ico_subdivide_range(int64_t i0, int64_t step, const int32_t* fin, int64_t F,
                                                       const int* first, const int32_t* nb, int V, int32_t* fout) {
  for (int64_t f = i0; f < F; f += step) {
    const int a = fin[3 * f], b = fin[3 * f + 1], c = fin[3 * f + 2];
    const int m0 = V + ag_edge_index(first, nb, a, b);
    const int m1 = V + ag_edge_index(first, nb, b, c);
    const int m2 = V + ag_edge_index(first, nb, c, a);
    int32_t* o = fout + 12 * f;
    o[0] = a;  o[1] = m0; o[2] = m2;
    o[3] = m0; o[4] = b;  o[5] = m1;
    o[6] = m2; o[7] = m1; o[8] = c;
    o[9] = m0; o[10] = m1; o[11] = m2;
  }
}

// unique edges of level R from the last subdivision, no sort: every level-(R-1) edge (a,b) with
// midpoint m gives (a,m) and (b,m); the central child (m0,m1,m2) of every parent face gives its 3 sides
__global__ void ag_ico_final_edges_kernel(const uint64_t* __restrict__ parent_edges, int64_t Eu, int64_t V,
                                          const int32_t* __restrict__ child_faces, int64_t F_parent,
                                          uint64_t* __restrict__ out) {
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < Eu + F_parent; i += int64_t(gridDim.x) * blockDim.x) {
    if (i < Eu) {
      const uint64_t k = parent_edges[i];
      const int a = static_cast<int>(static_cast<uint32_t>(k)), b = static_cast<int>(static_cast<uint32_t>(k >> 32));
      const int m = static_cast<int>(V + i);
      out[2 * i] = ag_edge_key(a, m);
      out[2 * i + 1] = ag_edge_key(b, m);
    } else {
      const int64_t f = i - Eu;
      const int m0 = child_faces[12 * f + 9], m1 = child_faces[12 * f + 10], m2 = child_faces[12 * f + 11];
      uint64_t* o = out + 2 * Eu + 3 * f;
      o[0] = ag_edge_key(m0, m1);
      o[1] = ag_edge_key(m1, m2);
      o[2] = ag_edge_key(m2, m0);
    }
  }
}

// (also clears the slot counters for the next level, V + 1 of them, when cnt is given)
__device__ __forceinline__ void ag_ico_project_range(int64_t i0, int64_t step, double* xyz, int64_t V, int* cnt) {
  for (int64_t i = i0; i < V; i += step) {
    if (cnt != nullptr) {
      cnt[i] = 0;
      if (i == 0) cnt[V] = 0;
    }
    const double x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    const double s = sqrt(x * x + y * y + z * z);
    const double off = 1.0 - s;
    xyz[3 * i] = x + (x / s) * off;
    xyz[3 * i + 1] = y + (y / s) * off;
    xyz[3 * i + 2] = z + (z / s) * off;
  }
}


#define AG_GRID_I0 (blockIdx.x * int64_t(blockDim.x) + threadIdx.x)
#define AG_GRID_STEP (int64_t(gridDim.x) * blockDim.x)

__global__ void ag_ico_lower_kernel(const int32_t* __restrict__ faces, int64_t F, int* __restrict__ cnt,
                                    int32_t* __restrict__ nb, int* __restrict__ error_flag) {
  ag_ico_lower_range(AG_GRID_I0, AG_GRID_STEP, faces, F, cnt, nb, error_flag);
}
__global__ void ag_ico_emit_kernel(int64_t V, int64_t Eu, const int* __restrict__ first, int32_t* __restrict__ nb,
                                   uint64_t* __restrict__ uniq, double* __restrict__ xyz, int write_midpoints,
                                   int* __restrict__ error_flag) {
  ag_ico_emit_range(AG_GRID_I0, AG_GRID_STEP, V, Eu, first, nb, uniq, xyz, write_midpoints, error_flag);
}
// Everything that follows the unique edges of a level, in one launch (the parts are independent of
// each other): the four children of every face; all V + Eu vertices pushed to the unit sphere; the
// slot counters (in the OTHER buffer than the one this level's offsets live in, which the face part
// is still reading) and the scan state cleared for the next level; and, for the last level, its
// unique edges with no further sort -- every parent edge (a, b) with midpoint m gives (a, m) and
// (b, m), every parent face the three sides of its central child.
__global__ void ag_ico_refine_kernel(const int32_t* __restrict__ fin, int64_t F, const int* __restrict__ first,
                                     const int32_t* __restrict__ nb, int64_t V, int64_t Eu, int32_t* __restrict__ fout,
                                     double* __restrict__ xyz, int* __restrict__ next_cnt,
                                     unsigned long long* __restrict__ scan_state, int64_t scan_words,
                                     const uint64_t* __restrict__ parent_edges, uint64_t* __restrict__ final_edges) {
  const int64_t i0 = AG_GRID_I0, step = AG_GRID_STEP;
  for (int64_t f = i0; f < F; f += step) {
    const int a = fin[3 * f], b = fin[3 * f + 1], c = fin[3 * f + 2];
    const int m0 = static_cast<int>(V) + ag_edge_index(first, nb, a, b);
    const int m1 = static_cast<int>(V) + ag_edge_index(first, nb, b, c);
    const int m2 = static_cast<int>(V) + ag_edge_index(first, nb, c, a);
    int32_t* o = fout + 12 * f;
    o[0] = a;  o[1] = m0; o[2] = m2;
    o[3] = m0; o[4] = b;  o[5] = m1;
    o[6] = m2; o[7] = m1; o[8] = c;
    o[9] = m0; o[10] = m1; o[11] = m2;
    if (final_edges != nullptr) {
      uint64_t* e = final_edges + 2 * Eu + 3 * f;
      e[0] = ag_edge_key(m0, m1);
      e[1] = ag_edge_key(m1, m2);
      e[2] = ag_edge_key(m2, m0);
    }
  }
  if (final_edges != nullptr) {
    for (int64_t i = i0; i < Eu; i += step) {
      const uint64_t k = parent_edges[i];
      const int a = static_cast<int>(static_cast<uint32_t>(k)), b = static_cast<int>(static_cast<uint32_t>(k >> 32));
      const int m = static_cast<int>(V + i);
      final_edges[2 * i] = ag_edge_key(a, m);
      final_edges[2 * i + 1] = ag_edge_key(b, m);
    }
  }
  ag_ico_project_range(i0, step, xyz, V + Eu, next_cnt);
  for (int64_t i = i0; i < scan_words; i += step) scan_state[i] = 0ull;
}

// The first levels are too small to fill even one CU: a launch per step costs more than the step.
// One block of 1024 threads takes the mesh from the icosahedron through `levels` subdivisions
// (at most AG_ICO_SMALL_LEVELS), with the slot counters, the vertex offsets and the neighbour lists
// in LDS and workgroup barriers between the steps.  Vertices, faces and edges go to the same
// global arrays the grid-wide kernels use, so the later levels continue from them.
#define AG_ICO_SMALL_LEVELS 4
#define AG_ICO_SMALL_V 642  // vertices of level AG_ICO_SMALL_LEVELS - 1, the last one subdivided here

__global__ __launch_bounds__(1024) void ag_ico_small_levels_kernel(int levels, double* xyz, int32_t* faces_a,
                                                                   int32_t* faces_b, uint64_t* level_edges,
                                                                   uint64_t* uniq_scratch, int* g_first,
                                                                   unsigned long long* scan_state, int scan_words,
                                                                   int* error_flag, int* zero_ptr, int zero_n) {
  __shared__ int s_first[AG_ICO_SMALL_V + 2];
  __shared__ int32_t s_nb[AG_MAX_DEG * AG_ICO_SMALL_V];
  __shared__ int s_scan[32 + 1024];
  const int tid = threadIdx.x;
  // counters that kernels further down the stream expect to be zero (the error flag of this build, the bucket
  // counters of the node-order sort)
  if (tid == 0) *error_flag = 0;
  for (int i = tid; i < zero_n; i += 1024) zero_ptr[i] = 0;
  {  // icosahedron (trimesh.creation.icosahedron)
    if (tid < 12) {
      const double t = (1.0 + sqrt(5.0)) / 2.0;
      const double n = sqrt(2.0 + t);
      for (int c = 0; c < 3; ++c) {
        const double raw = ag_ico_vertices[tid][c];
        const double v = (fabs(raw) > 1.5) ? (raw > 0 ? t : -t) : raw;
        xyz[3 * tid + c] = v / n;
      }
    }
    if (tid < 20)
      for (int c = 0; c < 3; ++c) faces_a[3 * tid + c] = ag_ico_faces[tid][c];
  }
  __syncthreads();
  int32_t* cur = faces_a;
  int32_t* nxt = faces_b;
  for (int l = 0; l < levels; ++l) {
    const int64_t F = 20ll << (2 * l), Eu = 30ll << (2 * l), V = (10ll << (2 * l)) + 2;
    for (int i = tid; i <= V; i += 1024) s_first[i] = 0;
    __syncthreads();
    ag_ico_lower_range(tid, 1024, cur, F, s_first, s_nb, error_flag);
    __syncthreads();
    {  // exclusive prefix sum of s_first[0 .. V]
      const int n = static_cast<int>(V) + 1;
      const int per = (n + 1023) / 1024;
      const int b = tid * per, e = min(n, b + per);
      int local = 0;
      for (int i = b; i < e; ++i) local += s_first[i];
      // inclusive scan of the 1024 per-thread sums: shuffles inside each wave, then the 16 wave totals
      int incl = local;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_up(incl, d, 64);
        if ((tid & 63) >= d) incl += v;
      }
      if ((tid & 63) == 63) s_scan[tid >> 6] = incl;
      __syncthreads();
      if (tid < 64) {
        int w = tid < 16 ? s_scan[tid] : 0;
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
          const int v = __shfl_up(w, d, 64);
          if (tid >= d) w += v;
        }
        if (tid < 16) s_scan[16 + tid] = w;  // inclusive totals of waves 0 .. tid
      }
      __syncthreads();
      if (tid >= 64) incl += s_scan[16 + (tid >> 6) - 1];
      s_scan[32 + tid] = incl;
      int run = incl - local;
      for (int i = b; i < e; ++i) {
        const int c = s_first[i];
        s_first[i] = run;
        run += c;
      }
    }
    __syncthreads();
    uint64_t* uniq = level_edges ? level_edges + 10 * ((int64_t(1) << (2 * l)) - 1) : uniq_scratch;
    ag_ico_emit_range(tid, 1024, V, Eu, s_first, s_nb, uniq, xyz, 1, error_flag);
    __syncthreads();
    ag_ico_subdivide_range(tid, 1024, cur, F, s_first, s_nb, static_cast<int>(V), nxt);
    __syncthreads();
    ag_ico_project_range(tid, 1024, xyz, V + Eu, nullptr);
    __syncthreads();
    int32_t* t = cur;
    cur = nxt;
    nxt = t;
  }
  // slot counters and scan state of the next (grid-wide) level
  const int64_t Vn = (10ll << (2 * levels)) + 2;
  for (int64_t i = tid; i <= Vn; i += 1024) g_first[i] = 0;
  for (int i = tid; i < scan_words; i += 1024) scan_state[i] = 0ull;
}

// descending order of a float as ascending order of an unsigned key
__device__ __forceinline__ uint32_t ag_desc_key(float v) {
  v = v + 0.0f;  // -0.0 -> +0.0
  uint32_t u = __float_as_uint(v);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // ascending
  return ~u;                                        // descending
}

// Node order (generate/utils.py:15-33): latitude descending, ties longitude descending, ties node id
// descending -- one stable sort of key = (descending-latitude key << 32) | descending-longitude key,
// fed in descending id order.

// sort input: position p holds node id = n - 1 - p with its key; for the icosphere the coordinates
// themselves are produced here too (generate/transforms.py:50-52)
//
// Node sets of up to AG_ORDER_SMALL_N nodes are not handed to a library sort (a merge sort of nine launches at
// these sizes): the nodes are dealt into latitude buckets of equal height -- a counting sort, one returning
// atomic per node -- and the nodes of every bucket are ranked among each other, 64 per wave.  Three
// launches; the bucket is a monotone function of the latitude, so the buckets are in order among themselves,
// and ties of a whole key go to the higher id, which makes the result the stable sort's.
#define AG_ORDER_SMALL_N (1 << 19)
#define AG_ORDER_MAX_BUCKETS 8192
#define AG_ORDER_WAVE_CAP 512  // nodes of a bucket a wave ranks from LDS (longer buckets: from global memory)
#define AG_ORDER_CHUNKS 8    // waves per bucket, 64 nodes each (the waves of a longer bucket take several)

static int ag_order_buckets(int64_t n) {
  int64_t b = n / 32;
  if (b < 64) b = 64;
  if (b > AG_ORDER_MAX_BUCKETS) b = AG_ORDER_MAX_BUCKETS;
  return static_cast<int>(b);
}

__device__ __forceinline__ int ag_order_bucket_of(float lat, int B) {
  // (NaN and latitudes beyond the poles land in the outermost buckets; the key order inside a bucket is exact)
  const float t = (static_cast<float>(AG_HALF_PI) - lat) * (static_cast<float>(B) * static_cast<float>(1.0 / AG_PI));
  int b = t > 0.0f ? static_cast<int>(t) : 0;
  return b < B ? b : B - 1;
}

// (lat, lon) of a vertex of the unit icosphere: cartesian_to_latlon_rad (generate/transforms.py:50-52) in float64,
// rounded to float32 as the node builder stores it
__device__ __forceinline__ float2 ag_vertex_latlon(const double* __restrict__ xyz, int64_t i) {
  const double x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
  const double lat = asin(z / (x * x + y * y + z * z));
  const double lon = atan2(y, x);
  return make_float2(static_cast<float>(lat), static_cast<float>(lon));
}

// the coordinates in vertex order, for the callers that order a part of the vertices themselves
__global__ void ag_vertex_latlon_kernel(const double* __restrict__ xyz, float2* __restrict__ latlon, int64_t n) {
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x)
    latlon[i] = ag_vertex_latlon(xyz, i);
}

__global__ void ag_order_keys_kernel(const double* __restrict__ xyz, float2* __restrict__ latlon, int64_t n,
                                     uint64_t* __restrict__ keys, int32_t* __restrict__ ids, int B,
                                     int* __restrict__ bucket_count, int* __restrict__ bucket_of,
                                     int* __restrict__ slot_of) {
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
    float2 p;
    if (xyz != nullptr) {
      p = ag_vertex_latlon(xyz, i);
      latlon[i] = p;
    } else {
      p = latlon[i];
    }
    const uint64_t key = (static_cast<uint64_t>(ag_desc_key(p.x)) << 32) | ag_desc_key(p.y);
    if (bucket_count != nullptr) {  // bucket path: keys stay in id order, every node takes a slot of its bucket
      keys[i] = key;
      // the bucket must not separate equal keys: it is taken from the latitude the key encodes (-0 as +0)
      const int b = ag_order_bucket_of(p.x + 0.0f, B);
      bucket_of[i] = b;
      slot_of[i] = atomicAdd(&bucket_count[b], 1);
    } else {
      const int64_t pos = n - 1 - i;
      keys[pos] = key;
      ids[pos] = static_cast<int32_t>(i);
    }
  }
}

// the inverse permutation and the ordered coordinates
__global__ void ag_order_finish_kernel(const int32_t* __restrict__ order, const float2* __restrict__ latlon, int64_t n,
                                       float2* __restrict__ out_latlon, int32_t* __restrict__ rank) {
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
    const int32_t id = order[i];
    rank[id] = static_cast<int32_t>(i);
    out_latlon[i] = latlon[id];
  }
}

// bucket path, second launch: every block turns the bucket counts into bucket starts in LDS (the table is small),
// then deals its share of the nodes into their buckets; block 0 leaves the starts in global memory for the third.
__global__ __launch_bounds__(256) void ag_order_scatter_kernel(int64_t n, int B, const int* __restrict__ bucket_count,
                                                               const int* __restrict__ bucket_of,
                                                               const int* __restrict__ slot_of,
                                                               const uint64_t* __restrict__ keys,
                                                               uint64_t* __restrict__ out_keys,
                                                               int32_t* __restrict__ out_ids,
                                                               int* __restrict__ bucket_start) {
  __shared__ int start[AG_ORDER_MAX_BUCKETS + 1];
  __shared__ int partial[256];
  const int tid = threadIdx.x;
  for (int b = tid; b < B; b += 256) start[b] = bucket_count[b];  // (coalesced; the serial part below reads LDS)
  __syncthreads();
  const int per = (B + 255) / 256;
  const int b0 = tid * per, b1 = min(B, b0 + per);
  int sum = 0;
  for (int b = b0; b < b1; ++b) sum += start[b];
  partial[tid] = sum;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {
    const int v = (tid >= d) ? partial[tid - d] : 0;
    __syncthreads();
    partial[tid] += v;
    __syncthreads();
  }
  int run = partial[tid] - sum;
  for (int b = b0; b < b1; ++b) {
    const int count = start[b];
    start[b] = run;
    run += count;
  }
  if (tid == 255) start[B] = partial[255];
  __syncthreads();
  if (blockIdx.x == 0)
    for (int b = tid; b <= B; b += 256) bucket_start[b] = start[b];
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + tid; i < n; i += int64_t(gridDim.x) * blockDim.x) {
    const int pos = start[bucket_of[i]] + slot_of[i];
    out_keys[pos] = keys[i];
    out_ids[pos] = static_cast<int32_t>(i);
  }
}

// bucket path, third launch: the nodes of every bucket are ranked among each other -- position = number of nodes
// of the bucket that come first: smaller key, or equal key and higher id -- and the order, the inverse permutation
// and the ordered coordinates are written.  A wave takes 64 nodes of a bucket (AG_ORDER_CHUNKS waves per bucket;
// those beyond the bucket's length leave at once) and stages the whole bucket in its slice of LDS (longer buckets
// are read from global memory); the counting loop is unrolled so that eight LDS reads are in flight.
__global__ __launch_bounds__(256) void ag_order_bucket_kernel(int B, const int* __restrict__ bucket_start,
                                                              const uint64_t* __restrict__ keys,
                                                              const int32_t* __restrict__ ids,
                                                              const float2* __restrict__ latlon,
                                                              int32_t* __restrict__ order, float2* __restrict__ out_latlon,
                                                              int32_t* __restrict__ rank) {
  __shared__ __attribute__((aligned(16))) uint64_t s_keys_all[4][AG_ORDER_WAVE_CAP];
  __shared__ __attribute__((aligned(16))) int32_t s_ids_all[4][AG_ORDER_WAVE_CAP];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int item = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);
  const int b = item / AG_ORDER_CHUNKS, chunk = item % AG_ORDER_CHUNKS;
  if (b >= B) return;
  const int first = bucket_start[b], m = bucket_start[b + 1] - first;  // (wave-uniform)
  if (chunk * 64 >= m) return;
  uint64_t* s_keys = s_keys_all[wave];
  int32_t* s_ids = s_ids_all[wave];
  const bool staged = m <= AG_ORDER_WAVE_CAP;
  const int m4 = (m + 3) & ~3;
  if (staged) {
    for (int j = lane; j < m4; j += 64) {
      // (padding to a multiple of four: the largest key with the smallest id comes before no node)
      s_keys[j] = j < m ? keys[first + j] : ~0ull;
      s_ids[j] = j < m ? ids[first + j] : static_cast<int32_t>(0x80000000u);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  for (int e = chunk * 64 + lane; e < m; e += 64 * AG_ORDER_CHUNKS) {
    const uint64_t ke = staged ? s_keys[e] : keys[first + e];
    const int32_t ie = staged ? s_ids[e] : ids[first + e];
    // no short-circuit: a branch per comparison would expose the latency of every LDS read
    auto before = [&](uint64_t kj, int32_t ij) {
      return static_cast<int>(kj < ke) | (static_cast<int>(kj == ke) & static_cast<int>(ij > ie));
    };
    int r = 0;
    if (staged) {
      const ulonglong2* k2 = reinterpret_cast<const ulonglong2*>(s_keys);
      const int4* i4 = reinterpret_cast<const int4*>(s_ids);
#pragma unroll 2
      for (int j = 0; j < m4 / 4; ++j) {
        const ulonglong2 ka = k2[2 * j], kb = k2[2 * j + 1];
        const int4 iv = i4[j];
        r += before(ka.x, iv.x) + before(ka.y, iv.y) + before(kb.x, iv.z) + before(kb.y, iv.w);
      }
    } else {
#pragma unroll 4
      for (int j = 0; j < m; ++j) r += before(keys[first + j], ids[first + j]);
    }
    const int pos = first + r;
    order[pos] = ie;
    if (rank != nullptr) {
      rank[ie] = pos;
      out_latlon[pos] = latlon[ie];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct AgTopoWs {
  int* first;          // [V_R + 2] slot counters, then first-edge offsets of every vertex
  int* first_alt;      // [V_R + 2] the counters of the following level (the two alternate)
  int32_t* nb;         // [6 V_R]   smaller neighbours of every vertex
  unsigned long long* scan_state;  // state words of the scan of the counters
  int64_t scan_words;
  uint64_t* uniq;      // [E_R]     unique edges of a level when the caller does not keep them
  uint64_t* keys_a;    // [V_R]     node-order sort keys
  uint64_t* keys_b;    // [V_R]
  int32_t* ids_a;      // [V_R]     node-order sort values
  int32_t* ids_b;      // [V_R]     bucket path of the node order: ids by bucket, bucket and slot of every node,
  int* bucket_of;      // [V_R]     bucket counters and starts
  int* slot_of;        // [V_R]
  int* bucket_count;   // [AG_ORDER_MAX_BUCKETS + 1]
  int* bucket_start;   // [AG_ORDER_MAX_BUCKETS + 1]
  void* sort_temp;
  size_t sort_temp_bytes;
  int32_t* faces_a;    // [3 F_R]
  int* error_flag;
};

static void ag_topo_carve(AgCarver& c, int R, AgTopoWs* w) {
  const int64_t F = ag_ico_F(R), V = ag_ico_V(R);
  AgTopoWs t;
  t.first = c.take<int>(V + 2);
  t.first_alt = c.take<int>(V + 2);
  t.nb = c.take<int32_t>(AG_MAX_DEG * V);
  t.scan_words = static_cast<int64_t>(ag_chained_scan_state_words_i32(V + 2));
  t.scan_state = c.take<unsigned long long>(t.scan_words);
  t.uniq = c.take<uint64_t>(ag_ico_E(R));
  t.keys_a = c.take<uint64_t>(V);
  t.keys_b = c.take<uint64_t>(V);
  t.ids_a = c.take<int32_t>(V);
  t.ids_b = c.take<int32_t>(V);
  t.bucket_of = c.take<int>(V);
  t.slot_of = c.take<int>(V);
  t.bucket_count = c.take<int>(AG_ORDER_MAX_BUCKETS + 1);
  t.bucket_start = c.take<int>(AG_ORDER_MAX_BUCKETS + 1);
  t.sort_temp_bytes = ag_sort_pairs_u64_i32_temp_bytes(V);
  t.sort_temp = c.take<char>(t.sort_temp_bytes);
  t.faces_a = c.take<int32_t>(3 * F);
  t.error_flag = c.take<int>(4);
  if (w) *w = t;
}

// unique edges of `faces` (level l) into `uniq`, in (max vertex, min vertex) order; leaves the
// per-vertex offsets (in `first`) and sorted neighbour lists in the workspace for the subdivision.  The
// slot counters first[0 .. V_l] and the scan state must be zero on entry.
static int ag_topo_unique_edges(const AgTopoWs& w, int* first, const int32_t* faces, int l, uint64_t* uniq, double* xyz,
                                bool write_midpoints, hipStream_t s) {
  const int64_t F = ag_ico_F(l), Eu = ag_ico_E(l), V = ag_ico_V(l);
  AG_KERNEL("ag_ico_lower_kernel", ag_ico_lower_kernel, dim3(ag_blocks(3 * F, 256, 8192)), dim3(256), s, faces, F,
            first, w.nb, w.error_flag);
  AG_LAUNCH_CHECK();
  int rc = ag_chained_scan_i32(first, first, V, w.scan_state, s);  // total -> first[V]
  if (rc) return rc;
  AG_KERNEL("ag_ico_emit_kernel", ag_ico_emit_kernel, dim3(ag_blocks(V, 256, 8192)), dim3(256), s, V, Eu, first, w.nb,
            uniq, xyz, write_midpoints ? 1 : 0, w.error_flag);
  AG_LAUNCH_CHECK();
  return AG_OK;
}

// node order of n nodes: `xyz` (nullable) are the unit vectors the coordinates are computed from first;
// d_order receives the node ids in order, out_latlon / rank (nullable) the ordered coordinates and the
// inverse permutation
struct AgOrderWs {
  uint64_t *keys_a, *keys_b;
  int32_t *ids_a, *ids_b;
  int *bucket_of, *slot_of, *bucket_count, *bucket_start;
  void* sort_temp;
  size_t sort_temp_bytes;
};

// `counters_clear`: the caller has a kernel upstream that zeroed w.bucket_count[0 .. AG_ORDER_MAX_BUCKETS]
static int ag_order_nodes(const double* xyz, float2* latlon, int64_t n, const AgOrderWs& w, int32_t* d_order,
                          float2* out_latlon, int32_t* rank, bool counters_clear, hipStream_t s) {
  const int nb = ag_blocks(n, 256, 8192);
  if (n <= AG_ORDER_SMALL_N) {
    const int B = ag_order_buckets(n);
    if (!counters_clear) {
      int rc = ag_zero_i32(w.bucket_count, B + 1, s);
      if (rc) return rc;
    }
    AG_KERNEL("ag_order_keys_kernel", ag_order_keys_kernel, dim3(nb), dim3(256), s, xyz, latlon, n, w.keys_a, w.ids_a, B,
              w.bucket_count, w.bucket_of, w.slot_of);
    AG_LAUNCH_CHECK();
    AG_KERNEL("ag_order_scatter_kernel", ag_order_scatter_kernel, dim3(ag_blocks(n, 1024, 1024)), dim3(256), s, n, B,
              w.bucket_count, w.bucket_of, w.slot_of, w.keys_a, w.keys_b, w.ids_b, w.bucket_start);
    AG_LAUNCH_CHECK();
    AG_KERNEL("ag_order_bucket_kernel", ag_order_bucket_kernel, dim3((B * AG_ORDER_CHUNKS + 3) / 4), dim3(256), s, B,
              w.bucket_start, w.keys_b, w.ids_b, latlon, d_order, out_latlon, rank);
    AG_LAUNCH_CHECK();
    return AG_OK;
  }
  AG_KERNEL("ag_order_keys_kernel", ag_order_keys_kernel, dim3(nb), dim3(256), s, xyz, latlon, n, w.keys_a, w.ids_a, 0,
            static_cast<int*>(nullptr), static_cast<int*>(nullptr), static_cast<int*>(nullptr));
  AG_LAUNCH_CHECK();
  int rc = ag_sort_pairs_u64_i32(w.sort_temp, w.sort_temp_bytes, w.keys_a, w.keys_b, w.ids_a, d_order, n, s);
  if (rc) return rc;
  if (rank != nullptr) {
    AG_KERNEL("ag_order_finish_kernel", ag_order_finish_kernel, dim3(nb), dim3(256), s, d_order, latlon, n, out_latlon,
              rank);
    AG_LAUNCH_CHECK();
  }
  return AG_OK;
}

static void ag_node_order_carve(AgCarver& c, int64_t n, AgOrderWs* w) {
  AgOrderWs t;
  t.keys_a = c.take<uint64_t>(n);
  t.keys_b = c.take<uint64_t>(n);
  t.ids_a = c.take<int32_t>(n);
  t.ids_b = c.take<int32_t>(n);
  t.bucket_of = c.take<int>(n);
  t.slot_of = c.take<int>(n);
  t.bucket_count = c.take<int>(AG_ORDER_MAX_BUCKETS + 1);
  t.bucket_start = c.take<int>(AG_ORDER_MAX_BUCKETS + 1);
  t.sort_temp_bytes = ag_sort_pairs_u64_i32_temp_bytes(n);
  t.sort_temp = c.take<char>(t.sort_temp_bytes);
  if (w) *w = t;
}

extern "C" size_t ag_node_order_workspace_bytes(int64_t n) {
  if (n < 1) return 0;
  AgCarver c(nullptr);
  ag_node_order_carve(c, n, nullptr);
  return c.total();
}

extern "C" int ag_node_order(const float* d_latlon, int64_t n, int32_t* d_order, void* d_ws, size_t ws_bytes,
                             void* stream) {
  AG_REQUIRE(n >= 0 && n < (int64_t(1) << 31), "ag_node_order: node count must fit int32");
  if (n == 0) return AG_OK;
  AG_REQUIRE(d_latlon && d_order && d_ws, "ag_node_order: null pointer");
  if (ws_bytes < ag_node_order_workspace_bytes(n)) return ag_fail(AG_ERR_WORKSPACE, "ag_node_order: workspace too small");
  AgCarver c(d_ws);
  AgOrderWs w;
  ag_node_order_carve(c, n, &w);
  return ag_order_nodes(nullptr, const_cast<float2*>(reinterpret_cast<const float2*>(d_latlon)), n, w, d_order, nullptr,
                        nullptr, false, static_cast<hipStream_t>(stream));
}

extern "C" int ag_icosphere_sizes(int R, int64_t* V, int64_t* F, int64_t* E_all) {
  AG_REQUIRE(R >= 0 && R <= AG_ICO_MAX_LEVEL, "ag_icosphere: resolution %d outside [0, %d]", R, AG_ICO_MAX_LEVEL);
  if (V) *V = ag_ico_V(R);
  if (F) *F = ag_ico_F(R);
  if (E_all) *E_all = ag_ico_edge_offset(R + 1);
  return AG_OK;
}

extern "C" size_t ag_icosphere_workspace_bytes(int R) {
  if (R < 0 || R > AG_ICO_MAX_LEVEL) return 0;
  AgCarver c(nullptr);
  ag_topo_carve(c, R, nullptr);
  (void)c.take<float2>(ag_ico_V(R));
  return c.total();
}

extern "C" int ag_icosphere(int R, double* d_xyz, int32_t* d_faces, float* d_latlon, int32_t* d_order, int32_t* d_rank,
                            uint64_t* d_level_edges, void* d_ws, size_t ws_bytes, void* stream) {
  AG_REQUIRE(R >= 0 && R <= AG_ICO_MAX_LEVEL, "ag_icosphere: resolution %d outside [0, %d]", R, AG_ICO_MAX_LEVEL);
  AG_REQUIRE(d_xyz && d_faces && d_latlon && d_ws, "ag_icosphere: null pointer");
  AG_REQUIRE((d_order == nullptr) == (d_rank == nullptr), "ag_icosphere: d_order and d_rank go together (both, or neither)");
  if (ws_bytes < ag_icosphere_workspace_bytes(R)) return ag_fail(AG_ERR_WORKSPACE, "ag_icosphere: workspace too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  AgCarver c(d_ws);
  AgTopoWs w;
  ag_topo_carve(c, R, &w);
  float2* latlon = c.take<float2>(ag_ico_V(R));
  // ping-pong between the caller's face buffer and the workspace so that level R lands in d_faces
  int32_t* cur = (R % 2 == 0) ? d_faces : w.faces_a;
  int32_t* nxt = (R % 2 == 0) ? w.faces_a : d_faces;
  int rc = AG_OK;
  // levels 0 .. small-1 in one block; the face ping-pong continues from where it ends.  The same kernel clears the
  // error flag and the bucket counters of the node-order sort at the end of this call.
  const int small = R < AG_ICO_SMALL_LEVELS ? R : AG_ICO_SMALL_LEVELS;
  AG_KERNEL("ag_ico_small_levels_kernel", ag_ico_small_levels_kernel, dim3(1), dim3(1024), s, small, d_xyz, cur, nxt,
            d_level_edges, w.uniq, w.first, w.scan_state, static_cast<int>(w.scan_words), w.error_flag, w.bucket_count,
            AG_ORDER_MAX_BUCKETS + 1);
  AG_LAUNCH_CHECK();
  if (small % 2 == 1) {
    int32_t* t = cur;
    cur = nxt;
    nxt = t;
  }
  if (d_level_edges && small == R && R > 0) {
    // the last subdivision happened inside the small kernel: its parent edges and child faces are in place
    const int l = R - 1;
    const int64_t F = ag_ico_F(l), Eu = ag_ico_E(l), V = ag_ico_V(l);
    AG_KERNEL("ag_ico_final_edges_kernel", ag_ico_final_edges_kernel, dim3(ag_blocks(Eu + F, 256, 8192)), dim3(256), s,
              d_level_edges + ag_ico_edge_offset(l), Eu, V, cur, F, d_level_edges + ag_ico_edge_offset(R));
    AG_LAUNCH_CHECK();
  }
  int* first = w.first;      // zeroed by the small-levels kernel
  int* first_next = w.first_alt;
  for (int l = small; l < R; ++l) {
    const int64_t F = ag_ico_F(l), Eu = ag_ico_E(l), V = ag_ico_V(l);
    // the unique edges of level l go straight into the per-level edge array when the caller wants it
    uint64_t* uniq = d_level_edges ? d_level_edges + ag_ico_edge_offset(l) : w.uniq;
    rc = ag_topo_unique_edges(w, first, cur, l, uniq, d_xyz, true, s);
    if (rc) return rc;
    const bool last = d_level_edges && l == R - 1;
    AG_KERNEL("ag_ico_refine_kernel", ag_ico_refine_kernel, dim3(ag_blocks(V + Eu, 256, 8192)), dim3(256), s, cur, F, first,
              w.nb, V, Eu, nxt, d_xyz, first_next, w.scan_state, w.scan_words, uniq,
              last ? d_level_edges + ag_ico_edge_offset(R) : nullptr);
    AG_LAUNCH_CHECK();
    int32_t* t = cur;
    cur = nxt;
    nxt = t;
    int* tf = first;
    first = first_next;
    first_next = tf;
  }
  if (d_level_edges && R == 0) {
    rc = ag_topo_unique_edges(w, first, cur, 0, d_level_edges, d_xyz, false, s);
    if (rc) return rc;
  }
  const int64_t V = ag_ico_V(R);
  const int vb = ag_blocks(V, 256, 8192);
  if (d_order == nullptr) {  // no node order wanted: the coordinates in vertex order
    AG_KERNEL("ag_vertex_latlon_kernel", ag_vertex_latlon_kernel, dim3(vb), dim3(256), s, d_xyz,
              reinterpret_cast<float2*>(d_latlon), V);
    AG_LAUNCH_CHECK();
    return AG_OK;
  }
  AgOrderWs ow;
  ow.keys_a = w.keys_a;
  ow.keys_b = w.keys_b;
  ow.ids_a = w.ids_a;
  ow.ids_b = w.ids_b;
  ow.bucket_of = w.bucket_of;
  ow.slot_of = w.slot_of;
  ow.bucket_count = w.bucket_count;
  ow.bucket_start = w.bucket_start;
  ow.sort_temp = w.sort_temp;
  ow.sort_temp_bytes = w.sort_temp_bytes;
  return ag_order_nodes(d_xyz, latlon, V, ow, d_order, reinterpret_cast<float2*>(d_latlon), d_rank, true, s);
}

// ---------------------------------------------------------------------------------------------
// reference distance of the complete icosphere, from its mesh
// ---------------------------------------------------------------------------------------------
// The triangulation of a refined icosahedron is a Delaunay triangulation of its vertices (every triangle is close to
// equilateral: the two angles opposite an edge add up to far less than 180 degrees), and the nearest neighbour of a
// point is always one of its Delaunay neighbours.  So the distance from a node to its nearest other node is the
// length of its shortest mesh edge, and the grid reference distance -- the largest such distance -- needs no search:
// one pass over the edges of the finest level (the reduced distance scikit-learn ranks by, from the float32
// coordinates of the nodes -- the same bits from either end: the sine used is an odd function bit for bit, its
// squares do not see the order of the two nodes, and the product of the two cosines commutes), a minimum per vertex,
// the maximum of the minima.
__global__ void ag_mesh_maxdist_clear_kernel(unsigned long long* __restrict__ vertex_min, int64_t V,
                                             unsigned long long* __restrict__ max_bits) {
  for (int64_t v = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; v < V; v += int64_t(gridDim.x) * blockDim.x)
    vertex_min[v] = 0x7ff0000000000000ull;  // +infinity
  if (blockIdx.x == 0 && threadIdx.x == 0) max_bits[0] = max_bits[1] = 0ull;  // the maximum, the blocks done
}

__global__ void ag_mesh_maxdist_edges_kernel(const uint64_t* __restrict__ edges, int64_t E, const float2* __restrict__ latlon,
                                             const int32_t* __restrict__ rank, int64_t V,
                                             unsigned long long* __restrict__ vertex_min) {
  for (int64_t e = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; e < E; e += int64_t(gridDim.x) * blockDim.x) {
    const uint64_t key = edges[e];
    const int64_t a = static_cast<int64_t>(key & 0xffffffffull), b = static_cast<int64_t>(key >> 32);
    if (a >= V || b >= V || a == b) continue;  // (not an edge of this mesh)
    const int ra = rank[a], rb = rank[b];
    if (ra < 0 || ra >= V || rb < 0 || rb >= V) continue;
    const float2 pa = latlon[ra], pb = latlon[rb];
    // (positive float64 values order like their bit patterns)
    const unsigned long long r = static_cast<unsigned long long>(__double_as_longlong(ag_rdist(pa.x, pa.y, pb.x, pb.y)));
    atomicMin(vertex_min + a, r);
    atomicMin(vertex_min + b, r);
  }
}

// (the block that finishes last turns the maximum into the distance: the blocks are few, so their counting in costs
// less than a launch)
__global__ __launch_bounds__(256) void ag_mesh_maxdist_max_kernel(const unsigned long long* __restrict__ vertex_min, int64_t V,
                                                                  unsigned long long* __restrict__ max_bits,
                                                                  double* __restrict__ out) {
  __shared__ unsigned long long partial[4];
  unsigned long long best = 0ull;
  for (int64_t v = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; v < V; v += int64_t(gridDim.x) * blockDim.x) {
    const unsigned long long m = vertex_min[v];
    // as dists[dists > 0].max() of the reference: a vertex without a positive distance does not count
    if (m != 0x7ff0000000000000ull && m != 0ull) best = m > best ? m : best;
  }
#pragma unroll
  for (int ofs = 32; ofs > 0; ofs >>= 1) {
    const unsigned long long other = __shfl_xor(best, ofs, 64);
    best = other > best ? other : best;
  }
  if ((threadIdx.x & 63) == 0) partial[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) best = partial[w] > best ? partial[w] : best;
    if (best) atomicMax(max_bits, best);
    __threadfence();
    if (atomicAdd(max_bits + 1, 1ull) == gridDim.x - 1) {
      const double r = __longlong_as_double(static_cast<long long>(atomicMax(max_bits, 0ull)));
      *out = (r > 0.0) ? ag_rdist_to_dist(r) : 0.0;
    }
  }
}

extern "C" size_t ag_mesh_maxdist_workspace_bytes(int R) {
  if (R < 0 || R > AG_ICO_MAX_LEVEL) return 0;
  return sizeof(unsigned long long) * (ag_ico_V(R) + 2) + 512;
}

extern "C" int ag_mesh_maxdist(int R, const uint64_t* d_level_edges, const float* d_latlon, const int32_t* d_rank,
                               double* d_out_max, void* d_ws, size_t ws_bytes, void* stream) {
  AG_REQUIRE(R >= 0 && R <= AG_ICO_MAX_LEVEL, "ag_mesh_maxdist: resolution %d outside [0, %d]", R, AG_ICO_MAX_LEVEL);
  AG_REQUIRE(d_level_edges && d_latlon && d_rank && d_out_max && d_ws, "ag_mesh_maxdist: null pointer");
  if (ws_bytes < ag_mesh_maxdist_workspace_bytes(R)) return ag_fail(AG_ERR_WORKSPACE, "ag_mesh_maxdist: workspace too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t V = ag_ico_V(R), E = ag_ico_E(R);
  AgCarver c(d_ws);
  unsigned long long* vertex_min = c.take<unsigned long long>(V);
  unsigned long long* max_bits = c.take<unsigned long long>(2);
  AG_KERNEL("ag_mesh_maxdist_clear_kernel", ag_mesh_maxdist_clear_kernel, dim3(ag_blocks(V, 256, 4096)), dim3(256), s, vertex_min, V,
            max_bits);
  AG_LAUNCH_CHECK();
  AG_KERNEL("ag_mesh_maxdist_edges_kernel", ag_mesh_maxdist_edges_kernel, dim3(ag_blocks(E, 256, 8192)), dim3(256), s,
            d_level_edges + ag_ico_edge_offset(R), E, reinterpret_cast<const float2*>(d_latlon), d_rank, V, vertex_min);
  AG_LAUNCH_CHECK();
  AG_KERNEL("ag_mesh_maxdist_max_kernel", ag_mesh_maxdist_max_kernel, dim3(ag_blocks(V, 1024, 64)), dim3(256), s, vertex_min, V,
            max_bits, d_out_max);
  AG_LAUNCH_CHECK();
  return AG_OK;
}

// ---------------------------------------------------------------------------------------------
// multi-scale edges
// ---------------------------------------------------------------------------------------------
static inline int ag_hops_capacity(int x_hops) { return 3 * x_hops * (x_hops + 1); }

// Pair keys are (target rank << B) | source rank with B = bits of the node count, so that the sort
// only has to look at 2B bits; the all-ones 2B-bit value is larger than every real key and marks
// an unused slot.
__device__ __forceinline__ uint64_t ag_pair_key(int src_node, int tgt_node, int B) {
  return (static_cast<uint64_t>(static_cast<uint32_t>(tgt_node)) << B) | static_cast<uint32_t>(src_node);
}
__host__ __device__ __forceinline__ uint64_t ag_no_pair(int B) { return (uint64_t(1) << (2 * B)) - 1; }

#define AG_MS_MAX_LEVELS (AG_ICO_MAX_LEVEL + 1)
struct AgMsLevels {
  int n;
  int64_t edge_off[AG_MS_MAX_LEVELS];  // offset of the level in the per-level edge array
  int64_t end[AG_MS_MAX_LEVELS];       // running total of edges over the selected levels
};

// x_hops == 1: both directions of every (active) mesh edge of every selected level (one launch
// for all levels).  No pair can occur twice -- an edge of one level is split at the next -- so the
// pairs only have to be grouped by target and ordered by source: every pair takes a slot in its
// target's row here (one atomic counter per node), a prefix sum over the counters gives the row
// offsets, the fill pass scatters the sources into their rows and sorts each short row.
__global__ void ag_ms_edge_pairs_kernel(const uint64_t* __restrict__ level_edges, AgMsLevels lv,
                                        const uint8_t* __restrict__ active, const int32_t* __restrict__ rank, int B,
                                        uint64_t* __restrict__ pairs, int* __restrict__ slot, int* __restrict__ row_count) {
  const int64_t total = lv.end[lv.n - 1];
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < total; i += int64_t(gridDim.x) * blockDim.x) {
    int64_t start = 0, eoff = lv.edge_off[0];
#pragma unroll
    for (int j = 1; j < AG_MS_MAX_LEVELS; ++j) {
      if (i >= lv.end[j - 1]) {  // never true past the last selected level: end[] is padded with the total
        start = lv.end[j - 1];
        eoff = lv.edge_off[j];
      }
    }
    const uint64_t k = level_edges[eoff + (i - start)];
    const int a = static_cast<int>(static_cast<uint32_t>(k)), b = static_cast<int>(static_cast<uint32_t>(k >> 32));
    uint64_t p0 = ag_no_pair(B), p1 = ag_no_pair(B);
    int s0 = 0, s1 = 0;
    if (active == nullptr || (active[a] && active[b])) {
      const int ra = rank[a], rb = rank[b];
      if (ra >= 0 && rb >= 0) {
        p0 = ag_pair_key(ra, rb, B);
        p1 = ag_pair_key(rb, ra, B);
        s0 = atomicAdd(&row_count[rb], 1);
        s1 = atomicAdd(&row_count[ra], 1);
      }
    }
    pairs[2 * i] = p0;
    pairs[2 * i + 1] = p1;
    slot[2 * i] = s0;
    slot[2 * i + 1] = s1;
  }
}

// x_hops == 1 on the COMPLETE icosphere (no vertex mask): the rows need no counting pass.  A vertex v exists
// from its birth level on (the first level with more than v vertices) and has 5 neighbours on every level if it
// is one of the 12 icosahedron vertices, else 6, so its row holds deg(v) entries per selected level it exists on.
// Also clears the per-row counters of the placement pass and the state of the scan that follows.
__global__ void ag_ms_row_sizes_kernel(int64_t V, unsigned level_mask, int R, const int32_t* __restrict__ rank,
                                       int* __restrict__ row_count, int2* __restrict__ placed,
                                       unsigned long long* __restrict__ scan_state, int64_t scan_words) {
  const int64_t i0 = blockIdx.x * int64_t(blockDim.x) + threadIdx.x, step = int64_t(gridDim.x) * blockDim.x;
  for (int64_t v = i0; v < V; v += step) {
    int levels_alive = 0;
    for (int l = 0; l <= R; ++l)
      if (((level_mask >> l) & 1u) && (10ll << (2 * l)) + 2 > v) ++levels_alive;
    const int r = rank[v];
    if (r >= 0 && r < V) row_count[r] = (v < 12 ? 5 : 6) * levels_alive;
    placed[v] = make_int2(0, 0);
  }
  if (i0 == 0) row_count[V] = row_count[V + 1] = 0;
  for (int64_t i = i0; i < scan_words; i += step) scan_state[i] = 0ull;
}

// Placement on the complete icosphere, one thread per mesh edge (min, max) of the selected levels; every edge
// puts `min` into the row of `max` and `max` into the row of `min`.
//  * A vertex has smaller neighbours on its birth level only (from the next level on all its neighbours are
//    midpoints created later), so the entries `min` of a row form one block, kept at the front of the row; the
//    entries `max` are taken from the back of the row with one atomic counter per row.  The rows are ordered by
//    the pass that follows, so the order inside the two blocks is free.
//  * Front slots mostly need no counter.  The edges of every level but the mesh's finest are ordered by
//    (max, min): the slot is the number of edges with the same `max` right before this one.  The finest level R is
//    stored as ag_icosphere leaves it -- for every edge i of level R-1 with midpoint m = V(R-1) + i the pair
//    (a, m), (b, m) at 2i and 2i+1, then the three sides of every central child face: the pair takes slots 0 and
//    1 of the row of m, the sides take the slots from 2 on with a second counter.
__global__ void ag_ms_place_kernel(const uint64_t* __restrict__ level_edges, AgMsLevels lv, int final_index,
                                   int64_t final_split, const int32_t* __restrict__ rank,
                                   const int* __restrict__ row_start, int64_t V, int64_t E, int2* __restrict__ placed,
                                   int32_t* __restrict__ out_src, int32_t* __restrict__ out_tgt) {
  const int64_t total = lv.end[lv.n - 1];
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < total; i += int64_t(gridDim.x) * blockDim.x) {
    int64_t start = 0, eoff = lv.edge_off[0];
    int which = 0;
#pragma unroll
    for (int j = 1; j < AG_MS_MAX_LEVELS; ++j) {
      if (i >= lv.end[j - 1]) {
        start = lv.end[j - 1];
        eoff = lv.edge_off[j];
        which = j;
      }
    }
    const int64_t local = i - start;
    const uint64_t* edges = level_edges + eoff;
    const uint64_t k = edges[local];
    const int vmin = static_cast<int>(static_cast<uint32_t>(k)), vmax = static_cast<int>(static_cast<uint32_t>(k >> 32));
    if (vmin < 0 || vmax >= V || vmin >= vmax) continue;  // not an edge of this mesh
    const int rmin = rank[vmin], rmax = rank[vmax];
    if (rmin < 0 || rmin >= V || rmax < 0 || rmax >= V) continue;  // not a complete mesh: nothing is placed
    int front_slot;
    if (which != final_index) {
      int before = 0;
      bool same = true;
#pragma unroll
      for (int d = 1; d < AG_MAX_DEG; ++d) {
        const int64_t q = local - d;
        same = same && q >= 0 && static_cast<int>(static_cast<uint32_t>(edges[q >= 0 ? q : 0] >> 32)) == vmax;
        before += same ? 1 : 0;
      }
      front_slot = before;
    } else if (local < final_split) {
      front_slot = static_cast<int>(local & 1);
    } else {
      front_slot = 2 + atomicAdd(&placed[vmax].x, 1);
    }
    const int front = row_start[rmax] + front_slot;
    if (front >= 0 && front < row_start[rmax + 1] && front < E) {
      out_src[front] = rmin;
      out_tgt[front] = rmax;
    }
    const int back = row_start[rmin + 1] - 1 - atomicAdd(&placed[vmin].y, 1);
    if (back >= 0 && back >= row_start[rmin] && back < E) {
      out_src[back] = rmax;
      out_tgt[back] = rmin;
    }
  }
}

__global__ void ag_ms_scatter_kernel(const uint64_t* __restrict__ pairs, const int* __restrict__ slot, int64_t n, int B,
                                     const int* __restrict__ row_start, int32_t* __restrict__ out_src,
                                     int32_t* __restrict__ out_tgt) {
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
    const uint64_t k = pairs[i];
    if (k == ag_no_pair(B)) continue;
    const int src = static_cast<int>(k & ((uint64_t(1) << B) - 1)), tgt = static_cast<int>(k >> B);
    const int64_t pos = int64_t(row_start[tgt]) + slot[i];
    out_src[pos] = src;
    out_tgt[pos] = tgt;
  }
}

// one thread per edge: ascending sources inside every row.  Sources are distinct inside a row, so
// the position of an entry is the number of smaller entries of its row (rows are short: at most
// 6 per level).
__global__ void ag_ms_row_rank_kernel(const int* __restrict__ row_start, const int32_t* __restrict__ unsorted,
                                      const int32_t* __restrict__ out_tgt, int64_t E, int64_t V,
                                      int32_t* __restrict__ out_src) {
  for (int64_t e = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; e < E; e += int64_t(gridDim.x) * blockDim.x) {
    const int t = out_tgt[e];
    if (t < 0 || t >= V) continue;  // (an entry nothing was placed into)
    const int first = row_start[t], n = row_start[t + 1] - first;
    if (first < 0 || n < 0 || int64_t(first) + n > E) continue;  // (row offsets of an inconsistent input)
    const int32_t mine = unsorted[e];
    int r = 0;
    for (int j = 0; j < n; ++j) r += (unsorted[first + j] < mine) ? 1 : 0;
    out_src[first + r] = mine;
  }
}

__global__ void ag_ms_adjacency_kernel(const uint64_t* __restrict__ uniq, int64_t Eu, const uint8_t* __restrict__ active,
                                       int* __restrict__ deg, int* __restrict__ adj) {
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < Eu; i += int64_t(gridDim.x) * blockDim.x) {
    const uint64_t k = uniq[i];
    const int a = static_cast<int>(static_cast<uint32_t>(k)), b = static_cast<int>(static_cast<uint32_t>(k >> 32));
    if (active != nullptr && !(active[a] && active[b])) continue;
    const int sa = atomicAdd(&deg[a], 1);
    if (sa < AG_MAX_DEG) adj[a * AG_MAX_DEG + sa] = b;
    const int sb = atomicAdd(&deg[b], 1);
    if (sb < AG_MAX_DEG) adj[b * AG_MAX_DEG + sb] = a;
  }
}

// x_hops > 1, one thread per level vertex: breadth-first expansion, one fixed slot per possible neighbour
template <int CAP>
__global__ __launch_bounds__(64) void ag_ms_pairs_kernel(int64_t V, int x_hops, const int* __restrict__ deg,
                                                         const int* __restrict__ adj, const int32_t* __restrict__ rank,
                                                         int B, uint64_t* __restrict__ pairs) {
  const int64_t v = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (v >= V) return;
  int list[CAP];
  int n = 0;
  const int cap = min(CAP, 3 * x_hops * (x_hops + 1));
  int fb = 0, fe = 0;
  {
    const int d = min(deg[v], AG_MAX_DEG);
    for (int j = 0; j < d; ++j) {
      const int u = adj[v * AG_MAX_DEG + j];
      if (n < cap) list[n++] = u;
    }
    fe = n;
  }
  for (int hop = 2; hop <= x_hops; ++hop) {
    for (int i = fb; i < fe; ++i) {
      const int w = list[i];
      const int d = min(deg[w], AG_MAX_DEG);
      for (int j = 0; j < d; ++j) {
        const int u = adj[w * AG_MAX_DEG + j];
        if (u == static_cast<int>(v)) continue;
        bool seen = false;
        for (int q = 0; q < n; ++q) seen = seen || (list[q] == u);
        if (!seen && n < cap) list[n++] = u;
      }
    }
    fb = fe;
    fe = n;
  }
  const int rv = rank[v];
  uint64_t* out = pairs + v * cap;
  for (int j = 0; j < cap; ++j) {
    uint64_t key = ag_no_pair(B);
    if (j < n && rv >= 0) {
      const int ru = rank[list[j]];
      if (ru >= 0) key = ag_pair_key(rv, ru, B);
    }
    out[j] = key;
  }
}

// sorted pair keys -> 1 where a new distinct non-sentinel key starts
__global__ void ag_ms_flag_unique_kernel(const uint64_t* __restrict__ sorted, int64_t n, int B,
                                         int64_t* __restrict__ flags) {
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
    const uint64_t k = sorted[i];
    flags[i] = (k != ag_no_pair(B) && (i == 0 || sorted[i - 1] != k)) ? 1 : 0;
  }
}

__global__ void ag_ms_copy_total_kernel(const int64_t* __restrict__ offsets, int64_t n, int64_t* __restrict__ out) {
  *out = offsets[n];
}

__global__ void ag_ms_emit_kernel(const uint64_t* __restrict__ sorted, const int64_t* __restrict__ offsets, int64_t n,
                                  int B, int64_t num_edges, int32_t* __restrict__ out_src,
                                  int32_t* __restrict__ out_tgt) {
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
    const uint64_t k = sorted[i];
    if (k != ag_no_pair(B) && (i == 0 || sorted[i - 1] != k)) {
      const int64_t o = offsets[i];
      if (o < num_edges) {
        out_src[o] = static_cast<int32_t>(k & ((uint64_t(1) << B) - 1));
        out_tgt[o] = static_cast<int32_t>(k >> B);
      }
    }
  }
}

struct AgMsWs {
  int* deg;           // [V_R]        (x_hops > 1)
  int* adj;           // [V_R * 6]    (x_hops > 1)
  uint64_t* pairs_a;  // [total slots]
  uint64_t* pairs_b;
  int64_t* offsets;   // [total slots + 1]
  int* slot;          // [total slots]  position of a pair inside its target's row   (x_hops == 1)
  int* row_start;     // [V_R + 1]      row counters, then row offsets               (x_hops == 1)
  int64_t row_clear;  // ints to clear from row_start on: the counters and the scan state behind them
  unsigned long long* row_scan_state;
  int32_t* unsorted;  // [total slots]  sources grouped by target, unsorted inside a row (x_hops == 1)
  int2* placed;       // [V_R]          entries placed at the front / from the back of a row (complete icosphere)
  void* temp;
  size_t temp_bytes;
  int64_t slots;
};

static int64_t ag_ms_level_slots(int l, int x_hops) {
  return x_hops == 1 ? 2 * ag_ico_E(l) : ag_ico_V(l) * ag_hops_capacity(x_hops);
}

static int ag_ms_carve(AgCarver& c, int R, const int32_t* levels, int nlevels, int x_hops, AgMsWs* w) {
  AgMsWs m;
  m.deg = x_hops > 1 ? c.take<int>(ag_ico_V(R)) : nullptr;
  m.adj = x_hops > 1 ? c.take<int>(ag_ico_V(R) * AG_MAX_DEG) : nullptr;
  // sizing without the level list (levels == nullptr) assumes the nlevels finest levels
  int64_t slots = 0;
  for (int i = 0; i < nlevels; ++i) {
    const int l = levels ? levels[i] : (R - i < 0 ? 0 : R - i);
    slots += ag_ms_level_slots(l, x_hops);
  }
  m.slots = slots;
  m.pairs_a = c.take<uint64_t>(slots);
  m.pairs_b = x_hops > 1 ? c.take<uint64_t>(slots) : nullptr;
  m.offsets = x_hops > 1 ? c.take<int64_t>(slots + 1) : nullptr;
  m.slot = x_hops == 1 ? c.take<int>(slots) : nullptr;
  // row counters and, right behind them, the state words of their scan: one clearing launch covers both
  const int64_t row_ints = ag_ico_V(R) + 2 + (ag_ico_V(R) & 1);  // even, so that the state is 8-byte aligned
  const int64_t scan_words = static_cast<int64_t>(ag_chained_scan_state_words_i32(ag_ico_V(R) + 2));
  m.row_start = x_hops == 1 ? c.take<int>(row_ints + 2 * scan_words) : nullptr;
  m.row_clear = row_ints + 2 * scan_words;
  m.row_scan_state = x_hops == 1 ? reinterpret_cast<unsigned long long*>(m.row_start + row_ints) : nullptr;
  m.unsorted = x_hops == 1 ? c.take<int32_t>(slots) : nullptr;
  m.placed = x_hops == 1 ? c.take<int2>(ag_ico_V(R)) : nullptr;
  size_t t1 = ag_sort_u64_temp_bytes(slots), t2 = ag_scan_i64_temp_bytes(slots);
  if (x_hops == 1) {
    t1 = ag_scan_i32_temp_bytes(ag_ico_V(R) + 2);
    t2 = 0;
  }
  m.temp_bytes = t1 > t2 ? t1 : t2;
  m.temp = c.take<char>(m.temp_bytes);
  if (w) *w = m;
  return AG_OK;
}

extern "C" size_t ag_multiscale_workspace_bytes(int R, int nlevels, int x_hops) {
  if (R < 0 || R > AG_ICO_MAX_LEVEL || nlevels < 1 || x_hops < 1 || x_hops > AG_MULTISCALE_MAX_HOPS) return 0;
  AgCarver c(nullptr);
  ag_ms_carve(c, R, nullptr, nlevels, x_hops, nullptr);
  return c.total();
}

static int ag_ms_check(int R, const int32_t* levels, int nlevels, int x_hops) {
  AG_REQUIRE(R >= 0 && R <= AG_ICO_MAX_LEVEL, "ag_multiscale: resolution %d outside [0, %d]", R, AG_ICO_MAX_LEVEL);
  AG_REQUIRE(levels != nullptr && nlevels >= 1 && nlevels <= R + 1, "ag_multiscale: bad level list");
  AG_REQUIRE(x_hops >= 1 && x_hops <= AG_MULTISCALE_MAX_HOPS, "ag_multiscale: x_hops %d outside [1, %d]", x_hops,
             AG_MULTISCALE_MAX_HOPS);
  for (int i = 0; i < nlevels; ++i) {
    AG_REQUIRE(levels[i] >= 0 && levels[i] <= R, "ag_multiscale: level %d outside [0, %d]", levels[i], R);
    for (int j = 0; j < i; ++j) AG_REQUIRE(levels[i] != levels[j], "ag_multiscale: duplicate level %d", levels[i]);
  }
  return AG_OK;
}

template <int CAP>
static void ag_ms_launch_pairs(int64_t V, int x_hops, const AgMsWs& m, const int32_t* rank, int B, uint64_t* pairs,
                               hipStream_t s) {
  AG_KERNEL("ag_ms_pairs_kernel", ag_ms_pairs_kernel<CAP>, dim3(ag_blocks(V, 64)), dim3(64), s, V, x_hops, m.deg, m.adj,
            rank, B, pairs);
}

extern "C" int ag_multiscale_count(int R, const int32_t* levels, int nlevels, int x_hops, const int32_t* d_rank,
                                   const uint8_t* d_active, const uint64_t* d_level_edges, int64_t* d_num_edges,
                                   void* d_ws, size_t ws_bytes, void* stream) {
  int rc = ag_ms_check(R, levels, nlevels, x_hops);
  if (rc) return rc;
  AG_REQUIRE(d_rank && d_level_edges && d_num_edges && d_ws, "ag_multiscale_count: null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  AgCarver c(d_ws);
  AgMsWs m;
  ag_ms_carve(c, R, levels, nlevels, x_hops, &m);
  if (ws_bytes < c.total()) return ag_fail(AG_ERR_WORKSPACE, "ag_multiscale_count: workspace too small");
  const int cap = ag_hops_capacity(x_hops);
  const int B = ag_vertex_bits(R);
  int64_t slot_base = 0;
  if (x_hops == 1) {
    AgMsLevels lv;
    lv.n = nlevels;
    int64_t run = 0;
    for (int i = 0; i < nlevels; ++i) {
      lv.edge_off[i] = ag_ico_edge_offset(levels[i]);
      run += ag_ico_E(levels[i]);
      lv.end[i] = run;
    }
    for (int i = nlevels; i < AG_MS_MAX_LEVELS; ++i) {
      lv.edge_off[i] = 0;
      lv.end[i] = run;
    }
    const int64_t V = ag_ico_V(R);
    rc = ag_zero_i32(m.row_start, m.row_clear, s);
    if (rc) return rc;
    AG_KERNEL("ag_ms_edge_pairs_kernel", ag_ms_edge_pairs_kernel, dim3(ag_blocks(run, 256, 8192)), dim3(256), s,
              d_level_edges, lv, d_active, d_rank, B, m.pairs_a, m.slot, m.row_start);
    AG_LAUNCH_CHECK();
    // row offsets; the total lands in row_start[V] and, as int64, in *d_num_edges
    return ag_chained_scan_i32(m.row_start, m.row_start, V, m.row_scan_state, s,
                               reinterpret_cast<long long*>(d_num_edges));
  } else {
    for (int i = 0; i < nlevels; ++i) {
      const int l = levels[i];
      const int64_t Eu = ag_ico_E(l), V = ag_ico_V(l);
      const uint64_t* edges = d_level_edges + ag_ico_edge_offset(l);
      uint64_t* pairs = m.pairs_a + slot_base;
      rc = ag_zero_i32(m.deg, V, s);
      if (rc) return rc;
      AG_KERNEL("ag_ms_adjacency_kernel", ag_ms_adjacency_kernel, dim3(ag_blocks(Eu, 256, 8192)), dim3(256), s, edges,
                Eu, d_active, m.deg, m.adj);
      AG_LAUNCH_CHECK();
      if (cap <= 18) ag_ms_launch_pairs<18>(V, x_hops, m, d_rank, B, pairs, s);
      else if (cap <= 36) ag_ms_launch_pairs<36>(V, x_hops, m, d_rank, B, pairs, s);
      else if (cap <= 90) ag_ms_launch_pairs<90>(V, x_hops, m, d_rank, B, pairs, s);
      else ag_ms_launch_pairs<216>(V, x_hops, m, d_rank, B, pairs, s);
      AG_LAUNCH_CHECK();
      slot_base += ag_ms_level_slots(l, x_hops);
    }
  }
  const int64_t n = slot_base;
  rc = ag_sort_u64(m.temp, m.temp_bytes, m.pairs_a, m.pairs_b, n, 0, 2 * B, s);
  if (rc) return rc;
  AG_KERNEL("ag_ms_flag_unique_kernel", ag_ms_flag_unique_kernel, dim3(ag_blocks(n, 256, 8192)), dim3(256), s, m.pairs_b,
            n, B, m.offsets);
  AG_LAUNCH_CHECK();
  rc = ag_exclusive_scan_i64(m.temp, m.temp_bytes, m.offsets, m.offsets, n, s);
  if (rc) return rc;
  AG_KERNEL("ag_ms_copy_total_kernel", ag_ms_copy_total_kernel, dim3(1), dim3(1), s, m.offsets, n, d_num_edges);
  AG_LAUNCH_CHECK();
  return AG_OK;
}

extern "C" int ag_multiscale_fill(int R, const int32_t* levels, int nlevels, int x_hops, int64_t num_edges,
                                  int32_t* d_out_src, int32_t* d_out_tgt, void* d_ws, size_t ws_bytes, void* stream) {
  int rc = ag_ms_check(R, levels, nlevels, x_hops);
  if (rc) return rc;
  if (num_edges <= 0) return AG_OK;
  AG_REQUIRE(d_out_src && d_out_tgt && d_ws, "ag_multiscale_fill: null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  AgCarver c(d_ws);
  AgMsWs m;
  ag_ms_carve(c, R, levels, nlevels, x_hops, &m);
  if (ws_bytes < c.total()) return ag_fail(AG_ERR_WORKSPACE, "ag_multiscale_fill: workspace too small");
  if (x_hops == 1) {
    const int64_t V = ag_ico_V(R);
    AG_KERNEL("ag_ms_scatter_kernel", ag_ms_scatter_kernel, dim3(ag_blocks(m.slots, 256, 8192)), dim3(256), s, m.pairs_a,
              m.slot, m.slots, ag_vertex_bits(R), m.row_start, m.unsorted, d_out_tgt);
    AG_LAUNCH_CHECK();
    AG_KERNEL("ag_ms_row_rank_kernel", ag_ms_row_rank_kernel, dim3(ag_blocks(num_edges, 256, 8192)), dim3(256), s,
              m.row_start, m.unsorted, d_out_tgt, num_edges, V, d_out_src);
    AG_LAUNCH_CHECK();
    return AG_OK;
  }
  AG_KERNEL("ag_ms_emit_kernel", ag_ms_emit_kernel, dim3(ag_blocks(m.slots, 256, 8192)), dim3(256), s, m.pairs_b,
            m.offsets, m.slots, ag_vertex_bits(R), num_edges, d_out_src, d_out_tgt);
  AG_LAUNCH_CHECK();
  return AG_OK;
}

// number of multi-scale edges of the complete icosphere for x_hops == 1: level l has 30 * 4^l mesh edges, each
// listed in both directions, and no pair occurs on two levels (an edge of one level is split on the next)
extern "C" int64_t ag_multiscale_complete_edges(const int32_t* levels, int nlevels) {
  int64_t n = 0;
  for (int i = 0; levels != nullptr && i < nlevels; ++i)
    if (levels[i] >= 0 && levels[i] <= AG_ICO_MAX_LEVEL) n += 2 * ag_ico_E(levels[i]);
  return n;
}

extern "C" int ag_multiscale_complete(int R, const int32_t* levels, int nlevels, const int32_t* d_rank,
                                      const uint64_t* d_level_edges, int32_t* d_out_src, int32_t* d_out_tgt, void* d_ws,
                                      size_t ws_bytes, void* stream) {
  int rc = ag_ms_check(R, levels, nlevels, 1);
  if (rc) return rc;
  AG_REQUIRE(d_rank && d_level_edges && d_out_src && d_out_tgt && d_ws, "ag_multiscale_complete: null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  AgCarver c(d_ws);
  AgMsWs m;
  ag_ms_carve(c, R, levels, nlevels, 1, &m);
  if (ws_bytes < c.total()) return ag_fail(AG_ERR_WORKSPACE, "ag_multiscale_complete: workspace too small");
  AgMsLevels lv;
  lv.n = nlevels;
  int64_t run = 0;
  unsigned level_mask = 0u;
  int final_index = -1;  // position in the level list of the mesh's finest level, whose edges are stored unsorted
  for (int i = 0; i < nlevels; ++i) {
    lv.edge_off[i] = ag_ico_edge_offset(levels[i]);
    run += ag_ico_E(levels[i]);
    lv.end[i] = run;
    level_mask |= 1u << levels[i];
    if (levels[i] == R && R > 0) final_index = i;
  }
  const int64_t final_split = R > 0 ? 2 * ag_ico_E(R - 1) : 0;
  for (int i = nlevels; i < AG_MS_MAX_LEVELS; ++i) {
    lv.edge_off[i] = 0;
    lv.end[i] = run;
  }
  const int64_t V = ag_ico_V(R), num_edges = 2 * run;
  AG_REQUIRE(num_edges < (int64_t(1) << 31), "ag_multiscale_complete: too many edges for 32-bit row offsets");
  const int64_t scan_words = static_cast<int64_t>(ag_chained_scan_state_words_i32(V + 2));
  AG_KERNEL("ag_ms_row_sizes_kernel", ag_ms_row_sizes_kernel, dim3(ag_blocks(V, 256, 8192)), dim3(256), s, V, level_mask,
            R, d_rank, m.row_start, m.placed, m.row_scan_state, scan_words);
  AG_LAUNCH_CHECK();
  rc = ag_chained_scan_i32(m.row_start, m.row_start, V, m.row_scan_state, s, nullptr);
  if (rc) return rc;
  AG_KERNEL("ag_ms_place_kernel", ag_ms_place_kernel, dim3(ag_blocks(run, 256, 8192)), dim3(256), s, d_level_edges, lv,
            final_index, final_split, d_rank, m.row_start, V, num_edges, m.placed, m.unsorted, d_out_tgt);
  AG_LAUNCH_CHECK();
  AG_KERNEL("ag_ms_row_rank_kernel", ag_ms_row_rank_kernel, dim3(ag_blocks(num_edges, 256, 8192)), dim3(256), s,
            m.row_start, m.unsorted, d_out_tgt, num_edges, V, d_out_src);
  AG_LAUNCH_CHECK();
  return AG_OK;
}
