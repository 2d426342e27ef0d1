// Icosphere subdivision (TriNodes) and multi-scale mesh edges (MultiScaleEdges) for gfx950.
//
// The mesh follows trimesh.creation.icosphere: every subdivision appends one midpoint per
// unique undirected edge after the existing vertices -- unique edges enumerated in increasing
// order of (max vertex, min vertex) -- splits each face (a,b,c) into (a,ab,ca) (ab,b,bc)
// (ca,bc,c) (ab,bc,ca), and then pushes ALL vertices back to the unit sphere in float64.
// On the GPU one level is: pack the 3F half-edges into 64-bit keys, radix-sort them (every
// edge of the closed mesh then appears exactly twice, so entry 2i is unique edge i), one thread
// per unique edge writes the midpoint, one thread per face finds its three midpoints by binary
// search in the sorted edge list and writes the four child faces.
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

__global__ void ag_ico_init_kernel(double* __restrict__ xyz, int32_t* __restrict__ faces) {
  const int i = threadIdx.x;
  if (xyz != nullptr && i < 12) {
    const double t = (1.0 + sqrt(5.0)) / 2.0;
    const double n = sqrt(2.0 + t);
    for (int c = 0; c < 3; ++c) {
      // the table holds t as a literal; rebuild it from the same expression trimesh uses
      const double raw = ag_ico_vertices[i][c];
      const double v = (fabs(raw) > 1.5) ? (raw > 0 ? t : -t) : raw;
      xyz[3 * i + c] = v / n;
    }
  }
  if (i < 20)
    for (int c = 0; c < 3; ++c) faces[3 * i + c] = ag_ico_faces[i][c];
}

__device__ __forceinline__ uint64_t ag_edge_key(int a, int b) {
  const uint32_t lo = static_cast<uint32_t>(min(a, b)), hi = static_cast<uint32_t>(max(a, b));
  return (static_cast<uint64_t>(hi) << 32) | lo;
}

__global__ void ag_ico_edge_keys_kernel(const int32_t* __restrict__ faces, int64_t F, uint64_t* __restrict__ keys) {
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < 3 * F; i += int64_t(gridDim.x) * blockDim.x) {
    const int64_t f = i / 3;
    const int j = static_cast<int>(i - 3 * f);
    const int a = faces[3 * f + j], b = faces[3 * f + (j == 2 ? 0 : j + 1)];
    keys[i] = ag_edge_key(a, b);
  }
}

// sorted half-edge keys -> unique edges (every edge of the closed mesh appears exactly twice)
__global__ void ag_ico_unique_kernel(const uint64_t* __restrict__ sorted, int64_t Eu, uint64_t* __restrict__ uniq,
                                     int* __restrict__ error_flag) {
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < Eu; i += int64_t(gridDim.x) * blockDim.x) {
    const uint64_t k = sorted[2 * i];
    if (k != sorted[2 * i + 1]) *error_flag = 1;
    uniq[i] = k;
  }
}

__global__ void ag_ico_midpoints_kernel(const uint64_t* __restrict__ uniq, int64_t Eu, int64_t V,
                                        double* __restrict__ xyz) {
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < Eu; i += int64_t(gridDim.x) * blockDim.x) {
    const uint64_t k = uniq[i];
    const int64_t a = static_cast<uint32_t>(k), b = static_cast<uint32_t>(k >> 32);
    for (int c = 0; c < 3; ++c) xyz[3 * (V + i) + c] = (xyz[3 * a + c] + xyz[3 * b + c]) / 2.0;
  }
}

__device__ __forceinline__ int ag_find_edge(const uint64_t* __restrict__ uniq, int64_t Eu, uint64_t key) {
  int64_t lo = 0, hi = Eu;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (uniq[mid] < key) lo = mid + 1; else hi = mid;
  }
  return static_cast<int>(lo);
}

__global__ void ag_ico_subdivide_kernel(const int32_t* __restrict__ fin, int64_t F, const uint64_t* __restrict__ uniq,
                                        int64_t Eu, int V, int32_t* __restrict__ fout) {
  for (int64_t f = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; f < F; f += int64_t(gridDim.x) * blockDim.x) {
    const int a = fin[3 * f], b = fin[3 * f + 1], c = fin[3 * f + 2];
    const int m0 = V + ag_find_edge(uniq, Eu, ag_edge_key(a, b));
    const int m1 = V + ag_find_edge(uniq, Eu, ag_edge_key(b, c));
    const int m2 = V + ag_find_edge(uniq, Eu, ag_edge_key(c, a));
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

__global__ void ag_ico_project_kernel(double* __restrict__ xyz, int64_t V) {
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < V; i += int64_t(gridDim.x) * blockDim.x) {
    const double x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    const double s = sqrt(x * x + y * y + z * z);
    const double off = 1.0 - s;
    xyz[3 * i] = x + (x / s) * off;
    xyz[3 * i + 1] = y + (y / s) * off;
    xyz[3 * i + 2] = z + (z / s) * off;
  }
}

// generate/transforms.py:50-52
__global__ void ag_ico_latlon_kernel(const double* __restrict__ xyz, int64_t V, float2* __restrict__ latlon) {
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < V; i += int64_t(gridDim.x) * blockDim.x) {
    const double x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    const double lat = asin(z / (x * x + y * y + z * z));
    const double lon = atan2(y, x);
    latlon[i] = make_float2(static_cast<float>(lat), static_cast<float>(lon));
  }
}

// descending order of a float as ascending order of an unsigned key
__device__ __forceinline__ uint32_t ag_desc_key(float v) {
  v = v + 0.0f;  // -0.0 -> +0.0
  uint32_t u = __float_as_uint(v);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // ascending
  return ~u;                                        // descending
}

// Node order = latitude descending, ties longitude descending, ties vertex id descending, as two
// full-width 64-bit sorts (no reliance on partial-bit sorting):
//   pass 0 key = (descending-longitude key << 32) | (0xffffffff - id)
//   pass 1 key = (descending-latitude key  << 32) | position after pass 0
__global__ void ag_ico_order_keys0_kernel(const float2* __restrict__ latlon, int64_t V, uint64_t* __restrict__ keys) {
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < V; i += int64_t(gridDim.x) * blockDim.x) {
    const uint32_t id = static_cast<uint32_t>(i);
    keys[i] = (static_cast<uint64_t>(ag_desc_key(latlon[i].y)) << 32) | (0xffffffffu - id);
  }
}

__global__ void ag_ico_order_keys1_kernel(const float2* __restrict__ latlon, int64_t V,
                                          const uint64_t* __restrict__ sorted0, uint64_t* __restrict__ keys) {
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < V; i += int64_t(gridDim.x) * blockDim.x) {
    const uint32_t id = 0xffffffffu - static_cast<uint32_t>(sorted0[i]);
    keys[i] = (static_cast<uint64_t>(ag_desc_key(latlon[id].x)) << 32) | static_cast<uint32_t>(i);
  }
}

__global__ void ag_ico_finish_order_kernel(const uint64_t* __restrict__ sorted0, const uint64_t* __restrict__ sorted1,
                                           const float2* __restrict__ latlon, int64_t V,
                                           float2* __restrict__ out_latlon, int32_t* __restrict__ order,
                                           int32_t* __restrict__ rank) {
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < V; i += int64_t(gridDim.x) * blockDim.x) {
    const uint32_t pos0 = static_cast<uint32_t>(sorted1[i]);
    const uint32_t id = 0xffffffffu - static_cast<uint32_t>(sorted0[pos0]);
    order[i] = static_cast<int32_t>(id);
    rank[id] = static_cast<int32_t>(i);
    out_latlon[i] = latlon[id];
  }
}

// ---------------------------------------------------------------------------------------------
// host side: topology stepping shared by ag_icosphere and ag_multiscale_*
// ---------------------------------------------------------------------------------------------
struct AgTopoWs {
  uint64_t* keys_a;   // [3 F_R]
  uint64_t* keys_b;   // [3 F_R]
  uint64_t* uniq;     // [E_R]
  void* sort_temp;
  size_t sort_temp_bytes;
  int32_t* faces_a;   // [3 F_R]
  int32_t* faces_b;   // [3 F_R]
  int* error_flag;
};

static void ag_topo_carve(AgCarver& c, int R, bool two_face_buffers, AgTopoWs* w) {
  const int64_t F = ag_ico_F(R);
  AgTopoWs t;
  t.keys_a = c.take<uint64_t>(3 * F);
  t.keys_b = c.take<uint64_t>(3 * F);
  t.uniq = c.take<uint64_t>(ag_ico_E(R));
  t.sort_temp_bytes = ag_sort_u64_temp_bytes(3 * F);
  t.sort_temp = c.take<char>(t.sort_temp_bytes);
  t.faces_a = c.take<int32_t>(3 * F);
  t.faces_b = two_face_buffers ? c.take<int32_t>(3 * F) : nullptr;
  t.error_flag = c.take<int>(4);
  if (w) *w = t;
}

static int ag_vertex_bits(int l) {
  int bits = 1;
  while ((int64_t(1) << bits) <= ag_ico_V(l)) ++bits;
  return bits;
}

// unique edges of `faces` (level l) into `uniq`
static int ag_topo_unique_edges(const AgTopoWs& w, const int32_t* faces, int l, uint64_t* uniq, hipStream_t s) {
  const int64_t F = ag_ico_F(l), Eu = ag_ico_E(l);
  AG_KERNEL("ag_ico_edge_keys_kernel", ag_ico_edge_keys_kernel, dim3(ag_blocks(3 * F, 256, 8192)), dim3(256), s, faces,
            F, w.keys_a);
  AG_LAUNCH_CHECK();
  int rc = ag_sort_u64(w.sort_temp, w.sort_temp_bytes, w.keys_a, w.keys_b, 3 * F, 0, 32 + ag_vertex_bits(l), s);
  if (rc) return rc;
  AG_KERNEL("ag_ico_unique_kernel", ag_ico_unique_kernel, dim3(ag_blocks(Eu, 256, 8192)), dim3(256), s, w.keys_b, Eu,
            uniq, w.error_flag);
  AG_LAUNCH_CHECK();
  return AG_OK;
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
  ag_topo_carve(c, R, false, nullptr);
  (void)c.take<float2>(ag_ico_V(R));
  return c.total();
}

extern "C" int ag_icosphere(int R, double* d_xyz, int32_t* d_faces, float* d_latlon, int32_t* d_order, int32_t* d_rank,
                            uint64_t* d_level_edges, void* d_ws, size_t ws_bytes, void* stream) {
  AG_REQUIRE(R >= 0 && R <= AG_ICO_MAX_LEVEL, "ag_icosphere: resolution %d outside [0, %d]", R, AG_ICO_MAX_LEVEL);
  AG_REQUIRE(d_xyz && d_faces && d_latlon && d_order && d_rank && d_ws, "ag_icosphere: null pointer");
  if (ws_bytes < ag_icosphere_workspace_bytes(R)) return ag_fail(AG_ERR_WORKSPACE, "ag_icosphere: workspace too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  AgCarver c(d_ws);
  AgTopoWs w;
  ag_topo_carve(c, R, false, &w);
  float2* latlon = c.take<float2>(ag_ico_V(R));
  // ping-pong between the caller's face buffer and the workspace so that level R lands in d_faces
  int32_t* cur = (R % 2 == 0) ? d_faces : w.faces_a;
  int32_t* nxt = (R % 2 == 0) ? w.faces_a : d_faces;
  int rc = ag_zero_i32(w.error_flag, 1, s);
  if (rc) return rc;
  AG_KERNEL("ag_ico_init_kernel", ag_ico_init_kernel, dim3(1), dim3(64), s, d_xyz, cur);
  AG_LAUNCH_CHECK();
  for (int l = 0; l < R; ++l) {
    const int64_t F = ag_ico_F(l), Eu = ag_ico_E(l), V = ag_ico_V(l);
    // the unique edges of level l go straight into the per-level edge array when the caller wants it
    uint64_t* uniq = d_level_edges ? d_level_edges + ag_ico_edge_offset(l) : w.uniq;
    rc = ag_topo_unique_edges(w, cur, l, uniq, s);
    if (rc) return rc;
    AG_KERNEL("ag_ico_midpoints_kernel", ag_ico_midpoints_kernel, dim3(ag_blocks(Eu, 256, 8192)), dim3(256), s, uniq,
              Eu, V, d_xyz);
    AG_LAUNCH_CHECK();
    AG_KERNEL("ag_ico_subdivide_kernel", ag_ico_subdivide_kernel, dim3(ag_blocks(F, 256, 8192)), dim3(256), s, cur, F,
              uniq, Eu, static_cast<int>(V), nxt);
    AG_LAUNCH_CHECK();
    AG_KERNEL("ag_ico_project_kernel", ag_ico_project_kernel, dim3(ag_blocks(V + Eu, 256, 8192)), dim3(256), s, d_xyz,
              V + Eu);
    AG_LAUNCH_CHECK();
    if (d_level_edges && l == R - 1) {
      AG_KERNEL("ag_ico_final_edges_kernel", ag_ico_final_edges_kernel, dim3(ag_blocks(Eu + F, 256, 8192)), dim3(256),
                s, uniq, Eu, V, nxt, F, d_level_edges + ag_ico_edge_offset(R));
      AG_LAUNCH_CHECK();
    }
    int32_t* t = cur;
    cur = nxt;
    nxt = t;
  }
  if (d_level_edges && R == 0) {
    rc = ag_topo_unique_edges(w, cur, 0, d_level_edges, s);
    if (rc) return rc;
  }
  const int64_t V = ag_ico_V(R);
  const int vb = ag_blocks(V, 256, 8192);
  AG_KERNEL("ag_ico_latlon_kernel", ag_ico_latlon_kernel, dim3(vb), dim3(256), s, d_xyz, V, latlon);
  AG_LAUNCH_CHECK();
  // node order: latitude descending, ties longitude descending, ties vertex id descending
  uint64_t* sorted0 = w.uniq;  // E_R >= V_R entries
  AG_KERNEL("ag_ico_order_keys0_kernel", ag_ico_order_keys0_kernel, dim3(vb), dim3(256), s, latlon, V, w.keys_a);
  AG_LAUNCH_CHECK();
  rc = ag_sort_u64(w.sort_temp, w.sort_temp_bytes, w.keys_a, sorted0, V, 0, 64, s);
  if (rc) return rc;
  AG_KERNEL("ag_ico_order_keys1_kernel", ag_ico_order_keys1_kernel, dim3(vb), dim3(256), s, latlon, V, sorted0,
            w.keys_a);
  AG_LAUNCH_CHECK();
  rc = ag_sort_u64(w.sort_temp, w.sort_temp_bytes, w.keys_a, w.keys_b, V, 0, 64, s);
  if (rc) return rc;
  AG_KERNEL("ag_ico_finish_order_kernel", ag_ico_finish_order_kernel, dim3(vb), dim3(256), s, sorted0, w.keys_b, latlon,
            V, reinterpret_cast<float2*>(d_latlon), d_order, d_rank);
  AG_LAUNCH_CHECK();
  return AG_OK;
}

// ---------------------------------------------------------------------------------------------
// multi-scale edges
// ---------------------------------------------------------------------------------------------
#define AG_NO_PAIR 0xffffffffffffffffull

static inline int ag_hops_capacity(int x_hops) { return 3 * x_hops * (x_hops + 1); }

__device__ __forceinline__ uint64_t ag_pair_key(int src_node, int tgt_node) {
  return (static_cast<uint64_t>(static_cast<uint32_t>(tgt_node)) << 32) | static_cast<uint32_t>(src_node);
}

// x_hops == 1: both directions of every (active) mesh edge, one fixed pair of slots per edge
__global__ void ag_ms_edge_pairs_kernel(const uint64_t* __restrict__ edges, int64_t Eu, const uint8_t* __restrict__ active,
                                        const int32_t* __restrict__ rank, uint64_t* __restrict__ pairs) {
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < Eu; i += int64_t(gridDim.x) * blockDim.x) {
    const uint64_t k = edges[i];
    const int a = static_cast<int>(static_cast<uint32_t>(k)), b = static_cast<int>(static_cast<uint32_t>(k >> 32));
    uint64_t p0 = AG_NO_PAIR, p1 = AG_NO_PAIR;
    if (active == nullptr || (active[a] && active[b])) {
      const int ra = rank[a], rb = rank[b];
      if (ra >= 0 && rb >= 0) {
        p0 = ag_pair_key(ra, rb);
        p1 = ag_pair_key(rb, ra);
      }
    }
    pairs[2 * i] = p0;
    pairs[2 * i + 1] = p1;
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
                                                         uint64_t* __restrict__ pairs) {
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
    uint64_t key = AG_NO_PAIR;
    if (j < n && rv >= 0) {
      const int ru = rank[list[j]];
      if (ru >= 0) key = ag_pair_key(rv, ru);
    }
    out[j] = key;
  }
}

// sorted pair keys -> 1 where a new distinct non-sentinel key starts
__global__ void ag_ms_flag_unique_kernel(const uint64_t* __restrict__ sorted, int64_t n, int64_t* __restrict__ flags) {
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
    const uint64_t k = sorted[i];
    flags[i] = (k != AG_NO_PAIR && (i == 0 || sorted[i - 1] != k)) ? 1 : 0;
  }
}

__global__ void ag_ms_copy_total_kernel(const int64_t* __restrict__ offsets, int64_t n, int64_t* __restrict__ out) {
  *out = offsets[n];
}

__global__ void ag_ms_emit_kernel(const uint64_t* __restrict__ sorted, const int64_t* __restrict__ offsets, int64_t n,
                                  int64_t num_edges, int32_t* __restrict__ out_src, int32_t* __restrict__ out_tgt) {
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
    const uint64_t k = sorted[i];
    if (k != AG_NO_PAIR && (i == 0 || sorted[i - 1] != k)) {
      const int64_t o = offsets[i];
      if (o < num_edges) {
        out_src[o] = static_cast<int32_t>(static_cast<uint32_t>(k));
        out_tgt[o] = static_cast<int32_t>(static_cast<uint32_t>(k >> 32));
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
  m.pairs_b = c.take<uint64_t>(slots);
  m.offsets = c.take<int64_t>(slots + 1);
  const size_t t1 = ag_sort_u64_temp_bytes(slots), t2 = ag_scan_i64_temp_bytes(slots);
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
static void ag_ms_launch_pairs(int64_t V, int x_hops, const AgMsWs& m, const int32_t* rank, uint64_t* pairs,
                               hipStream_t s) {
  AG_KERNEL("ag_ms_pairs_kernel", ag_ms_pairs_kernel<CAP>, dim3(ag_blocks(V, 64)), dim3(64), s, V, x_hops, m.deg, m.adj,
            rank, pairs);
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
  int64_t slot_base = 0;
  for (int i = 0; i < nlevels; ++i) {
    const int l = levels[i];
    const int64_t Eu = ag_ico_E(l), V = ag_ico_V(l);
    const uint64_t* edges = d_level_edges + ag_ico_edge_offset(l);
    uint64_t* pairs = m.pairs_a + slot_base;
    if (x_hops == 1) {
      AG_KERNEL("ag_ms_edge_pairs_kernel", ag_ms_edge_pairs_kernel, dim3(ag_blocks(Eu, 256, 8192)), dim3(256), s, edges,
                Eu, d_active, d_rank, pairs);
      AG_LAUNCH_CHECK();
    } else {
      rc = ag_zero_i32(m.deg, V, s);
      if (rc) return rc;
      AG_KERNEL("ag_ms_adjacency_kernel", ag_ms_adjacency_kernel, dim3(ag_blocks(Eu, 256, 8192)), dim3(256), s, edges,
                Eu, d_active, m.deg, m.adj);
      AG_LAUNCH_CHECK();
      if (cap <= 18) ag_ms_launch_pairs<18>(V, x_hops, m, d_rank, pairs, s);
      else if (cap <= 36) ag_ms_launch_pairs<36>(V, x_hops, m, d_rank, pairs, s);
      else if (cap <= 90) ag_ms_launch_pairs<90>(V, x_hops, m, d_rank, pairs, s);
      else ag_ms_launch_pairs<216>(V, x_hops, m, d_rank, pairs, s);
      AG_LAUNCH_CHECK();
    }
    slot_base += ag_ms_level_slots(l, x_hops);
  }
  const int64_t n = slot_base;
  rc = ag_sort_u64(m.temp, m.temp_bytes, m.pairs_a, m.pairs_b, n, 0, 64, s);
  if (rc) return rc;
  AG_KERNEL("ag_ms_flag_unique_kernel", ag_ms_flag_unique_kernel, dim3(ag_blocks(n, 256, 8192)), dim3(256), s, m.pairs_b,
            n, m.offsets);
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
  AG_KERNEL("ag_ms_emit_kernel", ag_ms_emit_kernel, dim3(ag_blocks(m.slots, 256, 8192)), dim3(256), s, m.pairs_b,
            m.offsets, m.slots, num_edges, d_out_src, d_out_tgt);
  AG_LAUNCH_CHECK();
  return AG_OK;
}
