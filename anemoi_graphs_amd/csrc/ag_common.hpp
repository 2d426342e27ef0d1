// Shared host/device helpers of libanemoi_graphs_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <string>

#include "../../include/anemoi_graphs_hip.h"

#define AG_PI 3.14159265358979323846
#define AG_TWO_PI 6.28318530717958647692
#define AG_HALF_PI 1.57079632679489661923
#define AG_INV_TWO_PI 0.15915494309189533577

// ---------------------------------------------------------------------------------------------
// error reporting
// ---------------------------------------------------------------------------------------------
std::string& ag_error_slot();
int ag_fail(int code, const char* fmt, ...);

#define AG_HIP(call)                                                                                  \
  do {                                                                                                \
    hipError_t e__ = (call);                                                                          \
    if (e__ != hipSuccess) return ag_fail(AG_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e__)); \
  } while (0)

#define AG_REQUIRE(cond, ...)                            \
  do {                                                   \
    if (!(cond)) return ag_fail(AG_ERR_INVALID, __VA_ARGS__); \
  } while (0)

#define AG_LAUNCH_CHECK() AG_HIP(hipGetLastError())

// ---------------------------------------------------------------------------------------------
// optional per-kernel timing with HIP events recorded on the launch stream (off by default)
// ---------------------------------------------------------------------------------------------
void ag_prof_begin(const char* name, hipStream_t stream);
void ag_prof_end(hipStream_t stream);
struct AgProfScope {
  hipStream_t s;
  AgProfScope(const char* name, hipStream_t stream) : s(stream) { ag_prof_begin(name, s); }
  ~AgProfScope() { ag_prof_end(s); }
};

#define AG_KERNEL(name, kernel, grid, block, stream, ...)              \
  do {                                                                 \
    AgProfScope ag_prof_scope__(name, stream);                         \
    hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__);   \
  } while (0)

// ---------------------------------------------------------------------------------------------
// workspace carving: the same code sizes the workspace (base == nullptr) and hands out pointers
// ---------------------------------------------------------------------------------------------
struct AgCarver {
  char* base;
  size_t used = 0;
  explicit AgCarver(void* p) : base(static_cast<char*>(p)) {}
  template <class T>
  T* take(size_t count) {
    used = (used + 255) & ~size_t(255);
    T* p = base ? reinterpret_cast<T*>(base + used) : nullptr;
    used += count * sizeof(T);
    return p;
  }
  size_t total() const { return (used + 255) & ~size_t(255); }
};

static inline int ag_blocks(int64_t n, int per_block, int64_t cap = (1 << 30)) {
  int64_t b = (n + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return static_cast<int>(b);
}

// ---------------------------------------------------------------------------------------------
// the binning grid: latitude bands of equal height, each cut into an integer number of
// longitude cells of roughly the band height (in great-circle terms); the polar bands
// degenerate to one cell.  Cells are numbered band-major, so the cells of one band that cover a
// longitude interval are consecutive and their points are consecutive in the binned array.
// ---------------------------------------------------------------------------------------------
struct AgGrid {
  int nb;             // number of latitude bands
  double dlat;        // pi / nb
  double inv_dlat;    // nb / pi
  const int* ncols;   // [nb]    cells in band b           (device)
  const int* off;     // [nb+1]  first cell id of band b   (device); off[nb] = number of cells
  const double* win;  // [nb]    longitude half-width that covers the angular radius 0.98*dlat around
                      //         any point of band b; negative = all longitudes   (device)
};

#ifdef __HIPCC__

__device__ __forceinline__ double ag_wrap_lon(double lon) {
  double t = lon - AG_TWO_PI * floor(lon * AG_INV_TWO_PI);
  if (t < 0.0) t += AG_TWO_PI;
  if (t >= AG_TWO_PI) t -= AG_TWO_PI;
  if (t < 0.0) t = 0.0;
  return t;
}

__device__ __forceinline__ int ag_band_of(const AgGrid& g, double lat) {
  int b = static_cast<int>(floor((AG_HALF_PI - lat) * g.inv_dlat));
  return min(max(b, 0), g.nb - 1);
}

__device__ __forceinline__ int ag_cell_of(const AgGrid& g, double lat, double lon) {
  const int b = ag_band_of(g, lat);
  const int nc = g.ncols[b];
  int c = static_cast<int>(ag_wrap_lon(lon) * (nc * AG_INV_TWO_PI));
  c = min(max(c, 0), nc - 1);
  return g.off[b] + c;
}

// float64 sine and cosine for coordinate-sized arguments (|x| < 1e6 rad; latitudes, longitudes and
// their half differences), < 1 ulp: two-constant Cody-Waite reduction by pi/2 with fused
// multiply-adds, then the fdlibm kernels (__kernel_sin / __kernel_cos, 13th / 14th degree) on
// [-pi/4, pi/4] with the reduction's tail.  Unlike the general library routines there is no
// large-argument path, which keeps the register footprint of the calling kernels small.
__device__ __forceinline__ void ag_sincos_f64(double x, double* s_out, double* c_out) {
  const double k = rint(x * 0x1.45f306dc9c883p-1);  // x * 2/pi
  const double r0 = __builtin_fma(-k, 0x1.921fb54442d18p+0, x);
  const double r = __builtin_fma(-k, 0x1.1a62633145c07p-54, r0);
  const double tail = __builtin_fma(-k, 0x1.1a62633145c07p-54, r0 - r);  // (r0 - r) - k*P2
  const double z = r * r;
  // sine kernel
  const double v = z * r;
  double ps = __builtin_fma(z, 0x1.5d93a5acfd57cp-33, -0x1.ae5e68a2b9cebp-26);
  ps = __builtin_fma(z, ps, 0x1.71de357b1fe7dp-19);
  ps = __builtin_fma(z, ps, -0x1.a01a019c161d5p-13);
  ps = __builtin_fma(z, ps, 0x1.111111110f8a6p-7);
  const double sn = r - ((z * (0.5 * tail - v * ps) - tail) - v * -0x1.5555555555549p-3);
  // cosine kernel
  double pc = __builtin_fma(z, -0x1.8fae9be8838d4p-37, 0x1.1ee9ebdb4b1c4p-29);
  pc = __builtin_fma(z, pc, -0x1.27e4f809c52adp-22);
  pc = __builtin_fma(z, pc, 0x1.a01a019cb159p-16);
  pc = __builtin_fma(z, pc, -0x1.6c16c16c15177p-10);
  pc = __builtin_fma(z, pc, 0x1.555555555554cp-5);
  const double hz = 0.5 * z;
  const double w = 1.0 - hz;
  const double cs = w + (((1.0 - w) - hz) + (z * (z * pc) - r * tail));
  const int q = static_cast<int>(k) & 3;
  const double s = (q & 1) ? cs : sn;
  const double c = (q & 1) ? sn : cs;
  *s_out = (q & 2) ? -s : s;
  *c_out = ((q + 1) & 2) ? -c : c;
}

__device__ __forceinline__ double ag_sin_f64(double x) {
  double s, c;
  ag_sincos_f64(x, &s, &c);
  return s;
}

__device__ __forceinline__ double ag_cos_f64(double x) {
  double s, c;
  ag_sincos_f64(x, &s, &c);
  return c;
}

// The float64 "reduced distance" scikit-learn ranks by (sklearn/metrics/_dist_metrics.pyx.tp,
// HaversineDistance.rdist): separate roundings, no fused multiply-add (built with -ffp-contract=off).
__device__ __forceinline__ double ag_rdist(double lat1, double lon1, double lat2, double lon2) {
  const double s0 = ag_sin_f64(0.5 * (lat1 - lat2));
  const double s1 = ag_sin_f64(0.5 * (lon1 - lon2));
  return (s0 * s0 + ag_cos_f64(lat1) * ag_cos_f64(lat2) * s1 * s1);
}

__device__ __forceinline__ double ag_rdist_to_dist(double r) { return 2.0 * asin(sqrt(r)); }

// numpy's SIMD float32 sin/cos restated (Cody-Waite reduction by pi/2 in three float32 pieces,
// degree-9 / degree-8 polynomials, fused multiply-adds, same constants and operation order): the
// results are bit-identical to np.sin / np.cos of float32 arrays on FMA-capable x86, maximum error
// 1.49 ulp.  Valid for |x| < 71476, far beyond any latitude / longitude.
__device__ __forceinline__ void ag_np_sincos_f32(float x, float* s_out, float* c_out) {
  const float two_over_pi = 0x1.45f306p-1f;
  const float magic = 0x1.8p+23f;
  const float qm = x * two_over_pi + magic;
  const float q = qm - magic;  // round to nearest integer
  float r = __fmaf_rn(q, -0x1.921fb0p+00f, x);
  r = __fmaf_rn(q, -0x1.5110b4p-22f, r);
  r = __fmaf_rn(q, -0x1.846988p-48f, r);
  const float r2 = r * r;
  float sp = __fmaf_rn(0x1.7d3bbcp-19f, r2, -0x1.a06bbap-13f);
  sp = __fmaf_rn(sp, r2, 0x1.11119ap-07f);
  sp = __fmaf_rn(sp, r2, -0x1.555556p-03f);
  sp = __fmaf_rn(sp, r2, 0.0f);
  sp = __fmaf_rn(sp, r, r);
  float cp = __fmaf_rn(0x1.98e616p-16f, r2, -0x1.6c06dcp-10f);
  cp = __fmaf_rn(cp, r2, 0x1.55553cp-05f);
  cp = __fmaf_rn(cp, r2, -0x1.000000p-01f);
  cp = __fmaf_rn(cp, r2, 1.0f);
  // round(q) sits in the low mantissa bits of q + magic; quadrant n: sin = (n odd ? cp : sp), negated
  // for n = 2, 3 (mod 4); cos is the same with n + 1.  (The negations are sign-bit flips: a value that
  // is negated here is never zero.)
  const unsigned n = __float_as_uint(qm);
  const bool odd = (n & 1u) != 0u;
  const float s = __uint_as_float(__float_as_uint(odd ? cp : sp) ^ ((n << 30) & 0x80000000u));
  const float c = __uint_as_float(__float_as_uint(odd ? sp : cp) ^ (((n + 1u) << 30) & 0x80000000u));
  *s_out = s;
  *c_out = c;
}

// float32 unit vector computed entirely in float32 (coordinate error <= 2.1e-7)
__device__ __forceinline__ float3 ag_unit_vec_f32(float lat, float lon) {
  float sl, cl, so, co;
  ag_np_sincos_f32(lat, &sl, &cl);
  ag_np_sincos_f32(lon, &so, &co);
  return make_float3(cl * co, cl * so, sl);
}

// float32 unit vector of a (lat, lon) point, rounded from float64
__device__ __forceinline__ float3 ag_unit_vec(float lat, float lon) {
  double sl, cl, so, co;
  ag_sincos_f64(static_cast<double>(lat), &sl, &cl);
  ag_sincos_f64(static_cast<double>(lon), &so, &co);
  return make_float3(static_cast<float>(cl * co), static_cast<float>(cl * so), static_cast<float>(sl));
}

// Upper bound of |fp32 chord^2 - true chord^2| when one unit vector comes from ag_unit_vec
// (coordinate error <= 3e-8) and the other from ag_unit_vec_f32 (<= 2.1e-7): per-coordinate
// difference error e <= 2.4e-7 + 6e-8 |d|, so |err| <= 2 sqrt(3) e sqrt(c2) + 3 e^2 + roundings.
// (The root is the hardware's, within one ulp and a tenth of the cost of the correctly rounded library root; the
// coefficient carries that ulp, and a subnormal c2, which the instruction may flush, is below the constant term.)
__device__ __forceinline__ float ag_chord2_err(float c2) {
  return 1.0000003e-6f * __builtin_amdgcn_sqrtf(c2) + 6.0e-7f * c2 + 4.0e-13f;
}
// The same bound for a chord^2 whose low 8 mantissa bits were replaced by a tag (the value read back
// with the tag cleared is at most 255 ulp = 3.05e-5 relative below the computed one).
__device__ __forceinline__ float ag_chord2_err_tagged(float c2) {
  return 1.0000003e-6f * __builtin_amdgcn_sqrtf(c2) + 3.2e-5f * c2 + 4.0e-13f;
}

// The window of a box of targets (or of a single point): the bands b0..b1 it covers and, unless `full` (every
// cell of those bands), the longitude interval [lo, hi] (unwrapped: it may leave [0, 2 pi] on either side).
struct AgPointWindow {
  int b0, b1;
  bool full;
  double lo, hi;
};

// Window of every point within angular distance R of any target in the box lat in [lat_lo, lat_hi], wrapped
// longitude in [lon_lo, lon_hi] (lon_lo <= lon_hi, both in [0, 2pi]).
__device__ __forceinline__ AgPointWindow ag_box_window(const AgGrid& g, double lat_lo, double lat_hi, double lon_lo,
                                                       double lon_hi, double R) {
  AgPointWindow w;
  const double pad = 1.0e-7 + 1.0e-6 * R;
  w.b0 = ag_band_of(g, lat_hi + R + pad);
  w.b1 = ag_band_of(g, lat_lo - R - pad);
  const double phit = fmax(fabs(lat_lo), fabs(lat_hi));
  w.full = (phit + R + pad >= AG_HALF_PI) || (R >= AG_HALF_PI);
  w.lo = 0.0;
  w.hi = 0.0;
  if (!w.full) {
    const double q = ag_sin_f64(R) / ag_cos_f64(phit);
    if (q >= 1.0) {
      w.full = true;
    } else {
      const double delta = asin(q) * (1.0 + 1.0e-6) + pad;
      w.lo = lon_lo - delta;
      w.hi = lon_hi + delta;
      if (w.hi - w.lo >= AG_TWO_PI) w.full = true;
    }
  }
  return w;
}

__device__ __forceinline__ AgPointWindow ag_point_window(const AgGrid& g, double lat, double lonw, double R) {
  return ag_box_window(g, lat, lat, lonw, lonw, R);
}

// Cells [c_first, c_last) of a band of nc cells that the window covers: its first run, or its second one (empty
// unless the longitude interval wraps).
__device__ __forceinline__ void ag_window_cells(const AgPointWindow& w, int nc, bool second, int* c_first, int* c_last) {
  // 32-bit cell numbers: a band has far fewer than 2^29 cells, and lo, hi lie within a few turns of [0, 2 pi].
  // (64-bit integers here cost an emulated conversion and an emulated remainder per run -- most of the window
  // kernels' instructions.)  The interval starts within one turn of [0, 2 pi) whenever the longitudes given are
  // wrapped, so the remainder is an addition or a subtraction; anything else takes the division.
  const double wd = nc * AG_INV_TWO_PI;
  const int ca = static_cast<int>(floor(w.lo * wd));
  const int cb = static_cast<int>(floor(w.hi * wd));
  const int len = cb - ca + 1;
  *c_first = 0;
  *c_last = 0;
  if (w.full || nc == 1 || len >= nc) {
    *c_last = second ? 0 : nc;
  } else {
    int cam = ca < 0 ? ca + nc : (ca >= nc ? ca - nc : ca);
    if (cam < 0 || cam >= nc) {
      cam = ca % nc;
      if (cam < 0) cam += nc;
    }
    const int end = cam + len;
    if (!second) {
      *c_first = cam;
      *c_last = min(end, nc);
    } else if (end > nc) {
      *c_last = end - nc;
    }
  }
}

// Enumerate the runs of binned points that can hold a point within angular distance R of any
// target in the box lat in [lat_lo, lat_hi], wrapped longitude in [lon_lo, lon_hi] (lon_lo <=
// lon_hi, both in [0, 2pi]).  f(first, last) receives half-open ranges into the binned array.
// Every value is identical across the wave when the arguments are.
template <class F>
__device__ __forceinline__ void ag_for_each_range(const AgGrid& g, const int* __restrict__ cell_start, double lat_lo,
                                                  double lat_hi, double lon_lo, double lon_hi, double R, F&& f) {
  const AgPointWindow w = ag_box_window(g, lat_lo, lat_hi, lon_lo, lon_hi, R);
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

__device__ __forceinline__ int ag_lane() { return threadIdx.x & 63; }

// The same enumeration with a run per lane: run j of the window w (two per covered band, the second one empty
// unless the longitude interval wraps; 2 (w.b1 - w.b0 + 1) in all) comes back as [first, last).  Lanes that ask
// for different runs have the dependent table look-ups of all their bands in flight together.
__device__ __forceinline__ void ag_window_run(const AgGrid& g, const int* __restrict__ cell_start, const AgPointWindow& w,
                                              int j, int* first, int* last) {
  *first = 0;
  *last = 0;
  if (j < 2 * (w.b1 - w.b0 + 1)) {
    const int b = w.b0 + (j >> 1);
    const int nc = g.ncols[b];
    const int base = g.off[b];
    int c_first, c_last;
    ag_window_cells(w, nc, (j & 1) != 0, &c_first, &c_last);
    if (c_last > c_first) {
      *first = cell_start[base + c_first];
      *last = cell_start[base + c_last];
    }
  }
}

#endif  // __HIPCC__

// ---------------------------------------------------------------------------------------------
// host-side declarations shared between translation units
// ---------------------------------------------------------------------------------------------
struct AgBinned {
  AgGrid grid;
  int64_t n = 0;
  int64_t ncells_ub = 0;        // host-side upper bound of the number of cells
  const float4* pts = nullptr;  // [n]  (x, y, z, bits of original id) in binned order (optional)
  const int* ids = nullptr;     // [n]  original id of every binned position
  const int* cell_of = nullptr; // [n]  cell of every point, by original id
  const int* cell_start = nullptr; // [ncells_ub + 1]
};

// Number of latitude bands for cells of about `delta` radians.
int ag_grid_bands(double delta);
// Bins n points into a grid of nb bands; carves everything it needs from `ws`, with the cell
// tables sized for nb_reserve (>= nb) bands so that the carve can be made independent of nb.
// out == nullptr: sizing pass only.
// `mult` multiplies the number of longitude cells of every band (finer cells for dense target sets).
int ag_bin_points(AgCarver& ws, const float* d_pts, int64_t n, int nb, int nb_reserve, int mult, bool want_pts,
                  hipStream_t stream, AgBinned* out);

// The status word of the current device (device address; nullptr if it cannot be had) and the values it takes.
int* ag_status_word();
#define AG_STATUS_SCAN_GAVE_UP 1  // a single-launch scan met state words that were not clear, or waited in vain
// Test hook (ag_debug_poison_scans): the kernels that clear the state words of a scan fill them with ones instead.
bool ag_debug_scans_poisoned();

// device-wide primitives (rocPRIM behind a plain interface)
size_t ag_sort_u64_temp_bytes(int64_t n);
int ag_sort_u64(void* d_temp, size_t temp_bytes, const uint64_t* d_in, uint64_t* d_out, int64_t n, int begin_bit,
                int end_bit, hipStream_t stream);
size_t ag_sort_pairs_u64_i32_temp_bytes(int64_t n);
int ag_sort_pairs_u64_i32(void* d_temp, size_t temp_bytes, const uint64_t* d_keys_in, uint64_t* d_keys_out,
                          const int32_t* d_vals_in, int32_t* d_vals_out, int64_t n, hipStream_t stream);
size_t ag_segmented_sort_i32_temp_bytes(int64_t segments);
int ag_segmented_sort_i32(void* d_temp, size_t temp_bytes, int32_t* d_a, int32_t* d_b, int64_t size, int64_t segments,
                          const uint32_t* d_begin, const uint32_t* d_end, int end_bit, bool* sorted_in_b,
                          hipStream_t stream);
size_t ag_scan_i32_temp_bytes(int64_t n);
// exclusive prefix sum of n int32 values, in place allowed
int ag_exclusive_scan_i32(void* d_temp, size_t temp_bytes, const int* d_in, int* d_out, int64_t n, hipStream_t stream);
int ag_zero_i32(int* d_ptr, int64_t n, hipStream_t stream);
// single-launch exclusive scans; d_state = *_state_words(n) 64-bit words the caller has zeroed
size_t ag_chained_scan_state_words_i32(int64_t n);
size_t ag_chained_scan_state_words_i64(int64_t n);
int ag_chained_scan_i32(const int* d_in, int* d_out, int64_t n, unsigned long long* d_state, hipStream_t stream,
                        long long* d_total = nullptr);
int ag_chained_scan_i64(const int64_t* d_in, int64_t* d_out, int64_t n, unsigned long long* d_state, int want_max,
                        hipStream_t stream);
size_t ag_scan_i64_temp_bytes(int64_t n);
// exclusive prefix sum of n int64 counts into d_out[0..n] (n+1 entries; d_out[n] = total); in-place allowed
int ag_exclusive_scan_i64(void* d_temp, size_t temp_bytes, const int64_t* d_in, int64_t* d_out, int64_t n,
                          hipStream_t stream);
