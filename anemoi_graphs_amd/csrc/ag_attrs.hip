// Edge attributes (EdgeLength, EdgeDirection) and global normalisation for gfx950.
//
// One thread per edge: gather the two endpoints, evaluate the reference's formulas in the
// reference's precisions, store.  The kernel is bound by the random 8-byte gathers (36 B of
// compulsory traffic per edge) and by float64 arithmetic for the rotation.
//
// Precision contract
//  * The reference evaluates cos/sin of float32 coordinates with numpy's float32 kernels.  numpy's
//    SIMD float32 sin/cos (Cody-Waite reduction by pi/2 in three float32 pieces, degree-9/8
//    polynomials evaluated with fused multiply-adds) is restated here with the same constants
//    and the same operation order, so the float32 unit vectors are bit-identical to numpy's on
//    FMA-capable x86.  Arguments are latitudes/longitudes (|x| << 71476, the range of that path).
//  * float32 products, sums and square roots are single IEEE operations (the library is built
//    with -ffp-contract=off).  atan2 is evaluated in float64 and rounded once.
//  * the rotation follows scipy's Rotation.from_rotvec(...).apply(...) in float64: unit
//    quaternion -> matrix -> product (the quaternion is formed with half-angle identities).
#include "ag_common.hpp"

__device__ __forceinline__ float ag_np_sin_f32(float x) {
  float s, c;
  ag_np_sincos_f32(x, &s, &c);
  return s;
}

// utils.haversine_distance (utils.py:99-103) in float32
__device__ __forceinline__ float ag_edge_length(float2 sp, float2 tp) {
  const float dlat = tp.x - sp.x;
  const float dlon = tp.y - sp.y;
  const float s0 = ag_np_sin_f32(dlat / 2.0f);
  const float s1 = ag_np_sin_f32(dlon / 2.0f);
  float ss, cs, st, ct;
  ag_np_sincos_f32(sp.x, &ss, &cs);
  ag_np_sincos_f32(tp.x, &st, &ct);
  const float a = s0 * s0 + cs * ct * (s1 * s1);
  const float y = sqrtf(a), x = sqrtf(1.0f - a);
  return 2.0f * static_cast<float>(atan2(static_cast<double>(y), static_cast<double>(x)));
}

// generate/transforms.py:106-110 in float32
__device__ __forceinline__ void ag_xyz_f32(float2 p, float* x, float* y, float* z) {
  float sl, cl, so, co;
  ag_np_sincos_f32(p.x, &sl, &cl);
  ag_np_sincos_f32(p.y, &so, &co);
  *x = cl * co;
  *y = cl * so;
  *z = sl;
}

// edges/directional.py:40-65 with generate/transforms.py:133-140
__device__ __forceinline__ void ag_edge_direction(float2 sp, float2 tp, float* out_x, float* out_y) {
  const float eps32 = 10e-11f;
  const double eps = 10e-11;
  float sxf, syf, szf, txf, tyf, tzf;
  ag_xyz_f32(sp, &sxf, &syf, &szf);
  ag_xyz_f32(tp, &txf, &tyf, &tzf);
  // rotation taking the target to the pole: axis = target x pole, angle = acos(target . pole)
  double vx = static_cast<double>(tyf), vy = -static_cast<double>(txf);
  double n2 = vx * vx + vy * vy;
  if (n2 < eps) {  // target at a pole: the reference nudges the float32 point in place
    txf = txf + eps32;
    tyf = tyf + eps32;
    tzf = tzf + eps32;
    vx = static_cast<double>(tyf);
    vy = -static_cast<double>(txf);
    n2 = vx * vx + vy * vy;
  }
  // scipy builds the unit quaternion (axis * sin(theta/2), cos(theta/2)) from the rotation vector
  // axis * theta with theta = acos(z).  Since cos(theta) = z by definition, the half-angle identities
  // give the same quaternion without inverse trigonometry: sin(theta/2) = sqrt((1-z)/2),
  // cos(theta/2) = sqrt((1+z)/2); 1 -/+ z are exact in float64 for a float32 z.
  const double norm_v = sqrt(n2);
  const double z = static_cast<double>(tzf);
  const double sh = sqrt(0.5 * (1.0 - z)), ch = sqrt(0.5 * (1.0 + z));
  double qx = (vx / norm_v) * sh, qy = (vy / norm_v) * sh, qz = 0.0, qw = ch;
  const double qn = sqrt(qx * qx + qy * qy + qz * qz + qw * qw);
  qx /= qn;
  qy /= qn;
  qw /= qn;
  // scipy Rotation.as_matrix / apply
  const double x2 = qx * qx, y2 = qy * qy, z2 = qz * qz, w2 = qw * qw;
  const double xy = qx * qy, zw = qz * qw, xz = qx * qz, yw = qy * qw, yz = qy * qz, xw = qx * qw;
  const double m00 = x2 - y2 - z2 + w2, m01 = 2.0 * (xy - zw), m02 = 2.0 * (xz + yw);
  const double m10 = 2.0 * (xy + zw), m11 = -x2 + y2 - z2 + w2, m12 = 2.0 * (yz - xw);
  const double sx = sxf, sy = syf, sz = szf;
  double px = m00 * sx + m01 * sy + m02 * sz;
  double py = m10 * sx + m11 * sy + m12 * sz;
  // direction of the rotated source with respect to the pole
  double dx = py, dy = -px;
  double m2 = dx * dx + dy * dy;
  if (m2 < eps) {
    px += eps;
    py += eps;
    dx = py;
    dy = -px;
    m2 = dx * dx + dy * dy;
  }
  const double m = sqrt(m2);
  dx /= m;
  dy /= m;
  const double renorm = sqrt(dx * dx + dy * dy);
  *out_x = static_cast<float>(dx / renorm);
  *out_y = static_cast<float>(dy / renorm);
}

__global__ __launch_bounds__(256) void ag_edge_attrs_kernel(const float2* __restrict__ src_x,
                                                            const float2* __restrict__ tgt_x,
                                                            const int32_t* __restrict__ es,
                                                            const int32_t* __restrict__ et, int64_t E, int flags,
                                                            float* __restrict__ out_len, float2* __restrict__ out_dir) {
  for (int64_t e = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; e < E; e += int64_t(gridDim.x) * blockDim.x) {
    const float2 sp = src_x[es[e]];
    const float2 tp = tgt_x[et[e]];
    if (out_len != nullptr) out_len[e] = ag_edge_length(sp, tp);
    if (out_dir != nullptr) {
      float dx, dy;
      if (flags & AG_ATTR_ROTATED) {
        ag_edge_direction(sp, tp, &dx, &dy);
      } else {
        dx = tp.x - sp.x;
        dy = tp.y - sp.y;
      }
      out_dir[e] = make_float2(dx, dy);
    }
  }
}

extern "C" int ag_edge_attrs(const float* d_src_x, int64_t ns, const float* d_tgt_x, int64_t nt,
                             const int32_t* d_edge_src, const int32_t* d_edge_tgt, int64_t E, int flags, float* d_len,
                             float* d_dir, void* stream) {
  AG_REQUIRE(E >= 0 && ns >= 0 && nt >= 0, "ag_edge_attrs: negative size");
  if (E == 0 || (d_len == nullptr && d_dir == nullptr)) return AG_OK;
  AG_REQUIRE(d_src_x && d_tgt_x && d_edge_src && d_edge_tgt, "ag_edge_attrs: null pointer");
  AG_KERNEL("ag_edge_attrs_kernel", ag_edge_attrs_kernel, dim3(ag_blocks(E, 256, 1 << 20)), dim3(256), static_cast<hipStream_t>(stream), reinterpret_cast<const float2*>(d_src_x), reinterpret_cast<const float2*>(d_tgt_x), d_edge_src, d_edge_tgt, E, flags, d_len, reinterpret_cast<float2*>(d_dir));
  AG_LAUNCH_CHECK();
  return AG_OK;
}

// ---------------------------------------------------------------------------------------------
// global normalisation
// ---------------------------------------------------------------------------------------------
#define AG_RED_BLOCKS 1024

struct AgStats {
  double sum, sumsq, vmin, vmax, csumsq;
};

__device__ __forceinline__ double ag_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double ag_wave_min(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double ag_wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}

// pass 0: sum, sum of squares, min, max.  pass 1: sum of squared deviations from the mean.
__global__ __launch_bounds__(256) void ag_reduce_partial_kernel(const float* __restrict__ v, int64_t n, int pass,
                                                                const AgStats* __restrict__ stats,
                                                                AgStats* __restrict__ partial) {
  __shared__ AgStats sh[4];
  double sum = 0.0, sumsq = 0.0, vmin = 1.0e300, vmax = -1.0e300, csumsq = 0.0;
  const double mean = pass ? stats->sum / static_cast<double>(n) : 0.0;
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
    const double x = v[i];
    if (pass == 0) {
      sum += x;
      sumsq += x * x;
      vmin = fmin(vmin, x);
      vmax = fmax(vmax, x);
    } else {
      const double d = x - mean;
      csumsq += d * d;
    }
  }
  sum = ag_wave_sum(sum);
  sumsq = ag_wave_sum(sumsq);
  csumsq = ag_wave_sum(csumsq);
  vmin = ag_wave_min(vmin);
  vmax = ag_wave_max(vmax);
  const int wave = threadIdx.x >> 6;
  if (ag_lane() == 0) sh[wave] = {sum, sumsq, vmin, vmax, csumsq};
  __syncthreads();
  if (threadIdx.x == 0) {
    AgStats r = sh[0];
    for (int w = 1; w < 4; ++w) {
      r.sum += sh[w].sum;
      r.sumsq += sh[w].sumsq;
      r.csumsq += sh[w].csumsq;
      r.vmin = fmin(r.vmin, sh[w].vmin);
      r.vmax = fmax(r.vmax, sh[w].vmax);
    }
    partial[blockIdx.x] = r;
  }
}

__global__ __launch_bounds__(256) void ag_reduce_final_kernel(const AgStats* __restrict__ partial, int nblocks, int pass,
                                                              AgStats* __restrict__ stats) {
  __shared__ AgStats sh[4];
  double sum = 0.0, sumsq = 0.0, vmin = 1.0e300, vmax = -1.0e300, csumsq = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += blockDim.x) {
    const AgStats p = partial[i];
    sum += p.sum;
    sumsq += p.sumsq;
    csumsq += p.csumsq;
    vmin = fmin(vmin, p.vmin);
    vmax = fmax(vmax, p.vmax);
  }
  sum = ag_wave_sum(sum);
  sumsq = ag_wave_sum(sumsq);
  csumsq = ag_wave_sum(csumsq);
  vmin = ag_wave_min(vmin);
  vmax = ag_wave_max(vmax);
  const int wave = threadIdx.x >> 6;
  if (ag_lane() == 0) sh[wave] = {sum, sumsq, vmin, vmax, csumsq};
  __syncthreads();
  if (threadIdx.x == 0) {
    AgStats r = sh[0];
    for (int w = 1; w < 4; ++w) {
      r.sum += sh[w].sum;
      r.sumsq += sh[w].sumsq;
      r.csumsq += sh[w].csumsq;
      r.vmin = fmin(r.vmin, sh[w].vmin);
      r.vmax = fmax(r.vmax, sh[w].vmax);
    }
    if (pass == 0) {
      *stats = r;
    } else {
      stats->csumsq = r.csumsq;
    }
  }
}

__global__ __launch_bounds__(256) void ag_normalise_apply_kernel(float* __restrict__ v, int64_t n, int norm, int invert,
                                                                 const AgStats* __restrict__ stats) {
  float scale = 1.0f, shift = 0.0f;
  bool skip = false;
  if (norm == 1) scale = static_cast<float>(stats->sum);
  else if (norm == 2) scale = static_cast<float>(sqrt(stats->sumsq));
  else if (norm == 3) scale = static_cast<float>(stats->vmax);
  else if (norm == 4) {
    shift = static_cast<float>(stats->vmin);
    scale = static_cast<float>(stats->vmax) - shift;
  } else if (norm == 5) {
    scale = static_cast<float>(sqrt(stats->csumsq / static_cast<double>(n)));
    skip = (scale == 0.0f);  // the reference leaves the values untouched when the std is 0
  } else {
    skip = true;
  }
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
    float x = v[i];
    if (!skip) x = (x - shift) / scale;
    if (invert) x = 1.0f - x;
    v[i] = x;
  }
}

extern "C" size_t ag_normalise_workspace_bytes(int64_t n) {
  (void)n;
  return (AG_RED_BLOCKS + 2) * sizeof(AgStats) + 512;
}

extern "C" int ag_normalise(float* d_values, int64_t n, int norm, int invert, void* d_ws, size_t ws_bytes,
                            void* stream) {
  AG_REQUIRE(norm >= 0 && norm <= 5, "ag_normalise: unknown normalisation %d", norm);
  AG_REQUIRE(n >= 0, "ag_normalise: negative size");
  if (n == 0 || (norm == 0 && !invert)) return AG_OK;
  AG_REQUIRE(d_values && d_ws, "ag_normalise: null pointer");
  if (ws_bytes < ag_normalise_workspace_bytes(n)) return ag_fail(AG_ERR_WORKSPACE, "ag_normalise: workspace too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  AgCarver c(d_ws);
  AgStats* stats = c.take<AgStats>(1);
  AgStats* partial = c.take<AgStats>(AG_RED_BLOCKS);
  const int nblocks = ag_blocks(n, 256 * 8, AG_RED_BLOCKS);
  if (norm != 0) {
    AG_KERNEL("ag_reduce_partial_kernel", ag_reduce_partial_kernel, dim3(nblocks), dim3(256), s, d_values, n, 0, stats, partial);
    AG_LAUNCH_CHECK();
    AG_KERNEL("ag_reduce_final_kernel", ag_reduce_final_kernel, dim3(1), dim3(256), s, partial, nblocks, 0, stats);
    AG_LAUNCH_CHECK();
    if (norm == 5) {
      AG_KERNEL("ag_reduce_partial_kernel", ag_reduce_partial_kernel, dim3(nblocks), dim3(256), s, d_values, n, 1, stats, partial);
      AG_LAUNCH_CHECK();
      AG_KERNEL("ag_reduce_final_kernel", ag_reduce_final_kernel, dim3(1), dim3(256), s, partial, nblocks, 1, stats);
      AG_LAUNCH_CHECK();
    }
  }
  AG_KERNEL("ag_normalise_apply_kernel", ag_normalise_apply_kernel, dim3(ag_blocks(n, 256, 1 << 16)), dim3(256), s, d_values, n, norm, invert, stats);
  AG_LAUNCH_CHECK();
  return AG_OK;
}
