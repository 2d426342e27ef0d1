// Edge attributes (EdgeLength, EdgeDirection) and global normalisation for gfx950.
//
// The attribute kernel is bound by vector arithmetic (four float32 sin/cos evaluations, one
// float32 arctangent and the float64 rotation per edge), not by its 20 B/edge of streamed
// traffic plus the cached endpoint gathers, so the design removes work rather than bytes:
//  * everything that depends on the target node only (its unit vector, the quaternion that takes
//    it to the pole and the rotation matrix) is computed once per run of equal targets inside a
//    block's chunk of edges and staged in LDS.  Edge lists from the builders are grouped by
//    target, so a run is a target's whole row; an ungrouped list degrades to one run per edge,
//    never to a wrong result;
//  * float64 square roots use reciprocal square root seeds refined by two Newton steps; the
//    arctangent of the length is a 9-term float32 polynomial after one float32 division.
//
// Precision contract
//  * The reference evaluates cos/sin of float32 coordinates with numpy's float32 kernels.  numpy's
//    SIMD float32 sin/cos (Cody-Waite reduction by pi/2 in three float32 pieces, degree-9/8
//    polynomials evaluated with fused multiply-adds) is restated here with the same constants
//    and the same operation order, so the float32 unit vectors are bit-identical to numpy's on
//    FMA-capable x86.  Arguments are latitudes/longitudes (|x| << 71476, the range of that path).
//  * float32 products, sums and square roots are single IEEE operations (the library is built
//    with -ffp-contract=off).
//  * the rotation (scipy's Rotation.from_rotvec(...).apply(...): unit quaternion -> matrix ->
//    product) is evaluated in float64 to within a few float64 ulp and rounded to float32 once, so a
//    stored direction differs from the float32 rounding of the exact result in about one value per
//    10^7, by one float32 ulp.  The quaternion is formed with half-angle identities;
//    normalisations of vectors that are unit by construction to within one float64 ulp are not
//    repeated.
//  * the length's atan2 is evaluated in float32 to within 1.6 ulp of the exact value (the
//    reference's numpy float32 arctan2 is within 2 ulp; the two differ by at most 3.1e-7
//    relative, inside the 1e-6 the parity tests allow).
#include "ag_common.hpp"

#define AG_ATTR_CHUNK 128  // edges per wave iteration (two per lane)
// Blocks of a launch: twice what fits the chip at once (6 blocks of 4 waves per CU).  Every wave takes an equal share
// of the chunks, so a grid that does not divide by the resident blocks ends with a round of a few blocks: measured
// per launch of 10.5 million edges, 2048 blocks 77.4 us, 1536 77.2, 3072 74.6, 4096 74.1, 8192 75.1, 16384 79.3 --
// and from 4096 on the kernels of a second stream wait longer behind the queued blocks.
#define AG_ATTR_MAX_BLOCKS (256 * 12)

__device__ __forceinline__ float ag_np_sin_f32(float x) {
  float s, c;
  ag_np_sincos_f32(x, &s, &c);
  return s;
}

// The same sine for |x| < 0.78: there the quadrant of ag_np_sincos_f32 is 0 (|x| 2/pi < 0.4966 rounds to 0), its
// reduced argument is x itself and its result the sine polynomial -- the same bits, without the reduction, the
// cosine polynomial and the quadrant logic (also for a NaN and for either zero).
#define AG_NP_SIN_SMALL 0.78f
__device__ __forceinline__ float ag_np_sin_small_f32(float r) {
  const float r2 = r * r;
  float sp = __fmaf_rn(0x1.7d3bbcp-19f, r2, -0x1.a06bbap-13f);
  sp = __fmaf_rn(sp, r2, 0x1.11119ap-07f);
  sp = __fmaf_rn(sp, r2, -0x1.555556p-03f);
  sp = __fmaf_rn(sp, r2, 0.0f);
  return __fmaf_rn(sp, r, r);
}

// 1/sqrt(x) for a normal, positive x: hardware seed (2^-23) + two Newton steps
__device__ __forceinline__ double ag_rsqrt_f64(double x) {
  double y = __builtin_amdgcn_rsq(x);
  const double h = 0.5 * x;
  double e = __builtin_fma(-(h * y), y, 0.5);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-(h * y), y, 0.5);
  y = __builtin_fma(y, e, y);
  return y;
}

// sqrt(x) for x >= 0 (normal or zero)
__device__ __forceinline__ double ag_sqrt_f64(double x) { return x > 0.0 ? x * ag_rsqrt_f64(x) : 0.0; }

// atan2(y, x) for y, x >= 0, not both zero, entirely in float32: atan(t) = t + t s P(s), s = t^2, on t = min/max in
// [0, 1], P of degree 8 interpolating (atan(t)/t - 1)/s at 9 Chebyshev nodes (tools/fit_atan_f32.py); the quotient
// is a reciprocal seed with one Newton step and a residual correction.  Over the range of the haversine term the
// length 2 atan2(sqrt(a), sqrt(1 - a)) is within 1.6 float32 ulp (1.53e-7 relative) of the exact value -- numpy's own
// float32 arctan2, which the reference calls (utils.py:103), is within 2.5e-7 on x86, and the two differ by at most
// 3.1e-7 -- at a quarter of the issue slots of the float64 polynomial.  NaN if either argument is NaN.
__device__ __forceinline__ float ag_atan2_pos_f32(float y, float x) {
  const float hi = fmaxf(x, y), lo = fminf(x, y);
  float r = __builtin_amdgcn_rcpf(hi);
  r = __fmaf_rn(__fmaf_rn(-hi, r, 1.0f), r, r);
  float t = lo * r;
  t = __fmaf_rn(__fmaf_rn(-hi, t, lo), r, t);
  const float s = t * t;
  float p = -0x1.38de56p-9f;
  p = __fmaf_rn(p, s, 0x1.ba9f66p-7f);
  p = __fmaf_rn(p, s, -0x1.25dc12p-5f);
  p = __fmaf_rn(p, s, 0x1.0001c6p-4f);
  p = __fmaf_rn(p, s, -0x1.6295f8p-4f);
  p = __fmaf_rn(p, s, 0x1.c3f168p-4f);
  p = __fmaf_rn(p, s, -0x1.246cd2p-3f);
  p = __fmaf_rn(p, s, 0x1.99983cp-3f);
  p = __fmaf_rn(p, s, -0x1.555554p-2f);
  const float at = __fmaf_rn(t * s, p, t);
  float res = (y > x) ? ((0x1.921fb6p+0f - at) + -0x1.777a5cp-25f) : at;  // pi/2 in two float32 pieces
  if (x != x || y != y) res = x + y;
  return res;
}

// Correctly rounded float32 square root of the haversine terms (0 <= x <= 1, or x < 0 -> NaN): the
// hardware root is within one ulp, and the exact residuals of its two neighbours (one fused
// multiply-add each) tell which of the three is the rounded root.  Without the rescaling a library
// sqrtf carries for x < 2^-96 (an edge shorter than 4e-15 rad), where the residuals underflow and the
// result may be one ulp off.
__device__ __forceinline__ float ag_sqrt_f32_rn(float x) {
  const float y = __builtin_amdgcn_sqrtf(x);
  const float ym = __uint_as_float(__float_as_uint(y) - 1u), yp = __uint_as_float(__float_as_uint(y) + 1u);
  const float rm = __fmaf_rn(-ym, y, x), rp = __fmaf_rn(-yp, y, x);
  float r = rm <= 0.0f ? ym : y;
  r = rp > 0.0f ? yp : r;
  return r;
}

// Per-run (per distinct target) data staged in LDS
struct AgAttrRuns {
  double m00[AG_ATTR_CHUNK], m01[AG_ATTR_CHUNK], m02[AG_ATTR_CHUNK], m11[AG_ATTR_CHUNK], m12[AG_ATTR_CHUNK];
  float lat[AG_ATTR_CHUNK], lon[AG_ATTR_CHUNK], cos_lat[AG_ATTR_CHUNK];
};

// Target part of edges/directional.py:40-65 with generate/transforms.py:106-110,133-140: unit
// vector of the target in float32, rotation taking it to the pole as a matrix in float64.
__device__ __forceinline__ void ag_attr_target(float2 tp, bool rotated, AgAttrRuns& R, int r) {
  float st, ct, so, co;
  ag_np_sincos_f32(tp.x, &st, &ct);
  R.lat[r] = tp.x;
  R.lon[r] = tp.y;
  R.cos_lat[r] = ct;
  if (!rotated) return;
  ag_np_sincos_f32(tp.y, &so, &co);
  float txf = ct * co, tyf = ct * so, tzf = st;
  const float eps32 = 10e-11f;
  const double eps = 10e-11;
  // axis = target x pole, angle = acos(target . pole)
  double vx = static_cast<double>(tyf), vy = -static_cast<double>(txf);
  double n2 = __builtin_fma(vx, vx, vy * vy);
  if (n2 < eps) {  // target at a pole: the reference nudges the float32 point in place
    txf = txf + eps32;
    tyf = tyf + eps32;
    tzf = tzf + eps32;
    vx = static_cast<double>(tyf);
    vy = -static_cast<double>(txf);
    n2 = __builtin_fma(vx, vx, vy * vy);
  }
  // scipy builds the unit quaternion q = (u sin(theta/2), cos(theta/2)), u = axis / |axis|, from the
  // rotation vector u theta with theta = acos(z), and from it the matrix.  With qz = 0 and the
  // half-angle identities sin^2(theta/2) = (1-z)/2, cos^2(theta/2) = (1+z)/2, 2 sin cos = sin(theta)
  // the matrix rows 0 and 1 need one reciprocal root (|axis|) and one root (sin(theta)), no inverse
  // trigonometry; 1 -/+ z are exact in float64 for a float32 z.
  const double inv_norm = ag_rsqrt_f64(n2);
  const double ux = vx * inv_norm, uy = vy * inv_norm;
  const double z = static_cast<double>(tzf);
  const double omz = 1.0 - z, opz = 1.0 + z;
  const double sin_t = ag_sqrt_f64(omz * opz);
  const double a = 0.5 * omz, b = 0.5 * opz;
  const double d = a * __builtin_fma(ux, ux, -(uy * uy));
  R.m00[r] = b + d;
  R.m11[r] = b - d;
  R.m01[r] = (ux * uy) * omz;
  R.m02[r] = uy * sin_t;
  R.m12[r] = -(ux * sin_t);
}

// One edge: utils.haversine_distance (utils.py:99-103) in float32 and the direction of the
// rotated source with respect to the pole (edges/directional.py:68-94).
__device__ __forceinline__ void ag_attr_edge(float2 sp, const AgAttrRuns& R, int r, bool rotated, bool want_len,
                                             bool want_dir, float* len, double2* dir) {
  const float tlat = R.lat[r], tlon = R.lon[r];
  float ss, cs;
  ag_np_sincos_f32(sp.x, &ss, &cs);
  if (want_len) {
    const float ct = R.cos_lat[r];
    const float dlat = tlat - sp.x;
    const float dlon = tlon - sp.y;
    // the half differences of the edges of a wave are nearly always all small (an edge is short unless it crosses
    // the date line): one wave-wide test picks the short form of the two sines
    const float h0 = dlat / 2.0f, h1 = dlon / 2.0f;
    float s0, s1;
    if (__builtin_amdgcn_ballot_w64(fmaxf(fabsf(h0), fabsf(h1)) >= AG_NP_SIN_SMALL) == 0ull) {
      s0 = ag_np_sin_small_f32(h0);
      s1 = ag_np_sin_small_f32(h1);
    } else {
      s0 = ag_np_sin_f32(h0);
      s1 = ag_np_sin_f32(h1);
    }
    const float a = s0 * s0 + cs * ct * (s1 * s1);
    const float y = ag_sqrt_f32_rn(a), x = ag_sqrt_f32_rn(1.0f - a);
    *len = 2.0f * ag_atan2_pos_f32(y, x);
  }
  if (want_dir) {
    if (!rotated) {
      *dir = make_double2(static_cast<double>(tlat - sp.x), static_cast<double>(tlon - sp.y));
      return;
    }
    float so, co;
    ag_np_sincos_f32(sp.y, &so, &co);
    const double sx = static_cast<double>(cs * co), sy = static_cast<double>(cs * so), sz = static_cast<double>(ss);
    const double m01 = R.m01[r];
    double px = __builtin_fma(R.m00[r], sx, __builtin_fma(m01, sy, R.m02[r] * sz));
    double py = __builtin_fma(m01, sx, __builtin_fma(R.m11[r], sy, R.m12[r] * sz));
    const double eps = 10e-11;
    double dx = py, dy = -px;
    double m2 = __builtin_fma(dx, dx, dy * dy);
    if (m2 < eps) {
      px += eps;
      py += eps;
      dx = py;
      dy = -px;
      m2 = __builtin_fma(dx, dx, dy * dy);
    }
    const double inv = ag_rsqrt_f64(m2);
    *dir = make_double2(dx * inv, dy * inv);
  }
}

// Orders the LDS traffic of one wave (its lanes run in lockstep, so no barrier is needed): global
// loads issued before it -- the prefetch of the next chunk, the endpoint gathers of this one -- stay
// in flight across it.
__device__ __forceinline__ void ag_lds_wave_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// One wave owns a chunk of AG_ATTR_CHUNK consecutive edges (two per lane) and a private slice of LDS
// for the runs of equal targets inside it, so waves never wait for each other.  The loop is
// software-pipelined so that no global-memory latency is exposed inside an iteration (see below).
__global__ __launch_bounds__(256) void ag_edge_attrs_kernel(const float2* __restrict__ src_x,
                                                            const float2* __restrict__ tgt_x,
                                                            const int32_t* __restrict__ es,
                                                            const int32_t* __restrict__ et, int64_t E, int flags,
                                                            float* __restrict__ out_len, float2* __restrict__ out_dir,
                                                            unsigned* __restrict__ len_range, int ns_last, int nt_last,
                                                            int* __restrict__ bad) {
  __shared__ AgAttrRuns R_all[4];
  const bool rotated = (flags & AG_ATTR_ROTATED) != 0 && out_dir != nullptr;
  const bool want_len = out_len != nullptr, want_dir = out_dir != nullptr;
  const bool dir_f64 = (flags & AG_ATTR_DIR_F64) != 0;  // out_dir holds float64 pairs
  const int lane = ag_lane();
  AgAttrRuns& R = R_all[__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)];
  const unsigned long long below = (1ull << lane) - 1ull;
  const int64_t nchunks = (E + AG_ATTR_CHUNK - 1) / AG_ATTR_CHUNK;
  const int64_t wave_global = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int64_t nwaves = (int64_t(gridDim.x) * blockDim.x) >> 6;
  // lane l owns edges base + l and base + 64 + l.  Three chunks are in flight per wave: the ids of
  // chunk c+2 are being fetched and the endpoints of chunk c+1 gathered while chunk c is evaluated.
  // Loads are never predicated: an index past the end of the list (or of the chunk sequence) is
  // clamped to the last edge, whose values are loaded again and ignored.
  struct Ids {
    int t[2], s[2], p[2];
  };
  struct Ends {
    float2 sp[2], tp[2];
    bool first[2];  // the edge starts a run of equal targets inside its chunk
  };
  auto fetch_ids = [&](int64_t chunk) {
    Ids r;
    const int64_t c = chunk < nchunks ? chunk : nchunks - 1;
    const int64_t base = c * AG_ATTR_CHUNK;
    const int64_t left = E - base;
    const int last = (left < AG_ATTR_CHUNK ? static_cast<int>(left) : AG_ATTR_CHUNK) - 1;
    const int32_t* __restrict__ etc = et + base;
    const int32_t* __restrict__ esc = es + base;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int o = min(j * 64 + lane, last);
      r.t[j] = etc[o];
      r.s[j] = esc[o];
      r.p[j] = etc[max(o - 1, 0)];
    }
    return r;
  };
  // An id outside its node set (the reference's numpy indexing raises IndexError for it,
  // edges/attributes.py:144-148) is never followed: the gather reads the nearest valid row instead, the
  // edge's values are meaningless and the caller's error word is set.
  bool bad_id = false;
  auto gather = [&](const Ids& ids) {
    Ends g;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int s = min(max(ids.s[j], 0), ns_last), t = min(max(ids.t[j], 0), nt_last);
      bad_id = bad_id || s != ids.s[j] || t != ids.t[j];
      g.sp[j] = src_x[s];
      g.tp[j] = tgt_x[t];
      g.first[j] = (j * 64 + lane == 0) || ids.p[j] != ids.t[j];
    }
    return g;
  };
  // running range of the lengths this lane has written, as bit patterns: lengths are >= +0 (or NaN),
  // so their bits order like the values, and NaN ranks above every number as numpy's max / min have it
  unsigned hi_bits = 0u, lo_inv_bits = 0u;
  auto process = [&](int64_t chunk, const Ends& g) {
    const int64_t base = chunk * AG_ATTR_CHUNK;
    const int64_t left = E - base;
    const int count = left < AG_ATTR_CHUNK ? static_cast<int>(left) : AG_ATTR_CHUNK;
    int run[2];
    bool valid[2], first[2];
    unsigned long long bal[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      valid[j] = j * 64 + lane < count;
      first[j] = valid[j] && g.first[j];
      bal[j] = __ballot(first[j]);
    }
    const int n0 = __popcll(bal[0]);
    const int nruns = n0 + __popcll(bal[1]);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      run[j] = (j ? n0 : 0) + __popcll(bal[j] & below) + (first[j] ? 1 : 0) - 1;
      if (first[j]) {
        R.lat[run[j]] = g.tp[j].x;
        R.lon[run[j]] = g.tp[j].y;
      }
    }
    ag_lds_wave_sync();
    for (int r = lane; r < nruns; r += 64) ag_attr_target(make_float2(R.lat[r], R.lon[r]), rotated, R, r);
    ag_lds_wave_sync();
    float* __restrict__ len_c = out_len + (want_len ? base : 0);
    float2* __restrict__ dir_c = out_dir + (want_dir ? base : 0);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (!valid[j]) continue;
      float len;
      double2 dir;
      ag_attr_edge(g.sp[j], R, run[j], rotated, want_len, want_dir, &len, &dir);
      if (want_len) {
        len_c[j * 64 + lane] = len;
        const unsigned b = __float_as_uint(len);
        hi_bits = max(hi_bits, b);
        lo_inv_bits = max(lo_inv_bits, (b > 0x7f800000u) ? 0u : ~b);  // (a NaN is seen through the maximum)
      }
      if (want_dir) {
        if (dir_f64) reinterpret_cast<double2*>(out_dir)[base + j * 64 + lane] = dir;
        else dir_c[j * 64 + lane] = make_float2(static_cast<float>(dir.x), static_cast<float>(dir.y));
      }
    }
    ag_lds_wave_sync();  // the staged runs are overwritten by the next chunk
  };
  if (wave_global >= nchunks) return;
  // two chunks per trip, so that the two sets of gathered endpoints alternate without register copies;
  // the ids are dead once their gather is issued and the next fetch reuses their registers
  Ids ids = fetch_ids(wave_global);
  Ends ga = gather(ids);
  ids = fetch_ids(wave_global + nwaves);
  for (int64_t chunk = wave_global; chunk < nchunks; chunk += 2 * nwaves) {
    const Ends gb = gather(ids);
    ids = fetch_ids(chunk + 2 * nwaves);
    process(chunk, ga);
    if (chunk + nwaves >= nchunks) break;
    ga = gather(ids);
    ids = fetch_ids(chunk + 3 * nwaves);
    process(chunk + nwaves, gb);
  }
  if (bad_id && bad != nullptr) *bad = 1;  // (every writer stores the same value)
  if (len_range != nullptr && want_len) {
    // every wave that had a chunk leaves its range in a slot of its own (atomics of thousands of waves on
    // one word are far slower than the kernel); ag_len_range_finish_kernel folds the slots
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      hi_bits = max(hi_bits, static_cast<unsigned>(__shfl_xor(static_cast<int>(hi_bits), o, 64)));
      lo_inv_bits = max(lo_inv_bits, static_cast<unsigned>(__shfl_xor(static_cast<int>(lo_inv_bits), o, 64)));
    }
    if (lane == 0) reinterpret_cast<uint2*>(len_range)[1 + wave_global] = make_uint2(hi_bits, lo_inv_bits);
  }
}

// slots 1 .. nslots of the waves -> words 0 and 1
__global__ __launch_bounds__(1024) void ag_len_range_finish_kernel(unsigned* __restrict__ range, int nslots) {
  __shared__ uint2 sh[16];
  unsigned hi = 0u, lo_inv = 0u;
  const uint2* slots = reinterpret_cast<const uint2*>(range) + 1;
  for (int i = threadIdx.x; i < nslots; i += blockDim.x) {
    const uint2 v = slots[i];
    hi = max(hi, v.x);
    lo_inv = max(lo_inv, v.y);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    hi = max(hi, static_cast<unsigned>(__shfl_xor(static_cast<int>(hi), o, 64)));
    lo_inv = max(lo_inv, static_cast<unsigned>(__shfl_xor(static_cast<int>(lo_inv), o, 64)));
  }
  if (ag_lane() == 0) sh[threadIdx.x >> 6] = make_uint2(hi, lo_inv);
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w) {
      hi = max(hi, sh[w].x);
      lo_inv = max(lo_inv, sh[w].y);
    }
    range[0] = hi;
    range[1] = lo_inv;
  }
}

extern "C" int ag_edge_attrs_range(const float* d_src_x, int64_t ns, const float* d_tgt_x, int64_t nt,
                                   const int32_t* d_edge_src, const int32_t* d_edge_tgt, int64_t E, int flags,
                                   float* d_len, float* d_dir, uint32_t* d_len_range, int32_t* d_bad, void* stream) {
  AG_REQUIRE(E >= 0 && ns >= 0 && nt >= 0, "ag_edge_attrs: negative size");
  AG_REQUIRE(ns < (int64_t(1) << 31) && nt < (int64_t(1) << 31), "ag_edge_attrs: node counts must fit int32");
  AG_REQUIRE(E == 0 || (ns > 0 && nt > 0), "ag_edge_attrs: %lld edges between node sets of %lld and %lld nodes",
             static_cast<long long>(E), static_cast<long long>(ns), static_cast<long long>(nt));
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool want_range = d_len_range != nullptr && d_len != nullptr;
  const bool empty = E == 0 || (d_len == nullptr && d_dir == nullptr);
  if (!empty) {
    AG_REQUIRE(d_src_x && d_tgt_x && d_edge_src && d_edge_tgt, "ag_edge_attrs: null pointer");
    const int blocks = ag_blocks(E, 4 * AG_ATTR_CHUNK, AG_ATTR_MAX_BLOCKS);
    AG_KERNEL("ag_edge_attrs_kernel", ag_edge_attrs_kernel, dim3(blocks), dim3(256), s, reinterpret_cast<const float2*>(d_src_x), reinterpret_cast<const float2*>(d_tgt_x), d_edge_src, d_edge_tgt, E, flags, d_len, reinterpret_cast<float2*>(d_dir), want_range ? d_len_range : nullptr, static_cast<int>(ns - 1), static_cast<int>(nt - 1), d_bad);
    AG_LAUNCH_CHECK();
  }
  if (d_len_range != nullptr) {
    // the waves that had a chunk: all of them, or one per chunk when the list is shorter than the grid
    const int64_t nchunks = (E + AG_ATTR_CHUNK - 1) / AG_ATTR_CHUNK;
    const int64_t nwaves = int64_t(ag_blocks(E, 4 * AG_ATTR_CHUNK, AG_ATTR_MAX_BLOCKS)) * 4;
    const int nslots = (empty || !want_range) ? 0 : static_cast<int>(nchunks < nwaves ? nchunks : nwaves);
    AG_KERNEL("ag_len_range_finish_kernel", ag_len_range_finish_kernel, dim3(1), dim3(1024), s, d_len_range, nslots);
    AG_LAUNCH_CHECK();
  }
  return AG_OK;
}

extern "C" size_t ag_len_range_words(void) { return 2 * (1 + size_t(AG_ATTR_MAX_BLOCKS) * 4); }

extern "C" int ag_edge_attrs(const float* d_src_x, int64_t ns, const float* d_tgt_x, int64_t nt,
                             const int32_t* d_edge_src, const int32_t* d_edge_tgt, int64_t E, int flags, float* d_len,
                             float* d_dir, int32_t* d_bad, void* stream) {
  return ag_edge_attrs_range(d_src_x, ns, d_tgt_x, nt, d_edge_src, d_edge_tgt, E, flags, d_len, d_dir, nullptr, d_bad,
                             stream);
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

// number of leading floats before the first 16-byte boundary of v (at most n)
__device__ __forceinline__ int64_t ag_align_head(const float* v, int64_t n) {
  const int64_t h = static_cast<int64_t>(((16 - (reinterpret_cast<uintptr_t>(v) & 15)) & 15) >> 2);
  return h < n ? h : n;
}

// pass 0: sum, sum of squares, min, max.  pass 1: sum of squared deviations from the mean.
// Four values per lane and iteration (16-byte loads); the tail is taken by the first threads.
__global__ __launch_bounds__(256) void ag_reduce_partial_kernel(const float* __restrict__ v, int64_t n, int pass,
                                                                const AgStats* __restrict__ stats,
                                                                AgStats* __restrict__ partial) {
  __shared__ AgStats sh[4];
  double sum = 0.0, sumsq = 0.0, csumsq = 0.0;
  float vmin = 3.0e38f, vmax = -3.0e38f;
  const double mean = pass ? stats->sum / static_cast<double>(n) : 0.0;
  auto take = [&](float xf) {
    const double x = xf;
    if (pass == 0) {
      sum += x;
      sumsq += x * x;
      vmin = fminf(vmin, xf);
      vmax = fmaxf(vmax, xf);
    } else {
      const double d = x - mean;
      csumsq += d * d;
    }
  };
  // elements before the first 16-byte boundary, 16-byte body, tail
  const int64_t head = ag_align_head(v, n);
  const int64_t n4 = (n - head) >> 2;
  const float4* v4 = reinterpret_cast<const float4*>(v + head);
  const int64_t tid = blockIdx.x * int64_t(blockDim.x) + threadIdx.x, nthreads = int64_t(gridDim.x) * blockDim.x;
  for (int64_t i = tid; i < n4; i += nthreads) {
    const float4 q = v4[i];
    take(q.x);
    take(q.y);
    take(q.z);
    take(q.w);
  }
  if (tid < head) take(v[tid]);
  if (tid < ((n - head) & 3)) take(v[head + 4 * n4 + tid]);
  sum = ag_wave_sum(sum);
  sumsq = ag_wave_sum(sumsq);
  csumsq = ag_wave_sum(csumsq);
  const double dmin = ag_wave_min(static_cast<double>(vmin)), dmax = ag_wave_max(static_cast<double>(vmax));
  const int wave = threadIdx.x >> 6;
  if (ag_lane() == 0) sh[wave] = {sum, sumsq, dmin, dmax, csumsq};
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
                                                                 const AgStats* __restrict__ stats,
                                                                 const unsigned* __restrict__ range) {
  float scale = 1.0f, shift = 0.0f;
  bool skip = false;
  if (range != nullptr) {  // maximum and minimum known beforehand (bit patterns, the minimum complemented)
    const float vmax = __uint_as_float(range[0]);
    const float vmin = (vmax != vmax) ? vmax : __uint_as_float(~range[1]);
    if (norm == 3) scale = vmax;
    else if (norm == 4) {
      shift = vmin;
      scale = vmax - vmin;
    } else skip = true;
  } else if (norm == 1) scale = static_cast<float>(stats->sum);
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
  auto apply = [&](float x) {
    if (!skip) x = (x - shift) / scale;
    if (invert) x = 1.0f - x;
    return x;
  };
  const int64_t head = ag_align_head(v, n);
  const int64_t n4 = (n - head) >> 2;
  float4* v4 = reinterpret_cast<float4*>(v + head);
  const int64_t tid = blockIdx.x * int64_t(blockDim.x) + threadIdx.x, nthreads = int64_t(gridDim.x) * blockDim.x;
  for (int64_t i = tid; i < n4; i += nthreads) {
    float4 q = v4[i];
    q.x = apply(q.x);
    q.y = apply(q.y);
    q.z = apply(q.z);
    q.w = apply(q.w);
    v4[i] = q;
  }
  if (tid < head) v[tid] = apply(v[tid]);
  if (tid < ((n - head) & 3)) v[head + 4 * n4 + tid] = apply(v[head + 4 * n4 + tid]);
}

// float64 values (node attributes): the same statistics, applied in float64 and rounded to float32
// once, as NormaliserMixin.normalise on a float64 array followed by astype(float32) does.
__global__ __launch_bounds__(256) void ag_reduce_partial_f64_kernel(const double* __restrict__ v, int64_t n, int pass,
                                                                    const AgStats* __restrict__ stats,
                                                                    AgStats* __restrict__ partial) {
  __shared__ AgStats sh[4];
  double sum = 0.0, sumsq = 0.0, csumsq = 0.0, vmin = 1.0e300, vmax = -1.0e300;
  const double mean = pass ? stats->sum / static_cast<double>(n) : 0.0;
  const int64_t tid = blockIdx.x * int64_t(blockDim.x) + threadIdx.x, nthreads = int64_t(gridDim.x) * blockDim.x;
  for (int64_t i = tid; i < n; i += nthreads) {
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

__global__ __launch_bounds__(256) void ag_normalise_apply_f64_kernel(const double* __restrict__ v, int64_t n, int norm,
                                                                     const AgStats* __restrict__ stats,
                                                                     float* __restrict__ out,
                                                                     double* __restrict__ out64) {
  double scale = 1.0, shift = 0.0;
  if (norm == 1) scale = stats->sum;
  else if (norm == 2) scale = sqrt(stats->sumsq);
  else if (norm == 3) scale = stats->vmax;
  else if (norm == 4) {
    shift = stats->vmin;
    scale = stats->vmax - shift;
  } else if (norm == 5) {
    scale = sqrt(stats->csumsq / static_cast<double>(n));
    if (scale == 0.0) scale = 1.0;  // the reference leaves the values untouched when the std is 0
  }
  const int64_t tid = blockIdx.x * int64_t(blockDim.x) + threadIdx.x, nthreads = int64_t(gridDim.x) * blockDim.x;
  for (int64_t i = tid; i < n; i += nthreads) {
    const double r = (v[i] - shift) / scale;
    if (out64 != nullptr) out64[i] = r;
    else out[i] = static_cast<float>(r);
  }
}

extern "C" int ag_normalise_f64(const double* d_values, int64_t n, int norm, void* d_out, int out_f64, void* d_ws,
                                size_t ws_bytes, void* stream) {
  AG_REQUIRE(norm >= 0 && norm <= 5, "ag_normalise_f64: unknown normalisation %d", norm);
  AG_REQUIRE(n >= 0, "ag_normalise_f64: negative size");
  if (n == 0) return AG_OK;
  AG_REQUIRE(d_values && d_out && d_ws, "ag_normalise_f64: null pointer");
  if (ws_bytes < (AG_RED_BLOCKS + 2) * sizeof(AgStats) + 512)
    return ag_fail(AG_ERR_WORKSPACE, "ag_normalise_f64: workspace too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  AgCarver c(d_ws);
  AgStats* stats = c.take<AgStats>(1);
  AgStats* partial = c.take<AgStats>(AG_RED_BLOCKS);
  const int nblocks = ag_blocks(n, 256 * 8, AG_RED_BLOCKS);
  if (norm != 0) {
    for (int pass = 0; pass < (norm == 5 ? 2 : 1); ++pass) {
      AG_KERNEL("ag_reduce_partial_f64_kernel", ag_reduce_partial_f64_kernel, dim3(nblocks), dim3(256), s, d_values, n, pass, stats, partial);
      AG_LAUNCH_CHECK();
      AG_KERNEL("ag_reduce_final_kernel", ag_reduce_final_kernel, dim3(1), dim3(256), s, partial, nblocks, pass, stats);
      AG_LAUNCH_CHECK();
    }
  }
  AG_KERNEL("ag_normalise_apply_f64_kernel", ag_normalise_apply_f64_kernel, dim3(ag_blocks(n, 256 * 4, 1 << 14)), dim3(256), s, d_values, n, norm, stats, out_f64 ? nullptr : static_cast<float*>(d_out), out_f64 ? static_cast<double*>(d_out) : nullptr);
  AG_LAUNCH_CHECK();
  return AG_OK;
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
  const int nblocks = ag_blocks(n, 256 * 16, AG_RED_BLOCKS);
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
  AG_KERNEL("ag_normalise_apply_kernel", ag_normalise_apply_kernel, dim3(ag_blocks(n, 256 * 8, 1 << 14)), dim3(256), s, d_values, n, norm, invert, stats, static_cast<const unsigned*>(nullptr));
  AG_LAUNCH_CHECK();
  return AG_OK;
}

extern "C" int ag_normalise_range(float* d_values, int64_t n, int norm, int invert, const uint32_t* d_range,
                                  void* stream) {
  AG_REQUIRE(norm == 0 || norm == 3 || norm == 4, "ag_normalise_range: normalisation %d does not follow from a range", norm);
  AG_REQUIRE(n >= 0, "ag_normalise_range: negative size");
  if (n == 0 || (norm == 0 && !invert)) return AG_OK;
  AG_REQUIRE(d_values && d_range, "ag_normalise_range: null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  AG_KERNEL("ag_normalise_apply_kernel", ag_normalise_apply_kernel, dim3(ag_blocks(n, 256 * 8, 1 << 14)), dim3(256), s, d_values, n, norm, invert, static_cast<const AgStats*>(nullptr), d_range);
  AG_LAUNCH_CHECK();
  return AG_OK;
}
