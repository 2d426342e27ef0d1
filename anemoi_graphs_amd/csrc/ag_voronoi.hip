// Areas of the spherical Voronoi cells of a node set (SphericalAreaWeights) for gfx950.
//
// The reference builds scipy's SphericalVoronoi (convex hull of the generators, one Voronoi vertex
// per hull facet = the facet's unit normal) and sums, per generator p, the solid angles of the
// triangles (p, v_k, v_k+1) of its region.  A facet normal v satisfies v.(p - q) = 0 for the
// generators p, q on the facet and v.(p - s) >= 0 for every other generator s, so the region of p is
// exactly the set of directions x with x.(p - q) >= 0 for all q -- also for generators that are only
// approximately on the sphere (they are float32 unit vectors).  Each cell is therefore built on its
// own, with no global triangulation: one thread per generator clips a convex polygon by the
// half-space of one neighbour after the other, nearest first, and stops when no farther neighbour can
// reach the polygon.
//
// The polygon lives in the gnomonic chart of p: x = h + u e1 + v e2 with h = p/|p| and (e1, e2, h)
// right-handed.  Great circles are straight lines there, the half-space of q is
// c0 + a u + b v >= 0 with (c0, a, b) = (h, e1, e2).(p - q), evaluated in float64 from the exact
// float32 differences, and a cell of angular size 1e-3 keeps full float64 relative precision.  The
// solid angle of (p, v1, v2) in these coordinates needs no cancellation:
//   det(p, v1, v2) = |p| (u1 v2 - u2 v1) / (L1 L2),  L = sqrt(1 + u^2 + v^2),
//   1 + p.v1 + v1.v2 + v2.p = 1 + |p|/L1 + (1 + u1 u2 + v1 v2)/(L1 L2) + |p|/L2,
// the two arguments of scipy's calculate_solid_angles (Van Oosterom-Strackee).
#include "ag_common.hpp"

// Half-width of the initial square in gnomonic units.  Cells are built in a square reaching 63 degrees
// from the generator, which keeps every intermediate vertex near the size of the final cell (full
// relative precision); only a cell that still touches that square is rebuilt in one reaching to within
// 1e-5 rad of 90 degrees.  A cell that extends farther still (a handful of nodes on the whole sphere,
// or the rim of a node set that leaves most of the sphere empty) cannot be represented in the chart of
// its generator and is reported as unbounded.
#define AG_VOR_T_NEAR 2.0
#define AG_VOR_T_FAR 1.0e5

// How much closer to the generator than its neighbour order suggests a boundary line can lie.  The
// generators are float32 unit vectors: their norms differ by up to 8e-8, which moves the boundary line
// of a neighbour at angular distance t by up to 8e-8 / t (the cells are those of a power diagram), and
// the neighbour tables are ordered by the haversine distance of the float32 coordinates, not by chart
// distance.  `d` is the distance of a line from the generator (about t / 2).
__device__ __forceinline__ double ag_vor_slack(double d) { return 1.0e-6 + 2.0e-7 / fmax(d, 1.0e-9); }

// The polygon of every lane: two buffers of MAXV vertices, lane-interleaved so that lanes working on
// the same vertex index hit different LDS banks.
template <int MAXV>
struct AgVorPoly {
  double u[2][MAXV][64];
  double v[2][MAXV][64];
};

struct AgVorFrame {
  double px, py, pz;     // generator: the float32 unit vector of the node
  double hx, hy, hz;     // h = p / |p|
  double e1x, e1y, e1z;  // east
  double e2x, e2y, e2z;  // north (h x e1)
  double norm;           // |p|
};

// generators as the reference forms them: latlon_rad_to_cartesian on float32 coordinates
// (generate/transforms.py:106-110), i.e. numpy's float32 sin/cos and float32 products
__device__ __forceinline__ void ag_vor_xyz(const float2* __restrict__ x, int64_t i, double* px, double* py, double* pz) {
  const float2 c = x[i];
  const float3 p = ag_unit_vec_f32(c.x, c.y);
  *px = static_cast<double>(p.x);
  *py = static_cast<double>(p.y);
  *pz = static_cast<double>(p.z);
}

__device__ __forceinline__ AgVorFrame ag_vor_frame(double px, double py, double pz) {
  AgVorFrame f;
  f.px = px;
  f.py = py;
  f.pz = pz;
  f.norm = sqrt(px * px + py * py + pz * pz);
  const double inv = 1.0 / f.norm;
  f.hx = px * inv;
  f.hy = py * inv;
  f.hz = pz * inv;
  // e1 = axis x h normalised, axis = z unless h is within ~8 degrees of a pole
  double ax = 0.0, ay = 0.0, az = 1.0;
  if (fabs(f.hz) > 0.99) {
    ax = 1.0;
    az = 0.0;
  }
  double cx = ay * f.hz - az * f.hy, cy = az * f.hx - ax * f.hz, cz = ax * f.hy - ay * f.hx;
  const double ci = 1.0 / sqrt(cx * cx + cy * cy + cz * cz);
  f.e1x = cx * ci;
  f.e1y = cy * ci;
  f.e1z = cz * ci;
  f.e2x = f.hy * f.e1z - f.hz * f.e1y;
  f.e2y = f.hz * f.e1x - f.hx * f.e1z;
  f.e2z = f.hx * f.e1y - f.hy * f.e1x;
  return f;
}

// State of one cell under construction
struct AgVorCell {
  int m;           // vertices
  int cur;         // buffer holding them
  double rho2max;  // largest u^2 + v^2 of a vertex
  bool overflow;   // more than MAXV vertices were needed
  bool duplicate;  // another generator coincides with this one
};

template <int MAXV>
__device__ __forceinline__ void ag_vor_init(AgVorPoly<MAXV>& P, int lane, AgVorCell& c, double T) {
  P.u[0][0][lane] = -T;
  P.v[0][0][lane] = -T;
  P.u[0][1][lane] = T;
  P.v[0][1][lane] = -T;
  P.u[0][2][lane] = T;
  P.v[0][2][lane] = T;
  P.u[0][3][lane] = -T;
  P.v[0][3][lane] = T;
  c.m = 4;
  c.cur = 0;
  c.rho2max = 2.0 * T * T;
  c.overflow = false;
  c.duplicate = false;
}

// Half-space of the generator q.  Returns the distance of its boundary line from the chart origin
// (negative if the origin itself is cut off), or -1 when q counts as a duplicate of p: scipy refuses
// generators closer than its threshold of 1e-6 (SphericalVoronoi.__init__, query_pairs).
__device__ __forceinline__ double ag_vor_line(const AgVorFrame& f, double qx, double qy, double qz, double* c0, double* a,
                                              double* b) {
  const double dx = f.px - qx, dy = f.py - qy, dz = f.pz - qz;
  *c0 = f.hx * dx + f.hy * dy + f.hz * dz;
  *a = f.e1x * dx + f.e1y * dy + f.e1z * dz;
  *b = f.e2x * dx + f.e2y * dy + f.e2z * dz;
  const double g2 = *a * *a + *b * *b;
  if (dx * dx + dy * dy + dz * dz <= 1.0e-12 || g2 == 0.0) return -1.0;
  return *c0 / sqrt(g2);
}

// Sutherland-Hodgman step: keep the part of the polygon with c0 + a u + b v >= 0.  Vertices closer
// than 1e-11 of the polygon's radius are merged: where many generators are equidistant from a point
// (a ring of nodes around a pole) all their boundary lines pass through one vertex, and every clip
// would otherwise split it into two vertices a rounding error apart.
template <int MAXV>
__device__ __forceinline__ void ag_vor_clip(AgVorPoly<MAXV>& P, int lane, AgVorCell& c, double c0, double a, double b) {
  const int cur = c.cur, m = c.m;
  bool cut = false;
  for (int k = 0; k < m; ++k) cut = cut || (c0 + a * P.u[cur][k][lane] + b * P.v[cur][k][lane] < 0.0);
  if (!cut) return;
  const int nxt = cur ^ 1;
  const double tol2 = 1.0e-22 * c.rho2max;
  int mo = 0;
  bool over = false;
  double up = P.u[cur][m - 1][lane], vp = P.v[cur][m - 1][lane];
  double sp = c0 + a * up + b * vp;
  double rho2 = 0.0;
  double ul = 0.0, vl = 0.0;  // the vertex added last
  auto add = [&](double un, double vn) {
    if (mo > 0 && (un - ul) * (un - ul) + (vn - vl) * (vn - vl) <= tol2) return;
    if (mo < MAXV) {
      P.u[nxt][mo][lane] = un;
      P.v[nxt][mo][lane] = vn;
      rho2 = fmax(rho2, un * un + vn * vn);
      ul = un;
      vl = vn;
      ++mo;
    } else {
      over = true;
    }
  };
  for (int k = 0; k < m; ++k) {
    const double uk = P.u[cur][k][lane], vk = P.v[cur][k][lane];
    const double sk = c0 + a * uk + b * vk;
    if ((sp < 0.0) != (sk < 0.0)) {
      const double t = sp / (sp - sk);
      add(up + t * (uk - up), vp + t * (vk - vp));
    }
    if (sk >= 0.0) add(uk, vk);
    up = uk;
    vp = vk;
    sp = sk;
  }
  if (over) {
    c.overflow = true;
    return;
  }
  if (mo > 1) {  // the last vertex against the first
    const double du = P.u[nxt][0][lane] - ul, dv = P.v[nxt][0][lane] - vl;
    if (du * du + dv * dv <= tol2) --mo;
  }
  c.m = mo;
  c.cur = nxt;
  c.rho2max = rho2;
}

template <int MAXV>
__device__ __forceinline__ double ag_vor_area(const AgVorPoly<MAXV>& P, int lane, const AgVorCell& c, double norm) {
  if (c.m < 3) return 0.0;
  double sum = 0.0;
  double u1 = P.u[c.cur][c.m - 1][lane], v1 = P.v[c.cur][c.m - 1][lane];
  double l1 = sqrt(1.0 + u1 * u1 + v1 * v1);
  for (int k = 0; k < c.m; ++k) {
    const double u2 = P.u[c.cur][k][lane], v2 = P.v[c.cur][k][lane];
    const double l2 = sqrt(1.0 + u2 * u2 + v2 * v2);
    const double ll = l1 * l2;
    const double num = norm * (u1 * v2 - u2 * v1) / ll;
    const double den = 1.0 + norm / l1 + (1.0 + u1 * u2 + v1 * v2) / ll + norm / l2;
    sum += fabs(2.0 * atan2(num, den));
    u1 = u2;
    v1 = v2;
    l1 = l2;
  }
  return sum;
}

// What a finished cell reports besides its area: 0 = complete; > 0 = the angular radius (radians)
// within which every generator must be offered before the cell is final (infinity while the
// initial square is still part of it); AG_VOR_DUPLICATE / AG_VOR_OVERFLOW for the two failures.
#define AG_VOR_DUPLICATE (-1.0)
#define AG_VOR_OVERFLOW (-2.0)

__device__ __forceinline__ bool ag_vor_touches_square(double rho2max, double T) {
  return rho2max >= T * T * (1.0 - 1.0e-9);
}

__device__ __forceinline__ double ag_vor_security_radius(double rho2max, double T) {
  if (ag_vor_touches_square(rho2max, T)) return __builtin_huge_val();
  const double rho = sqrt(rho2max);
  return 2.0 * atan(rho + ag_vor_slack(rho)) * (1.0 + 1.0e-6) + 1.0e-6;
}

// Neighbours from a table: d_nbr[i*K + j] = j-th nearest node of node i (nearest first; the node
// itself may be among them).  A farther neighbour cannot cut the polygon once the boundary line of
// the current one passes outside every vertex by more than ag_vor_slack.
template <int MAXV>
__global__ __launch_bounds__(64) void ag_voronoi_knn_kernel(const float2* __restrict__ x, int64_t n,
                                                            const int32_t* __restrict__ nbr, int K,
                                                            double* __restrict__ area, double* __restrict__ reach) {
  __shared__ AgVorPoly<MAXV> P;
  const int lane = threadIdx.x;
  const int64_t i = blockIdx.x * int64_t(64) + lane;
  if (i >= n) return;
  double px, py, pz;
  ag_vor_xyz(x, i, &px, &py, &pz);
  const AgVorFrame f = ag_vor_frame(px, py, pz);
  AgVorCell c;
  ag_vor_init(P, lane, c, AG_VOR_T_NEAR);
  bool closed = false;
  for (int j = 0; j < K && !c.overflow; ++j) {
    const int64_t q = nbr[i * K + j];
    if (q == i || q < 0 || q >= n) continue;
    double qx, qy, qz, c0, a, b;
    ag_vor_xyz(x, q, &qx, &qy, &qz);
    const double dist = ag_vor_line(f, qx, qy, qz, &c0, &a, &b);
    if (dist == -1.0) {
      c.duplicate = true;
      continue;
    }
    if (dist - ag_vor_slack(dist) > sqrt(c.rho2max) * (1.0 + 1.0e-6)) {
      closed = true;
      break;
    }
    ag_vor_clip(P, lane, c, c0, a, b);
  }
  // (a polygon that outgrew this kernel's buffer is still a superset of the cell: its security radius
  // is valid, and the row kernel with its larger buffer finishes the cell)
  // (a polygon that still has a side of the initial square is not a cell yet, whatever the ordering says)
  if (ag_vor_touches_square(c.rho2max, AG_VOR_T_NEAR)) closed = false;
  double r = (closed && !c.overflow) ? 0.0 : ag_vor_security_radius(c.rho2max, AG_VOR_T_NEAR);
  if (c.duplicate) r = AG_VOR_DUPLICATE;
  area[i] = ag_vor_area(P, lane, c, f.norm);
  reach[i] = r;
}

// Neighbours from rows of a radius query (any order): row t holds every node within `radius` of node
// ids[t].  The cell is final when its security radius does not exceed `radius`.
// The order matters for the size of the intermediate polygons -- far nodes applied first shave the
// initial square into a polygon with one edge per node -- so the row is swept in bands of doubling
// distance, nearest band first; after the first bands the polygon is close to the final cell and
// the farther bands are rejected by the security test without touching it.
template <int MAXV>
__global__ __launch_bounds__(64) void ag_voronoi_rows_kernel(const float2* __restrict__ x, int64_t n,
                                                             const int32_t* __restrict__ ids, int64_t nt,
                                                             const int64_t* __restrict__ offsets,
                                                             const int32_t* __restrict__ nbr, double radius,
                                                             double* __restrict__ area, double* __restrict__ reach) {
  __shared__ AgVorPoly<MAXV> P;
  const int lane = threadIdx.x;
  const int64_t t = blockIdx.x * int64_t(64) + lane;
  if (t >= nt) return;
  const int64_t i = ids[t];
  if (i < 0 || i >= n) return;
  double px, py, pz;
  ag_vor_xyz(x, i, &px, &py, &pz);
  const AgVorFrame f = ag_vor_frame(px, py, pz);
  AgVorCell c;
  c.duplicate = false;
  const int64_t e0 = offsets[t], e1 = offsets[t + 1];
  // distance range of the boundary lines of the row
  double dmin = __builtin_huge_val(), dmax = 0.0;
  for (int64_t e = e0; e < e1; ++e) {
    const int64_t q = nbr[e];
    if (q == i || q < 0 || q >= n) continue;
    double qx, qy, qz, c0, a, b;
    ag_vor_xyz(x, q, &qx, &qy, &qz);
    const double dist = ag_vor_line(f, qx, qy, qz, &c0, &a, &b);
    if (dist == -1.0) {
      c.duplicate = true;
      continue;
    }
    if (dist > 0.0) dmin = fmin(dmin, dist);
    dmax = fmax(dmax, dist);
  }
  const bool duplicate = c.duplicate;
  double T = AG_VOR_T_NEAR;
  for (int attempt = 0; attempt < 2; ++attempt) {
  T = attempt == 0 ? AG_VOR_T_NEAR : AG_VOR_T_FAR;
  ag_vor_init(P, lane, c, T);
  c.duplicate = duplicate;
  double lo = -__builtin_huge_val(), hi = 2.0 * dmin;
  while (!c.overflow && !c.duplicate && dmin <= dmax) {
    for (int64_t e = e0; e < e1 && !c.overflow; ++e) {
      const int64_t q = nbr[e];
      if (q == i || q < 0 || q >= n) continue;
      double qx, qy, qz, c0, a, b;
      ag_vor_xyz(x, q, &qx, &qy, &qz);
      const double dist = ag_vor_line(f, qx, qy, qz, &c0, &a, &b);
      if (!(dist >= lo && dist < hi)) continue;
      if (dist - ag_vor_slack(dist) > sqrt(c.rho2max) * (1.0 + 1.0e-6)) continue;
      ag_vor_clip(P, lane, c, c0, a, b);
    }
    if (hi > dmax) break;                                                    // every line was offered
    if (hi - ag_vor_slack(hi) > sqrt(c.rho2max) * (1.0 + 1.0e-6)) break;  // no farther line reaches the polygon
    lo = hi;
    hi *= 2.0;
  }
  if (c.overflow || c.duplicate || !ag_vor_touches_square(c.rho2max, T)) break;
  }
  double r = ag_vor_security_radius(c.rho2max, T);
  if (r <= radius) r = 0.0;
  if (c.overflow) r = AG_VOR_OVERFLOW;
  if (c.duplicate) r = AG_VOR_DUPLICATE;
  area[i] = ag_vor_area(P, lane, c, f.norm);
  reach[i] = r;
}

// Cells that reach 90 degrees or more away from their generator (the rim of a node set that leaves
// part of the sphere empty, or a handful of nodes on the whole sphere) do not fit the chart of their
// generator -- but a cell with at least two neighbours lies inside some open hemisphere, and the chart
// may be centred anywhere in it: the half-space of q is still c0 + a u + b v >= 0, now with
// (c0, a, b) = (c, e1, e2).(p - q) for the chart centre c.  One thread per such cell: build the part
// of the cell visible from p, re-centre the chart on the mean direction of its vertices (a point of
// the cell, looking towards the side that ran off the chart), rebuild, repeat while a side of the
// initial square survives.  Every node is offered (a wide cell has no useful security radius), in
// bands of doubling line distance as in the row kernel; the unit vectors come from a table.  The
// area is scipy's formula on the three-dimensional vertices.
__global__ void ag_voronoi_xyz_kernel(const float2* __restrict__ x, int64_t n, float4* __restrict__ xyz) {
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
    const float2 c = x[i];
    const float3 p = ag_unit_vec_f32(c.x, c.y);
    xyz[i] = make_float4(p.x, p.y, p.z, 0.0f);
  }
}

#define AG_VOR_RECENTRE 8  // charts tried per cell

template <int MAXV>
__global__ __launch_bounds__(64) void ag_voronoi_far_kernel(const float4* __restrict__ xyz, int64_t n,
                                                            const int32_t* __restrict__ ids, int64_t nt,
                                                            double* __restrict__ area, double* __restrict__ reach) {
  __shared__ AgVorPoly<MAXV> P;
  const int lane = threadIdx.x;
  const int64_t t = blockIdx.x * int64_t(64) + lane;
  if (t >= nt) return;
  const int64_t i = ids[t];
  if (i < 0 || i >= n) return;
  const float4 pf = xyz[i];
  const double px = pf.x, py = pf.y, pz = pf.z;
  AgVorFrame f = ag_vor_frame(px, py, pz);
  const double pnorm = f.norm;
  double cx = f.hx, cy = f.hy, cz = f.hz;  // chart centre, starts at the generator
  AgVorCell c;
  bool done = false, duplicate = false;
  for (int attempt = 0; attempt < AG_VOR_RECENTRE && !done; ++attempt) {
    f = ag_vor_frame(cx, cy, cz);
    f.px = px;
    f.py = py;
    f.pz = pz;
    double dmin = __builtin_huge_val(), dmax = 0.0;
    for (int64_t q = 0; q < n; ++q) {
      if (q == i) continue;
      const float4 qf = xyz[q];
      double c0, a, b;
      const double dist = ag_vor_line(f, qf.x, qf.y, qf.z, &c0, &a, &b);
      if (dist == -1.0) {
        duplicate = true;
        continue;
      }
      if (dist > 0.0) dmin = fmin(dmin, dist);
      dmax = fmax(dmax, dist);
    }
    ag_vor_init(P, lane, c, AG_VOR_T_FAR);
    double lo = -__builtin_huge_val(), hi = 2.0 * dmin;
    while (!c.overflow && !duplicate && dmin <= dmax) {
      for (int64_t q = 0; q < n && !c.overflow; ++q) {
        if (q == i) continue;
        const float4 qf = xyz[q];
        double c0, a, b;
        const double dist = ag_vor_line(f, qf.x, qf.y, qf.z, &c0, &a, &b);
        if (!(dist >= lo && dist < hi)) continue;
        if (dist - ag_vor_slack(dist) > sqrt(c.rho2max) * (1.0 + 1.0e-6)) continue;
        ag_vor_clip(P, lane, c, c0, a, b);
      }
      if (hi > dmax) break;
      if (hi - ag_vor_slack(hi) > sqrt(c.rho2max) * (1.0 + 1.0e-6)) break;
      lo = hi;
      hi *= 2.0;
    }
    if (c.overflow || duplicate) break;
    if (!ag_vor_touches_square(c.rho2max, AG_VOR_T_FAR)) {
      done = true;
      break;
    }
    // next chart centre: mean direction of the vertices of what this chart shows of the cell
    double sx = 0.0, sy = 0.0, sz = 0.0;
    for (int k = 0; k < c.m; ++k) {
      const double u = P.u[c.cur][k][lane], v = P.v[c.cur][k][lane];
      const double il = 1.0 / sqrt(1.0 + u * u + v * v);
      sx += (f.hx + u * f.e1x + v * f.e2x) * il;
      sy += (f.hy + u * f.e1y + v * f.e2y) * il;
      sz += (f.hz + u * f.e1z + v * f.e2z) * il;
    }
    const double sn = sqrt(sx * sx + sy * sy + sz * sz);
    if (!(sn > 1.0e-9)) break;
    cx = sx / sn;
    cy = sy / sn;
    cz = sz / sn;
  }
  double total = 0.0;
  if (done && c.m >= 3) {
    // scipy's calculate_solid_angles on (p, v_k, v_k+1) with the unit vertices v = (h + u e1 + v e2) / L
    auto vertex = [&](int k, double* vx, double* vy, double* vz) {
      const double u = P.u[c.cur][k][lane], v = P.v[c.cur][k][lane];
      const double il = 1.0 / sqrt(1.0 + u * u + v * v);
      *vx = (f.hx + u * f.e1x + v * f.e2x) * il;
      *vy = (f.hy + u * f.e1y + v * f.e2y) * il;
      *vz = (f.hz + u * f.e1z + v * f.e2z) * il;
    };
    double ax, ay, az;
    vertex(c.m - 1, &ax, &ay, &az);
    for (int k = 0; k < c.m; ++k) {
      double bx, by, bz;
      vertex(k, &bx, &by, &bz);
      const double det = px * (ay * bz - az * by) + py * (az * bx - ax * bz) + pz * (ax * by - ay * bx);
      const double den = 1.0 + (px * ax + py * ay + pz * az) + (ax * bx + ay * by + az * bz) + (bx * px + by * py + bz * pz);
      total += fabs(2.0 * atan2(det, den));
      ax = bx;
      ay = by;
      az = bz;
    }
  }
  (void)pnorm;
  double r = done ? 0.0 : __builtin_huge_val();
  if (c.overflow) r = AG_VOR_OVERFLOW;
  if (duplicate) r = AG_VOR_DUPLICATE;
  area[i] = total;
  reach[i] = r;
}

extern "C" size_t ag_voronoi_far_workspace_bytes(int64_t n) { return static_cast<size_t>(n > 0 ? n : 1) * sizeof(float4) + 256; }

extern "C" int ag_voronoi_areas_far(const float* d_x, int64_t n, const int32_t* d_ids, int64_t nt, double* d_area,
                                    double* d_reach, void* d_ws, size_t ws_bytes, void* stream) {
  AG_REQUIRE(n >= 0 && nt >= 0, "ag_voronoi_areas_far: negative size");
  if (n == 0 || nt == 0) return AG_OK;
  AG_REQUIRE(d_x && d_ids && d_area && d_reach && d_ws, "ag_voronoi_areas_far: null pointer");
  if (ws_bytes < ag_voronoi_far_workspace_bytes(n)) return ag_fail(AG_ERR_WORKSPACE, "ag_voronoi_areas_far: workspace too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  AgCarver c(d_ws);
  float4* xyz = c.take<float4>(n);
  AG_KERNEL("ag_voronoi_xyz_kernel", ag_voronoi_xyz_kernel, dim3(ag_blocks(n, 256, 8192)), dim3(256), s,
            reinterpret_cast<const float2*>(d_x), n, xyz);
  AG_LAUNCH_CHECK();
  AG_KERNEL("ag_voronoi_far_kernel", ag_voronoi_far_kernel<48>, dim3(ag_blocks(nt, 64)), dim3(64), s, xyz, n, d_ids, nt,
            d_area, d_reach);
  AG_LAUNCH_CHECK();
  return AG_OK;
}

extern "C" int ag_voronoi_areas_knn(const float* d_x, int64_t n, const int32_t* d_nbr, int K, double* d_area,
                                    double* d_reach, void* stream) {
  AG_REQUIRE(n >= 0 && K >= 0, "ag_voronoi_areas_knn: negative size");
  if (n == 0) return AG_OK;
  AG_REQUIRE(d_x && d_area && d_reach && (d_nbr || K == 0), "ag_voronoi_areas_knn: null pointer");
  AG_KERNEL("ag_voronoi_knn_kernel", ag_voronoi_knn_kernel<16>, dim3(ag_blocks(n, 64)), dim3(64),
            static_cast<hipStream_t>(stream), reinterpret_cast<const float2*>(d_x), n, d_nbr, K, d_area, d_reach);
  AG_LAUNCH_CHECK();
  return AG_OK;
}

extern "C" int ag_voronoi_areas_rows(const float* d_x, int64_t n, const int32_t* d_ids, int64_t nt,
                                     const int64_t* d_offsets, const int32_t* d_nbr, double radius, double* d_area,
                                     double* d_reach, void* stream) {
  AG_REQUIRE(n >= 0 && nt >= 0, "ag_voronoi_areas_rows: negative size");
  if (n == 0 || nt == 0) return AG_OK;
  AG_REQUIRE(d_x && d_ids && d_offsets && d_nbr && d_area && d_reach, "ag_voronoi_areas_rows: null pointer");
  AG_KERNEL("ag_voronoi_rows_kernel", ag_voronoi_rows_kernel<48>, dim3(ag_blocks(nt, 64)), dim3(64),
            static_cast<hipStream_t>(stream), reinterpret_cast<const float2*>(d_x), n, d_ids, nt, d_offsets, d_nbr,
            radius, d_area, d_reach);
  AG_LAUNCH_CHECK();
  return AG_OK;
}
