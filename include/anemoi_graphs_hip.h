/*
 * C ABI of libanemoi_graphs_hip.so -- the MI355X (gfx950) graph-construction hot path.
 *
 * The reference (ecmwf/anemoi-graphs) has no FFI on this path: its builders are Python
 * classes that call scikit-learn / trimesh / networkx / scipy on the CPU.  Each entry point
 * below replaces one of those third-party calls; the comment above it cites the reference
 * call site it stands in for (paths relative to /root/reference/src/anemoi/graphs).
 *
 * Conventions
 *  - every pointer named d_* is DEVICE memory owned by the caller; nothing is allocated or
 *    freed inside the library.  Scratch memory comes from a caller-provided workspace whose
 *    minimum size the matching *_workspace_bytes() function returns.
 *  - coordinates are float32 [N,2] row-major (latitude, longitude) in radians, the layout of
 *    `graph[name].x` (nodes/builders/base.py:25,99-101).
 *  - node ids are int32 (edges/builder.py:87), sizes and offsets int64.
 *  - all work is enqueued on `stream` (a hipStream_t passed as void*); the functions do not
 *    synchronise unless stated.
 *  - return value 0 = success, negative = error; ag_last_error() gives the message of the
 *    last failure on the calling thread.
 */
#ifndef ANEMOI_GRAPHS_HIP_H
#define ANEMOI_GRAPHS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AG_OK 0
#define AG_ERR_INVALID (-1)
#define AG_ERR_WORKSPACE (-2)
#define AG_ERR_HIP (-3)

const char* ag_last_error(void);
int ag_version(void);

/* Per-kernel timing (no counterpart in the reference).  While enabled, every kernel launch of
 * the library is bracketed by HIP events recorded on the launch stream.  ag_profile_collect
 * synchronises on those events and writes one line per kernel name, "<name> <launches>
 * <total_ms>\n", into buf (NUL-terminated, truncated to buf_bytes); returns the untruncated
 * length.  ag_profile_enable(0|1) also discards what was recorded so far. */
int ag_profile_enable(int on);
int ag_profile_filter(const char* kernel_name); /* NULL or "": every kernel; else only launches of that name */
int ag_profile_collect(char* buf, size_t buf_bytes);

/* Status of the current device (no counterpart in the reference).  The kernels never stop a launch half-way: one
 * that cannot do its work -- a single-launch prefix sum that finds its state words not cleared or waits in vain
 * for a tile (value 1), a binning whose cell starts did not add up to the number of points (value 2: the binning
 * is then left EMPTY but sound, every record initialised, so that the searches that read it stay inside their
 * arrays and find nothing) -- sets the status word of its device instead.  ag_status_take copies the word to
 * *status and clears it; it waits for the device work queued before it.  The Python layer calls it wherever it
 * synchronises for the caller and raises LibraryError on a non-zero value.
 * ag_debug_poison_scans(1) is a test hook: until it is switched off, binning calls fill the state words of their
 * prefix sum with ones instead of clearing them, so that the path above can be exercised with every kernel still
 * ending normally. */
int ag_status_take(int* status);
int ag_debug_poison_scans(int on);

/* ---------------------------------------------------------------------------------------
 * k nearest neighbours on the sphere.
 * Replaces NearestNeighbors(metric="haversine").fit(src).kneighbors_graph(tgt, k, "distance")
 * (edges/builder.py:259-265).  For every target t, d_out_src[t*k + j] is the id of its j-th
 * nearest source, ascending in the float64 haversine "rdist"
 * sin^2(dlat/2) + cos(lat_t) cos(lat_s) sin^2(dlon/2) of the float32 inputs; exact float64
 * ties are broken towards the lower source id.  d_out_dist (nullable) receives the distance
 * 2*asin(sqrt(rdist)) in radians.  d_stats (nullable, int64[4]) receives: [0] targets resolved
 * by the float64 wave-per-target path, [1] passes of that path beyond the first window.
 * Requires 1 <= k <= ns.  Up to AG_KNN_MAX_K neighbours the targets share tiles of sources in a first stage; beyond
 * that every target is searched on its own by a wavefront (any k: the reference has no limit either, edges/builder.py:222-233).
 *
 * src_density, tgt_density (no counterpart in the reference, whose ball tree adapts to its points by itself):
 * how many times more nodes per steradian a set has, where it has nodes, than the same number of nodes spread
 * over the whole sphere -- 1 for a global grid or mesh, 1/f for a regional one that covers the fraction f of the
 * sphere; values below 1 count as 1.  The searches bin the sources into cells sized for the density given (a
 * regional set binned as if it were global crowds thirty times the intended sources into every cell) and choose
 * how the targets are handed to the lanes from theirs.  The densities only decide how fast a search runs; its
 * results are the same for any values.  ag_point_density measures the density of a set whose extent the caller
 * does not know.  The workspace is sized for the source density that the search is then given.
 */
#define AG_KNN_MAX_K 64
size_t ag_knn_workspace_bytes(int64_t ns, int64_t nt, int k, double src_density);
int ag_knn(const float* d_src, int64_t ns, const float* d_tgt, int64_t nt, int k, double src_density,
           double tgt_density, int32_t* d_out_src, double* d_out_dist, int64_t* d_stats, void* d_ws, size_t ws_bytes,
           void* stream);

/* The density of a node set as the k-NN entry points take it: the mean over the nodes of the density around them
 * (counted in up to 32768 cells of equal area), relative to that of as many nodes spread over the sphere.  1 for
 * a global grid, 1/f for a grid that covers the fraction f of the sphere, about the value of its dense part for a
 * stretched grid.  Writes one float64 to *d_out (device memory). */
size_t ag_point_density_workspace_bytes(int64_t n);
int ag_point_density(const float* d_pts, int64_t n, double* d_out, void* d_ws, size_t ws_bytes, void* stream);

/* ag_knn that does not look beyond max_dist (radians): an entry whose neighbour would be farther than that is
 * -1 in d_out_src and +infinity in d_out_dist.  Replaces `kneighbors(coords, n_neighbors=1)` followed by the
 * threshold `dists * EARTH_RADIUS <= margin_radius_km` of KNNAreaMaskBuilder.get_mask (generate/masks.py:94-99):
 * the mask only asks whether a reference node lies within the margin, so mesh vertices far from a regional
 * reference set need no search over the whole sphere.  Workspace: ag_knn_workspace_bytes(ns, nt, k, src_density). */
int ag_knn_within(const float* d_src, int64_t ns, const float* d_tgt, int64_t nt, int k, double max_dist,
                  double src_density, double tgt_density, int32_t* d_out_src, double* d_out_dist, int64_t* d_stats,
                  void* d_ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------
 * Grid reference distance.
 * Replaces utils.get_grid_reference_distance (utils.py:61-63): kneighbors(pts, 2) on the set
 * itself and dists[dists > 0].max().  Writes the float64 result (radians) to *d_out_max
 * (0.0 if no positive distance exists).  density: of the set, as for ag_knn.
 */
size_t ag_nn_maxdist_workspace_bytes(int64_t n, double density);
int ag_nn_maxdist(const float* d_pts, int64_t n, double density, double* d_out_max, void* d_ws, size_t ws_bytes,
                  void* stream);

/* ag_nn_maxdist without a workspace argument, for callers that cannot allocate device memory (code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc).  It has no counterpart in the
 * reference.  The scratch memory is an array built into the library (48 MiB on every GPU the library runs on,
 * held until the process ends), which bounds the input: 2 <= n <= AG_NN_MAXDIST_STATIC_MAX_N, more is refused
 * with an error.  Nothing is allocated or freed, by the library or by a routine it calls; the first call on a
 * device looks the array's address up (hipGetSymbolAddress), so make that call outside a capture.  Same
 * result as ag_nn_maxdist, bit for bit.  All calls on one GPU share that GPU's array: calls that overlap on one
 * GPU (two streams, two threads) race; successive calls on one stream do not.  Call with the device of
 * `stream` current.  The set is binned as one spread over the sphere (density 1). */
#define AG_NN_MAXDIST_STATIC_MAX_N 262144
int ag_nn_maxdist_static(const float* d_pts, int64_t n, double* d_out_max, void* stream);

/* ---------------------------------------------------------------------------------------
 * Radius (cut-off) query.
 * Replaces NearestNeighbors(metric="haversine").fit(src).radius_neighbors_graph(tgt, radius)
 * (edges/builder.py:364-366): all (source, target) pairs with rdist <= sin^2(radius/2).
 * Two calls sharing one workspace: ag_radius_count bins the sources and writes the
 * exclusive prefix sum of the per-target match counts to d_offsets[0..nt] (d_offsets[nt] =
 * number of edges E) and the longest row to d_offsets[nt + 1] -- d_offsets holds nt + 2
 * entries; the caller reads both, allocates, and calls ag_radius_fill with the SAME arguments,
 * the two values and the SAME untouched workspace.  Rows are target-major; inside a row source
 * ids ascend.  Rows of up to 256 entries are sorted in LDS by the count pass and only copied by
 * the fill pass; longer rows are searched again and ordered by one segmented radix sort (that
 * path needs E < 2^32).  d_out_tgt [E] is required: it receives the target id of every edge
 * and serves as scratch for the long rows.
 */
size_t ag_radius_workspace_bytes(int64_t ns, int64_t nt);
int ag_radius_count(const float* d_src, int64_t ns, const float* d_tgt, int64_t nt, double radius,
                    int64_t* d_offsets, void* d_ws, size_t ws_bytes, void* stream);
int ag_radius_fill(const float* d_src, int64_t ns, const float* d_tgt, int64_t nt, double radius,
                   const int64_t* d_offsets, int64_t num_edges, int64_t max_row, int32_t* d_out_src,
                   int32_t* d_out_tgt, void* d_ws, size_t ws_bytes, void* stream);
/* The order in which the count pass takes the targets (no counterpart in the reference): 0, the default, the
 * caller's; 1, sorted into groups of neighbours that share their source cells (three more launches, a third of the
 * cache-to-memory traffic of the search, the same duration on quasi-uniform node sets and a longer one on uneven
 * ones).  The edge lists are identical either way.  Process-wide; returns the previous setting. */
int ag_radius_group_order(int on);

/* ---------------------------------------------------------------------------------------
 * A binning the caller keeps (no counterpart in the reference, whose every builder fits a ball tree of its
 * own: edges/builder.py:259-260,364-365).  One node set often serves several searches of a graph -- the data
 * nodes are the sources of the cut-off encoder and the targets of the k-NN decoder -- and each of the entry
 * points above bins its inputs inside its workspace.  ag_binning_build bins a node set once into a caller-owned
 * buffer of ag_binning_bytes(n) bytes, with cells of about `cell` radians (for a cut-off search:
 * ag_radius_cell(ns, radius), the size ag_radius_count would choose itself); the *_binned entry points take the
 * buffer together with the SAME n and cell and skip their own binning of that set:
 *  - ag_radius_count_binned / ag_radius_fill_binned: the binning of the SOURCES (any cell size works; the
 *    search is fastest with ag_radius_cell).  Workspace: ag_radius_binned_workspace_bytes(ns, nt).
 *  - ag_knn_binned: the binning of the TARGETS (their order and unit vectors are taken from it).  Workspace:
 *    ag_knn_binned_workspace_bytes(ns, nt, k, src_density).  src_density: of the sources, as for ag_knn (the binning
 *    says where the targets are; their density is not asked for).
 * Results are identical to the entry points without a kept binning.  The buffer must stay untouched while
 * a search that was given it is in flight. */
size_t ag_binning_bytes(int64_t n);
double ag_radius_cell(int64_t ns, double radius);
int ag_binning_build(const float* d_pts, int64_t n, double cell, void* d_binning, size_t binning_bytes, void* stream);
size_t ag_radius_binned_workspace_bytes(int64_t ns, int64_t nt);
int ag_radius_count_binned(const float* d_src, int64_t ns, const void* d_src_binning, double src_cell, const float* d_tgt,
                           int64_t nt, double radius, int64_t* d_offsets, void* d_ws, size_t ws_bytes, void* stream);
int ag_radius_fill_binned(const float* d_src, int64_t ns, const void* d_src_binning, double src_cell, const float* d_tgt,
                          int64_t nt, double radius, const int64_t* d_offsets, int64_t num_edges, int64_t max_row,
                          int32_t* d_out_src, int32_t* d_out_tgt, void* d_ws, size_t ws_bytes, void* stream);
size_t ag_knn_binned_workspace_bytes(int64_t ns, int64_t nt, int k, double src_density);
int ag_knn_binned(const float* d_src, int64_t ns, const float* d_tgt, int64_t nt, const void* d_tgt_binning,
                  double tgt_cell, int k, double src_density, int32_t* d_out_src, double* d_out_dist, int64_t* d_stats,
                  void* d_ws, size_t ws_bytes, void* stream);

/* The k-NN graph as an edge list [2, nt*k] (edges/builder.py:259-265: the COO form of kneighbors_graph): d_edge_src =
 * what ag_knn writes, d_edge_tgt = the target of every entry (0,0,0,1,1,1,... for k = 3), written by the same
 * kernels instead of a pass of its own (ag_fill_row_ids).  d_tgt_binning: a binning of the targets as for
 * ag_knn_binned (workspace ag_knn_binned_workspace_bytes), or NULL (workspace ag_knn_workspace_bytes).  src_density,
 * tgt_density: as for ag_knn (tgt_density is not read when a binning of the targets is given). */
int ag_knn_edges(const float* d_src, int64_t ns, const float* d_tgt, int64_t nt, const void* d_tgt_binning,
                 double tgt_cell, int k, double src_density, double tgt_density, int32_t* d_edge_src, int32_t* d_edge_tgt,
                 int64_t* d_stats, void* d_ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------
 * Icosphere nodes.
 * Replaces trimesh.creation.icosphere(subdivisions=R, radius=1.0) + cartesian_to_latlon_rad +
 * get_coordinates_ordering (generate/tri_icosahedron.py:47-49,121-122; generate/transforms.py:
 * 50-52; generate/utils.py:30-32).  Sizes: V = 10*4^R + 2, F = 20*4^R, E_all = 10*(4^(R+1) - 1).
 *   d_xyz         float64 [V,3]  unit vertices in trimesh vertex order (level-l vertices are a prefix)
 *   d_faces       int32   [F,3]  faces of level R
 *   d_latlon      float32 [V,2]  (lat, lon) of the vertices in NODE order: latitude descending,
 *                                ties longitude descending, remaining ties vertex id descending
 *                                (what generate/utils.py:30-32 yields with stable sorts)
 *   d_order       int32   [V]    node -> vertex  (the reference's `node_ordering`)
 *   d_rank        int32   [V]    vertex -> node  (inverse permutation)
 *                                d_order = d_rank = NULL: no node order is computed and d_latlon receives the
 *                                coordinates in VERTEX order -- for the limited-area and stretched meshes
 *                                (generate/tri_icosahedron.py:47-58,61-105), which keep a part of the vertices
 *                                and order that part (ag_node_order)
 *   d_level_edges uint64  [E_all] (nullable) the unique undirected mesh edges of every level 0..R as
 *                                (max vertex << 32 | min vertex), level l at offset 10*(4^l - 1).  This
 *                                is what the reference keeps in `_nx_graph` for MultiScaleEdges
 *                                (nodes/builders/from_refined_icosahedron.py:50-55).
 */
#define AG_ICO_MAX_LEVEL 11
int ag_icosphere_sizes(int R, int64_t* V, int64_t* F, int64_t* E_all);
size_t ag_icosphere_workspace_bytes(int R);
int ag_icosphere(int R, double* d_xyz, int32_t* d_faces, float* d_latlon, int32_t* d_order, int32_t* d_rank,
                 uint64_t* d_level_edges, void* d_ws, size_t ws_bytes, void* stream);

/* The grid reference distance (utils.py:61-63, what ag_nn_maxdist computes by a search) of the COMPLETE icosphere of
 * level R, from its mesh: the triangulation is a Delaunay triangulation of its vertices, the nearest neighbour of a
 * vertex is therefore a mesh neighbour, and the largest nearest-neighbour distance is the maximum over the vertices
 * of their shortest finest-level edge.  Same float64 result as ag_nn_maxdist on d_latlon, bit for bit, in three small
 * launches.  d_level_edges, d_latlon (node order), d_rank: as ag_icosphere returns them for the same R.  Not for
 * meshes that keep a part of the vertices (a node at the rim may have lost its nearest neighbour). */
size_t ag_mesh_maxdist_workspace_bytes(int R);
int ag_mesh_maxdist(int R, const uint64_t* d_level_edges, const float* d_latlon, const int32_t* d_rank, double* d_out_max,
                    void* d_ws, size_t ws_bytes, void* stream);

/* Node order of a set of (lat, lon) float32 coordinates: latitude descending, ties longitude
 * descending, ties node id descending -- what get_coordinates_ordering (generate/utils.py:15-33)
 * yields when its two argsorts are stable.  d_order[i] = id of the node at position i. */
size_t ag_node_order_workspace_bytes(int64_t n);
int ag_node_order(const float* d_latlon, int64_t n, int32_t* d_order, void* d_ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------
 * Multi-scale icosahedral edges.
 * Replaces tri_icosahedron.add_edges_to_nx_graph (generate/tri_icosahedron.py:165-188) and
 * nx.to_scipy_sparse_array (edges/builder.py:453): for every level in `levels` (host array) all
 * ordered pairs of distinct level-l vertices within x_hops mesh hops, in node numbering (d_rank
 * and d_level_edges of ag_icosphere for the same R), de-duplicated over levels.  Output is sorted
 * by (target, source).  Two calls sharing one workspace: count -> *d_num_edges (int64, device),
 * then fill.  d_active (nullable, uint8 [V]) restricts every level's mesh to edges whose two
 * vertices are active (generate/tri_icosahedron.py:214-215); vertices with d_rank < 0 are not
 * nodes of the graph and produce no edges.
 */
#define AG_MULTISCALE_MAX_HOPS 8
size_t ag_multiscale_workspace_bytes(int R, int nlevels, int x_hops);
int ag_multiscale_count(int R, const int32_t* levels, int nlevels, int x_hops, const int32_t* d_rank,
                        const uint8_t* d_active, const uint64_t* d_level_edges, int64_t* d_num_edges, void* d_ws,
                        size_t ws_bytes, void* stream);
int ag_multiscale_fill(int R, const int32_t* levels, int nlevels, int x_hops, int64_t num_edges, int32_t* d_out_src,
                       int32_t* d_out_tgt, void* d_ws, size_t ws_bytes, void* stream);

/* The same edges for x_hops = 1 on the COMPLETE icosphere (every vertex a node: d_rank a permutation, no
 * mask) -- what MultiScaleEdges builds on TriNodes (edges/builder.py:412-420) -- in one call and without
 * a host round trip: the edge count is a closed form, ag_multiscale_complete_edges(levels, nlevels) =
 * sum of 60 * 4^l (docs/graphs/edges/tri_refined_edges.csv), which sizes d_out_src / d_out_tgt.
 * Workspace: ag_multiscale_workspace_bytes(R, nlevels, 1). */
int64_t ag_multiscale_complete_edges(const int32_t* levels, int nlevels);
int ag_multiscale_complete(int R, const int32_t* levels, int nlevels, const int32_t* d_rank,
                           const uint64_t* d_level_edges, int32_t* d_out_src, int32_t* d_out_tgt, void* d_ws,
                           size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------
 * Edge attributes.
 * Replaces EdgeLength.get_raw_values -> utils.haversine_distance (edges/attributes.py:144-148,
 * utils.py:99-103; float32 arithmetic) and EdgeDirection.get_raw_values ->
 * directional_edge_features (edges/attributes.py:100-104, edges/directional.py:40-94; float32
 * unit vectors, float64 rotation).  Either output may be NULL.  flags: bit0 = rotated
 * direction features (luse_rotated_features); bit1 = d_dir points to float64 [E,2] and receives the
 * directions before their rounding to float32 (the reference normalises the float64 array and rounds
 * afterwards, edges/attributes.py:33-40: `unit-range` needs that, see ag_normalise_f64).
 *   d_len  float32 [E]     great-circle angle in radians
 *   d_dir  float32 [E,2]   (float64 [E,2] with AG_ATTR_DIR_F64)
 * Edge ids are checked against ns / nt: an id outside its node set -- numpy raises IndexError for it in the
 * reference (edges/attributes.py:144-148) -- is not followed, the values of that edge are meaningless and
 * *d_bad (nullable, int32, device) is set to 1.  The library never clears *d_bad: the caller zeroes it,
 * and reads it when it next synchronises.
 */
#define AG_ATTR_ROTATED 1
#define AG_ATTR_DIR_F64 2
int ag_edge_attrs(const float* d_src_x, int64_t ns, const float* d_tgt_x, int64_t nt, const int32_t* d_edge_src,
                  const int32_t* d_edge_tgt, int64_t E, int flags, float* d_len, float* d_dir, int32_t* d_bad,
                  void* stream);

/* ag_edge_attrs that also reports the range of the lengths it wrote, so that the `unit-max` /
 * `unit-range` normalisation of EdgeLength (normalise.py:45-48: values.max(), values.min()) needs no
 * pass over the values of its own.  d_len_range: ag_len_range_words() 32-bit words, written by the
 * call (no need to clear them): [0] = bits of the largest length, [1] = complement of the bits of
 * the smallest, the rest scratch; a NaN length makes both NaN, as numpy's max / min do.
 * NULL = plain ag_edge_attrs. */
size_t ag_len_range_words(void);
int ag_edge_attrs_range(const float* d_src_x, int64_t ns, const float* d_tgt_x, int64_t nt, const int32_t* d_edge_src,
                        const int32_t* d_edge_tgt, int64_t E, int flags, float* d_len, float* d_dir,
                        uint32_t* d_len_range, int32_t* d_bad, void* stream);

/* ag_normalise for norm 3 (unit-max), 4 (unit-range) or 0 from a range reported by
 * ag_edge_attrs_range: one pass, no workspace.  Same results as ag_normalise on NaN-free values. */
int ag_normalise_range(float* d_values, int64_t n, int norm, int invert, const uint32_t* d_range, void* stream);

/* Global normalisation of an attribute array in place.
 * Replaces NormaliserMixin.normalise (normalise.py:36-55) + the float32 cast and the
 * optional `1 - v` of EdgeLength.post_process (edges/attributes.py:33-40,150-157).
 * norm: 0 none, 1 l1, 2 l2, 3 unit-max, 4 unit-range, 5 unit-std.  n = number of float32
 * elements (E*M).  The reductions run over ALL elements, in float64. */
size_t ag_normalise_workspace_bytes(int64_t n);
int ag_normalise(float* d_values, int64_t n, int norm, int invert, void* d_ws, size_t ws_bytes, void* stream);

/* The same normalisations of a float64 array, applied in float64 and rounded to float32 once:
 * BaseNodeAttribute.post_process (nodes/attributes.py:45-52) = normalise(values).astype(dtype), and what
 * BaseEdgeAttribute.post_process does to the float64 directions (edges/attributes.py:33-40).  d_out is
 * float32 [n], or float64 [n] when out_f64 != 0 (dtype "float64": no rounding).  Workspace as ag_normalise. */
int ag_normalise_f64(const double* d_values, int64_t n, int norm, void* d_out, int out_f64, void* d_ws,
                     size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------
 * Areas of the spherical Voronoi cells of a node set.
 * Replaces SphericalVoronoi(points, radius=1).calculate_areas() in SphericalAreaWeights
 * (nodes/attributes.py:198-205), points = latlon_rad_to_cartesian of the float32 coordinates: the
 * area of node i is the sum over its Voronoi vertices of the solid angles |2 atan2(det(p, v_k, v_k+1),
 * 1 + p.v_k + v_k.v_k+1 + v_k+1.p)|.  Every cell is built on its own by clipping with the
 * half-spaces of nearby nodes, so the caller supplies candidates and learns which cells are final:
 *  - ag_voronoi_areas_knn: d_nbr[i*K + j] = j-th nearest node of node i, nearest first (ag_knn of
 *    the set on itself; node i itself may be listed);
 *  - ag_voronoi_areas_rows: for the nt nodes d_ids[t], every node within `radius` (radians) of
 *    it, in any order, as rows d_nbr[d_offsets[t] .. d_offsets[t+1]) (ag_radius_count/fill).
 * d_area[i] receives the area and d_reach[i]: 0 = final; > 0 = angular radius within which all
 * nodes must be offered to finish the cell (infinity: not bounded yet); -1 = another node
 * coincides with node i (scipy raises "duplicate generators"); -2 (from _rows only) = the cell needs
 * more than the 48 polygon vertices the kernel holds (_knn, which holds 16, reports such a cell as
 * not final instead).  ag_voronoi_areas_rows writes entries d_ids[t] only.
 */
int ag_voronoi_areas_knn(const float* d_x, int64_t n, const int32_t* d_nbr, int K, double* d_area, double* d_reach,
                         void* stream);
/* Cells that stay unbounded (d_reach = infinity) even when every node was offered reach 90 degrees or
 * more away from their node (the rim of a node set that leaves part of the sphere empty; a handful
 * of nodes): ag_voronoi_areas_far rebuilds the nt cells d_ids[t] in charts centred inside the cell
 * instead of on the node, offering every node.  d_reach[i] = 0 on success, infinity if no chart held
 * the cell (it spans a hemisphere: fewer than three effective neighbours). */
size_t ag_voronoi_far_workspace_bytes(int64_t n);
int ag_voronoi_areas_far(const float* d_x, int64_t n, const int32_t* d_ids, int64_t nt, double* d_area, double* d_reach,
                         void* d_ws, size_t ws_bytes, void* stream);
int ag_voronoi_areas_rows(const float* d_x, int64_t n, const int32_t* d_ids, int64_t nt, const int64_t* d_offsets,
                          const int32_t* d_nbr, double radius, double* d_area, double* d_reach, void* stream);

/* ---------------------------------------------------------------------------------------
 * Edge list utilities.
 * ag_sort_edges: canonical sort of E (source, target) pairs in place, by (target, source) when
 * by_target != 0 else by (source, target).
 * ag_unique_edges: sort by (source, target) and drop duplicates; replaces utils.concat_edges'
 * torch.unique(dim=1) (utils.py:81) on the concatenated list.  *d_num_unique (int64, device).
 */
size_t ag_sort_edges_workspace_bytes(int64_t E);
int ag_sort_edges(int32_t* d_src, int32_t* d_tgt, int64_t E, int by_target, void* d_ws, size_t ws_bytes,
                  void* stream);
int ag_unique_edges(int32_t* d_src, int32_t* d_tgt, int64_t E, int64_t* d_num_unique, void* d_ws,
                    size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------------------
 * Node masking (edges/builder.py:162-193).
 * ag_mask_compact: ids of the true entries of a uint8 mask, ascending -> d_ids, count ->
 * *d_count (int64, device); ag_gather_coords: out[i] = x[ids[i]] (n_x rows in x); ag_remap_ids:
 * v[i] = map[v[i]] (n_map entries in map).
 * Ids that index a table (here and in ag_mark_ids / ag_invert_ids) are checked against the table's
 * length: an id outside [0, length) -- an IndexError of numpy / torch in the reference -- is never
 * followed (the output entry is 0 / -1, or nothing is written) and *d_bad (nullable, int32, device) is
 * set to 1.  The library never clears *d_bad.
 */
size_t ag_mask_compact_workspace_bytes(int64_t n);
int ag_mask_compact(const uint8_t* d_mask, int64_t n, int32_t* d_ids, int64_t* d_count, void* d_ws,
                    size_t ws_bytes, void* stream);
int ag_gather_coords(const float* d_x, int64_t n_x, const int32_t* d_ids, int64_t n, float* d_out, int32_t* d_bad,
                     void* stream);
int ag_remap_ids(int32_t* d_values, int64_t n, const int32_t* d_map, int64_t n_map, int32_t* d_bad, void* stream);

/* Post-processing of node sets (processors/post_process.py:48-61,133-149).
 * ag_mark_ids: d_mask[d_ids[i]] = 1 for every i (the mask of nodes that occur in an edge list;
 *   replaces connected_mask[edges.edge_index[r]] = True, :142-147).
 * ag_invert_ids: d_map[d_ids[j]] = j (old id -> new id for the ids kept by ag_mask_compact; with
 *   ag_remap_ids this replaces the per-edge Python dict lookup of update_edge_indices, :51-61). */
int ag_mark_ids(const int32_t* d_ids, int64_t n, uint8_t* d_mask, int64_t n_mask, int32_t* d_bad, void* stream);
int ag_invert_ids(const int32_t* d_ids, int64_t n, int32_t* d_map, int64_t n_map, int32_t* d_bad, void* stream);

/* Exclusive prefix sum of n int32 values in ONE launch (chained tiles with look-back), the primitive behind the
 * cell starts of every binning and the row offsets of the cut-off query; no counterpart in the reference.
 * d_out holds n + 1 entries (d_out[n] = total).  d_state: ag_scan_i32_state_words(n) 64-bit words that MUST be zero
 * when the call is made (inside the library a kernel that runs upstream anyway clears them).  A tile waits for its
 * predecessors at most spin_limit polls: if the words were stale the wait gives up, every block still ends, and
 * d_out[n] is -1 instead of the total. */
size_t ag_scan_i32_state_words(int64_t n);
int ag_scan_i32(const int32_t* d_in, int32_t* d_out, int64_t n, unsigned long long* d_state, int spin_limit, void* stream);

/* Target column of a fixed-degree edge list: d_out[r * k + j] = r for r < rows, j < k.
 * Replaces the COO row array the reference takes from kneighbors_graph(...).tocoo()
 * (edges/builder.py:80-84: edge_index row 1 = adjacency row = target id). */
int ag_fill_row_ids(int32_t* d_out, int64_t rows, int k, void* stream);

#ifdef __cplusplus
}
#endif
#endif
