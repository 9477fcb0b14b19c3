/* meshvae_host.h -- C ABI of libmeshvae_host.so: the HOST half of the hierarchy generator
 * (SURVEY 8(f) "next" #1: reference mesh_operations.py:45-70, 87-199, 202-250).
 *
 * Runs once per template, before any step: plain C++ (g++, no HIP, no GPU needed), plain host
 * pointers and sizes.  It holds the three parts of the generator that are loops over mesh elements:
 * the quadric accumulation, the heap edge-collapse decimator and the closest-point search behind the
 * upsampling matrices.  The two LAPACK calls of the reference (the per-face np.linalg.svd that yields
 * a face's plane, mesh_operations.py:58-61, and the per-vertex np.linalg.lstsq that yields the
 * barycentric weights, :229-243) stay with numpy on the Python side of mesh-vae_amd/mesh_operations.py:
 * their last bits decide collapse ties on coplanar neighbourhoods, so they must be the reference's own calls.
 *
 * Every function returns 0 on success, a negative code otherwise (MVHH_ERR_*); nothing is allocated
 * for the caller, every output buffer is caller-provided with the stated size.  All arithmetic is
 * IEEE double without contraction (-ffp-contract=off), written in the evaluation order stated per
 * function so that results do not depend on a BLAS build.
 */
#ifndef MESHVAE_HOST_H
#define MESHVAE_HOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVHH_OK 0
#define MVHH_ERR_INVALID (-1)     /* bad sizes / indices out of range */
#define MVHH_ERR_EXHAUSTED (-2)   /* the edge queue ran dry before n_keep vertices were reached
                                     (the reference raises IndexError from heapq.heappop there) */

#define MVHH_ABI_VERSION 100
int32_t mvhh_version(void);

/* Undirected edges of a triangle list, each once as (r, c) with r < c, in the reference's queue
 * order (mesh_operations.py:33-43 and :109-111, 137-144: the COO entries of the symmetrised CSC pair
 * matrix, i.e. sorted by column c, then row r).  edges_out has room for 3 * n_faces pairs;
 * *n_edges_out receives the count. */
int32_t mvhh_unique_edges(const int64_t* faces, int64_t n_faces, int64_t n_verts,
                          int64_t* edges_out, int64_t* n_edges_out);

/* Per-vertex quadrics (mesh_operations.py:45-70): q[v] = sum over the faces f containing v, in
 * face-major order (f ascending, corner 0,1,2), of planes[f] planes[f]^T.  planes [n_faces][4]
 * (unit-normal plane equations from the caller's SVD), q_out [n_verts][16] row-major. */
int32_t mvhh_vertex_quadrics(const double* planes, const int64_t* faces, int64_t n_faces,
                             int64_t n_verts, double* q_out);

/* Optional: the caller's CBLAS for the pair cost.  The reference evaluates p^T Q p as two numpy dots (mesh_operations.py:
 * 121-122), which numpy turns into cblas_dgemv(RowMajor, Trans, 4, 4, 1, Q, 4, p, 1, 0, t, 1) and cblas_ddot(4, t, 1, p, 1);
 * on exactly coplanar neighbourhoods (a subdivided template) both costs of a pair are rounding noise around zero and the
 * collapse DIRECTION follows the last bits of that BLAS build's kernels (FMA, lane order).  With the two entry points of the
 * BLAS the caller's numpy is linked to, the decimator reproduces them; with blas == NULL it uses the plain evaluation
 * order stated below (same D on every fixture; on the subdivided template the surviving faces then come out in another
 * rotation / order, which moves closest-point ties in U). */
typedef struct mvhh_blas {
  void* cblas_dgemv;   /* void (int order, int trans, blasint m, blasint n, double alpha, const double* a, blasint lda,
                                const double* x, blasint incx, double beta, double* y, blasint incy) */
  void* cblas_ddot;    /* double (blasint n, const double* x, blasint incx, const double* y, blasint incy) */
  int32_t ilp64;       /* blasint is int64_t (numpy's scipy_openblas64_) instead of int32_t */
} mvhh_blas_t;

/* QSlim-style vertex-pair decimation (mesh_operations.py:87-199) down to n_keep vertices.
 *   verts [n_verts][3], faces [n_faces][3], q [n_verts][16] (in/out: the quadrics, updated as
 *   collapsed pairs inherit their sum, :180-181), edges [n_edges][2] in queue order (mvhh_unique_edges).
 * Reproduces the reference's collapse ORDER: a binary heap with CPython heapq's sift procedures over
 * records ordered as the tuple (cost, r, c), records renamed in place when a vertex vanishes (:170-175;
 * found through per-vertex record lists instead of two scans of the whole queue), stale costs
 * re-queued (:152-156), collapse onto the cheaper endpoint without moving it (:161-166).
 *   cost(r, c) = min(p_r^T Q p_r, p_c^T Q p_c), Q = q[r] + q[c], p = (x, y, z, 1), evaluated as
 *   t_j = ((p0 Q0j + p1 Q1j) + p2 Q2j) + p3 Q3j, then ((t0 p0 + t1 p1) + t2 p2) + t3 p3.
 * faces_out [n_faces][3] receives the surviving (non-degenerate) faces in their original order with
 * ORIGINAL vertex numbering; *n_faces_out their count; *n_collapses_out (may be NULL) the number of
 * collapses performed; blas: see mvhh_blas_t (NULL = built-in arithmetic). */
int32_t mvhh_qslim_decimate(const double* verts, int64_t n_verts, const int64_t* faces, int64_t n_faces,
                            double* q, const int64_t* edges, int64_t n_edges, int64_t n_keep,
                            int64_t* faces_out, int64_t* n_faces_out, int64_t* n_collapses_out,
                            const mvhh_blas_t* blas);

/* Closest point on a triangle mesh for every query point (the AABB-tree query behind
 * mesh_operations.py:208-209; psbody's `nearest` returns the same triple).
 *   sv [n_sv][3], sf [n_sf][3] source mesh; pts [n_pts][3].
 *   face_out [n_pts], region_out [n_pts] (0 interior, 1/2/3 edge ab/bc/ca, 4/5/6 vertex a/b/c),
 *   hit_out [n_pts][3].
 * The result is exactly that of testing every triangle (Ericson 5.1.5 per pair, squared distance as
 * ((dx^2 + dy^2) + dz^2), first minimum = lowest face index wins); a bounding-volume hierarchy over
 * the triangles only skips boxes that cannot hold the minimum.  exhaustive != 0 forces the full scan
 * (it is also taken by itself when a triangle is degenerate or a distance is not a number). */
int32_t mvhh_closest_points(const double* sv, int64_t n_sv, const int64_t* sf, int64_t n_sf,
                            const double* pts, int64_t n_pts, int32_t exhaustive,
                            int64_t* face_out, int64_t* region_out, double* hit_out);

#ifdef __cplusplus
}
#endif
#endif
