/* meshvae_hip.h -- C ABI of libmeshvae_hip.so (MI355X / gfx950).
 *
 * The drop-in boundary for the Mesh-VAE ChebConv-VAE hot path.  Every entry point
 * takes plain device pointers + sizes + a HIP stream (no torch types) and replaces
 * the arithmetic of one reference interface, cited per function as file:line under
 * the reference tree.  The reference is pure Python on torch/torch_scatter, so the
 * "FFI" a maintainer adds is a ctypes stub; the host-side mirror
 * of the reference's own classes lives in mesh-vae_amd/{nn,models,model.py,logpdf.py}.
 *
 * Conventions
 *   - all tensors are dense, contiguous, row-major fp32 unless stated; activations
 *     are [B, N, C] (batch, vertex, channel) exactly as the reference's modules
 *     receive them (nn/conv.py:557, nn/pool.py:17);
 *   - sparse operators are CSR over OUTPUT rows with int32 indices; the order of a
 *     row's entries is the order of the reference's COO edge list, so sums run in
 *     the reference's accumulation order;
 *   - `stream` is a hipStream_t (0 = default stream); all work is asynchronous on it;
 *   - return value 0 = success, anything else = failure with a message available
 *     from mvh_last_error() (thread-local).  Nothing here allocates device memory:
 *     workspaces are caller-provided and sized by the *_ws_bytes queries, so every
 *     call is hipGraph-capturable.
 */
#ifndef MESHVAE_HIP_H
#define MESHVAE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mvh_stream_t;

#define MVH_OK 0
#define MVH_ERR_INVALID 1
#define MVH_ERR_HIP 2
#define MVH_ERR_UNSUPPORTED 3

#define MVH_ACT_NONE 0
#define MVH_ACT_RELU 1

/* CSR operator over output rows: y[r] = sum_{e in [rowptr[r], rowptr[r+1])} val[e] * x[col[e]]. */
typedef struct mvh_csr {
  int32_t n_rows;        /* output rows  */
  int32_t n_cols;        /* input rows   */
  int32_t nnz;
  const int32_t* rowptr; /* [n_rows+1] device */
  const int32_t* col;    /* [nnz]      device */
  const float* val;      /* [nnz]      device */
  /* Optional compact form for the LDS-resident kernels (NULL / 0 = not available, the
   * general kernels are used).  rowinfo[r] = (rowptr[r] << 8) | row_length.  ell is the
   * column list in padded ELL form, two uint16 columns per word, vertex-major:
   * ell[r * PW + p] = col(r, 2p) | col(r, 2p+1) << 16 with PW = 4 words per row when
   * ell_pairs <= 4, else 8; out-of-row slots are set to n_cols (an all-zero dummy row the
   * kernels keep in LDS).  The kernels copy it to LDS with 16-byte loads.
   * With MVH_CSR_ELL_OVERFLOW the list holds only the FIRST 8 columns of a row (ell_pairs = 4) and
   * rows with 9..12 entries continue in col[rowptr[r] + 8 ..]: a decimated level has 2-6 % of such
   * vertices, which would otherwise double the padded width (and the LDS gather work) for everyone. */
  const uint32_t* rowinfo; /* [n_rows] device */
  const uint32_t* ell;     /* [n_rows * PW] device, 16-byte aligned */
  int32_t ell_pairs;       /* ceil(max_row_nnz / 2) */
  int32_t max_row_nnz;
  int32_t flags;           /* MVH_CSR_* */
  /* Entries only touch rows/cols < n_active (0 = unknown, treat as n_rows).  When a Laplacian is
   * mostly empty (the final layer applies the 20-vertex edge list to 4998 vertices,
   * cheb_VAE.py:288) `sub` may point at the same operator restricted to its leading
   * n_active x n_active block: the convolution then splits into that small problem plus a
   * per-vertex linear map for the isolated rows (T_k(0) = cos(k pi/2)). */
  int32_t n_active;
  const struct mvh_csr* sub;
  /* MVH_CSR_SELECTION operators (the one-hot downsampling matrices D, nn/pool.py): every row has
   * exactly one entry of value 1 and no column repeats; sel_inv[c] = the row that selects column
   * c, or -1.  Lets the step engine fold the pooling into the neighbouring convolutions. */
  const int32_t* sel_inv; /* [n_cols] device */
  /* Optional vertex-patch plan of a level's Laplacian (NULL = none): lets the 16 -> 16 convolutions of a 2 049 ..
   * 5 119-vertex level run as (mesh, vertex patch) workgroups with all channels on chip and the K Cin x Cout
   * contraction on v_mfma_f32_16x16x4_f32 (csrc/cheb_patch.hip) instead of (mesh, 4-channel slab) workgroups. */
  const struct mvh_patch_plan* patch;
} mvh_csr_t;

/* Vertex-patch plan (built on the host by meshvae_hip/patches.py; every array on the device).  A level's vertices are
 * partitioned into n_patches EXCLUSIVE sets; patch p additionally carries the rows a fused pooling needs (its CORE) and
 * n_rings breadth-first rings around the core -- what the recurrence T_k = 2 L T_{k-1} - T_{k-2} (nn/conv.py:568-572)
 * of a layer with K - 1 <= n_rings needs to be exact on the core.  Local numbering of a patch: exclusive vertices,
 * rest of the core, ring 1, ..., ring n_rings; padded to whole 16-vertex tiles.
 *   poff[p]            first local slot of patch p in pinfo / ell (multiples of 16); poff[n_patches] = total slots
 *   cnt[p][0]          exclusive vertices;  cnt[p][1 + r] = local vertices of ring <= r (r = 0: the core)
 *   pinfo[slot]        global vertex id | degree << 16 | ring << 24 | exclusive << 28  (pad slots: ring 15)
 *   ell[slot][4]       8 neighbours as LOCAL ids x 5 (LDS row stride in 16-byte units), two uint16 per word;
 *                      pad = 5 x (slots of the patch) = the zero row the kernels keep behind the last tile
 *   pooling (n_pool_rows > 0: rows of a pooling operator's transpose, nn/pool.py:17-20 backward, e.g. U^T):
 *   prow_off[p]        first assigned coarse row of patch p in prow_gid; rows prow_off[p] .. prow_off[p+1]
 *   prow_gid[i]        coarse row id;  prow_ptr[prow_off[p] + p + i .. + 1] = its entries in pcol / pval
 *   pcol / pval        LOCAL column (a core vertex of the patch) and value, in the operator's own entry order */
typedef struct mvh_patch_plan {
  int32_t n_patches, n_rings, n_vertices;
  int32_t max_rows, max_core, max_excl;   /* largest patch: padded slots, core vertices, exclusive vertices */
  int32_t n_pool_rows, min_core;          /* min_core: core vertices of the smallest patch */
  const int32_t* poff;
  const int32_t* cnt;      /* [n_patches][n_rings + 2] */
  const uint32_t* pinfo;
  const uint32_t* ell;
  const int32_t* prow_off;
  const int32_t* prow_gid;
  const int32_t* prow_ptr;
  const int32_t* pcol;
  const float* pval;
  const int32_t* pool_rowptr; /* rowptr of the pooling operator the rows above were taken from (identity check), or NULL */
  int32_t max_pool_nnz;       /* most pooling entries (pcol / pval) of one patch */
  int32_t u_rows;             /* > 0: urec is there; rows of the COARSE level the un-pooling operator reads (its n_cols) */
  /* Optional (plans built with the level's un-pooling operator U, nn/pool.py:17-20 forward, whose rows all have <= 3
   * entries): urec[slot][6] = the three (coarse row id, fp32 weight bits) pairs of U's row for the local vertex, in the
   * operator's own entry order, zero-weight pairs past the row's end.  The forward kernel of the level's 16 -> 16 layer
   * can then take the COARSE tensor as its input and un-pool it while it loads (ConvIO::x_unpool inside the library:
   * the step engine's last decoder stage), in the arithmetic of mvh_pool_fwd (one rounding per product and per sum). */
  const uint32_t* urec;
} mvh_patch_plan_t;

/* val[e] == -d[row] * d[col] with d = rowlen^-1/2 (0 for empty rows): the normalised mesh
 * Laplacian of ChebConv_batch.norm (nn/conv.py:541-555) on unit edge weights. */
#define MVH_CSR_NORMALIZED_LAPLACIAN 1
/* the operator equals its transpose (same pattern, same values) */
#define MVH_CSR_SYMMETRIC 2
#define MVH_CSR_SELECTION 4
#define MVH_CSR_ELL_OVERFLOW 8

/* ABI version of this header: 100 * round + revision.  It changes whenever a struct layout or an argument list
 * changes incompatibly; a binding must compare mvh_version() with the MVH_ABI_VERSION it was written against before
 * any other call (the Python binding does, meshvae_hip/__init__.py).  History: 100 = round 1; 300 = `storage`
 * inserted into mvh_vae_desc_t, skip_lo / skip_hi appended to mvh_adam_step / mvh_adam_step_counted (round 2,
 * shipped unversioned), version check introduced (round 3). */
#define MVH_ABI_VERSION 321
int mvh_version(void);
const char* mvh_last_error(void);
/* Device properties of the current HIP device (arch string e.g. "gfx950"). */
int mvh_device_info(int* n_cu, int* lds_bytes_per_cu, char* arch, int arch_len);
/* Debug / A-B switches (no reference counterpart).  They live in ONE struct that is filled when the
 * library is loaded from MESHVAE_DEBUG="key=value,..." and is never re-read from the environment;
 * keys: force_generic, l0_wide, side_prio, no_side, no_tstack, tail_main, fork_batch, fork_small,
 * l0_lane, l0_lane_any, l0_lane_bf, l0_hold, enc_dense (the level-0 lane of the train step),
 * no_gstack_mfma, no_dw_mfma, no_xcd_remap, no_prefetch, no_l0h, no_head_fuse, no_big, no_dx_tstack,
 * no_dx_first, no_bwd_fused, no_dw_rows, keep_enc_out, dw_lane2, tstack_tall,
 * prefetch_at, dw_tie_x, no_src3, no_final_fuse, skip_conv_dw (timing only: results invalid), roctx (roctx ranges per layer of the step for rocprofv3 --marker-trace).  mvh_debug_set changes one switch in-process (the tests run
 * both kernel families that way); mvh_debug_get returns its value, -1 for an unknown key. */
int mvh_debug_set(const char* key, int32_t value);
int32_t mvh_debug_get(const char* key);

/* ---- row P: MessagePassing.propagate (nn/conv.py:242-331 with __collect__ :171-229,
 * message :579-581, aggregate :346-364).  y[b,r,:] = add[b,r,:] + alpha * sum_e val[e] *
 * x[b,col[e],:] + beta * z[b,r,:].  add / z may be NULL.  exact != 0 rounds every product
 * and every add separately in edge order (bit-identical to the reference's fp32
 * index_select -> mul -> scatter_add_); exact == 0 allows FMA contraction. */
int mvh_spmm(mvh_stream_t stream, const mvh_csr_t* op, const float* x, float* y,
             const float* add, const float* z, float alpha, float beta,
             int32_t B, int32_t C, int32_t exact);

/* ---- row S: SurfacePool.forward (nn/pool.py:17-20): y[B,n_rows,C] = P x[B,n_cols,C],
 * bit-exact (D: pure row gather; U: 3 taps in COO order). */
int mvh_pool_fwd(mvh_stream_t stream, const mvh_csr_t* pool, const float* x, float* y,
                 int32_t B, int32_t C);
/* Backward of row S (autograd of index_select/mul/scatter_add in the reference):
 * dx[B,n_cols,C] = P^T dy; `pool_t` is the CSR of P^T (rows = n_cols of P). */
int mvh_pool_bwd(mvh_stream_t stream, const mvh_csr_t* pool_t, const float* dy, float* dx,
                 int32_t B, int32_t C);

/* ---- rows C + Q: ChebConv_batch.forward (nn/conv.py:557-577) (+ F.relu, cheb_VAE.py:264,285).
 * out[B,N,Cout] = act( sum_k T_k(L) x W[k] + bias ), T_0 = x, T_1 = L x,
 * T_k = 2 L T_{k-1} - T_{k-2}.  `lap` is the CSR (by target vertex) of the edge list
 * with the norm of nn/conv.py:541-555 as values; lap->n_rows == lap->n_cols == N, and
 * rows without edges are simply empty (the final-layer quirk, cheb_VAE.py:288).
 * W is [K,Cin,Cout]; bias [Cout] or NULL.  `ws` must hold mvh_cheb_conv_ws_bytes().
 * If `tx_saved` is non-NULL it receives T_1..T_{K-1} for the backward: (K-1)*B*N*Cin floats in a layout private to
 * the library ([K-1,B,N,Cin] rows, or pair-major planes [K-1,B,Cin/2,N,2] where csrc/cheb_big.hip builds the
 * stack); hand the buffer to mvh_cheb_conv_bwd unchanged. */
size_t mvh_cheb_conv_ws_bytes(int32_t B, int32_t N, int32_t Cin, int32_t Cout, int32_t K);
int mvh_cheb_conv_fwd(mvh_stream_t stream, const mvh_csr_t* lap, const float* x, const float* W,
                      const float* bias, float* out, float* tx_saved,
                      int32_t B, int32_t N, int32_t Cin, int32_t Cout, int32_t K, int32_t act,
                      void* ws, size_t ws_bytes);
/* Backward of rows C + Q (the reference relies on autograd; analytic form: SURVEY 7.6).
 * dout is the gradient w.r.t. the activated output; `out` is the forward output (used as
 * the ReLU mask when act == MVH_ACT_RELU, may be NULL otherwise).  `lap_t` is the CSR of
 * L^T (== lap for the symmetric mesh Laplacian).  tx_saved may be NULL (T_k recomputed).
 * dx may be NULL (first layer).  dW [K,Cin,Cout] and db [Cout] (NULL if no bias) are
 * overwritten (not accumulated); dW may be NULL for a dX-only call (dW and dX are independent
 * and may run on different streams with separate workspaces). */
size_t mvh_cheb_conv_bwd_ws_bytes(int32_t B, int32_t N, int32_t Cin, int32_t Cout, int32_t K);
int mvh_cheb_conv_bwd(mvh_stream_t stream, const mvh_csr_t* lap, const mvh_csr_t* lap_t,
                      const float* x, const float* W, const float* out, const float* dout,
                      const float* tx_saved, float* dx, float* dW, float* db,
                      int32_t B, int32_t N, int32_t Cin, int32_t Cout, int32_t K, int32_t act,
                      void* ws, size_t ws_bytes);
/* The fused conv + ReLU of cheb_VAE.py:264/:285 with the ReLU SIGNS kept as one byte per vertex and
 * four output channels (relu_signs [B,N,Cout/4]: bit j of byte c/4 = out[b,v,c+j] > 0; Cout % 4 == 0):
 * the forward writes them from its epilogue, the backward reads them instead of the fp32 output
 * (1/16 of the bytes).  `out` must still be passed to the backward (fallback paths use it). */
int mvh_cheb_conv_fwd_signs(mvh_stream_t stream, const mvh_csr_t* lap, const float* x, const float* W,
                            const float* bias, float* out, uint8_t* relu_signs,
                            int32_t B, int32_t N, int32_t Cin, int32_t Cout, int32_t K,
                            void* ws, size_t ws_bytes);
int mvh_cheb_conv_bwd_signs(mvh_stream_t stream, const mvh_csr_t* lap, const mvh_csr_t* lap_t,
                            const float* x, const float* W, const float* out, const uint8_t* relu_signs,
                            const float* dout, float* dx, float* dW, float* db,
                            int32_t B, int32_t N, int32_t Cin, int32_t Cout, int32_t K,
                            void* ws, size_t ws_bytes);

/* The same two ops on bf16-STORED activations (BASELINE configs[1] "bf16"; reference arithmetic nn/conv.py:557-577):
 * x, out, dout, dx are [B,N,C] tensors of 2-byte bfloat16 (round-to-nearest-even at every store), W / bias / dW / db
 * stay fp32 and every sum is accumulated in fp32 (the recurrence state lives in LDS as fp32).  Cin % 4 == Cout % 4 == 0.
 * act == MVH_ACT_RELU: the forward writes relu_signs [B,N,Cout/4] and the backward reads them (the bf16 output is
 * not needed again); act == MVH_ACT_NONE: relu_signs may be NULL.  Workspaces as for the fp32 ops.  Only layers the
 * LDS-resident kernels can take (N + 1 <= 5120 etc.) are supported: others return MVH_ERR_UNSUPPORTED. */
int mvh_cheb_conv_fwd_bf16(mvh_stream_t stream, const mvh_csr_t* lap, const void* x, const float* W,
                           const float* bias, void* out, uint8_t* relu_signs,
                           int32_t B, int32_t N, int32_t Cin, int32_t Cout, int32_t K, int32_t act,
                           void* ws, size_t ws_bytes);
int mvh_cheb_conv_bwd_bf16(mvh_stream_t stream, const mvh_csr_t* lap, const mvh_csr_t* lap_t,
                           const void* x, const float* W, const uint8_t* relu_signs, const void* dout,
                           void* dx, float* dW, float* db,
                           int32_t B, int32_t N, int32_t Cin, int32_t Cout, int32_t K, int32_t act,
                           void* ws, size_t ws_bytes);

/* ---- strided inputs across the boundary (last row).  The reference's modules hand NON-contiguous views
 * to their arithmetic -- ChebConv_batch.forward transposes to [N, B, C] (nn/conv.py:560, 565, 570), SurfacePool.forward
 * does the same (nn/pool.py:18-20) -- so a caller may hold its activations vertex-major.  These entries take x as a
 * [B, N, C] VIEW given by element strides (x_mesh_stride, x_row_stride, 1): e.g. (C, B*C) for the transpose of an
 * [N, B, C]-physical tensor, (2*N*C, C) for every other mesh of a batch.  No copy is made: the LDS-resident kernels
 * read the rows in place.  Requirements: both strides multiples of Cin (pool: any positive row stride), rows 16-byte
 * aligned when Cin % 4 == 0, x_mesh_stride > 0.  Outputs and gradients are contiguous; everything else is as in
 * mvh_cheb_conv_fwd_signs / _bwd_signs (relu_signs != NULL, act = RELU) or mvh_cheb_conv_fwd / _bwd (relu_signs ==
 * NULL; the backward's ReLU mask is then `out`).  Layers the LDS-resident kernels do not take (N + 1 > 5120, the
 * mostly-isolated final-layer form, Cin not in {3, 8, 16, 32}, ...) return MVH_ERR_UNSUPPORTED and nothing is written:
 * the caller copies x and uses the contiguous entry.  ws: mvh_cheb_conv_strided_ws_bytes for both directions. */
size_t mvh_cheb_conv_strided_ws_bytes(int32_t B, int32_t N, int32_t Cin, int32_t Cout, int32_t K);
int mvh_cheb_conv_fwd_strided(mvh_stream_t stream, const mvh_csr_t* lap, const float* x, int64_t x_mesh_stride,
                              int64_t x_row_stride, const float* W, const float* bias, float* out, uint8_t* relu_signs,
                              int32_t B, int32_t N, int32_t Cin, int32_t Cout, int32_t K, int32_t act,
                              void* ws, size_t ws_bytes);
int mvh_cheb_conv_bwd_strided(mvh_stream_t stream, const mvh_csr_t* lap, const mvh_csr_t* lap_t, const float* x,
                              int64_t x_mesh_stride, int64_t x_row_stride, const float* W, const float* out,
                              const uint8_t* relu_signs, const float* dout, float* dx, float* dW, float* db,
                              int32_t B, int32_t N, int32_t Cin, int32_t Cout, int32_t K, int32_t act,
                              void* ws, size_t ws_bytes);
/* SurfacePool.forward (nn/pool.py:17-20) on such a view: y[B,n_rows,C] (contiguous) = P x, bit for bit
 * mvh_pool_fwd on the contiguous copy (any level; its backward reads only the contiguous dy: mvh_pool_bwd). */
int mvh_pool_fwd_strided(mvh_stream_t stream, const mvh_csr_t* pool, const float* x, int64_t x_mesh_stride,
                         int64_t x_row_stride, float* y, int32_t B, int32_t C);

/* ---- rows E/D (dense parts): nn.Linear + F.relu + nn.Dropout (cheb_VAE.py:270-272,277-280).
 * y[B,out] = drop( act( x[B,in] W[out,in]^T + bias ) ); drop keeps element i when
 * drop_u[i] >= p and scales by 1/(1-p); drop_u == NULL or p == 0 disables it. */
int mvh_linear_fwd(mvh_stream_t stream, const float* x, const float* W, const float* bias,
                   float* y, int32_t B, int32_t in_f, int32_t out_f, int32_t act,
                   const float* drop_u, float p);
/* dx may be NULL; dW (with db) may be NULL instead (dX-only call).  `y` is the forward output
 * (mask for relu/dropout).  ws is unused (kept for ABI stability; may be NULL). */
int mvh_linear_bwd(mvh_stream_t stream, const float* x, const float* W, const float* y,
                   const float* dy, float* dx, float* dW, float* db,
                   int32_t B, int32_t in_f, int32_t out_f, int32_t act, float p,
                   void* ws, size_t ws_bytes);

/* ---- rows K + Z + R: classifier (cheb_VAE.py:253-258), latent heads (:209-226) and
 * reparameterize (:309-319).  h [B,H] (encoder output), y [B,C] one-hot as float,
 * drop_u [B,H] uniforms for the classifier's second dropout (NULL = eval), eps [B,Z]
 * host-drawn N(0,1) (NULL => z = mu, the m_type="test" path).
 * Outputs: y_hat [B,C] softmax, mu/logvar/z [B,Z], zy [B,C+Z] = cat[y, z]. */
int mvh_vae_latent_fwd(mvh_stream_t stream, const float* h, const float* y, const float* drop_u,
                       float p, const float* Wc, const float* bc, const float* Wm, const float* bm,
                       const float* Wv, const float* bv, const float* eps,
                       float* y_hat, float* mu, float* logvar, float* z, float* zy,
                       int32_t B, int32_t H, int32_t C, int32_t Z);
/* Backward: incoming d_yhat [B,C], d_mu/d_logvar [B,Z] (from the loss), d_zy [B,C+Z] (from
 * the decoder).  Produces dh [B,H] and the six parameter gradients (overwritten).
 * ws: B*(C+2Z) floats. */
int mvh_vae_latent_bwd(mvh_stream_t stream, const float* h, const float* y, const float* drop_u,
                       float p, const float* Wc, const float* Wm, const float* Wv,
                       const float* eps, const float* y_hat, const float* logvar,
                       const float* d_yhat, const float* d_mu, const float* d_logvar,
                       const float* d_zy, float* dh, float* dWc, float* dbc, float* dWm,
                       float* dbm, float* dWv, float* dbv,
                       int32_t B, int32_t H, int32_t C, int32_t Z, void* ws, size_t ws_bytes);

/* ---- row L: cheb_VAE.loss_function (cheb_VAE.py:321-346) with logpdf.KLD (logpdf.py:7-8),
 * softclip (:24-28) and gaussian_nll (:22-23).  x_gt is fp32 (gt_f64 == 0) or fp64
 * (gt_f64 != 0, the main.py path); rec/loss are written in that same type.
 * loss (1 element), rec [B], kld [B] fp32, correct: one int64.  ws: mvh_vae_loss_ws_bytes(B). */
size_t mvh_vae_loss_ws_bytes(int32_t B);
int mvh_vae_loss_fwd(mvh_stream_t stream, const float* recon, const void* x_gt, int32_t gt_f64,
                     const float* mu, const float* logvar, const float* y, const float* y_hat,
                     float log_sigma, void* loss, void* rec, float* kld, int64_t* correct,
                     int32_t B, int32_t NV /* vertices*features per mesh */, int32_t C, int32_t Z,
                     void* ws, size_t ws_bytes);
/* Gradients of `loss` (scaled by the upstream scalar g read from device pointer d_loss, fp32
 * or fp64 per gt_f64; NULL means 1): d_recon [B,NV], d_mu, d_logvar [B,Z], d_yhat [B,C]. */
int mvh_vae_loss_bwd(mvh_stream_t stream, const float* recon, const void* x_gt, int32_t gt_f64,
                     const float* mu, const float* logvar, const float* y, const float* y_hat,
                     float log_sigma, const void* d_loss, float* d_recon, float* d_mu,
                     float* d_logvar, float* d_yhat, int32_t B, int32_t NV, int32_t C, int32_t Z);

/* ---- train-loop optimizer (reference main.py:251, :80-81: torch.optim.Adam with coupled L2
 * weight_decay) over the flat parameter buffer.  step_count is a device int32 incremented by
 * the call (so the step is hipGraph-replayable); grad is multiplied by grad_scale first
 * (1/world_size after a sum all-reduce).  Elements [skip_lo, skip_hi) of the buffers are left untouched
 * (skip_lo >= skip_hi: none): torch.optim.Adam skips parameters whose .grad is None -- dec_lin_1,
 * which the forward never uses (cheb_VAE.py:165), keeps its initial values in the reference. */
int mvh_adam_step(mvh_stream_t stream, float* param, const float* grad, float* exp_avg,
                  float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2, float eps,
                  float weight_decay, float grad_scale, int32_t* step_count, int64_t skip_lo,
                  int64_t skip_hi);

/* The same update with the step number t >= 1 counted by the caller (eager launch sequences: saves the counter-tick
 * launch of mvh_adam_step, which exists so that a captured graph can be replayed); *step_count is set to t, so a later
 * mvh_adam_step / graph capture continues from there. */
int mvh_adam_step_counted(mvh_stream_t stream, float* param, const float* grad, float* exp_avg,
                          float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2, float eps,
                          float weight_decay, float grad_scale, int32_t* step_count, int32_t step,
                          int64_t skip_lo, int64_t skip_hi);

/* ---- around the step: what main.py:88-93 / :139-145 do on the HOST with numpy after
 * every batch -- de-normalise the reconstruction (out * std + mean, per vertex), undo the Procrustes
 * alignment of data.py:144 (bmm(mesh * s, R) + m) and take the per-vertex Euclidean distance to the
 * original mesh (inference.py:50-51) -- as one device pass, so the step needs no D2H sync.
 * recon [B,N,3]; std, mean [N,3]; R [B,3,3] row-major; m [B,3]; s [B]; gt [B,N,3] or NULL;
 * mesh_out [B,N,3] or NULL; dist_out [B,N] or NULL (needs gt). */
int mvh_recon_postprocess(mvh_stream_t stream, const float* recon, const float* std, const float* mean,
                          const float* R, const float* m, const float* s, const float* gt, float* mesh_out,
                          float* dist_out, int32_t B, int32_t N);

/* ---- around the step, input side: the Procrustes alignment of every mesh to the
 * template when a dataset is built (data.py:144 -> utils.py:58-157, numpy double on the host there)
 * as two device passes over meshes resident in HBM, with the 3x3 SVD between them left to the
 * caller (LAPACK on 13 numbers per mesh).  All tensors fp64.
 *   tmpl [N,3]: the standardised template mtx1 (centred, unit Frobenius norm; utils.py:136,147).
 *   pts [B,N,3]; stats [B,13] = {centroid[3], norm2, M[3][3]} with M = mtx1^T ((pts-centroid)/norm2),
 *   i.e. the matrix scipy's orthogonal_procrustes(mtx1, mtx2) decomposes (utils.py:151):
 *   U,w,Vt = svd(M); R = U Vt; s = sum(w).
 *   apply: aligned [B,N,3] = ((pts-centroid)/norm2) @ R^T * s (utils.py:152); disparity [B] or NULL
 *   = sum((mtx1 - aligned)^2) (utils.py:155).  R [B,3,3] row-major, s [B]. */
int mvh_procrustes_stats(mvh_stream_t stream, const double* tmpl, const double* pts, double* stats,
                         int32_t B, int32_t N);
int mvh_procrustes_apply(mvh_stream_t stream, const double* tmpl, const double* pts, const double* stats,
                         const double* R, const double* s, double* aligned, double* disparity,
                         int32_t B, int32_t N);

/* MeshData.__getitem__ for a whole batch (data.py:103-111): x64[b] = (data[idx[b]] - mean) / std in fp64
 * (the loss target x_gt) and x32 = its float cast (the network input), gathered from the device-resident
 * aligned dataset data [n_meshes, n3] (n3 = N*3).  Bit-identical to the reference's torch CPU ops.
 * idx [B] int64 on the device; mean, std [n3] fp64; x32 / x64 [B, n3], either may be NULL. */
int mvh_gather_normalize(mvh_stream_t stream, const double* data, int64_t n_meshes, const int64_t* idx,
                         const double* mean, const double* stdv, float* x32, double* x64, int32_t B,
                         int64_t n3);

/* ---- row F: cheb_VAE.forward (cheb_VAE.py:190-251) and loss.backward() (main.py:80) as one
 * native launch sequence.  `desc` describes the model the reference builds in
 * cheb_VAE.__init__ (cheb_VAE.py:106-172): filters = [num_features] + num_conv_filters,
 * K = polygon_order, the per-level operators in the CSR form above.  lap[i] (i < n_layers)
 * is the level-i Laplacian; lap[n_layers] is the COARSEST edge list on the FINEST vertex set
 * (the final layer's quirk, cheb_VAE.py:288).  down[i]/up[i] map level i <-> i+1; *_t are the
 * transposes.  `params` / `grads` are host arrays of mvh_vae_param_count() device pointers in
 * state_dict order (cheb.i.{weight,bias}, cheb_dec.i.{weight,bias}, cheb_dec.n.weight,
 * classifier_layer, z_mean, z_log_var, enc_lin, dec_lin, dec_lin_1, dec_lin_2 .{weight,bias}).
 * y is the one-hot label as float [B,C]; eps [B,Z] host-drawn N(0,1) or NULL (m_type="test");
 * drop_u [B*(3H + flat)] uniforms for the four dropout sites in order (encoder h, classifier,
 * dec_lin, dec_lin_2), NULL = eval.  Activations and their gradients live in `ws`
 * (mvh_vae_step_ws_bytes), which must be passed unchanged from forward to backward.
 * backward overwrites every gradient except dec_lin_1's, which it never touches (the forward never uses
 * that layer, cheb_VAE.py:165, so torch leaves its .grad None: hand in zeros if the span is all-reduced
 * with the rest, and skip it in the optimizer, mvh_adam_step); its weight-gradient kernels run on
 * internal side streams forked from and joined to `stream` with events (hipGraph-capturable).
 * d_loss: device scalar (dtype of the loss) scaling every gradient, or NULL = 1: the forward
 * already leaves the d_loss = 1 gradient seeds of the loss in `ws`, so NULL costs no launch. */
#define MVH_VAE_MAX_LAYERS 8
/* storage of the conv-level activations (and their gradients) between two layers inside `ws`:
 * MVH_STORAGE_F32 = the reference's dtype (default); MVH_STORAGE_BF16 = BASELINE configs[1] "bf16": 2-byte
 * tensors in HBM, every kernel converts at its loads / stores, arithmetic and the Chebyshev recurrence state
 * (LDS) stay fp32, parameters / gradients / Adam state stay fp32.  The network input x, recon, the dense head
 * and the loss are fp32 in both modes.  bf16 exists on the LDS-resident kernels only: a model with a level
 * that needs the general stack pipeline (N + 1 > 5120) returns MVH_ERR_UNSUPPORTED. */
#define MVH_STORAGE_F32 0
#define MVH_STORAGE_BF16 1
typedef struct mvh_vae_desc {
  int32_t n_layers, num_features, num_hidden, num_classes, num_style;
  float dropout_p;
  int32_t storage;
  int32_t filters[MVH_VAE_MAX_LAYERS + 2];
  int32_t K[MVH_VAE_MAX_LAYERS + 1];
  int32_t num_nodes[MVH_VAE_MAX_LAYERS + 1];
  mvh_csr_t lap[MVH_VAE_MAX_LAYERS + 1], lap_t[MVH_VAE_MAX_LAYERS + 1];
  mvh_csr_t down[MVH_VAE_MAX_LAYERS], down_t[MVH_VAE_MAX_LAYERS];
  mvh_csr_t up[MVH_VAE_MAX_LAYERS], up_t[MVH_VAE_MAX_LAYERS];
} mvh_vae_desc_t;

/* struct sizes as compiled (binding self-check) */
size_t mvh_sizeof_vae_desc(void);
size_t mvh_sizeof_csr(void);
size_t mvh_vae_step_ws_bytes(const mvh_vae_desc_t* desc, int32_t B);
int32_t mvh_vae_param_count(const mvh_vae_desc_t* desc);
/* Test aid (no reference counterpart): where one activation / activation-gradient tensor of the step lives in `ws`,
 * so that parity tests can hold the INTERMEDIATE tensors of mvh_vae_forward / mvh_vae_backward against the oracle's,
 * not only the module outputs.  name: "encA" | "encP" | "decU" | "decC" (conv output / pooled output of encoder stage
 * `index`; un-pooled input / conv output of decoder stage `index`), their gradients "g_encA" | "g_encP" | "g_decU" |
 * "g_decC", and (index ignored) "h" "zy" "d1" "d2" "g_h" "g_zy" "g_d1" "g_d2" "g_recon".  Returns the byte offset and
 * stores the element count, or -1 for an unknown name / index.  Which of these a given configuration really writes
 * (fused pooling keeps some tensors in LDS only) is private to the library: the tests state what they read. */
int64_t mvh_vae_ws_offset(const mvh_vae_desc_t* desc, int32_t B, const char* name, int32_t index, int64_t* n_elems);
int mvh_vae_forward(mvh_stream_t stream, const mvh_vae_desc_t* desc, const float* const* params,
                    const float* x, const float* y, const void* x_gt, int32_t gt_f64, const float* eps,
                    const float* drop_u, int32_t B, float log_sigma, void* loss, int64_t* correct,
                    float* recon, float* kld, void* rec, float* z, float* y_hat, float* mu, float* logvar,
                    void* ws, size_t ws_bytes);
int mvh_vae_backward(mvh_stream_t stream, const mvh_vae_desc_t* desc, const float* const* params,
                     float* const* grads, const float* x, const float* y, const void* x_gt, int32_t gt_f64,
                     const float* eps, const float* drop_u, int32_t B, float log_sigma, const void* d_loss,
                     const float* recon, const float* y_hat, const float* mu, const float* logvar,
                     void* ws, size_t ws_bytes, mvh_stream_t side_stream /* NULL = internal */);
/* Optional, BEFORE mvh_vae_forward of a step whose backward will follow (same x, ws, side_stream, same host thread):
 * starts the input-only part of the backward -- T_k(L) x of the first layer at the rows its pooling keeps, what
 * autograd would recompute from x at main.py:80 -- on the weight-gradient lane, forked from `stream` with an event;
 * mvh_vae_backward then only waits for it.  A forward with no backward after it must not be preceded by this call
 * inside a stream capture (the fork would stay unjoined).  No effect on results. */
int mvh_vae_backward_prefetch(mvh_stream_t stream, const mvh_vae_desc_t* desc, const float* x, int32_t B,
                              void* ws, size_t ws_bytes, mvh_stream_t side_stream /* NULL = internal */);

/* The two halves of the forward on their own, for the inference-side callers that use the model piecewise
 * (inference.py:98-131, crecon.py:167-192: net.encoder(x) ... net.sample(y, z)): the same launch sequences as inside
 * mvh_vae_forward, same workspace.  encode: x [B,N,F] -> h [B,num_hidden] (cheb_VAE.py:261-273; drop_u_enc [B,H]
 * uniforms or NULL).  decode: zy [B, num_classes + num_style] -> recon [B,N,F] (cheb_VAE.py:275-292; drop_u NULL or a
 * buffer in the full-step layout [B*H | B*H | B*H | B*flat] of which the last two segments are read). */
int mvh_vae_encode(mvh_stream_t stream, const mvh_vae_desc_t* desc, const float* const* params, const float* x,
                   const float* drop_u_enc, int32_t B, float* h, void* ws, size_t ws_bytes);
int mvh_vae_decode(mvh_stream_t stream, const mvh_vae_desc_t* desc, const float* const* params, const float* zy,
                   const float* drop_u, int32_t B, float* recon, void* ws, size_t ws_bytes);

/* Data-parallel overlap: make `stream` wait until the DENSE-layer weight gradients written by the
 * most recent mvh_vae_backward on the current device (from any host thread: the gradient lanes are per device) are final -- classifier_layer,
 * z_mean, z_log_var, enc_lin, dec_lin, dec_lin_1, dec_lin_2: 98 % of the parameter bytes at default.cfg, and they
 * are complete before the encoder half of the backward starts.  A caller that keeps those parameters contiguous
 * can all-reduce them on `stream` underneath the rest of the backward and only the convolution weights
 * (80 KB) after it.  The reference has no distributed code; this replaces nothing there.  Error if no backward
 * was issued on this device yet (or the last one ran inside a stream capture, where no event is recorded). */
int mvh_vae_wait_dense_grads(mvh_stream_t stream);

/* ---- the asynchronous launcher (csrc/launcher.hip) ---------------------------------------------------------------
 * The reference's train loop (main.py:74-81: optimizer.zero_grad() -> model(...) -> loss.backward() -> optimizer.step())
 * drives this library from ONE Python thread, and on this path that thread is the bottleneck: ~0.35 ms of kernel-launch
 * calls per step inside mvh_vae_forward / mvh_vae_backward, then ~0.35 ms of torch.optim.Adam's Python, for ~0.45 ms of
 * GPU work.  A launcher owns a worker thread and a stream S of its own; mvh_vae_forward_async / mvh_vae_backward_async
 * take the arguments of the synchronous entries (copies of the descriptor and the pointer tables are made), hand the
 * launch sequence to that thread and return at once.  Stream semantics are those of the synchronous call on
 * `user_stream`: the job starts behind everything enqueued on `user_stream` before the call (an event), and
 * `user_stream` continues only once the job's work has finished on the GPU (hipStreamWaitValue64 on a ticket the job's
 * last packet writes) -- so a later consumer of an output, or a later allocation that reuses a freed input, is ordered
 * behind the job.  The CALLER keeps every buffer a job touches allocated until that point (the Python binding holds
 * references to the tensors of the last few jobs).  S is created at the highest stream priority: the runtime pools
 * hardware queues per priority, so S can never sit behind the caller's blocked stream in a shared queue.  The value
 * wait should run on the command processor (environment GPU_STREAMOPS_CP_WAIT=1, read when the HIP runtime starts): the
 * runtime's default is a shader that spins on a compute unit for as long as the job runs (measured: the step's kernels
 * then take 0.75 instead of 0.55 ms); the Python binding only uses the launcher when that variable was in place in time.
 * A job's failure is kept and returned by the next call on the launcher (mvh_last_error carries its message).
 * mvh_launcher_supported: 1 if the current device has hipStreamWaitValue64 (hipDeviceAttributeCanUseStreamWaitValue).
 * mvh_launcher_sync: the calling host thread waits until the worker has enqueued everything handed over so far
 * (host-side only; the GPU work is then ordinary stream work).  The reference has no counterpart: it replaces nothing
 * there, it removes host time from the loop the reference drives. */
typedef struct mvh_launcher mvh_launcher_t;
int mvh_launcher_supported(void);
int mvh_launcher_create(mvh_launcher_t** out);
int mvh_launcher_sync(mvh_launcher_t* launcher);
int mvh_launcher_destroy(mvh_launcher_t* launcher);
/* Test aid (no reference counterpart): queue a job that does no work and, on the worker thread, mode 0 succeeds, 1 throws a C++
 * exception, 2 returns an error code -- in every case its ticket is written (nothing stays blocked on user_stream) and the failure
 * is reported by the next launcher call / mvh_launcher_sync. */
int mvh_launcher_test_job(mvh_launcher_t* launcher, mvh_stream_t user_stream, int32_t mode);
int mvh_vae_forward_async(mvh_launcher_t* launcher, mvh_stream_t user_stream, const mvh_vae_desc_t* desc,
                          const float* const* params, const float* x, const float* y, const void* x_gt, int32_t gt_f64,
                          const float* eps, const float* drop_u, int32_t B, float log_sigma, void* loss, int64_t* correct,
                          float* recon, float* kld, void* rec, float* z, float* y_hat, float* mu, float* logvar,
                          void* ws, size_t ws_bytes);
int mvh_vae_backward_async(mvh_launcher_t* launcher, mvh_stream_t user_stream, const mvh_vae_desc_t* desc,
                           const float* const* params, float* const* grads, const float* x, const float* y,
                           const void* x_gt, int32_t gt_f64, const float* eps, const float* drop_u, int32_t B,
                           float log_sigma, const void* d_loss, const float* recon, const float* y_hat, const float* mu,
                           const float* logvar, void* ws, size_t ws_bytes);

#ifdef __cplusplus
}
#endif
#endif /* MESHVAE_HIP_H */
