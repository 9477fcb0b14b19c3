// First-layer weight gradient through a saved Chebyshev stack.
//
// cheb.0 of the VAE (cheb_VAE.py:264) reads the 3-channel input mesh and is followed by the one-hot
// downsampling D, so the gradient of its output is non-zero only at the n_sel rows D selects
// (1250 of 4998).  Its weight gradient
//     dW_k[ci][co] = sum_{b,v} T_k(L) x [b,v,ci] * dpre[b,v,co]
// therefore needs T_k(L) x only at those rows.  Two kernels:
//   k_cheb_tstack : ONE workgroup per mesh runs the K-order recurrence on the (<= 4 channel) input in
//                   the 160 KB LDS image of the other LDS kernels and stores T_k x of the SELECTED rows only,
//                   as one plane per order: stack [B][K][n_sel + 1][4].  The stores are decoupled from the
//                   thread-owns-vertex recurrence: once an order's slab is complete every thread copies three rows of
//                   the pooled-row order out of LDS (slab[D.col[r]] * deg^1/2), so a wave stores 1 KB of consecutive
//                   bytes and no store is divergent.  (Round 2 stored every vertex as [B][N+1][K][4]: each lane's 16
//                   bytes in a 96-byte-strided line of its own -- 122 MB of partial-line writes per launch for the
//                   7.7 MB k_stack_dw reads, 48.6 us.  The first form of this round let slots own the selected
//                   vertices instead: same bytes, but the permuted ownership lost the conflict-aware ELL order -- LDS
//                   conflict share 0.52, 34.4 us.)
//                   64 workgroups: it leaves 3/4 of the chip to the small kernels of the main chain,
//                   which is where the step engine schedules it.
//   k_stack_dw    : streaming reduction  stack^T * (dout masked by the ReLU sign bytes)  over the
//                   B * n_sel rows (fixed order: per-block partials + one finishing block).
// Replaces the LDS-resident dW kernel for this layer (4 workgroups per mesh each re-running the
// recurrence, 64 + 5 us at the end of the backward critical path) by a 5 us reduction.
#include "common.hpp"
#include "bf16.hpp"

namespace mvh {

struct TstackDims {
  int B, N, K, Cin, n_sel;
};

template <int VPT, int TCT, int PW>
__global__ void __launch_bounds__(TCT)
k_cheb_tstack(const float* __restrict__ p_x, const uint32_t* __restrict__ p_rowinfo, const uint32_t* __restrict__ p_ell,
              const int32_t* __restrict__ p_sel_col, float* __restrict__ p_stack, TstackDims a) {
  constexpr int THREADS = TCT, VS = VPT * THREADS;
  extern __shared__ __align__(16) unsigned char smem[];
  float4* slab = reinterpret_cast<float4*>(smem);     // [VS] scaled t~_k = D^-1/2 T_k x; rows >= N stay zero
  uint4* ellv = reinterpret_cast<uint4*>(slab + VS);  // [VS][PW/4]
  const int mesh = blockIdx.x, tid = threadIdx.x, N = a.N;
  {
    const unsigned pad = (unsigned)N | ((unsigned)N << 16);
    const uint4 pad4 = make_uint4(pad, pad, pad, pad);
    const uint4* src = reinterpret_cast<const uint4*>(p_ell);
    for (int i = tid; i < VS * (PW / 4); i += THREADS) ellv[i] = (i / (PW / 4) < N) ? src[i] : pad4;
  }
  float ka2[VPT];
  float4 R[VPT];
  // The recurrence keeps the thread-owns-vertex layout of the other LDS kernels (vertex = slot: the conflict-aware ELL
  // order applies).  The STORES are decoupled from it: after an order's slab is complete, thread t copies the rows
  // r = t + j THREADS (j < kStoreRows) of the pooled-row order out of LDS -- slab[D.col[r]] times that vertex's deg^1/2 --
  // so a wave stores 1 KB of consecutive bytes of plane k and no store is divergent (three extra LDS reads per thread and
  // order beside 80 gathers).  Rows past n_sel go to the spare row n_sel of the plane.
  constexpr int kStoreRows = (1536 + THREADS - 1) / THREADS;   // n_sel <= 1536 (host check)
  const int prow = a.n_sel + 1;
  float4* const sbase = reinterpret_cast<float4*>(p_stack) + (long long)mesh * a.K * prow;
  const float* xb = p_x + (long long)mesh * N * a.Cin;
  int sv[kStoreRows];
  float sis[kStoreRows];
#pragma unroll
  for (int j = 0; j < kStoreRows; ++j) {
    const int r = tid + j * THREADS;
    sv[j] = p_sel_col[min(r, a.n_sel - 1)];
    const float deg = (float)(p_rowinfo[sv[j]] & 255u);
    sis[j] = deg > 0.f ? __builtin_amdgcn_sqrtf(deg) : 1.0f;
    float t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c) t[c] = (c < a.Cin) ? xb[(long long)sv[j] * a.Cin + min(c, a.Cin - 1)] : 0.f;
    sbase[min(r, a.n_sel)] = make_float4(t[0], t[1], t[2], t[3]);   // plane 0: T_0 x = x
  }
#pragma unroll
  for (int vi = 0; vi < VPT; ++vi) {
    const int v = tid + vi * THREADS;
    const bool valid = v < N;
    const int vl = min(v, N - 1);
    const float deg = valid ? (float)(p_rowinfo[vl] & 255u) : 0.f;
    ka2[vi] = deg > 0.f ? -2.0f * __builtin_amdgcn_rcpf(deg) : 0.f;
    const float s = valid ? (deg > 0.f ? __builtin_amdgcn_rsqf(deg) : 1.0f) : 0.f;
    float t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c)  // branch-free (clamped index, selected afterwards): the vertices' loads overlap
      t[c] = (c < a.Cin) ? xb[(long long)vl * a.Cin + min(c, a.Cin - 1)] : 0.f;
    if (!valid) t[0] = t[1] = t[2] = t[3] = 0.f;
    slab[v] = make_float4(t[0] * s, t[1] * s, t[2] * s, t[3] * s);
    R[vi] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  auto add4 = [](float4& x, const float4& y) {
    x.x += y.x;
    x.y += y.y;
    x.z += y.z;
    x.w += y.w;
  };
  auto gather = [&](int v) {
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int q = 0; q < PW / 4; ++q) {
      const uint4 id = ellv[v * (PW / 4) + q];
      {
        const float4 n0 = slab[id.x & 0xffffu], n1 = slab[id.x >> 16];
        const float4 n2 = slab[id.y & 0xffffu], n3 = slab[id.y >> 16];
        add4(g, n0); add4(g, n1); add4(g, n2); add4(g, n3);
      }
      asm volatile("" ::: "memory");
      {
        const float4 n0 = slab[id.z & 0xffffu], n1 = slab[id.z >> 16];
        const float4 n2 = slab[id.w & 0xffffu], n3 = slab[id.w >> 16];
        add4(g, n0); add4(g, n1); add4(g, n2); add4(g, n3);
      }
      asm volatile("" ::: "memory");
    }
    return g;
  };
  for (int k = 1; k < a.K; ++k) {
    const float sc = (k == 1) ? 0.5f : 1.0f;  // T_1 = L T_0 ; T_k = 2 L T_{k-1} - T_{k-2}
#pragma unroll
    for (int vi = 0; vi < VPT; ++vi) {
      const float4 g = gather(tid + vi * THREADS);
      const float kk = ka2[vi] * sc;
      R[vi] = make_float4(fmaf(kk, g.x, -R[vi].x), fmaf(kk, g.y, -R[vi].y), fmaf(kk, g.z, -R[vi].z),
                          fmaf(kk, g.w, -R[vi].w));
    }
    __syncthreads();  // every gather of t~_{k-1} is done
#pragma unroll
    for (int vi = 0; vi < VPT; ++vi) {
      const int v = tid + vi * THREADS;
      const float4 old = slab[v];
      slab[v] = R[vi];
      R[vi] = old;
    }
    __syncthreads();  // the slab holds t~_k (the next swap is two barriers away: the copies below are safe)
#pragma unroll
    for (int j = 0; j < kStoreRows; ++j) {
      const float4 t = slab[sv[j]];
      sbase[(long long)k * prow + min(tid + j * THREADS, a.n_sel)] = make_float4(t.x * sis[j], t.y * sis[j], t.z * sis[j], t.w * sis[j]);
    }
  }
}

// ---- dW / db from the stack: out[k][ci][co] = sum_rows stack[row][k][ci] * dpre[row][co]
constexpr int kSdwGrid = 256;

struct SdwDims {
  int rows, n_sel, N, K, Cin, Cout;
  int dout_bf16;  // dout (the gradient of the pooled output) is stored as bf16
};

constexpr int kSdwRows = 64;  // rows per LDS tile

__global__ void __launch_bounds__(512)
k_stack_dw(const float* __restrict__ stack, const float* __restrict__ dout, const uint8_t* __restrict__ bits,
           const float* __restrict__ out_mask, const int* __restrict__ sel_col, float* __restrict__ partial,
           SdwDims a) {
  // dpre[row][co] = dout[row][co] where the forward output at the row's fine vertex is > 0.
  // Tiles of 64 rows go through LDS (8 lanes per row, one float4 each; the next tile's loads are
  // issued before the current tile is consumed); thread t < n_w owns dW entry (k, ci, co) and
  // n_w <= t < n_out owns db[co].  (A barrier-free variant with wave-uniform rows and broadcast loads
  // was 2-3x slower: 6 dependent small loads per row and lane.)
  __shared__ float sA[kSdwRows][52];  // [row][k*4 + ci], K <= 12 (+4: bank spread, keeps 16-byte alignment)
  __shared__ float sB[kSdwRows][36];  // [row][co], Cout <= 32
  const int n_w = a.K * a.Cin * a.Cout, n_out = n_w + a.Cout;
  const int t = threadIdx.x, ty = t >> 3, tx = t & 7;
  const int rpb = (a.rows + gridDim.x - 1) / gridDim.x;
  const int r0 = blockIdx.x * rpb, r1 = min(a.rows, r0 + rpb);
  const int co = t < n_w ? t % a.Cout : t - n_w;
  const int kc = t < n_w ? (t / a.Cout / a.Cin) * 4 + (t / a.Cout) % a.Cin : 0;
  const int CQ4 = a.Cout >> 2;
  float4 pa = make_float4(0.f, 0.f, 0.f, 0.f), pb = pa, pa2 = pa;    // (pa2: planes 8 .. 11 of the deeper layers, K > 8)
  auto fetch = [&](int base) {  // this thread's pieces of the tile starting at `base`, into registers
    pa = make_float4(0.f, 0.f, 0.f, 0.f);
    pb = pa;
    pa2 = pa;
    const int r = base + ty;
    if (r >= r1) return;
    const int b = r / a.n_sel, v = sel_col[r - b * a.n_sel];
    if (tx < a.K) pa = *reinterpret_cast<const float4*>(stack + (((long long)b * a.K + tx) * (a.n_sel + 1) + (r - b * a.n_sel)) * 4);
    if (tx + 8 < a.K) pa2 = *reinterpret_cast<const float4*>(stack + (((long long)b * a.K + tx + 8) * (a.n_sel + 1) + (r - b * a.n_sel)) * 4);
    if (tx < CQ4) {
      float4 d = load4_any(dout, (long long)r * a.Cout + tx * 4, a.dout_bf16 != 0);
      if (bits) {
        const uint32_t m = bits[((long long)b * a.N + v) * CQ4 + tx];
        d.x = (m & 1u) ? d.x : 0.f;
        d.y = (m & 2u) ? d.y : 0.f;
        d.z = (m & 4u) ? d.z : 0.f;
        d.w = (m & 8u) ? d.w : 0.f;
      } else if (out_mask) {
        const float4 o = *reinterpret_cast<const float4*>(out_mask + ((long long)b * a.N + v) * a.Cout + tx * 4);
        d.x = o.x > 0.f ? d.x : 0.f;
        d.y = o.y > 0.f ? d.y : 0.f;
        d.z = o.z > 0.f ? d.z : 0.f;
        d.w = o.w > 0.f ? d.w : 0.f;
      }
      pb = d;
    }
  };
  float acc = 0.f;
  fetch(r0);
  for (int base = r0; base < r1; base += kSdwRows) {
    const int nr = min(kSdwRows, r1 - base);
    if (tx < a.K) *reinterpret_cast<float4*>(&sA[ty][tx * 4]) = pa;
    if (tx + 8 < a.K) *reinterpret_cast<float4*>(&sA[ty][(tx + 8) * 4]) = pa2;
    if (tx < CQ4) *reinterpret_cast<float4*>(&sB[ty][tx * 4]) = pb;
    __syncthreads();
    fetch(base + kSdwRows);  // in flight while this tile is consumed
    if (t < n_w) {
      float a0 = 0.f, a1 = 0.f;
      int r = 0;
      for (; r + 1 < nr; r += 2) {
        a0 = fmaf(sA[r][kc], sB[r][co], a0);
        a1 = fmaf(sA[r + 1][kc], sB[r + 1][co], a1);
      }
      if (r < nr) a0 = fmaf(sA[r][kc], sB[r][co], a0);
      acc += a0 + a1;
    } else if (t < n_out) {
      float a0 = 0.f;
      for (int r = 0; r < nr; ++r) a0 += sB[r][co];
      acc += a0;
    }
    __syncthreads();
  }
  // partial tile in the layout of the LDS dW kernel (slab 0, [k][q = co][j = ci], plane K = db), so the
  // step engine's one reduction launch (k_dw_reduce_all) finishes this layer with all the others
  const int tile = (a.K + 1) * a.Cout * 4;
  float* pt = partial + (long long)blockIdx.x * tile;
  if (t < n_w) {
    const int k = t / (a.Cout * a.Cin), ci = (t / a.Cout) % a.Cin;
    pt[(k * a.Cout + co) * 4 + ci] = acc;
  } else if (t < n_out) {
    pt[(a.K * a.Cout + co) * 4] = acc;
  }
}

// workspace: the stack [B][K][n_sel + 1][4] (row n_sel of a plane is the store target of the unselected slots), then the
// per-block partials of the reduction
size_t tstack_stack_floats(int B, int n_sel, int K) { return (size_t)B * (n_sel + 1) * K * 4; }
size_t tstack_ws_floats(int B, int n_sel, int K, int Cin, int Cout) {
  (void)Cin;
  return tstack_stack_floats(B, n_sel, K) + (size_t)kSdwGrid * ((size_t)(K + 1) * Cout * 4) + 64;
}

bool tstack_eligible(const mvh_csr_t* lap, const mvh_csr_t* pool, int N, int Cin, int Cout, int K) {
  const int need = MVH_CSR_NORMALIZED_LAPLACIAN | MVH_CSR_SYMMETRIC;
  if (!lap || !pool || !lap->rowinfo || !lap->ell || (lap->flags & need) != need) return false;
  if (lap->ell_pairs <= 0 || lap->ell_pairs > 8 || (lap->flags & MVH_CSR_ELL_OVERFLOW)) return false;
  if (!pool->sel_inv || !pool->col || pool->n_cols != N || pool->n_rows <= 0) return false;
  if (Cin < 1 || Cin > 4 || Cout < 4 || Cout > 32 || Cout % 4 != 0 || K < 1 || K > 12) return false;
  if (K * Cin * Cout + Cout > 512) return false;
  if (N + 1 > 5120)   // levels of the streaming kernels: cheb_big.hip builds the same stack (launch_tstack dispatches), 16 outputs
    return !dbg().no_big_tstack && K >= 2 && Cout == 16 && cheb_big_eligible(lap, 1, N, Cin, K);
  if (K > 8 || pool->n_rows > 3 * 512) return false;
  if (N + 1 <= 2048) return false;  // only the 160 KB configuration pays: smaller levels keep the LDS dW kernel
  const int pw = lap->ell_pairs > 4 ? 8 : 4;
  if ((size_t)5120 * (16 + pw * 4) > 160 * 1024) return false;
  return true;
}

// stack [B][K][n_sel][4] <- T_k(L) x at the selected rows
int launch_tstack(hipStream_t st, const mvh_csr_t* lap, const mvh_csr_t* pool, const float* x, float* stack, int B,
                  int N, int Cin, int K) {
  if (N + 1 > 5120) return launch_big_tstack(st, lap, pool, x, stack, B, N, Cin, K);
  TstackDims d{B, N, K, Cin, pool->n_rows};
  const size_t lds = (size_t)5120 * (16 + 4 * 4);
  // 512 threads x 10 vertices; debug switch tstack_tall: 1024 x 5 (61 VGPRs, 4 waves per SIMD) -- MEASURED the same within
  // the noise of one box (550.1 vs 546.9 us per step over four alternating runs)
  const bool wide = dbg().tstack_tall == 0;
  auto kern = wide ? k_cheb_tstack<10, 512, 4> : k_cheb_tstack<5, 1024, 4>;
  static LdsAttr attr[2];   // (one per kernel)
  if (int rc = attr[wide ? 1 : 0].ensure(reinterpret_cast<const void*>(kern), lds)) return rc;
  hipLaunchKernelGGL(kern, dim3(B), dim3(wide ? 512 : 1024), lds, st, x, lap->rowinfo, lap->ell, pool->col, stack, d);
  MVH_LAUNCH_CHECK();
  return MVH_OK;
}

// ---- the layer's own forward from the stack: pooled[b][r][0..16) = relu(bias + sum_k sum_ci stack[b][k][r][ci] W[k][ci][:])
// at the selected rows only, in k_cheb_contract's fma order (k outer, input channel inner), + the sign bytes of those rows at
// their FINE vertex (what k_stack_dw masks with).  One thread per pooled row.
__global__ void __launch_bounds__(256)
k_stack_contract(const float* __restrict__ stack, const float* __restrict__ W, const float* __restrict__ bias,
                 const int* __restrict__ sel_col, float* __restrict__ pooled, uint8_t* __restrict__ bits, int rows,
                 int n_sel, int N, int K, int Cin) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const int b = r / n_sel, pr = r - b * n_sel;
  float acc[16];
#pragma unroll
  for (int co = 0; co < 16; ++co) acc[co] = bias ? bias[co] : 0.f;
  for (int k = 0; k < K; ++k) {
    const float4 t4 = *reinterpret_cast<const float4*>(stack + (((long long)b * K + k) * (n_sel + 1) + pr) * 4);
    const float tv[4] = {t4.x, t4.y, t4.z, t4.w};
    for (int ci = 0; ci < Cin; ++ci) {
      const float* w = W + ((long long)k * Cin + ci) * 16;
#pragma unroll
      for (int co = 0; co < 16; ++co) acc[co] = fmaf(tv[ci], w[co], acc[co]);
    }
  }
#pragma unroll
  for (int co = 0; co < 16; ++co) acc[co] = fmaxf(acc[co], 0.f);
  float* o = pooled + (long long)r * 16;
#pragma unroll
  for (int co = 0; co < 16; co += 4) *reinterpret_cast<float4*>(o + co) = make_float4(acc[co], acc[co + 1], acc[co + 2], acc[co + 3]);
  if (bits) {
    const long long vb = ((long long)b * N + sel_col[pr]) * 4;
#pragma unroll
    for (int co = 0; co < 16; co += 4)
      bits[vb + co / 4] = (uint8_t)((acc[co] > 0.f ? 1 : 0) | (acc[co + 1] > 0.f ? 2 : 0) | (acc[co + 2] > 0.f ? 4 : 0) |
                                    (acc[co + 3] > 0.f ? 8 : 0));
  }
}

int launch_stack_contract(hipStream_t st, const mvh_csr_t* pool, const float* stack, const float* W, const float* bias,
                          float* pooled, uint8_t* bits, int B, int N, int Cin, int K) {
  MVH_REQUIRE(stack && W && pooled && ((uintptr_t)pooled & 15) == 0 && ((uintptr_t)stack & 15) == 0, "stack_contract: bad tensor");
  const int rows = B * pool->n_rows;
  hipLaunchKernelGGL(k_stack_contract, dim3(cdiv(rows, 256)), dim3(256), 0, st, stack, W, bias, pool->col, pooled, bits, rows,
                     pool->n_rows, N, K, Cin);
  MVH_LAUNCH_CHECK();
  return MVH_OK;
}

// dW [K][Cin][Cout], db [Cout] from the stack and the gradient of the POOLED output dout [B][n_sel][Cout]
int launch_stack_dw(hipStream_t st, const mvh_csr_t* pool, const float* stack, const float* dout, const uint8_t* bits,
                    const float* out_mask, float* dW, float* db, float* partial, int B, int N, int Cin, int Cout, int K,
                    DwReduceEntry* defer, bool dout_bf16) {
  SdwDims d{B * pool->n_rows, pool->n_rows, N, K, Cin, Cout, dout_bf16 ? 1 : 0};
  int grid = (d.rows + kSdwRows - 1) / kSdwRows;
  if (grid > kSdwGrid) grid = kSdwGrid;
  hipLaunchKernelGGL(k_stack_dw, dim3(grid), dim3(512), 0, st, stack, dout, bits, out_mask, pool->col, partial, d);
  MVH_LAUNCH_CHECK();
  // the partial tiles are summed by the dW reduction kernel (fixed order over the blocks)
  const DwReduceEntry ent{partial, grid, 1, K, Cout, Cin, 1, Cin, Cout, db ? 1 : 0, dW, db};
  if (defer) {
    *defer = ent;
    return MVH_OK;
  }
  DwReduceTable t;
  t.n = 1;
  t.e[0] = ent;
  return launch_dw_reduce_all(st, t);
}

}  // namespace mvh
