// Chebyshev recurrence of the levels that are too big for the (mesh, 4-channel slab) LDS kernels (cheb_lds.hip stops at
// 5119 vertices): ONE launch instead of the K - 1 streaming SpMM passes of the general pipeline (SURVEY row C,
// nn/conv.py:564-572, and the adjoint chain of its backward).  At 19 992 vertices a plane of TWO channels is 160 KB --
// a whole CU's LDS -- so a workgroup owns (mesh, channel pair):
//   * LDS   : u_{k-1} of the pair for every vertex as one float2 (+ a zero row N that the padded ELL slots gather);
//   * VGPRs : u_{k-2} of the thread's own VPT = 20 vertices (vertex v = tid + 1024 j), their degrees (4 bits each);
//   * the neighbour ids (8 x uint16 per vertex, one 16-byte load) come from the L2-resident ELL list every order
//     through a ring of eight vertices' loads in flight per thread;
//   * per order: 8 unweighted ds_read_b64 gathers per vertex, then after a barrier each thread swaps its own rows
//     (registers <-> LDS; MODE 1 adds the order's input piece here, away from the gathers' register pressure) and a
//     second barrier opens the next order.  HBM sees one read of the input piece or one write of the output piece
//     per order -- not two reads and a write of whole [B, N, C] planes plus the gathers.
// L = -D^-1/2 A D^-1/2 is applied in scaled variables u = s t, s = deg^-1/2 (s = 1, no gather for isolated vertices),
// as in cheb_lds.hip:   u_k = -(alpha / deg) sum_{j in N(v)} u_{k-1}[j] - u_{k-2}[v].
// MODE 0  T stack (forward, dW):  T_0 = x;  tx[k-1] = T_k = 2 L T_{k-1} - T_{k-2} (T_1 = L T_0)
// MODE 1  Clenshaw sum (dX):      b_k = G_k + 2 L b_{k+1} - b_{k+2};  out = G_0 + L b_1 - b_2   from the stack G [K][B][N][C]
// The stacks are row layout [B][N][C] or, for the 16 -> 16 layers whose producer / consumer kernels know it, pair-major
// (BigArgs::pm).  The 8 pair-workgroups of a mesh sit on one XCD (blockIdx & 7 = mesh & 7), which helps the single
// row-layout plane each mode still touches (x in, dx out).
#include <type_traits>

#include "common.hpp"

namespace mvh {

constexpr int kBigThreads = 1024, kBigVpt = 20, kBigRing = 8;

struct BigArgs {
  const float* in;      // MODE 0: x [B][N][C];  MODE 1: G [K][B][N][C]
  float* out;           // MODE 0: tx [K-1][B][N][C];  MODE 1: dx [B][N][C]
  const uint32_t* ell;      // [N][4]: eight uint16 neighbour ids per vertex, padded with N
  const uint32_t* rowinfo;  // [N]: (rowptr << 8) | degree
  int B, N, C, K;
  long long plane;      // B * N * C
  const float* mask;    // MODE 0, optional: rows like `in`; an input element counts only where mask > 0 (ReLU backward)
  int with_t0;          // MODE 0: the (masked) input itself is stored as plane 0 and T_k as plane k (K planes)
  int half_ids;         // TIMING ONLY (debug switch big_half_ids): 8 id bytes per vertex and order instead of 16
  const int32_t* sel_inv;   // MODE 0, optional (with_t0, C <= 4, row layout): ONLY the rows a one-hot pooling selects are stored
  int n_sel;                // (sel_inv[v] = pooled row or -1), in the layout of cheb_tstack.hip: out = stack [B][K][n_sel + 1][4]
  int pm;               // the stack (MODE 0: tx, MODE 1: G) is PAIR-MAJOR: plane k = [B][C/2][N][2] (C even), so that a
                        // workgroup streams 8 contiguous bytes per vertex instead of 8-byte pieces of 64-byte rows --
                        // with row layout the 8 pair-workgroups of a mesh drift apart, every one of them pulls the whole
                        // line through L2 again (measured: 833 us for the dX sum at 20k, no better than the 9 SpMM passes)
};

template <bool VEC>
__device__ __forceinline__ float2 big_load2(const float* p, unsigned idx, bool has1) {
  if (VEC) return *reinterpret_cast<const float2*>(p + idx);
  return make_float2(p[idx], has1 ? p[idx + 1] : 0.f);
}
// (nontemporal: the stack planes are written once and read after the whole 0.8 GB stack has passed -- keeping them out
//  of L2 took the T kernel from 238 to 212 us and its consumers' re-reads of x / weights stay cached)
template <bool VEC>
__device__ __forceinline__ void big_store2(float* p, unsigned idx, bool has1, float2 v) {
  if (VEC) {
    typedef float f2nt __attribute__((ext_vector_type(2)));
    __builtin_nontemporal_store((f2nt){v.x, v.y}, reinterpret_cast<f2nt*>(p + idx));
  } else {   // (4-byte pieces of 12-byte rows: plain stores -- nontemporal ones took the 3-channel launch from 108 to 168 us)
    p[idx] = v.x;
    if (has1) p[idx + 1] = v.y;
  }
}

// VEC: the channel count is even, every piece is one aligned float2 (odd counts -- the 3-channel first layer -- go
// through scalar accesses and never use the pair-major layout)
template <int MODE, bool VEC>
__global__ void __launch_bounds__(kBigThreads) k_cheb_big(BigArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  float2* pl = reinterpret_cast<float2*>(smem);   // [N + 1]
  const int tid = threadIdx.x, N = a.N, C = a.C;
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3, pairs = (C + 1) >> 1;
  const int mesh = (jj / pairs) * 8 + xcd, pr = jj % pairs;
  if (mesh >= a.B) return;   // uniform per block, before any barrier
  const int c0 = 2 * pr;
  const bool has1 = c0 + 1 < C;
  // element (mesh, v, c0) of a row-layout tensor = row_base + v * C; of a pair-major plane = pm_base + v * 2
  const long long row_base = ((long long)mesh * N) * C + c0, pm_base = ((long long)(mesh * pairs + pr) * N) * 2;
  const bool in_pm = MODE == 1 && a.pm, out_pm = MODE == 0 && a.pm;
  const float* in_m = a.in + (in_pm ? pm_base : row_base);
  float* out_m = a.out + (out_pm ? pm_base : row_base);
  const int istr = in_pm ? 2 : C, ostr = out_pm ? 2 : C;
  const uint4* ellv = reinterpret_cast<const uint4*>(a.ell);

  // degrees of the own vertices, 4 bits each (<= 8: the launcher checks max_row_nnz)
  uint32_t degp[(kBigVpt + 7) / 8];
#pragma unroll
  for (int w = 0; w < (kBigVpt + 7) / 8; ++w) degp[w] = 0u;
#pragma unroll
  for (int j = 0; j < kBigVpt; ++j) {
    const int v = tid + kBigThreads * j;
    const uint32_t d = v < N ? (a.rowinfo[v] & 255u) : 0u;
    degp[j >> 3] |= d << (4 * (j & 7));
  }
  // (1/deg, deg^+-1/2 below are the hardware's 1-ulp v_rcp / v_rsq / v_sqrt: the IEEE forms cost a dozen instructions
  //  and a branch per vertex and order; the degrees are integers <= 8)
  auto deg_of = [&](int j) -> float { return (float)((degp[j >> 3] >> (4 * (j & 7))) & 15u); };

  float2 P[kBigVpt];   // u_{k-2} (MODE 1: w_{k+2}) of the own vertices; after an order's first phase the new u_k
  // ---- first plane
  if (MODE == 0) {  // u_0 = s x
#pragma unroll
    for (int j = 0; j < kBigVpt; ++j) {
      const int v = tid + kBigThreads * j;
      P[j] = make_float2(0.f, 0.f);
      if (v < N) {
        const float dg = deg_of(j), s = dg > 0.f ? __builtin_amdgcn_rsqf(dg) : 1.f;
        float2 x = big_load2<VEC>(in_m, (unsigned)(v * istr), has1);
        if (a.mask) {
          const float2 mk = big_load2<VEC>(a.mask + row_base, (unsigned)(v * istr), has1);
          x = make_float2(mk.x > 0.f ? x.x : 0.f, mk.y > 0.f ? x.y : 0.f);
        }
        pl[v] = make_float2(s * x.x, s * x.y);
        if (a.sel_inv) {          // compact stack of the selected rows (plane 0 = x)
          const int pr = a.sel_inv[v];
          if (pr >= 0) {
            float* dst = a.out + (((long long)mesh * a.K) * (a.n_sel + 1) + pr) * 4 + c0;
            dst[0] = x.x;
            if (has1) dst[1] = x.y;
          }
        } else if (a.with_t0) big_store2<VEC>(out_m, (unsigned)(v * ostr), has1, x);
      }
    }
  } else {  // w_{K-1} = s G_{K-1}
    const float* gk = in_m + (long long)(a.K - 1) * a.plane;
#pragma unroll
    for (int j = 0; j < kBigVpt; ++j) {
      const int v = tid + kBigThreads * j;
      P[j] = make_float2(0.f, 0.f);
      if (v < N) {
        const float dg = deg_of(j), s = dg > 0.f ? __builtin_amdgcn_rsqf(dg) : 1.f;
        const float2 g = big_load2<VEC>(gk, (unsigned)(v * istr), has1);
        pl[v] = make_float2(s * g.x, s * g.y);
      }
    }
  }
  if (tid == 0) pl[N] = make_float2(0.f, 0.f);
  __syncthreads();

  // MODE 0: orders k = 1 .. K-1 (step s = k);  MODE 1: k = K-2 .. 0 (step s = K-1-k)
  const int steps = a.K - 1;
  for (int s = 1; s <= steps; ++s) {
    const bool last = s == steps;
    const int k = (MODE == 0) ? s : a.K - 1 - s;
    const float alpha = (MODE == 0) ? (k == 1 ? 1.f : 2.f) : (k == 0 ? 1.f : 2.f);
    const float* gk = (MODE == 1) ? in_m + (long long)k * a.plane : nullptr;
    float* tk = (MODE == 0) ? out_m + (long long)(a.with_t0 ? k : k - 1) * a.plane : out_m;
    // (the thread id and the degrees behind empty asm statements: otherwise the 20 vertices' offsets, LDS addresses,
    //  64-bit list addresses and 1/deg, deg^+-1/2 are hoisted out of the order loop as invariants -- spills)
    int tid_k = tid;
    asm volatile("" : "+v"(tid_k));
#pragma unroll
    for (int w = 0; w < (kBigVpt + 7) / 8; ++w) asm volatile("" : "+v"(degp[w]));
    // ---- phase A: P <- -(alpha / deg) sum_nbrs u_{k-1} - P     (MODE 0: = u_k, stored as T_k; MODE 1: = w_k - s G_k)
    // neighbour ids: a ring of kBigRing vertices' 16-byte words stays in flight (an order needs 320 KB of them per
    // workgroup from L2; with two vertices ahead the CU had ~32 KB outstanding and an order took 14 us, all latency)
    uint4 ring[kBigRing];
    auto ids_of = [&](int v) -> uint4 {
      if (a.half_ids) {   // (timing experiment: the first four ids twice)
        const uint2 h = reinterpret_cast<const uint2*>(a.ell)[2 * (v < N ? v : N - 1)];
        return make_uint4(h.x, h.y, h.x, h.y);
      }
      return ellv[v < N ? v : N - 1];
    };
#pragma unroll
    for (int j = 0; j < kBigRing; ++j) ring[j] = ids_of(tid_k + kBigThreads * j);
#pragma unroll
    for (int j = 0; j < kBigVpt; ++j) {
      const int v = tid_k + kBigThreads * j;
      const uint4 id = ring[j % kBigRing];
      if (j + kBigRing < kBigVpt) {
        ring[j % kBigRing] = ids_of(tid_k + kBigThreads * (j + kBigRing));
      }
      if (v < N) {
        const float2 n0 = pl[id.x & 0xFFFFu], n1 = pl[id.x >> 16], n2 = pl[id.y & 0xFFFFu], n3 = pl[id.y >> 16];
        const float2 n4 = pl[id.z & 0xFFFFu], n5 = pl[id.z >> 16], n6 = pl[id.w & 0xFFFFu], n7 = pl[id.w >> 16];
        const float sx = ((n0.x + n1.x) + (n2.x + n3.x)) + ((n4.x + n5.x) + (n6.x + n7.x));
        const float sy = ((n0.y + n1.y) + (n2.y + n3.y)) + ((n4.y + n5.y) + (n6.y + n7.y));
        const float dg = deg_of(j);
        const float coef = dg > 0.f ? -alpha * __builtin_amdgcn_rcpf(dg) : 0.f;
        const float2 nu = make_float2(fmaf(coef, sx, -P[j].x), fmaf(coef, sy, -P[j].y));
        P[j] = nu;
        if (MODE == 0) {   // T_k = u_k / s
          const float is = dg > 0.f ? __builtin_amdgcn_sqrtf(dg) : 1.f;
          if (a.sel_inv) {
            const int pr = a.sel_inv[v];
            if (pr >= 0) {
              float* dst = a.out + (((long long)mesh * a.K + k) * (a.n_sel + 1) + pr) * 4 + c0;
              dst[0] = nu.x * is;
              if (has1) dst[1] = nu.y * is;
            }
          } else {
            big_store2<VEC>(tk, (unsigned)(v * ostr), has1, make_float2(nu.x * is, nu.y * is));
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);   // (keeps the unrolled loop from hoisting every vertex's loads: 80 registers)
    }
    if (MODE == 1 && last) {  // out = G_0 + (w_0 - s G_0) / s
#pragma unroll
      for (int h = 0; h < 4; ++h) {   // (five vertices' pieces in flight at a time)
        constexpr int HV = kBigVpt / 4;
        float2 g0[HV];
#pragma unroll
        for (int jj2 = 0; jj2 < HV; ++jj2) {
          const int v = tid_k + kBigThreads * (h * HV + jj2);
          g0[jj2] = big_load2<VEC>(gk, (unsigned)((v < N ? v : N - 1) * istr), has1);
        }
#pragma unroll
        for (int jj2 = 0; jj2 < HV; ++jj2) {
          const int j = h * HV + jj2;
          const int v = tid_k + kBigThreads * j;
          if (v < N) {
            const float dg = deg_of(j), is = dg > 0.f ? __builtin_amdgcn_sqrtf(dg) : 1.f;
            big_store2<VEC>(tk, (unsigned)(v * ostr), has1,
                            make_float2(fmaf(P[j].x, is, g0[jj2].x), fmaf(P[j].y, is, g0[jj2].y)));
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (last) break;         // nothing gathers from the last plane
    __syncthreads();         // every gather of this order is done
    // ---- phase B: own rows, registers <-> LDS (MODE 1: + s G_k, whose pieces are fetched here, away from the gathers'
    //      register pressure: one global round trip per order)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int h = 0; h < 4; ++h) {   // five vertices at a time: their input pieces in flight together (clamped rows)
      constexpr int HV = kBigVpt / 4;
      float2 gpc[MODE == 1 ? HV : 1];
      if (MODE == 1) {
#pragma unroll
        for (int jj2 = 0; jj2 < HV; ++jj2) {
          const int v = tid_k + kBigThreads * (h * HV + jj2);
          gpc[jj2] = big_load2<VEC>(gk, (unsigned)((v < N ? v : N - 1) * istr), has1);
        }
      }
#pragma unroll
      for (int jj2 = 0; jj2 < HV; ++jj2) {
        const int j = h * HV + jj2;
        const int v = tid_k + kBigThreads * j;
        if (v < N) {
          float2 nw = P[j];
          if (MODE == 1) {
            const float dg = deg_of(j), sc = dg > 0.f ? __builtin_amdgcn_rsqf(dg) : 1.f;
            nw = make_float2(fmaf(sc, gpc[jj2].x, nw.x), fmaf(sc, gpc[jj2].y, nw.y));
          }
          const float2 t = pl[v];
          pl[v] = nw;
          P[j] = t;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }
}

bool cheb_big_eligible(const mvh_csr_t* lap, int B, int N, int C, int K) {
  const int need = MVH_CSR_NORMALIZED_LAPLACIAN | MVH_CSR_SYMMETRIC;
  if (dbg().force_generic || dbg().no_big) return false;
  if (!lap->rowinfo || !lap->ell || (lap->flags & need) != need || (lap->flags & MVH_CSR_ELL_OVERFLOW)) return false;
  if (lap->ell_pairs < 1 || lap->ell_pairs > 4 || lap->max_row_nnz > 8) return false;   // one 16-byte id word per vertex
  if (N + 1 <= 5120 || N > kBigThreads * kBigVpt || (size_t)(N + 1) * 8 > 160 * 1024 || N + 1 >= 65535) return false;
  return B >= 1 && C >= 1 && K >= 2;
}

template <int MODE>
static int big_launch(hipStream_t st, const mvh_csr_t* lap, const float* in, float* out, int B, int N, int C, int K,
                      bool pm, const float* mask = nullptr, bool with_t0 = false, const int32_t* sel_inv = nullptr,
                      int n_sel = 0) {
  if (pm && (C & 1)) return fail(MVH_ERR_INVALID, "cheb_big: the pair-major stack needs an even channel count");
  if (sel_inv && (MODE != 0 || pm || C > 4 || !with_t0 || n_sel <= 0))
    return fail(MVH_ERR_INVALID, "cheb_big: the selected-rows stack is a MODE 0, row-layout, <= 4-channel form with T_0");
  BigArgs a{in, out, lap->ell, lap->rowinfo, B, N, C, K, (long long)B * N * C, mask, with_t0 ? 1 : 0, dbg().big_half_ids,
            sel_inv, n_sel, pm ? 1 : 0};
  const size_t lds = (size_t)(N + 1) * 8;
  const int grid = ((B + 7) / 8) * 8 * ((C + 1) / 2);
  auto go = [&](auto kern) -> int {
    static LdsAttr attr;   // (one per instantiation of this lambda, i.e. per kernel; per device, thread-safe)
    if (int rc = attr.ensure(reinterpret_cast<const void*>(kern), lds)) return rc;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kBigThreads), lds, st, a);
    return MVH_OK;
  };
  if (int rc = (C & 1) ? go(k_cheb_big<MODE, false>) : go(k_cheb_big<MODE, true>)) return rc;
  MVH_LAUNCH_CHECK();
  return MVH_OK;
}

// tx[k-1] = T_k(L) x, k = 1 .. K-1 (what tx_forward's K - 1 SpMM launches produce); *handled == false: not eligible.
// mask: x counts only where mask > 0; with_t0: K planes, plane 0 = the (masked) x itself, plane k = T_k
int try_cheb_big_tx(hipStream_t st, const mvh_csr_t* lap, const float* x, float* tx, int B, int N, int C, int K,
                    bool pm, bool* handled, const float* mask, bool with_t0) {
  *handled = false;
  if (!cheb_big_eligible(lap, B, N, C, K)) return MVH_OK;
  if (int rc = big_launch<0>(st, lap, x, tx, B, N, C, K, pm, mask, with_t0)) return rc;
  *handled = true;
  return MVH_OK;
}

// stack [B][K][n_sel + 1][4] <- T_k(L) x, k = 0 .. K-1, at the rows the one-hot pooling `pool` selects (cheb_tstack.hip's
// layout, for a level too big for its kernel): what k_stack_dw and k_stack_contract read
int launch_big_tstack(hipStream_t st, const mvh_csr_t* lap, const mvh_csr_t* pool, const float* x, float* stack, int B,
                      int N, int Cin, int K) {
  MVH_REQUIRE(cheb_big_eligible(lap, B, N, Cin, K) && pool && pool->sel_inv && pool->n_cols == N && Cin <= 4,
              "cheb_big: selected-rows stack on a level it does not take");
  if (int rc = big_launch<0>(st, lap, x, stack, B, N, Cin, K, false, nullptr, true, pool->sel_inv, pool->n_rows)) return rc;
  return MVH_OK;
}

// out = sum_k T_k(L) G_k from the stack G [K][B][N][C] (Clenshaw; L symmetric, so also the adjoint chain of the backward)
int try_cheb_big_clenshaw(hipStream_t st, const mvh_csr_t* lap, const float* G, float* out, int B, int N, int C, int K,
                          bool pm, bool* handled) {
  *handled = false;
  if (!cheb_big_eligible(lap, B, N, C, K)) return MVH_OK;
  if (int rc = big_launch<1>(st, lap, G, out, B, N, C, K, pm)) return rc;
  *handled = true;
  return MVH_OK;
}

}  // namespace mvh
