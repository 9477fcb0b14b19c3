// Rows C + Q: ChebConv_batch.forward (nn/conv.py:557-577) and its analytic backward.
//
// Baseline ("stack") pipeline -- every stage is a separate launch over [B*N] rows:
//   forward : T_1..T_{K-1} by K-1 sparse propagates (spmm.hip), then ONE contraction
//             out = act([T_0|..|T_{K-1}] W + bias)            (k_cheb_contract)
//   backward: G_k = dpre W_k^T for all k in one launch        (k_cheb_gstack)
//             dx  = sum_k T_k(L^T) G_k by Clenshaw, in place   (spmm.hip, K-1 launches)
//             dW_k = T_k^T dpre, db = sum dpre: register-tiled split over row chunks with
//             per-block partials and a fixed-order final reduce (k_cheb_dw, k_reduce_partials)
//             -- no atomics, so gradients are bitwise reproducible run to run.
// Levels of 5120 .. 20480 vertices (BASELINE configs[3]) replace the K-1 propagates by cheb_big.hip's one-launch
// recurrence over pair-major stack planes, and the 16 -> 16 layer's backward by ONE pass over the T_k(dpre) stack
// (k_big_bwd16 below: dx and dW / db together).
// Weights are wave-uniform and read through the scalar cache (s_load) so the inner loops are
// v_fma with an SGPR operand; activations move as 16-byte vectors.
#include "common.hpp"
#include "bf16.hpp"

namespace mvh {

typedef float f2nt_c __attribute__((ext_vector_type(2)));
typedef float f4nt_c __attribute__((ext_vector_type(4)));
static bool dw_is_mfma(long long rows, int Cout);

// ------------------------------------------------------------------ forward contraction
// optional extras of the contraction's epilogue (Cout % 4 == 0 forms): sign bytes, and for 16 output channels a per-vertex map
struct ContractExtra {
  uint8_t* bits = nullptr;
  const float* map_w = nullptr;
  float* map_out = nullptr;
  int map_c = 0, map_n0 = 0, N = 1;
};

template <int COUT_T, bool FULL, bool VIN>
__global__ void __launch_bounds__(256)
k_cheb_contract(const float* __restrict__ x, const float* __restrict__ tx, const float* __restrict__ W,
                const float* __restrict__ bias, float* __restrict__ out, long long rows, int Cin,
                int Cout, int K, int act, int x_bf16, int pmN, ContractExtra ex) {
  // x_bf16: x is stored as bf16 (K == 1 only: the W_eff pass of the split path; the launcher checks)
  // pmN != 0: the stack planes tx are pair-major [B][8][|pmN|][2] (cheb_big.hip; 16 -> 16 only, the launcher checks);
  // pmN < 0: T_0 is plane 0 of that stack as well (x is not read) and T_k plane k -- the backward's T_k(dpre) stack
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  float acc[COUT_T];
#pragma unroll
  for (int co = 0; co < COUT_T; ++co) acc[co] = (bias && (FULL || co < Cout)) ? bias[co] : 0.f;
  if constexpr (VIN && FULL && COUT_T == 16) {
    if (Cin == 16) {  // the stack planes are the kernel's HBM stream: plane k+1 is in flight while plane k is used
      float4 cur[4], nxt[4];
      const int pN = pmN < 0 ? -pmN : pmN, koff = pmN < 0 ? 0 : 1;   // stack plane of T_k = k - koff
      long long pm_off = 0;   // (float2 units) of (mesh, pair 0, vertex) inside a pair-major plane
      if (pmN) {
        const int rr = (int)r, b = rr / pN;
        pm_off = (long long)b * 8 * pN + (rr - b * pN);
      }
      if (pmN < 0) {
        const float2* sp = reinterpret_cast<const float2*>(tx) + pm_off;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float2 lo = sp[(long long)(2 * j) * pN], hi = sp[(long long)(2 * j + 1) * pN];
          cur[j] = make_float4(lo.x, lo.y, hi.x, hi.y);
        }
      } else {
        const float4* s0 = reinterpret_cast<const float4*>(x + r * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) cur[j] = s0[j];
      }
      for (int k = 0; k < K; ++k) {
        const int kn = min(k + 1, K - 1);  // (the last trip re-reads its own plane: no branch around the loads)
        if (pmN && (kn > 0 || pmN < 0)) {
          // (streamed once: nontemporal, so that 0.8 GB of stack do not sweep L2)
          const f2nt_c* sp = reinterpret_cast<const f2nt_c*>(tx + (long long)(kn - koff) * rows * 16) + pm_off;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f2nt_c lo = __builtin_nontemporal_load(sp + (long long)(2 * j) * pN);
            const f2nt_c hi = __builtin_nontemporal_load(sp + (long long)(2 * j + 1) * pN);
            nxt[j] = make_float4(lo.x, lo.y, hi.x, hi.y);
          }
        } else {
          const float4* sn = reinterpret_cast<const float4*>((kn == 0 ? x : tx + (long long)(kn - 1) * rows * 16) + r * 16);
#pragma unroll
          for (int j = 0; j < 4; ++j) nxt[j] = sn[j];
        }
        const float* Wk = W + (long long)k * 16 * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float tv[4] = {cur[j].x, cur[j].y, cur[j].z, cur[j].w};
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const float* w = Wk + (j * 4 + t) * 16;
#pragma unroll
            for (int co = 0; co < 16; ++co) acc[co] = fmaf(tv[t], w[co], acc[co]);
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) cur[j] = nxt[j];
      }
      K = 0;  // done: skip the generic loop below
    }
  }
  for (int k = 0; k < K; ++k) {
    const float* src = (k == 0 ? x : tx + (long long)(k - 1) * rows * Cin) + r * Cin;
    const float* Wk = W + (long long)k * Cin * Cout;
    if constexpr (VIN) {
      for (int c4 = 0; c4 < Cin; c4 += 4) {
        const float4 t = x_bf16 ? load4_any(x, r * Cin + c4, true) : *reinterpret_cast<const float4*>(src + c4);
        const float tv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float* w = Wk + (long long)(c4 + j) * Cout;
#pragma unroll
          for (int co = 0; co < COUT_T; ++co)
            if (FULL || co < Cout) acc[co] = fmaf(tv[j], w[co], acc[co]);
        }
      }
    } else {
      for (int ci = 0; ci < Cin; ++ci) {
        const float t = src[ci];
        const float* w = Wk + (long long)ci * Cout;
#pragma unroll
        for (int co = 0; co < COUT_T; ++co)
          if (FULL || co < Cout) acc[co] = fmaf(t, w[co], acc[co]);
      }
    }
  }
  float* o = out + r * Cout;
  if (act == MVH_ACT_RELU) {
#pragma unroll
    for (int co = 0; co < COUT_T; ++co) acc[co] = fmaxf(acc[co], 0.f);
  }
  if constexpr (FULL && (COUT_T % 4 == 0)) {
#pragma unroll
    for (int co = 0; co < COUT_T; co += 4)
      *reinterpret_cast<float4*>(o + co) = make_float4(acc[co], acc[co + 1], acc[co + 2], acc[co + 3]);
    if (ex.bits) {   // the ReLU sign bytes the backward masks with (one per 4 channels), instead of a k_relu_bits pass over `out`
#pragma unroll
      for (int co = 0; co < COUT_T; co += 4)
        ex.bits[r * (COUT_T / 4) + co / 4] = (uint8_t)((acc[co] > 0.f ? 1 : 0) | (acc[co + 1] > 0.f ? 2 : 0) |
                                                         (acc[co + 2] > 0.f ? 4 : 0) | (acc[co + 3] > 0.f ? 8 : 0));
    }
    if constexpr (COUT_T == 16) {
      if (ex.map_out) {   // per-vertex map behind the layer (ConvIO::map_*): rows >= map_n0 of a mesh, the fma chain of this kernel's K = 1 form
        const int v = (int)(r % ex.N);
        if (v >= ex.map_n0) {
          for (int oo = 0; oo < ex.map_c; ++oo) {
            float m = 0.f;
#pragma unroll
            for (int cc = 0; cc < 16; ++cc) m = fmaf(acc[cc], ex.map_w[cc * ex.map_c + oo], m);
            ex.map_out[r * ex.map_c + oo] = m;
          }
        }
      }
    }
  } else {
#pragma unroll
    for (int co = 0; co < COUT_T; ++co)
      if (FULL || co < Cout) o[co] = acc[co];
  }
}

static int launch_contract(hipStream_t st, const float* x, const float* tx, const float* W,
                           const float* bias, float* out, long long rows, int Cin, int Cout, int K,
                           int act, bool x_bf16 = false, int pmN = 0, ContractExtra ex = ContractExtra()) {
  if (ex.bits && (Cout % 4 != 0 || !(Cout == 8 || Cout == 16 || Cout == 32) || act != MVH_ACT_RELU))
    return fail(MVH_ERR_INVALID, "cheb_conv: sign bytes out of the contraction need ReLU and 8 / 16 / 32 output channels");
  if (ex.map_out && (Cout != 16 || !ex.map_w || ex.map_c < 1 || ex.map_c > 4 || ex.N < 1))
    return fail(MVH_ERR_INVALID, "cheb_conv: the per-vertex map rides on the 16-channel contraction only");
  const bool vin = (Cin % 4 == 0) && (((uintptr_t)x | (uintptr_t)tx) % 16 == 0);
  // pmN != 0: pair-major stack planes -- only the 16 -> 16 fast path of the kernel reads them (< 0: T_0 in the stack)
  if (pmN && !(vin && Cin == 16 && Cout == 16)) return fail(MVH_ERR_INVALID, "cheb_conv: pair-major stack outside the 16 -> 16 contraction");
  if (x_bf16 && (!vin || K != 1 || (Cin == 16 && Cout == 16)))
    return fail(MVH_ERR_UNSUPPORTED, "cheb_conv: bf16 rows reach the contraction only through the K = 1 split pass");
  const int xb = x_bf16 ? 1 : 0;
  const int grid = cdiv(rows, 256);
#define MVH_C(CT, FULL, VIN)                                                                    \
  hipLaunchKernelGGL((k_cheb_contract<CT, FULL, VIN>), dim3(grid), dim3(256), 0, st, x, tx, W,  \
                     bias, out, rows, Cin, Cout, K, act, xb, pmN, ex)
#define MVH_CV(CT, FULL) \
  do { if (vin) MVH_C(CT, FULL, true); else MVH_C(CT, FULL, false); } while (0)
  if (Cout == 3) MVH_CV(3, true);
  else if (Cout == 8) MVH_CV(8, true);
  else if (Cout == 16) MVH_CV(16, true);
  else if (Cout == 32) MVH_CV(32, true);
  else if (Cout <= 8) MVH_CV(8, false);
  else if (Cout <= 32) MVH_CV(32, false);
  else return fail(MVH_ERR_UNSUPPORTED, "cheb_conv: out_channels %d > 32 not supported", Cout);
#undef MVH_CV
#undef MVH_C
  MVH_LAUNCH_CHECK();
  return MVH_OK;
}

// ------------------------------------------------------------------ backward: G_k = dpre W_k^T
template <int COUT_T, bool FULL>
__global__ void __launch_bounds__(256)
k_cheb_gstack(const float* __restrict__ dout, const float* __restrict__ out, const float* __restrict__ W,
              float* __restrict__ G, float* __restrict__ g0, long long rows, int Cin, int Cout, int K,
              int act) {
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  float dp[COUT_T];
#pragma unroll
  for (int co = 0; co < COUT_T; ++co) {
    float d = 0.f;
    if (FULL || co < Cout) {
      d = dout[r * Cout + co];
      if (act == MVH_ACT_RELU && !(out[r * Cout + co] > 0.f)) d = 0.f;
    }
    dp[co] = d;
  }
  for (int k = 0; k < K; ++k) {
    // plane 0 may live in a separate buffer (dx itself when K == 1)
    float* dst = (k == 0 ? g0 : G + (long long)k * rows * Cin) + r * Cin;
    const float* Wk = W + (long long)k * Cin * Cout;
    int ci = 0;
    if ((Cin & 3) == 0 && ((uintptr_t)dst & 15) == 0) {  // four input channels per 16-byte store
      for (; ci < Cin; ci += 4) {
        float g4[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float* w = Wk + (long long)(ci + t) * Cout;
          float g = 0.f;
#pragma unroll
          for (int co = 0; co < COUT_T; ++co)
            if (FULL || co < Cout) g = fmaf(dp[co], w[co], g);
          g4[t] = g;
        }
        *reinterpret_cast<float4*>(dst + ci) = make_float4(g4[0], g4[1], g4[2], g4[3]);
      }
    }
    for (; ci < Cin; ++ci) {
      const float* w = Wk + (long long)ci * Cout;
      float g = 0.f;
#pragma unroll
      for (int co = 0; co < COUT_T; ++co)
        if (FULL || co < Cout) g = fmaf(dp[co], w[co], g);
      dst[ci] = g;
    }
  }
}

// K == 1 with a narrow dout (the split path of the final 16 -> 3 layer): g0[r][ci] = sum_co dpre[r][co] W[ci][co].
// One lane per (row, 4 input channels): the lanes of a row write 16 consecutive bytes each, so a wave
// stores whole rows back to back (the row-per-lane kernel above scatters 16-byte pieces 64 B apart).
template <int COUT_T>
__global__ void __launch_bounds__(256)
k_gstack_rows(const float* __restrict__ dout, const float* __restrict__ out, const float* __restrict__ W,
              float* __restrict__ g0, long long total, int Cin, int act, int g_bf16) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int QV = Cin >> 2, q = (int)(idx % QV);
  const long long r = idx / QV;
  float w[4][COUT_T];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int co = 0; co < COUT_T; ++co) w[t][co] = W[(long long)(4 * q + t) * COUT_T + co];
  float g[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int co = 0; co < COUT_T; ++co) {
    float d = dout[r * COUT_T + co];
    if (act == MVH_ACT_RELU && !(out[r * COUT_T + co] > 0.f)) d = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) g[t] = fmaf(d, w[t][co], g[t]);
  }
  store4_any(g0, r * Cin + 4 * q, g_bf16 != 0, g[0], g[1], g[2], g[3]);  // (bf16 storage of the gradient rows)
}

// 16 -> 16 channels on a big level (the 20k-vertex level of BASELINE configs[3]: 82 MB of dout in, 819 MB of
// G stack out): G_k^T = W_k dpre^T on the matrix pipe.  A = W_k [ci x co], B = dpre^T [co x 16 rows]; the
// reduction slot (s, q) stands for co = 4 q + s so that both operands are 16-byte loads, and the result
// C[ci = 4 q + j][row = lane % 16] leaves as one float4 per lane: every wave load and store is 1 KB
// contiguous.  (The one-thread-per-row kernel reads dout with a 64-byte lane stride: 985 us here.)
typedef float f32x4_g __attribute__((ext_vector_type(4)));

template <int KMAX, bool RELU>
__global__ void __launch_bounds__(256)
k_gstack_mfma16(const float* __restrict__ dout, const float* __restrict__ out, const float* __restrict__ W,
                float* __restrict__ G, float* __restrict__ g0, long long rows, int K, long long blocks_per_wave,
                int pmN) {
  // pmN > 0: the planes leave pair-major, [B][8][pmN][2] (what cheb_big.hip's Clenshaw kernel streams): lane (m, q)
  // holds the channels 4 q .. 4 q + 3 of its row = the pairs 2 q and 2 q + 1, 16 lanes write 128 contiguous bytes
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = lane & 15, q = lane >> 4;
  float4 wa[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k)
    wa[k] = k < K ? *reinterpret_cast<const float4*>(W + ((long long)k * 16 + m) * 16 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
  const long long nblk = (rows + 15) / 16;
  const long long wid = (long long)blockIdx.x * 4 + wave;
  const long long b_end = min(nblk, (wid + 1) * blocks_per_wave);
  for (long long blk = wid * blocks_per_wave; blk < b_end; ++blk) {
    const long long row_raw = blk * 16 + m;
    const long long row = min(row_raw, rows - 1);
    float4 d = *reinterpret_cast<const float4*>(dout + row * 16 + 4 * q);
    if constexpr (RELU) {
      const float4 o = *reinterpret_cast<const float4*>(out + row * 16 + 4 * q);
      d.x = o.x > 0.f ? d.x : 0.f; d.y = o.y > 0.f ? d.y : 0.f; d.z = o.z > 0.f ? d.z : 0.f; d.w = o.w > 0.f ? d.w : 0.f;
    }
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      if (k < K) {
        f32x4_g acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[k].x, d.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[k].y, d.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[k].z, d.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[k].w, d.w, acc, 0, 0, 0);
        float* dst = (k == 0 ? g0 : G + (long long)k * rows * 16);
        if (row_raw < rows) {
          if (pmN && (pmN & 1) == 0) {
            // rows m and m ^ 1 swap halves (pmN even: both in one mesh): the even row's lane then holds pair 2 q of
            // both rows, the odd one pair 2 q + 1 -- one 16-byte store per lane, 128 contiguous bytes per 8 lanes
            const bool odd = m & 1;
            const float rx = __shfl_xor(odd ? acc[0] : acc[2], 1, 64), ry = __shfl_xor(odd ? acc[1] : acc[3], 1, 64);
            const int rr = (int)row_raw & ~1, b = rr / pmN, v = rr - b * pmN;
            float2* pp = reinterpret_cast<float2*>(dst) + ((long long)(b * 8 + 2 * q + (odd ? 1 : 0)) * pmN + v);
            *reinterpret_cast<float4*>(pp) = odd ? make_float4(rx, ry, acc[2], acc[3]) : make_float4(acc[0], acc[1], rx, ry);
          } else if (pmN) {
            const int rr = (int)row_raw, b = rr / pmN, v = rr - b * pmN;
            float2* pp = reinterpret_cast<float2*>(dst) + ((long long)(b * 8 + 2 * q) * pmN + v);
            pp[0] = make_float2(acc[0], acc[1]);
            pp[pmN] = make_float2(acc[2], acc[3]);
          } else {
            *reinterpret_cast<float4*>(dst + row_raw * 16 + 4 * q) = make_float4(acc[0], acc[1], acc[2], acc[3]);
          }
        }
      }
    }
  }
}

static bool gstack_is_mfma(const float* dout, const float* out, const float* W, const float* G, const float* g0,
                           long long rows, int Cin, int Cout, int K) {
  return Cin == 16 && Cout == 16 && K <= 12 && rows >= 4096 && !dbg().no_gstack_mfma &&
         (((uintptr_t)dout | (uintptr_t)out | (uintptr_t)W | (uintptr_t)G | (uintptr_t)g0) & 15) == 0;
}

// pmN > 0 (only where gstack_is_mfma holds): the planes are written pair-major for cheb_big.hip
static int launch_gstack(hipStream_t st, const float* dout, const float* out, const float* W, float* G,
                         float* g0, long long rows, int Cin, int Cout, int K, int act, bool g_bf16 = false, int pmN = 0) {
  if (K == 1 && (Cin & 3) == 0 && ((uintptr_t)g0 & 15) == 0 && (Cout == 3 || Cout == 4)) {
    const long long total = rows * (Cin >> 2);
    const int gb = g_bf16 ? 1 : 0;
    if (Cout == 3)
      hipLaunchKernelGGL((k_gstack_rows<3>), dim3(cdiv(total, 256)), dim3(256), 0, st, dout, out, W, g0, total, Cin, act, gb);
    else
      hipLaunchKernelGGL((k_gstack_rows<4>), dim3(cdiv(total, 256)), dim3(256), 0, st, dout, out, W, g0, total, Cin, act, gb);
    MVH_LAUNCH_CHECK();
    return MVH_OK;
  }
  if (g_bf16) return fail(MVH_ERR_UNSUPPORTED, "cheb_conv: bf16 gradient rows leave the G-stack only through the K = 1 split pass");
  if (gstack_is_mfma(dout, out, W, G, g0, rows, Cin, Cout, K)) {
    const long long nblk = (rows + 15) / 16;
    const long long waves = min(nblk, 8192ll);
    const long long bpw = (nblk + waves - 1) / waves;
    const int g = (int)((nblk + bpw * 4 - 1) / (bpw * 4));
    const bool relu = act == MVH_ACT_RELU;
#define MVH_GM(KM, R) \
  hipLaunchKernelGGL((k_gstack_mfma16<KM, R>), dim3(g), dim3(256), 0, st, dout, out, W, G, g0, rows, K, bpw, pmN)
    if (K <= 6) { if (relu) MVH_GM(6, true); else MVH_GM(6, false); }
    else { if (relu) MVH_GM(12, true); else MVH_GM(12, false); }
#undef MVH_GM
    MVH_LAUNCH_CHECK();
    return MVH_OK;
  }
  if (pmN) return fail(MVH_ERR_INVALID, "cheb_conv: pair-major G stack outside the matrix-pipe producer");
  const int grid = cdiv(rows, 256);
#define MVH_G(CT, FULL)                                                                          \
  hipLaunchKernelGGL((k_cheb_gstack<CT, FULL>), dim3(grid), dim3(256), 0, st, dout, out, W, G, g0, \
                     rows, Cin, Cout, K, act)
  if (Cout == 3) MVH_G(3, true);
  else if (Cout == 8) MVH_G(8, true);
  else if (Cout == 16) MVH_G(16, true);
  else if (Cout == 32) MVH_G(32, true);
  else if (Cout <= 8) MVH_G(8, false);
  else if (Cout <= 32) MVH_G(32, false);
  else return fail(MVH_ERR_UNSUPPORTED, "cheb_conv: out_channels %d > 32 not supported", Cout);
#undef MVH_G
  MVH_LAUNCH_CHECK();
  return MVH_OK;
}

// ------------------------------------------------------------------ backward: dW, db
// A^T D over the row dimension: A = [T_0 | T_1 | .. | T_{K-1} | 1] (rows x KC1), D = dpre
// (rows x Cout).  Each thread owns a 4x4 tile of the (KC1 x Cout) result; a block walks row
// chunks of R rows staged in LDS and writes its partial tile set; blockIdx.y selects the group
// of 256 tiles when KC1/4 * Cout/4 > 256.
constexpr int kDwRows = 32;

__global__ void __launch_bounds__(256)
k_cheb_dw(const float* __restrict__ x, const float* __restrict__ tx, const float* __restrict__ dout,
          const float* __restrict__ out, float* __restrict__ partial, long long rows, int Cin, int Cout,
          int K, int act, int TI, int TJ, int nchunks) {
  extern __shared__ float lds[];
  const int KC = K * Cin;
  const int lda = TI * 4;  // padded KC+1
  const int ldd = TJ * 4;  // padded Cout
  float* Tl = lds;                   // [R][lda]
  float* Dl = lds + kDwRows * lda;   // [R][ldd]
  const int tile = blockIdx.y * blockDim.x + threadIdx.x;
  const bool active = tile < TI * TJ;
  const int ti = active ? tile / TJ : 0, tj = active ? tile % TJ : 0;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

  for (int chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    const long long r0 = (long long)chunk * kDwRows;
    const int nr = (int)min((long long)kDwRows, rows - r0);
    __syncthreads();
    // stage T planes: element (r, k, ci) -> Tl[r][k*Cin+ci]
    for (int i = threadIdx.x; i < kDwRows * lda; i += blockDim.x) {
      const int rr = i / lda, c = i % lda;
      float v = 0.f;
      if (rr < nr) {
        if (c < KC) {
          const int k = c / Cin, ci = c % Cin;
          const float* src = (k == 0 ? x : tx + (long long)(k - 1) * rows * Cin);
          v = src[(r0 + rr) * Cin + ci];
        } else if (c == KC) {
          v = 1.f;  // ones column -> db
        }
      }
      Tl[i] = v;
    }
    for (int i = threadIdx.x; i < kDwRows * ldd; i += blockDim.x) {
      const int rr = i / ldd, c = i % ldd;
      float v = 0.f;
      if (rr < nr && c < Cout) {
        v = dout[(r0 + rr) * Cout + c];
        if (act == MVH_ACT_RELU && !(out[(r0 + rr) * Cout + c] > 0.f)) v = 0.f;
      }
      Dl[i] = v;
    }
    __syncthreads();
    if (active) {
#pragma unroll 4
      for (int rr = 0; rr < kDwRows; ++rr) {
        const float4 a = *reinterpret_cast<const float4*>(Tl + rr * lda + ti * 4);
        const float4 d = *reinterpret_cast<const float4*>(Dl + rr * ldd + tj * 4);
        const float av[4] = {a.x, a.y, a.z, a.w};
        const float dv[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], dv[j], acc[i][j]);
      }
    }
  }
  if (active) {
    float* p = partial + (long long)blockIdx.x * (KC + 1) * Cout;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = ti * 4 + i;
      if (row > KC) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = tj * 4 + j;
        if (c < Cout) p[(long long)row * Cout + c] = acc[i][j];
      }
    }
  }
}

// out[i] = sum_g partial[g][i] in fixed order; first n_w entries go to dW, the rest to db.
__global__ void __launch_bounds__(1024)
k_reduce_partials(const float* __restrict__ partial, int G, int n, int n_w, float* __restrict__ dW,
                  float* __restrict__ db) {
  // 64 outputs per block, NW = blockDim / 64 waves (4 or 16): wave w sums the partials g = w, w + NW, ... with four
  // independent chains, then wave 0 adds the waves' sums in wave order.  (16 waves where G is large: with 4, an
  // output was 32 dependent load rounds deep -- 126 us for the 496 outputs of the 20k first layer, at the very end
  // of the step's critical path.)
  __shared__ float red[15][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, NW = blockDim.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < n) {
    int g = w;
    for (; g + 3 * NW < G; g += 4 * NW) {
      s0 += partial[(long long)g * n + i];
      s1 += partial[(long long)(g + NW) * n + i];
      s2 += partial[(long long)(g + 2 * NW) * n + i];
      s3 += partial[(long long)(g + 3 * NW) * n + i];
    }
    for (; g < G; g += NW) s0 += partial[(long long)g * n + i];
  }
  float s = (s0 + s1) + (s2 + s3);
  if (w > 0) red[w - 1][lane] = s;
  __syncthreads();
  if (w == 0 && i < n) {
    for (int u = 1; u < NW; ++u) s += red[u - 1][lane];
    if (i < n_w) dW[i] = s;
    else if (db) db[i - n_w] = s;
  }
}

// Streaming weight gradient on the matrix pipe for big levels (the 20k-vertex level of BASELINE configs[3]:
// rows = B*N = 1.28 M, stack = 819 MB): dW[kc][c] = sum_rows T[row][kc] * dpre[row][c] is a tall-skinny
// GEMM whose reduction index is the row, so each wave walks its own contiguous row range four rows at a time
// and feeds v_mfma_f32_16x16x4_f32 (exact fp32) straight from global memory: lane (m = lane%16, q = lane/16)
// loads T[row+q][16 t + m] for its M-tiles and dpre[row+q][16 n + m] -- for Cin = 16 every wave load is 256
// contiguous bytes of one plane.  No LDS staging, no barriers in the loop; the 4 waves of a block are summed
// through LDS once at the end and the block writes one partial in the layout k_reduce_partials expects
// ([KC+1][Cout], row KC = db via a ones column).  blockIdx.y selects a group of MT M-tiles (MT*16 stack
// columns), so the accumulators stay at MT*TN*4 VGPRs; dpre is re-read per group (1/10 of the stack bytes).
typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <int TN, int MT, bool RELU>
__global__ void __launch_bounds__(256)
k_cheb_dw_mfma(const float* __restrict__ x, const float* __restrict__ tx, const float* __restrict__ dout,
               const float* __restrict__ out, float* __restrict__ partial, long long rows, int Cin, int K,
               long long rows_per_wave, int pmN, const int* __restrict__ sel, int selN, int fullN) {
  // pmN > 0: the stack planes tx are pair-major [B][Cin/2][pmN][2] (cheb_big.hip); x stays a row-layout tensor
  // sel != NULL: dout is the gradient of the POOLED output [B][selN][Cout] (one-hot downsampling, nn/pool.py D): the
  // un-pooled gradient is zero everywhere else, so the reduction runs over the B * selN pooled rows only (`rows`
  // counts those); row (b, v') reads x / T_k / the ReLU mask at the fine vertex (b, sel[v']) of a mesh of fullN rows
  // -- the scatter into zeros of the reference's autograd, its pool launch and 3/4 of the matrix work disappear
  constexpr int Cout = 16 * TN;
  __shared__ float red[3][MT * TN][64][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = lane & 15, q = lane >> 4;
  const int KC = K * Cin;
  const long long plane_rows = sel ? (rows / selN) * fullN : rows;   // rows of one stack plane
  const float* ap[MT];
  float a_load[MT], a_one[MT];  // a = v * a_load + a_one: stack column, ones column (db) or padding
  bool a_pm[MT];                // this column lives in a pair-major plane
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int kc = ((int)blockIdx.y * MT + t) * 16 + m;
    ap[t] = x;
    a_load[t] = 0.f;
    a_one[t] = 0.f;
    a_pm[t] = false;
    if (kc < KC) {
      const int k = kc / Cin, ci = kc - k * Cin;
      ap[t] = (k == 0 ? x : tx + (long long)(k - 1) * plane_rows * Cin) + ci;
      if (k > 0 && pmN) {  // element (b, pair, v, c) = b N Cin + pair N 2 + 2 v + c = row Cin - v (Cin - 2) + pair N 2 + c
        ap[t] = tx + (long long)(k - 1) * plane_rows * Cin + (long long)(ci >> 1) * pmN * 2 + (ci & 1);
        a_pm[t] = true;
      }
      a_load[t] = 1.f;
    } else if (kc == KC) {
      a_one[t] = 1.f;
    }
  }
  f32x4_t acc[MT][TN];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int n = 0; n < TN; ++n) acc[t][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const long long wid = (long long)blockIdx.x * 4 + wave;
  const long long r_begin = wid * rows_per_wave;
  const long long r_end = min(rows, r_begin + rows_per_wave);
  for (long long r = r_begin; r < r_end; r += 16) {  // 16 rows per trip: four independent load groups in flight
    float a[4][MT], d[4][TN];
    if (pmN) {
      // pair-major columns: the two lanes of a channel pair fetch DIFFERENT rows as float2 (the even channel's lane
      // the row of instruction 2 h, the odd one's the row of 2 h + 1) and swap halves -- 8 lanes x 8 bytes = one
      // 64-byte line per pair and instruction, half the loads of the scalar form below
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const long long row_mine = min(r + 4 * (2 * h + (m & 1)) + q, rows - 1);
        const long long off = row_mine * Cin - (long long)((int)row_mine % pmN) * (Cin - 2);
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          if (a_pm[t]) {
            const float2 f = *reinterpret_cast<const float2*>(ap[t] - (m & 1) + off);
            const float gx = __shfl_xor(f.x, 1, 64), gy = __shfl_xor(f.y, 1, 64);
            a[2 * h][t] = (m & 1) ? gy : f.x;
            a[2 * h + 1][t] = (m & 1) ? f.y : gx;
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long row_raw = r + 4 * u + q;
      const float live = row_raw < r_end ? 1.f : 0.f;  // (rows_per_wave is a multiple of 16: only the global tail)
      const long long row = min(row_raw, rows - 1);
      long long frow = row;   // the row of x / T_k / out
      if (sel) {
        const int rr = (int)row, b = rr / selN;
        frow = (long long)b * fullN + sel[rr - b * selN];
      }
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        if (pmN && a_pm[t]) a[u][t] *= live;
        else a[u][t] = fmaf(ap[t][frow * Cin], a_load[t], a_one[t]) * live;
      }
#pragma unroll
      for (int n = 0; n < TN; ++n) {
        float v = dout[row * Cout + n * 16 + m] * live;
        if constexpr (RELU) v = out[frow * Cout + n * 16 + m] > 0.f ? v : 0.f;
        d[u][n] = v;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int n = 0; n < TN; ++n)
          acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][t], d[u][n], acc[t][n], 0, 0, 0);
  }
  if (wave > 0) {
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int n = 0; n < TN; ++n)
#pragma unroll
        for (int j = 0; j < 4; ++j) red[wave - 1][t * TN + n][lane][j] = acc[t][n][j];
  }
  __syncthreads();
  if (wave == 0) {
    float* p = partial + (long long)blockIdx.x * (KC + 1) * Cout;
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int n = 0; n < TN; ++n)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int kc = ((int)blockIdx.y * MT + t) * 16 + 4 * q + j;  // C[m' = 4 q + j][n' = lane % 16]
          const float v = ((acc[t][n][j] + red[0][t * TN + n][lane][j]) + red[1][t * TN + n][lane][j]) +
                          red[2][t * TN + n][lane][j];
          if (kc <= KC) p[(long long)kc * Cout + n * 16 + m] = v;
        }
  }
}

static int dw_grid(long long rows) { return (int)min((long long)512, (rows + kDwRows - 1) / kDwRows); }

static bool dw_is_mfma(long long rows, int Cout) { return (Cout == 16 || Cout == 32) && rows >= 4096 && !dbg().no_dw_mfma; }

// pmN > 0 (only where dw_is_mfma holds): the stack planes tx are pair-major (cheb_big.hip)
static int launch_dw(hipStream_t st, const float* x, const float* tx, const float* dout, const float* out,
                     float* partial, float* dW, float* db, long long rows, int Cin, int Cout, int K,
                     int act, int pmN = 0, const int* sel = nullptr, int selN = 0, int fullN = 0) {
  const int KC = K * Cin;
  const int n = (KC + 1) * Cout;
  int G = dw_grid(rows);
  if ((pmN || sel) && !dw_is_mfma(rows, Cout))
    return fail(MVH_ERR_INVALID, "cheb_conv dW: pair-major stack / pooled-row list outside the matrix-pipe kernel");
  if (pmN && sel) return fail(MVH_ERR_INVALID, "cheb_conv dW: pooled-row list over a pair-major stack");
  if (dw_is_mfma(rows, Cout)) {
    // big levels: streaming MFMA reduction (k_cheb_dw_mfma); G blocks x 4 waves, contiguous row ranges
    const long long waves = max(4ll, min(2048ll, rows / 256));
    G = min(G, (int)((waves + 3) / 4));
    long long rpw = (rows + (long long)G * 4 - 1) / ((long long)G * 4);
    rpw = (rpw + 15) / 16 * 16;
    const int tiles_m = cdiv(KC + 1, 16);
    const bool relu = act == MVH_ACT_RELU;
#define MVH_DWM(TN, MT, R)                                                                                          \
  hipLaunchKernelGGL((k_cheb_dw_mfma<TN, MT, R>), dim3(G, cdiv(tiles_m, MT)), dim3(256), 0, st, x, tx, dout, out, \
                     partial, rows, Cin, K, rpw, pmN, sel, selN, fullN)
    if (Cout == 16) {  // (MT = 6: two passes over dpre instead of three for 11 M-tiles; all 11 at once was slower)
      if (tiles_m <= 2) { if (relu) MVH_DWM(1, 2, true); else MVH_DWM(1, 2, false); }
      else if (tiles_m <= 4 || tiles_m > 12) { if (relu) MVH_DWM(1, 4, true); else MVH_DWM(1, 4, false); }
      else { if (relu) MVH_DWM(1, 6, true); else MVH_DWM(1, 6, false); }
    } else {
      if (tiles_m <= 2) { if (relu) MVH_DWM(2, 2, true); else MVH_DWM(2, 2, false); }
      else { if (relu) MVH_DWM(2, 4, true); else MVH_DWM(2, 4, false); }
    }
#undef MVH_DWM
    MVH_LAUNCH_CHECK();
  } else {
    const int TI = cdiv(KC + 1, 4), TJ = cdiv(Cout, 4);
    const int tiles = TI * TJ;
    const int threads = min(256, cdiv(tiles, 64) * 64);
    const int gy = cdiv(tiles, threads);
    const int nchunks = cdiv(rows, kDwRows);
    const size_t lds = (size_t)kDwRows * (TI * 4 + TJ * 4) * sizeof(float);
    if (lds > 64 * 1024) return fail(MVH_ERR_UNSUPPORTED, "cheb_conv dW: K*Cin=%d too large", KC);
    hipLaunchKernelGGL(k_cheb_dw, dim3(G, gy), dim3(threads), lds, st, x, tx, dout, out, partial, rows, Cin,
                       Cout, K, act, TI, TJ, nchunks);
    MVH_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(k_reduce_partials, dim3(cdiv(n, 64)), dim3(G >= 128 ? 1024 : 256), 0, st, partial, G, n, KC * Cout,
                     dW, db);
  MVH_LAUNCH_CHECK();
  return MVH_OK;
}

// ------------------------------------------------------------------ 16 -> 16 backward on a streaming level, one pass
// BASELINE configs[3]'s level-0 layer: both gradients that need the Chebyshev stack from ONE pass over
//   S_k = T_k(L)(dpre)   (K pair-major planes from cheb_big.hip; plane 0 = dpre = dout masked by the forward output):
//   dx    = sum_k S_k W_k^T                       (the input-side form of the dX)
//   dW_k  = x^T S_k,  db = column sums of S_0     (L is symmetric: <T_k x, dpre> = <x, T_k dpre>)
// so the 0.8 GB stack is read once instead of once by the contraction and once by the weight-gradient kernel (which
// also shut each other out of the CUs when they ran side by side), and the forward's own T_k(x) stack is not needed
// by the backward at all.  A wave walks a contiguous row range 16 rows per trip, everything on v_mfma_f32_16x16x4_f32:
//   dW_k tile [ci x co] += x^T[ci x 4 rows] S_k[4 rows x co]     A = x[r + 4 u + q][m], B = S_k[r + 4 u + q][m]
//   dx tile [16 rows x ci] += S_k[16 rows x 4 co] W_k^T[4 co x ci]   A = S_k[r + m][4 j + q], B = Wt_k[4 j + q][m] (LDS)
// The pair-major B operand of the first product is fetched as float2 by the two lanes of a channel pair from two
// different rows and swapped (as in k_cheb_dw_mfma); the partial tiles leave in k_reduce_partials' layout.
template <int KMAX>
__global__ void __launch_bounds__(256)
k_big_bwd16(const float* __restrict__ S, const float* __restrict__ x, const float* __restrict__ Wt,
            float* __restrict__ dx, float* __restrict__ partial, long long rows, int N, int K, long long rows_per_wave) {
  extern __shared__ __align__(16) float bsm[];
  float* wl = bsm;                                   // [K][16 co][16 ci]
  float* red = bsm + KMAX * 256;                     // [3][KMAX + 1][64][4]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = lane & 15, q = lane >> 4;
  for (int i = threadIdx.x; i < K * 256; i += blockDim.x) wl[i] = Wt[i];
  __syncthreads();
  f32x4_t acc_w[KMAX], acc_b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < KMAX; ++k) acc_w[k] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const long long plane = rows * 16;
  const long long wid = (long long)blockIdx.x * 4 + wave;
  const long long r_begin = wid * rows_per_wave;
  const long long r_end = min(rows, r_begin + rows_per_wave);
  // element (row, channel c) of a pair-major plane = row * 16 - (row % N) * 14 + (c / 2) * 2 N + (c & 1)
  const long long pair_off = (long long)(m >> 1) * N * 2;   // this lane's pair as the B operand of the dW product
  for (long long r = r_begin; r < r_end; r += 16) {
    float ax[4], live[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long row_raw = r + 4 * u + q;
      live[u] = row_raw < r_end ? 1.f : 0.f;
      ax[u] = x[min(row_raw, rows - 1) * 16 + m] * live[u];
    }
    // offsets of the rows this lane fetches for the dW product: h = 0 -> instruction 2 h + (m & 1)
    long long off_w[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const long long row_mine = min(r + 4 * (2 * h + (m & 1)) + q, rows - 1);
      off_w[h] = row_mine * 16 - (long long)((int)row_mine % N) * 14 + pair_off;
    }
    // ... and of the row it feeds to the dX product (A operand: row r + m, channels 4 j + q)
    const long long row_x = min(r + m, rows - 1);
    const long long off_x = row_x * 16 - (long long)((int)row_x % N) * 14;
    f32x4_t acc_x = {0.f, 0.f, 0.f, 0.f};
    // plane k + 2's operands are in flight while plane k is multiplied (two waves per SIMD: the latency of the
    // per-plane loads is otherwise a dozen serial global round trips per trip)
    constexpr int PF = 2;
    float2 fw[PF + 1][2];
    float fa[PF + 1][4];
    auto fetch = [&](int k, int slot) {
      const float* pk = S + (long long)k * plane;
#pragma unroll
      for (int h = 0; h < 2; ++h) fw[slot][h] = *reinterpret_cast<const float2*>(pk + off_w[h]);
      // (plain loads: the dX operands below touch the same lines a second time -- nontemporal ones here cost 16 us)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = 4 * j + q;
        fa[slot][j] = pk[off_x + (long long)(c >> 1) * N * 2 + (c & 1)];   // (second touch of the same lines: cached)
      }
    };
#pragma unroll
    for (int k = 0; k < PF; ++k)
      if (k < K) fetch(k, k % (PF + 1));
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      if (k < K) {
        if (k + PF < K) fetch(k + PF, (k + PF) % (PF + 1));
        const int slot = k % (PF + 1);
        float bs[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const float2 f = fw[slot][h];
          const float gx = __shfl_xor(f.x, 1, 64), gy = __shfl_xor(f.y, 1, 64);
          bs[2 * h] = (m & 1) ? gy : f.x;
          bs[2 * h + 1] = (m & 1) ? f.y : gx;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc_w[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[u], bs[u], acc_w[k], 0, 0, 0);
        if (k == 0) {
#pragma unroll
          for (int u = 0; u < 4; ++u) acc_b = __builtin_amdgcn_mfma_f32_16x16x4f32(live[u], bs[u], acc_b, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc_x = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[slot][j], wl[(k * 16 + 4 * j + q) * 16 + m], acc_x, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // dx tile: lane (ci = m, q) holds the rows r + 4 q + j
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long long row = r + 4 * q + j;
      if (row < r_end) __builtin_nontemporal_store(acc_x[j], dx + row * 16 + m);
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j) red[(((wave - 1) * (KMAX + 1) + k) * 64 + lane) * 4 + j] = acc_w[k][j];
#pragma unroll
    for (int j = 0; j < 4; ++j) red[(((wave - 1) * (KMAX + 1) + KMAX) * 64 + lane) * 4 + j] = acc_b[j];
  }
  __syncthreads();
  if (wave == 0) {
    float* p = partial + (long long)blockIdx.x * (K * 16 + 1) * 16;
    auto total = [&](float v, int t, int j) {
      return ((v + red[((0 * (KMAX + 1) + t) * 64 + lane) * 4 + j]) + red[((1 * (KMAX + 1) + t) * 64 + lane) * 4 + j]) +
             red[((2 * (KMAX + 1) + t) * 64 + lane) * 4 + j];
    };
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (k < K) {
#pragma unroll
        for (int j = 0; j < 4; ++j) p[(long long)(k * 16 + 4 * q + j) * 16 + m] = total(acc_w[k][j], k, j);  // [ci = 4 q + j][co = m]
      }
    if (q == 0) p[(long long)(K * 16) * 16 + m] = total(acc_b[0], KMAX, 0);   // every row of the ones tile is db
  }
}

// Wt[k][co][ci] = W[k][ci][co]
__global__ void __launch_bounds__(256) k_w_transpose(const float* __restrict__ W, float* __restrict__ Wt, int K, int Cin,
                                                     int Cout) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= K * Cin * Cout) return;
  const int k = i / (Cin * Cout), rem = i - k * Cin * Cout, co = rem / Cin, ci = rem - co * Cin;
  Wt[i] = W[((long long)k * Cin + ci) * Cout + co];
}

// ------------------------------------------------------------------ host entry points
// Layout of the T_k stack of a layer (forward -> contraction, dW; also what tx_saved carries from mvh_cheb_conv_fwd to
// mvh_cheb_conv_bwd): pair-major planes where cheb_big.hip builds the stack AND both consumers are the kernels that
// read that layout (16 -> 16 channels on a level of 5120 .. 20480 vertices, i.e. BASELINE configs[3]'s level 0), rows
// [B][N][Cin] everywhere else.  Returns N for pair-major, 0 for rows.
static int tx_pair_major(const mvh_csr_t* lap, const float* x, const float* tx, int B, int N, int Cin, int Cout, int K) {
  if (K < 2 || Cin != 16 || Cout != 16 || !cheb_big_eligible(lap, B, N, Cin, K)) return 0;
  if (!dw_is_mfma((long long)B * N, Cout) || (((uintptr_t)x | (uintptr_t)tx) & 15) != 0) return 0;
  return N;
}

static int tx_forward(hipStream_t st, const mvh_csr_t* lap, const float* x, float* tx, long long plane,
                      int B, int Cin, int K, int pmN = 0) {
  // T_1 = L x ; T_k = 2 L T_{k-1} - T_{k-2}   (nn/conv.py:564-569)
  if (K > 1 && B > 0) {  // 5120 .. 20480 vertices: the whole stack in one launch (cheb_big.hip)
    bool big = false;
    if (int rc = try_cheb_big_tx(st, lap, x, tx, B, (int)(plane / ((long long)B * Cin)), Cin, K, pmN != 0, &big)) return rc;
    if (big) return MVH_OK;
  }
  if (pmN) return fail(MVH_ERR_INVALID, "cheb_conv: pair-major stack without the cheb_big kernel");
  for (int k = 1; k < K; ++k) {
    const float* prev = (k == 1) ? x : tx + (long long)(k - 2) * plane;
    const float* prev2 = (k == 1) ? nullptr : (k == 2 ? x : tx + (long long)(k - 3) * plane);
    int rc = launch_spmm(st, lap, prev, tx + (long long)(k - 1) * plane, nullptr, prev2,
                         k == 1 ? 1.f : 2.f, -1.f, B, Cin, false);
    if (rc) return rc;
  }
  return MVH_OK;
}


// ------------------------------------------------------------------ mostly-isolated Laplacians
// When the edge list only touches the leading n_active vertices (the final layer applies the
// 20-vertex edge list to 4998 vertices, cheb_VAE.py:288) every other vertex has L x = 0, so
// T_k x = c_k x with c_k = T_k(0) = 1, 0, -1, 0, ... : the convolution there is the per-vertex
// linear map x W_eff, W_eff = sum_k c_k W_k, and only the n_active x n_active block needs the
// recurrence (run as a strided sub-problem that overwrites the leading rows).
__device__ __forceinline__ float cheb_at_zero(int k) { return (k & 1) ? 0.f : ((k & 2) ? -1.f : 1.f); }

__global__ void __launch_bounds__(256) k_weff(const float* __restrict__ W, float* __restrict__ Weff, int K, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int k = 0; k < K; k += 2) s += cheb_at_zero(k) * W[(long long)k * n + i];
  Weff[i] = s;
}

// dW_k = dWsub_k + c_k (S_all - dWsub_0):  S_all = sum over ALL rows of x^T dpre (a K=1 pass)
__global__ void __launch_bounds__(256)
k_dw_combine(float* __restrict__ dW, const float* __restrict__ dWsub, const float* __restrict__ S, int K, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= K * n) return;
  const int k = i / n, e = i - k * n;
  dW[i] = dWsub[i] + cheb_at_zero(k) * (S[e] - dWsub[e]);
}

constexpr size_t kSplitScratchBytes = 64 * 1024;

// K = 1 weight gradient with a narrow dpre (the S_all pass of the split path, final 16 -> 3 layer):
//   S[ci][co] = sum_rows x[r][ci] dpre[r][co],  db[co] = sum_rows dpre[r][co].
// A streaming reduction (HBM-bound): QV = CX/4 lanes share a row, each holding 16 B of x and the
// CD dpre values; lanes with the same channel quad are summed with shuffles, waves through LDS,
// blocks through `partial` [grid][(CX + 1) * CD] and a one-block finish (fixed order).
constexpr int kXtyGrid = 256;   // (1024 partial sets made the one-block finish 16 us: 51 dependent loads per thread)

template <int CD>
__global__ void __launch_bounds__(256)
k_xty_small(const float* __restrict__ x, const float* __restrict__ d, float* __restrict__ partial, long long rows,
            int CX, int x_bf16) {
  const int QV = CX >> 2, RPW = 64 / QV;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane % QV, rsub = lane / QV;
  float acc[4][CD], dsum[CD];
#pragma unroll
  for (int c = 0; c < CD; ++c) {
    dsum[c] = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t][c] = 0.f;
  }
  const long long stride = (long long)gridDim.x * 4 * RPW;
#pragma unroll 4
  for (long long r = ((long long)blockIdx.x * 4 + wave) * RPW + rsub; r < rows; r += stride) {
    const float4 xv = load4_any(x, r * CX + 4 * q, x_bf16 != 0);  // (wave-uniform select, one load per trip)
    float dv[CD];
#pragma unroll
    for (int c = 0; c < CD; ++c) dv[c] = d[r * CD + c];
#pragma unroll
    for (int c = 0; c < CD; ++c) {
      acc[0][c] = fmaf(xv.x, dv[c], acc[0][c]);
      acc[1][c] = fmaf(xv.y, dv[c], acc[1][c]);
      acc[2][c] = fmaf(xv.z, dv[c], acc[2][c]);
      acc[3][c] = fmaf(xv.w, dv[c], acc[3][c]);
      dsum[c] += dv[c];
    }
  }
  // lanes q, q + QV, q + 2 QV, ... hold the same channel quad
  for (int off = QV; off < 64; off <<= 1) {
#pragma unroll
    for (int c = 0; c < CD; ++c) {
      dsum[c] += __shfl_xor(dsum[c], off, 64);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t][c] += __shfl_xor(acc[t][c], off, 64);
    }
  }
  __shared__ float red[4][(32 + 1) * 4];  // [wave][(CX + 1) * CD], CX <= 32, CD <= 4
  if (lane < QV) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int c = 0; c < CD; ++c) red[wave][(4 * q + t) * CD + c] = acc[t][c];
    if (q == 0)
#pragma unroll
      for (int c = 0; c < CD; ++c) red[wave][CX * CD + c] = dsum[c];
  }
  __syncthreads();
  const int nout = (CX + 1) * CD;
  if ((int)threadIdx.x < nout)
    partial[(long long)blockIdx.x * nout + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ void __launch_bounds__(1024)
k_xty_finish(const float* __restrict__ partial, int nblocks, int nout, int n_s, float* __restrict__ S,
             float* __restrict__ db) {
  // thread (g, e): group g of 1024/nout_pad sums blocks g, g + G, ...; groups are then summed in order
  __shared__ float red[1024];
  const int G = 1024 / nout;  // >= 7 for nout <= 132
  const int g = threadIdx.x / nout, e = threadIdx.x - g * nout;
  float s = 0.f;
  if (g < G)
    for (int b = g; b < nblocks; b += G) s += partial[(long long)b * nout + e];
  red[threadIdx.x] = (g < G) ? s : 0.f;
  __syncthreads();
  if ((int)threadIdx.x < nout) {
    float t = 0.f;
    for (int k = 0; k < G; ++k) t += red[k * nout + threadIdx.x];
    if ((int)threadIdx.x < n_s) S[threadIdx.x] = t;
    else if (db) db[threadIdx.x - n_s] = t;
  }
}

// *handled == false: not eligible (caller keeps the LDS-kernel K = 1 pass)
static int try_xty_small(hipStream_t st, const float* x, const float* dpre, float* S, float* db, long long rows,
                         int Cin, int Cout, float* partial, size_t part_bytes, bool* handled, bool x_bf16 = false) {
  *handled = false;
  if ((Cout != 3 && Cout != 4) || (Cin != 8 && Cin != 16 && Cin != 32)) return MVH_OK;
  if (((uintptr_t)x % 16) != 0 || rows <= 0) return MVH_OK;
  const int nout = (Cin + 1) * Cout;
  const int rpb = 4 * (64 / (Cin >> 2));
  int grid = (int)((rows + rpb - 1) / rpb);
  if (grid > kXtyGrid) grid = kXtyGrid;
  if (!partial || part_bytes < (size_t)grid * nout * sizeof(float)) return MVH_OK;
  const int xb = x_bf16 ? 1 : 0;
  if (dbg().skip_xty) {}   // TIMING ONLY (results invalid): what would S from the loss launch be worth?
  else if (Cout == 3) hipLaunchKernelGGL((k_xty_small<3>), dim3(grid), dim3(256), 0, st, x, dpre, partial, rows, Cin, xb);
  else hipLaunchKernelGGL((k_xty_small<4>), dim3(grid), dim3(256), 0, st, x, dpre, partial, rows, Cin, xb);
  MVH_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_xty_finish, dim3(1), dim3(1024), 0, st, partial, grid, nout, Cin * Cout, S, db);
  MVH_LAUNCH_CHECK();
  *handled = true;
  return MVH_OK;
}

// fallback producer of the ReLU sign bytes (the LDS kernel writes them in its epilogue)
__global__ void __launch_bounds__(256)
k_relu_bits(const float* __restrict__ out, uint8_t* __restrict__ bits, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 o = reinterpret_cast<const float4*>(out)[i];
  bits[i] = (uint8_t)((o.x > 0.f ? 1 : 0) | (o.y > 0.f ? 2 : 0) | (o.z > 0.f ? 4 : 0) | (o.w > 0.f ? 8 : 0));
}

static bool split_eligible(const mvh_csr_t* lap, int N, int Cin, int Cout, int K);
bool conv_split_eligible(const mvh_csr_t* lap, int N, int Cin, int Cout, int K) { return split_eligible(lap, N, Cin, Cout, K); }
static bool split_eligible(const mvh_csr_t* lap, int N, int Cin, int Cout, int K) {
  if (dbg().force_generic) return false;
  return lap->sub && lap->n_active > 0 && 4 * lap->n_active <= N && lap->sub->n_rows == lap->n_active &&
         (size_t)(K + 2) * Cin * Cout * sizeof(float) + 1024 <= kSplitScratchBytes;
}

}  // namespace mvh

using namespace mvh;

extern "C" size_t mvh_cheb_conv_ws_bytes(int32_t B, int32_t N, int32_t Cin, int32_t Cout, int32_t K) {
  (void)Cout;
  return kLdsWpackBytes + kSplitScratchBytes + align_up((size_t)(K > 1 ? K - 1 : 0) * B * N * Cin * sizeof(float), 256) + 256;
}

static int conv_args_ok(const mvh_csr_t* lap, int B, int N, int Cin, int Cout, int K) {
  if (int rc = check_csr(lap, "lap")) return rc;
  MVH_REQUIRE(B >= 0 && N >= 0 && Cin > 0 && Cout > 0, "cheb_conv: bad sizes B=%d N=%d Cin=%d Cout=%d", B, N, Cin, Cout);
  MVH_REQUIRE(K > 0, "cheb_conv: K must be > 0 (nn/conv.py:445)");
  MVH_REQUIRE(lap->n_rows == N && lap->n_cols == N,
              "cheb_conv: Laplacian is %dx%d but x has %d vertices", lap->n_rows, lap->n_cols, N);
  return MVH_OK;
}

extern "C" int mvh_cheb_conv_fwd(mvh_stream_t stream, const mvh_csr_t* lap, const float* x, const float* W,
                                 const float* bias, float* out, float* tx_saved, int32_t B, int32_t N,
                                 int32_t Cin, int32_t Cout, int32_t K, int32_t act, void* ws,
                                 size_t ws_bytes) {
  return cheb_conv_fwd_impl((hipStream_t)stream, lap, x, W, bias, out, tx_saved, B, N, Cin, Cout, K, act, ws, ws_bytes,
                            nullptr, nullptr, nullptr);
}

extern "C" int mvh_cheb_conv_fwd_signs(mvh_stream_t stream, const mvh_csr_t* lap, const float* x, const float* W,
                                       const float* bias, float* out, uint8_t* relu_signs, int32_t B, int32_t N,
                                       int32_t Cin, int32_t Cout, int32_t K, void* ws, size_t ws_bytes) {
  MVH_REQUIRE(relu_signs != nullptr, "cheb_conv_fwd_signs: null relu_signs");
  return cheb_conv_fwd_impl((hipStream_t)stream, lap, x, W, bias, out, nullptr, B, N, Cin, Cout, K, MVH_ACT_RELU, ws,
                            ws_bytes, nullptr, nullptr, nullptr, relu_signs);
}

int mvh::cheb_conv_fwd_impl(hipStream_t stream, const mvh_csr_t* lap, const float* x, const float* W,
                            const float* bias, float* out, float* tx_saved, int B, int N, int Cin, int Cout, int K,
                            int act, void* ws, size_t ws_bytes, const float* prepacked, const mvh_csr_t* pool,
                            float* pooled, uint8_t* bits_out, const float* weff_pre, const ConvIO& io) {
  if (int rc = conv_args_ok(lap, B, N, Cin, Cout, K)) return rc;
  const bool bf = io.any();
  MVH_REQUIRE(!bf || !tx_saved, "cheb_conv_fwd: bf16 storage does not keep a T_k stack");
  MVH_REQUIRE(!bits_out || Cout % 4 == 0, "cheb_conv_fwd: sign bytes need Cout %% 4 == 0");
  auto finish = [&](bool bits_done) -> int {  // pooling / sign bytes the main kernel did not produce itself
    if (bits_out && !bits_done) {
      const long long n = (long long)B * N * (Cout / 4);
      hipLaunchKernelGGL(k_relu_bits, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, out, bits_out, n);
      MVH_LAUNCH_CHECK();
    }
    return MVH_OK;
  };
  MVH_REQUIRE(x && W && out, "cheb_conv_fwd: null tensor");
  if ((long long)B * N == 0) return MVH_OK;
  hipStream_t st = (hipStream_t)stream;
  const long long rows = (long long)B * N, plane = rows * Cin;
  if (io.x_map) {  // strided x: the LDS-resident kernel reads it through its row map, nothing else can
    MVH_REQUIRE(!tx_saved && !bf && io.x_bs > 0, "cheb_conv_fwd: a strided x comes without a saved stack, as fp32");
    if (split_eligible(lap, N, Cin, Cout, K))
      return fail(MVH_ERR_UNSUPPORTED, "cheb_conv_fwd: strided x on a mostly-isolated Laplacian (per-vertex map path)");
  }
  if (!tx_saved && ws && ws_bytes >= kLdsWpackBytes + kSplitScratchBytes && split_eligible(lap, N, Cin, Cout, K)) {
    const float* weff = weff_pre;
    if (!weff) {
      float* wbuf = (float*)((char*)ws + kLdsWpackBytes);
      hipLaunchKernelGGL(k_weff, dim3(cdiv(Cin * Cout, 256)), dim3(256), 0, st, W, wbuf, K, Cin * Cout);
      MVH_LAUNCH_CHECK();
      weff = wbuf;
    }
    MVH_REQUIRE(!io.out, "cheb_conv_fwd: the split path writes fp32 rows");
    if (!io.out_lazy)
      if (int rc = launch_contract(st, x, nullptr, weff, bias, out, rows, Cin, Cout, 1, act, io.x)) return rc;
    bool handled = false;
    LdsConvOpts so;
    so.prepacked = prepacked; so.in_bs = N; so.out_bs = N; so.in_bf16 = io.x;
    if (int rc = try_cheb_lds(st, lap->sub, x, nullptr, W, bias, out, B, lap->n_active, Cin, Cout, K, act, false,
                              (float*)ws, &handled, so)) return rc;
    if (handled) {
      if (pool && pooled)
        if (int rc = launch_spmm(st, pool, out, pooled, nullptr, nullptr, 1.f, 0.f, B, Cout, true, io.pooled)) return rc;
      return finish(false);
    }  // otherwise fall through: the full path rewrites every row
    MVH_REQUIRE(!io.out_lazy, "cheb_conv_fwd: out_lazy without the LDS kernel for the connected block");
    if (bf) return fail(MVH_ERR_UNSUPPORTED, "cheb_conv_fwd: bf16 storage needs the LDS-resident kernel for the active block");
  }
  MVH_REQUIRE(!io.out_lazy, "cheb_conv_fwd: out_lazy on a layer that is not on the split path");
  // 16 -> 16 on a level that carries a vertex-patch plan: (mesh, patch) workgroups with all channels, contraction on
  // v_mfma_f32_16x16x4_f32 (cheb_patch.hip).  No fused pooling there: the decoder's LAST stage and module-level calls.
  if (!tx_saved && !bf && !pool && patch_eligible(lap, N, Cin, Cout, K) &&
      (((uintptr_t)x | (uintptr_t)out | (uintptr_t)bias | (uintptr_t)W) & 15) == 0)
  {
    const bool map = io.map_out && io.map_w && !dbg().no_patch_map;
    if (int rc = launch_patch_fwd(st, lap, x, W, bias, out, bits_out, B, N, K, act, io.x_map, io.x_bs, io.x_unpool, io.x_store,
                                  map ? io.map_w : nullptr, map ? io.map_out : nullptr, io.map_c, io.map_n0)) return rc;
    if (map && io.map_done) *io.map_done = true;
    return MVH_OK;
  }
  MVH_REQUIRE(!io.x_unpool, "cheb_conv_fwd: x_unpool on a layer that does not take the vertex-patch kernel");
  // the first layer (<= 4 -> 16) in front of its one-hot pooling, nobody reading the other rows: recurrence on the input
  // side in the patch image, pooled rows + their sign bytes + the weight gradient's T_k stack out of one launch (`out`
  // itself is never written: io.out, its storage type, does not matter)
  if (!tx_saved && !io.x && !io.x_map && pool && pooled && io.out_dead && !dbg().keep_enc_out &&
      act == MVH_ACT_RELU && patch_enc0_eligible(lap, pool, N, Cin, Cout, K) &&
      (((uintptr_t)pooled | (uintptr_t)bias | (uintptr_t)io.stack_out) & 15) == 0) {
    if (int rc = launch_patch_enc0(st, lap, pool, x, W, bias, pooled, io.pooled, bits_out, io.stack_out, B, N, Cin, K, act))
      return rc;
    if (io.stack_out && io.stack_done) *io.stack_done = true;
    return MVH_OK;
  }
  if (!tx_saved) {  // fused path: one launch, no T_k stack
    bool handled = false;
    float* wpack = (ws && ws_bytes >= kLdsWpackBytes) ? (float*)ws : nullptr;
    LdsConvOpts fo;
    fo.prepacked = prepacked;
    fo.in_bf16 = io.x; fo.out_bf16 = io.out; fo.pooled_bf16 = io.pooled; fo.prepacked_h = io.wh;
    if (io.x_map) { fo.in_map = io.x_map; fo.in_bs = io.x_bs; }
    if (pool && pool->sel_inv && pooled) { fo.pool_inv = pool->sel_inv; fo.pooled = pooled; fo.pooled_bs = pool->n_rows; }
    fo.bits_out = bits_out;
    fo.out_dead = io.out_dead && fo.pool_inv != nullptr && bits_out != nullptr;
    bool pooled_in_kernel = fo.pool_inv != nullptr;
    if (pool && pooled && !fo.pool_inv) {  // general operator (the decoder's upsampling): pooled from LDS in the epilogue
      fo.out_pool_t = pool; fo.pooled = pooled;
      if (int rc = try_cheb_lds(st, lap, x, nullptr, W, bias, out, B, N, Cin, Cout, K, act, false, wpack, &handled, fo))
        return rc;
      if (handled) return finish(true);
      fo.out_pool_t = nullptr; fo.pooled = nullptr;
    }
    if (int rc = try_cheb_lds(st, lap, x, nullptr, W, bias, out, B, N, Cin, Cout, K, act, false, wpack, &handled, fo))
      return rc;
    if (handled && pool && !pooled_in_kernel) {
      MVH_REQUIRE(!io.out, "cheb_conv_fwd: bf16 rows are pooled inside the conv kernel only");
      if (int rc = launch_spmm(st, pool, out, pooled, nullptr, nullptr, 1.f, 0.f, B, Cout, true, io.pooled)) return rc;
    }
    if (handled) return finish(true);
  }
  if (bf)
    return fail(MVH_ERR_UNSUPPORTED, "cheb_conv_fwd: bf16 storage exists on the LDS-resident kernels only (N=%d %d->%d K=%d)",
                N, Cin, Cout, K);
  if (io.x_map)
    return fail(MVH_ERR_UNSUPPORTED, "cheb_conv_fwd: strided x is read by the LDS-resident kernels only (N=%d %d->%d K=%d)",
                N, Cin, Cout, K);
  // the first layer of a streaming level in front of its one-hot pooling, with the caller's stack buffer (the step engine,
  // ConvIO::stack_out): T_k x is kept at the SELECTED rows only (cheb_big.hip -> the stack of cheb_tstack.hip) and the
  // layer's output is contracted from it at those rows -- no full [K][B][N][Cin] stack, no [B][N][Cout] output, no pooling pass
  if (io.stack_out && pool && pooled && io.out_dead && bits_out && act == MVH_ACT_RELU && Cout == 16 && N + 1 > 5120 &&
      !io.pooled && tstack_eligible(lap, pool, N, Cin, Cout, K)) {
    if (int rc = launch_big_tstack(st, lap, pool, x, io.stack_out, B, N, Cin, K)) return rc;
    if (int rc = launch_stack_contract(st, pool, io.stack_out, W, bias, pooled, bits_out, B, N, Cin, K)) return rc;
    if (io.stack_done) *io.stack_done = true;
    return MVH_OK;
  }
  float* tx = tx_saved;
  if (!tx && K > 1) {
    MVH_REQUIRE(ws && ws_bytes >= mvh_cheb_conv_ws_bytes(B, N, Cin, Cout, K), "cheb_conv_fwd: workspace too small");
    tx = (float*)((char*)ws + kLdsWpackBytes + kSplitScratchBytes);
  }
  const int pmN = tx_pair_major(lap, x, tx, B, N, Cin, Cout, K);
  if (int rc = tx_forward(st, lap, x, tx, plane, B, Cin, K, pmN)) return rc;
  ContractExtra ex;
  const bool bits_here = bits_out && act == MVH_ACT_RELU && (Cout == 8 || Cout == 16 || Cout == 32) && !dbg().no_contract_extras;
  if (bits_here) ex.bits = bits_out;
  const bool map_here = io.map_out && io.map_w && Cout == 16 && !dbg().no_patch_map && !dbg().no_contract_extras;
  if (map_here) { ex.map_w = io.map_w; ex.map_out = io.map_out; ex.map_c = io.map_c; ex.map_n0 = io.map_n0; ex.N = N; }
  if (int rc = launch_contract(st, x, tx, W, bias, out, rows, Cin, Cout, K, act, false, pmN, ex)) return rc;
  if (map_here && io.map_done) *io.map_done = true;
  if (pool && pooled)
    if (int rc = launch_spmm(st, pool, out, pooled, nullptr, nullptr, 1.f, 0.f, B, Cout, true)) return rc;
  return finish(bits_here);
}

extern "C" size_t mvh_cheb_conv_bwd_ws_bytes(int32_t B, int32_t N, int32_t Cin, int32_t Cout, int32_t K) {
  const size_t rows = (size_t)B * N;
  size_t tx = align_up((size_t)(K > 1 ? K - 1 : 0) * rows * Cin * sizeof(float), 256);
  size_t g = align_up((size_t)K * rows * Cin * sizeof(float), 256);
  size_t part = align_up((size_t)dw_grid((long long)rows) * ((size_t)K * Cin + 1) * Cout * sizeof(float), 256);
  const size_t part_lds = align_up(cheb_dw_lds_ws_bytes(B, N, Cin, Cout, K), 256);
  if (part_lds > part) part = part_lds;
  return kLdsWpackBytes + kSplitScratchBytes + tx + g + part + 256;
}

extern "C" int mvh_cheb_conv_bwd(mvh_stream_t stream, const mvh_csr_t* lap, const mvh_csr_t* lap_t,
                                 const float* x, const float* W, const float* out, const float* dout,
                                 const float* tx_saved, float* dx, float* dW, float* db, int32_t B,
                                 int32_t N, int32_t Cin, int32_t Cout, int32_t K, int32_t act, void* ws,
                                 size_t ws_bytes) {
  return cheb_conv_bwd_impl((hipStream_t)stream, lap, lap_t, x, W, out, dout, tx_saved, dx, dW, db, B, N, Cin, Cout, K,
                            act, ws, ws_bytes, nullptr, nullptr, nullptr);
}

extern "C" int mvh_cheb_conv_bwd_signs(mvh_stream_t stream, const mvh_csr_t* lap, const mvh_csr_t* lap_t,
                                       const float* x, const float* W, const float* out, const uint8_t* relu_signs,
                                       const float* dout, float* dx, float* dW, float* db, int32_t B, int32_t N,
                                       int32_t Cin, int32_t Cout, int32_t K, void* ws, size_t ws_bytes) {
  MVH_REQUIRE(relu_signs != nullptr && Cout % 4 == 0, "cheb_conv_bwd_signs: relu_signs need Cout %% 4 == 0");
  return cheb_conv_bwd_impl((hipStream_t)stream, lap, lap_t, x, W, out, dout, nullptr, dx, dW, db, B, N, Cin, Cout, K,
                            MVH_ACT_RELU, ws, ws_bytes, nullptr, nullptr, nullptr, relu_signs);
}

// bf16-storage forms of the two fused ops (BASELINE configs[1] "bf16"): x / out / dout / dx are bf16 tensors,
// parameters and their gradients fp32, accumulation fp32.  LDS-resident kernels only (MVH_ERR_UNSUPPORTED otherwise).
extern "C" int mvh_cheb_conv_fwd_bf16(mvh_stream_t stream, const mvh_csr_t* lap, const void* x, const float* W,
                                      const float* bias, void* out, uint8_t* relu_signs, int32_t B, int32_t N,
                                      int32_t Cin, int32_t Cout, int32_t K, int32_t act, void* ws, size_t ws_bytes) {
  MVH_REQUIRE(Cin % 4 == 0 && Cout % 4 == 0, "cheb_conv_fwd_bf16: channel counts must be multiples of 4");
  MVH_REQUIRE(act != MVH_ACT_RELU || relu_signs, "cheb_conv_fwd_bf16: the ReLU form returns its sign bytes");
  ConvIO io;
  io.x = io.out = true;
  return cheb_conv_fwd_impl((hipStream_t)stream, lap, (const float*)x, W, bias, (float*)out, nullptr, B, N, Cin, Cout, K, act,
                            ws, ws_bytes, nullptr, nullptr, nullptr, act == MVH_ACT_RELU ? relu_signs : nullptr, nullptr, io);
}

extern "C" int mvh_cheb_conv_bwd_bf16(mvh_stream_t stream, const mvh_csr_t* lap, const mvh_csr_t* lap_t, const void* x,
                                      const float* W, const uint8_t* relu_signs, const void* dout, void* dx, float* dW,
                                      float* db, int32_t B, int32_t N, int32_t Cin, int32_t Cout, int32_t K, int32_t act,
                                      void* ws, size_t ws_bytes) {
  MVH_REQUIRE(Cin % 4 == 0 && Cout % 4 == 0, "cheb_conv_bwd_bf16: channel counts must be multiples of 4");
  MVH_REQUIRE(act != MVH_ACT_RELU || relu_signs, "cheb_conv_bwd_bf16: the ReLU form reads the forward's sign bytes");
  ConvIO io;
  io.x = io.dout = io.dx = true;
  return cheb_conv_bwd_impl((hipStream_t)stream, lap, lap_t, (const float*)x, W, nullptr, (const float*)dout, nullptr,
                            (float*)dx, dW, db, B, N, Cin, Cout, K, act, ws, ws_bytes, nullptr, nullptr, nullptr,
                            act == MVH_ACT_RELU ? relu_signs : nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr,
                            nullptr, io);
}

int mvh::cheb_conv_bwd_impl(hipStream_t stream, const mvh_csr_t* lap, const mvh_csr_t* lap_t, const float* x,
                            const float* W, const float* out, const float* dout, const float* tx_saved, float* dx,
                            float* dW, float* db, int B, int N, int Cin, int Cout, int K, int act, void* ws,
                            size_t ws_bytes, const float* prepacked_bwd, const mvh_csr_t* dout_pool, bool* fused_ok,
                            const uint8_t* out_bits, const float* weff_pre, DwReduceEntry* defer,
                            float* defer_part, size_t defer_bytes, bool* deferred, const mvh_csr_t* dx_pool_t,
                            float* dx_pooled, const ConvIO& io) {
  if (int rc = conv_args_ok(lap, B, N, Cin, Cout, K)) return rc;
  const bool bf = io.any();
  MVH_REQUIRE(!bf || !tx_saved, "cheb_conv_bwd: bf16 storage does not keep a T_k stack");
  MVH_REQUIRE(!bf || act != MVH_ACT_RELU || out_bits, "cheb_conv_bwd: bf16 storage takes the ReLU mask as sign bytes");
  if (int rc = check_csr(lap_t, "lap_t")) return rc;
  MVH_REQUIRE(lap_t->n_rows == N && lap_t->n_cols == N, "cheb_conv_bwd: lap_t shape mismatch");
  MVH_REQUIRE(x && W && dout && (dW || dx), "cheb_conv_bwd: null tensor");
  MVH_REQUIRE(act != MVH_ACT_RELU || out || out_bits, "cheb_conv_bwd: relu backward needs the forward output (or its sign bytes)");
  MVH_REQUIRE(ws && ws_bytes >= mvh_cheb_conv_bwd_ws_bytes(B, N, Cin, Cout, K), "cheb_conv_bwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const long long rows = (long long)B * N, plane = rows * Cin;
  if (act != MVH_ACT_RELU || Cout % 4 != 0) out_bits = nullptr;
  if (deferred) *deferred = false;
  if (!defer || !defer_part || !deferred) defer = nullptr;
  char* p = (char*)ws;
  float* wpack = (float*)p;
  p += kLdsWpackBytes;
  float* split = (float*)p;
  p += kSplitScratchBytes;
  float* tx_ws = (float*)p;
  p += align_up((size_t)(K > 1 ? K - 1 : 0) * plane * sizeof(float), 256);
  float* G = (float*)p;
  p += align_up((size_t)K * plane * sizeof(float), 256);
  float* partial = (float*)p;
  if (rows == 0) {
    if (dW) MVH_HIP(hipMemsetAsync(dW, 0, (size_t)K * Cin * Cout * sizeof(float), st));
    if (db) MVH_HIP(hipMemsetAsync(db, 0, (size_t)Cout * sizeof(float), st));
    return MVH_OK;
  }
  if (dout_pool) {
    // `dout` is the gradient of the POOLED output: the un-pooling (a scatter into zeros in the
    // reference's autograd) is folded into the loads of the LDS kernels; all or nothing.
    MVH_REQUIRE(fused_ok != nullptr, "cheb_conv_bwd: fused_ok required with dout_pool");
    *fused_ok = false;
    if (dW && !dx && (dout_pool->flags & MVH_CSR_SELECTION) && dout_pool->col && K > 1 && N + 1 > 5120 && !bf &&
        dw_is_mfma((long long)B * dout_pool->n_rows, Cout) && !dbg().force_generic && !dbg().no_dw_rows &&
        tx_pair_major(lap, x, tx_saved ? tx_saved : tx_ws, B, N, Cin, Cout, K) == 0) {
      // streaming level (BASELINE configs[3]'s first layer): the matrix-pipe weight-gradient kernel walks the pooled
      // rows only -- no un-pooling launch, a quarter of the reduction
      const float* tx = tx_saved;
      if (!tx) {
        if (int rc = tx_forward(st, lap, x, tx_ws, plane, B, Cin, K, 0)) return rc;
        tx = tx_ws;
      }
      if (int rc = launch_dw(st, x, tx, dout, out, partial, dW, db, (long long)B * dout_pool->n_rows, Cin, Cout, K, act, 0,
                             dout_pool->col, dout_pool->n_rows, N)) return rc;
      *fused_ok = true;
      return MVH_OK;
    }
    if (tx_saved || !dout_pool->sel_inv) return MVH_OK;
    const float* mask = act == MVH_ACT_RELU ? out : nullptr;
    size_t pbytes = (size_t)((char*)ws + ws_bytes - (char*)partial);
    if (defer) { partial = defer_part; pbytes = defer_bytes; }
    bool ok_dw = (dW == nullptr), ok_dx = (dx == nullptr);
    LdsConvOpts bo;
    bo.prepacked = prepacked_bwd; bo.in_map = dout_pool->sel_inv; bo.in_bs = dout_pool->n_rows; bo.mask_bs = N;
    bo.mask_bits = out_bits;
    bo.in_bf16 = io.dout; bo.out_bf16 = io.dx;
    bo.dry_run = true;
    if (!ok_dw)
      if (int rc = try_cheb_dw_lds(st, lap, x, dout, mask, dW, db, B, N, Cin, Cout, K, partial, pbytes, &ok_dw, 0,
                                   dout_pool->sel_inv, dout_pool->n_rows, true, out_bits, nullptr, io.x, io.dout)) return rc;
    if (!ok_dx)
      if (int rc = try_cheb_lds(st, lap_t, dout, mask, W, nullptr, dx, B, N, Cin, Cout, K, act, true, wpack, &ok_dx, bo))
        return rc;
    if (!ok_dw || !ok_dx) return MVH_OK;  // caller un-pools explicitly and calls again without dout_pool
    bool h = false;
    if (dW) {
      if (int rc = try_cheb_dw_lds(st, lap, x, dout, mask, dW, db, B, N, Cin, Cout, K, partial, pbytes, &h, 0,
                                   dout_pool->sel_inv, dout_pool->n_rows, false, out_bits, defer, io.x, io.dout)) return rc;
      if (defer && h) *deferred = true;
    }
    if (dx) {
      bo.dry_run = false;
      if (int rc = try_cheb_lds(st, lap_t, dout, mask, W, nullptr, dx, B, N, Cin, Cout, K, act, true, wpack, &h, bo))
        return rc;
    }
    *fused_ok = true;
    return MVH_OK;
  }
  // vertex-patch plan on this level (cheb_patch.hip): dX (+ its pooling) and the dW / db partial tiles from ONE launch
  // (bf16 storage: x, dout and dx all bf16 -- the whole step's mode --, fp32 arithmetic; the debug switch no_l0h, which
  //  selects the general kernels for the bf16 rows of this level, switches this form off as well)
  const bool bf_patch = io.x && io.dout && !io.x_map && (dx == nullptr || (io.dx && (!dx_pooled || io.dx_pooled))) && !dbg().no_l0h &&
                        !dbg().no_patch_bf16;
  if (!dout_pool && !tx_saved && (!bf || bf_patch) && !dbg().no_patch_bwd && patch_eligible(lap_t, N, Cin, Cout, K) &&
      (act != MVH_ACT_RELU || out_bits) &&
      (((uintptr_t)x | (uintptr_t)dout | (uintptr_t)dx | (uintptr_t)dx_pooled | (uintptr_t)W | (uintptr_t)partial |
        (uintptr_t)defer_part) & 15) == 0) {
    const mvh_patch_plan_t* pl = lap_t->patch;
    const bool want_pool = dx && dx_pool_t && dx_pooled;
    const bool pooled = want_pool && pl->n_pool_rows == dx_pool_t->n_rows && pl->pool_rowptr == dx_pool_t->rowptr;
    float* part = defer ? defer_part : partial;
    const size_t pbytes = defer ? defer_bytes : (size_t)((char*)ws + ws_bytes - (char*)partial);
    if (!dW || pbytes >= patch_part_bytes(lap_t, B, K)) {
      DwReduceEntry e{};
      if (int rc = launch_patch_bwd(st, lap_t, x, W, dout, act == MVH_ACT_RELU ? out_bits : nullptr, io.src3_g, io.src3_w,
                                    io.src3_n, pooled ? dx_pooled : dx, pooled, part, pbytes, &e, dW, db, B, N, K, io.x_map,
                                    io.x_bs, io.x, io.dout, pooled ? io.dx_pooled : io.dx)) return rc;
      if (dW) {
        if (defer) {
          *defer = e;
          *deferred = true;
        } else {
          DwReduceTable t;
          t.n = 1;
          t.e[0] = e;
          if (int rc = launch_dw_reduce_all(st, t)) return rc;
        }
      }
      if (want_pool && !pooled) {
        MVH_REQUIRE(!bf, "cheb_conv_bwd: bf16 storage pools dx inside the patch kernel only");
        return launch_spmm(st, dx_pool_t, dx, dx_pooled, nullptr, nullptr, 1.f, 0.f, B, Cin, true);
      }
      return MVH_OK;
    }
  }
  bool dw_done = (dW == nullptr);  // dW == NULL: dX-only call (the step engine runs dW on a side stream)
  bool dx_done = (dx == nullptr);
  if (io.x_map && dW) {  // strided x: only the LDS-resident dW kernel reads it (through its row map)
    MVH_REQUIRE(!tx_saved && !bf && io.x_bs > 0 && !dout_pool, "cheb_conv_bwd: a strided x comes without a saved stack, as fp32");
    bool ok = false;
    if (!split_eligible(lap, N, Cin, Cout, K) && N + 1 <= 5120)
      if (int rc = try_cheb_dw_lds(stream, lap, x, dout, nullptr, dW, db, B, N, Cin, Cout, K, partial,
                                   (size_t)((char*)ws + ws_bytes - (char*)partial), &ok, 0, nullptr, 0, true)) return rc;
    if (!ok)
      return fail(MVH_ERR_UNSUPPORTED, "cheb_conv_bwd: strided x is read by the LDS-resident dW kernel only (N=%d %d->%d K=%d)",
                  N, Cin, Cout, K);
  }
  // 16 -> 16 on a level of 5120 .. 20480 vertices with BOTH gradients asked for: one T_k(dpre) stack, one pass over it
  // (k_big_bwd16); the forward's stack (tx_saved) is not read
  if (dW && dx && K > 1 && K <= 12 && Cin == 16 && Cout == 16 && cheb_big_eligible(lap_t, B, N, Cout, K) &&
      (lap->flags & MVH_CSR_SYMMETRIC) && (((uintptr_t)dout | (uintptr_t)out | (uintptr_t)G | (uintptr_t)dx | (uintptr_t)x) & 15) == 0 &&
      !dbg().no_dx_tstack && !dbg().no_bwd_fused && rows >= 4096 && (act != MVH_ACT_RELU || out) && !bf) {   // (the mask and every
    // operand are read as fp32 rows: bf16 tensors must reach the MVH_ERR_UNSUPPORTED check below, never this kernel)
    float* Wt = split;   // [K][Cout][Cin] = W_k^T
    hipLaunchKernelGGL(k_w_transpose, dim3(cdiv(K * Cin * Cout, 256)), dim3(256), 0, st, W, Wt, K, Cin, Cout);
    MVH_LAUNCH_CHECK();
    bool big = false;
    if (int rc = try_cheb_big_tx(st, lap_t, dout, G, B, N, Cout, K, true, &big, act == MVH_ACT_RELU ? out : nullptr, true))
      return rc;
    if (big) {
      int Gb = dw_grid(rows);
      const long long waves = max(4ll, min(2048ll, rows / 256));
      Gb = min(Gb, (int)((waves + 3) / 4));
      long long rpw = (rows + (long long)Gb * 4 - 1) / ((long long)Gb * 4);
      rpw = (rpw + 15) / 16 * 16;
      const int n = (K * 16 + 1) * 16;
      if (K <= 6) {
        const size_t lds = (size_t)(6 * 256 + 3 * 7 * 256) * sizeof(float);
        hipLaunchKernelGGL((k_big_bwd16<6>), dim3(Gb), dim3(256), lds, st, G, x, Wt, dx, partial, rows, N, K, rpw);
      } else {
        const size_t lds = (size_t)(12 * 256 + 3 * 13 * 256) * sizeof(float);
        hipLaunchKernelGGL((k_big_bwd16<12>), dim3(Gb), dim3(256), lds, st, G, x, Wt, dx, partial, rows, N, K, rpw);
      }
      MVH_LAUNCH_CHECK();
      hipLaunchKernelGGL(k_reduce_partials, dim3(cdiv(n, 64)), dim3(Gb >= 128 ? 1024 : 256), 0, st, partial, Gb, n, K * 256, dW, db);
      MVH_LAUNCH_CHECK();
      if (dx_pool_t && dx_pooled)
        return launch_spmm(st, dx_pool_t, dx, dx_pooled, nullptr, nullptr, 1.f, 0.f, B, Cin, true);
      return MVH_OK;
    }
  }
  const bool split_ok = !tx_saved && split_eligible(lap, N, Cin, Cout, K) && split_eligible(lap_t, N, Cin, Cout, K);
  const int CC = Cin * Cout;
  if (split_ok && !dw_done) {  // dW_k = dWsub_k + c_k (S_all - dWsub_0), see k_dw_combine
    float* S = (defer && io.s_keep) ? io.s_keep : split + CC;   // [CC]
    float* dWsub = split + 2 * CC;         // [K][CC]
    const size_t pbytes = (size_t)((char*)ws + ws_bytes - (char*)partial);
    const float* mask = act == MVH_ACT_RELU ? out : nullptr;
    bool h1 = false, h2 = false;
    if (!mask && !io.dout)  // S_all is a plain x^T dpre reduction: stream it instead of running the LDS kernel with K = 1
      if (int rc = try_xty_small(st, x, dout, S, db, rows, Cin, Cout, partial, pbytes, &h1, io.x)) return rc;
    if (!h1)
      if (int rc = try_cheb_dw_lds(st, lap, x, dout, mask, S, db, B, N, Cin, Cout, 1, partial, pbytes, &h1, 0, nullptr, 0,
                                   false, out_bits, nullptr, io.x, io.dout)) return rc;
    if (h1 && defer && io.s_keep && !db) {
      // deferred form: the connected block's partial tiles stay in the caller's buffer and launch_dw_reduce_all
      // applies dW_k = dWsub_k + T_k(0) (S - dWsub_0) itself (two launches less on the weight-gradient lane)
      if (int rc = try_cheb_dw_lds(st, lap->sub, x, dout, mask, dW, nullptr, B, lap->n_active, Cin, Cout, K, defer_part,
                                   defer_bytes, &h2, N, nullptr, 0, false, out_bits, defer, io.x, io.dout)) return rc;
      if (h2) {
        defer->S = S;
        *deferred = true;
        dw_done = true;
        h1 = false;   // (nothing left for the immediate form below)
      }
    }
    if (h1 && !dw_done)
      if (int rc = try_cheb_dw_lds(st, lap->sub, x, dout, mask, dWsub, nullptr, B, lap->n_active, Cin, Cout, K, partial,
                                   pbytes, &h2, N, nullptr, 0, false, out_bits, nullptr, io.x, io.dout)) return rc;
    if (h1 && h2 && !dw_done) {
      hipLaunchKernelGGL(k_dw_combine, dim3(cdiv(K * CC, 256)), dim3(256), 0, st, dW, dWsub, S, K, CC);
      MVH_LAUNCH_CHECK();
      dw_done = true;
    }
  }
  if (split_ok && !dx_done) {  // dx = dpre W_eff^T everywhere, then the connected block overwrites its rows
    const float* weff = weff_pre;
    if (!weff) {
      hipLaunchKernelGGL(k_weff, dim3(cdiv(CC, 256)), dim3(256), 0, st, W, split, K, CC);
      MVH_LAUNCH_CHECK();
      weff = split;
    }
    MVH_REQUIRE(!io.dout, "cheb_conv_bwd: the split path reads an fp32 output gradient");
    // (dx_lazy: the rows off the connected block are rebuilt by the consumer from dout and W_eff, ConvIO::src3_*)
    if (!io.dx_lazy)
      if (int rc = launch_gstack(st, dout, out, weff, dx, dx, rows, Cin, Cout, 1, act, io.dx)) return rc;
    bool handled = false;
    LdsConvOpts so;
    so.prepacked = prepacked_bwd; so.in_bs = N; so.out_bs = N; so.out_bf16 = io.dx;
    if (int rc = try_cheb_lds(st, lap_t->sub, dout, act == MVH_ACT_RELU ? out : nullptr, W, nullptr, dx, B,
                              lap_t->n_active, Cin, Cout, K, act, true, wpack, &handled, so)) return rc;
    dx_done = handled;
    MVH_REQUIRE(handled || !io.dx_lazy, "cheb_conv_bwd: dx_lazy without the LDS kernel for the connected block");
  }
  MVH_REQUIRE(!io.dx_lazy || dx_done || !dx, "cheb_conv_bwd: dx_lazy on a layer that is not on the split path");
  if (!dw_done && !tx_saved) {  // fused dW/db: recurrence in LDS, contraction over vertices on the matrix pipe
    const size_t pbytes = defer ? defer_bytes : (size_t)((char*)ws + ws_bytes - (char*)partial);
    if (int rc = try_cheb_dw_lds(st, lap, x, dout, act == MVH_ACT_RELU ? out : nullptr, dW, db, B, N, Cin, Cout, K,
                                 defer ? defer_part : partial, pbytes, &dw_done, 0, nullptr, 0, false, out_bits, defer,
                                 io.x, io.dout, &io))
      return rc;
    if (defer && dw_done) *deferred = true;
  }
  MVH_REQUIRE(!io.src3_g || dw_done, "cheb_conv_bwd: lazy output-gradient rows (src3) need the LDS-resident dW kernel");
  // strided x: the dry run above asked without the mask operands; if the real call declined after all (e.g. a mask
  // pointer off its 16-byte alignment) nothing below can read x through its row map -- the caller makes the copy
  if (io.x_map && dW && !dw_done)
    return fail(MVH_ERR_UNSUPPORTED, "cheb_conv_bwd: the LDS-resident dW kernel declined a strided x (N=%d %d->%d K=%d)",
                N, Cin, Cout, K);
  if (!dw_done && bf)
    return fail(MVH_ERR_UNSUPPORTED, "cheb_conv_bwd: bf16 storage exists on the LDS-resident dW kernels only (N=%d %d->%d K=%d)",
                N, Cin, Cout, K);
  if (!dw_done) {
    const float* tx = tx_saved;
    const int pmN = tx_pair_major(lap, x, tx ? tx : tx_ws, B, N, Cin, Cout, K);   // (the forward's rule: same answer)
    if (!tx && K > 1) {
      if (int rc = tx_forward(st, lap, x, tx_ws, plane, B, Cin, K, pmN)) return rc;
      tx = tx_ws;
    }
    if (int rc = launch_dw(st, x, tx, dout, out, partial, dW, db, rows, Cin, Cout, K, act, pmN)) return rc;
  }
  auto pool_dx = [&]() -> int {  // fallback for dx_pool_t: dx was materialised, pool it with one more launch
    if (dx_pool_t && dx_pooled && dx)
      return launch_spmm(st, dx_pool_t, dx, dx_pooled, nullptr, nullptr, 1.f, 0.f, B, Cin, true);
    return MVH_OK;
  };
  if (dx_done) return pool_dx();
  {  // fused dX: the same LDS-resident Clenshaw kernel with W^T and the masked dout as input
    bool handled = false;
    LdsConvOpts bo;
    bo.prepacked = prepacked_bwd;
    bo.mask_bits = out_bits;
    bo.in_bf16 = io.dout; bo.prepacked_h = io.wh;
    bo.src3_g = io.src3_g; bo.src3_w = io.src3_w; bo.src3_n = io.src3_n; bo.src3_c = io.src3_c;
    if (dx_pool_t && dx_pooled) {  // pooled rows straight from the kernel (dx itself is not materialised)
      bo.out_pool_t = dx_pool_t;
      bo.out_bf16 = io.dx_pooled;
      if (int rc = try_cheb_lds(st, lap_t, dout, act == MVH_ACT_RELU ? out : nullptr, W, nullptr, dx_pooled, B, N, Cin,
                                Cout, K, act, true, wpack, &handled, bo)) return rc;
      if (handled) return MVH_OK;
      bo.out_pool_t = nullptr;
      if (bf) return fail(MVH_ERR_UNSUPPORTED, "cheb_conv_bwd: bf16 storage pools dx inside the dX kernel only");
    }
    bo.out_bf16 = io.dx;
    if (int rc = try_cheb_lds(st, lap_t, dout, act == MVH_ACT_RELU ? out : nullptr, W, nullptr, dx, B, N, Cin,
                              Cout, K, act, true, wpack, &handled, bo)) return rc;
    if (handled) {
      if (dx_pool_t && dx_pooled) return launch_spmm(st, dx_pool_t, dx, dx_pooled, nullptr, nullptr, 1.f, 0.f, B, Cin, true);
      return MVH_OK;
    }
  }
  MVH_REQUIRE(!io.src3_g, "cheb_conv_bwd: lazy output-gradient rows (src3) need the LDS-resident dX kernel");
  if (bf)
    return fail(MVH_ERR_UNSUPPORTED, "cheb_conv_bwd: bf16 storage exists on the LDS-resident dX kernels only (N=%d %d->%d K=%d)",
                N, Cin, Cout, K);
  // 16 -> 16 channels on a level of 5120 .. 20480 vertices (BASELINE configs[3]'s level 0): the INPUT-side form
  //   dx = sum_k T_k(L^T)(dpre) W_k^T  --  cheb_big.hip's T-stack kernel on the masked dout (pair-major planes, plane 0 =
  // dpre itself) and the forward's contraction kernel with the transposed weights.  It moves the same bytes as the
  // output-side form below (G stack out, Clenshaw sum in) but both of its kernels run without scratch: 254 + 228 us
  // against 290 + 400 us measured in isolation.
  if (K > 1 && Cin == 16 && Cout == 16 && dx && cheb_big_eligible(lap_t, B, N, Cout, K) &&
      (((uintptr_t)dout | (uintptr_t)out | (uintptr_t)G | (uintptr_t)dx) & 15) == 0 && !dbg().no_dx_tstack &&
      (size_t)K * Cin * Cout * sizeof(float) <= kSplitScratchBytes && (act != MVH_ACT_RELU || out)) {
    float* Wt = split;   // [K][Cout][Cin] = W_k^T (the split-path scratch is free on this path)
    hipLaunchKernelGGL(k_w_transpose, dim3(cdiv(K * Cin * Cout, 256)), dim3(256), 0, st, W, Wt, K, Cin, Cout);
    MVH_LAUNCH_CHECK();
    bool big = false;
    if (int rc = try_cheb_big_tx(st, lap_t, dout, G, B, N, Cout, K, true, &big, act == MVH_ACT_RELU ? out : nullptr, true))
      return rc;
    if (big) {
      if (int rc = launch_contract(st, nullptr, G, Wt, nullptr, dx, rows, Cout, Cin, K, MVH_ACT_NONE, false, -N)) return rc;
      return pool_dx();
    }
  }
  // dx = sum_k T_k(L^T) G_k via Clenshaw: b_k = G_k + 2 L^T b_{k+1} - b_{k+2}; dx = G_0 + L^T b_1 - b_2
  float* g0 = (K == 1) ? dx : G;
  // (the stack is pair-major where cheb_big.hip consumes it and the matrix-pipe kernel produces it)
  const int g_pm = (K > 1 && cheb_big_eligible(lap_t, B, N, Cin, K) && gstack_is_mfma(dout, out, W, G, g0, rows, Cin, Cout, K)) ? N : 0;
  if (int rc = launch_gstack(st, dout, out, W, G, g0, rows, Cin, Cout, K, act, false, g_pm)) return rc;
  if (K == 1) return pool_dx();
  {  // 5120 .. 20480 vertices: the Clenshaw sum over the stack in one launch (cheb_big.hip)
    bool big = false;
    if (int rc = try_cheb_big_clenshaw(st, lap_t, G, dx, B, N, Cin, K, g_pm != 0, &big)) return rc;
    if (big) return pool_dx();
    if (g_pm) return fail(MVH_ERR_INVALID, "cheb_conv_bwd: pair-major G stack without the cheb_big kernel");
  }
  for (int k = K - 2; k >= 1; --k) {
    float* bk = G + (long long)k * plane;
    const float* bk2 = (k + 2 < K) ? G + (long long)(k + 2) * plane : nullptr;
    if (int rc = launch_spmm(st, lap_t, G + (long long)(k + 1) * plane, bk, bk, bk2, 2.f, -1.f, B, Cin, false))
      return rc;
  }
  if (int rc = launch_spmm(st, lap_t, G + plane, dx, G, (K > 2) ? G + 2 * plane : nullptr, 1.f, -1.f, B, Cin, false))
    return rc;
  if (dx_pool_t && dx_pooled) return launch_spmm(st, dx_pool_t, dx, dx_pooled, nullptr, nullptr, 1.f, 0.f, B, Cin, true);
  return MVH_OK;
}


// ------------------------------------------------------------------ strided x at the module boundary
// x is a [B, N, Cin] VIEW with element strides (x_mesh_stride, x_row_stride, 1) -- e.g. the transpose of an
// [N, B, Cin]-physical tensor, as the reference's own modules hand around (nn/conv.py:560, nn/pool.py:18).  Both strides
// must be multiples of Cin (so that a row is `Cin` consecutive elements at a row index of the flattened buffer) and the
// rows 16-byte aligned.  The LDS-resident kernels read the rows in place through their row map (built here, in the
// tail of `ws`: N int32); shapes those kernels do not cover return MVH_ERR_UNSUPPORTED and the caller copies.
__global__ void __launch_bounds__(256) k_row_map(int32_t* __restrict__ map, int n, int step) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) map[i] = i * step;
}

static size_t strided_map_bytes(int N) { return align_up((size_t)N * sizeof(int32_t), 256); }

extern "C" size_t mvh_cheb_conv_strided_ws_bytes(int32_t B, int32_t N, int32_t Cin, int32_t Cout, int32_t K) {
  const size_t f = mvh_cheb_conv_ws_bytes(B, N, Cin, Cout, K), b = mvh_cheb_conv_bwd_ws_bytes(B, N, Cin, Cout, K);
  return (f > b ? f : b) + 256 + strided_map_bytes(N);
}

static int strided_setup(hipStream_t st, const float* x, int64_t ms, int64_t rs, int N, int Cin, void* ws, size_t ws_bytes,
                         ConvIO& io, size_t* inner_bytes) {
  MVH_REQUIRE(x && Cin > 0 && N > 0, "cheb_conv_strided: null tensor or bad sizes");
  MVH_REQUIRE(ms >= 0 && rs > 0 && ms % Cin == 0 && rs % Cin == 0, "cheb_conv_strided: strides (%lld, %lld) must be multiples of Cin = %d",
              (long long)ms, (long long)rs, Cin);
  MVH_REQUIRE(((uintptr_t)x % 16) == 0 && (Cin % 4 != 0 || ((ms | rs) % 4) == 0), "cheb_conv_strided: rows must be 16-byte aligned");
  MVH_REQUIRE(rs / Cin * (long long)(N - 1) < (1ll << 31) && ms / Cin < (1ll << 31), "cheb_conv_strided: strides too large");
  const size_t mb = strided_map_bytes(N);
  MVH_REQUIRE(ws && ws_bytes > mb + 256, "cheb_conv_strided: workspace too small (mvh_cheb_conv_strided_ws_bytes)");
  *inner_bytes = (ws_bytes - mb) / 256 * 256;
  int32_t* map = reinterpret_cast<int32_t*>((char*)ws + *inner_bytes);
  hipLaunchKernelGGL(k_row_map, dim3(cdiv(N, 256)), dim3(256), 0, st, map, N, (int)(rs / Cin));
  MVH_LAUNCH_CHECK();
  io.x_map = map;
  io.x_bs = (int)(ms / Cin);
  MVH_REQUIRE(io.x_bs > 0, "cheb_conv_strided: mesh stride 0 (a broadcast batch) is not supported");
  return MVH_OK;
}

extern "C" int mvh_cheb_conv_fwd_strided(mvh_stream_t stream, const mvh_csr_t* lap, const float* x, int64_t x_mesh_stride,
                                         int64_t x_row_stride, const float* W, const float* bias, float* out,
                                         uint8_t* relu_signs, int32_t B, int32_t N, int32_t Cin, int32_t Cout, int32_t K,
                                         int32_t act, void* ws, size_t ws_bytes) {
  MVH_REQUIRE(!relu_signs || (act == MVH_ACT_RELU && Cout % 4 == 0), "cheb_conv_fwd_strided: relu_signs need act = RELU and Cout %% 4 == 0");
  ConvIO io;
  size_t inner = 0;
  if (int rc = strided_setup((hipStream_t)stream, x, x_mesh_stride, x_row_stride, N, Cin, ws, ws_bytes, io, &inner)) return rc;
  return cheb_conv_fwd_impl((hipStream_t)stream, lap, x, W, bias, out, nullptr, B, N, Cin, Cout, K, act, ws, inner, nullptr,
                            nullptr, nullptr, relu_signs, nullptr, io);
}

extern "C" int mvh_cheb_conv_bwd_strided(mvh_stream_t stream, const mvh_csr_t* lap, const mvh_csr_t* lap_t, const float* x,
                                         int64_t x_mesh_stride, int64_t x_row_stride, const float* W, const float* out,
                                         const uint8_t* relu_signs, const float* dout, float* dx, float* dW, float* db,
                                         int32_t B, int32_t N, int32_t Cin, int32_t Cout, int32_t K, int32_t act, void* ws,
                                         size_t ws_bytes) {
  MVH_REQUIRE(!relu_signs || (act == MVH_ACT_RELU && Cout % 4 == 0), "cheb_conv_bwd_strided: relu_signs need act = RELU and Cout %% 4 == 0");
  ConvIO io;
  size_t inner = 0;
  if (int rc = strided_setup((hipStream_t)stream, x, x_mesh_stride, x_row_stride, N, Cin, ws, ws_bytes, io, &inner)) return rc;
  return cheb_conv_bwd_impl((hipStream_t)stream, lap, lap_t, x, W, out, dout, nullptr, dx, dW, db, B, N, Cin, Cout, K, act,
                            ws, inner, nullptr, nullptr, nullptr, relu_signs, nullptr, nullptr, nullptr, 0, nullptr, nullptr,
                            nullptr, io);
}
