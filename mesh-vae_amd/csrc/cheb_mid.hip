// Small-level sub-networks of the decoder / encoder as ONE launch per direction (cheb_VAE.py:281-286
// and :262-270): at the coarse levels (<= 512 vertices) a whole mesh with ALL its channels fits one CU's LDS, so a
// workgroup owns a mesh and walks   unpool -> ChebConv + ReLU -> unpool -> ChebConv + ReLU -> unpool   without leaving
// the CU -- three pool launches and two convolution launches of 7-14 us each (launch-latency bound: a 79-vertex
// convolution costs the same 8 us at K = 1 and K = 6) become one.
//
// Convolution inside a workgroup, Clenshaw form on the OUTPUT channels (the recurrence then runs on min(Cin, Cout)
// = Cout channels for the 32 -> 16 stage, and the matrix pipe reads x once instead of T_k per order):
//   y_k = x W_k                         v_mfma_f32_16x16x4_f32 (exact fp32): a wave owns up to two 16-vertex tiles, the
//                                       A operand (its rows of x) is loaded from LDS ONCE into registers, B = W_k comes
//                                       from global one order ahead, D = a 16 x 16 tile of y_k
//   b_k = y_k + 2 L b_{k+1} - b_{k+2}   two fp32 planes [N][Cout + 4] in LDS: the tile lanes store y_k - b_{k+2} over
//                                       b_{k+2}, barrier, then a thread per (vertex, 4 channels) adds 2 L b_{k+1} with
//                                       the CSR edges (col, val packed in 8 bytes) staged in LDS, four edges in flight
//   out = y_0 + L b_1 - b_2             + bias, ReLU, sign byte, all in the last gather's threads.
// The pooled rows are produced from the activated plane in the operator's CSR order with separately rounded products
// and sums (the arithmetic of k_spmm<.., EXACT>, nn/pool.py:17-20).
// Results differ from the per-layer kernels only by fp32 summation order.
#include <initializer_list>

#include "common.hpp"
#include "bf16.hpp"

namespace mvh {
unsigned long long* g_mid_tlog = nullptr;
#define MARK(i) do { if (a.tlog && blockIdx.x == 0 && threadIdx.x == 0) { a.tlog[i] = wall_clock64(); if (i == 0 || i == 6) a.tlog[56 + i] = clock64(); } } while (0)

constexpr int kMidThreads = 1024, kMidWaves = kMidThreads / 64, kMidTiles = 2;   // <= 2 x 16 x 16 = 512 vertices

typedef float v4f_m __attribute__((ext_vector_type(4)));

struct MidCsr {
  const int* rowptr;
  const int* col;
  const float* val;
  const uint32_t* rowinfo;   // (rowptr[r] << 8) | row length
  int n_rows, nnz;
};

struct MidDecArgs {
  const float* d2;             // [B][n4][CA] fp32 (dec_lin_2's output)
  MidCsr up3, lap3, up2, lap2, up1;
  const float *W0, *b0, *W1, *b1;   // dec stage 0: [K0][CA][CB]; stage 1: [K1][CB][CC]
  float *decU0, *decU1, *decU2;     // [B][n3][CA], [B][n2][CB], [B][n1][CC]  (fp32 or bf16)
  uint8_t *bits0, *bits1;           // [B][n3][CB/4], [B][n2][CC/4]
  int B, n4, n3, n2, n1, K0, K1, out_bf16;
  int r1_floats, r23_floats;        // LDS regions: R1 (x of stage 1), R2 and R3 (its two Clenshaw planes)
  int wl_floats, emax, nmax;
  unsigned long long* tlog;
  int exp;
};

__device__ __forceinline__ float bf16_round(float x) { return bf16_lo(bf16_pack2(x, 0.f)); }

// ---- register prefetch: N dwords per thread of a global array, issued early and parked in LDS once the target
//      region is free (the global latency, 1-2 us, then runs under the preceding phase)
template <int N>
struct PfBuf {
  int v[N];
};
template <int N>
__device__ __forceinline__ void pf_load(PfBuf<N>& b, const void* g, int count) {
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const int i = threadIdx.x + j * kMidThreads;
    b.v[j] = i < count ? reinterpret_cast<const int*>(g)[i] : 0;
  }
}
template <int N>
__device__ __forceinline__ void pf_store(const PfBuf<N>& b, void* lds, int count) {
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const int i = threadIdx.x + j * kMidThreads;
    if (i < count) reinterpret_cast<int*>(lds)[i] = b.v[j];
  }
}
template <int N>
__device__ __forceinline__ void pf_store_edges(const PfBuf<N>& c, const PfBuf<N>& v, int2* edges, int count) {
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const int i = threadIdx.x + j * kMidThreads;
    if (i < count) edges[i] = make_int2(c.v[j], v.v[j]);
  }
}
// weights [rows][CO] -> LDS [rows][CO + 4] (the four k-slices a wave's lanes read then sit in different banks)
template <int N, int CO>
__device__ __forceinline__ void pf_store_w(const PfBuf<N>& b, float* wl, int count) {
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const int i = threadIdx.x + j * kMidThreads;
    if (i < count) wl[(i / CO) * (CO + 4) + (i % CO)] = __int_as_float(b.v[j]);
  }
}

// pooling operator (<= 3 entries per row, checked on the host): info[r] = (first entry << 8) | row length
struct PoolOp {
  const uint32_t* info;
  const int* col;
  const float* val;
  int n_rows;
};
template <int PB>
struct PoolTaps {
  uint32_t info[PB];
  int c[PB][3];
  float w[PB][3];
};
// the taps of this thread's PB row items, starting at item `i0` (item = row * C/4 + channel quad)
template <int C, int PB>
__device__ __forceinline__ void pool_fetch(PoolTaps<PB>& t, const PoolOp& P, int i0) {
  constexpr int Q = C / 4;
  const int total = P.n_rows * Q;
#pragma unroll
  for (int j = 0; j < PB; ++j) {
    const int i = i0 + j * kMidThreads;
    t.info[j] = i < total ? P.info[i / Q] : 0u;
  }
#pragma unroll
  for (int j = 0; j < PB; ++j)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const bool has = k < (int)(t.info[j] & 255u);
      const int e = (int)(t.info[j] >> 8) + k;
      t.c[j][k] = has ? P.col[e] : 0;
      t.w[j][k] = has ? P.val[e] : 0.f;
    }
}
// rows of `dst` (LDS plane, stride S) and / or `gdst` (global [n_out][C]) = P * src (LDS plane, stride SS), in the
// operator's entry order with separately rounded products and sums
template <int C, int PB>
__device__ __forceinline__ void pool_apply(const PoolTaps<PB>& t, int n_rows, int i0, const float* src, int SS, float* dst,
                                           int S, float* gdst, long long gbase, bool g_bf16, bool round_lds) {
  constexpr int Q = C / 4;
  const int total = n_rows * Q;
#pragma unroll
  for (int j = 0; j < PB; ++j) {
    const int i = i0 + j * kMidThreads;
    if (i < total) {
      const int r = i / Q, q = i - r * Q, len = (int)(t.info[j] & 255u);
      float4 x[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) x[k] = *reinterpret_cast<const float4*>(src + t.c[j][k] * SS + 4 * q);
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if (k < len) {
          acc.x = __fadd_rn(acc.x, __fmul_rn(t.w[j][k], x[k].x));
          acc.y = __fadd_rn(acc.y, __fmul_rn(t.w[j][k], x[k].y));
          acc.z = __fadd_rn(acc.z, __fmul_rn(t.w[j][k], x[k].z));
          acc.w = __fadd_rn(acc.w, __fmul_rn(t.w[j][k], x[k].w));
        }
      if (gdst) store4_any(gdst, gbase + (long long)r * C + 4 * q, g_bf16, acc.x, acc.y, acc.z, acc.w);
      if (dst) {
        if (round_lds) acc = make_float4(bf16_round(acc.x), bf16_round(acc.y), bf16_round(acc.z), bf16_round(acc.w));
        *reinterpret_cast<float4*>(dst + r * S + 4 * q) = acc;
      }
    }
  }
}
template <int C, int PB>
__device__ __forceinline__ void mid_pool(const PoolOp& P, const float* src, int SS, float* dst, int S, float* gdst,
                                         long long gbase, bool g_bf16, bool round_lds) {
  for (int i0 = threadIdx.x; i0 < P.n_rows * (C / 4); i0 += PB * kMidThreads) {
    PoolTaps<PB> t;
    pool_fetch<C, PB>(t, P, i0);
    pool_apply<C, PB>(t, P.n_rows, i0, src, SS, dst, S, gdst, gbase, g_bf16, round_lds);
  }
}

// ELL form of a staged Laplacian for the gather: 8 slots (col, val) per vertex, slot 0's col carries the row length
// in its upper half; rows longer than 8 continue in the CSR copy (a decimated level has 2-6 % of them)
__device__ __forceinline__ void mid_build_ell(int n, const int* rp, const int2* edges, int2* ell) {
  for (int i = threadIdx.x; i < n * 8; i += kMidThreads) {
    const int v = i >> 3, j = i & 7;
    const int e0 = rp[v], len = rp[v + 1] - e0;
    int2 s = j < len ? edges[e0 + j] : make_int2(0, 0);
    if (j == 0) s.x |= len << 16;
    ell[i] = s;
  }
}

// q[v] += alpha * sum_e w_e p[col_e]   per (vertex, 4 channels);  FINAL: + bias, ReLU, sign byte
template <int CO, bool FINAL>
__device__ __forceinline__ void mid_gather(const float* p, float* q, int n, float alpha, bool has_edges, const int* rp,
                                           const int2* edges, const int4* ell, const float* __restrict__ bias,
                                           uint8_t* __restrict__ bits, long long bit_base, int exp = 0) {
  constexpr int S = CO + 4, Q = CO / 4;
  for (int i = threadIdx.x; i < n * Q; i += kMidThreads) {
    const int v = i / Q, qd = i - v * Q;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 o = *reinterpret_cast<const float4*>(q + v * S + 4 * qd);
    if (has_edges) {
      const int4 s0 = ell[v * 4], s1 = ell[v * 4 + 1];
      const int len = s0.x >> 16;
      {
        const float4 x0 = *reinterpret_cast<const float4*>(p + (s0.x & 0xFFFF) * S + 4 * qd);
        const float4 x1 = *reinterpret_cast<const float4*>(p + s0.z * S + 4 * qd);
        const float4 x2 = *reinterpret_cast<const float4*>(p + s1.x * S + 4 * qd);
        const float4 x3 = *reinterpret_cast<const float4*>(p + s1.z * S + 4 * qd);
        const float w0 = __int_as_float(s0.y), w1 = __int_as_float(s0.w), w2 = __int_as_float(s1.y), w3 = __int_as_float(s1.w);
        g.x = fmaf(w3, x3.x, fmaf(w2, x2.x, fmaf(w1, x1.x, w0 * x0.x)));
        g.y = fmaf(w3, x3.y, fmaf(w2, x2.y, fmaf(w1, x1.y, w0 * x0.y)));
        g.z = fmaf(w3, x3.z, fmaf(w2, x2.z, fmaf(w1, x1.z, w0 * x0.z)));
        g.w = fmaf(w3, x3.w, fmaf(w2, x2.w, fmaf(w1, x1.w, w0 * x0.w)));
      }
      if (len > 4 && !(exp & 4)) {
        const int4 s2 = ell[v * 4 + 2], s3 = ell[v * 4 + 3];
        const float4 x0 = *reinterpret_cast<const float4*>(p + s2.x * S + 4 * qd);
        const float4 x1 = *reinterpret_cast<const float4*>(p + s2.z * S + 4 * qd);
        const float4 x2 = *reinterpret_cast<const float4*>(p + s3.x * S + 4 * qd);
        const float4 x3 = *reinterpret_cast<const float4*>(p + s3.z * S + 4 * qd);
        const float w0 = __int_as_float(s2.y), w1 = __int_as_float(s2.w), w2 = __int_as_float(s3.y), w3 = __int_as_float(s3.w);
        g.x = fmaf(w3, x3.x, fmaf(w2, x2.x, fmaf(w1, x1.x, fmaf(w0, x0.x, g.x))));
        g.y = fmaf(w3, x3.y, fmaf(w2, x2.y, fmaf(w1, x1.y, fmaf(w0, x0.y, g.y))));
        g.z = fmaf(w3, x3.z, fmaf(w2, x2.z, fmaf(w1, x1.z, fmaf(w0, x0.z, g.z))));
        g.w = fmaf(w3, x3.w, fmaf(w2, x2.w, fmaf(w1, x1.w, fmaf(w0, x0.w, g.w))));
        if (len > 8) {
          const int e1 = rp[v + 1];
          for (int e = rp[v] + 8; e < e1; ++e) {
            const int2 ev = edges[e];
            const float4 x = *reinterpret_cast<const float4*>(p + ev.x * S + 4 * qd);
            const float w = __int_as_float(ev.y);
            g.x = fmaf(w, x.x, g.x);
            g.y = fmaf(w, x.y, g.y);
            g.z = fmaf(w, x.z, g.z);
            g.w = fmaf(w, x.w, g.w);
          }
        }
      }
    }
    o = make_float4(fmaf(alpha, g.x, o.x), fmaf(alpha, g.y, o.y), fmaf(alpha, g.z, o.z), fmaf(alpha, g.w, o.w));
    if (FINAL) {
      const float4 bb = *reinterpret_cast<const float4*>(bias + 4 * qd);
      o = make_float4(fmaxf(o.x + bb.x, 0.f), fmaxf(o.y + bb.y, 0.f), fmaxf(o.z + bb.z, 0.f), fmaxf(o.w + bb.w, 0.f));
      bits[bit_base + (long long)v * Q + qd] =
          (uint8_t)((o.x > 0.f ? 1 : 0) | (o.y > 0.f ? 2 : 0) | (o.z > 0.f ? 4 : 0) | (o.w > 0.f ? 8 : 0));
    }
    *reinterpret_cast<float4*>(q + v * S + 4 * qd) = o;
  }
}

// ChebConv + bias + ReLU of one mesh, all channels: x in the plane X [n][CI + 4], pa / pb two planes [n][CO + 4],
// wl = the weights in LDS ([K * CI][CO + 4]).  `after_x` runs once every wave holds its rows of x in registers (X is
// free from then on).  Returns the plane that holds the activated output; ends with a barrier.
// Matrix-pipe work units: (16-vertex tile, 16-channel block) u = wave + 16 j, so that the four SIMDs get equal shares.
template <int CI, int CO, class F>
__device__ __forceinline__ float* mid_conv(const float* X, float* pa, float* pb, int n, int K, const int* rp,
                                           const int2* edges, const int4* ell, const float* wl,
                                           const float* __restrict__ bias, uint8_t* __restrict__ bits, long long bit_base,
                                           F after_x, unsigned long long* tl = nullptr, int exp = 0) {
  constexpr int SX = CI + 4, S = CO + 4, H = CI / 16, NB = CO / 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int units = ((n + 15) >> 4) * NB;
  // A operand: lane (li, lk) keeps x[v0 + li][4 (lk + 4 h) .. + 3]; MFMA (h, m) contracts the channels 4 (lk + 4 h) + m
  float4 a[kMidTiles][H];
#pragma unroll
  for (int j = 0; j < kMidTiles; ++j) {
    const int u = wave + kMidWaves * j, v0 = (u / NB) * 16;
    const int vr = min(v0 + li, n - 1);
#pragma unroll
    for (int h = 0; h < H; ++h)
      a[j][h] = (u < units) ? *reinterpret_cast<const float4*>(X + vr * SX + 4 * (lk + 4 * h)) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  after_x();
  float *p = pa, *q = pb;   // b_{k+1}, b_{k+2} (the latter becomes b_k)
  for (int k = K - 1; k >= 0; --k) {
    v4f_m acc[kMidTiles];
#pragma unroll
    for (int j = 0; j < kMidTiles; ++j) acc[j] = (v4f_m){0.f, 0.f, 0.f, 0.f};
    if (wave < units && !(exp & 1)) {  // wave-uniform
#pragma unroll
      for (int h = 0; h < H; ++h) {
#pragma unroll
        for (int j = 0; j < kMidTiles; ++j) {
          const int u = wave + kMidWaves * j;
          if (u < units) {
            const float* wrow = wl + (k * CI + 4 * (lk + 4 * h)) * S + (u % NB) * 16 + li;
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][h].x, wrow[0], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][h].y, wrow[S], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][h].z, wrow[2 * S], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][h].w, wrow[3 * S], acc[j], 0, 0, 0);
          }
        }
      }
    }
    if (tl && blockIdx.x == 0 && threadIdx.x == 0) tl[3 * k] = wall_clock64();
    const bool has_sub = k + 2 <= K - 1, has_g = k + 1 <= K - 1;
    // the tile lanes: q = y_k - b_{k+2}   (lane (li, lk) holds y_k[v0 + 4 lk + r][16 nb + li]); all reads, then all writes
    float old[kMidTiles][4];
#pragma unroll
    for (int j = 0; j < kMidTiles; ++j) {
      const int u = wave + kMidWaves * j, v0 = (u / NB) * 16 + 4 * lk, c = (u % NB) * 16 + li;
#pragma unroll
      for (int r = 0; r < 4; ++r) old[j][r] = (has_sub && u < units && v0 + r < n) ? q[(v0 + r) * S + c] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < kMidTiles; ++j) {
      const int u = wave + kMidWaves * j, v0 = (u / NB) * 16 + 4 * lk, c = (u % NB) * 16 + li;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (u < units && v0 + r < n) q[(v0 + r) * S + c] = acc[j][r] - old[j][r];
    }
    __syncthreads();
    if (tl && blockIdx.x == 0 && threadIdx.x == 0) tl[3 * k + 1] = wall_clock64();
    if (k == 0) {
      mid_gather<CO, true>(p, q, n, 1.f, has_g, rp, edges, ell, bias, bits, bit_base);
      __syncthreads();
    } else if (has_g) {
      mid_gather<CO, false>(p, q, n, 2.f, !(exp & 2), rp, edges, ell, nullptr, nullptr, 0, exp);
      __syncthreads();
    }
    if (tl && blockIdx.x == 0 && threadIdx.x == 0) tl[3 * k + 2] = wall_clock64();
    float* t = p;
    p = q;
    q = t;
  }
  return p;
}

// ---- decoder head, forward:  d2 -> U3 -> conv(CA -> CB) + ReLU -> U2 -> conv(CB -> CC) + ReLU -> U1
// LDS: R1 [n2][CB + 4] (d2, then x of stage 1, then U1's entries) | R2, R3 [n2][CC + 4] each (stage 0's three planes,
// then stage 1's two) | WL (W0, then W1 + U2's entries) | edges + rowptr of the current Laplacian.
// Every global read is issued at the start of the kernel or one phase ahead (register prefetch).
template <int CA, int CB, int CC>
__global__ void __launch_bounds__(kMidThreads) k_mid_dec_fwd(MidDecArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  float* r1 = reinterpret_cast<float*>(smem);
  float* r2 = r1 + a.r1_floats;
  float* r3 = r2 + a.r23_floats;
  float* wl = r3 + a.r23_floats;
  int2* edges = reinterpret_cast<int2*>(wl + a.wl_floats);
  int* rp = reinterpret_cast<int*>(edges + a.emax);
  int2* ell = reinterpret_cast<int2*>(rp + ((a.nmax + 7) & ~3));
  const int b = blockIdx.x;
  const bool hb = a.out_bf16 != 0;
  constexpr int NW1 = (6 * CB * CC + kMidThreads - 1) / kMidThreads;   // K1 <= 6 (host check)

  MARK(0);
  // now: d2 -> R1 (stride CA + 4), Laplacian of level 3, W0, U3's taps
  for (int i = threadIdx.x; i < a.n4 * (CA / 4); i += kMidThreads) {
    const int r = i / (CA / 4), q = i - r * (CA / 4);
    *reinterpret_cast<float4*>(r1 + r * (CA + 4) + 4 * q) =
        *reinterpret_cast<const float4*>(a.d2 + ((long long)b * a.n4 + r) * CA + 4 * q);
  }
  for (int i = threadIdx.x; i <= a.n3; i += kMidThreads) rp[i] = a.lap3.rowptr[i];
  for (int i = threadIdx.x; i < a.lap3.nnz; i += kMidThreads)
    edges[i] = make_int2(a.lap3.col[i], __float_as_int(a.lap3.val[i]));
  for (int i = threadIdx.x; i < a.K0 * CA * CB; i += kMidThreads) wl[(i / CB) * (CB + 4) + (i % CB)] = a.W0[i];
  const PoolOp up3{a.up3.rowinfo, a.up3.col, a.up3.val, a.n3};
  PoolTaps<1> t3;
  pool_fetch<CA, 1>(t3, up3, threadIdx.x);            // (n3 * CA / 4 <= 1024 items, host check)
  // one phase ahead: level 2's Laplacian, U2's entries, W1
  PfBuf<1> p_rp, p_ui;
  PfBuf<4> p_ec, p_ev;
  PfBuf<2> p_uc, p_uv;
  PfBuf<NW1> p_w1;
  pf_load(p_rp, a.lap2.rowptr, a.n2 + 1);
  pf_load(p_ec, a.lap2.col, a.lap2.nnz);
  pf_load(p_ev, a.lap2.val, a.lap2.nnz);
  pf_load(p_ui, a.up2.rowinfo, a.n2);
  pf_load(p_uc, a.up2.col, a.up2.nnz);
  pf_load(p_uv, a.up2.val, a.up2.nnz);
  pf_load(p_w1, a.W1, a.K1 * CB * CC);
  __syncthreads();
  MARK(1);
  // stage 0 lives in R2 + R3: x0 [n3][CA + 4], two planes [n3][CB + 4]
  float* x0 = r2;
  float* pa0 = x0 + a.n3 * (CA + 4);
  float* pb0 = pa0 + a.n3 * (CB + 4);
  pool_apply<CA, 1>(t3, a.n3, threadIdx.x, r1, CA + 4, x0, CA + 4, a.decU0, (long long)b * a.n3 * CA, hb, hb);
  mid_build_ell(a.n3, rp, edges, ell);
  __syncthreads();
  MARK(2);
  const float* out0 = mid_conv<CA, CB>(x0, pa0, pb0, a.n3, a.K0, rp, edges, reinterpret_cast<const int4*>(ell), wl, a.b0, a.bits0,
                                       (long long)b * a.n3 * (CB / 4), [] {}, a.tlog ? a.tlog + 8 : nullptr, a.exp);
  MARK(3);
  // (W0 and level 3's edges are dead after the convolution's final barrier)
  float* w1 = wl;
  uint32_t* u2i = reinterpret_cast<uint32_t*>(wl + a.K1 * CB * (CC + 4));
  int* u2c = reinterpret_cast<int*>(u2i + a.n2);
  float* u2v = reinterpret_cast<float*>(u2c + a.up2.nnz);
  pf_store(p_rp, rp, a.n2 + 1);
  pf_store_edges(p_ec, p_ev, edges, a.lap2.nnz);
  pf_store(p_ui, u2i, a.n2);
  pf_store(p_uc, u2c, a.up2.nnz);
  pf_store(p_uv, u2v, a.up2.nnz);
  pf_store_w<NW1, CC>(p_w1, w1, a.K1 * CB * CC);
  __syncthreads();
  mid_build_ell(a.n2, rp, edges, ell);
  // U2: activated stage-0 rows -> R1 = x of stage 1, and decU1
  mid_pool<CB, 3>(PoolOp{u2i, u2c, u2v, a.n2}, out0, CB + 4, r1, CB + 4, a.decU1, (long long)b * a.n2 * CB, hb, hb);
  // one phase ahead: U1's entries (parked in R1 as soon as stage 1 holds its x in registers)
  PfBuf<2> q_ui;
  PfBuf<4> q_uc, q_uv;
  pf_load(q_ui, a.up1.rowinfo, a.n1);
  pf_load(q_uc, a.up1.col, a.up1.nnz);
  pf_load(q_uv, a.up1.val, a.up1.nnz);
  __syncthreads();
  MARK(4);
  uint32_t* u1i = reinterpret_cast<uint32_t*>(r1);
  int* u1c = reinterpret_cast<int*>(u1i + a.n1);
  float* u1v = reinterpret_cast<float*>(u1c + a.up1.nnz);
  const float* out1 = mid_conv<CB, CC>(r1, r2, r3, a.n2, a.K1, rp, edges, reinterpret_cast<const int4*>(ell), w1, a.b1, a.bits1, (long long)b * a.n2 * (CC / 4), [&] {
    pf_store(q_ui, u1i, a.n1);
    pf_store(q_uc, u1c, a.up1.nnz);
    pf_store(q_uv, u1v, a.up1.nnz);
  }, a.tlog ? a.tlog + 32 : nullptr, a.exp);
  MARK(5);
  // U1: -> decU2 (the next level's convolution input) straight to global
  mid_pool<CC, 5>(PoolOp{u1i, u1c, u1v, a.n1}, out1, CC + 4, nullptr, 0, a.decU2, (long long)b * a.n1 * CC, hb, false);
  MARK(6);
}

static MidCsr mid_csr(const mvh_csr_t* c) { return MidCsr{c->rowptr, c->col, c->val, c->rowinfo, c->n_rows, c->nnz}; }

// The three pools + two convolutions of the decoder's coarse end (d2 -> ... -> decU[2]); *handled == false:
// not eligible (sizes / channel counts / operator form), the caller keeps the per-layer launches.
int try_mid_dec_fwd(hipStream_t st, const mvh_csr_t* up3, const mvh_csr_t* lap3, const mvh_csr_t* up2,
                    const mvh_csr_t* lap2, const mvh_csr_t* up1, const float* d2, const float* W0, const float* b0,
                    const float* W1, const float* b1, float* decU0, float* decU1, float* decU2, uint8_t* bits0,
                    uint8_t* bits1, int B, int CA, int CB, int CC, int K0, int K1, bool out_bf16, bool* handled) {
  *handled = false;
  if (dbg().force_generic || dbg().no_mid) return MVH_OK;
  if (!bits0 || !bits1 || !b0 || !b1) return MVH_OK;
  const int n4 = up3->n_cols, n3 = up3->n_rows, n2 = up2->n_rows, n1 = up1->n_rows;
  if (lap3->n_rows != n3 || lap2->n_rows != n2 || up2->n_cols != n3 || up1->n_cols != n2) return MVH_OK;
  if (n2 > 16 * kMidWaves * kMidTiles || n3 > n2 || n4 > n3 || B < 1 || K0 < 1 || K1 < 1) return MVH_OK;
  if (!(CA == 32 && CB == 32 && CC == 16)) return MVH_OK;
  if (K0 > 6 || K1 > 6) return MVH_OK;
  for (const mvh_csr_t* u : {up3, up2, up1})
    if (!u->rowinfo || u->max_row_nnz > 3) return MVH_OK;
  // what the kernel's fixed-size register prefetches and one-pass loops cover
  if (n3 * (CA / 4) > kMidThreads || n2 + 1 > kMidThreads || lap2->nnz > 4 * kMidThreads || up2->nnz > 2 * kMidThreads ||
      n1 > 2 * kMidThreads || up1->nnz > 4 * kMidThreads)
    return MVH_OK;
  MidDecArgs a;
  a.d2 = d2;
  a.up3 = mid_csr(up3); a.lap3 = mid_csr(lap3); a.up2 = mid_csr(up2); a.lap2 = mid_csr(lap2); a.up1 = mid_csr(up1);
  a.W0 = W0; a.b0 = b0; a.W1 = W1; a.b1 = b1;
  a.decU0 = decU0; a.decU1 = decU1; a.decU2 = decU2; a.bits0 = bits0; a.bits1 = bits1;
  a.B = B; a.n4 = n4; a.n3 = n3; a.n2 = n2; a.n1 = n1; a.K0 = K0; a.K1 = K1; a.out_bf16 = out_bf16 ? 1 : 0;
  a.r1_floats = n2 * (CB + 4);
  a.r23_floats = n2 * (CC + 4);
  const int w0f = K0 * CA * (CB + 4), w1f = K1 * CB * (CC + 4) + n2 + 2 * up2->nnz;
  a.wl_floats = ((w0f > w1f ? w0f : w1f) + 3) & ~3;
  a.emax = ((lap3->nnz > lap2->nnz ? lap3->nnz : lap2->nnz) + 1) & ~1;
  a.nmax = n2;
  a.tlog = g_mid_tlog;
  a.exp = dbg().l0_wide;
  if (n4 * (CA + 4) > a.r1_floats || n3 * (CA + 4 + 2 * (CB + 4)) > 2 * a.r23_floats || n1 + 2 * up1->nnz > a.r1_floats)
    return MVH_OK;
  const size_t lds = ((size_t)a.r1_floats + 2 * (size_t)a.r23_floats + a.wl_floats) * 4 + (size_t)a.emax * 8 +
                     (size_t)((a.nmax + 7) & ~3) * 4 + (size_t)a.nmax * 64;
  if (lds > 160 * 1024) return MVH_OK;
#define MVH_MID(A_, B_, C_)                                                                                               \
  do {                                                                                                                    \
    auto kern = k_mid_dec_fwd<A_, B_, C_>;                                                                                \
    static size_t attr = 0;                                                                                               \
    if (lds > attr) {                                                                                                     \
      MVH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
      attr = lds;                                                                                                         \
    }                                                                                                                     \
    hipLaunchKernelGGL(kern, dim3(B), dim3(kMidThreads), lds, st, a);                                                     \
  } while (0)
  MVH_MID(32, 32, 16);
#undef MVH_MID
  MVH_LAUNCH_CHECK();
  *handled = true;
  return MVH_OK;
}

}  // namespace mvh
extern "C" void mvh_debug_mid_tlog(unsigned long long* p) { mvh::g_mid_tlog = p; }

