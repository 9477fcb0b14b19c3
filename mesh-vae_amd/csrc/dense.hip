// Dense parts of rows E/D/K/Z/R: nn.Linear (+relu +dropout), the classifier / latent heads
// and the reparameterisation (cheb_VAE.py:203-226, 253-258, 270-280, 309-319).
// These are skinny problems (M = batch), a few MFLOP per mesh; one LDS-tiled fp32 GEMM with
// generic strides serves forward, dX and dW, and the latent head is one fused kernel per pass.
#include <type_traits>

#include "common.hpp"

namespace mvh {

// ------------------------------------------------------------------ small strided GEMM (MFMA)
// C[M,N] = sum_k A(m,k) B(k,n) with A(m,k) = A[m*sam + k*sak], B(k,n) = Bm[k*sbk + n*sbn]
// (+bias[n]) -> act -> dropout.  The FC layers are skinny (M = batch = 64): one wave owns a
// 16x16 output tile and runs the exact-fp32 matrix instruction v_mfma_f32_16x16x4_f32 over K
// (lane l feeds A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15]); operands that are contiguous
// in K are fetched as 16-byte vectors covering four k-steps (the k order inside a 16-chunk is
// permuted identically for A and B, which a dot product does not care about).
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool KC>
__device__ __forceinline__ void load_k4(const float* __restrict__ base, long long s_row, long long s_k, int row,
                                        int nrows, int k, int K, float (&v)[4]) {
  v[0] = v[1] = v[2] = v[3] = 0.f;
  if (row >= nrows) return;
  const float* p = base + (long long)row * s_row;
  if constexpr (KC) {
    if (k + 3 < K) {
      const float4 t = *reinterpret_cast<const float4*>(p + k);
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
      return;
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t)
    if (k + t < K) v[t] = p[(long long)(k + t) * s_k];
}

constexpr int kGemmWaves = 8;

template <bool A_KC, bool B_KC, bool A_MASK>
__global__ void __launch_bounds__(64 * kGemmWaves)
k_gemm16(const float* __restrict__ A, long long sam, long long sak, const float* __restrict__ Bm, long long sbk,
         long long sbn, float* __restrict__ C, int M, int N, int K, const float* __restrict__ bias, int act,
         const float* __restrict__ drop_u, float p, const float* __restrict__ a_mask, float a_scale,
         float* __restrict__ ones_out) {
  // one block = one 16x16 output tile; its 8 waves split K (interleaved 16-chunks) and are
  // summed through LDS in fixed order.  These GEMMs are latency-bound (a few MFLOP): the full
  // 16-chunks run as branch-free groups of up to eight so that many chunks of loads are in flight per
  // wave; rows/columns past the edge are clamped (their results are never stored) and only
  // the K tail takes the bounds-checked loads.
  // a_mask (same indexing as A): A elements become (mask > 0 ? a * a_scale : 0) on load -- the
  // relu/dropout backward mask of nn.Linear, so no separate "dpre" pass exists.
  // ones_out: a virtual column n == N with B = 1 whose results (row sums of A^T...) go to
  // ones_out[m]: the bias gradient comes out of the dW GEMM for free.
  __shared__ float red[kGemmWaves - 1][64][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 16;
  const int i = lane & 15, kq = lane >> 4;
  const bool ones_col = ones_out && (n0 + i == N);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int arow = min(m0 + i, M - 1), brow = min(n0 + i, N - 1);
  const float* __restrict__ ap = A + (long long)arow * sam;
  const float* __restrict__ mp = a_mask ? a_mask + (long long)arow * sam : nullptr;
  const float* __restrict__ bp = Bm + (long long)brow * sbn;
  auto load_chunk = [&](int k, float (&a)[4], float (&b)[4]) {  // k .. k+3 all < K
    if constexpr (A_KC) {
      const float4 t = *reinterpret_cast<const float4*>(ap + k);
      a[0] = t.x; a[1] = t.y; a[2] = t.z; a[3] = t.w;
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) a[t] = ap[(long long)(k + t) * sak];
    }
    if constexpr (A_MASK) {  // compile-time: a run-time branch here fences each chunk's loads
      float mk[4];
      if constexpr (A_KC) {
        const float4 t = *reinterpret_cast<const float4*>(mp + k);
        mk[0] = t.x; mk[1] = t.y; mk[2] = t.z; mk[3] = t.w;
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) mk[t] = mp[(long long)(k + t) * sak];
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) a[t] = mk[t] > 0.f ? a[t] * a_scale : 0.f;
    }
    if constexpr (B_KC) {
      const float4 t = *reinterpret_cast<const float4*>(bp + k);
      b[0] = t.x; b[1] = t.y; b[2] = t.z; b[3] = t.w;
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) b[t] = bp[(long long)(k + t) * sbk];
    }
    if (ones_col) b[0] = b[1] = b[2] = b[3] = 1.f;
  };
  auto mma = [&](const float (&a)[4], const float (&b)[4]) {
#pragma unroll
    for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[t], acc, 0, 0, 0);
  };
  const int Kfull = K & ~15;
  const int NW = blockDim.x >> 6;  // 1..kGemmWaves waves, chosen from K by the launcher
  const int kStep = 16 * NW;
  int k0 = wave * 16;
  // this wave's 16-chunks k0, k0 + kStep, ...: the loads of up to eight of them are in flight together (a layer of
  // K = 640 is five chunks per wave: one memory round trip, not two); the matrix instructions stay in chunk order
  auto run = [&](auto n_tag) {
    constexpr int NCH = decltype(n_tag)::value;
    float a[NCH][4], b[NCH][4];
#pragma unroll
    for (int u = 0; u < NCH; ++u) load_chunk(k0 + u * kStep + 4 * kq, a[u], b[u]);
#pragma unroll
    for (int u = 0; u < NCH; ++u) mma(a[u], b[u]);
    k0 += NCH * kStep;
  };
  int rem = k0 < Kfull ? (Kfull - k0 + kStep - 1) / kStep : 0;
  for (; rem >= 8; rem -= 8) run(std::integral_constant<int, 8>());
  switch (rem) {
    case 7: run(std::integral_constant<int, 7>()); break;
    case 6: run(std::integral_constant<int, 6>()); break;
    case 5: run(std::integral_constant<int, 5>()); break;
    case 4: run(std::integral_constant<int, 4>()); break;
    case 3: run(std::integral_constant<int, 3>()); break;
    case 2: run(std::integral_constant<int, 2>()); break;
    case 1: run(std::integral_constant<int, 1>()); break;
    default: break;
  }
  if (Kfull < K && wave == ((Kfull >> 4) % NW)) {  // K tail: bounds-checked loads
    float a[4], b[4];
    load_k4<A_KC>(A, sam, sak, m0 + i, M, Kfull + 4 * kq, K, a);
    if (a_mask) {
      float mk[4];
      load_k4<A_KC>(a_mask, sam, sak, m0 + i, M, Kfull + 4 * kq, K, mk);
#pragma unroll
      for (int t = 0; t < 4; ++t) a[t] = mk[t] > 0.f ? a[t] * a_scale : 0.f;
    }
    load_k4<B_KC>(Bm, sbn, sbk, n0 + i, N, Kfull + 4 * kq, K, b);
    if (ones_col) {
#pragma unroll
      for (int t = 0; t < 4; ++t) b[t] = (Kfull + 4 * kq + t < K) ? 1.f : 0.f;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[t], acc, 0, 0, 0);
  }
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave - 1][lane][r] = acc[r];
  }
  __syncthreads();
  if (wave > 0) return;
#pragma unroll
  for (int w = 0; w < NW - 1; ++w)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] += red[w][lane][r];
  const float keep_scale = (drop_u && p > 0.f) ? 1.f / (1.f - p) : 1.f;
  const int n = n0 + i;
  if (ones_col) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + 4 * kq + r;
      if (m < M) ones_out[m] = acc[r];
    }
    return;
  }
  if (n >= N) return;
  const float bv = bias ? bias[n] : 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int m = m0 + 4 * kq + r;
    if (m >= M) continue;
    float v = acc[r] + bv;
    if (act == MVH_ACT_RELU) v = fmaxf(v, 0.f);
    if (drop_u && p > 0.f) v = (drop_u[(long long)m * N + n] >= p) ? v * keep_scale : 0.f;
    C[(long long)m * N + n] = v;
  }
}

static int launch_gemm_ex(hipStream_t st, const float* A, long long sam, long long sak, const float* Bm,
                          long long sbk, long long sbn, float* C, int M, int N, int K, const float* bias,
                          int act, const float* drop_u, float p, const float* a_mask, float a_scale,
                          float* ones_out) {
  if (M == 0 || N == 0) return MVH_OK;
  const bool akc = (sak == 1) && (sam % 4 == 0) && ((uintptr_t)A % 16 == 0) && ((uintptr_t)a_mask % 16 == 0);
  const bool bkc = (sbk == 1) && (sbn % 4 == 0) && ((uintptr_t)Bm % 16 == 0);
  const dim3 grid(cdiv(N + (ones_out ? 1 : 0), 16), cdiv(M, 16));
  int waves = cdiv(K, 16);  // one 16-chunk of K per wave at least
  waves = waves < 1 ? 1 : (waves > kGemmWaves ? kGemmWaves : waves);
#define MVH_GEMM(AK, BK)                                                                                   \
  do { if (a_mask) MVH_GEMM_M(AK, BK, true); else MVH_GEMM_M(AK, BK, false); } while (0)
#define MVH_GEMM_M(AK, BK, MK)                                                                             \
  hipLaunchKernelGGL((k_gemm16<AK, BK, MK>), grid, dim3(64 * waves), 0, st, A, sam, sak, Bm, sbk, sbn, C, M, N, K, bias, \
                     act, drop_u, p, a_mask, a_scale, ones_out)
  if (akc && bkc) MVH_GEMM(true, true);
  else if (akc) MVH_GEMM(true, false);
  else if (bkc) MVH_GEMM(false, true);
  else MVH_GEMM(false, false);
#undef MVH_GEMM
#undef MVH_GEMM_M
  MVH_LAUNCH_CHECK();
  return MVH_OK;
}

int launch_gemm(hipStream_t st, const float* A, long long sam, long long sak, const float* Bm,
                long long sbk, long long sbn, float* C, int M, int N, int K, const float* bias,
                int act, const float* drop_u, float p) {
  return launch_gemm_ex(st, A, sam, sak, Bm, sbk, sbn, C, M, N, K, bias, act, drop_u, p, nullptr, 1.f, nullptr);
}

// dpre = dy masked by the forward output (relu and/or dropout zeroes) and rescaled by 1/(1-p)
__global__ void __launch_bounds__(256)
k_linear_dpre(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dpre,
              long long n, int masked, float scale) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float d = dy[i];
  if (masked) d = (y[i] > 0.f) ? d * scale : 0.f;
  dpre[i] = d;
}

__global__ void __launch_bounds__(256)
k_colsum(const float* __restrict__ a, float* __restrict__ out, int M, int N) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;  // independent chains: loads pipeline, order fixed
  int m = 0;
  for (; m + 3 < M; m += 4) {
    s0 += a[(long long)m * N + n];
    s1 += a[(long long)(m + 1) * N + n];
    s2 += a[(long long)(m + 2) * N + n];
    s3 += a[(long long)(m + 3) * N + n];
  }
  for (; m < M; ++m) s0 += a[(long long)m * N + n];
  out[n] = (s0 + s1) + (s2 + s3);
}

// ------------------------------------------------------------------ latent head, forward
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

__global__ void __launch_bounds__(1024)
k_latent_fwd(const float* __restrict__ h, const float* __restrict__ y, const float* __restrict__ drop_u,
             float p, const float* __restrict__ Wc, const float* __restrict__ bc,
             const float* __restrict__ Wm, const float* __restrict__ bm, const float* __restrict__ Wv,
             const float* __restrict__ bv, const float* __restrict__ eps, float* __restrict__ y_hat,
             float* __restrict__ mu, float* __restrict__ logvar, float* __restrict__ z,
             float* __restrict__ zy, int H, int C, int Z, const float* __restrict__ Wd,
             const float* __restrict__ bd, const float* __restrict__ drop_d, float* __restrict__ d1) {
  extern __shared__ float lds[];
  float* hy = lds;            // [C+H]  = cat[y, h]                      (cheb_VAE.py:209)
  float* hd = lds + C + H;    // [H]    = classifier's dropout(h)        (cheb_VAE.py:255)
  float* outs = hd + H;       // [C+2Z] logits | mu | logvar
  float* zyl = outs + C + 2 * Z;  // [C+Z] = cat[y, z] for the fused dec_lin
  const int b = blockIdx.x;
  const float scale = (drop_u && p > 0.f) ? 1.f / (1.f - p) : 1.f;
  // Fused dec_lin (cheb_VAE.py:277, Wd != NULL; the launcher checks C + Z < 32 and H <= 512): d1 = dropout(relu(
  // cat[y, z] Wd^T + bd)) for this mesh, in the arithmetic of k_gemm16 (the module path's mvh_linear_fwd) to the
  // last bit -- same matrix instructions on the same lanes, its per-wave partial sums added in its order -- so one
  // launch less changes no result.  The weight tiles do not depend on z: they are fetched now, under the heads.
  const int Kd = C + Z;
  float wdr[2][2][4];   // [tile][full 16-chunk | K tail][t]
  float bdr[2] = {0.f, 0.f}, udr[2] = {1.f, 1.f};   // bias and dropout uniform of the lane's output
  if (Wd) {
    const int lane_ = threadIdx.x & 63, i_ = lane_ & 15, kq_ = lane_ >> 4, wave_ = threadIdx.x >> 6;
#pragma unroll
    for (int tl = 0; tl < 2; ++tl) {
      const int n0 = (wave_ + 16 * tl) * 16;
      const float* wr = Wd + (long long)min(n0 + i_, H - 1) * Kd;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int k = 4 * kq_ + t, kt = (Kd & ~15) + k;
        wdr[tl][0][t] = (n0 < H && k < (Kd & ~15)) ? wr[k] : 0.f;
        wdr[tl][1][t] = (n0 < H && kt < Kd) ? wr[kt] : 0.f;
      }
      const bool mine = kq_ == 0 && n0 + i_ < H;
      bdr[tl] = mine ? bd[n0 + i_] : 0.f;
      udr[tl] = (mine && drop_d && p > 0.f) ? drop_d[(long long)b * H + n0 + i_] : 1.f;
    }
  }
  // The head weights do not depend on h either: when a lane's share of the (at most three) outputs its wave owns
  // fits in registers, every global load of the kernel is issued here, in front of the first barrier
  const int wave = threadIdx.x / 64, lane = threadIdx.x % 64, nw = blockDim.x / 64;
  constexpr int kHR = 3, kHC = 9;
  const bool fast = (C + H <= 64 * kHC) && (C + 2 * Z <= kHR * nw);
  float wq[kHR][kHC], bq[kHR];
  if (fast) {
#pragma unroll
    for (int r = 0; r < kHR; ++r) {
      const int o = wave + r * nw;
      const bool live = o < C + 2 * Z, cls = o < C;
      const int zz = cls ? 0 : (o - C) % Z;
      const bool is_mu = (o - C) < Z;
      const float* w = cls ? Wc + (long long)o * H : (is_mu ? Wm : Wv) + (long long)zz * (C + H);
      const int len = cls ? H : C + H;
      bq[r] = live ? (cls ? bc[o] : (is_mu ? bm[zz] : bv[zz])) : 0.f;
#pragma unroll
      for (int c = 0; c < kHC; ++c) wq[r][c] = (live && lane + 64 * c < len) ? w[lane + 64 * c] : 0.f;
    }
  }
  for (int j = threadIdx.x; j < H; j += blockDim.x) {
    const float v = h[(long long)b * H + j];
    hy[C + j] = v;
    hd[j] = (drop_u && p > 0.f) ? ((drop_u[(long long)b * H + j] >= p) ? v * scale : 0.f) : v;
  }
  for (int j = threadIdx.x; j < C; j += blockDim.x) hy[j] = y[(long long)b * C + j];
  __syncthreads();
  if (fast) {
#pragma unroll
    for (int r = 0; r < kHR; ++r) {
      const int o = wave + r * nw;
      if (o >= C + 2 * Z) break;  // wave-uniform
      const float* x = (o < C) ? hd : hy;
      const int len = (o < C) ? H : C + H;
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < kHC; ++c) s = fmaf(lane + 64 * c < len ? x[lane + 64 * c] : 0.f, wq[r][c], s);
      s = wave_sum(s);
      if (lane == 0) outs[o] = s + bq[r];
    }
  } else {
    for (int o = wave; o < C + 2 * Z; o += nw) {
      float s = 0.f;
      const bool cls = o < C;
      const int zz = cls ? 0 : (o - C) % Z;
      const bool is_mu = (o - C) < Z;
      const float* w = cls ? Wc + (long long)o * H : (is_mu ? Wm : Wv) + (long long)zz * (C + H);
      const float* x = cls ? hd : hy;
      const int len = cls ? H : C + H;
      float s1 = 0.f, s2 = 0.f, s3 = 0.f;  // four independent chains: the row's loads are in flight together
      int j = lane;
      for (; j + 192 < len; j += 256) {
        s = fmaf(x[j], w[j], s);
        s1 = fmaf(x[j + 64], w[j + 64], s1);
        s2 = fmaf(x[j + 128], w[j + 128], s2);
        s3 = fmaf(x[j + 192], w[j + 192], s3);
      }
      for (; j < len; j += 64) s = fmaf(x[j], w[j], s);
      s = (s + s1) + (s2 + s3);
      s = wave_sum(s);
      if (lane == 0) outs[o] = s + (cls ? bc[o] : (is_mu ? bm[zz] : bv[zz]));
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {  // softmax over the C logits (cheb_VAE.py:256)
    float mx = outs[0];
    for (int c = 1; c < C; ++c) mx = fmaxf(mx, outs[c]);
    float den = 0.f;
    for (int c = 0; c < C; ++c) den += expf(outs[c] - mx);
    for (int c = 0; c < C; ++c) y_hat[(long long)b * C + c] = expf(outs[c] - mx) / den;
  }
  for (int t = threadIdx.x; t < Z; t += blockDim.x) {
    const float m = outs[C + t], lv = outs[C + Z + t];
    // reparameterize (cheb_VAE.py:309-319) or z = mu in test mode (:221)
    const float zz = eps ? fmaf(eps[(long long)b * Z + t], expf(lv * 0.5f), m) : m;
    mu[(long long)b * Z + t] = m;
    logvar[(long long)b * Z + t] = lv;
    z[(long long)b * Z + t] = zz;
    zy[(long long)b * (C + Z) + C + t] = zz;
    zyl[C + t] = zz;
  }
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    zy[(long long)b * (C + Z) + c] = hy[c];
    zyl[c] = hy[c];
  }
  if (!Wd) return;
  __syncthreads();
  {
    const int lane = threadIdx.x & 63, i = lane & 15, kq = lane >> 4, wave = threadIdx.x >> 6;
    const int Kfull = Kd & ~15;
    const float keep = (drop_d && p > 0.f) ? 1.f / (1.f - p) : 1.f;
#pragma unroll
    for (int tl = 0; tl < 2; ++tl) {
      const int n0 = (wave + 16 * tl) * 16;
      if (n0 >= H) continue;  // wave-uniform
      // k_gemm16 with K = Kd <= 32 runs NW = cdiv(Kd, 16) waves: wave 0 the full chunk [0, 16), the K tail on wave
      // (Kfull / 16) % NW, and wave 0 then adds the others' accumulators in wave order
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      if (Kfull >= 16) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(zyl[4 * kq + t], wdr[tl][0][t], acc0, 0, 0, 0);
      }
      if (Kfull < Kd) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int k = Kfull + 4 * kq + t;
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(k < Kd ? zyl[k] : 0.f, wdr[tl][1][t], acc1, 0, 0, 0);
        }
      }
      // NW == 1 (Kd < 16): the tail lands on wave 0 itself, i.e. on the same accumulator; NW == 2: a separate one
      float v = (Kfull >= 16) ? ((Kfull < Kd) ? acc0[0] + acc1[0] : acc0[0]) : acc1[0];
      const int n = n0 + i;
      if (kq == 0 && n < H) {
        v += bdr[tl];
        v = fmaxf(v, 0.f);
        if (drop_d && p > 0.f) v = (udr[tl] >= p) ? v * keep : 0.f;
        d1[(long long)b * H + n] = v;
      }
    }
  }
}

// ------------------------------------------------------------------ latent head, backward
__global__ void __launch_bounds__(512)
k_latent_bwd(const float* __restrict__ drop_u, float p, const float* __restrict__ Wc,
             const float* __restrict__ Wm, const float* __restrict__ Wv, const float* __restrict__ eps,
             const float* __restrict__ y_hat, const float* __restrict__ logvar,
             const float* __restrict__ d_yhat, const float* __restrict__ d_mu,
             const float* __restrict__ d_logvar, const float* __restrict__ d_zy, float* __restrict__ dh,
             float* __restrict__ dpre, int H, int C, int Z, const float* __restrict__ Wd,
             const float* __restrict__ d1, const float* __restrict__ g_d1, float* __restrict__ g_zy) {
  extern __shared__ float lds[];
  float* g = lds;  // [C+2Z]: dlogit | dmu | dlogvar
  const int b = blockIdx.x;
  float* part = g + C + 2 * Z;      // [2 tiles][8 waves][16]
  float* dzl = part + 2 * 8 * 16;   // [C+Z]
  float* yl = dzl + C + Z;          // [2C]: y_hat | d_yhat
  // ---- the head inputs of thread t < Z / c < C do not depend on g_zy: their loads are issued first
  const int no = C + 2 * Z;
  float p_mu = 0.f, p_lv = 0.f, p_eps = 0.f, p_logvar = 0.f, p_y = 0.f, p_dy = 0.f;
  if ((int)threadIdx.x < Z) {
    p_mu = d_mu[(long long)b * Z + threadIdx.x];
    p_lv = d_logvar[(long long)b * Z + threadIdx.x];
    if (eps) {
      p_eps = eps[(long long)b * Z + threadIdx.x];
      p_logvar = logvar[(long long)b * Z + threadIdx.x];
    }
  }
  if ((int)threadIdx.x < C) {
    p_y = y_hat[(long long)b * C + threadIdx.x];
    p_dy = d_yhat[(long long)b * C + threadIdx.x];
  }
  // Fused dX of dec_lin (Wd != NULL; launcher: C + Z <= 32, 512 threads): g_zy[b] = (g_d1[b] masked by d1[b] > 0 and
  // rescaled) Wd, i.e. mvh_linear_bwd's dX GEMM for this mesh in the arithmetic of k_gemm16 to the last bit: wave w
  // plays that kernel's wave w for both 16-column tiles (its interleaved 16-chunks of K = H and, on one wave, the K
  // tail, same lanes, same instruction order; the A operand is shared by the tiles); the accumulators are then added
  // in wave order.  (8 waves and ~70 VGPRs on purpose: this launch runs beside the chip-filling weight-gradient
  // kernels of the side lane and must fit into what they leave free.)
  int NW = (H + 15) / 16;
  NW = NW < 1 ? 1 : (NW > kGemmWaves ? kGemmWaves : NW);
  if (Wd) {
    const int Kd = C + Z;
    const int lane = threadIdx.x & 63, i = lane & 15, kq = lane >> 4, w = threadIdx.x >> 6;
    const bool two = Kd > 16;
    if (w < NW) {  // wave-uniform
      const int Kfull = H & ~15, kStep = 16 * NW;
      const float sc = p > 0.f ? 1.f / (1.f - p) : 1.f;
      const float* __restrict__ ap = g_d1 + (long long)b * H;
      const float* __restrict__ mp = d1 + (long long)b * H;
      const float* __restrict__ bp0 = Wd + min(i, Kd - 1);
      const float* __restrict__ bp1 = Wd + min(16 + i, Kd - 1);
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      auto load_chunk = [&](int k, float (&a)[4], float (&b0)[4], float (&b1)[4]) {  // k .. k + 3 all < H
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float av = ap[k + t], mv = mp[k + t];
          a[t] = mv > 0.f ? av * sc : 0.f;
          b0[t] = bp0[(long long)(k + t) * Kd];
          b1[t] = two ? bp1[(long long)(k + t) * Kd] : 0.f;
        }
      };
      auto mma = [&](const float (&a)[4], const float (&b0)[4], const float (&b1)[4]) {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b0[t], acc0, 0, 0, 0);
        if (two) {
#pragma unroll
          for (int t = 0; t < 4; ++t) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b1[t], acc1, 0, 0, 0);
        }
      };
      int k0 = 16 * w;
      for (; k0 + 3 * kStep < Kfull; k0 += 4 * kStep) {  // four chunks of loads in flight
        float a[4][4], b0[4][4], b1[4][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) load_chunk(k0 + c * kStep + 4 * kq, a[c], b0[c], b1[c]);
#pragma unroll
        for (int c = 0; c < 4; ++c) mma(a[c], b0[c], b1[c]);
      }
      for (; k0 < Kfull; k0 += kStep) {
        float a[4], b0[4], b1[4];
        load_chunk(k0 + 4 * kq, a, b0, b1);
        mma(a, b0, b1);
      }
      if (Kfull < H && w == ((Kfull >> 4) % NW)) {  // K tail
        float a[4], b0[4], b1[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int k = Kfull + 4 * kq + t;
          a[t] = b0[t] = b1[t] = 0.f;
          if (k < H) {
            a[t] = mp[k] > 0.f ? ap[k] * sc : 0.f;
            b0[t] = bp0[(long long)k * Kd];
            if (two) b1[t] = bp1[(long long)k * Kd];
          }
        }
        mma(a, b0, b1);
      }
      if (kq == 0) {   // output row 0 (every row of the tile is this mesh)
        part[w * 16 + i] = acc0[0];
        part[(kGemmWaves + w) * 16 + i] = acc1[0];
      }
    }
  }
  if ((int)threadIdx.x < C) {
    yl[threadIdx.x] = p_y;
    yl[C + threadIdx.x] = p_dy;
  }
  __syncthreads();
  if (Wd) {
    const int Kd = C + Z;
    for (int n = threadIdx.x; n < Kd; n += blockDim.x) {
      const int tile = n >> 4, i2 = n & 15;
      float tot = part[(tile * kGemmWaves) * 16 + i2];
      for (int w = 1; w < NW; ++w) tot += part[(tile * kGemmWaves + w) * 16 + i2];
      dzl[n] = tot;
      g_zy[(long long)b * Kd + n] = tot;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int c = 0; c < C; ++c) s = fmaf(yl[C + c], yl[c], s);
    for (int c = 0; c < C; ++c) g[c] = yl[c] * (yl[C + c] - s);
  }
  if ((int)threadIdx.x < Z) {
    const int t = threadIdx.x;
    const float dz = Wd ? dzl[C + t] : d_zy[(long long)b * (C + Z) + C + t];
    float dm = p_mu + dz;
    float dl = p_lv;
    if (eps) dl = fmaf(dz * p_eps, 0.5f * expf(0.5f * p_logvar), dl);
    g[C + t] = dm;
    g[C + Z + t] = dl;
  }
  __syncthreads();
  for (int o = threadIdx.x; o < no; o += blockDim.x) dpre[(long long)b * no + o] = g[o];
  const float scale = (drop_u && p > 0.f) ? 1.f / (1.f - p) : 1.f;
  for (int j = threadIdx.x; j < H; j += blockDim.x) {
    float a = 0.f;
    for (int c = 0; c < C; ++c) a = fmaf(g[c], Wc[(long long)c * H + j], a);
    if (drop_u && p > 0.f) a = (drop_u[(long long)b * H + j] >= p) ? a * scale : 0.f;
    for (int t = 0; t < Z; ++t) {
      a = fmaf(g[C + t], Wm[(long long)t * (C + H) + C + j], a);
      a = fmaf(g[C + Z + t], Wv[(long long)t * (C + H) + C + j], a);
    }
    dh[(long long)b * H + j] = a;
  }
}

// parameter gradients of the three heads: thread (o, j); j == ld is the bias slot
__global__ void __launch_bounds__(256)
k_latent_wgrad(const float* __restrict__ h, const float* __restrict__ y, const float* __restrict__ drop_u,
               float p, const float* __restrict__ dpre, float* __restrict__ dWc, float* __restrict__ dbc,
               float* __restrict__ dWm, float* __restrict__ dbm, float* __restrict__ dWv,
               float* __restrict__ dbv, int B, int H, int C, int Z) {
  const int ld = C + H;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int o = blockIdx.y;
  if (j > ld) return;
  const int no = C + 2 * Z;
  const float scale = (drop_u && p > 0.f) ? 1.f / (1.f - p) : 1.f;
  float s = 0.f;
  // Sixteen meshes' operands are fetched before their products are added (clamped rows past the batch, predicated adds):
  // the sums run in the order of the plain loop -- chain t takes the meshes b = t mod 4, ascending -- but the kernel waits
  // for B / 16 rounds of loads instead of B / 4 (22.7 -> 11.3 us in the step at B = 64; among the last launches of the dense lane)
  constexpr int kU = 16;
  if (o < C) {
    if (j > H) return;  // classifier input is only [H] (+ bias slot at j == H)
    float s4[4] = {0.f, 0.f, 0.f, 0.f};
    for (int b0 = 0; b0 < B; b0 += kU) {
      float dp[kU], in[kU], u[kU];
#pragma unroll
      for (int t = 0; t < kU; ++t) {
        const int b = min(b0 + t, B - 1);
        dp[t] = dpre[(long long)b * no + o];
        in[t] = (j < H) ? h[(long long)b * H + j] : 1.f;
        u[t] = (j < H && drop_u && p > 0.f) ? drop_u[(long long)b * H + j] : 1.f;
      }
#pragma unroll
      for (int t = 0; t < kU; ++t) {
        if (b0 + t >= B) break;
        float v = in[t];
        if (j < H && drop_u && p > 0.f) v = (u[t] >= p) ? v * scale : 0.f;
        s4[t & 3] = fmaf(dp[t], v, s4[t & 3]);
      }
    }
    s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    if (j < H) dWc[(long long)o * H + j] = s; else dbc[o] = s;
  } else {
    float s4[4] = {0.f, 0.f, 0.f, 0.f};
    for (int b0 = 0; b0 < B; b0 += kU) {
      float dp[kU], in[kU];
#pragma unroll
      for (int t = 0; t < kU; ++t) {
        const int b = min(b0 + t, B - 1);
        dp[t] = dpre[(long long)b * no + o];
        in[t] = (j == ld) ? 1.f : (j < C ? y[(long long)b * C + j] : h[(long long)b * H + (j - C)]);
      }
#pragma unroll
      for (int t = 0; t < kU; ++t) {
        if (b0 + t >= B) break;
        s4[t & 3] = fmaf(dp[t], in[t], s4[t & 3]);
      }
    }
    s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    const int zz = (o - C) % Z;
    const bool is_mu = (o - C) < Z;
    if (j < ld) (is_mu ? dWm : dWv)[(long long)zz * ld + j] = s;
    else (is_mu ? dbm : dbv)[zz] = s;
  }
}

}  // namespace mvh

using namespace mvh;

extern "C" int mvh_linear_fwd(mvh_stream_t stream, const float* x, const float* W, const float* bias,
                              float* y, int32_t B, int32_t in_f, int32_t out_f, int32_t act,
                              const float* drop_u, float p) {
  MVH_REQUIRE(x && W && y, "linear_fwd: null tensor");
  MVH_REQUIRE(B >= 0 && in_f > 0 && out_f > 0, "linear_fwd: bad sizes");
  MVH_REQUIRE(p >= 0.f && p < 1.f, "linear_fwd: dropout p=%f out of range", p);
  // y[b,o] = sum_i x[b,i] W[o,i]
  return launch_gemm((hipStream_t)stream, x, in_f, 1, W, 1, in_f, y, B, out_f, in_f, bias, act, drop_u, p);
}

extern "C" int mvh_linear_bwd(mvh_stream_t stream, const float* x, const float* W, const float* y,
                              const float* dy, float* dx, float* dW, float* db, int32_t B, int32_t in_f,
                              int32_t out_f, int32_t act, float p, void* ws, size_t ws_bytes) {
  MVH_REQUIRE(x && W && dy && (dW || dx), "linear_bwd: null tensor");
  const bool masked = (act == MVH_ACT_RELU);
  MVH_REQUIRE(masked || p == 0.f, "linear_bwd: dropout without relu is not supported");
  MVH_REQUIRE(!masked || y, "linear_bwd: relu/dropout backward needs the forward output");
  hipStream_t st = (hipStream_t)stream;
  (void)ws;
  (void)ws_bytes;
  // dpre = masked(dy) is formed on the fly inside the GEMM A-operand loads
  const float* mask = masked ? y : nullptr;
  const float scale = p > 0.f ? 1.f / (1.f - p) : 1.f;
  if (dx)  // dx[b,i] = sum_o dpre[b,o] W[o,i]
    if (int rc = launch_gemm_ex(st, dy, out_f, 1, W, in_f, 1, dx, B, in_f, out_f, nullptr, 0, nullptr, 0.f, mask, scale,
                                nullptr)) return rc;
  if (!dW) return MVH_OK;  // dX-only call (the step engine runs the weight gradients on a side stream)
  // dW[o,i] = sum_b dpre[b,o] x[b,i]  and  db[o] = sum_b dpre[b,o] (virtual ones column)
  return launch_gemm_ex(st, dy, 1, out_f, x, in_f, 1, dW, out_f, in_f, B, nullptr, 0, nullptr, 0.f, mask, scale, db);
}

// latent heads (+ optionally dec_lin, see k_latent_fwd) of the whole batch in one launch
int mvh::latent_fwd_impl(hipStream_t st, const float* h, const float* y, const float* drop_u, float p, const float* Wc,
                         const float* bc, const float* Wm, const float* bm, const float* Wv, const float* bv,
                         const float* eps, float* y_hat, float* mu, float* logvar, float* z, float* zy, int B, int H,
                         int C, int Z, const float* Wd, const float* bd, const float* drop_d, float* d1, bool* fused) {
  MVH_REQUIRE(h && y && Wc && bc && Wm && bm && Wv && bv && y_hat && mu && logvar && z && zy, "latent_fwd: null tensor");
  MVH_REQUIRE(H > 0 && C > 0 && Z > 0 && B >= 0, "latent_fwd: bad sizes");
  MVH_REQUIRE(p >= 0.f && p < 1.f, "latent_fwd: dropout p out of range");
  const bool fuse = Wd && bd && d1 && C + Z < 32 && H <= 512 && !dbg().force_generic && !dbg().no_head_fuse;
  if (fused) *fused = fuse;
  if (B == 0) return MVH_OK;
  const size_t lds = (size_t)(C + H + H + C + 2 * Z + C + Z) * sizeof(float);
  MVH_REQUIRE(lds <= 64 * 1024, "latent_fwd: hidden size %d too large", H);
  hipLaunchKernelGGL(k_latent_fwd, dim3(B), dim3(1024), lds, st, h, y, drop_u, p, Wc, bc, Wm, bm, Wv, bv, eps, y_hat, mu,
                     logvar, z, zy, H, C, Z, fuse ? Wd : nullptr, bd, drop_d, d1);
  MVH_LAUNCH_CHECK();
  return MVH_OK;
}

extern "C" int mvh_vae_latent_fwd(mvh_stream_t stream, const float* h, const float* y, const float* drop_u,
                                  float p, const float* Wc, const float* bc, const float* Wm,
                                  const float* bm, const float* Wv, const float* bv, const float* eps,
                                  float* y_hat, float* mu, float* logvar, float* z, float* zy, int32_t B,
                                  int32_t H, int32_t C, int32_t Z) {
  return latent_fwd_impl((hipStream_t)stream, h, y, drop_u, p, Wc, bc, Wm, bm, Wv, bv, eps, y_hat, mu, logvar, z, zy, B, H,
                         C, Z, nullptr, nullptr, nullptr, nullptr, nullptr);
}

extern "C" int mvh_vae_latent_bwd(mvh_stream_t stream, const float* h, const float* y, const float* drop_u,
                                  float p, const float* Wc, const float* Wm, const float* Wv,
                                  const float* eps, const float* y_hat, const float* logvar,
                                  const float* d_yhat, const float* d_mu, const float* d_logvar,
                                  const float* d_zy, float* dh, float* dWc, float* dbc, float* dWm,
                                  float* dbm, float* dWv, float* dbv, int32_t B, int32_t H, int32_t C,
                                  int32_t Z, void* ws, size_t ws_bytes) {
  MVH_REQUIRE(h && y && Wc && Wm && Wv && y_hat && logvar && d_yhat && d_mu && d_logvar && d_zy && dh &&
                  dWc && dbc && dWm && dbm && dWv && dbv, "latent_bwd: null tensor");
  const int no = C + 2 * Z;
  MVH_REQUIRE(ws && ws_bytes >= (size_t)B * no * sizeof(float), "latent_bwd: workspace too small");
  if (int rc = latent_bwd_heads((hipStream_t)stream, drop_u, p, Wc, Wm, Wv, eps, y_hat, logvar, d_yhat, d_mu, d_logvar,
                                d_zy, dh, (float*)ws, B, H, C, Z)) return rc;
  return latent_bwd_wgrad((hipStream_t)stream, h, y, drop_u, p, (const float*)ws, dWc, dbc, dWm, dbm, dWv, dbv, B, H, C, Z);
}

// the two halves of mvh_vae_latent_bwd (the step engine runs the second one on a side stream)
int mvh::latent_bwd_heads(hipStream_t st, const float* drop_u, float p, const float* Wc, const float* Wm,
                          const float* Wv, const float* eps, const float* y_hat, const float* logvar,
                          const float* d_yhat, const float* d_mu, const float* d_logvar, const float* d_zy,
                          float* dh, float* dpre, int B, int H, int C, int Z, const float* Wd, const float* d1,
                          const float* g_d1, float* g_zy, bool* fused) {
  const int no = C + 2 * Z;
  // optional: the dX GEMM of dec_lin (g_d1 -> g_zy, which then is an output) inside the same launch
  const bool fuse = Wd && d1 && g_d1 && g_zy && C + Z <= 32 && !dbg().force_generic && !dbg().no_head_fuse;
  if (fused) *fused = fuse;
  MVH_REQUIRE(fuse || d_zy, "latent_bwd: null tensor");
  if (B > 0) {
    const size_t lds = (size_t)(no + 2 * kGemmWaves * 16 + C + Z + 2 * C) * sizeof(float);
    hipLaunchKernelGGL(k_latent_bwd, dim3(B), dim3(fuse ? 512 : 256), lds, st, drop_u, p, Wc, Wm, Wv, eps, y_hat, logvar,
                       d_yhat, d_mu, d_logvar, d_zy, dh, dpre, H, C, Z, fuse ? Wd : nullptr, d1, g_d1, g_zy);
    MVH_LAUNCH_CHECK();
  }
  return MVH_OK;
}

int mvh::latent_bwd_wgrad(hipStream_t st, const float* h, const float* y, const float* drop_u, float p,
                          const float* dpre, float* dWc, float* dbc, float* dWm, float* dbm, float* dWv,
                          float* dbv, int B, int H, int C, int Z) {
  const int no = C + 2 * Z;
  hipLaunchKernelGGL(k_latent_wgrad, dim3(cdiv(C + H + 1, 256), no), dim3(256), 0, st, h, y, drop_u, p, dpre,
                     dWc, dbc, dWm, dbm, dWv, dbv, B, H, C, Z);
  MVH_LAUNCH_CHECK();
  return MVH_OK;
}
