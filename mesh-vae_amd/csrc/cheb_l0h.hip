// Level-0 ChebConv forward / dX for bf16-STORED activations (BASELINE configs[1] "bf16"): the 16 -> 16 layer at the
// 4998-vertex level, one workgroup per (mesh, slab of 4 output channels), 1024 threads x 5 vertices.
//
// Same algorithm as k_cheb_lds (cheb_lds.hip: Clenshaw on the output side in scaled variables u = D^-1/2 b, unweighted
// gathers, fp32 recurrence state in LDS), re-laid out around what bf16 storage frees:
//   * the thread's 5 input rows stay in VGPRs as PACKED bf16 (8 registers per vertex instead of 16), so that
//   * the neighbour ids of its vertices fit in VGPRs too (20 registers) and the LDS holds TWO fp32 slabs (u_{k+1}
//     gathered by everyone, u_{k+2} touched only through the thread's own rows: ONE barrier per order, no ELL image,
//     no staging pass, no register copy of u_{k+2}) -- the layout the small levels already use;
//   * the weight contraction runs on the matrix pipe from the packed registers as they are:
//     v_mfma_f32_4x4x4_16B_bf16, A = W_k[c0..c0+3][s0 + (lane & 3)] (bf16 copy of the fp32 master weights, packed per
//     step by k_pack_all), B = the lane's own four channels c0..c0+3: block b's column j -- the four output channels
//     of the vertex of lane 4 b + j -- lands in that lane's accumulator (fp32), four instructions per vertex and
//     order instead of 64 v_fma + 16 unpacks, and the VALU keeps only the gather adds.
// Precision: inputs are bf16 by definition of the storage mode; weights are rounded to bf16 for the products (exact in
// fp32: 8 x 8 significant bits), every sum is fp32, the Chebyshev state never leaves fp32.  tests/test_gpu_bf16.py
// holds it to the same bars as the unpack-and-v_fma form (cheb_lds.hip with in_bf16), which stays the fallback.
#include "common.hpp"
#include "bf16.hpp"

namespace mvh {

struct L0hDims {
  int B, N, K, CO, act, in_bs, out_bs, mask_bs, pooled_bs, pt_rows;
  int out_bf16, pooled_bf16;
};

typedef float v4f_h __attribute__((ext_vector_type(4)));
typedef short v4s_h __attribute__((ext_vector_type(4)));

constexpr int kL0hThreads = 1024, kL0hVpt = 5, kL0hSlots = kL0hThreads * kL0hVpt;

template <bool BWD>
__global__ void __launch_bounds__(kL0hThreads)
k_cheb_l0h(const uint16_t* __restrict__ p_in, const uint8_t* __restrict__ p_mask_bits,
           const uint32_t* __restrict__ p_wpk, const float* __restrict__ p_bias, float* __restrict__ p_out,
           uint8_t* __restrict__ p_bits_out, const uint32_t* __restrict__ p_rowinfo, const uint32_t* __restrict__ p_ell,
           const int* __restrict__ p_pt_rowptr, const int* __restrict__ p_pt_col, const float* __restrict__ p_pt_val,
           float* __restrict__ p_pooled, L0hDims a) {
  constexpr int CQ = 16, VPT = kL0hVpt, THREADS = kL0hThreads, VS = kL0hSlots;
  extern __shared__ __align__(16) unsigned char smem[];
  float4* slabA = reinterpret_cast<float4*>(smem);  // [VS]
  float4* slabB = slabA + VS;                        // [VS]

  const int NS = a.CO >> 2;
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;  // blocks b and b + 8 share an XCD: a mesh's slabs share one L2
  const int mesh = (jj / NS) * 8 + xcd, s0 = (jj % NS) * 4;
  if (mesh >= a.B) return;
  const int tid = threadIdx.x, N = a.N;

  // ---- own vertices: neighbour ids, -2/deg, s = deg^-1/2, the packed input rows (masked by the ReLU signs in dX)
  uint4 ids[VPT];
  uint32_t xp[VPT][CQ / 2];
  float ka2[VPT], sdeg[VPT];
  const uint16_t* inh = p_in + (long long)mesh * a.in_bs * CQ;
  const uint32_t* mbits = reinterpret_cast<const uint32_t*>(p_mask_bits + (long long)mesh * a.mask_bs * (CQ / 4));
  const unsigned padi = (unsigned)N | ((unsigned)N << 16);
#pragma unroll
  for (int vi = 0; vi < VPT; ++vi) {
    const int v = tid + vi * THREADS;
    const bool valid = v < N;
    const int vl = min(v, N - 1);
    const float deg = valid ? (float)(p_rowinfo[vl] & 255u) : 0.f;
    ka2[vi] = deg > 0.f ? -2.0f * __builtin_amdgcn_rcpf(deg) : 0.f;
    sdeg[vi] = valid ? (deg > 0.f ? __builtin_amdgcn_rsqf(deg) : 1.0f) : 0.f;
    ids[vi] = valid ? reinterpret_cast<const uint4*>(p_ell)[vl] : make_uint4(padi, padi, padi, padi);
    const uint4 r0 = *reinterpret_cast<const uint4*>(inh + (long long)vl * CQ);
    const uint4 r1 = *reinterpret_cast<const uint4*>(inh + (long long)vl * CQ + 8);
    uint32_t w[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
    if (BWD && p_mask_bits) {  // bit j of byte c/4 = out[v][c + j] > 0: keep those halves, clear the others
      const uint32_t m = mbits[vl];
#pragma unroll
      for (int d = 0; d < 8; ++d) {
        const uint32_t b2 = (m >> (8 * (d >> 1) + 2 * (d & 1))) & 3u;  // channels 2d (low half), 2d + 1 (high half)
        w[d] &= ((b2 & 1u) ? 0x0000ffffu : 0u) | ((b2 & 2u) ? 0xffff0000u : 0u);
      }
    }
#pragma unroll
    for (int d = 0; d < 8; ++d) xp[vi][d] = valid ? w[d] : 0u;
    slabB[v] = make_float4(0.f, 0.f, 0.f, 0.f);  // u_K = 0
  }

  // bf16 weight slab [slab][k][channel group][out channel i][2 dwords]: lane l reads the 8 bytes of i = l & 3
  const uint2* wl = reinterpret_cast<const uint2*>(p_wpk) + (long long)(s0 >> 2) * a.K * 16 + (tid & 3);
  uint2 wk[4];
  auto load_w = [&](int k) {
#pragma unroll
    for (int cg = 0; cg < 4; ++cg) wk[cg] = wl[(k * 4 + cg) * 4];
  };
  // (In W_k)[own vertex][slab] on the matrix pipe, unscaled
  auto contract = [&](int vi) -> float4 {
    v4f_h t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int cg = 0; cg < 4; ++cg) {
      const v4s_h wa = __builtin_bit_cast(v4s_h, wk[cg]);
      const v4s_h xb = __builtin_bit_cast(v4s_h, make_uint2(xp[vi][2 * cg], xp[vi][2 * cg + 1]));
      t = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wa, xb, t, 0, 0, 0);
    }
    return make_float4(t[0], t[1], t[2], t[3]);
  };
  auto gather = [&](int vi, const float4* slab) -> float4 {
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    const uint4 id = ids[vi];
    {
      const float4 n0 = slab[id.x & 0xffffu], n1 = slab[id.x >> 16], n2 = slab[id.y & 0xffffu], n3 = slab[id.y >> 16];
      g.x += n0.x; g.y += n0.y; g.z += n0.z; g.w += n0.w;
      g.x += n1.x; g.y += n1.y; g.z += n1.z; g.w += n1.w;
      g.x += n2.x; g.y += n2.y; g.z += n2.z; g.w += n2.w;
      g.x += n3.x; g.y += n3.y; g.z += n3.z; g.w += n3.w;
    }
    asm volatile("" ::: "memory");
    {
      const float4 n0 = slab[id.z & 0xffffu], n1 = slab[id.z >> 16], n2 = slab[id.w & 0xffffu], n3 = slab[id.w >> 16];
      g.x += n0.x; g.y += n0.y; g.z += n0.z; g.w += n0.w;
      g.x += n1.x; g.y += n1.y; g.z += n1.z; g.w += n1.w;
      g.x += n2.x; g.y += n2.y; g.z += n2.z; g.w += n2.w;
      g.x += n3.x; g.y += n3.y; g.z += n3.z; g.w += n3.w;
    }
    return g;
  };

  // (results are parked in the thread's own rows of the slab that nobody gathers from, not in registers: the
  //  kernel sits at the 128-VGPR budget of four waves per SIMD)
  float4* cur = slabA;   // u_{k+1}, gathered by everyone
  float4* oth = slabB;   // u_{k+2}, own rows only; receives u_k
  load_w(a.K - 1);
#pragma unroll
  for (int vi = 0; vi < VPT; ++vi) {
    const float4 c = contract(vi);
    const float s = sdeg[vi];
    slabA[tid + vi * THREADS] = make_float4(c.x * s, c.y * s, c.z * s, c.w * s);  // u_{K-1} (zero in the slots past N)
  }
  float4* res = slabA;   // where the thread's own result rows are at the end
  if (a.K >= 2) {
    load_w(a.K - 2);
    __syncthreads();
    for (int k = a.K - 2; k >= 0; --k) {
      const float sc = (k == 0) ? 0.5f : 1.0f;
#pragma unroll
      for (int vi = 0; vi < VPT; ++vi) {
        const int v = tid + vi * THREADS;
        const float4 o = oth[v];
        const float4 c = contract(vi);
        const float4 g = gather(vi, cur);
        const float kk = ka2[vi] * sc, s = sdeg[vi];
        oth[v] = make_float4(fmaf(kk, g.x, fmaf(s, c.x, -o.x)), fmaf(kk, g.y, fmaf(s, c.y, -o.y)),   // own row only:
                             fmaf(kk, g.z, fmaf(s, c.z, -o.z)), fmaf(kk, g.w, fmaf(s, c.w, -o.w)));  // no barrier before
      }
      res = oth;
      if (k == 0) break;  // `oth` holds the result rows; nobody gathers from it
      load_w(k - 1);      // (its latency sits under the barrier)
      __syncthreads();
      float4* t = cur;
      cur = oth;
      oth = t;
    }
  }
  float4* stage = res;  // pooled epilogue: the activated rows replace the thread's own result rows there

  // ---- epilogue: unscale (1/s = sqrt(deg)), bias, activation, sign bytes, one bf16 (or fp32) store per vertex
  float bj[4] = {0.f, 0.f, 0.f, 0.f};
  if (!BWD && p_bias) {
#pragma unroll
    for (int j = 0; j < 4; ++j) bj[j] = p_bias[s0 + j];
  }
  const bool scatter = p_pt_rowptr != nullptr;
#pragma unroll
  for (int vi = 0; vi < VPT; ++vi) {
    const int v = tid + vi * THREADS;
    if (v >= N) continue;
    const float inv_s = ka2[vi] < 0.f ? __builtin_amdgcn_rsqf(-0.5f * ka2[vi]) : 1.0f;
    const float4 Rv = res[v];
    float o[4] = {fmaf(Rv.x, inv_s, bj[0]), fmaf(Rv.y, inv_s, bj[1]), fmaf(Rv.z, inv_s, bj[2]), fmaf(Rv.w, inv_s, bj[3])};
    if (!BWD && a.act == MVH_ACT_RELU) {
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = fmaxf(o[j], 0.f);
    }
    if (!BWD && p_bits_out)
      p_bits_out[((long long)mesh * a.out_bs + v) * NS + (s0 >> 2)] =
          (uint8_t)((o[0] > 0.f ? 1 : 0) | (o[1] > 0.f ? 2 : 0) | (o[2] > 0.f ? 4 : 0) | (o[3] > 0.f ? 8 : 0));
    if (scatter) stage[v] = make_float4(o[0], o[1], o[2], o[3]);
    if (!(BWD && scatter))  // (dX with a fused U^T stores only the pooled rows)
      store4_any(p_out, ((long long)mesh * a.out_bs + v) * a.CO + s0, a.out_bf16 != 0, o[0], o[1], o[2], o[3]);
  }
  if (scatter) {  // pooled rows gathered from LDS in the operator's CSR order (the arithmetic of k_spmm<.., EXACT>)
    __syncthreads();
    for (int c = tid; c < a.pt_rows; c += THREADS) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      const int e0 = p_pt_rowptr[c], e1 = p_pt_rowptr[c + 1];
      for (int e = e0; e < e1; e += 4) {
        float w[4];
        int cc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int ee = min(e + t, e1 - 1);
          cc[t] = p_pt_col[ee];
          w[t] = (e + t < e1) ? p_pt_val[ee] : 0.f;
        }
        float4 n[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) n[t] = stage[cc[t]];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          acc.x = __fadd_rn(acc.x, __fmul_rn(w[t], n[t].x));
          acc.y = __fadd_rn(acc.y, __fmul_rn(w[t], n[t].y));
          acc.z = __fadd_rn(acc.z, __fmul_rn(w[t], n[t].z));
          acc.w = __fadd_rn(acc.w, __fmul_rn(w[t], n[t].w));
        }
      }
      if (BWD) store4_any(p_out, ((long long)mesh * a.out_bs + c) * a.CO + s0, a.out_bf16 != 0, acc.x, acc.y, acc.z, acc.w);
      else store4_any(p_pooled, ((long long)mesh * a.pooled_bs + c) * a.CO + s0, a.pooled_bf16 != 0, acc.x, acc.y, acc.z, acc.w);
    }
  }
}

// Wh[slab][k][cg][i][d] (one dword = two bf16): the four input channels c0 + 0..3 (c0 = 4 cg) of output channel
// 4 slab + i, low half first -- W[k][c][o] forward, W[k][o][c] backward (W^T); 16 channels on both sides.
__global__ void __launch_bounds__(256) k_pack_l0h(const float* __restrict__ W, uint32_t* __restrict__ Wh, int K, int bwd) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 4 * K * 32) return;
  Wh[i] = pack_l0h_dword(W, K, bwd, i);
}

int l0h_pack_dwords(int K) { return l0h_pack_dwords_hd(K); }

int launch_pack_l0h(hipStream_t st, const float* W, uint32_t* Wh, int K, bool bwd) {
  hipLaunchKernelGGL(k_pack_l0h, dim3(cdiv(l0h_pack_dwords(K), 256)), dim3(256), 0, st, W, Wh, K, bwd ? 1 : 0);
  MVH_LAUNCH_CHECK();
  return MVH_OK;
}

// Eligibility + launch; *handled == false -> the caller keeps the general bf16 form of cheb_lds.hip.
int try_cheb_l0h(hipStream_t st, const mvh_csr_t* lap, const float* in, const uint8_t* mask_bits, const float* W,
                 const float* bias, float* out, int B, int N, int Cin, int Cout, int K, int act, bool bwd, void* wpack,
                 bool* handled, const LdsConvOpts& o) {
  *handled = false;
  if (dbg().force_generic || dbg().no_l0h) return MVH_OK;
  const int need = MVH_CSR_NORMALIZED_LAPLACIAN | MVH_CSR_SYMMETRIC;
  if (!lap->rowinfo || !lap->ell || lap->ell_pairs <= 0 || lap->ell_pairs > 4 || (lap->flags & need) != need) return MVH_OK;
  if (lap->flags & MVH_CSR_ELL_OVERFLOW) return MVH_OK;
  if (Cin != 16 || Cout != 16 || K < 1 || N + 1 > kL0hSlots || N + 1 <= 2048) return MVH_OK;   // the 5k level only
  if (!o.in_bf16 || o.in_map || o.pool_inv || (o.in_bs > 0 && o.in_bs != N)) return MVH_OK;
  if (bwd && mask_bits == nullptr && act == MVH_ACT_RELU) return MVH_OK;
  if (!wpack && !o.prepacked_h) return MVH_OK;
  if (((uintptr_t)in | (uintptr_t)out | (uintptr_t)o.pooled) % 16 != 0) return MVH_OK;
  L0hDims d;
  d.B = B; d.N = N; d.K = K; d.CO = 16; d.act = act;
  d.in_bs = N; d.out_bs = o.out_bs > 0 ? o.out_bs : N; d.mask_bs = o.mask_bs > 0 ? o.mask_bs : N;
  d.pooled_bs = 0; d.pt_rows = 0;
  d.out_bf16 = o.out_bf16 ? 1 : 0; d.pooled_bf16 = o.pooled_bf16 ? 1 : 0;
  const int *pt_rowptr = nullptr, *pt_col = nullptr;
  const float* pt_val = nullptr;
  float* pooled = nullptr;
  if (o.out_pool_t) {
    const mvh_csr_t* pt = o.out_pool_t;
    if (pt->n_cols != N || !pt->rowptr || !pt->col || !pt->val) return MVH_OK;
    if (!bwd && !o.pooled) return MVH_OK;
    pt_rowptr = pt->rowptr; pt_col = pt->col; pt_val = pt->val; d.pt_rows = pt->n_rows;
    if (bwd) d.out_bs = pt->n_rows;
    else { pooled = o.pooled; d.pooled_bs = pt->n_rows; }
  }
  if (o.dry_run) {
    *handled = true;
    return MVH_OK;
  }
  const uint32_t* wh = o.prepacked_h;
  if (!wh) {
    if (int rc = launch_pack_l0h(st, W, (uint32_t*)wpack, K, bwd)) return rc;
    wh = (const uint32_t*)wpack;
  }
  const size_t lds = (size_t)kL0hSlots * 32;
  static LdsAttr attr_set[2];
  const int grid = ((B + 7) / 8) * 8 * 4;
#define MVH_L0H(BW)                                                                                                        \
  do {                                                                                                                     \
    auto kern = k_cheb_l0h<BW>;                                                                                            \
    if (int rc = attr_set[BW ? 1 : 0].ensure(reinterpret_cast<const void*>(kern), lds)) return rc;                         \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kL0hThreads), lds, st, reinterpret_cast<const uint16_t*>(in), mask_bits, wh, \
                       bias, out, o.bits_out, lap->rowinfo, lap->ell, pt_rowptr, pt_col, pt_val, pooled, d);               \
  } while (0)
  if (bwd) MVH_L0H(true);
  else MVH_L0H(false);
#undef MVH_L0H
  MVH_LAUNCH_CHECK();
  *handled = true;
  return MVH_OK;
}

}  // namespace mvh
