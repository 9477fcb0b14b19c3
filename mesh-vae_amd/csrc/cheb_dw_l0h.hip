// Level-0 ChebConv weight gradient for bf16-STORED activations (BASELINE configs[1] "bf16"): the 16 -> 16 layer at the
// 4998-vertex level,  dW_k[ci][co] = sum_{b,v} T_k(L) x [b,v,ci] * dpre[b,v,co],  db[co] = sum dpre.
//
// Same decomposition as k_cheb_dw_lds (cheb_dw_lds.hip): a workgroup owns (mesh, slab of 4 channels of the side P = x
// that runs the Chebyshev recurrence, fp32 in LDS, scaled variables), the other side Q = dpre stays in registers, and
// per (mesh, slab, wave, order) a 16 x 4 partial tile goes to the workspace that k_dw_reduce(_all) sums in fixed order.
// What bf16 storage changes (1024 threads x 5 vertices, as cheb_l0h.hip):
//   * the thread keeps the dpre rows of its OWN 5 vertices as packed bf16 (8 registers per vertex) -- the fp32 form
//     needed 160 registers per lane for the same rows in a 4-lanes-per-vertex layout;
//   * neighbour ids in VGPRs, TWO fp32 slabs in LDS: one barrier per order, and the contraction is fused into the
//     recurrence loop (T_k of the thread's own vertex is in registers when it is produced: no second pass over LDS);
//   * the contraction over vertices runs on v_mfma_f32_4x4x4_16B_bf16: a block = 4 lanes = 4 vertices is the
//     reduction index, A = dpre[v_0..3][4 g + i] (4 instructions for the 16 channels), B = T_k[v_0..3][j]; both
//     operands come from "one row per lane" by a 4 x 4 transpose of 16-bit values inside the quad (two DPP
//     quad-permutes and a few bit selects; the dpre side once per launch, the T_k side once per vertex and order).
//     20 matrix instructions per wave and order instead of 160 fp32 ones.
// Precision: dpre is bf16 by definition of the storage mode; T_k is rounded to bf16 for the products (fp32 sums).
#include "common.hpp"
#include "bf16.hpp"

namespace mvh {

struct DwL0hDims {
  int B, N, K, bs, db_mode, has_bits;
  int mesh0;   // first mesh of this launch (ConvIO::dw_split)
};

typedef float v4f_d __attribute__((ext_vector_type(4)));
typedef short v4s_d __attribute__((ext_vector_type(4)));

constexpr int kDwhThreads = 1024, kDwhVpt = 5, kDwhSlots = kDwhThreads * kDwhVpt;

template <int CTRL>
__device__ __forceinline__ uint32_t quad_perm(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}

// 4 x 4 transpose of 16-bit values inside each quad of lanes: lane i holds row i = (e0, e1 | e2, e3) in (d0 | d1)
// and receives column i = (row_0[i], row_1[i] | row_2[i], row_3[i]).
__device__ __forceinline__ void quad_transpose16(uint32_t& d0, uint32_t& d1, int qi) {
  const uint32_t send = (qi < 2) ? d1 : d0;                   // off-diagonal 2 x 2 blocks change lane pairs
  const uint32_t recv = quad_perm<0x4E>(send);                // quad_perm [2,3,0,1]
  if (qi < 2) d1 = recv;
  else d0 = recv;
  const uint32_t p0 = quad_perm<0xB1>(d0), p1 = quad_perm<0xB1>(d1);   // quad_perm [1,0,3,2]
  if (qi & 1) {
    d0 = (p0 >> 16) | (d0 & 0xffff0000u);
    d1 = (p1 >> 16) | (d1 & 0xffff0000u);
  } else {
    d0 = (d0 & 0x0000ffffu) | (p0 << 16);
    d1 = (d1 & 0x0000ffffu) | (p1 << 16);
  }
}

__device__ __forceinline__ float sum_quads(float x) {  // over the 16 lanes that share (lane & 3)
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x124, 0xf, 0xf, false));  // row_ror 4
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x128, 0xf, 0xf, false));  // row_ror 8
  x += __shfl_xor(x, 16, 64);
  x += __shfl_xor(x, 32, 64);
  return x;
}

// part layout (shared with cheb_dw_lds.hip): [slab][mesh][wave][K+1][16][4]; plane K = bias gradient at [q][0]
__global__ void __launch_bounds__(kDwhThreads)
k_cheb_dw_l0h(const uint16_t* __restrict__ p_x, const uint16_t* __restrict__ p_dout, const uint8_t* __restrict__ p_bits,
              const uint32_t* __restrict__ p_rowinfo, const uint32_t* __restrict__ p_ell, float* __restrict__ p_part,
              DwL0hDims a) {
  constexpr int C = 16, VPT = kDwhVpt, THREADS = kDwhThreads, VS = kDwhSlots, NW = THREADS / 64;
  extern __shared__ __align__(16) unsigned char smem[];
  float4* slabA = reinterpret_cast<float4*>(smem);
  float4* slabB = slabA + VS;
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int mesh = a.mesh0 + (jj >> 2) * 8 + xcd, sl = jj & 3, s0 = sl * 4;
  if (mesh >= a.B) return;
  const int tid = threadIdx.x, N = a.N, lane = tid & 63, wave = tid >> 6, qi = lane & 3;

  uint4 ids[VPT];
  uint32_t qa[VPT][8];     // A operands: [vertex step][group g][2 dwords], already quad-transposed
  float ka2[VPT], invs[VPT];
  const uint16_t* xh = p_x + (long long)mesh * a.bs * C;
  const uint16_t* dh = p_dout + (long long)mesh * a.bs * C;
  const uint32_t* mb = reinterpret_cast<const uint32_t*>(p_bits + (long long)mesh * a.bs * (C / 4));
  const unsigned padi = (unsigned)N | ((unsigned)N << 16);
  float4* cur = slabA;   // t~_{k-1}
  float4* oth = slabB;   // t~_{k-2} -> t~_k (own rows only)
#pragma unroll
  for (int vi = 0; vi < VPT; ++vi) {
    const int v = tid + vi * THREADS;
    const bool valid = v < N;
    const int vl = min(v, N - 1);
    const float deg = valid ? (float)(p_rowinfo[vl] & 255u) : 0.f;
    ka2[vi] = deg > 0.f ? -2.0f * __builtin_amdgcn_rcpf(deg) : 0.f;
    const float s = valid ? (deg > 0.f ? __builtin_amdgcn_rsqf(deg) : 1.0f) : 0.f;
    invs[vi] = valid ? (deg > 0.f ? __builtin_amdgcn_sqrtf(deg) : 1.0f) : 0.f;
    ids[vi] = valid ? reinterpret_cast<const uint4*>(p_ell)[vl] : make_uint4(padi, padi, padi, padi);
    // Q = dpre row of the own vertex, masked by the ReLU sign bytes of the forward
    const uint4 r0 = *reinterpret_cast<const uint4*>(dh + (long long)vl * C);
    const uint4 r1 = *reinterpret_cast<const uint4*>(dh + (long long)vl * C + 8);
    uint32_t w[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
    if (a.has_bits) {
      const uint32_t m = mb[vl];
#pragma unroll
      for (int d = 0; d < 8; ++d) {
        const uint32_t b2 = (m >> (8 * (d >> 1) + 2 * (d & 1))) & 3u;
        w[d] &= ((b2 & 1u) ? 0x0000ffffu : 0u) | ((b2 & 2u) ? 0xffff0000u : 0u);
      }
    }
#pragma unroll
    for (int d = 0; d < 8; ++d) w[d] = valid ? w[d] : 0u;
#pragma unroll
    for (int g = 0; g < 4; ++g) {  // lane i of the quad ends up with dpre[v_0..3][4 g + i]
      uint32_t d0 = w[2 * g], d1 = w[2 * g + 1];
      quad_transpose16(d0, d1, qi);
      qa[vi][2 * g] = d0;
      qa[vi][2 * g + 1] = d1;
    }
    // P = x slab of the own vertex: t~_0 = s x
    const float4 xv = bf16_unpack4(*reinterpret_cast<const uint2*>(xh + (long long)vl * C + s0));
    slabA[v] = valid ? make_float4(xv.x * s, xv.y * s, xv.z * s, xv.w * s) : make_float4(0.f, 0.f, 0.f, 0.f);
    slabB[v] = make_float4(0.f, 0.f, 0.f, 0.f);   // t~_{-1} = 0
  }
  float* part = p_part + (((long long)sl * a.B + mesh) * NW + wave) * (long long)(a.K + 1) * C * 4;
  __syncthreads();

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
  // acc[g][r] at lane (block b, j) += sum over the block's 4 vertices of dpre[v][4 g + r] * T_k[v][j]
  auto accumulate = [&](v4f_d(&acc)[4], int vi, const float4& tt) {   // tt = t~_k of the own vertex (scaled)
    const float is = invs[vi];
    uint32_t d0 = bf16_pack2(tt.x * is, tt.y * is), d1 = bf16_pack2(tt.z * is, tt.w * is);
    quad_transpose16(d0, d1, qi);                               // lane j of the quad: T_k[v_0..3][j]
    const v4s_d tb = __builtin_bit_cast(v4s_d, make_uint2(d0, d1));
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const v4s_d qv = __builtin_bit_cast(v4s_d, make_uint2(qa[vi][2 * g], qa[vi][2 * g + 1]));
      acc[g] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(qv, tb, acc[g], 0, 0, 0);
    }
  };
  auto flush = [&](v4f_d(&acc)[4], int k) {   // fold the 16 blocks; lanes 0..3 (j = lane) write the 16 x 4 tile
    float red[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[g][r] = sum_quads(acc[g][r]);
    if (lane < 4) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[((long long)k * C + 4 * g + r) * 4 + lane] = red[g][r];
    }
  };

  if (a.db_mode == 1 && sl == 0) {  // bias gradient = column sums of dpre: the same products against a column of ones
    v4f_d acc[4];                   // (plane K of the tile set; the reduce kernel reads entry [q][0])
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = (v4f_d){0.f, 0.f, 0.f, 0.f};
    const v4s_d ones = __builtin_bit_cast(v4s_d, make_uint2(0x3f803f80u, 0x3f803f80u));
#pragma unroll
    for (int vi = 0; vi < VPT; ++vi)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const v4s_d qv = __builtin_bit_cast(v4s_d, make_uint2(qa[vi][2 * g], qa[vi][2 * g + 1]));
        acc[g] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(qv, ones, acc[g], 0, 0, 0);
      }
    flush(acc, a.K);
  }
  {  // order 0: T_0 = x
    v4f_d acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = (v4f_d){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int vi = 0; vi < VPT; ++vi) accumulate(acc, vi, slabA[tid + vi * THREADS]);
    flush(acc, 0);
  }
  for (int k = 1; k < a.K; ++k) {
    const float sc = (k == 1) ? 0.5f : 1.0f;   // T_1 = L T_0 ; T_k = 2 L T_{k-1} - T_{k-2}
    v4f_d acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = (v4f_d){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int vi = 0; vi < VPT; ++vi) {
      const int v = tid + vi * THREADS;
      const float4 o = oth[v];
      const float4 g = gather(vi, cur);
      const float kk = ka2[vi] * sc;
      const float4 tt = make_float4(fmaf(kk, g.x, -o.x), fmaf(kk, g.y, -o.y), fmaf(kk, g.z, -o.z), fmaf(kk, g.w, -o.w));
      oth[v] = tt;                     // own row only: no barrier before
      accumulate(acc, vi, tt);
    }
    flush(acc, k);
    __syncthreads();                   // t~_k complete, every gather of t~_{k-1} done
    float4* t = cur;
    cur = oth;
    oth = t;
  }
}

size_t cheb_dw_l0h_ws_bytes(int B, int K) { return (size_t)B * 4 * (kDwhThreads / 64) * (size_t)(K + 1) * 16 * 4 * sizeof(float); }

// dW (+ db) of the 5k level's 16 -> 16 layer on bf16 rows; *handled == false -> the caller keeps cheb_dw_lds.hip.
int try_cheb_dw_l0h(hipStream_t st, const mvh_csr_t* lap, const float* x, const float* dout, const uint8_t* out_bits,
                    float* dW, float* db, int B, int N, int Cin, int Cout, int K, float* part, size_t part_bytes,
                    bool* handled, bool dry_run, DwReduceEntry* defer, int dw_split) {
  *handled = false;
  if (dbg().force_generic || dbg().no_l0h) return MVH_OK;
  const int need = MVH_CSR_NORMALIZED_LAPLACIAN | MVH_CSR_SYMMETRIC;
  if (!lap->rowinfo || !lap->ell || lap->ell_pairs <= 0 || lap->ell_pairs > 4 || (lap->flags & need) != need) return MVH_OK;
  if (lap->flags & MVH_CSR_ELL_OVERFLOW) return MVH_OK;
  if (Cin != 16 || Cout != 16 || K < 1 || N + 1 > kDwhSlots || N + 1 <= 2048 || B < 1) return MVH_OK;
  if (((uintptr_t)x | (uintptr_t)dout) % 16 != 0) return MVH_OK;
  if (!part || part_bytes < cheb_dw_l0h_ws_bytes(B, K)) return MVH_OK;
  if (dry_run) {
    *handled = true;
    return MVH_OK;
  }
  const int NW = kDwhThreads / 64;
  DwL0hDims d{B, N, K, N, db ? 1 : 0, out_bits ? 1 : 0, 0};
  const size_t lds = (size_t)kDwhSlots * 32;
  static LdsAttr attr;
  if (int rc = attr.ensure(reinterpret_cast<const void*>(k_cheb_dw_l0h), lds)) return rc;
  // the batch in dw_split launches of whole 8-mesh groups, one behind the other on this stream (ConvIO::dw_split)
  const int split = dw_split > 1 ? dw_split : 1;
  const int per = ((((B + split - 1) / split) + 7) / 8) * 8;
  for (int m0 = 0; m0 < B; m0 += per) {
    d.mesh0 = m0;
    const int nb = B - m0 < per ? B - m0 : per;
    const int grid = ((nb + 7) / 8) * 8 * 4;
    hipLaunchKernelGGL(k_cheb_dw_l0h, dim3(grid), dim3(kDwhThreads), lds, st, reinterpret_cast<const uint16_t*>(x),
                       reinterpret_cast<const uint16_t*>(dout), out_bits, lap->rowinfo, lap->ell, part, d);
    MVH_LAUNCH_CHECK();
  }
  // tile entry (q, j): q = dpre channel (co), j = x channel 4 slab + j (ci): P is x
  const DwReduceEntry ent{part, B * NW, 4, K, 16, 16, 1, Cin, Cout, db ? 1 : 0, dW, db};
  if (defer) {
    *defer = ent;
  } else {
    DwReduceTable t;
    t.n = 1;
    t.e[0] = ent;
    if (int rc = launch_dw_reduce_all(st, t)) return rc;
  }
  *handled = true;
  return MVH_OK;
}

}  // namespace mvh
