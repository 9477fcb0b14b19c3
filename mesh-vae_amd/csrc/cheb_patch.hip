// Patch ChebConv: the 16 -> 16 convolutions of a 2 049 .. 5 119-vertex level as (mesh, VERTEX PATCH) workgroups with all
// 16 channels on chip, so that the K Cin x Cout contraction of nn/conv.py:559-572 runs on v_mfma_f32_16x16x4_f32.
//
//   out = act( sum_k T_k(L) x W_k + bias ),  T_0 = x, T_1 = L x, T_k = 2 L T_{k-1} - T_{k-2}      (nn/conv.py:557-577)
//
// The slab kernels (cheb_lds.hip) own (mesh, 4 output channels): every contraction then has a 4-wide side (VALU FMAs with
// scalar weights, or dependent v_mfma_f32_4x4x1 chains in the weight gradient) and every workgroup re-reads all input
// channels.  Here a workgroup owns one PATCH of a mesh (meshvae_hip/patches.py: ~1 250 vertices it owns + the rings the
// recurrence needs around them; ring r is only computed up to order K - 1 - r) with ALL channels:
//   * LDS   : T_{k-1} of every local vertex as a 16-float row; row stride 80 B: the 16 B behind the row spread a 16-lane
//             group's float4 gathers over all banks AND hold the vertex's neighbour list (8 LOCAL ids, padded ELL);
//             -2 / deg per vertex behind the rows (a register per tile slot otherwise);
//   * lanes : lane l of a wave = (vertex l & 15 of a 16-vertex tile, channel quad l >> 4): the float4 a lane gathers /
//             holds IS the B operand of v_mfma_f32_16x16x4_f32 (k index = l >> 4) for the four k-steps of 16 channels,
//             and with A = the weight column W_k[4 (l >> 4) + s][l & 15] the product lands as D[cout][vertex]: lane l
//             holds the four output channels 4 (l >> 4) .. + 3 of ITS OWN vertex -- no shuffle, no LDS round trip;
//   * input-side recurrence (no Clenshaw): out accumulates T_k W_k in one accumulator tile per 16 vertices over k.
// L = -D^-1/2 A D^-1/2 on unit weights is applied in scaled variables u = D^-1/2 T (no edge values):
//   u_k = -(2 / deg) sum_{j in N(i)} u_{k-1}[j] - u_{k-2}   (u_1: factor 1),   T_k W_k = D^1/2 (u_k W_k),
// so the D^1/2 is applied once to the accumulated rows.  Isolated vertices (deg = 0): s = 1, no gather.
//
// Backward (one launch for BOTH gradients; nothing of this layer on the weight-gradient lanes):
//   the recurrence runs on dpre = dout * relu'(out);  dX = sum_k T_k(dpre) W_k^T  (L symmetric) accumulates like the
//   forward;  dW_k = x^T T_k(dpre): A = the patch's own x rows (registers, scaled by D^1/2), B = u_k rows read back
//   from LDS as [vertex][co] -- v_mfma_f32_16x16x4_f32 with k = 4 vertices; sums run over the EXCLUSIVE vertices of
//   the patch (every vertex once).  Per (patch, wave, order) partial tiles in the layout of launch_dw_reduce_all.
//   The pooling behind dX (U^T, nn/pool.py:17-20 backward) is formed from LDS for the coarse rows the plan assigns.
#include "common.hpp"

namespace mvh {

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int kRowF = 20;        // floats per LDS row (16 + 4 of padding) = 5 x 16 B, the unit of the plan's ELL ids
constexpr int kRowB = kRowF * 4;

struct PatchDims {
  int B, N, K, P, R, act;
  int n_pool_rows;      // > 0: dx is [B][n_pool_rows][16], formed by the plan's pooling rows
  int src3_n;           // lazy rows: dout rows >= src3_n are g3[v][0..3) w3^T (ConvIO::src3_*), -1 = all rows stored
  int n_part;           // B * P * waves (partial tiles per slab)
  int has_dw, has_dx;
};

__device__ __forceinline__ float4 f4add(const float4& a, const float4& b) {
  return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}

// sum of the 8 neighbour rows' quads: lane_base = LDS byte address of quad q of row 0, ids = 8 x (5 x local id)
__device__ __forceinline__ float4 gather8(const unsigned char* lane_base, const uint4& id) {
  float4 g;
  {
    const float4 n0 = *reinterpret_cast<const float4*>(lane_base + ((id.x & 0xffffu) << 4));
    const float4 n1 = *reinterpret_cast<const float4*>(lane_base + ((id.x >> 16) << 4));
    const float4 n2 = *reinterpret_cast<const float4*>(lane_base + ((id.y & 0xffffu) << 4));
    const float4 n3 = *reinterpret_cast<const float4*>(lane_base + ((id.y >> 16) << 4));
    g = f4add(f4add(n0, n1), f4add(n2, n3));
  }
  asm volatile("" ::: "memory");   // (four rows in flight at a time: 16 registers instead of 32)
  {
    const float4 n4 = *reinterpret_cast<const float4*>(lane_base + ((id.z & 0xffffu) << 4));
    const float4 n5 = *reinterpret_cast<const float4*>(lane_base + ((id.z >> 16) << 4));
    const float4 n6 = *reinterpret_cast<const float4*>(lane_base + ((id.w & 0xffffu) << 4));
    const float4 n7 = *reinterpret_cast<const float4*>(lane_base + ((id.w >> 16) << 4));
    g = f4add(g, f4add(f4add(n4, n5), f4add(n6, n7)));
  }
  asm volatile("" ::: "memory");
  return g;
}

// acc (D[c_out quad][vertex]) += W-column registers (A, k-steps 0..3) x the lane's float4 (B)
__device__ __forceinline__ void mfma4(v4f& acc, const float (&wa)[4], const float4& t) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[0], t.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[1], t.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[2], t.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[3], t.w, acc, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------------------ forward
// SLOTS >= tiles of the largest patch / waves; ASLOTS >= tiles of its exclusive vertices / waves
template <int THREADS, int SLOTS, int ASLOTS>
__global__ void __launch_bounds__(THREADS)
k_patch_fwd(const float* __restrict__ p_x, const float* __restrict__ p_W, const float* __restrict__ p_bias,
            float* __restrict__ p_out, uint8_t* __restrict__ p_bits, const int32_t* __restrict__ p_poff,
            const int32_t* __restrict__ p_cnt, const uint32_t* __restrict__ p_pinfo, const uint32_t* __restrict__ p_ell,
            PatchDims a) {
  constexpr int NW = THREADS / 64;
  extern __shared__ __align__(16) unsigned char smem[];
  // blocks b and b + 8 share an XCD: the patches of one mesh (which share their halo rows) stay on one L2 (speed only)
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int mesh = (jj / a.P) * 8 + xcd, pt = jj % a.P;
  if (mesh >= a.B) return;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);   // (w: wave-uniform)
  const int vi = lane & 15, q = lane >> 4;
  const int o = p_poff[pt], rows16 = p_poff[pt + 1] - o;
  const int* __restrict__ c = p_cnt + pt * (a.R + 2);
  const int K = a.K;
  float* u = reinterpret_cast<float*>(smem);                                            // [rows16 + 1][kRowF]
  float* coefv = reinterpret_cast<float*>(smem + (size_t)(rows16 + 1) * kRowB);          // [rows16]
  const unsigned char* lane_base = smem + 16 * q;
  MVH_STAMPX(0);

  for (int i = tid; i < rows16; i += THREADS) {
    *reinterpret_cast<uint4*>(smem + (size_t)i * kRowB + 64) = reinterpret_cast<const uint4*>(p_ell)[o + i];
    const float deg = (float)((p_pinfo[o + i] >> 16) & 255u);
    coefv[i] = deg > 0.f ? -2.0f * __builtin_amdgcn_rcpf(deg) : 0.f;
  }
  if (tid < kRowF / 4) reinterpret_cast<float4*>(u + (size_t)rows16 * kRowF)[tid] = make_float4(0.f, 0.f, 0.f, 0.f);

  const int n_excl = c[0];
  const int nt_all = rows16 >> 4;
  const int nt_out = (n_excl + 15) >> 4;
  const int nt0 = (c[1 + min(a.R, K - 1)] + 15) >> 4;     // tiles whose u_0 somebody needs

  // st[s]: between two orders u_{k-2} of the slot's own row quad; inside order k, from its gather on, u_k (the swap
  // behind the barrier trades it for the row's u_{k-1}, which the next order subtracts)
  float4 st[SLOTS];
  v4f acc[ASLOTS];
  float wa[4];
  auto load_w = [&](int k) {
#pragma unroll
    for (int s = 0; s < 4; ++s) wa[s] = p_W[k * 256 + (4 * q + s) * 16 + vi];
  };
  load_w(0);
  const float* xb = p_x + (long long)mesh * a.N * 16;
#pragma unroll
  for (int s = 0; s < ASLOTS; ++s) acc[s] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    const int t = s * NW + w;
    st[s] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int v = 16 * t + vi;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t < nt0) {
      const uint32_t info = p_pinfo[o + v];
      const float deg = (float)((info >> 16) & 255u);
      const bool valid = (info >> 24 & 15u) != 15u;
      const float sc = valid ? (deg > 0.f ? __builtin_amdgcn_rsqf(deg) : 1.0f) : 0.f;
      const float4 xv = *reinterpret_cast<const float4*>(xb + (long long)(info & 0xffffu) * 16 + 4 * q);
      r = make_float4(xv.x * sc, xv.y * sc, xv.z * sc, xv.w * sc);
    }
    if (t < nt_all) *reinterpret_cast<float4*>(u + (size_t)v * kRowF + 4 * q) = r;
    if (s < ASLOTS) mfma4(acc[s < ASLOTS ? s : 0], wa, r);     // (outside every run-time branch)
  }
  MVH_STAMPX(1);
  __syncthreads();
  MVH_STAMPX(2);

  for (int k = 1; k < K; ++k) {
    const int ntk = (c[1 + min(a.R, K - 1 - k)] + 15) >> 4;
    const float sc = (k == 1) ? 0.5f : 1.0f;
    load_w(k);
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const int t = s * NW + w;
      if (t < ntk) {
        const int v = 16 * t + vi;
        const uint4 id = *reinterpret_cast<const uint4*>(smem + (size_t)v * kRowB + 64);
        const float cc = coefv[v] * sc;
        const float4 g = gather8(lane_base, id);
        st[s] = make_float4(fmaf(cc, g.x, -st[s].x), fmaf(cc, g.y, -st[s].y), fmaf(cc, g.z, -st[s].z),
                            fmaf(cc, g.w, -st[s].w));
        if (s < ASLOTS) mfma4(acc[s < ASLOTS ? s : 0], wa, st[s]);   // (tiles past the last output tile: unused columns)
      }
      __builtin_amdgcn_sched_barrier(0);   // one slot's gathers in flight at a time (registers)
    }
    if (k + 1 == K) break;      // (the rows of the last order feed nobody)
    MVH_STAMPX(3 + 3 * (k - 1));
    __syncthreads();            // every gather of u_{k-1} is done
    MVH_STAMPX(4 + 3 * (k - 1));
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const int t = s * NW + w;
      if (t < ntk) {
        float4* own = reinterpret_cast<float4*>(u + (size_t)(16 * t + vi) * kRowF + 4 * q);
        const float4 old = *own;
        *own = st[s];
        st[s] = old;
      }
    }
    __syncthreads();
    MVH_STAMPX(5 + 3 * (k - 1));
  }
  MVH_STAMPX(26);

  // ---- epilogue: D^1/2, bias, activation, one 16-byte store per lane (+ one sign byte)
  float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p_bias) b4 = *reinterpret_cast<const float4*>(p_bias + 4 * q);
#pragma unroll
  for (int s = 0; s < ASLOTS; ++s) {
    const int t = s * NW + w;
    const int v = 16 * t + vi;
    if (t < nt_out && v < n_excl) {
      const uint32_t info = p_pinfo[o + v];
      const float deg = (float)((info >> 16) & 255u);
      const float is = deg > 0.f ? __builtin_sqrtf(deg) : 1.0f;
      float r0 = fmaf(acc[s][0], is, b4.x), r1 = fmaf(acc[s][1], is, b4.y), r2 = fmaf(acc[s][2], is, b4.z),
            r3 = fmaf(acc[s][3], is, b4.w);
      if (a.act == MVH_ACT_RELU) { r0 = fmaxf(r0, 0.f); r1 = fmaxf(r1, 0.f); r2 = fmaxf(r2, 0.f); r3 = fmaxf(r3, 0.f); }
      const long long row = (long long)mesh * a.N + (info & 0xffffu);
      *reinterpret_cast<float4*>(p_out + row * 16 + 4 * q) = make_float4(r0, r1, r2, r3);
      if (p_bits)
        p_bits[row * 4 + q] = (uint8_t)((r0 > 0.f ? 1 : 0) | (r1 > 0.f ? 2 : 0) | (r2 > 0.f ? 4 : 0) | (r3 > 0.f ? 8 : 0));
    }
  }
  MVH_STAMPX(27);
}

// ----------------------------------------------------------------------------------------------------------- backward
// GSLOTS >= 4-vertex groups of the largest exclusive set / waves.  The weight gradient of order k - 1 rides in order k's
// slot loop: slot s issues the x loads of its chunk of the wave's groups, gathers, and then feeds the matrix pipe with
// that chunk (A = D^1/2 x from L2, B = u_{k-1} rows from LDS), so the loads' latency sits under a slot's gathers and the
// matrix instructions of both gradients are spread between the LDS bursts.
// SU: the first SU tile slots of every wave are core tiles in every patch (needed at every order): no run-time test
// around them, i.e. no control flow between the x prefetch and its use (the waitcnt pass then counts exactly).
template <int THREADS, int SLOTS, int ASLOTS, int GSLOTS, int SU>
__global__ void __launch_bounds__(THREADS)
k_patch_bwd(const float* __restrict__ p_dout, const uint8_t* __restrict__ p_mbits, const float* __restrict__ p_g3,
            const float* __restrict__ p_w3, const float* __restrict__ p_x, const float* __restrict__ p_W,
            float* __restrict__ p_dx, float* __restrict__ p_part, const int32_t* __restrict__ p_poff,
            const int32_t* __restrict__ p_cnt, const uint32_t* __restrict__ p_pinfo, const uint32_t* __restrict__ p_ell,
            const int32_t* __restrict__ p_prow_off, const int32_t* __restrict__ p_prow_gid,
            const int32_t* __restrict__ p_prow_ptr, const int32_t* __restrict__ p_pcol, const float* __restrict__ p_pval,
            PatchDims a) {
  constexpr int NW = THREADS / 64;
  constexpr int CH = (GSLOTS + SLOTS - 1) / SLOTS;     // groups of a wave per slot
  extern __shared__ __align__(16) unsigned char smem[];
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int mesh = (jj / a.P) * 8 + xcd, pt = jj % a.P;
  if (mesh >= a.B) return;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);   // (w: wave-uniform)
  const int vi = lane & 15, q = lane >> 4;
  const int o = p_poff[pt], rows16 = p_poff[pt + 1] - o;
  const int* __restrict__ c = p_cnt + pt * (a.R + 2);
  const int K = a.K;
  const int n_excl = c[0], n_core = c[1];
  const int nt_all = rows16 >> 4;
  const int nt_out = a.has_dx ? (n_core + 15) >> 4 : 0;
  const int nt0 = (c[1 + min(a.R, K - 1)] + 15) >> 4;
  const int ng = a.has_dw ? (n_excl + 3) >> 2 : 0;          // 4-vertex groups of the weight gradient
  float* u = reinterpret_cast<float*>(smem);
  float* coefv = reinterpret_cast<float*>(smem + (size_t)(rows16 + 1) * kRowB);   // -2 / deg per local vertex
  float* wpart = coefv + rows16;                   // [NW][256] the waves' dW tiles of one order, [NW][16] their db sums
  float* wdb = wpart + NW * 256;
  uint32_t* xinfo = reinterpret_cast<uint32_t*>(wdb + NW * 16);   // [4 NW SLOTS CH] global id | max(deg, 1) << 16; 0 off the exclusive set
  const unsigned char* lane_base = smem + 16 * q;
  MVH_STAMPX(0);

  for (int i = tid; i < rows16; i += THREADS) {
    *reinterpret_cast<uint4*>(smem + (size_t)i * kRowB + 64) = reinterpret_cast<const uint4*>(p_ell)[o + i];
    const uint32_t info = p_pinfo[o + i];
    const uint32_t dg = (info >> 16) & 255u;
    coefv[i] = dg > 0u ? -2.0f * __builtin_amdgcn_rcpf((float)dg) : 0.f;
    if (i < 4 * NW * SLOTS * CH) xinfo[i] = ((info >> 28) & 1u) && a.has_dw ? ((info & 0xffffu) | (max(dg, 1u) << 16)) : 0u;
  }
  for (int i = rows16 + tid; i < 4 * NW * SLOTS * CH; i += THREADS) xinfo[i] = 0u;   // (every group slot of every wave exists)
  if (tid < kRowF / 4) reinterpret_cast<float4*>(u + (size_t)rows16 * kRowF)[tid] = make_float4(0.f, 0.f, 0.f, 0.f);

  float4 st[SLOTS];     // (see k_patch_fwd)
  v4f acc[ASLOTS];
  float wa[4];
  auto load_w = [&](int k) {   // A = W_k^T: [c_in = vi][c_out = 4 q + s]
    const float4 t = *reinterpret_cast<const float4*>(p_W + k * 256 + vi * 16 + 4 * q);
    wa[0] = t.x; wa[1] = t.y; wa[2] = t.z; wa[3] = t.w;
  };
  load_w(0);
  const long long mrow = (long long)mesh * a.N;
  const int wg = mesh * a.P + pt;          // this workgroup's partial tile (one per slab)
  const int tile = (K + 1) * 64;
#pragma unroll
  for (int s = 0; s < ASLOTS; ++s) acc[s] = (v4f){0.f, 0.f, 0.f, 0.f};
  {
    // lazy rows: this lane's four columns of W3 [16][3]
    float w3r[4][3];
    if (a.src3_n >= 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < 3; ++t) w3r[i][t] = p_w3[(4 * q + i) * 3 + t];
    }
    float4 dbacc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const int t = s * NW + w;
      st[s] = make_float4(0.f, 0.f, 0.f, 0.f);
      const int v = 16 * t + vi;
      float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
      if (t < nt0) {
        const uint32_t info = p_pinfo[o + v];
        const float deg = (float)((info >> 16) & 255u);
        const bool valid = (info >> 24 & 15u) != 15u;
        const float sc = valid ? (deg > 0.f ? __builtin_amdgcn_rsqf(deg) : 1.0f) : 0.f;
        const int gid = (int)(info & 0xffffu);
        float4 dv;
        if (a.src3_n >= 0 && gid >= a.src3_n) {
          const float* gr = p_g3 + (mrow + gid) * 3;
          const float g0 = gr[0], g1 = gr[1], g2 = gr[2];
          dv.x = fmaf(g2, w3r[0][2], fmaf(g1, w3r[0][1], g0 * w3r[0][0]));
          dv.y = fmaf(g2, w3r[1][2], fmaf(g1, w3r[1][1], g0 * w3r[1][0]));
          dv.z = fmaf(g2, w3r[2][2], fmaf(g1, w3r[2][1], g0 * w3r[2][0]));
          dv.w = fmaf(g2, w3r[3][2], fmaf(g1, w3r[3][1], g0 * w3r[3][0]));
        } else {
          dv = *reinterpret_cast<const float4*>(p_dout + (mrow + gid) * 16 + 4 * q);
        }
        if (p_mbits) {
          const uint32_t m = p_mbits[(mrow + gid) * 4 + q];
          dv.x = (m & 1u) ? dv.x : 0.f;
          dv.y = (m & 2u) ? dv.y : 0.f;
          dv.z = (m & 4u) ? dv.z : 0.f;
          dv.w = (m & 8u) ? dv.w : 0.f;
        }
        if (valid && v < n_excl) dbacc = f4add(dbacc, dv);
        r = make_float4(dv.x * sc, dv.y * sc, dv.z * sc, dv.w * sc);
      }
      if (t < nt_all) *reinterpret_cast<float4*>(u + (size_t)v * kRowF + 4 * q) = r;
      if (s < ASLOTS) mfma4(acc[s < ASLOTS ? s : 0], wa, r);     // (outside every run-time branch)
    }
    if (a.has_dw) {   // db: this wave's sums over its exclusive vertices
      float4 d = dbacc;
#pragma unroll
      for (int m = 1; m < 16; m <<= 1) {
        d.x += __shfl_xor(d.x, m, 64);
        d.y += __shfl_xor(d.y, m, 64);
        d.z += __shfl_xor(d.z, m, 64);
        d.w += __shfl_xor(d.w, m, 64);
      }
      if (vi == 0) *reinterpret_cast<float4*>(wdb + w * 16 + 4 * q) = d;
    }
  }
  MVH_STAMPX(1);
  __syncthreads();      // rows, lists, coefv, xinfo, wdb staged
  MVH_STAMPX(2);
  if (a.has_dw && tid < 16) {   // db partial of the workgroup: entries (order K, q = c_out, j = 0) of slab 0
    float d = 0.f;
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) d += wdb[ww * 16 + tid];
    p_part[(long long)wg * tile + (K * 16 + tid) * 4] = d;
  }
  // ---- weight gradient: wave w owns the groups g = w + NW i; chunk s = its groups i in [s CH, (s + 1) CH)
  const float* xlane = p_x + mrow * 16 + vi;     // A operand: [c_in = vi][vertex 4 g + q]
  constexpr int PD = 2;                          // chunks whose x loads are in flight ahead of the one being consumed
  uint32_t xi[PD + 1][CH];
  float xv[PD + 1][CH];
  v4f t0, t1;
  // (addresses: lane base + compile-time offsets -- no clamped indices, which the compiler would hoist out of the order
  //  loop as one register per group and spill; a group past the wave's last one reads LDS it does not use, weight 0)
  const uint32_t* xinfo_l = xinfo + 4 * w + q;
  const float* ub_l = u + (size_t)(4 * w + q) * kRowF + vi;
  auto dw_issue = [&](int s) {      // (branch-free: a slot past the exclusive set reads row 0 with weight 0)
    if (s >= SLOTS) return;
    uint32_t inf[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) inf[i] = xinfo_l[4 * NW * (s * CH + i)];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      xi[s % (PD + 1)][i] = inf[i];
      xv[s % (PD + 1)][i] = xlane[(long long)(inf[i] & 0xffffu) * 16];
    }
  };
  auto dw_consume = [&](int s) {      // (no run-time branch around a matrix instruction: the compiler spills the tiles there)
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const bool on = w + NW * (s * CH + i) < ng;
      const float is = __builtin_sqrtf((float)(xi[s % (PD + 1)][i] >> 16));   // D^1/2 (0 off the exclusive set / past the end)
      float b = ub_l[(size_t)4 * NW * (s * CH + i) * kRowF];
      b = on ? b : 0.f;
      if (i & 1) t1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[s % (PD + 1)][i] * is, b, t1, 0, 0, 0);
      else t0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[s % (PD + 1)][i] * is, b, t0, 0, 0, 0);
    }
  };
  auto dw_store = [&]() {   // this wave's tile of the order -> wpart[w]
    *reinterpret_cast<float4*>(wpart + w * 256 + lane * 4) =
        make_float4(t0[0] + t1[0], t0[1] + t1[1], t0[2] + t1[2], t0[3] + t1[3]);
  };
  // behind a barrier: the waves' tiles summed in wave order -> the workgroup's partial tile of order k
  auto dw_flush = [&](int k) {
    if (!a.has_dw || tid >= 256) return;
    float d = 0.f;
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) d += wpart[ww * 256 + tid];
    // element tid = (lane l = tid >> 2, register r = tid & 3) of the D tile: c_in = 4 (l >> 4) + r, c_out = l & 15
    const int l = tid >> 2, r = tid & 3;
    p_part[((long long)(l >> 4) * a.n_part + wg) * tile + (k * 16 + (l & 15)) * 4 + r] = d;
  };

  for (int k = 1; k < K; ++k) {
    const int ntk = (c[1 + min(a.R, K - 1 - k)] + 15) >> 4;
    const float sc = (k == 1) ? 0.5f : 1.0f;
    load_w(k);
    t0 = (v4f){0.f, 0.f, 0.f, 0.f};
    t1 = (v4f){0.f, 0.f, 0.f, 0.f};
    if (a.has_dw) {
#pragma unroll
      for (int s = 0; s < PD; ++s) dw_issue(s);
    }
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const int t = s * NW + w;
      if (a.has_dw) dw_issue(s + PD);
      __builtin_amdgcn_sched_barrier(0);   // (the scheduler otherwise sinks these loads down to their use, two slots on)
      if (s < SU || t < ntk) {
        const int v = 16 * t + vi;
        const uint4 id = *reinterpret_cast<const uint4*>(smem + (size_t)v * kRowB + 64);
        const float cc = coefv[v] * sc;
        const float4 g = gather8(lane_base, id);
        st[s] = make_float4(fmaf(cc, g.x, -st[s].x), fmaf(cc, g.y, -st[s].y), fmaf(cc, g.z, -st[s].z),
                            fmaf(cc, g.w, -st[s].w));
        if (s < ASLOTS) mfma4(acc[s < ASLOTS ? s : 0], wa, st[s]);   // (tiles past the last output tile: unused columns)
      }
      if (a.has_dw) dw_consume(s);        // dW_{k-1}: the rows in LDS are still u_{k-1}
      __builtin_amdgcn_sched_barrier(0);   // one slot's gathers in flight at a time (registers)
    }
    if (a.has_dw) dw_store();
    MVH_STAMPX(3 + 3 * (k - 1));
    __syncthreads();
    MVH_STAMPX(4 + 3 * (k - 1));
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const int t = s * NW + w;
      if (t < ntk) {
        float4* own = reinterpret_cast<float4*>(u + (size_t)(16 * t + vi) * kRowF + 4 * q);
        const float4 old = *own;
        *own = st[s];
        st[s] = old;
      }
    }
    dw_flush(k - 1);
    __syncthreads();
    MVH_STAMPX(5 + 3 * (k - 1));
  }
  if (a.has_dw) {   // dW_{K-1}: the rows are u_{K-1} now
    t0 = (v4f){0.f, 0.f, 0.f, 0.f};
    t1 = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < PD; ++s) dw_issue(s);
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      dw_issue(s + PD);
      __builtin_amdgcn_sched_barrier(0);
      dw_consume(s);
      __builtin_amdgcn_sched_barrier(0);
    }
    dw_store();
  }
  __syncthreads();      // (also: the last dW pass has read its rows)
  dw_flush(K - 1);
  MVH_STAMPX(26);
  if (!a.has_dx) return;

  // ---- dX rows (D^1/2 applied): straight to memory, or through LDS into the rows of the pooling operator
  if (a.n_pool_rows <= 0) {
#pragma unroll
    for (int s = 0; s < ASLOTS; ++s) {
      const int t = s * NW + w;
      const int v = 16 * t + vi;
      if (t < nt_out && v < n_excl) {
        const uint32_t info = p_pinfo[o + v];
        const float deg = (float)((info >> 16) & 255u);
        const float is = deg > 0.f ? __builtin_sqrtf(deg) : 1.0f;
        *reinterpret_cast<float4*>(p_dx + (mrow + (info & 0xffffu)) * 16 + 4 * q) =
            make_float4(acc[s][0] * is, acc[s][1] * is, acc[s][2] * is, acc[s][3] * is);
      }
    }
    return;
  }
#pragma unroll
  for (int s = 0; s < ASLOTS; ++s) {
    const int t = s * NW + w;
    const int v = 16 * t + vi;
    if (t < nt_out) {
      const uint32_t info = p_pinfo[o + v];
      const float deg = (float)((info >> 16) & 255u);
      const float is = deg > 0.f ? __builtin_sqrtf(deg) : 1.0f;
      *reinterpret_cast<float4*>(u + (size_t)v * kRowF + 4 * q) =
          make_float4(acc[s][0] * is, acc[s][1] * is, acc[s][2] * is, acc[s][3] * is);
    }
  }
  __syncthreads();
  MVH_STAMPX(27);
  const int r0 = p_prow_off[pt], nrow = p_prow_off[pt + 1] - r0;
  const int* __restrict__ rp = p_prow_ptr + r0 + pt;
  for (int it = tid; it < nrow * 4; it += THREADS) {
    const int i = it >> 2, qq = it & 3;
    const int e0 = rp[i], e1 = rp[i + 1];
    float4 sacc = make_float4(0.f, 0.f, 0.f, 0.f);
    // four taps per round (loads issued together); taps past the row end: a valid entry with weight 0, so the sums
    // stay in the operator's entry order (the arithmetic of k_spmm<.., EXACT>)
    for (int e = e0; e < e1; e += 4) {
      float wv[4];
      int cc[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int ee = min(e + t, e1 - 1);
        cc[t] = p_pcol[ee];
        wv[t] = (e + t < e1) ? p_pval[ee] : 0.f;
      }
      float4 n[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) n[t] = *reinterpret_cast<const float4*>(u + (size_t)cc[t] * kRowF + 4 * qq);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        sacc.x = __fadd_rn(sacc.x, __fmul_rn(wv[t], n[t].x));
        sacc.y = __fadd_rn(sacc.y, __fmul_rn(wv[t], n[t].y));
        sacc.z = __fadd_rn(sacc.z, __fmul_rn(wv[t], n[t].z));
        sacc.w = __fadd_rn(sacc.w, __fmul_rn(wv[t], n[t].w));
      }
    }
    *reinterpret_cast<float4*>(p_dx + ((long long)mesh * a.n_pool_rows + p_prow_gid[r0 + i]) * 16 + 4 * qq) = sacc;
  }
  MVH_STAMPX(28);
}

#ifdef MVH_STAMP
MVH_STAMP_READER(mvh_debug_read_stamps_patch)
#endif

// ------------------------------------------------------------------------------------------------------------- host
// rows + lists, -2 / deg, and (backward) the waves' dW tiles / db sums of one order
static size_t patch_lds_bytes(const mvh_patch_plan_t* pl, int bwd_waves = 16) {
  const size_t fwd = (size_t)(pl->max_rows + 1) * kRowB + (size_t)pl->max_rows * 4;
  if (bwd_waves <= 0) return fwd;
  return fwd + (size_t)bwd_waves * (256 + 16) * 4 + (size_t)1344 * 4;    // (4 NW SLOTS CH group slots, every block size)
}

// register-array sizes per block size (whole tiles / groups per wave of the largest patch the LDS admits: 106 tiles)
template <int THREADS> struct PatchCfg;
template <> struct PatchCfg<1024> { static constexpr int S = 7, AF = 6, AB = 6, G = 21; };
template <> struct PatchCfg<768> { static constexpr int S = 9, AF = 8, AB = 7, G = 27; };
template <> struct PatchCfg<512> { static constexpr int S = 14, AF = 11, AB = 11, G = 42; };

static int fwd_threads() { const int t = dbg().patch_fwd_threads; return (t == 512 || t == 768 || t == 1024) ? t : 1024; }
static int bwd_threads() { const int t = dbg().patch_bwd_threads; return (t == 512 || t == 768 || t == 1024) ? t : 512; }

template <int THREADS>
static bool cfg_fits(const mvh_patch_plan_t* pl) {
  constexpr int NW = THREADS / 64;
  using C = PatchCfg<THREADS>;
  return cdiv(pl->max_rows / 16, NW) <= C::S && cdiv(cdiv(pl->max_excl, 16), NW) <= C::AF &&
         cdiv(cdiv(pl->max_core, 16), NW) <= C::AB && cdiv(cdiv(pl->max_excl, 4), NW) <= C::G;
}
static bool cfg_fits_rt(const mvh_patch_plan_t* pl, int threads) {
  return threads == 1024 ? cfg_fits<1024>(pl) : threads == 768 ? cfg_fits<768>(pl) : cfg_fits<512>(pl);
}

bool patch_eligible(const mvh_csr_t* lap, int N, int Cin, int Cout, int K) {
  if (dbg().no_patch || dbg().force_generic) return false;
  const mvh_patch_plan_t* pl = lap ? lap->patch : nullptr;
  if (!pl || Cin != 16 || Cout != 16 || K < 1 || K > 12) return false;
  const int need = MVH_CSR_NORMALIZED_LAPLACIAN | MVH_CSR_SYMMETRIC;
  if ((lap->flags & need) != need) return false;
  if (pl->n_vertices != N || pl->n_patches < 1 || K - 1 > pl->n_rings) return false;
  if (patch_lds_bytes(pl, bwd_threads() / 64) > 160 * 1024 || pl->max_rows % 16 != 0 || pl->max_rows < 16) return false;
  return cfg_fits_rt(pl, fwd_threads()) && cfg_fits_rt(pl, bwd_threads());
}

size_t patch_part_bytes(const mvh_csr_t* lap, int B, int K) {
  const mvh_patch_plan_t* pl = lap ? lap->patch : nullptr;
  if (!pl) return 0;
  return (size_t)4 * B * pl->n_patches * (K + 1) * 64 * sizeof(float) + 256;
}

template <int THREADS>
static int launch_fwd_t(hipStream_t st, const mvh_patch_plan_t* pl, const float* x, const float* W, const float* bias,
                        float* out, uint8_t* bits, const PatchDims& d) {
  using C = PatchCfg<THREADS>;
  auto kern = k_patch_fwd<THREADS, C::S, C::AF>;
  const size_t lds = patch_lds_bytes(pl, 0);
  static LdsAttr attr;
  if (int rc = attr.ensure(reinterpret_cast<const void*>(kern), lds)) return rc;
  const int grid = ((d.B + 7) / 8) * 8 * d.P;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(THREADS), lds, st, x, W, bias, out, bits, pl->poff, pl->cnt, pl->pinfo,
                     pl->ell, d);
  MVH_LAUNCH_CHECK();
  return MVH_OK;
}

int launch_patch_fwd(hipStream_t st, const mvh_csr_t* lap, const float* x, const float* W, const float* bias, float* out,
                     uint8_t* bits, int B, int N, int K, int act) {
  const mvh_patch_plan_t* pl = lap->patch;
  MVH_REQUIRE((((uintptr_t)x | (uintptr_t)out | (uintptr_t)bias) & 15) == 0, "patch_fwd: tensors must be 16-byte aligned");
  PatchDims d{};
  d.B = B; d.N = N; d.K = K; d.P = pl->n_patches; d.R = pl->n_rings; d.act = act;
  const int th = fwd_threads();
  MVH_REQUIRE(cfg_fits_rt(pl, th), "patch_fwd: the plan does not fit the %d-thread kernel", th);
  if (th == 1024) return launch_fwd_t<1024>(st, pl, x, W, bias, out, bits, d);
  if (th == 768) return launch_fwd_t<768>(st, pl, x, W, bias, out, bits, d);
  return launch_fwd_t<512>(st, pl, x, W, bias, out, bits, d);
}

template <int THREADS, int SU>
static int launch_bwd_t(hipStream_t st, const mvh_patch_plan_t* pl, const float* dout, const uint8_t* mbits,
                        const float* g3, const float* w3, const float* x, const float* W, float* dx, float* part,
                        PatchDims d) {
  using C = PatchCfg<THREADS>;
  auto kern = k_patch_bwd<THREADS, C::S, C::AB, C::G, SU>;
  const size_t lds = patch_lds_bytes(pl, THREADS / 64);
  static LdsAttr attr;
  if (int rc = attr.ensure(reinterpret_cast<const void*>(kern), lds)) return rc;
  const int grid = ((d.B + 7) / 8) * 8 * d.P;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(THREADS), lds, st, dout, mbits, g3, w3, x, W, dx, part, pl->poff,
                     pl->cnt, pl->pinfo, pl->ell, pl->prow_off, pl->prow_gid, pl->prow_ptr, pl->pcol, pl->pval, d);
  MVH_LAUNCH_CHECK();
  return MVH_OK;
}

// dX (+ its pooling when pooled) and the dW / db partial tiles of a 16 -> 16 layer in one launch.  dW / db themselves
// come out of the reduction `defer` describes (launch_dw_reduce_all; the caller runs it, at once or deferred).
int launch_patch_bwd(hipStream_t st, const mvh_csr_t* lap, const float* x, const float* W, const float* dout,
                     const uint8_t* mbits, const float* g3, const float* w3, int src3_n, float* dx, bool pooled,
                     float* part, size_t part_bytes, DwReduceEntry* defer, float* dW, float* db, int B, int N, int K) {
  const mvh_patch_plan_t* pl = lap->patch;
  MVH_REQUIRE((((uintptr_t)x | (uintptr_t)dout | (uintptr_t)dx | (uintptr_t)W | (uintptr_t)part) & 15) == 0,
              "patch_bwd: tensors must be 16-byte aligned");
  MVH_REQUIRE(!pooled || pl->n_pool_rows > 0, "patch_bwd: the plan carries no pooling rows");
  MVH_REQUIRE(dx || dW, "patch_bwd: nothing to compute");
  const int th = bwd_threads();
  MVH_REQUIRE(cfg_fits_rt(pl, th), "patch_bwd: the plan does not fit the %d-thread kernel", th);
  PatchDims d{};
  d.B = B; d.N = N; d.K = K; d.P = pl->n_patches; d.R = pl->n_rings; d.act = 0;
  d.n_pool_rows = pooled ? pl->n_pool_rows : 0;
  d.src3_n = g3 ? src3_n : -1;
  d.n_part = B * pl->n_patches;
  d.has_dw = dW ? 1 : 0; d.has_dx = dx ? 1 : 0;
  if (dW) {
    MVH_REQUIRE(x && part && defer && part_bytes >= patch_part_bytes(lap, B, K), "patch_bwd: partial-tile buffer too small");
    *defer = DwReduceEntry{};
    defer->part = part; defer->n_part = d.n_part; defer->NS = 4; defer->K = K; defer->CQ = 16; defer->CP = 16;
    defer->p_is_x = 1; defer->Cin = 16; defer->Cout = 16; defer->db_mode = db ? 1 : 0; defer->dW = dW; defer->db = db;
    defer->S = nullptr;
  }
  // slots that are core tiles for every wave of every patch (>= min_core / 16 / waves, rounded down)
  const int su = (pl->min_core / 16) / (th / 64);
  if (th == 1024) return su >= 5 ? launch_bwd_t<1024, 5>(st, pl, dout, mbits, g3, w3, x, W, dx, part, d)
                                 : launch_bwd_t<1024, 0>(st, pl, dout, mbits, g3, w3, x, W, dx, part, d);
  if (th == 768) return su >= 6 ? launch_bwd_t<768, 6>(st, pl, dout, mbits, g3, w3, x, W, dx, part, d)
                                : launch_bwd_t<768, 0>(st, pl, dout, mbits, g3, w3, x, W, dx, part, d);
  return su >= 10 ? launch_bwd_t<512, 10>(st, pl, dout, mbits, g3, w3, x, W, dx, part, d)
                  : launch_bwd_t<512, 0>(st, pl, dout, mbits, g3, w3, x, W, dx, part, d);
}

}  // namespace mvh
