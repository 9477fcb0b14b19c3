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
//   * lanes : lane l of a wave = (vertex l & 15 of a 16-vertex tile, channel quad l >> 4): the float4 a lane reads of
//             its own vertex IS the B operand of v_mfma_f32_16x16x4_f32 (k index = l >> 4) for the four k-steps of 16
//             channels, and with A = the weight column W_k[4 (l >> 4) + s][l & 15] the product lands as D[cout][vertex]:
//             lane l holds the four output channels 4 (l >> 4) .. + 3 of ITS OWN vertex -- no shuffle;
//   * input-side recurrence (no Clenshaw): out accumulates T_k W_k in one accumulator tile per 16 vertices over k;
//   * forward: every wave gathers AND feeds the matrix pipe with the rows it has just formed (registers);
//   * backward, WAVE ROLES: the first NWR waves of a workgroup run the recurrence (LDS gathers + VALU, no accumulators:
//             registers for two tiles' gathers in flight), the other NWD waves do all the matrix work one order behind,
//             reading u_{k-1} rows from LDS while the recurrence waves gather them for u_k -- the two instruction streams
//             share each SIMD (matrix pipe beside VALU / LDS), no global load and no run-time branch sits between matrix
//             instructions, and the weight gradient's x rows stay in the matrix waves' registers.  (The forward in this
//             form: 44 against 35 us -- its matrix work is a quarter of the backward's and the split costs gather waves.)
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
#include "bf16.hpp"

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
  int pool_lds;         // the pooling entries of every patch fit the LDS behind its core rows
  int x_bs;             // rows per mesh of the layer input x (N, or the mesh stride of a strided view in rows)
  int cin;              // k_patch_enc0: input channels (<= 4)
  int out_bf16;         // k_patch_enc0: the pooled output rows are stored as bf16 (bf16.hpp); k_patch_bwd: dx / its pooled rows
  int x_bf16, dout_bf16;  // k_patch_bwd on bf16 STORAGE: x and the stored dout rows are bf16 tensors (fp32 arithmetic throughout)
  int u_rows;           // k_patch_fwd, > 0: x is the COARSE tensor [B][u_rows][16], un-pooled through the plan's urec while loading
  int map_c, map_n0;    // k_patch_fwd, map_c > 0: rows >= map_n0 of out3 [B][N][map_c] (map_c <= 4) = this layer's output x w3 [16][map_c]
};

// float4 sums / fused multiply-adds as TWO packed instructions (v_pk_add_f32 / v_pk_fma_f32: two fp32 lanes per issue slot,
// the same IEEE operations per component) -- the gather loops are bound by vector-instruction issue, and the build runs
// without the SLP vectoriser that would pair the scalar forms
typedef float f32x2_v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 f4add(const float4& a, const float4& b) {
#ifdef MVH_NO_PK
  return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
#endif
  const f32x2_v lo = (f32x2_v){a.x, a.y} + (f32x2_v){b.x, b.y}, hi = (f32x2_v){a.z, a.w} + (f32x2_v){b.z, b.w};
  return make_float4(lo[0], lo[1], hi[0], hi[1]);
}
// c g - s per component
__device__ __forceinline__ float4 f4fms(float c, const float4& g, const float4& s) {
#ifdef MVH_NO_PK
  return make_float4(fmaf(c, g.x, -s.x), fmaf(c, g.y, -s.y), fmaf(c, g.z, -s.z), fmaf(c, g.w, -s.w));
#endif
  const f32x2_v cc = {c, c};
  const f32x2_v lo = __builtin_elementwise_fma(cc, (f32x2_v){g.x, g.y}, -(f32x2_v){s.x, s.y});
  const f32x2_v hi = __builtin_elementwise_fma(cc, (f32x2_v){g.z, g.w}, -(f32x2_v){s.z, s.w});
  return make_float4(lo[0], lo[1], hi[0], hi[1]);
}

// LDS byte offset of quad q (qoff = 16 q) of the row whose id (5 x local id, one 16-bit half of w) the list holds:
// id * 16 + qoff in ONE instruction (the compiler's and / shift / add forms take 2-3; the gather loops are bound by
// vector-instruction issue)
__device__ __forceinline__ unsigned addr_lo(unsigned w, unsigned qoff) {
  unsigned r;
  asm("v_mad_u32_u16 %0, %1, 16, %2 op_sel:[0,0,0,0]" : "=v"(r) : "v"(w), "v"(qoff));
  return r;
}
__device__ __forceinline__ unsigned addr_hi(unsigned w, unsigned qoff) {
  unsigned r;
  asm("v_mad_u32_u16 %0, %1, 16, %2 op_sel:[1,0,0,0]" : "=v"(r) : "v"(w), "v"(qoff));
  return r;
}
__device__ __forceinline__ float4 ldq(const unsigned char* smem0, unsigned off) {
  return *reinterpret_cast<const float4*>(smem0 + off);
}

// sum of the neighbour rows' quads: smem0 = the LDS base, qoff = 16 q, ids = 8 x (5 x local id), padw = the pad id in both
// halves.  A list of <= 6 neighbours ends in two pads (plan: SHORT_DEG): when all 16 lists of the tile do -- 4 of 5 tiles
// on the 5k template -- the last two gathers are skipped (wave-uniform test on the word the lane already holds).
__device__ __forceinline__ float4 gather8(const unsigned char* smem0, unsigned qoff, const uint4& id, unsigned padw) {
  float4 g;
  {
    const float4 n0 = ldq(smem0, addr_lo(id.x, qoff)), n1 = ldq(smem0, addr_hi(id.x, qoff));
    const float4 n2 = ldq(smem0, addr_lo(id.y, qoff)), n3 = ldq(smem0, addr_hi(id.y, qoff));
    g = f4add(f4add(n0, n1), f4add(n2, n3));
  }
  asm volatile("" ::: "memory");   // (four rows in flight at a time: 16 registers instead of 32)
  {
    const float4 n4 = ldq(smem0, addr_lo(id.z, qoff)), n5 = ldq(smem0, addr_hi(id.z, qoff));
    g = f4add(g, f4add(n4, n5));
  }
  if (__builtin_amdgcn_ballot_w64(id.w != padw) != 0ull) {
    const float4 n6 = ldq(smem0, addr_lo(id.w, qoff)), n7 = ldq(smem0, addr_hi(id.w, qoff));
    g = f4add(g, f4add(n6, n7));
  }
  asm volatile("" ::: "memory");
  return g;
}

// the same for TWO tiles, software-pipelined by hand: rows of one tile are summed while rows of the other are in flight
// (one tile alone is three dependent LDS round trips with nothing of the wave in between)
__device__ __forceinline__ void gather8x2(const unsigned char* smem0, unsigned qoff, const uint4& ia, const uint4& ib,
                                          unsigned padw, float4& ga, float4& gb) {
  const float4 a0 = ldq(smem0, addr_lo(ia.x, qoff)), a1 = ldq(smem0, addr_hi(ia.x, qoff));
  const float4 a2 = ldq(smem0, addr_lo(ia.y, qoff)), a3 = ldq(smem0, addr_hi(ia.y, qoff));
  const float4 b0 = ldq(smem0, addr_lo(ib.x, qoff)), b1 = ldq(smem0, addr_hi(ib.x, qoff));
  const float4 b2 = ldq(smem0, addr_lo(ib.y, qoff)), b3 = ldq(smem0, addr_hi(ib.y, qoff));
  asm volatile("" ::: "memory");
  ga = f4add(f4add(a0, a1), f4add(a2, a3));
  const float4 a4 = ldq(smem0, addr_lo(ia.z, qoff)), a5 = ldq(smem0, addr_hi(ia.z, qoff));
  asm volatile("" ::: "memory");
  gb = f4add(f4add(b0, b1), f4add(b2, b3));
  const float4 b4 = ldq(smem0, addr_lo(ib.z, qoff)), b5 = ldq(smem0, addr_hi(ib.z, qoff));
  asm volatile("" ::: "memory");
  ga = f4add(ga, f4add(a4, a5));
  gb = f4add(gb, f4add(b4, b5));
  if (__builtin_amdgcn_ballot_w64(ia.w != padw || ib.w != padw) != 0ull) {     // (either tile has a long list: both pay)
    const float4 a6 = ldq(smem0, addr_lo(ia.w, qoff)), a7 = ldq(smem0, addr_hi(ia.w, qoff));
    const float4 b6 = ldq(smem0, addr_lo(ib.w, qoff)), b7 = ldq(smem0, addr_hi(ib.w, qoff));
    ga = f4add(ga, f4add(a6, a7));
    gb = f4add(gb, f4add(b6, b7));
  }
  asm volatile("" ::: "memory");
}

// pin a value where it is computed (without a consumer in the gather phase the compiler sinks the sums of a tile down to
// the swap behind the barrier and keeps -- spills -- the 32 gathered registers per tile until then)
__device__ __forceinline__ void pin(float4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }

// acc (D[c_out quad][vertex]) += W-column registers (A, k-steps 0..3) x the lane's float4 (B)
__device__ __forceinline__ void mfma4(v4f& acc, const float (&wa)[4], const float4& t) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[0], t.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[1], t.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[2], t.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[3], t.w, acc, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------------------ forward
// SLOTS >= tiles of the largest patch / waves; ASLOTS >= tiles of its exclusive vertices / waves.
// SU: the first SU tile slots of every wave are core tiles in every patch, i.e. needed at every order: no run-time test
// around them, so that they form one basic block and the scheduler overlaps the LDS round trips of neighbouring slots.
template <int THREADS, int SLOTS, int ASLOTS, int SU>
__global__ void __launch_bounds__(THREADS)
k_patch_fwd(const float* __restrict__ p_x, const int32_t* __restrict__ p_xmap, const float* __restrict__ p_W,
            const float* __restrict__ p_bias, float* __restrict__ p_out, uint8_t* __restrict__ p_bits,
            const int32_t* __restrict__ p_poff,
            const int32_t* __restrict__ p_cnt, const uint32_t* __restrict__ p_pinfo, const uint32_t* __restrict__ p_ell,
            const uint32_t* __restrict__ p_urec, float* __restrict__ p_xstore, const float* __restrict__ p_w3,
            float* __restrict__ p_out3, PatchDims a) {
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
  const unsigned padw = (unsigned)(rows16 * 5) * 0x10001u;      // the list pad (the zero row), both halves of a word
  float* u = reinterpret_cast<float*>(smem);                                            // [rows16 + 1][kRowF]
  float* coefv = reinterpret_cast<float*>(smem + (size_t)(rows16 + 1) * kRowB);          // [rows16]
  // this lane's row in tile slot s: byte offset (16 (s NW + w) + vi) 80 -- up to 133 KB, beyond the 16-bit offset field of
  // the DS instructions: three bases 48 KB apart that the compiler cannot fold (else it keeps one address register per
  // slot, hoisted out of the order loop: 7 .. 14 registers, spilled)
  int kb1 = 49152, kb2 = 98304;
  asm volatile("" : "+v"(kb1), "+v"(kb2));
  unsigned char* const row_b0 = smem + (size_t)(16 * w + vi) * kRowB;
  unsigned char* row_b1 = row_b0 + kb1;
  unsigned char* row_b2 = row_b0 + kb2;
  auto rowp = [&](int s) -> unsigned char* {
    const int off = s * NW * 16 * kRowB;
    return off < 49152 ? row_b0 + off : off < 98304 ? row_b1 + (off - 49152) : row_b2 + (off - 98304);
  };
  const float* coef_l = coefv + 16 * w + vi;      // + s NW 16
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
  const float* xb = p_x + (long long)mesh * a.x_bs * 16;      // (strided x, ConvIO::x_map: row v of a mesh at x_map[v])
  // Every load of the prologue in two rounds of independent loads, indices clamped and no branch around a load (written
  // slot by slot behind their tests these were 2 x SLOTS dependent round trips: 12 k of the kernel's 60 k cycles): first
  // the rows' plan words (and, un-pooling, their three taps), then what they point to.
  uint32_t info[SLOTS];
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) info[s] = p_pinfo[o + min(16 * (s * NW + w) + vi, rows16 - 1)];
  float4 xv[SLOTS];
  if (a.u_rows > 0) {
    // the layer input is U x_coarse (nn/pool.py:17-20): three taps per row from the coarse tensor (5 MB, on chip), in
    // the arithmetic of the pooling op -- every product and every sum rounded, the operator's entry order -- so the
    // values are those a stored un-pooled tensor would hold.  Exclusive rows are written out once for the backward.
    uint2 ur[SLOTS][3];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const uint2* up = reinterpret_cast<const uint2*>(p_urec + (size_t)(o + min(16 * (s * NW + w) + vi, rows16 - 1)) * 6);
      ur[s][0] = up[0]; ur[s][1] = up[1]; ur[s][2] = up[2];
    }
    const float* xc = p_x + (long long)mesh * a.u_rows * 16 + 4 * q;
    constexpr int H = (SLOTS + 1) / 2;      // the taps' rows in two halves (12 registers per slot in flight)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float4 n[H][3];
#pragma unroll
      for (int s = h * H; s < min((h + 1) * H, SLOTS); ++s)
#pragma unroll
        for (int j = 0; j < 3; ++j) n[s - h * H][j] = *reinterpret_cast<const float4*>(xc + (long long)ur[s][j].x * 16);
#pragma unroll
      for (int s = h * H; s < min((h + 1) * H, SLOTS); ++s) {
        const float4 n0 = n[s - h * H][0], n1 = n[s - h * H][1], n2 = n[s - h * H][2];
        const float w0 = __uint_as_float(ur[s][0].y), w1 = __uint_as_float(ur[s][1].y), w2 = __uint_as_float(ur[s][2].y);
        xv[s].x = __fadd_rn(__fadd_rn(__fadd_rn(0.f, __fmul_rn(w0, n0.x)), __fmul_rn(w1, n1.x)), __fmul_rn(w2, n2.x));
        xv[s].y = __fadd_rn(__fadd_rn(__fadd_rn(0.f, __fmul_rn(w0, n0.y)), __fmul_rn(w1, n1.y)), __fmul_rn(w2, n2.y));
        xv[s].z = __fadd_rn(__fadd_rn(__fadd_rn(0.f, __fmul_rn(w0, n0.z)), __fmul_rn(w1, n1.z)), __fmul_rn(w2, n2.z));
        xv[s].w = __fadd_rn(__fadd_rn(__fadd_rn(0.f, __fmul_rn(w0, n0.w)), __fmul_rn(w1, n1.w)), __fmul_rn(w2, n2.w));
        const int t = s * NW + w;
        if (p_xstore && t < nt_all && (info[s] >> 24 & 15u) != 15u && ((info[s] >> 28) & 1u))
          *reinterpret_cast<float4*>(p_xstore + ((long long)mesh * a.N + (info[s] & 0xffffu)) * 16 + 4 * q) = xv[s];
      }
    }
  } else if (p_xmap) {
    int xr[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) xr[s] = p_xmap[info[s] & 0xffffu];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) xv[s] = *reinterpret_cast<const float4*>(xb + (long long)xr[s] * 16 + 4 * q);
  } else {
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) xv[s] = *reinterpret_cast<const float4*>(xb + (long long)(info[s] & 0xffffu) * 16 + 4 * q);
  }
#pragma unroll
  for (int s = 0; s < ASLOTS; ++s) acc[s] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    const int t = s * NW + w;
    st[s] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int v = 16 * t + vi;
    const float deg = (float)((info[s] >> 16) & 255u);
    const bool valid = t < nt0 && (info[s] >> 24 & 15u) != 15u;
    const float sc = valid ? (deg > 0.f ? __builtin_amdgcn_rsqf(deg) : 1.0f) : 0.f;
    // (a clamped load of a slot nobody needs may hold anything: selected away, not multiplied by zero)
    const float4 r = valid ? make_float4(xv[s].x * sc, xv[s].y * sc, xv[s].z * sc, xv[s].w * sc) : make_float4(0.f, 0.f, 0.f, 0.f);
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
    // (the lists and coefficients never change, and without a run-time test around their loads the compiler would hoist
    //  them out of the order loop into ~10 registers per slot pair: the bases are re-derived opaquely every order)
    asm volatile("" : "+v"(kb1), "+v"(kb2));
    row_b1 = row_b0 + kb1;
    row_b2 = row_b0 + kb2;
    coef_l = coefv + 16 * w + vi + (kb2 - 2 * kb1);
    // the SU unconditional slots in pairs (gather8x2), the others one by one behind their run-time test
#pragma unroll
    for (int s = 0; s + 1 < SU; s += 2) {
      const uint4 ia = *reinterpret_cast<const uint4*>(rowp(s) + 64);
      const uint4 ib = *reinterpret_cast<const uint4*>(rowp(s + 1) + 64);
      const float ca = coef_l[s * NW * 16] * sc, cb = coef_l[(s + 1) * NW * 16] * sc;
      float4 ga, gb;
      gather8x2(smem, 16u * q, ia, ib, padw, ga, gb);
      st[s] = f4fms(ca, ga, st[s]);
      st[s + 1] = f4fms(cb, gb, st[s + 1]);
      if (s < ASLOTS) mfma4(acc[s < ASLOTS ? s : 0], wa, st[s]);
      if (s + 1 < ASLOTS) mfma4(acc[s + 1 < ASLOTS ? s + 1 : 0], wa, st[s + 1]);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int s = SU & ~1; s < SLOTS; ++s) {
      const int t = s * NW + w;
      if (s < SU || t < ntk) {
        const uint4 id = *reinterpret_cast<const uint4*>(rowp(s) + 64);
        const float cc = coef_l[s * NW * 16] * sc;
        const float4 g = gather8(smem, 16u * q, id, padw);
        st[s] = f4fms(cc, g, st[s]);
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
        float4* own = reinterpret_cast<float4*>(rowp(s) + 16 * q);
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
  // optional per-vertex map behind the layer (the final conv of the VAE, cheb_VAE.py:288, off its 20-vertex block:
  // recon[v][o] = sum_c out[v][c] W_eff[c][o]): lane (vertex, quad q) collects the vertex's other three quads from its
  // sister lanes and runs output o = q's 16-term fma chain in k_cheb_contract's order, so the values are bitwise those of
  // the separate launch (or of the loss launch's fused form) -- which then never reads this layer's 20 MB output
  float wq[16];
  const int mo = min(q, max(a.map_c - 1, 0));
  if (a.map_c > 0) {
#pragma unroll
    for (int cc = 0; cc < 16; ++cc) wq[cc] = p_w3[cc * a.map_c + mo];
  }
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
      if (a.map_c > 0) {      // (uniform; the four lanes of a vertex share the test around this block)
        float xa16[16];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          xa16[4 * qq + 0] = __shfl(r0, vi + 16 * qq, 64);
          xa16[4 * qq + 1] = __shfl(r1, vi + 16 * qq, 64);
          xa16[4 * qq + 2] = __shfl(r2, vi + 16 * qq, 64);
          xa16[4 * qq + 3] = __shfl(r3, vi + 16 * qq, 64);
        }
        float m = 0.f;
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) m = fmaf(xa16[cc], wq[cc], m);
        const int gid = (int)(info & 0xffffu);
        if (q < a.map_c && gid >= a.map_n0) p_out3[row * a.map_c + q] = m;
      }
    }
  }
  MVH_STAMPX(27);
}

// ----------------------------------------------------------------------------------------------------------- backward
// RS >= tiles of the largest patch / NWR; AS >= core tiles / NWD; GS >= 4-vertex groups of the largest exclusive set / NWD;
// SU: see k_patch_fwd
template <int NWR, int NWD, int RS, int AS, int GS, int SU>
__global__ void __launch_bounds__((NWR + NWD) * 64)
k_patch_bwd(const float* __restrict__ p_dout, const uint8_t* __restrict__ p_mbits, const float* __restrict__ p_g3,
            const float* __restrict__ p_w3, const float* __restrict__ p_x, const int32_t* __restrict__ p_xmap,
            const float* __restrict__ p_W, float* __restrict__ p_dx, float* __restrict__ p_part,
            const int32_t* __restrict__ p_poff,
            const int32_t* __restrict__ p_cnt, const uint32_t* __restrict__ p_pinfo, const uint32_t* __restrict__ p_ell,
            const int32_t* __restrict__ p_prow_off, const int32_t* __restrict__ p_prow_gid,
            const int32_t* __restrict__ p_prow_ptr, const int32_t* __restrict__ p_pcol, const float* __restrict__ p_pval,
            PatchDims a) {
  constexpr int THREADS = (NWR + NWD) * 64;
  extern __shared__ __align__(16) unsigned char smem[];
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int mesh = (jj / a.P) * 8 + xcd, pt = jj % a.P;
  if (mesh >= a.B) return;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);   // (w: wave-uniform)
  const int vi = lane & 15, q = lane >> 4;
  const int o = p_poff[pt], rows16 = p_poff[pt + 1] - o;
  const int* __restrict__ c = p_cnt + pt * (a.R + 2);
  const int K = a.K;
  const unsigned padw = (unsigned)(rows16 * 5) * 0x10001u;      // the list pad (the zero row), both halves of a word
  const int n_excl = c[0], n_core = c[1];
  const int nt_all = rows16 >> 4;
  const int nt0 = (c[1 + min(a.R, K - 1)] + 15) >> 4;
  float* u = reinterpret_cast<float*>(smem);
  float* coefv = reinterpret_cast<float*>(smem + (size_t)(rows16 + 1) * kRowB);   // -2 / deg per local vertex
  float* wpart = coefv + rows16;                   // [NWD][256] the matrix waves' dW tiles of one order
  float* wdb = wpart + NWD * 256;                  // [NWR][16] the recurrence waves' db sums
  MVH_STAMPX(0);

  for (int i = tid; i < rows16; i += THREADS) {
    *reinterpret_cast<uint4*>(smem + (size_t)i * kRowB + 64) = reinterpret_cast<const uint4*>(p_ell)[o + i];
    const float deg = (float)((p_pinfo[o + i] >> 16) & 255u);
    coefv[i] = deg > 0.f ? -2.0f * __builtin_amdgcn_rcpf(deg) : 0.f;
  }
  if (tid < kRowF / 4) reinterpret_cast<float4*>(u + (size_t)rows16 * kRowF)[tid] = make_float4(0.f, 0.f, 0.f, 0.f);
  const long long mrow = (long long)mesh * a.N;
  const int wg = mesh * a.P + pt;          // this workgroup's partial tile (one per slab)
  const int tile = (K + 1) * 64;
  // pooling epilogue (has_dx, n_pool_rows > 0): rows r0 .. r0 + nrow of the plan, entries ebase .. ebase + ne of pcol / pval
  const bool pool = a.has_dx && a.n_pool_rows > 0;
  const int r0 = pool ? p_prow_off[pt] : 0, nrow = pool ? p_prow_off[pt + 1] - r0 : 0;
  const int* __restrict__ rp = p_prow_ptr + r0 + pt;
  const int ebase = pool ? rp[0] : 0, ne = pool ? rp[nrow] - ebase : 0;
  uint2* ent = reinterpret_cast<uint2*>(smem + (size_t)((n_core + 15) & ~15) * kRowB);   // (behind the dX rows of the core)
  const bool staged = pool && a.pool_lds != 0;     // (host: the largest patch's entries fit there, <= 8 per recurrence lane)
  // first item of this thread (row it >> 2, quad it & 3): its entry range and output row, fetched ahead of the barriers
  int e0f = 0, e1f = 0, gidf = 0;
  auto pre_item = [&]() {
    if (pool && tid < nrow * 4) {
      e0f = rp[tid >> 2] - ebase;
      e1f = rp[(tid >> 2) + 1] - ebase;
      gidf = p_prow_gid[r0 + (tid >> 2)];
    }
  };

  if (w < NWR) {
    // ================================================== recurrence waves (see k_patch_fwd): rows = dpre = dout * relu'
      // (three row bases 48 KB apart, opaque to the compiler: see k_patch_fwd)
    int kb1 = 49152, kb2 = 98304;
    asm volatile("" : "+v"(kb1), "+v"(kb2));
    unsigned char* const row_b0 = smem + (size_t)(16 * w + vi) * kRowB;
    unsigned char* row_b1 = row_b0 + kb1;
    unsigned char* row_b2 = row_b0 + kb2;
    auto rowp = [&](int s) -> unsigned char* {
      const int off = s * NWR * 16 * kRowB;
      return off < 49152 ? row_b0 + off : off < 98304 ? row_b1 + (off - 49152) : row_b2 + (off - 98304);
    };
    const float* coef_l = coefv + 16 * w + vi;      // + s NWR 16
    float4 st[RS];
    {
      float w3r[4][3];      // lazy rows: this lane's four columns of W3 [16][3]
      if (a.src3_n >= 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int t = 0; t < 3; ++t) w3r[i][t] = p_w3[(4 * q + i) * 3 + t];
      }
      float4 dbacc = make_float4(0.f, 0.f, 0.f, 0.f);
      // two phases, no run-time branch in either: every slot's plan word first, then every slot's row (the loads of all
      // RS slots are in flight together; a slot past the patch reads its last row and is discarded)
      uint32_t inf[RS];
#pragma unroll
      for (int s = 0; s < RS; ++s) inf[s] = p_pinfo[o + min(16 * (s * NWR + w) + vi, rows16 - 1)];
      const bool lazy = a.src3_n >= 0;
      if (lazy) {
#pragma unroll
        for (int s = 0; s < RS; ++s) {
          const float* gr = p_g3 + (mrow + (inf[s] & 0xffffu)) * 3;
          st[s] = make_float4(gr[0], gr[1], gr[2], 0.f);
        }
      } else {
#pragma unroll
        for (int s = 0; s < RS; ++s) st[s] = load4_any(p_dout, (mrow + (inf[s] & 0xffffu)) * 16 + 4 * q, a.dout_bf16 != 0);
      }
      uint32_t mb[RS];
#pragma unroll
      for (int s = 0; s < RS; ++s) mb[s] = p_mbits ? (uint32_t)p_mbits[(mrow + (inf[s] & 0xffffu)) * 4 + q] : 15u;
      bool fix_any = false;
#pragma unroll
      for (int s = 0; s < RS; ++s) {
        const int t = s * NWR + w;
        const int v = 16 * t + vi;
        const uint32_t info = inf[s];
        const float deg = (float)((info >> 16) & 255u);
        const bool valid = t < nt0 && (info >> 24 & 15u) != 15u;
        const float sc = valid ? (deg > 0.f ? __builtin_amdgcn_rsqf(deg) : 1.0f) : 0.f;
        float4 dv = st[s];
        if (lazy) {
          const float g0 = dv.x, g1 = dv.y, g2 = dv.z;
          dv.x = fmaf(g2, w3r[0][2], fmaf(g1, w3r[0][1], g0 * w3r[0][0]));
          dv.y = fmaf(g2, w3r[1][2], fmaf(g1, w3r[1][1], g0 * w3r[1][0]));
          dv.z = fmaf(g2, w3r[2][2], fmaf(g1, w3r[2][1], g0 * w3r[2][0]));
          dv.w = fmaf(g2, w3r[3][2], fmaf(g1, w3r[3][1], g0 * w3r[3][0]));
          fix_any = fix_any || (valid && (int)(info & 0xffffu) < a.src3_n);
        }
        const uint32_t m = mb[s];
        dv.x = (m & 1u) ? dv.x : 0.f;
        dv.y = (m & 2u) ? dv.y : 0.f;
        dv.z = (m & 4u) ? dv.z : 0.f;
        dv.w = (m & 8u) ? dv.w : 0.f;
        if (valid && v < n_excl) dbacc = f4add(dbacc, dv);
        if (t < nt_all) *reinterpret_cast<float4*>(u + (size_t)v * kRowF + 4 * q) = make_float4(dv.x * sc, dv.y * sc, dv.z * sc, dv.w * sc);
        st[s] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      // the few STORED rows of a lazy gradient (the connected block of the layer above, ConvIO::src3_n rows of the mesh):
      // the lanes that own one redo their slot from the stored row (rare: one divergent pass, skipped by every other wave)
      if (__builtin_amdgcn_ballot_w64(fix_any) != 0ull) {
#pragma unroll
        for (int s = 0; s < RS; ++s) {
          const int t = s * NWR + w;
          const int v = 16 * t + vi;
          const uint32_t info = inf[s];
          const int gid = (int)(info & 0xffffu);
          if (t < nt0 && (info >> 24 & 15u) != 15u && gid < a.src3_n) {
            const float deg = (float)((info >> 16) & 255u);
            const float sc = deg > 0.f ? __builtin_amdgcn_rsqf(deg) : 1.0f;
            const float* gr = p_g3 + (mrow + gid) * 3;
            const float g0 = gr[0], g1 = gr[1], g2 = gr[2];
            float4 lz, dv = load4_any(p_dout, (mrow + gid) * 16 + 4 * q, a.dout_bf16 != 0);
            lz.x = fmaf(g2, w3r[0][2], fmaf(g1, w3r[0][1], g0 * w3r[0][0]));
            lz.y = fmaf(g2, w3r[1][2], fmaf(g1, w3r[1][1], g0 * w3r[1][0]));
            lz.z = fmaf(g2, w3r[2][2], fmaf(g1, w3r[2][1], g0 * w3r[2][0]));
            lz.w = fmaf(g2, w3r[3][2], fmaf(g1, w3r[3][1], g0 * w3r[3][0]));
            const uint32_t m = mb[s];
            dv.x = (m & 1u) ? dv.x : 0.f; lz.x = (m & 1u) ? lz.x : 0.f;
            dv.y = (m & 2u) ? dv.y : 0.f; lz.y = (m & 2u) ? lz.y : 0.f;
            dv.z = (m & 4u) ? dv.z : 0.f; lz.z = (m & 4u) ? lz.z : 0.f;
            dv.w = (m & 8u) ? dv.w : 0.f; lz.w = (m & 8u) ? lz.w : 0.f;
            if (v < n_excl) dbacc = make_float4(dbacc.x + (dv.x - lz.x), dbacc.y + (dv.y - lz.y), dbacc.z + (dv.z - lz.z), dbacc.w + (dv.w - lz.w));
            *reinterpret_cast<float4*>(u + (size_t)v * kRowF + 4 * q) = make_float4(dv.x * sc, dv.y * sc, dv.z * sc, dv.w * sc);
          }
        }
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
    __syncthreads();
    MVH_STAMPX(2);
    for (int k = 1; k < K; ++k) {
      const int ntk = (c[1 + min(a.R, K - 1 - k)] + 15) >> 4;
      const float sc = (k == 1) ? 0.5f : 1.0f;
      asm volatile("" : "+v"(kb1), "+v"(kb2));     // (see k_patch_fwd: no hoisting of the lists out of the order loop)
      row_b1 = row_b0 + kb1;
      row_b2 = row_b0 + kb2;
      coef_l = coefv + 16 * w + vi + (kb2 - 2 * kb1);
#pragma unroll
      for (int s = 0; s + 1 < SU; s += 2) {      // the unconditional slots in pairs (gather8x2)
        const uint4 ia = *reinterpret_cast<const uint4*>(rowp(s) + 64);
        const uint4 ib = *reinterpret_cast<const uint4*>(rowp(s + 1) + 64);
        const float ca = coef_l[s * NWR * 16] * sc, cb = coef_l[(s + 1) * NWR * 16] * sc;
        float4 ga, gb;
        gather8x2(smem, 16u * q, ia, ib, padw, ga, gb);
        st[s] = f4fms(ca, ga, st[s]);
        st[s + 1] = f4fms(cb, gb, st[s + 1]);
        pin(st[s]);
        pin(st[s + 1]);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int s = SU & ~1; s < RS; ++s) {
        const int t = s * NWR + w;
        if (s < SU || t < ntk) {
          const uint4 id = *reinterpret_cast<const uint4*>(rowp(s) + 64);
          const float cc = coef_l[s * NWR * 16] * sc;
          const float4 g = gather8(smem, 16u * q, id, padw);
          st[s] = f4fms(cc, g, st[s]);
          pin(st[s]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      MVH_STAMPX(3 + 3 * (k - 1));
      __syncthreads();
      MVH_STAMPX(4 + 3 * (k - 1));
#pragma unroll
      for (int s = 0; s < RS; ++s) {
        const int t = s * NWR + w;
        if (t < ntk) {
          float4* own = reinterpret_cast<float4*>(rowp(s) + 16 * q);
          const float4 old = *own;
          *own = st[s];
          st[s] = old;
        }
      }
      __syncthreads();
      MVH_STAMPX(5 + 3 * (k - 1));
    }
    // while the matrix waves run the last order: this patch's pooling entries -> registers (8 per lane), then into the LDS
    // the halo rows, coefficients and dW tiles leave free once that pass is done
    uint2 pe[8];
    if (staged) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int e = min(tid + NWR * 64 * j, max(ne - 1, 0));
        pe[j] = make_uint2((uint32_t)p_pcol[ebase + e], __float_as_uint(p_pval[ebase + e]));
      }
    }
    pre_item();
    __syncthreads();      // (the matrix waves' pass over u_{K-1})
    if (!pool) return;
    if (staged) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int e = tid + NWR * 64 * j;
        if (e < ne) ent[e] = pe[j];
      }
    }
    __syncthreads();      // dX rows (+ pooling entries) staged in LDS
  } else {
    // ================================================== matrix waves, one order behind the recurrence:
    //   dX tile a NWD + wd:  acc += W_k^T (A) x u_k rows (B);   dW_k: groups g = gs NWD + wd: x^T (A, registers) x u_k rows (B)
    const int wd = w - NWR;
    const int dtid = tid - NWR * 64;
    v4f acc[AS];
    float xa[GS];
    float wa[4];
    auto load_w = [&](int k) {   // A = W_k^T: [c_in = vi][c_out = 4 q + s]
      float4 t = *reinterpret_cast<const float4*>(p_W + k * 256 + vi * 16 + 4 * q);
      // bf16 storage: the forward of this layer (k_cheb_l0h) multiplies by a bf16 COPY of the weights; dX is the gradient
      // of THAT function, so it takes the same rounded weights (one RNE rounding per element, bf16.hpp)
      if (a.x_bf16) t = bf16_unpack4(bf16_pack4(t.x, t.y, t.z, t.w));
      wa[0] = t.x; wa[1] = t.y; wa[2] = t.z; wa[3] = t.w;
    };
    // A operand of the weight gradient: D^1/2 x of the exclusive vertices, [c_in = vi][vertex 4 g + q]; 0 elsewhere
    const int ng = a.has_dw ? (n_excl + 3) >> 2 : 0;
    {   // (two phases, branch-free: every group's plan word, then every group's x -- 2 round trips, not 2 GS)
      uint32_t xinf[GS];
#pragma unroll
      for (int gs = 0; gs < GS; ++gs) xinf[gs] = p_pinfo[o + min(4 * (gs * NWD + wd) + q, rows16 - 1)];
      const float* xb = p_x + (long long)mesh * a.x_bs * 16 + vi;
      if (a.x_bf16) {          // (bf16 storage: never with a row map)
        const long long xo = (long long)mesh * a.x_bs * 16 + vi;
#pragma unroll
        for (int gs = 0; gs < GS; ++gs) xa[gs] = load1_any(p_x, xo + (long long)(xinf[gs] & 0xffffu) * 16, true);
      } else if (p_xmap) {
#pragma unroll
        for (int gs = 0; gs < GS; ++gs) xa[gs] = xb[(long long)p_xmap[xinf[gs] & 0xffffu] * 16];
      } else {
#pragma unroll
        for (int gs = 0; gs < GS; ++gs) xa[gs] = xb[(long long)(xinf[gs] & 0xffffu) * 16];
      }
#pragma unroll
      for (int gs = 0; gs < GS; ++gs) {
        const float deg = (float)((xinf[gs] >> 16) & 255u);
        const bool on = (gs * NWD + wd) < ng && ((xinf[gs] >> 28) & 1u);
        xa[gs] *= on ? (deg > 0.f ? __builtin_sqrtf(deg) : 1.0f) : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < AS; ++i) acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};
    const float* tile_l = u + (size_t)(16 * wd + vi) * kRowF + 4 * q;     // dX: this lane's quad of tile wd (+ a NWD 16 rows)
    const float* grp_l = u + (size_t)(4 * wd + q) * kRowF + vi;           // dW: row 4 wd + q, channel vi (+ gs NWD 4 rows)
    auto pass = [&](int k) {      // both gradients' share of order k from the u_k rows in LDS
      if (a.has_dx) {
        load_w(k);
#pragma unroll
        for (int i = 0; i < AS; ++i) {
          const float4 b = *reinterpret_cast<const float4*>(tile_l + (size_t)i * NWD * 16 * kRowF);
          mfma4(acc[i], wa, b);
        }
      }
      if (a.has_dw) {
        v4f t0 = {0.f, 0.f, 0.f, 0.f}, t1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int gs = 0; gs < GS; gs += 2) {
          const float b0 = grp_l[(size_t)gs * NWD * 4 * kRowF];
          t0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[gs], b0, t0, 0, 0, 0);
          if (gs + 1 < GS) {
            const float b1 = grp_l[(size_t)(gs + 1) * NWD * 4 * kRowF];
            t1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[gs + 1 < GS ? gs + 1 : gs], b1, t1, 0, 0, 0);
          }
        }
        *reinterpret_cast<float4*>(wpart + wd * 256 + lane * 4) =
            make_float4(t0[0] + t1[0], t0[1] + t1[1], t0[2] + t1[2], t0[3] + t1[3]);
      }
    };
    // behind a barrier: the matrix waves' tiles summed in wave order -> the workgroup's partial tile of order k
    auto flush = [&](int k) {
      if (!a.has_dw || dtid >= 256) return;
      float d = 0.f;
#pragma unroll
      for (int ww = 0; ww < NWD; ++ww) d += wpart[ww * 256 + dtid];
      // element dtid = (lane l = dtid >> 2, register r = dtid & 3) of the D tile: c_in = 4 (l >> 4) + r, c_out = l & 15
      const int l = dtid >> 2, r = dtid & 3;
      p_part[((long long)(l >> 4) * a.n_part + wg) * tile + (k * 16 + (l & 15)) * 4 + r] = d;
    };
    MVH_STAMPX(1);
    __syncthreads();
    MVH_STAMPX(2);
    if (a.has_dw && dtid < 16) {   // db partial of the workgroup: entries (order K, q = c_out, j = 0) of slab 0
      float d = 0.f;
#pragma unroll
      for (int ww = 0; ww < NWR; ++ww) d += wdb[ww * 16 + dtid];
      p_part[(long long)wg * tile + (K * 16 + dtid) * 4] = d;
    }
    for (int k = 1; k < K; ++k) {
      pass(k - 1);
      MVH_STAMPX(3 + 3 * (k - 1));
      __syncthreads();
      MVH_STAMPX(4 + 3 * (k - 1));
      flush(k - 1);
      __syncthreads();
      MVH_STAMPX(5 + 3 * (k - 1));
    }
    pass(K - 1);
    MVH_STAMPX(26);
    float isd[AS];        // D^1/2 of this lane's dX rows (from the LDS coefficients, which the recurrence waves overwrite next)
#pragma unroll
    for (int i = 0; i < AS; ++i) {
      const float cf = coefv[min(16 * (i * NWD + wd) + vi, rows16 - 1)];      // -2 / deg (0: isolated)
      isd[i] = cf < 0.f ? __builtin_sqrtf(-2.0f * __builtin_amdgcn_rcpf(cf)) : 1.0f;
    }
    pre_item();
    __syncthreads();
    flush(K - 1);
    if (!a.has_dx) return;
    // ---- dX rows (D^1/2 applied): straight to memory, or through LDS into the rows of the pooling operator
    if (a.n_pool_rows <= 0) {
#pragma unroll
      for (int i = 0; i < AS; ++i) {
        const int v = 16 * (i * NWD + wd) + vi;
        if (v < n_excl) {
          const uint32_t info = p_pinfo[o + v];
          const float deg = (float)((info >> 16) & 255u);
          const float is = deg > 0.f ? __builtin_sqrtf(deg) : 1.0f;
          store4_any(p_dx, (mrow + (info & 0xffffu)) * 16 + 4 * q, a.out_bf16 != 0, acc[i][0] * is, acc[i][1] * is,
                     acc[i][2] * is, acc[i][3] * is);
        }
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < AS; ++i) {
      const int v = 16 * (i * NWD + wd) + vi;
      if (v < ((n_core + 15) & ~15)) {
        const float is = isd[i];
        *reinterpret_cast<float4*>(u + (size_t)v * kRowF + 4 * q) =
            make_float4(acc[i][0] * is, acc[i][1] * is, acc[i][2] * is, acc[i][3] * is);
      }
    }
    __syncthreads();
  }
  // ---- every wave: the rows of the pooling operator's transpose this patch forms, from the dX rows in LDS.  The patch's
  // (column, value) entries sit behind the core rows by now (the recurrence waves fetched them during the last order):
  // the row loops run on LDS alone -- the operator's rows reach 54 entries, 14 dependent global round trips otherwise
  MVH_STAMPX(27);
  for (int it = tid; it < nrow * 4; it += THREADS) {
    const int i = it >> 2, qq = it & 3;
    const bool first = it == tid;
    const int e0 = first ? e0f : rp[i] - ebase, e1 = first ? e1f : rp[i + 1] - ebase;
    const int grow = first ? gidf : p_prow_gid[r0 + i];
    float4 sacc = make_float4(0.f, 0.f, 0.f, 0.f);
    // eight taps per round (loads issued together); taps past the row end: a valid entry with weight 0, so the sums stay
    // in the operator's entry order
    for (int e = e0; e < e1; e += 8) {
      float wv[8];
      int cc[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int ee = min(e + t, e1 - 1);
        uint2 cv;
        if (staged) cv = ent[ee];
        else cv = make_uint2((uint32_t)p_pcol[ebase + ee], __float_as_uint(p_pval[ebase + ee]));
        cc[t] = (int)cv.x;
        wv[t] = (e + t < e1) ? __uint_as_float(cv.y) : 0.f;
      }
      float4 n[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) n[t] = *reinterpret_cast<const float4*>(u + (size_t)cc[t] * kRowF + 4 * qq);
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        sacc.x = __fadd_rn(sacc.x, __fmul_rn(wv[t], n[t].x));
        sacc.y = __fadd_rn(sacc.y, __fmul_rn(wv[t], n[t].y));
        sacc.z = __fadd_rn(sacc.z, __fmul_rn(wv[t], n[t].z));
        sacc.w = __fadd_rn(sacc.w, __fmul_rn(wv[t], n[t].w));
      }
    }
    store4_any(p_dx, ((long long)mesh * a.n_pool_rows + grow) * 16 + 4 * qq, a.out_bf16 != 0, sacc.x, sacc.y, sacc.z, sacc.w);
  }
  MVH_STAMPX(28);
}

// ------------------------------------------------------------------------------------- first layer: <= 4 -> 16 channels
// cheb.0 of the encoder (cheb_VAE.py:264) followed by its one-hot downsampling (nn/pool.py D): x has <= 4 channels, and
// only the rows D selects (1250 of 4998) are read by anybody -- the next layer, and the layer's own weight gradient, which
// needs T_k(L) x at exactly those rows (cheb_tstack.hip).  So the recurrence runs on the INPUT side, one float4 per vertex
// (a quarter of the 16-channel recurrence the output-side Clenshaw form of cheb_lds.hip runs for this layer), in the same
// patch image as the 16 -> 16 kernels: a lane owns a ROW here (no matrix operand layout to respect).  After every order
// the pooled rows the plan assigns to the patch are read back from LDS in the pooling operator's row order:
//   stack[mesh][k][row] = T_k x (D^1/2 applied)  -- the plane k_stack_dw reads in the backward (no k_cheb_tstack launch),
//   acc[row][16] += T_k x W_k                     -- one lane per (row, 4 output channels), weights from LDS,
// and the epilogue stores relu(acc + bias) and its sign byte at the pooled rows only.
// SLOTS >= rows of the largest patch / THREADS; TS >= 4 x pooled rows of a patch / THREADS.
template <int THREADS, int SLOTS, int TS>
__global__ void __launch_bounds__(THREADS)
k_patch_enc0(const float* __restrict__ p_x, const float* __restrict__ p_W, const float* __restrict__ p_bias,
             float* __restrict__ p_pooled, uint8_t* __restrict__ p_bits, float* __restrict__ p_stack,
             const int32_t* __restrict__ p_poff, const int32_t* __restrict__ p_cnt, const uint32_t* __restrict__ p_pinfo,
             const uint32_t* __restrict__ p_ell, const int32_t* __restrict__ p_prow_off,
             const int32_t* __restrict__ p_prow_gid, const int32_t* __restrict__ p_prow_ptr,
             const int32_t* __restrict__ p_pcol, PatchDims a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int mesh = (jj / a.P) * 8 + xcd, pt = jj % a.P;
  if (mesh >= a.B) return;
  const int tid = threadIdx.x;
  const int o = p_poff[pt], rows16 = p_poff[pt + 1] - o;
  const int* __restrict__ c = p_cnt + pt * (a.R + 2);
  const int K = a.K;
  const unsigned padw = (unsigned)(rows16 * 5) * 0x10001u;
  float* coefv = reinterpret_cast<float*>(smem + (size_t)(rows16 + 1) * kRowB);   // [rows16]
  float* wl = coefv + rows16;                                                     // [K][4][16], rows >= cin zero
  MVH_STAMPX(0);
  if (tid == 0) *reinterpret_cast<float4*>(smem + (size_t)rows16 * kRowB) = make_float4(0.f, 0.f, 0.f, 0.f);
  const int n0 = c[1 + min(a.R, K - 1)];           // rows whose u_0 somebody needs
  const float* xb = p_x + (long long)mesh * a.x_bs * a.cin;
  // the pooled rows of this patch: lane task (row i, output quad q); a dead task reads the zero row.  One entry per row
  // (a selection operator): entry ebase + i.
  const int r0 = p_prow_off[pt], nrow = p_prow_off[pt + 1] - r0;
  const int ebase = p_prow_ptr[r0 + pt];
  const int n_sel = a.n_pool_rows;
  // every load of the prologue in two rounds of independent loads (indices clamped, no branch around a load): first the
  // row words and the tasks' (column, row id), then what they point to -- x rows and the selected vertices' row words.
  // (Written slot by slot behind their tests these were ten dependent round trips: 7.7k of the kernel's 28k cycles.)
  uint32_t info[SLOTS];
  uint4 ellw[SLOTS];
  int lc[TS], grow[TS];
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    const int vi = o + min(s * THREADS + tid, rows16 - 1);
    info[s] = p_pinfo[vi];
    ellw[s] = reinterpret_cast<const uint4*>(p_ell)[vi];
  }
  float wv = 0.f;                  // this thread's weight of the LDS copy [K][4][16] (rows >= cin zero)
  if (tid < K * 64) {
    const int k = tid >> 6, ci = (tid >> 4) & 3, co = tid & 15;
    wv = p_W[(k * a.cin + min(ci, a.cin - 1)) * 16 + co];
    if (ci >= a.cin) wv = 0.f;
  }
#pragma unroll
  for (int j = 0; j < TS; ++j) {
    const int i = (j * THREADS + tid) >> 2;
    const bool live = i < nrow;
    lc[j] = p_pcol[live ? ebase + i : 0];
    grow[j] = p_prow_gid[live ? r0 + i : 0];
  }
  float xr[SLOTS][4];
  uint32_t tinfo[TS];
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    const float* xp = xb + (long long)(info[s] & 0xffffu) * a.cin;
#pragma unroll
    for (int t = 0; t < 4; ++t) xr[s][t] = xp[t < a.cin ? t : 0];
  }
#pragma unroll
  for (int j = 0; j < TS; ++j) tinfo[j] = p_pinfo[o + lc[j]];
  float4 st[SLOTS];
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    const int v = s * THREADS + tid;
    st[s] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float deg = (float)((info[s] >> 16) & 255u);
    const bool use = v < n0 && (info[s] >> 24 & 15u) != 15u;
    const float sc = use ? (deg > 0.f ? __builtin_amdgcn_rsqf(deg) : 1.0f) : 0.f;
    const float4 r = make_float4(xr[s][0] * sc, a.cin > 1 ? xr[s][1] * sc : 0.f, a.cin > 2 ? xr[s][2] * sc : 0.f,
                                 a.cin > 3 ? xr[s][3] * sc : 0.f);
    if (v < rows16) {
      *reinterpret_cast<float4*>(smem + (size_t)v * kRowB) = r;
      *reinterpret_cast<uint4*>(smem + (size_t)v * kRowB + 64) = ellw[s];
      coefv[v] = deg > 0.f ? -2.0f * __builtin_amdgcn_rcpf(deg) : 0.f;
    }
  }
  if (tid < K * 64) wl[tid] = wv;
  float is[TS];
  float4 acc[TS];
#pragma unroll
  for (int j = 0; j < TS; ++j) {
    const bool live = ((j * THREADS + tid) >> 2) < nrow;
    const float deg = (float)((tinfo[j] >> 16) & 255u);
    is[j] = deg > 0.f ? __builtin_sqrtf(deg) : 1.0f;
    if (!live) { lc[j] = rows16; grow[j] = -1; }
    acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int q = tid & 3;
  float4* const sbase = reinterpret_cast<float4*>(p_stack) + (long long)mesh * K * (n_sel + 1);
  auto contract = [&](int k) {
#pragma unroll
    for (int j = 0; j < TS; ++j) {
      if (j * THREADS + (tid & ~63) >= 4 * nrow) continue;        // (wave-uniform: no task of this wave is live)
      const float4 uv = *reinterpret_cast<const float4*>(smem + (size_t)lc[j] * kRowB);
      const float4 T = make_float4(uv.x * is[j], uv.y * is[j], uv.z * is[j], uv.w * is[j]);
      if (p_stack && q == 0 && grow[j] >= 0) sbase[(long long)k * (n_sel + 1) + grow[j]] = T;
      const float4 w0 = *reinterpret_cast<const float4*>(wl + (k * 4 + 0) * 16 + 4 * q);
      const float4 w1 = *reinterpret_cast<const float4*>(wl + (k * 4 + 1) * 16 + 4 * q);
      const float4 w2 = *reinterpret_cast<const float4*>(wl + (k * 4 + 2) * 16 + 4 * q);
      acc[j].x = fmaf(T.z, w2.x, fmaf(T.y, w1.x, fmaf(T.x, w0.x, acc[j].x)));
      acc[j].y = fmaf(T.z, w2.y, fmaf(T.y, w1.y, fmaf(T.x, w0.y, acc[j].y)));
      acc[j].z = fmaf(T.z, w2.z, fmaf(T.y, w1.z, fmaf(T.x, w0.z, acc[j].z)));
      acc[j].w = fmaf(T.z, w2.w, fmaf(T.y, w1.w, fmaf(T.x, w0.w, acc[j].w)));
      if (a.cin > 3) {          // (uniform; the mesh input has three channels)
        const float4 w3 = *reinterpret_cast<const float4*>(wl + (k * 4 + 3) * 16 + 4 * q);
        acc[j].x = fmaf(T.w, w3.x, acc[j].x); acc[j].y = fmaf(T.w, w3.y, acc[j].y);
        acc[j].z = fmaf(T.w, w3.z, acc[j].z); acc[j].w = fmaf(T.w, w3.w, acc[j].w);
      }
    }
  };
  MVH_STAMPX(1);
  __syncthreads();
  MVH_STAMPX(2);
  contract(0);
  for (int k = 1; k < K; ++k) {
    const int lim = c[1 + min(a.R, K - 1 - k)];      // rows whose u_k somebody needs
    const float sc = (k == 1) ? 0.5f : 1.0f;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const int v = s * THREADS + tid;
      if (v < lim) {
        const uint4 id = *reinterpret_cast<const uint4*>(smem + (size_t)v * kRowB + 64);
        const float cc = coefv[v] * sc;
        const float4 g = gather8(smem, 0u, id, padw);
        st[s] = f4fms(cc, g, st[s]);
      }
    }
    MVH_STAMPX(3 + 3 * (k - 1));
    __syncthreads();            // every gather of u_{k-1} is done
    MVH_STAMPX(4 + 3 * (k - 1));
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const int v = s * THREADS + tid;
      if (v < lim) {
        float4* own = reinterpret_cast<float4*>(smem + (size_t)v * kRowB);
        const float4 old = *own;
        *own = st[s];
        st[s] = old;
      }
    }
    __syncthreads();            // u_k is in LDS (its readers: contract(k) now, the gathers of order k + 1)
    MVH_STAMPX(5 + 3 * (k - 1));
    contract(k);
  }
  MVH_STAMPX(26);
  const float4 b4 = p_bias ? *reinterpret_cast<const float4*>(p_bias + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int j = 0; j < TS; ++j) {
    if (grow[j] < 0) continue;
    float r0v = acc[j].x + b4.x, r1v = acc[j].y + b4.y, r2v = acc[j].z + b4.z, r3v = acc[j].w + b4.w;
    if (a.act == MVH_ACT_RELU) { r0v = fmaxf(r0v, 0.f); r1v = fmaxf(r1v, 0.f); r2v = fmaxf(r2v, 0.f); r3v = fmaxf(r3v, 0.f); }
    store4_any(p_pooled, ((long long)mesh * n_sel + grow[j]) * 16 + 4 * q, a.out_bf16 != 0, r0v, r1v, r2v, r3v);
    if (p_bits) {
      const long long vrow = (long long)mesh * a.N + (tinfo[j] & 0xffffu);
      p_bits[vrow * 4 + q] = (uint8_t)((r0v > 0.f ? 1 : 0) | (r1v > 0.f ? 2 : 0) | (r2v > 0.f ? 4 : 0) | (r3v > 0.f ? 8 : 0));
    }
  }
  MVH_STAMPX(27);
}

#ifdef MVH_STAMP
MVH_STAMP_READER(mvh_debug_read_stamps_patch)
#endif

// ------------------------------------------------------------------------------------------------------------- host
// wave roles and register-array sizes (the largest patch the LDS admits: 106 tiles; exclusive sets of <= 1 280 vertices)
struct FwdCfg { static constexpr int THREADS = 1024, S = 7, A = 6, SU = 4; };      // (uniform waves: tile slots, output-tile slots)
struct BwdCfg { static constexpr int NWR = 8, NWD = 8, RS = 14, AS = 11, GS = 40, SU = 10; };

// rows + lists, -2 / deg, and (backward) the matrix waves' dW tiles of one order + the recurrence waves' db sums
static size_t patch_lds_bytes(const mvh_patch_plan_t* pl, bool bwd) {
  const size_t fwd = (size_t)(pl->max_rows + 1) * kRowB + (size_t)pl->max_rows * 4;
  return bwd ? fwd + (size_t)(BwdCfg::NWD * 256 + BwdCfg::NWR * 16) * 4 : fwd;
}

static bool cfg_fits(const mvh_patch_plan_t* pl) {
  const int tiles = pl->max_rows / 16;
  return cdiv(tiles, FwdCfg::THREADS / 64) <= FwdCfg::S && cdiv(cdiv(pl->max_excl, 16), FwdCfg::THREADS / 64) <= FwdCfg::A &&
         cdiv(tiles, BwdCfg::NWR) <= BwdCfg::RS && cdiv(cdiv(pl->max_core, 16), BwdCfg::NWD) <= BwdCfg::AS &&
         cdiv(cdiv(pl->max_excl, 4), BwdCfg::NWD) <= BwdCfg::GS;
}

bool patch_eligible(const mvh_csr_t* lap, int N, int Cin, int Cout, int K) {
  if (dbg().no_patch || dbg().force_generic) return false;
  const mvh_patch_plan_t* pl = lap ? lap->patch : nullptr;
  if (!pl || Cin != 16 || Cout != 16 || K < 1 || K > 12) return false;
  const int need = MVH_CSR_NORMALIZED_LAPLACIAN | MVH_CSR_SYMMETRIC;
  if ((lap->flags & need) != need) return false;
  if (pl->n_vertices != N || pl->n_patches < 1 || K - 1 > pl->n_rings) return false;
  if (patch_lds_bytes(pl, true) > 160 * 1024 || pl->max_rows % 16 != 0 || pl->max_rows < 16) return false;
  return cfg_fits(pl);
}

// the level's 16 -> 16 forward can take the coarse tensor and un-pool it in its loads: the plan was built with THIS
// un-pooling operator (identity of its transpose's rowptr is not available here: the row counts and the operator's
// shape are checked, the step engine builds both from the same Operator) and carries its rows
bool patch_unpool_eligible(const mvh_csr_t* lap, const mvh_csr_t* up, int N, int Cin, int Cout, int K) {
  if (dbg().no_patch_unpool == 1 || !patch_eligible(lap, N, Cin, Cout, K) || !up) return false;
  const mvh_patch_plan_t* pl = lap->patch;
  return pl->urec != nullptr && pl->u_rows > 0 && up->n_rows == N && up->n_cols == pl->u_rows && pl->n_pool_rows == pl->u_rows;
}

size_t patch_part_bytes(const mvh_csr_t* lap, int B, int K) {
  const mvh_patch_plan_t* pl = lap ? lap->patch : nullptr;
  if (!pl) return 0;
  return (size_t)4 * B * pl->n_patches * (K + 1) * 64 * sizeof(float) + 256;
}

int launch_patch_fwd(hipStream_t st, const mvh_csr_t* lap, const float* x, const float* W, const float* bias, float* out,
                     uint8_t* bits, int B, int N, int K, int act, const int32_t* x_map, int x_bs, bool x_unpool,
                     float* x_store, const float* map_w, float* map_out, int map_c, int map_n0) {
  const mvh_patch_plan_t* pl = lap->patch;
  MVH_REQUIRE(!x_unpool || (pl->urec && pl->u_rows > 0 && !x_map), "patch_fwd: the plan carries no un-pooling rows");
  MVH_REQUIRE(((uintptr_t)x_store & 15) == 0, "patch_fwd: tensors must be 16-byte aligned");
  MVH_REQUIRE((((uintptr_t)x | (uintptr_t)out | (uintptr_t)bias) & 15) == 0, "patch_fwd: tensors must be 16-byte aligned");
  MVH_REQUIRE(cfg_fits(pl), "patch_fwd: the plan does not fit the kernel's register arrays");
  PatchDims d{};
  d.B = B; d.N = N; d.K = K; d.P = pl->n_patches; d.R = pl->n_rings; d.act = act;
  d.x_bs = x_map ? x_bs : N;
  d.u_rows = x_unpool ? pl->u_rows : 0;
  MVH_REQUIRE(!map_out || (map_w && map_c >= 1 && map_c <= 4 && map_n0 >= 0), "patch_fwd: bad per-vertex map");
  d.map_c = map_out ? map_c : 0; d.map_n0 = map_n0;
  using C = FwdCfg;
  // slots that are core tiles for every wave of every patch
  const bool su_ok = (pl->min_core / 16) / (C::THREADS / 64) >= C::SU;
  auto kern = su_ok ? k_patch_fwd<C::THREADS, C::S, C::A, C::SU> : k_patch_fwd<C::THREADS, C::S, C::A, 0>;
  const size_t lds = patch_lds_bytes(pl, false);
  static LdsAttr attr[2];
  if (int rc = attr[su_ok].ensure(reinterpret_cast<const void*>(kern), lds)) return rc;
  const int grid = ((d.B + 7) / 8) * 8 * d.P;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(C::THREADS), lds, st, x, x_map, W, bias, out, bits, pl->poff, pl->cnt,
                     pl->pinfo, pl->ell, pl->urec, x_unpool ? x_store : nullptr, map_w, map_out, d);
  MVH_LAUNCH_CHECK();
  return MVH_OK;
}

// dX (+ its pooling when pooled) and the dW / db partial tiles of a 16 -> 16 layer in one launch.  dW / db themselves
// come out of the reduction `defer` describes (launch_dw_reduce_all; the caller runs it, at once or deferred).
int launch_patch_bwd(hipStream_t st, const mvh_csr_t* lap, const float* x, const float* W, const float* dout,
                     const uint8_t* mbits, const float* g3, const float* w3, int src3_n, float* dx, bool pooled,
                     float* part, size_t part_bytes, DwReduceEntry* defer, float* dW, float* db, int B, int N, int K,
                     const int32_t* x_map, int x_bs, bool x_bf16, bool dout_bf16, bool dx_bf16) {
  const mvh_patch_plan_t* pl = lap->patch;
  MVH_REQUIRE((((uintptr_t)x | (uintptr_t)dout | (uintptr_t)dx | (uintptr_t)W | (uintptr_t)part) & 15) == 0,
              "patch_bwd: tensors must be 16-byte aligned");
  MVH_REQUIRE(!(x_bf16 && x_map), "patch_bwd: a strided x is an fp32 tensor");
  MVH_REQUIRE(!pooled || pl->n_pool_rows > 0, "patch_bwd: the plan carries no pooling rows");
  MVH_REQUIRE(dx || dW, "patch_bwd: nothing to compute");
  MVH_REQUIRE(cfg_fits(pl), "patch_bwd: the plan does not fit the kernel's register arrays");
  PatchDims d{};
  d.B = B; d.N = N; d.K = K; d.P = pl->n_patches; d.R = pl->n_rings; d.act = 0;
  d.x_bs = x_map ? x_bs : N;
  d.n_pool_rows = pooled ? pl->n_pool_rows : 0;
  d.x_bf16 = x_bf16 ? 1 : 0; d.dout_bf16 = dout_bf16 ? 1 : 0; d.out_bf16 = dx_bf16 ? 1 : 0;
  d.src3_n = g3 ? src3_n : -1;
  d.n_part = B * pl->n_patches;
  d.has_dw = dW ? 1 : 0; d.has_dx = dx ? 1 : 0;
  // free LDS behind the dX rows of the largest core at the pooling epilogue: halo rows, zero row, coefficients, dW tiles
  d.pool_lds = pooled && pl->max_pool_nnz <= BwdCfg::NWR * 64 * 8 &&
               (size_t)pl->max_pool_nnz * 8 + (size_t)((pl->max_core + 15) / 16 * 16) * kRowB <= patch_lds_bytes(pl, true);
  if (dW) {
    MVH_REQUIRE(x && part && defer && part_bytes >= patch_part_bytes(lap, B, K), "patch_bwd: partial-tile buffer too small");
    *defer = DwReduceEntry{};
    defer->part = part; defer->n_part = d.n_part; defer->NS = 4; defer->K = K; defer->CQ = 16; defer->CP = 16;
    defer->p_is_x = 1; defer->Cin = 16; defer->Cout = 16; defer->db_mode = db ? 1 : 0; defer->dW = dW; defer->db = db;
    defer->S = nullptr;
  }
  using C = BwdCfg;
  const bool su_ok = (pl->min_core / 16) / C::NWR >= C::SU;
  auto kern = su_ok ? k_patch_bwd<C::NWR, C::NWD, C::RS, C::AS, C::GS, C::SU> : k_patch_bwd<C::NWR, C::NWD, C::RS, C::AS, C::GS, 0>;
  const size_t lds = patch_lds_bytes(pl, true);
  static LdsAttr attr[2];
  if (int rc = attr[su_ok].ensure(reinterpret_cast<const void*>(kern), lds)) return rc;
  const int grid = ((d.B + 7) / 8) * 8 * d.P;
  hipLaunchKernelGGL(kern, dim3(grid), dim3((C::NWR + C::NWD) * 64), lds, st, dout, mbits, g3, w3, x, x_map, W, dx, part,
                     pl->poff, pl->cnt, pl->pinfo, pl->ell, pl->prow_off, pl->prow_gid, pl->prow_ptr, pl->pcol, pl->pval, d);
  MVH_LAUNCH_CHECK();
  return MVH_OK;
}

// ---- first layer (<= 4 -> 16 channels + one-hot pooling): the plan hangs off the POOLING operator (down->patch), built
// from the level's Laplacian with that operator's rows as its pooling rows (topology.patch_plan(.., down_op=))
struct Enc0Cfg { static constexpr int THREADS = 1024, SLOTS = 2, TS = 2; };

static size_t enc0_lds_bytes(const mvh_patch_plan_t* pl, int K) {
  return (size_t)(pl->max_rows + 1) * kRowB + (size_t)pl->max_rows * 4 + (size_t)K * 64 * 4;
}

bool patch_enc0_eligible(const mvh_csr_t* lap, const mvh_csr_t* down, int N, int Cin, int Cout, int K) {
  if (dbg().no_patch == 1 || dbg().no_enc0_patch || dbg().force_generic) return false;   // (no_patch = 2: the 16 -> 16 kernels only, diagnostics)
  const mvh_patch_plan_t* pl = down ? down->patch : nullptr;
  if (!pl || !lap || Cin < 1 || Cin > 4 || Cout != 16 || K < 1 || K > 12) return false;
  const int need = MVH_CSR_NORMALIZED_LAPLACIAN | MVH_CSR_SYMMETRIC;
  if ((lap->flags & need) != need || !(down->flags & MVH_CSR_SELECTION)) return false;
  if (pl->n_vertices != N || down->n_cols != N || pl->n_patches < 1 || K - 1 > pl->n_rings) return false;
  if (pl->n_pool_rows != down->n_rows || pl->pool_rowptr != down->rowptr) return false;
  if (pl->max_rows % 16 != 0 || pl->max_rows < 16 || pl->max_rows > Enc0Cfg::THREADS * Enc0Cfg::SLOTS) return false;
  if (4 * pl->max_pool_nnz > Enc0Cfg::THREADS * Enc0Cfg::TS) return false;      // (one entry per pooled row)
  return enc0_lds_bytes(pl, K) <= 160 * 1024;
}

int launch_patch_enc0(hipStream_t st, const mvh_csr_t* lap, const mvh_csr_t* down, const float* x, const float* W,
                      const float* bias, float* pooled, bool pooled_bf16, uint8_t* bits, float* stack, int B, int N,
                      int Cin, int K, int act) {
  const mvh_patch_plan_t* pl = down->patch;
  MVH_REQUIRE(x && W && pooled, "patch_enc0: null tensor");
  MVH_REQUIRE((((uintptr_t)pooled | (uintptr_t)bias | (uintptr_t)stack) & 15) == 0 && ((uintptr_t)x & 3) == 0,
              "patch_enc0: tensors must be 16-byte aligned");
  PatchDims d{};
  d.B = B; d.N = N; d.K = K; d.P = pl->n_patches; d.R = pl->n_rings; d.act = act;
  d.x_bs = N; d.cin = Cin; d.out_bf16 = pooled_bf16 ? 1 : 0; d.n_pool_rows = pl->n_pool_rows;
  using C = Enc0Cfg;
  auto kern = k_patch_enc0<C::THREADS, C::SLOTS, C::TS>;
  const size_t lds = enc0_lds_bytes(pl, K);
  static LdsAttr attr;
  if (int rc = attr.ensure(reinterpret_cast<const void*>(kern), lds)) return rc;
  const int grid = ((d.B + 7) / 8) * 8 * d.P;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(C::THREADS), lds, st, x, W, bias, pooled, bits, stack, pl->poff, pl->cnt,
                     pl->pinfo, pl->ell, pl->prow_off, pl->prow_gid, pl->prow_ptr, pl->pcol, d);
  MVH_LAUNCH_CHECK();
  return MVH_OK;
}

}  // namespace mvh
