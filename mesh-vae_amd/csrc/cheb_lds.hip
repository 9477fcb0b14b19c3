// LDS-resident ChebConv: the whole K-order recurrence of ONE mesh stays on one CU.
//
//   out = act( sum_k T_k(L) (In W_k) + bias )       In = x (forward)  or  dpre (backward dX,
//                                                    with W_k^T; L is symmetric)
// is evaluated by Clenshaw's recurrence on the OUTPUT side,
//   b_k = In W_k + 2 L b_{k+1} - b_{k+2},   out = In W_0 + L b_1 - b_2,
// so every output channel is independent and a workgroup owns (mesh, slab of 4 output channels):
//   * LDS   : b_{k+1} of the slab for every vertex as one float4 (80 KB at N=4998) and the
//             neighbour lists in padded ELL form, vertex-major, two uint16 per word (80 KB):
//             together exactly the CU's 160 KB at the 5k template;
//   * VGPRs : the input rows In[v][0:CQ] of the thread's own VPT vertices (512 threads x 10
//             vertices at N=4998; read from HBM once, 16 B/lane coalesced) and b_{k+2}[own];
//   * SGPRs : the weight slab (wave-uniform scalar loads);
//   * per order: In W_k as v_fma with SGPR weights, one ds_read_b128 of 8 neighbour ids and one
//             unweighted ds_read_b128 gather per neighbour, two barriers.  No global traffic
//             between orders.
// L = -D^-1/2 A D^-1/2 is applied in scaled variables u = D^-1/2 b so the edge list needs no
// values:  u_k = s (In W_k) - (2/deg) sum_{j in N(i)} u_{k+1}[j] - u_{k+2},  s = deg^-1/2
// (s = 1 and no gather for isolated vertices, which covers the final-layer quirk,
// cheb_VAE.py:288).  Slots past N carry zeros (s = 0) and own a zero row, so the padded ELL
// entries (index N) gather zeros and no branch on validity exists before the final store.
// Replaces 5 propagate launches + 1 contraction of the stack pipeline with one launch whose
// HBM traffic is the module boundary: read In once (per slab, L2-shared), write out once.
#include "common.hpp"
#include "bf16.hpp"

#include <type_traits>

namespace mvh {

struct LdsConvArgs {
  const float* in;
  const float* mask;
  const float* W;
  const float* bias;
  float* out;
  const uint32_t* rowinfo;
  const uint32_t* ell;  // [pairs][N] pair-slot major (global)
  int B, N, K, CO, Cin, Cout, pairs, act;
  int in_bs, out_bs;    // rows per mesh in the in/mask and out buffers (>= N: strided sub-problem)
  int mask_bs, pooled_bs;
  int mask_bits;            // mask points at ReLU sign bytes (one per vertex and 4 channels) instead of floats
  uint8_t* bits_out;        // forward: also store the ReLU sign bytes of the output
  const int32_t* in_map;    // optional: row v of the input is in[in_map[v]] (zero when < 0)
  const int32_t* pool_inv;  // optional fused one-hot pooling: out row v also goes to pooled[pool_inv[v]]
  float* pooled;
  const int* col;           // CSR columns: rows longer than the 8 ELL slots continue there (ovf)
  int ovf;
  const int* pt_rowptr;     // optional (backward): the output rows are pooled by this CSR (n_rows = pt_rows) inside
  const int* pt_col;        // the kernel and ONLY the pooled rows [B][pt_rows][CO] are stored to `out`
  const float* pt_val;
  int pt_rows;
  int in_bf16, out_bf16, pooled_bf16;  // storage type of in / out / pooled: bf16 instead of fp32 (bf16.hpp)
  int out_dead;                        // forward + pool_inv: only the selected rows (and their sign bytes) are stored
  const float* g3;                     // ConvIO::src3_*
  const float* w3;
  int src3_n, src3_c;
};

__device__ __forceinline__ void add4(float4& a, const float4& b) {
  a.x += b.x;
  a.y += b.y;
  a.z += b.z;
  a.w += b.w;
}

// acc (lo | hi pair) += x_c * w[c][0..3] and += x_{c+1} * w[c+1][0..3] as four v_pk_fma_f32: the x operand is ONE aligned
// register pair (x_c, x_{c+1}) whose low / high half is broadcast by op_sel, the weights are SGPR pairs.  Each half of a
// packed FMA is the scalar FMA (results bitwise those of the v_fma form); written as inline asm because the compiler's
// own packing of this loop (float2 vectors, or -fslp-vectorize) duplicates operands and spills thousands of registers.
// MEASURED: 42.7 -> 41.8 us per forward launch and 29 -> 16 spilled VGPRs: v_pk_fma_f32 halves the instruction count, not
// the cycles (a wave64 packed FMA takes twice the cycles of a plain one on the SIMD-32), so the gain is the registers.
typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void fma4x2(f2v& lo, f2v& hi, const f2v& xpair, const float* __restrict__ w) {
  const f2v w0 = {w[0], w[1]}, w1 = {w[2], w[3]}, w2 = {w[4], w[5]}, w3 = {w[6], w[7]};
  asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(lo) : "v"(xpair), "s"(w0));
  asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(hi) : "v"(xpair), "s"(w1));
  asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(lo) : "v"(xpair), "s"(w2));
  asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(hi) : "v"(xpair), "s"(w3));
}

// PW = ELL words per vertex in LDS (4 -> up to 8 neighbours, 8 -> up to 16)
struct LdsConvDims {
  int B, N, K, CO, Cin, Cout, pairs, act, in_bs, out_bs, mask_bs, pooled_bs, mask_bits, pt_rows, ovf;
  int in_bf16, out_bf16, pooled_bf16, out_dead;
  int src3_n, src3_c;   // > 0: input rows >= src3_n are p_g3[v][0..src3_c) W3^T (lazy output gradient of a split-path layer)
};

// Pointers are separate __restrict__ kernel arguments (not struct members) so that hipcc can
// prove the weight loads are uniform + unclobbered and emit scalar s_load for them.
// TCT > 0: block size fixed at compile time (the 512 x 10 configuration of the 5k level, so its
// LDS offsets are immediates); TCT == 0: any block size <= 1024 (small levels run VPT = 1 or 2
// with one thread slot per vertex, which gives 5..16 waves per CU for latency hiding).
template <int CQ, int VPT, int TCT, int PW, bool BWD>
__global__ void __launch_bounds__(TCT > 0 ? TCT : 1024)
k_cheb_lds(const float* __restrict__ p_in, const float* __restrict__ p_mask, const float* __restrict__ p_W,
           const float* __restrict__ p_bias, float* __restrict__ p_out, const uint32_t* __restrict__ p_rowinfo,
           const uint32_t* __restrict__ p_ell, const int32_t* __restrict__ p_in_map,
           const int32_t* __restrict__ p_pool_inv, float* __restrict__ p_pooled, uint8_t* __restrict__ p_bits_out,
           const int* __restrict__ p_pt_rowptr, const int* __restrict__ p_pt_col, const float* __restrict__ p_pt_val,
           const int* __restrict__ p_col, const float* __restrict__ p_g3, const float* __restrict__ p_w3, LdsConvDims a) {
  const int THREADS = TCT > 0 ? TCT : (int)blockDim.x;
  const int VS = VPT * THREADS;  // vertex slots (> N)
  extern __shared__ __align__(16) unsigned char smem[];
  float4* slab = reinterpret_cast<float4*>(smem);             // [VS]; rows >= N stay zero
  uint4* ellv = reinterpret_cast<uint4*>(slab + VS);           // [VS][PW/4]  (kDB: the second slab instead)
  // Small levels (TCT == 0: one or two vertices per thread, registers to spare): the neighbour ids of the
  // thread's own vertices live in VGPRs (nobody else needs them) and the LDS holds TWO slabs, u_{k+1}
  // and u_{k+2}: an order gathers from one, reads/overwrites only its OWN rows of the other, so it needs
  // ONE barrier instead of two, no register copy of u_{k+2}, no ELL image and no staging pass.
  constexpr bool kDB = (TCT == 0);
  float4* slabB = slab + VS;

  // blocks b and b+8 share an XCD: keep the slabs of one mesh on one L2 (speed only)
  const int NS = (a.CO + 3) >> 2;
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int mesh = (jj / NS) * 8 + xcd, s0 = (jj % NS) * 4;
  if (mesh >= a.B) return;  // uniform per block, before any barrier
  const int tid = threadIdx.x, N = a.N;
  MVH_STAMPX(0);

  if constexpr (!kDB) {  // stage the vertex-major ELL lists with 16-byte copies; slots past N point at the zero row N
    const unsigned pad = (unsigned)N | ((unsigned)N << 16);
    const uint4 pad4 = make_uint4(pad, pad, pad, pad);
    const uint4* src = reinterpret_cast<const uint4*>(p_ell);
    for (int i = tid; i < VS * (PW / 4); i += THREADS) {
      const int v = i / (PW / 4);
      ellv[i] = (v < N) ? src[i] : pad4;
    }
  }
  uint4 ids[kDB ? VPT : 1][PW / 4];
  MVH_STAMPX(1);

  // ---- own vertices: -2/deg and the input rows scaled by s = deg^-1/2 (0 for slots past N)
  float ka2[VPT];
  float xs[VPT][CQ];
  float4 R[VPT];
  // MVH_CSR_ELL_OVERFLOW (small levels only, TCT == 0): columns 8..11 of the few long rows, two per
  // word like the ELL (pad = zero row N), fetched once; ovf_any[vi] keeps the extra gathers branchy
  constexpr bool kOvf = (TCT == 0);
  uint32_t ovf0[kOvf ? VPT : 1], ovf1[kOvf ? VPT : 1];
  bool ovf_any[kOvf ? VPT : 1];
  const float* inb = p_in + (long long)mesh * a.in_bs * CQ;
  // bf16 storage: the same rows as 2-byte elements (CQ % 4 == 0; the host checks)
  const uint16_t* inh = reinterpret_cast<const uint16_t*>(p_in) + (long long)mesh * a.in_bs * CQ;
  const bool use_bits = BWD && p_mask && a.mask_bits && (CQ % 4 == 0);
  const float* mkb = (BWD && p_mask && !use_bits) ? p_mask + (long long)mesh * a.mask_bs * CQ : nullptr;
  // ReLU sign bytes written by the forward kernel: CQ/4 bytes per vertex, bit j of byte c/4 = out[v][c+j] > 0
  const uint8_t* mbits = reinterpret_cast<const uint8_t*>(p_mask) + (long long)mesh * a.mask_bs * (CQ / 4);
  // The row map and the mask mode are wave-uniform; as run-time branches they would fence every vertex's
  // loads into its own basic block (one memory round trip per vertex).  The loop is therefore a generic
  // lambda instantiated per (map, mask mode) and dispatched once, so each copy is straight-line code.
  auto load_rows = [&](auto map_tag, auto mask_tag, auto bf_tag, auto src3_tag) {
    constexpr bool kMap = decltype(map_tag)::value;
    constexpr int kMask = decltype(mask_tag)::value;  // 0 none, 1 fp32 mask, 2 sign bytes
    constexpr bool kBF = decltype(bf_tag)::value;     // input rows stored as bf16
    // lazy rows (ConvIO::src3_*): row v >= src3_n is g3[v][0..src3_c) W3^T; the first src3_n rows (all owned by threads
    // tid < src3_n at vi == 0) are stored rows.  5k-level dX kernel only (CQ == 16, 1024 x 5, fp32, sign bytes).
    constexpr bool kSrc3 = decltype(src3_tag)::value;
  #pragma unroll
    for (int vi = 0; vi < VPT; ++vi) {
      const int v = tid + vi * THREADS;
      const bool valid = v < N;
      const int vl = min(v, N - 1);
      const uint32_t rinfo = p_rowinfo[vl];
      const float deg = valid ? (float)(rinfo & 255u) : 0.f;
      ka2[vi] = deg > 0.f ? -2.0f * __builtin_amdgcn_rcpf(deg) : 0.f;
      float s = valid ? (deg > 0.f ? __builtin_amdgcn_rsqf(deg) : 1.0f) : 0.f;
      R[vi] = make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (kDB) {
        const unsigned padi = (unsigned)N | ((unsigned)N << 16);
  #pragma unroll
        for (int q = 0; q < PW / 4; ++q)
          ids[vi][q] = valid ? reinterpret_cast<const uint4*>(p_ell)[vl * (PW / 4) + q] : make_uint4(padi, padi, padi, padi);
      }
      if constexpr (kOvf) {
        const unsigned padw = (unsigned)N | ((unsigned)N << 16);
        ovf0[vi] = ovf1[vi] = padw;
        const int dg = valid ? (int)(rinfo & 255u) : 0;
        ovf_any[vi] = a.ovf && dg > 8;
        if (ovf_any[vi]) {
          const int* cp = p_col + (rinfo >> 8) + 8;
          const unsigned c0 = (unsigned)cp[0], c1 = dg > 9 ? (unsigned)cp[1] : (unsigned)N;
          const unsigned c2 = dg > 10 ? (unsigned)cp[2] : (unsigned)N, c3 = dg > 11 ? (unsigned)cp[3] : (unsigned)N;
          ovf0[vi] = c0 | (c1 << 16);
          ovf1[vi] = c2 | (c3 << 16);
        }
      }
      int rl = vl;  // input row (through the optional selection map: un-pooled gradient rows)
      if constexpr (kMap) {
        const int rr = p_in_map[vl];
        if (rr < 0) s = 0.f;
        rl = max(rr, 0);
      }
      if constexpr (CQ % 4 == 0) {
        uint32_t mw[(CQ + 15) / 16];
  #pragma unroll
        for (int h = 0; h < (CQ + 15) / 16; ++h) mw[h] = 0xffffffffu;
        if constexpr (kMask == 2) {
          if constexpr (CQ % 16 == 0) {
  #pragma unroll
            for (int h = 0; h < CQ / 16; ++h) mw[h] = reinterpret_cast<const uint32_t*>(mbits)[vl * (CQ / 16) + h];
          } else if constexpr (CQ == 8) {
            mw[0] = reinterpret_cast<const uint16_t*>(mbits)[vl];
          } else {
  #pragma unroll
            for (int c = 0; c < CQ; c += 4) {
              if (c % 16 == 0) mw[c / 16] = 0;
              mw[c / 16] |= (uint32_t)mbits[vl * (CQ / 4) + c / 4] << (2 * (c % 16));
            }
          }
        }
        float4 tv[CQ / 4];
        if constexpr (kBF && CQ % 8 == 0) {          // 16-byte loads: 8 channels each
  #pragma unroll
          for (int c = 0; c < CQ; c += 8) {
            const uint4 w = *reinterpret_cast<const uint4*>(inh + (long long)rl * CQ + c);
            tv[c / 4] = bf16_unpack4(make_uint2(w.x, w.y));
            tv[c / 4 + 1] = bf16_unpack4(make_uint2(w.z, w.w));
          }
        } else if constexpr (kBF) {
  #pragma unroll
          for (int c = 0; c < CQ; c += 4)
            tv[c / 4] = bf16_unpack4(*reinterpret_cast<const uint2*>(inh + (long long)rl * CQ + c));
        } else if constexpr (kSrc3) {
          const float* gr = p_g3 + ((long long)mesh * N + vl) * 3;   // (src3_c == 3: host check)
          const float g0 = gr[0], g1 = gr[1], g2 = gr[2];
  #pragma unroll
          for (int c = 0; c < CQ; c += 4) {
            float r4[4];
  #pragma unroll
            for (int j = 0; j < 4; ++j)   // (wave-uniform weights at constant offsets: scalar loads)
              r4[j] = fmaf(g2, p_w3[(c + j) * 3 + 2], fmaf(g1, p_w3[(c + j) * 3 + 1], g0 * p_w3[(c + j) * 3]));
            tv[c / 4] = make_float4(r4[0], r4[1], r4[2], r4[3]);
          }
          if (vi == 0 && tid < a.src3_n) {  // (one divergent block in the first wave: the connected block's stored rows)
  #pragma unroll
            for (int c = 0; c < CQ; c += 4) tv[c / 4] = *reinterpret_cast<const float4*>(inb + (long long)rl * CQ + c);
          }
        } else {
  #pragma unroll
          for (int c = 0; c < CQ; c += 4) tv[c / 4] = *reinterpret_cast<const float4*>(inb + (long long)rl * CQ + c);
        }
  #pragma unroll
        for (int c = 0; c < CQ; c += 4) {
          float4 t = tv[c / 4];
          if constexpr (kMask == 2) {
            const uint32_t m = mw[c / 16] >> (2 * (c % 16));
            t.x = (m & 1u) ? t.x : 0.f;
            t.y = (m & 2u) ? t.y : 0.f;
            t.z = (m & 4u) ? t.z : 0.f;
            t.w = (m & 8u) ? t.w : 0.f;
          }
          if constexpr (kMask == 1) {
            const float4 m = *reinterpret_cast<const float4*>(mkb + (long long)vl * CQ + c);
            t.x = m.x > 0.f ? t.x : 0.f;
            t.y = m.y > 0.f ? t.y : 0.f;
            t.z = m.z > 0.f ? t.z : 0.f;
            t.w = m.w > 0.f ? t.w : 0.f;
          }
          xs[vi][c] = t.x * s;
          xs[vi][c + 1] = t.y * s;
          xs[vi][c + 2] = t.z * s;
          xs[vi][c + 3] = t.w * s;
        }
      } else {
  #pragma unroll
        for (int c = 0; c < CQ; ++c) {
          float t = inb[(long long)rl * CQ + c];
          if constexpr (kMask == 1) {
            if (!(mkb[(long long)vl * CQ + c] > 0.f)) t = 0.f;
          }
          xs[vi][c] = t * s;
        }
      }
    }
  };
  {
    using T = std::true_type;
    using F = std::false_type;
    using M0 = std::integral_constant<int, 0>;
    using M1 = std::integral_constant<int, 1>;
    using M2 = std::integral_constant<int, 2>;
    const int mm = use_bits ? 2 : (mkb ? 1 : 0);
    bool done = false;
    if constexpr (CQ == 16 && TCT == 1024 && BWD) {
      if (a.src3_n > 0) {  // (the host admits this only with sign bytes or no mask, no row map, fp32 rows)
        if (mm == 2) load_rows(F{}, M2{}, F{}, T{});
        else load_rows(F{}, M0{}, F{}, T{});
        done = true;
      }
    }
    if constexpr (CQ % 4 == 0) {
      if (a.in_bf16 && !done) {  // (bf16 rows come with sign bytes or no mask: the host refuses an fp32 mask)
        if (p_in_map) {
          if (mm == 2) load_rows(T{}, M2{}, T{}, F{});
          else load_rows(T{}, M0{}, T{}, F{});
        } else {
          if (mm == 2) load_rows(F{}, M2{}, T{}, F{});
          else load_rows(F{}, M0{}, T{}, F{});
        }
      }
    }
    if ((CQ % 4 != 0 || !a.in_bf16) && !done) {
      if (p_in_map) {
        if (mm == 2) load_rows(T{}, M2{}, F{}, F{});
        else if (mm == 1) load_rows(T{}, M1{}, F{}, F{});
        else load_rows(T{}, M0{}, F{}, F{});
      } else {
        if (mm == 2) load_rows(F{}, M2{}, F{}, F{});
        else if (mm == 1) load_rows(F{}, M1{}, F{}, F{});
        else load_rows(F{}, M0{}, F{}, F{});
      }
    }
  }

  MVH_STAMPX(2);
  // acc[vi] += xs[vi][:] . W_k[:, slab]   (weights are wave-uniform: scalar loads, SGPR operands)
  const bool slab_full = s0 + 4 <= a.CO;
  const float* __restrict__ wslab = p_W + (long long)(s0 >> 2) * a.K * CQ * 4;
  // On the matrix pipe: v_mfma_f32_4x4x1 (16 independent 4x4 outer products, K = 1) with A = W_k[c][0..3]
  // replicated in every block (lane l supplies W_k[c][l & 3]) and B = the lane's own x[c]: block b's result
  // column j -- the four output channels of the vertex of lane 4 b + j -- lands in that very lane's accumulator,
  // i.e. the thread-per-vertex layout needs no shuffles.  One instruction does the work of four v_fma on a pipe
  // that runs beside the VALU, and the weights arrive as ONE round of CQ vector loads per order (4 distinct
  // addresses per wave) instead of up to 7 dependent scalar-load rounds when CQ * 4 exceeds the SGPR file.
  typedef float v4f __attribute__((ext_vector_type(4)));
  const float* __restrict__ wlane = wslab + (tid & 3);
  // (Small levels only: at the 5k level the CQ extra VGPRs of the weight column spill in the 1024 x 5 shape
  //  -- 49.8 vs 42.7 us -- so those variants keep the scalar-operand v_fma form.)
  // The weight column of the NEXT order is fetched right after the current one has been used, so its latency sits
  // under the gathers / the barrier instead of in front of the matrix instructions.
  float wv[TCT == 0 ? CQ : 1];
  auto load_w = [&](int k) {
    if constexpr (TCT == 0) {
#pragma unroll
      for (int c = 0; c < CQ; ++c) wv[c] = wlane[(k * CQ + c) * 4];
    }
  };
  auto contract = [&](float4(&acc)[VPT], int k) {
    if constexpr (TCT == 0) {
#pragma unroll
      for (int vi = 0; vi < VPT; ++vi) {
        v4f t = {acc[vi].x, acc[vi].y, acc[vi].z, acc[vi].w};
#pragma unroll
        for (int c = 0; c < CQ; ++c) t = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[c], xs[vi][c], t, 0, 0, 0);
        acc[vi] = make_float4(t[0], t[1], t[2], t[3]);
      }
      if (k > 0) load_w(k - 1);
    } else {
      // p_W is the slab-packed copy [slab][k][c][4] (k_pack_w): contiguous, so one order's
      // weights arrive in a few wide s_load instead of CQ*4 dependent scalar loads
      if constexpr (CQ % 2 == 0) {
        f2v lo[VPT], hi[VPT];
#pragma unroll
        for (int vi = 0; vi < VPT; ++vi) { lo[vi] = (f2v){acc[vi].x, acc[vi].y}; hi[vi] = (f2v){acc[vi].z, acc[vi].w}; }
#pragma unroll
        for (int c = 0; c < CQ; c += 2) {
          const float* w = wslab + (k * CQ + c) * 4;
#pragma unroll
          for (int vi = 0; vi < VPT; ++vi) fma4x2(lo[vi], hi[vi], (f2v){xs[vi][c], xs[vi][c + 1]}, w);
        }
#pragma unroll
        for (int vi = 0; vi < VPT; ++vi) acc[vi] = make_float4(lo[vi].x, lo[vi].y, hi[vi].x, hi[vi].y);
      } else {
#pragma unroll
        for (int c = 0; c < CQ; ++c) {
          const float* w = wslab + (k * CQ + c) * 4;
#pragma unroll
          for (int vi = 0; vi < VPT; ++vi) {
            acc[vi].x = fmaf(xs[vi][c], w[0], acc[vi].x);
            acc[vi].y = fmaf(xs[vi][c], w[1], acc[vi].y);
            acc[vi].z = fmaf(xs[vi][c], w[2], acc[vi].z);
            acc[vi].w = fmaf(xs[vi][c], w[3], acc[vi].w);
          }
        }
      }
    }
  };
  // acc[vi] += (scale * -2/deg) * sum_{j in N(v)} slab[j]
  auto gather_axpy = [&](float4(&acc)[VPT], float scale, const float4* slab) {  // (shadows the kernel's `slab`)
#pragma unroll
    for (int vi = 0; vi < VPT; ++vi) {
      const int v = tid + vi * THREADS;
      float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int q = 0; q < PW / 4; ++q) {
        uint4 id;
        if constexpr (kDB) id = ids[vi][q];
        else id = ellv[v * (PW / 4) + q];
        {
          const float4 n0 = slab[id.x & 0xffffu], n1 = slab[id.x >> 16];
          const float4 n2 = slab[id.y & 0xffffu], n3 = slab[id.y >> 16];
          add4(g, n0);
          add4(g, n1);
          add4(g, n2);
          add4(g, n3);
        }
        asm volatile("" ::: "memory");
        {
          const float4 n0 = slab[id.z & 0xffffu], n1 = slab[id.z >> 16];
          const float4 n2 = slab[id.w & 0xffffu], n3 = slab[id.w >> 16];
          add4(g, n0);
          add4(g, n1);
          add4(g, n2);
          add4(g, n3);
        }
        asm volatile("" ::: "memory");
      }
      if constexpr (kOvf) {
        if (ovf_any[vi]) {  // columns 8..11 of a long row, after the ELL slots (same order as a 16-wide list)
          const float4 n0 = slab[ovf0[vi] & 0xffffu], n1 = slab[ovf0[vi] >> 16];
          const float4 n2 = slab[ovf1[vi] & 0xffffu], n3 = slab[ovf1[vi] >> 16];
          add4(g, n0);
          add4(g, n1);
          add4(g, n2);
          add4(g, n3);
        }
      }
      const float kk = ka2[vi] * scale;
      acc[vi].x = fmaf(kk, g.x, acc[vi].x);
      acc[vi].y = fmaf(kk, g.y, acc[vi].y);
      acc[vi].z = fmaf(kk, g.z, acc[vi].z);
      acc[vi].w = fmaf(kk, g.w, acc[vi].w);
    }
  };

  float4* stage = slab;  // where a fused pooling parks the result rows
  load_w(a.K - 1);
  if (a.K >= 2) {
    contract(R, a.K - 1);
#pragma unroll
    for (int vi = 0; vi < VPT; ++vi) {
      slab[tid + vi * THREADS] = R[vi];         // u_{K-1} (zero in the slots past N)
      if constexpr (kDB) slabB[tid + vi * THREADS] = make_float4(0.f, 0.f, 0.f, 0.f);  // u_K = 0
      R[vi] = make_float4(0.f, 0.f, 0.f, 0.f);  // u_K = 0
    }
    MVH_STAMPX(3);
    __syncthreads();  // slab (+ ELL image) staged
    MVH_STAMPX(4);
    // Waves w and w + NW/2 share a SIMD: the second half gathers first and contracts after, so at any
    // time one partner is on the VALU (weight FMAs) while the other waits on LDS gathers
    // (MI355X_MICROARCH "two waves per SIMD": split roles by wave number >= NW/2, not by parity).
    const bool gather_first = (tid >> 6) >= (THREADS >> 7) && THREADS >= 128;
    if constexpr (kDB) {
      float4* cur = slab;   // u_{k+1}, gathered by everyone
      float4* oth = slabB;  // u_{k+2}, touched only through the thread's own rows
      for (int k = a.K - 2; k >= 0; --k) {
#pragma unroll
        for (int vi = 0; vi < VPT; ++vi) {
          const float4 o = oth[tid + vi * THREADS];
          R[vi] = make_float4(-o.x, -o.y, -o.z, -o.w);
        }
        const float sc = (k == 0) ? 0.5f : 1.0f;
        if (gather_first) {
          gather_axpy(R, sc, cur);
          contract(R, k);
        } else {
          contract(R, k);
          gather_axpy(R, sc, cur);
        }
        if (k == 0) break;  // R is the result; `oth` is free for the epilogue
#pragma unroll
        for (int vi = 0; vi < VPT; ++vi) oth[tid + vi * THREADS] = R[vi];  // own rows only: no barrier before
        __syncthreads();
        float4* t = cur;
        cur = oth;
        oth = t;
      }
      stage = oth;
    } else {
      for (int k = a.K - 2; k >= 1; --k) {
#pragma unroll
        for (int vi = 0; vi < VPT; ++vi) R[vi] = make_float4(-R[vi].x, -R[vi].y, -R[vi].z, -R[vi].w);
        if (gather_first) {
          gather_axpy(R, 1.0f, slab);
          contract(R, k);
        } else {
          contract(R, k);
          gather_axpy(R, 1.0f, slab);
        }
        MVH_STAMPX(5 + 3 * (a.K - 2 - k));
        __syncthreads();  // every gather of u_{k+1} is done
        MVH_STAMPX(6 + 3 * (a.K - 2 - k));
#pragma unroll
        for (int vi = 0; vi < VPT; ++vi) {
          const int v = tid + vi * THREADS;
          const float4 old = slab[v];
          slab[v] = R[vi];
          R[vi] = old;
        }
        __syncthreads();
        MVH_STAMPX(7 + 3 * (a.K - 2 - k));
      }
#pragma unroll
      for (int vi = 0; vi < VPT; ++vi) R[vi] = make_float4(-R[vi].x, -R[vi].y, -R[vi].z, -R[vi].w);
      if (gather_first) {
        gather_axpy(R, 0.5f, slab);
        contract(R, 0);
      } else {
        contract(R, 0);
        gather_axpy(R, 0.5f, slab);
      }
    }
  } else {
    contract(R, 0);
  }

  MVH_STAMPX(26);
  // ---- epilogue: unscale (1/s = sqrt(deg)), bias, activation, one store per vertex
  float bj[4] = {0.f, 0.f, 0.f, 0.f};
  if (!BWD && p_bias) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (s0 + j < a.CO) bj[j] = p_bias[s0 + j];
  }
  float* outb = p_out + (long long)mesh * a.out_bs * a.CO;
  const bool vec_store = slab_full && (a.CO % 4 == 0);
  // fused pooling of the result by a general CSR operator (nn/pool.py U forward, U^T backward): the
  // rows also go to the slab and every thread then gathers its pooled rows from LDS in the
  // operator's CSR order (the arithmetic of k_spmm<.., EXACT>).  Backward: ONLY the pooled rows are
  // stored (to `out`; no [B, N, C] gradient tensor); forward: `out` as usual + pooled rows to `pooled`.
  const bool scatter = p_pt_rowptr != nullptr;
  // (single slab: wait for the last gathers of u_1; two slabs: the rows go to the one nobody gathers from)
  if (scatter && !(kDB && a.K >= 2)) __syncthreads();
#pragma unroll
  for (int vi = 0; vi < VPT; ++vi) {
    const int v = tid + vi * THREADS;
    if (BWD && scatter && v < N) {
      const float inv_s = ka2[vi] < 0.f ? __builtin_amdgcn_rsqf(-0.5f * ka2[vi]) : 1.0f;
      stage[v] = make_float4(R[vi].x * inv_s, R[vi].y * inv_s, R[vi].z * inv_s, R[vi].w * inv_s);
      continue;
    }
    if (v >= N) continue;
    // -0.5 * ka2 = 1/deg  ->  rsq(1/deg) = sqrt(deg)
    const float inv_s = ka2[vi] < 0.f ? __builtin_amdgcn_rsqf(-0.5f * ka2[vi]) : 1.0f;
    float o[4] = {fmaf(R[vi].x, inv_s, bj[0]), fmaf(R[vi].y, inv_s, bj[1]), fmaf(R[vi].z, inv_s, bj[2]),
                  fmaf(R[vi].w, inv_s, bj[3])};
    if (!BWD && a.act == MVH_ACT_RELU) {
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = fmaxf(o[j], 0.f);
    }
    const int pr = p_pool_inv ? p_pool_inv[v] : -1;  // fused one-hot downsampling (nn/pool.py D)
    const bool dead = a.out_dead && pr < 0;          // a row nobody reads: neither its values nor its sign byte
    if (!BWD && p_bits_out && !dead)  // CO % 4 == 0 (checked on the host): one sign byte per (vertex, slab)
      p_bits_out[((long long)mesh * a.out_bs + v) * (a.CO >> 2) + (s0 >> 2)] =
          (uint8_t)((o[0] > 0.f ? 1 : 0) | (o[1] > 0.f ? 2 : 0) | (o[2] > 0.f ? 4 : 0) | (o[3] > 0.f ? 8 : 0));
    if (!BWD && scatter) stage[v] = make_float4(o[0], o[1], o[2], o[3]);
    float* dst = outb + (long long)v * a.CO + s0;
    float* pdst = p_pooled + ((long long)mesh * a.pooled_bs + max(pr, 0)) * a.CO + s0;
    if (vec_store) {  // (bf16 storage: the host guarantees CO % 4 == 0, i.e. this branch)
      if (!a.out_dead) store4_any(p_out, ((long long)mesh * a.out_bs + v) * a.CO + s0, a.out_bf16 != 0, o[0], o[1], o[2], o[3]);
      if (pr >= 0)
        store4_any(p_pooled, ((long long)mesh * a.pooled_bs + pr) * a.CO + s0, a.pooled_bf16 != 0, o[0], o[1], o[2], o[3]);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (s0 + j < a.CO) {
          if (!a.out_dead) dst[j] = o[j];
          if (pr >= 0) pdst[j] = o[j];
        }
    }
  }
  MVH_STAMPX(27);
  if (scatter) {
    __syncthreads();
    for (int c = tid; c < a.pt_rows; c += THREADS) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      const int e0 = p_pt_rowptr[c], e1 = p_pt_rowptr[c + 1];
      // four taps per round: their (col, val) loads and LDS reads are issued together (a tap-by-tap loop
      // is one dependent global round trip per tap, ~12 of them per row of U^T); taps past the row
      // end are clamped to a valid entry and given weight 0, so the sums stay in CSR order
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
#ifdef MVH_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (diagnostic build: the stores have left the wave)
#endif
  MVH_STAMPX(28);
}

#ifdef MVH_STAMP
MVH_STAMP_READER(mvh_debug_read_stamps_lds)
#endif

// Wp[slab][k][c][j] = W[k][c][4 slab + j] (forward) or W[k][4 slab + j][c] (backward, W^T);
// entries past the last output channel are zero.  K*Cin*Cout <= ~10k floats: one tiny launch.
__global__ void __launch_bounds__(256)
k_pack_w(const float* __restrict__ W, float* __restrict__ Wp, int K, int Cin, int Cout, int CQ, int CO, int bwd) {
  const int NS = (CO + 3) >> 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= NS * K * CQ * 4) return;
  const int j = i & 3, c = (i >> 2) % CQ, k = (i >> 2) / CQ % K, sl = (i >> 2) / CQ / K;
  const int o = sl * 4 + j;
  float v = 0.f;
  if (o < CO) v = bwd ? W[((long long)k * Cin + o) * Cout + c] : W[((long long)k * Cin + c) * Cout + o];
  Wp[i] = v;
}

// All convolution layers of a model in ONE launch (weights are constant within a step):
// blockIdx.y selects the table entry.
__global__ void __launch_bounds__(256) k_pack_all(PackTable t) {
  const PackEntry e = t.e[blockIdx.y];
  const int NS = (e.CO + 3) >> 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (e.bwd >= 3) {  // bf16 slabs of the level-0 matrix-pipe kernel
    if (i < l0h_pack_dwords_hd(e.K)) reinterpret_cast<uint32_t*>(e.dst)[i] = pack_l0h_dword(e.W, e.K, e.bwd == 4, i);
    return;
  }
  if (e.bwd == 2) {  // W_eff = sum_k T_k(0) W_k (isolated vertices of the split path): T_k(0) = 1, 0, -1, 0, ...
    if (i >= e.Cin * e.Cout) return;
    float s = 0.f;
    for (int k = 0; k < e.K; k += 2) s += ((k & 2) ? -1.f : 1.f) * e.W[(long long)k * e.Cin * e.Cout + i];
    e.dst[i] = s;
    return;
  }
  if (i >= NS * e.K * e.CQ * 4) return;
  const int j = i & 3, c = (i >> 2) % e.CQ, k = (i >> 2) / e.CQ % e.K, sl = (i >> 2) / e.CQ / e.K;
  const int o = sl * 4 + j;
  float v = 0.f;
  if (o < e.CO)
    v = e.bwd ? e.W[((long long)k * e.Cin + o) * e.Cout + c] : e.W[((long long)k * e.Cin + c) * e.Cout + o];
  e.dst[i] = v;
}

int pack_entry_floats(int Cin, int Cout, int K, bool bwd) {
  const int CQ = bwd ? Cout : Cin, CO = bwd ? Cin : Cout;
  return ((CO + 3) / 4) * K * CQ * 4;
}

int launch_pack_all(hipStream_t st, const PackTable& t) {
  if (t.n == 0) return MVH_OK;
  int mx = 0;
  for (int i = 0; i < t.n; ++i) {
    const int nf = t.e[i].bwd >= 3 ? l0h_pack_dwords(t.e[i].K)
                   : t.e[i].bwd == 2 ? t.e[i].Cin * t.e[i].Cout : ((t.e[i].CO + 3) / 4) * t.e[i].K * t.e[i].CQ * 4;
    if (nf > mx) mx = nf;
  }
  hipLaunchKernelGGL(k_pack_all, dim3(cdiv(mx, 256), t.n), dim3(256), 0, st, t);
  MVH_LAUNCH_CHECK();
  return MVH_OK;
}

template <int CQ, int VPT, int TCT, int PW, bool BWD>
static int launch_one(hipStream_t st, const LdsConvArgs& a, int threads) {
  auto kern = k_cheb_lds<CQ, VPT, TCT, PW, BWD>;
  const size_t lds = (size_t)VPT * threads * (TCT == 0 ? 32 : 16 + PW * 4);
  static LdsAttr attr;
  if (int rc = attr.ensure(reinterpret_cast<const void*>(kern), lds)) return rc;
  const int NS = (a.CO + 3) / 4;
  const int grid = ((a.B + 7) / 8) * 8 * NS;
  LdsConvDims d{a.B, a.N, a.K, a.CO, a.Cin, a.Cout, a.pairs, a.act, a.in_bs, a.out_bs, a.mask_bs, a.pooled_bs,
                a.mask_bits, a.pt_rows, a.ovf, a.in_bf16, a.out_bf16, a.pooled_bf16, a.out_dead, a.src3_n, a.src3_c};
  hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, st, a.in, a.mask, a.W, a.bias, a.out, a.rowinfo, a.ell,
                     a.in_map, a.pool_inv, a.pooled, a.bits_out, a.pt_rowptr, a.pt_col, a.pt_val, a.col, a.g3, a.w3, d);
  MVH_LAUNCH_CHECK();
  return MVH_OK;
}

template <int CQ, int VPT, int TCT>
static int launch_cfg(hipStream_t st, const LdsConvArgs& a, bool bwd, int threads) {
  if (a.pairs > 4)
    return bwd ? launch_one<CQ, VPT, TCT, 8, true>(st, a, threads) : launch_one<CQ, VPT, TCT, 8, false>(st, a, threads);
  return bwd ? launch_one<CQ, VPT, TCT, 4, true>(st, a, threads) : launch_one<CQ, VPT, TCT, 4, false>(st, a, threads);
}

template <int CQ>
static int launch_cq(hipStream_t st, const LdsConvArgs& a, bool bwd, int vpt, int threads) {
  if (vpt == 1) return launch_cfg<CQ, 1, 0>(st, a, bwd, threads);
  if (vpt == 2) return launch_cfg<CQ, 2, 0>(st, a, bwd, threads);
  if constexpr (CQ <= 16) {
    if (vpt == 10 && threads == 512) return launch_cfg<CQ, 10, 512>(st, a, bwd, threads);
    if (vpt == 5 && threads == 1024) return launch_cfg<CQ, 5, 1024>(st, a, bwd, threads);
  }
  return -1;
}

// Returns MVH_OK and sets *handled when the LDS-resident kernel ran; *handled == false means
// "not eligible, use the general pipeline".
int try_cheb_lds(hipStream_t st, const mvh_csr_t* lap, const float* in, const float* mask, const float* W,
                 const float* bias, float* out, int B, int N, int Cin, int Cout, int K, int act, bool bwd,
                 float* wpack, bool* handled, const LdsConvOpts& o) {
  *handled = false;
  if (o.in_bf16) {  // the 5k level's 16 -> 16 layer on bf16 rows: packed registers + matrix-pipe contraction
    if (int rc = try_cheb_l0h(st, lap, in, o.mask_bits, W, bias, out, B, N, Cin, Cout, K, act, bwd, wpack, handled, o)) return rc;
    if (*handled) return MVH_OK;
  }
  const float* prepacked = o.prepacked;
  if (!wpack && !prepacked) return MVH_OK;
  if (dbg().force_generic) return MVH_OK;
  const int need = MVH_CSR_NORMALIZED_LAPLACIAN | MVH_CSR_SYMMETRIC;
  if (!lap->rowinfo || !lap->ell || lap->ell_pairs <= 0 || lap->ell_pairs > 8 || (lap->flags & need) != need)
    return MVH_OK;
  const int CQ = bwd ? Cout : Cin, CO = bwd ? Cin : Cout;
  if (N < 1 || N + 1 >= 65535) return MVH_OK;
  if (CQ != 3 && CQ != 8 && CQ != 16 && CQ != 32) return MVH_OK;
  if (((uintptr_t)in | (uintptr_t)mask | (uintptr_t)out) % 16 != 0) return MVH_OK;
  // vertex slots: one thread per vertex while a block of <= 1024 threads covers the mesh (many
  // waves hide the LDS latency), two per thread up to 2047 vertices, 512 x 10 for the 5k level
  int vpt, threads;
  if (N + 1 <= 1024) { vpt = 1; threads = ((N + 1 + 63) / 64) * 64; }
  else if (N + 1 <= 2048) { vpt = 2; threads = (((N + 2) / 2 + 63) / 64) * 64; }
  else if (N + 1 <= 5120 && CQ <= 16 && !(lap->flags & MVH_CSR_ELL_OVERFLOW)) {
    // 1024 threads x 5 vertices (4 waves/SIMD, 128 VGPRs) measured 1-5 % faster per step than 512 x 10
    // (2 waves/SIMD, 256 VGPRs) in both directions; the debug switch l0_wide selects the latter for A/B runs
    // (the GPU tests pass under both)
    if (dbg().l0_wide) { vpt = 10; threads = 512; } else { vpt = 5; threads = 1024; }
  }
  else return MVH_OK;
  const int pw = lap->ell_pairs > 4 ? 8 : 4;
  if ((size_t)vpt * threads * (16 + pw * 4) > 160 * 1024) return MVH_OK;
  const int n_pack = ((CO + 3) / 4) * K * CQ * 4;
  if ((size_t)n_pack * sizeof(float) > kLdsWpackBytes) return MVH_OK;

  LdsConvArgs a;
  a.mask_bits = 0; a.bits_out = nullptr;
  a.in_bf16 = o.in_bf16 ? 1 : 0; a.out_bf16 = o.out_bf16 ? 1 : 0; a.pooled_bf16 = o.pooled_bf16 ? 1 : 0;
  a.out_dead = (o.out_dead && o.pool_inv && !bwd) ? 1 : 0;
  a.g3 = nullptr; a.w3 = nullptr; a.src3_n = 0; a.src3_c = 0;
  if (o.src3_g) {  // lazy input rows: the 5k level's fp32 dX kernel only -- anything else must not pretend
    const bool ok = bwd && CQ == 16 && !dbg().l0_wide && N + 1 > 2048 && N + 1 <= 5120 && !o.in_bf16 && !o.in_map &&
                    (o.in_bs == 0 || o.in_bs == N) && (!mask || o.mask_bits) && o.src3_w && o.src3_c == 3 &&
                    o.src3_n >= 1 && o.src3_n <= 1024 && !(lap->flags & MVH_CSR_ELL_OVERFLOW);
    if (!ok) return fail(MVH_ERR_UNSUPPORTED, "cheb_lds: lazy output-gradient rows (src3) on a layer without that kernel");
    a.g3 = o.src3_g; a.w3 = o.src3_w; a.src3_n = o.src3_n; a.src3_c = o.src3_c;
  }
  // bf16 rows are read / written in 4-channel words, and a ReLU mask comes as sign bytes (never the fp32 output)
  if (o.in_bf16 && (CQ % 4 != 0 || (mask && !o.mask_bits))) return MVH_OK;
  if ((o.out_bf16 || o.pooled_bf16) && CO % 4 != 0) return MVH_OK;
  if (o.mask_bits) {  // sign bytes take the place of the float mask
    if (!bwd || CQ % 4 != 0) return MVH_OK;
    mask = reinterpret_cast<const float*>(o.mask_bits);
    a.mask_bits = 1;
  }
  if (o.bits_out) {
    if (bwd || CO % 4 != 0) return MVH_OK;
    a.bits_out = o.bits_out;
  }
  a.in = in; a.mask = mask; a.W = prepacked ? prepacked : wpack; a.bias = bias; a.out = out;
  a.rowinfo = lap->rowinfo; a.ell = lap->ell;
  a.col = lap->col; a.ovf = (lap->flags & MVH_CSR_ELL_OVERFLOW) ? 1 : 0;
  a.B = B; a.N = N; a.K = K; a.CO = CO; a.Cin = Cin; a.Cout = Cout;
  a.pairs = lap->ell_pairs; a.act = act;
  a.in_bs = o.in_bs > 0 ? o.in_bs : N;
  a.out_bs = o.out_bs > 0 ? o.out_bs : N;
  a.mask_bs = o.mask_bs > 0 ? o.mask_bs : a.in_bs;
  a.in_map = o.in_map; a.pool_inv = o.pool_inv; a.pooled = o.pooled; a.pooled_bs = o.pooled_bs;
  if (o.pool_inv && (!o.pooled || ((uintptr_t)o.pooled % 16) != 0)) return MVH_OK;
  a.pt_rowptr = nullptr; a.pt_col = nullptr; a.pt_val = nullptr; a.pt_rows = 0;
  if (o.out_pool_t) {  // `out` is the pooled [B][n_rows][CO] buffer
    const mvh_csr_t* pt = o.out_pool_t;
    if (CO % 4 != 0 || pt->n_cols != N || !pt->rowptr || !pt->col || !pt->val || o.pool_inv) return MVH_OK;
    if (!bwd && (!o.pooled || ((uintptr_t)o.pooled % 16) != 0)) return MVH_OK;
    a.pt_rowptr = pt->rowptr; a.pt_col = pt->col; a.pt_val = pt->val; a.pt_rows = pt->n_rows;
    if (bwd) a.out_bs = pt->n_rows;
    else { a.pooled = o.pooled; a.pooled_bs = pt->n_rows; }
  }
  if (o.dry_run) {  // eligibility probe only: nothing is launched
    *handled = true;
    return MVH_OK;
  }
  if (!prepacked) {  // slab-packed weights for the scalar loads of the main kernel
    hipLaunchKernelGGL(k_pack_w, dim3(cdiv(n_pack, 256)), dim3(256), 0, st, W, wpack, K, Cin, Cout, CQ, CO, bwd ? 1 : 0);
    MVH_LAUNCH_CHECK();
  }
  // (MEASURED, not kept -- round 3, tools/scratch/cheb_l0m.hip.txt: this kernel with the contraction on the matrix pipe and the
  //  weight slab [K][16][4] in the LDS slots nobody uses (rows N + 1 .. of the slab and of the ELL image), read four values at
  //  a time: parity-green, 28 .. 34 spilled VGPRs, 49.3 us per forward launch against 42.3.  v_mfma_f32_4x4x1 is an 8-cycle
  //  instruction that holds the SIMD's vector issue for its whole length (MI355X_MICROARCH: 8 of an MFMA's cycles), so 16 of
  //  them per vertex and order cost the VALU exactly what the 64 v_fma cost (2 cycles each): nothing moves off the pipe that
  //  binds this kernel (SQ_ACTIVE_INST_VALU 76 %).  The bf16 form wins because 4x4x4 does four times the work per issue.)
  // (MEASURED, not kept -- round 3, history: cheb_l0f.hip: the bf16 kernel's layout in fp32 -- two slabs, no ELL image, one
  //  barrier per order -- with the neighbour ids re-read from the L2-resident ELL table every order because 80 VGPRs of
  //  fp32 rows leave no room for them: 6 .. 13 spilled VGPRs instead of 29 .. 119, but 58.2 us per forward launch against
  //  42.8 with one id load in flight ahead and 76.7 with three: a global load inside the gather loop costs more than the
  //  ELL image and the second barrier it replaces.)
  int rc = -1;
  if (CQ == 3) rc = launch_cq<3>(st, a, bwd, vpt, threads);
  else if (CQ == 8) rc = launch_cq<8>(st, a, bwd, vpt, threads);
  else if (CQ == 16) rc = launch_cq<16>(st, a, bwd, vpt, threads);
  else rc = launch_cq<32>(st, a, bwd, vpt, threads);
  if (rc < 0) return fail(MVH_ERR_UNSUPPORTED, "cheb_lds: no kernel for vpt=%d threads=%d", vpt, threads);
  if (rc == MVH_OK) *handled = true;
  return rc;
}

}  // namespace mvh
