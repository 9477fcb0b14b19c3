// LDS-resident weight gradient of ChebConv:  dW_k[ci][co] = sum_{b,v} T_k(L) x [b,v,ci] * dpre[b,v,co]
// (the reference gets it from autograd through K matmuls, nn/conv.py:559-571).
//
// One side ("P", the one with FEWER channels) runs the Chebyshev recurrence, the other ("Q")
// stays fixed:  dW_k = sum_v Q[v][q] * T_k(P)[v][p]   (L is symmetric, so T_k may act on either).
// A workgroup owns (mesh, slab of 4 P-channels):
//   * LDS   : t_k of the slab for every vertex as one float4 + the padded ELL neighbour lists --
//             the same 160 KB image as the forward kernel (cheb_lds.hip), in scaled variables
//             t~ = D^-1/2 T so the edge list needs no values;
//   * recurrence: thread-owns-vertex, unweighted ds_read_b128 gathers, in-place swap of the
//             previous order through registers, two barriers per order;
//   * contraction over vertices on the MATRIX pipe: v_mfma_f32_4x4x1_16B_f32 with one block
//             per vertex (4 lanes per vertex): A = 4 Q-channels of the vertex (pre-divided by s),
//             B = its 4 slab values read back from LDS, so 16 vertices x (4x4) products per
//             instruction accumulate straight into the dW tile -- no cross-lane reduction until
//             the end of an order, and the VALU/LDS pipes stay free for the next order's gather;
//   * Q rows live in VGPRs in that 4-lanes-per-vertex layout (read from HBM once per slab,
//             1 KB contiguous per wave load).
// Per (mesh, slab, wave, order) a 16 x 4 partial tile goes to a workspace; a second launch sums
// the partials in fixed order (bitwise reproducible, no atomics) and scatters them into dW / db.
#include "common.hpp"
#include "bf16.hpp"

#include <type_traits>

namespace mvh {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct DwDims {
  int B, N, K, CP, CQtot, pairs, db_mode;  // db_mode: 0 none, 1 = column sums of Q, 2 = of P
  int bs;                                  // rows per mesh in the P/Q/mask buffers (>= N)
  int map_side, map_bs;                    // fused un-pooling of dout: 1 = P rows, 2 = Q rows come through p_map
                                           // from a compact buffer of map_bs rows per mesh (masks stay full-size)
  int mask_bits;                           // the mask pointer holds ReLU sign bytes (one per vertex and 4 channels)
  int ovf;                                 // MVH_CSR_ELL_OVERFLOW: rows longer than 8 continue in the CSR columns
  int p_bf16, q_bf16;                      // the P / Q tensor is stored as bf16 (bf16.hpp); masks are sign bytes then
  int src3_n, src3_c;                      // > 0: P rows >= src3_n are p_g3[v][0..src3_c) W3^T (ConvIO::src3_*; P = dout, fp32)
  int mesh0;                               // first mesh of this launch (ConvIO::dw_split: the batch in several launches)
};

__device__ __forceinline__ void add4f(float4& a, const float4& b) {
  a.x += b.x;
  a.y += b.y;
  a.z += b.z;
  a.w += b.w;
}

__device__ __forceinline__ float xor_add(float v, int mask) { return v + __shfl_xor(v, mask, 64); }

// v + (v rotated right by N lanes inside each row of 16): a VALU DPP op, no LDS round trip
template <int N>
__device__ __forceinline__ float row_ror_add(float v) {
  const int r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + N, 0xf, 0xf, false);
  return v + __int_as_float(r);
}

// sum over the 16 lanes that share (lane & 3): two in-row DPP rotations, then the 4 rows
__device__ __forceinline__ float sum_blocks(float x) {
  x = row_ror_add<4>(x);
  x = row_ror_add<8>(x);
  x = xor_add(x, 16);
  x = xor_add(x, 32);
  return x;
}

// part layout: [slab][mesh][wave][K+1][CQ][4]
template <int CQ, int VPT, int TCT, int PW>
__global__ void __launch_bounds__(TCT > 0 ? TCT : 1024)
k_cheb_dw_lds(const float* __restrict__ p_P, const float* __restrict__ p_Pmask, const float* __restrict__ p_Q,
              const float* __restrict__ p_Qmask, const uint32_t* __restrict__ p_rowinfo,
              const uint32_t* __restrict__ p_ell, float* __restrict__ p_part, const int32_t* __restrict__ p_map,
              const int* __restrict__ p_col, const float* __restrict__ p_g3, const float* __restrict__ p_w3, DwDims a) {
  // CQ % 8 == 0: the workgroup holds all CQ channels of Q (16 per float4-per-lane register group);
  // CQ == 4   : "Q-split" mode for P sides of <= 4 channels (cheb.0 and the final layer): the
  //             a.CQtot channels of Q are split over CQtot/4 workgroups (one channel per lane,
  //             one MFMA per step), so those layers still fill the chip instead of 1 WG per mesh.
  static_assert(CQ == 4 || CQ % 8 == 0, "unsupported Q width");
  constexpr bool SPLIT = (CQ == 4);
  constexpr int QH = SPLIT ? 1 : (CQ + 15) / 16;  // float4 Q registers per vertex step per lane
  const int THREADS = TCT > 0 ? TCT : (int)blockDim.x;
  const int VS = VPT * THREADS;
  const int NW = THREADS / 64;
  constexpr int STEPS_CT = TCT > 0 ? (VPT * TCT / 16) / (TCT / 64) : VPT * 4;  // 16-vertex steps per wave
  extern __shared__ __align__(16) unsigned char smem[];
  float4* slab = reinterpret_cast<float4*>(smem);   // [VS]; rows >= N stay zero
  uint4* ellv = reinterpret_cast<uint4*>(slab + VS);  // [VS][PW/4]   (kDB: the second slab instead)
  // Small levels (TCT == 0): two slabs (t~_{k-1} gathered by everyone, t~_{k-2} touched only through the
  // thread's own rows and overwritten in place by t~_k) and the neighbour ids in VGPRs, as in
  // cheb_lds.hip: one barrier per order instead of two, no ELL image, no staging pass.
  constexpr bool kDB = (TCT == 0) && (PW == 4);
  float4* slabB = slab + VS;

  const int NS = (a.CP + 3) >> 2;
  const int QP = SPLIT ? a.CQtot / 4 : 1, CQT = SPLIT ? a.CQtot : CQ;
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int per_mesh = NS * QP, rem = jj % per_mesh;
  const int mesh = a.mesh0 + (jj / per_mesh) * 8 + xcd, sl = rem / QP, s0 = sl * 4, q0 = 4 * (rem % QP);
  if (mesh >= a.B) return;
  const int tid = threadIdx.x, N = a.N, lane = tid & 63, wave = tid >> 6;
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

  MVH_STAMPX(1);
  // ---- Q rows in the MFMA layout: lane (b = lane>>2, i = lane&3) of step s holds
  //      Q[v][16h + 4i .. +3] / s_v  for vertex v = 16 (s NW + wave) + b
  float4 qreg[SPLIT ? 1 : STEPS_CT][QH];
  float qone[SPLIT ? STEPS_CT : 1];  // split mode: one channel per lane
  float4 qsum[QH];
#pragma unroll
  for (int h = 0; h < QH; ++h) qsum[h] = make_float4(0.f, 0.f, 0.f, 0.f);
  const float* Qb = p_Q + (long long)mesh * (a.map_side == 2 ? a.map_bs : a.bs) * CQT;
  const uint16_t* Qh = reinterpret_cast<const uint16_t*>(p_Q) + (long long)mesh * (a.map_side == 2 ? a.map_bs : a.bs) * CQT;
  const bool qbf = a.q_bf16 != 0, pbf = a.p_bf16 != 0;
  // element offsets of this mesh in the Q / P tensors (load*_any index from the tensor's base: the element size differs)
  const long long qoff = (long long)mesh * (a.map_side == 2 ? a.map_bs : a.bs) * CQT;
  const long long poff = (long long)mesh * (a.map_side == 1 ? a.map_bs : a.bs) * a.CP;
  const float* Qm = (p_Qmask && !a.mask_bits) ? p_Qmask + (long long)mesh * a.bs * CQT : nullptr;
  const uint8_t* Qbits = (p_Qmask && a.mask_bits)
                             ? reinterpret_cast<const uint8_t*>(p_Qmask) + (long long)mesh * a.bs * (CQT / 4) : nullptr;
  // Fast path of the train step's decoder layers (sign bytes, no row map, full channel groups): the loop
  // body is branch-free, so the scheduler can keep the loads of many steps in flight; the general loop
  // below has wave-uniform branches per step, which serialise its 40 global loads at the 5k level.
  const bool fast_q = !SPLIT && (Qbits != nullptr || Qm == nullptr) && a.map_side != 2 && (CQ % 16 == 0);
  auto load_q_fast = [&](auto bf_tag, auto bits_tag) {
    constexpr bool kBF = decltype(bf_tag)::value;
    constexpr bool kBits = decltype(bits_tag)::value;   // Q carries a ReLU mask as sign bytes (else: no mask at all)
    // the degree of a step's vertex is the same for the 4 lanes of the vertex: lane i of the quad fetches the one of step
    // s4 + i and the quad shares the four by DPP -- one row-info load per FOUR steps instead of one per step
    static_assert(STEPS_CT % 4 == 0, "steps come in groups of four");
    int deg4[STEPS_CT / 4];
#pragma unroll
    for (int s4 = 0; s4 < STEPS_CT / 4; ++s4) {
      const int vq = 16 * ((4 * s4 + (lane & 3)) * NW + wave) + (lane >> 2);
      deg4[s4] = (int)(p_rowinfo[min(vq, N - 1)] & 255u);
    }
#pragma unroll
    for (int s = 0; s < STEPS_CT; ++s) {
      const int v = 16 * (s * NW + wave) + (lane >> 2);
      const int vl = min(v, N - 1);
      int di;   // quad_perm [j, j, j, j] broadcast of lane j's value, j = s mod 4
      switch (s & 3) {
        case 0: di = __builtin_amdgcn_update_dpp(0, deg4[s >> 2], 0x00, 0xf, 0xf, false); break;
        case 1: di = __builtin_amdgcn_update_dpp(0, deg4[s >> 2], 0x55, 0xf, 0xf, false); break;
        case 2: di = __builtin_amdgcn_update_dpp(0, deg4[s >> 2], 0xAA, 0xf, 0xf, false); break;
        default: di = __builtin_amdgcn_update_dpp(0, deg4[s >> 2], 0xFF, 0xf, 0xf, false); break;
      }
      const float deg = (float)di;
      float inv_s = deg > 0.f ? __builtin_amdgcn_sqrtf(deg) : 1.0f;
      inv_s = (v < N) ? inv_s : 0.f;
      const float live = (v < N) ? 1.f : 0.f;
#pragma unroll
      for (int h = 0; h < QH; ++h) {
        const int c0 = 16 * h + 4 * (lane & 3);
        float4 t;
        if constexpr (kBF) t = bf16_unpack4(*reinterpret_cast<const uint2*>(Qh + (long long)vl * CQ + c0));
        else t = *reinterpret_cast<const float4*>(Qb + (long long)vl * CQ + c0);
        if constexpr (kBits) {
          const uint32_t m = Qbits[vl * (CQ / 4) + (c0 >> 2)];
          t.x = (m & 1u) ? t.x : 0.f;
          t.y = (m & 2u) ? t.y : 0.f;
          t.z = (m & 4u) ? t.z : 0.f;
          t.w = (m & 8u) ? t.w : 0.f;
        }
        qsum[h].x = fmaf(live, t.x, qsum[h].x);
        qsum[h].y = fmaf(live, t.y, qsum[h].y);
        qsum[h].z = fmaf(live, t.z, qsum[h].z);
        qsum[h].w = fmaf(live, t.w, qsum[h].w);
        qreg[SPLIT ? 0 : s][h] = make_float4(t.x * inv_s, t.y * inv_s, t.z * inv_s, t.w * inv_s);
      }
    }
  };
  if (fast_q) {
    if (Qbits) {
      if (a.q_bf16) load_q_fast(std::true_type{}, std::true_type{});
      else load_q_fast(std::false_type{}, std::true_type{});
    } else {
      if (a.q_bf16) load_q_fast(std::true_type{}, std::false_type{});
      else load_q_fast(std::false_type{}, std::false_type{});
    }
  } else
#pragma unroll
  for (int s = 0; s < STEPS_CT; ++s) {
    const int v = 16 * (s * NW + wave) + (lane >> 2);
    const bool valid = v < N;
    const int vl = min(v, N - 1);
    const float deg = (float)(p_rowinfo[vl] & 255u);
    float inv_s = valid ? (deg > 0.f ? __builtin_amdgcn_sqrtf(deg) : 1.0f) : 0.f;
    int ql = vl;  // Q row (through the selection map when Q = un-pooled dout)
    bool qhave = true;
    if (a.map_side == 2) {
      const int rr = p_map[vl];
      qhave = rr >= 0;
      ql = max(rr, 0);
    }
    if constexpr (SPLIT) {
      const long long off = (long long)ql * CQT + q0 + (lane & 3);
      float t = qhave ? load1_any(p_Q, qoff + off, qbf) : 0.f;
      if (Qm && !(Qm[(long long)vl * CQT + q0 + (lane & 3)] > 0.f)) t = 0.f;
      if (Qbits && !((Qbits[vl * (CQT / 4) + (q0 >> 2)] >> (lane & 3)) & 1)) t = 0.f;
      if (valid) qsum[0].x += t;
      qone[s] = t * inv_s;
      continue;
    }
#pragma unroll
    for (int h = 0; h < QH; ++h) {
      const int c0 = 16 * h + 4 * (lane & 3);
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c0 < CQ) {
        t = load4_any(p_Q, qoff + (long long)ql * CQ + c0, qbf);
        if (!qhave) t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (Qm) {
          const float4 m = *reinterpret_cast<const float4*>(Qm + (long long)vl * CQ + c0);
          t.x = m.x > 0.f ? t.x : 0.f;
          t.y = m.y > 0.f ? t.y : 0.f;
          t.z = m.z > 0.f ? t.z : 0.f;
          t.w = m.w > 0.f ? t.w : 0.f;
        }
        if (Qbits) {
          const uint32_t m = Qbits[vl * (CQ / 4) + (c0 >> 2)];
          t.x = (m & 1u) ? t.x : 0.f;
          t.y = (m & 2u) ? t.y : 0.f;
          t.z = (m & 4u) ? t.z : 0.f;
          t.w = (m & 8u) ? t.w : 0.f;
        }
      }
      if (valid) add4f(qsum[h], t);
      qreg[s][h] = make_float4(t.x * inv_s, t.y * inv_s, t.z * inv_s, t.w * inv_s);
    }
  }

  MVH_STAMPX(2);
  float* part = p_part + (((long long)sl * a.B + mesh) * NW + wave) * (long long)(a.K + 1) * CQT * 4;
  // bias gradient (plane K of the tile set, layout [q][j]): column sums of Q, written right away
  if (a.db_mode == 1 && sl == 0) {  // lane (b, i) holds q = 16h + 4i + {0..3}  (split: q0 + i)
    float* pk = part + (long long)a.K * CQT * 4;
    if constexpr (SPLIT) {
      const float x = sum_blocks(qsum[0].x);
      if (lane < 4) pk[(q0 + lane) * 4] = x;
    } else
#pragma unroll
    for (int h = 0; h < QH; ++h) {
      float v4[4] = {qsum[h].x, qsum[h].y, qsum[h].z, qsum[h].w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float x = v4[c];
        x = sum_blocks(x);
        const int q = 16 * h + 4 * lane + c;
        if (lane < 4 && q < CQ) pk[q * 4] = x;
      }
    }
  }

  MVH_STAMPX(3);
  // ---- own vertices (thread-owns-vertex layout): t~_0 = s P[:, slab]
  float ka2[VPT];
  float4 R[VPT];
  constexpr bool kOvf = (TCT == 0);  // columns 8..11 of long rows (see cheb_lds.hip), small levels only
  uint32_t ovf0[kOvf ? VPT : 1], ovf1[kOvf ? VPT : 1];
  bool ovf_any[kOvf ? VPT : 1];
  uint4 ids[kDB ? VPT : 1];
  float4 psum = make_float4(0.f, 0.f, 0.f, 0.f);
  const float* Pb = p_P + (long long)mesh * (a.map_side == 1 ? a.map_bs : a.bs) * a.CP;
  const uint16_t* Ph = reinterpret_cast<const uint16_t*>(p_P) + (long long)mesh * (a.map_side == 1 ? a.map_bs : a.bs) * a.CP;
  const float* Pm = (p_Pmask && !a.mask_bits) ? p_Pmask + (long long)mesh * a.bs * a.CP : nullptr;
  const uint8_t* Pbits = (p_Pmask && a.mask_bits)
                             ? reinterpret_cast<const uint8_t*>(p_Pmask) + (long long)mesh * a.bs * (a.CP >> 2) : nullptr;
  const bool slab_full = (s0 + 4 <= a.CP) && (a.CP % 4 == 0);
  // (as in cheb_lds.hip: the wave-uniform mode branches would fence each vertex's loads, so the loop is
  //  instantiated per mode -- 0 plain full slab, 1 fp32 mask, 2 sign bytes, 3 general (row map, partial slab))
  auto load_p = [&](auto mode_tag, auto bf_tag, auto src3_tag) {
    constexpr int kMode = decltype(mode_tag)::value;
    constexpr bool kBF = decltype(bf_tag)::value;
    constexpr bool kSrc3 = decltype(src3_tag)::value;   // lazy P rows (ConvIO::src3_*): modes 0 / 2, fp32, full slab
  #pragma unroll
    for (int vi = 0; vi < VPT; ++vi) {
      const int v = tid + vi * THREADS;
      const bool valid = v < N;
      const int vl = min(v, N - 1);
      const uint32_t rinfo = p_rowinfo[vl];
      const float deg = valid ? (float)(rinfo & 255u) : 0.f;
      ka2[vi] = deg > 0.f ? -2.0f * __builtin_amdgcn_rcpf(deg) : 0.f;
      const float s = valid ? (deg > 0.f ? __builtin_amdgcn_rsqf(deg) : 1.0f) : 0.f;
      if constexpr (kDB) {
        const unsigned padi = (unsigned)N | ((unsigned)N << 16);
        ids[vi] = valid ? reinterpret_cast<const uint4*>(p_ell)[vl] : make_uint4(padi, padi, padi, padi);
        slabB[v] = make_float4(0.f, 0.f, 0.f, 0.f);  // t~_{-1} = 0
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
      float t[4] = {0.f, 0.f, 0.f, 0.f};
      int pl = vl;
      bool phave = true;
      if constexpr (kMode == 3) {
        if (a.map_side == 1) {
          const int rr = p_map[vl];
          phave = rr >= 0;
          pl = max(rr, 0);
        }
      }
      if (kMode != 3 || slab_full) {
        float4 tv;
        if constexpr (kSrc3) {
          const float* gr = p_g3 + ((long long)mesh * N + vl) * 3;   // (src3_c == 3: host check)
          const float g0 = gr[0], g1 = gr[1], g2 = gr[2];
          float r4[4];
  #pragma unroll
          for (int j = 0; j < 4; ++j)
            r4[j] = fmaf(g2, p_w3[(s0 + j) * 3 + 2], fmaf(g1, p_w3[(s0 + j) * 3 + 1], g0 * p_w3[(s0 + j) * 3]));
          tv = make_float4(r4[0], r4[1], r4[2], r4[3]);
          if (vi == 0 && tid < a.src3_n) tv = *reinterpret_cast<const float4*>(Pb + (long long)pl * a.CP + s0);   // stored rows
        } else if constexpr (kBF) tv = bf16_unpack4(*reinterpret_cast<const uint2*>(Ph + (long long)pl * a.CP + s0));
        else tv = *reinterpret_cast<const float4*>(Pb + (long long)pl * a.CP + s0);
        if (!phave) tv = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((kMode == 1 || kMode == 3) && Pm) {
          const float4 m = *reinterpret_cast<const float4*>(Pm + (long long)vl * a.CP + s0);
          tv.x = m.x > 0.f ? tv.x : 0.f;
          tv.y = m.y > 0.f ? tv.y : 0.f;
          tv.z = m.z > 0.f ? tv.z : 0.f;
          tv.w = m.w > 0.f ? tv.w : 0.f;
        }
        if ((kMode == 2 || kMode == 3) && Pbits) {  // CP % 4 == 0 (host check)
          const uint32_t m = Pbits[vl * (a.CP >> 2) + sl];
          tv.x = (m & 1u) ? tv.x : 0.f;
          tv.y = (m & 2u) ? tv.y : 0.f;
          tv.z = (m & 4u) ? tv.z : 0.f;
          tv.w = (m & 8u) ? tv.w : 0.f;
        }
        t[0] = tv.x; t[1] = tv.y; t[2] = tv.z; t[3] = tv.w;
      } else {
  #pragma unroll
        for (int j = 0; j < 4; ++j)
          if (s0 + j < a.CP) {
            float x = phave ? load1_any(p_P, poff + (long long)pl * a.CP + s0 + j, kBF) : 0.f;
            if (Pm && !(Pm[(long long)vl * a.CP + s0 + j] > 0.f)) x = 0.f;
            t[j] = x;
          }
      }
      {
        const float live = valid ? 1.f : 0.f;  // (branch-free)
        psum.x = fmaf(live, t[0], psum.x); psum.y = fmaf(live, t[1], psum.y);
        psum.z = fmaf(live, t[2], psum.z); psum.w = fmaf(live, t[3], psum.w);
      }
      slab[v] = make_float4(t[0] * s, t[1] * s, t[2] * s, t[3] * s);  // zero in the slots past N
      R[vi] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  {
    using T = std::true_type;
    using F = std::false_type;
    const int pm = (a.map_side == 1 || !slab_full) ? 3 : (Pbits ? 2 : (Pm ? 1 : 0));
    bool done = false;
    if constexpr (CQ == 16 && TCT == 512) {
      if (a.src3_n > 0) {  // (host: fp32 P = dout, full slabs, no row map, sign bytes or no mask)
        if (pm == 2) load_p(std::integral_constant<int, 2>{}, F{}, T{});
        else load_p(std::integral_constant<int, 0>{}, F{}, T{});
        done = true;
      }
    }
    if (done) {
    } else if (pbf) {  // (bf16 rows never come with an fp32 mask: mode 1 does not exist for them)
      if (pm == 3) load_p(std::integral_constant<int, 3>{}, T{}, F{});
      else if (pm == 2) load_p(std::integral_constant<int, 2>{}, T{}, F{});
      else load_p(std::integral_constant<int, 0>{}, T{}, F{});
    } else {
      if (pm == 3) load_p(std::integral_constant<int, 3>{}, F{}, F{});
      else if (pm == 2) load_p(std::integral_constant<int, 2>{}, F{}, F{});
      else if (pm == 1) load_p(std::integral_constant<int, 1>{}, F{}, F{});
      else load_p(std::integral_constant<int, 0>{}, F{}, F{});
    }
  }
  if (a.db_mode == 2 && q0 == 0) {  // column sums of the P slab (dpre): every lane holds its own vertices' sum
    float* pk = part + (long long)a.K * CQT * 4;
    float v4[4] = {psum.x, psum.y, psum.z, psum.w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float x = v4[c];
      x = xor_add(x, 1);
      x = xor_add(x, 2);
      x = xor_add(x, 4);
      x = xor_add(x, 8);
      x = xor_add(x, 16);
      x = xor_add(x, 32);
      if (lane == 0) pk[c] = x;
    }
  }
  MVH_STAMPX(4);
  __syncthreads();  // slab = t~_0, ELL staged
  MVH_STAMPX(5);

  auto gather = [&](int v, int vi, const float4* slab) {  // (shadows the kernel's `slab`: the slab to gather from)
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int q = 0; q < PW / 4; ++q) {
      uint4 id;
      if constexpr (kDB) id = ids[vi];
      else id = ellv[v * (PW / 4) + q];
      {
        const float4 n0 = slab[id.x & 0xffffu], n1 = slab[id.x >> 16];
        const float4 n2 = slab[id.y & 0xffffu], n3 = slab[id.y >> 16];
        add4f(g, n0); add4f(g, n1); add4f(g, n2); add4f(g, n3);
      }
      asm volatile("" ::: "memory");
      {
        const float4 n0 = slab[id.z & 0xffffu], n1 = slab[id.z >> 16];
        const float4 n2 = slab[id.w & 0xffffu], n3 = slab[id.w >> 16];
        add4f(g, n0); add4f(g, n1); add4f(g, n2); add4f(g, n3);
      }
      asm volatile("" ::: "memory");
    }
    if constexpr (kOvf) {
      if (ovf_any[vi]) {
        const float4 n0 = slab[ovf0[vi] & 0xffffu], n1 = slab[ovf0[vi] >> 16];
        const float4 n2 = slab[ovf1[vi] & 0xffffu], n3 = slab[ovf1[vi] >> 16];
        add4f(g, n0); add4f(g, n1); add4f(g, n2); add4f(g, n3);
      }
    }
    return g;
  };

  // dW tile of order k from the slab: 4 (x QH) MFMAs per 16 vertices, then one cross-block reduce
  auto mfma_pass = [&](int k, const float* slabf) {
    f32x4 acc[QH][4];
#pragma unroll
    for (int h = 0; h < QH; ++h)
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[h][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < STEPS_CT; ++s) {
      const int v = 16 * (s * NW + wave) + (lane >> 2);
      const float tb = slabf[v * 4 + (lane & 3)];
      if constexpr (SPLIT) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(qone[s], tb, acc[0][0], 0, 0, 0);
        if ((s & 7) == 7) asm volatile("" ::: "memory");
        continue;
      }
#pragma unroll
      for (int h = 0; h < QH; ++h) {
        acc[h][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(qreg[s][h].x, tb, acc[h][0], 0, 0, 0);
        acc[h][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(qreg[s][h].y, tb, acc[h][1], 0, 0, 0);
        acc[h][2] = __builtin_amdgcn_mfma_f32_4x4x1f32(qreg[s][h].z, tb, acc[h][2], 0, 0, 0);
        acc[h][3] = __builtin_amdgcn_mfma_f32_4x4x1f32(qreg[s][h].w, tb, acc[h][3], 0, 0, 0);
      }
      if ((s & 3) == 3) asm volatile("" ::: "memory");  // at most 4 slab reads in flight (VGPR budget)
    }
    // lane (b, j), register r of acc[h][m]  =  sum over this block's vertices of
    // Q~[v][16h + 4r + m] * t~_k[v][j]; fold the 16 blocks, lanes 0..3 write the tile
    if constexpr (SPLIT) {  // register r <-> channel q0 + r
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float x = sum_blocks(acc[0][0][r]);
        if (lane < 4) part[((long long)k * CQT + q0 + r) * 4 + lane] = x;
      }
      return;
    }
    // all cross-lane sums first (independent chains), then ONE predicated block of stores
    float red[QH][4][4];
#pragma unroll
    for (int h = 0; h < QH; ++h)
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[h][m][r] = sum_blocks(acc[h][m][r]);
    if (lane < 4) {
#pragma unroll
      for (int h = 0; h < QH; ++h)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int q = 16 * h + 4 * r + m;
            if (q < CQ) part[((long long)k * CQT + q) * 4 + lane] = red[h][m][r];
          }
    }
  };

  mfma_pass(0, reinterpret_cast<const float*>(slab));
  MVH_STAMPX(6);
  if constexpr (kDB) {
    float4* cur = slab;   // t~_{k-1}
    float4* oth = slabB;  // t~_{k-2}, becomes t~_k (own rows only)
    for (int k = 1; k < a.K; ++k) {
      const float sc = (k == 1) ? 0.5f : 1.0f;  // T_1 = L T_0 ; T_k = 2 L T_{k-1} - T_{k-2}
#pragma unroll
      for (int vi = 0; vi < VPT; ++vi) {
        const int v = tid + vi * THREADS;
        const float4 o = oth[v];
        const float4 g = gather(v, vi, cur);
        const float kk = ka2[vi] * sc;
        oth[v] = make_float4(fmaf(kk, g.x, -o.x), fmaf(kk, g.y, -o.y), fmaf(kk, g.z, -o.z), fmaf(kk, g.w, -o.w));
      }
      __syncthreads();  // t~_k complete; every gather of t~_{k-1} and the previous MFMA pass (which read `cur`) done
      mfma_pass(k, reinterpret_cast<const float*>(oth));
      float4* t = cur;
      cur = oth;
      oth = t;
    }
  } else {
    for (int k = 1; k < a.K; ++k) {
      const float sc = (k == 1) ? 0.5f : 1.0f;  // T_1 = L T_0 ; T_k = 2 L T_{k-1} - T_{k-2}
#pragma unroll
      for (int vi = 0; vi < VPT; ++vi) {
        const float4 g = gather(tid + vi * THREADS, vi, slab);
        const float kk = ka2[vi] * sc;
        R[vi] = make_float4(fmaf(kk, g.x, -R[vi].x), fmaf(kk, g.y, -R[vi].y), fmaf(kk, g.z, -R[vi].z),
                            fmaf(kk, g.w, -R[vi].w));
      }
      MVH_STAMPX(3 + 4 * k);
      __syncthreads();  // all reads of t~_{k-1} (gathers and the previous MFMA pass) are done
      MVH_STAMPX(4 + 4 * k);
#pragma unroll
      for (int vi = 0; vi < VPT; ++vi) {
        const int v = tid + vi * THREADS;
        const float4 old = slab[v];
        slab[v] = R[vi];
        R[vi] = old;
      }
      __syncthreads();
      MVH_STAMPX(5 + 4 * k);
      mfma_pass(k, reinterpret_cast<const float*>(slab));
      MVH_STAMPX(6 + 4 * k);
    }
  }
#ifdef MVH_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  MVH_STAMPX(30);
}

#ifdef MVH_STAMP
MVH_STAMP_READER(mvh_debug_read_stamps_dw)
#endif

// Sum the per-(mesh, wave) partial tiles in fixed order and scatter into dW [K][Cin][Cout] / db.
// part layout [slab][mesh*NW + wave][tile]; one block = 64 consecutive tile entries x 16 groups
// of partials, every thread keeps 4 independent running sums (loads pipelined, fixed order).
__device__ __forceinline__ void dw_reduce_body(const DwReduceEntry& t, int block) {
  const float* __restrict__ part = t.part;
  const int n_part = t.n_part, NS = t.NS, K = t.K, CQ = t.CQ, CP = t.CP, p_is_x = t.p_is_x, Cin = t.Cin, Cout = t.Cout,
            db_mode = t.db_mode;
  float* __restrict__ dW = t.dW;
  float* __restrict__ db = t.db;
  const float* __restrict__ S = t.S;
  __shared__ float red[16][64];
  __shared__ float red0[16][64];   // split path: the order-0 sums of the same (q, j)
  const int tile = (K + 1) * CQ * 4;  // floats per (slab, mesh, wave)
  const int n_out = NS * tile;
  const int lo = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int o = block * 64 + lo;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, z0 = 0.f;
  if (o < n_out) {
    const int sl = o / tile, e = o - sl * tile;
    const float* src = part + (long long)sl * n_part * tile + e;
    int p = grp;
    for (; p + 48 < n_part; p += 64) {
      s0 += src[(long long)p * tile];
      s1 += src[(long long)(p + 16) * tile];
      s2 += src[(long long)(p + 32) * tile];
      s3 += src[(long long)(p + 48) * tile];
    }
    for (; p < n_part; p += 16) s0 += src[(long long)p * tile];
    if (S) {  // the order-0 entry of the same (q, j): e0 = e mod (CQ * 4)
      const float* src0 = part + (long long)sl * n_part * tile + (e % (CQ * 4));
      for (int p2 = grp; p2 < n_part; p2 += 16) z0 += src0[(long long)p2 * tile];
    }
  }
  red[grp][lo] = (s0 + s1) + (s2 + s3);
  red0[grp][lo] = z0;
  __syncthreads();
  if (grp != 0 || o >= n_out) return;
  float s = 0.f, sz = 0.f;
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    s += red[g][lo];
    sz += red0[g][lo];
  }
  const int sl = o / tile, e = o - sl * tile;
  const int k = e / (CQ * 4), q = (e / 4) % CQ, j = e & 3;
  const int p = sl * 4 + j;
  if (k < K) {
    if (p >= CP) return;
    const int ci = p_is_x ? p : q, co = p_is_x ? q : p;
    if (S) {
      const float ck = (k & 1) ? 0.f : ((k & 2) ? -1.f : 1.f);   // T_k(0)
      s += ck * (S[ci * Cout + co] - sz);
    }
    dW[((long long)k * Cin + ci) * Cout + co] = s;
  } else if (db) {
    if (db_mode == 1 && sl == 0 && j == 0) db[q] = s;  // Q = dpre: db[co = q]
    if (db_mode == 2 && q == 0 && p < CP) db[p] = s;    // P = dpre: db[co = p]
  }
}

__global__ void __launch_bounds__(1024) k_dw_reduce(DwReduceEntry t) { dw_reduce_body(t, blockIdx.x); }

// Every deferred layer of a step in ONE launch (blockIdx.y = layer): the step engine leaves the
// partial tiles of each conv layer in its own buffer and reduces them all after the join, which
// takes ~10 dependent 5 us launches off the weight-gradient lane.
__global__ void __launch_bounds__(1024) k_dw_reduce_all(DwReduceTable t) {
  const DwReduceEntry& e = t.e[blockIdx.y];
  const int n_out = e.NS * (e.K + 1) * e.CQ * 4;
  if ((int)blockIdx.x * 64 >= n_out) return;  // uniform per block
  dw_reduce_body(e, blockIdx.x);
}

int launch_dw_reduce_all(hipStream_t st, const DwReduceTable& t) {
  if (t.n == 0) return MVH_OK;
  int mx = 0;
  for (int i = 0; i < t.n; ++i) {
    const int n_out = t.e[i].NS * (t.e[i].K + 1) * t.e[i].CQ * 4;
    if (n_out > mx) mx = n_out;
  }
  hipLaunchKernelGGL(k_dw_reduce_all, dim3(cdiv(mx, 64), t.n), dim3(1024), 0, st, t);
  MVH_LAUNCH_CHECK();
  return MVH_OK;
}

template <int CQ, int VPT, int TCT, int PW>
static int launch_dw_one(hipStream_t st, const float* P, const float* Pm, const float* Q, const float* Qm,
                         const mvh_csr_t* lap, float* part, const DwDims& d, int threads, const int32_t* map,
                         const float* g3, const float* w3) {
  auto kern = k_cheb_dw_lds<CQ, VPT, TCT, PW>;
  const size_t lds = (size_t)VPT * threads * (16 + PW * 4);
  static LdsAttr attr;
  if (int rc = attr.ensure(reinterpret_cast<const void*>(kern), lds)) return rc;
  const int NS = (d.CP + 3) / 4, QP = (CQ == 4) ? d.CQtot / 4 : 1;
  // d.mesh0 < 0: the batch in -mesh0 launches of whole 8-mesh groups, one behind the other on this stream (ConvIO::dw_split)
  const int split = d.mesh0 < 0 ? -d.mesh0 : 1;
  const int per = ((((d.B + split - 1) / split) + 7) / 8) * 8;
  for (int m0 = 0; m0 < d.B; m0 += per) {
    DwDims dd = d;
    dd.mesh0 = m0;
    const int nb = d.B - m0 < per ? d.B - m0 : per;
    const int grid = ((nb + 7) / 8) * 8 * NS * QP;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, st, P, Pm, Q, Qm, lap->rowinfo, lap->ell, part, map, lap->col, g3, w3, dd);
    MVH_LAUNCH_CHECK();
  }
  return MVH_OK;
}

template <int CQ>
static int launch_dw_cq(hipStream_t st, const float* P, const float* Pm, const float* Q, const float* Qm,
                        const mvh_csr_t* lap, float* part, const DwDims& d, int vpt, int threads, const int32_t* map,
                        const float* g3 = nullptr, const float* w3 = nullptr) {
  const bool pw8 = d.pairs > 4;
#define MVH_DW(V, T)                                                                                         \
  return pw8 ? launch_dw_one<CQ, V, T, 8>(st, P, Pm, Q, Qm, lap, part, d, threads, map, g3, w3)              \
             : launch_dw_one<CQ, V, T, 4>(st, P, Pm, Q, Qm, lap, part, d, threads, map, g3, w3)
  if (vpt == 1) { MVH_DW(1, 0); }
  if (vpt == 2) { MVH_DW(2, 0); }
  if constexpr (CQ <= 16) {
    if (vpt == 10 && threads == 512) { MVH_DW(10, 512); }
  }
#undef MVH_DW
  return -1;
}

size_t cheb_dw_lds_ws_bytes(int B, int N, int Cin, int Cout, int K) {
  (void)N;
  const int cq = Cin > Cout ? Cin : Cout, cp = Cin > Cout ? Cout : Cin;
  // worst case 16 waves per block
  return (size_t)B * ((cp + 3) / 4) * 16 * (size_t)(K + 1) * cq * 4 * sizeof(float) + 256;
}

// dW (+ db) through the LDS-resident kernels; *handled == false -> use the general pipeline.
int try_cheb_dw_lds(hipStream_t st, const mvh_csr_t* lap, const float* x, const float* dout, const float* out_mask,
                    float* dW, float* db, int B, int N, int Cin, int Cout, int K, float* part, size_t part_bytes,
                    bool* handled, int bstride, const int32_t* dout_map, int dout_rows, bool dry_run,
                    const uint8_t* out_bits, DwReduceEntry* defer, bool x_bf16, bool dout_bf16, const ConvIO* src3) {
  *handled = false;
  const int dw_split = src3 ? src3->dw_split : 1;
  if (src3 && !src3->src3_g && !src3->x_map) src3 = nullptr;
  if (dbg().force_generic) return MVH_OK;
  if (x_bf16 && dout_bf16 && !dout_map && bstride == 0 && (out_bits || !out_mask)) {
    // the 5k level's 16 -> 16 layer on bf16 rows: packed registers, contraction on the bf16 matrix pipe
    if (int rc = try_cheb_dw_l0h(st, lap, x, dout, Cout % 4 == 0 ? out_bits : nullptr, dW, db, B, N, Cin, Cout, K, part,
                                 part_bytes, handled, dry_run, defer, dw_split)) return rc;
    if (*handled) return MVH_OK;
  }
  const int need = MVH_CSR_NORMALIZED_LAPLACIAN | MVH_CSR_SYMMETRIC;
  if (!lap->rowinfo || !lap->ell || lap->ell_pairs <= 0 || lap->ell_pairs > 8 || (lap->flags & need) != need)
    return MVH_OK;
  if (N < 1 || N + 1 >= 65535 || B < 1) return MVH_OK;
  // P = the side with fewer channels runs the recurrence; Q stays in registers.  Ties: dout, whose ReLU mask then costs
  // one byte per vertex on the slab side instead of one per vertex and lane group on the register side (MEASURED: 549.4 vs
  // 554.3 us per step in three alternating runs; debug switch dw_tie_x = round 2's choice, x)
  const bool p_is_x = Cin < Cout || (Cin == Cout && dbg().dw_tie_x);
  const int CP = p_is_x ? Cin : Cout, CQ = p_is_x ? Cout : Cin;
  if (CQ != 8 && CQ != 16 && CQ != 32) return MVH_OK;
  if (((uintptr_t)x | (uintptr_t)dout | (uintptr_t)out_mask) % 16 != 0) return MVH_OK;
  int vpt, threads;
  if (N + 1 <= 1024) { vpt = 1; threads = ((N + 1 + 63) / 64) * 64; }
  else if (N + 1 <= 2048) { vpt = 2; threads = (((N + 2) / 2 + 63) / 64) * 64; }
  else if (N + 1 <= 5120 && CQ <= 16 && !(lap->flags & MVH_CSR_ELL_OVERFLOW)) { vpt = 10; threads = 512; }
  else return MVH_OK;
  const int pw = lap->ell_pairs > 4 ? 8 : 4;
  if ((size_t)vpt * threads * (16 + pw * 4) > 160 * 1024) return MVH_OK;
  const int NS = (CP + 3) / 4, NW = threads / 64;
  const size_t need_bytes = (size_t)B * NS * NW * (K + 1) * CQ * 4 * sizeof(float);
  if (!part || part_bytes < need_bytes) return MVH_OK;

  if (out_bits && Cout % 4 != 0) out_bits = nullptr;  // sign bytes cover whole 4-channel groups only
  // bf16 tensors are read in 4-channel words; a ReLU mask then comes as sign bytes (never the fp32 output)
  if ((x_bf16 && Cin % 4 != 0) || (dout_bf16 && Cout % 4 != 0)) return MVH_OK;
  if ((x_bf16 || dout_bf16) && out_mask && !out_bits) return MVH_OK;
  if (dry_run) {
    *handled = true;
    return MVH_OK;
  }
  DwDims d;
  d.mask_bits = out_bits ? 1 : 0;
  d.ovf = (lap->flags & MVH_CSR_ELL_OVERFLOW) ? 1 : 0;
  if (out_bits) out_mask = reinterpret_cast<const float*>(out_bits);
  d.B = B; d.N = N; d.K = K; d.CP = CP; d.CQtot = CQ; d.pairs = lap->ell_pairs;
  d.db_mode = db ? (p_is_x ? 1 : 2) : 0;
  d.bs = bstride > 0 ? bstride : N;
  d.map_side = dout_map ? (p_is_x ? 2 : 1) : 0;  // dout is Q when x runs the recurrence, else P
  d.map_bs = dout_rows;
  if (src3 && src3->x_map) {  // strided x (ConvIO::x_map): the ONE row map of the kernel goes to x's side
    if (dout_map || bstride != 0 || x_bf16 || src3->src3_g)
      return fail(MVH_ERR_UNSUPPORTED, "cheb_dw_lds: a strided x beside another row map / sub-problem / bf16 rows");
    d.map_side = p_is_x ? 1 : 2;
    d.map_bs = src3->x_bs;
    dout_map = src3->x_map;   // (p_map of the launch below)
  }
  d.p_bf16 = (p_is_x ? x_bf16 : dout_bf16) ? 1 : 0;
  d.q_bf16 = (p_is_x ? dout_bf16 : x_bf16) ? 1 : 0;
  d.src3_n = 0; d.src3_c = 0;
  d.mesh0 = (dw_split > 1 && vpt == 10) ? -dw_split : 0;
  if (src3 && src3->src3_g) {  // lazy dout rows: the 5k level's fp32 16 -> 16 kernel with the recurrence on dout only
    const bool ok = !p_is_x && CQ == 16 && CP == 16 && vpt == 10 && threads == 512 && !dout_bf16 && !dout_map && bstride == 0 &&
                    (!out_mask || out_bits) && src3->src3_w && src3->src3_c == 3 && src3->src3_n >= 1 &&
                    src3->src3_n <= 512;
    if (!ok) return fail(MVH_ERR_UNSUPPORTED, "cheb_dw_lds: lazy output-gradient rows (src3) on a layer without that kernel");
    d.src3_n = src3->src3_n; d.src3_c = src3->src3_c;
  }
  const float* P = p_is_x ? x : dout;
  const float* Pm = p_is_x ? nullptr : out_mask;
  const float* Q = p_is_x ? dout : x;
  const float* Qm = p_is_x ? out_mask : nullptr;
  int rc;
  if (NS == 1 && CQ % 4 == 0 && CQ >= 8) rc = launch_dw_cq<4>(st, P, Pm, Q, Qm, lap, part, d, vpt, threads, dout_map);  // Q-split
  else if (CQ == 8) rc = launch_dw_cq<8>(st, P, Pm, Q, Qm, lap, part, d, vpt, threads, dout_map);
  else if (CQ == 16) rc = launch_dw_cq<16>(st, P, Pm, Q, Qm, lap, part, d, vpt, threads, dout_map, src3 ? src3->src3_g : nullptr,
                                           (src3 && src3->src3_g) ? src3->src3_w : nullptr);
  else rc = launch_dw_cq<32>(st, P, Pm, Q, Qm, lap, part, d, vpt, threads, dout_map);
  if (rc < 0) return fail(MVH_ERR_UNSUPPORTED, "cheb_dw_lds: no kernel for vpt=%d threads=%d", vpt, threads);
  if (rc) return rc;
  const int n_out = NS * (K + 1) * CQ * 4;
  DwReduceEntry ent{part, B * NW, NS, K, CQ, CP, p_is_x ? 1 : 0, Cin, Cout, d.db_mode, dW, db};
  if (defer) {
    *defer = ent;  // the caller reduces later (launch_dw_reduce_all); `part` must stay untouched until then
  } else {
    hipLaunchKernelGGL(k_dw_reduce, dim3(cdiv(n_out, 64)), dim3(1024), 0, st, ent);
    MVH_LAUNCH_CHECK();
  }
  *handled = true;
  return MVH_OK;
}

}  // namespace mvh
