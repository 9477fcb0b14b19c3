// Row P (nn/conv.py:242-331): the sparse propagate  y = add + alpha * (P x) + beta * z
// over [B, N, C] activations, without materialising the reference's [E, B, C] message
// tensor.  One lane owns one (mesh, output row, VEC-channel group): the row's CSR entries
// are walked in the reference's edge order and the neighbour rows are gathered with
// 16-byte loads; the 4..8 lanes of a row read the same col/val words (wave broadcast).
// HBM-bound; the gathers are served by L2 (one mesh level is <= 320 KB).
#include "common.hpp"
#include "bf16.hpp"

namespace mvh {

template <int VEC>
struct VecT;
template <>
struct VecT<1> {
  using type = float;
};
template <>
struct VecT<4> {
  using type = float4;
};

// C == 3 (the network input, x y z): one lane owns a whole (mesh, row): the three channels are 12 contiguous
// bytes, a wave writes 768 contiguous bytes, and the row's col/val words are read once instead of three times
// (the one-lane-per-channel form took 60 us per propagate at the 20k level for 15 MB of data).
__global__ void __launch_bounds__(256)
k_spmm3(const int* __restrict__ rowptr, const int* __restrict__ col, const float* __restrict__ val, int n_rows,
        int n_cols, const float* __restrict__ x, float* y, const float* add, const float* z, float alpha, float beta,
        int wg_per_mesh, int xcd_remap) {
  long long bid = blockIdx.x, b;
  int tile;
  if (xcd_remap) {
    const int xcd = (int)(bid & 7);
    const long long slot = bid >> 3;
    b = xcd + 8 * (slot / wg_per_mesh);
    tile = (int)(slot % wg_per_mesh);
  } else {
    b = bid / wg_per_mesh;
    tile = (int)(bid % wg_per_mesh);
  }
  const int r = tile * (int)blockDim.x + (int)threadIdx.x;
  if (r >= n_rows) return;
  const float* xb = x + b * (long long)n_cols * 3;
  const long long o = (b * n_rows + r) * 3;
  float zp[3] = {0.f, 0.f, 0.f}, ap[3] = {0.f, 0.f, 0.f};  // the HBM-resident terms first (see k_spmm)
  if (z) { zp[0] = z[o]; zp[1] = z[o + 1]; zp[2] = z[o + 2]; }
  if (add) { ap[0] = add[o]; ap[1] = add[o + 1]; ap[2] = add[o + 2]; }
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  const int e1 = rowptr[r + 1];
  for (int e = rowptr[r]; e < e1; ++e) {
    const float* p = xb + (long long)col[e] * 3;
    const float v = val[e];
    a0 = fmaf(v, p[0], a0);
    a1 = fmaf(v, p[1], a1);
    a2 = fmaf(v, p[2], a2);
  }
  const bool plain = (add == nullptr) && (z == nullptr) && alpha == 1.f;
  float res[3] = {a0, a1, a2};
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    if (!plain) {
      res[i] = alpha * res[i];
      if (add) res[i] += ap[i];
      if (z) res[i] = fmaf(beta, zp[i], res[i]);
    }
    y[o + i] = res[i];
  }
}

template <bool EXACT>
__device__ __forceinline__ void mac(float& acc, float v, float x) {
  if constexpr (EXACT) {
    acc = __fadd_rn(acc, __fmul_rn(v, x));  // two roundings, as index_select*norm then scatter_add_
  } else {
    acc = fmaf(v, x, acc);
  }
}

template <int VEC, bool EXACT>
__global__ void __launch_bounds__(256)
k_spmm(const int* __restrict__ rowptr, const int* __restrict__ col, const float* __restrict__ val,
       int n_rows, int n_cols, const float* __restrict__ x, float* y,
       const float* add, const float* z,  // y may alias add (in-place Clenshaw update)
       float alpha, float beta, int C,
       int wg_per_mesh, int xcd_remap, int y_bf16) {
  // One mesh = wg_per_mesh consecutive tiles.  Workgroup ids are dealt round-robin to the 8 XCDs, each with
  // its own 4 MB L2; with the linear order every XCD touches every mesh and the ~7 gathers per row (a 20k-vertex
  // level is 1.3 MB per mesh at 16 channels) miss L2 and go to MALL/HBM.  Remapped, XCD x walks the meshes
  // b = x, x+8, ... one after another, so a mesh's rows stay in that XCD's L2 while its tiles run.
  long long bid = blockIdx.x;
  long long b;
  int tile;
  if (xcd_remap) {
    const int xcd = (int)(bid & 7);
    const long long slot = bid >> 3;
    b = xcd + 8 * (slot / wg_per_mesh);
    tile = (int)(slot % wg_per_mesh);
  } else {
    b = bid / wg_per_mesh;
    tile = (int)(bid % wg_per_mesh);
  }
  const int CV = C / VEC;
  const int in_mesh = tile * (int)blockDim.x + (int)threadIdx.x;
  if (in_mesh >= n_rows * CV) return;
  const int cv = in_mesh % CV;
  const int r = in_mesh / CV;
  const float* xb = x + b * (long long)n_cols * C + (long long)cv * VEC;
  const int e0 = rowptr[r], e1 = rowptr[r + 1];
  // the recurrence's T_{k-2} term is the one stream of this kernel that comes from HBM rather than L2: issue its
  // load first so that it is in flight underneath the gathers
  const long long o_pre = (b * n_rows + r) * (long long)C + (long long)cv * VEC;
  float zpre[VEC];
  if constexpr (VEC == 4) {
    const float4 t = z ? *reinterpret_cast<const float4*>(z + o_pre) : make_float4(0.f, 0.f, 0.f, 0.f);
    zpre[0] = t.x; zpre[1] = t.y; zpre[2] = t.z; zpre[3] = t.w;
  } else {
    zpre[0] = z ? z[o_pre] : 0.f;
  }
  float apre[VEC];  // (same for the additive term: G_k of the Clenshaw update; y may alias it, it is read before the store)
  if constexpr (VEC == 4) {
    const float4 t = add ? *reinterpret_cast<const float4*>(add + o_pre) : make_float4(0.f, 0.f, 0.f, 0.f);
    apre[0] = t.x; apre[1] = t.y; apre[2] = t.z; apre[3] = t.w;
  } else {
    apre[0] = add ? add[o_pre] : 0.f;
  }
  float acc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
  int e = e0;
  if constexpr (VEC == 4) {
    // four independent gathers in flight per lane (long rows, e.g. the transposed upsampling
    // operator with ~12 taps per row, are otherwise one dependent HBM/L2 round trip per tap);
    // the adds stay in edge order, so EXACT results are unchanged
    for (; e + 3 < e1; e += 4) {
      const int c0 = col[e], c1 = col[e + 1], c2 = col[e + 2], c3 = col[e + 3];
      const float v0 = val[e], v1 = val[e + 1], v2 = val[e + 2], v3 = val[e + 3];
      const float4 x0 = *reinterpret_cast<const float4*>(xb + (long long)c0 * C);
      const float4 x1 = *reinterpret_cast<const float4*>(xb + (long long)c1 * C);
      const float4 x2 = *reinterpret_cast<const float4*>(xb + (long long)c2 * C);
      const float4 x3 = *reinterpret_cast<const float4*>(xb + (long long)c3 * C);
      mac<EXACT>(acc[0], v0, x0.x); mac<EXACT>(acc[1], v0, x0.y); mac<EXACT>(acc[2], v0, x0.z); mac<EXACT>(acc[3], v0, x0.w);
      mac<EXACT>(acc[0], v1, x1.x); mac<EXACT>(acc[1], v1, x1.y); mac<EXACT>(acc[2], v1, x1.z); mac<EXACT>(acc[3], v1, x1.w);
      mac<EXACT>(acc[0], v2, x2.x); mac<EXACT>(acc[1], v2, x2.y); mac<EXACT>(acc[2], v2, x2.z); mac<EXACT>(acc[3], v2, x2.w);
      mac<EXACT>(acc[0], v3, x3.x); mac<EXACT>(acc[1], v3, x3.y); mac<EXACT>(acc[2], v3, x3.z); mac<EXACT>(acc[3], v3, x3.w);
    }
    if (e + 2 < e1) {  // exactly three left: the barycentric upsampling row (nn/pool.py U: 3 taps)
      const int c0 = col[e], c1 = col[e + 1], c2 = col[e + 2];
      const float v0 = val[e], v1 = val[e + 1], v2 = val[e + 2];
      const float4 x0 = *reinterpret_cast<const float4*>(xb + (long long)c0 * C);
      const float4 x1 = *reinterpret_cast<const float4*>(xb + (long long)c1 * C);
      const float4 x2 = *reinterpret_cast<const float4*>(xb + (long long)c2 * C);
      mac<EXACT>(acc[0], v0, x0.x); mac<EXACT>(acc[1], v0, x0.y); mac<EXACT>(acc[2], v0, x0.z); mac<EXACT>(acc[3], v0, x0.w);
      mac<EXACT>(acc[0], v1, x1.x); mac<EXACT>(acc[1], v1, x1.y); mac<EXACT>(acc[2], v1, x1.z); mac<EXACT>(acc[3], v1, x1.w);
      mac<EXACT>(acc[0], v2, x2.x); mac<EXACT>(acc[1], v2, x2.y); mac<EXACT>(acc[2], v2, x2.z); mac<EXACT>(acc[3], v2, x2.w);
      e += 3;
    }
  }
  for (; e < e1; ++e) {
    const int c = col[e];
    const float v = val[e];
    if constexpr (VEC == 4) {
      const float4 xv = *reinterpret_cast<const float4*>(xb + (long long)c * C);
      mac<EXACT>(acc[0], v, xv.x);
      mac<EXACT>(acc[1], v, xv.y);
      mac<EXACT>(acc[2], v, xv.z);
      mac<EXACT>(acc[3], v, xv.w);
    } else {
      mac<EXACT>(acc[0], v, xb[(long long)c * C]);
    }
  }
  const long long o = (b * n_rows + r) * (long long)C + (long long)cv * VEC;
  const bool plain = (add == nullptr) && (z == nullptr) && alpha == 1.f;
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    float res = acc[i];
    if (!plain) {
      res = alpha * res;
      if (add) res += apre[i];
      if (z) res = fmaf(beta, zpre[i], res);
    }
    acc[i] = res;
  }
  if constexpr (VEC == 4) {
    store4_any(y, o, y_bf16 != 0, acc[0], acc[1], acc[2], acc[3]);
  } else {
    y[o] = acc[0];
  }
}

int launch_spmm(hipStream_t st, const mvh_csr_t* op, const float* x, float* y, const float* add,
                const float* z, float alpha, float beta, int B, int C, bool exact, bool y_bf16) {
  if (B == 0 || op->n_rows == 0 || C == 0) return MVH_OK;
  const bool v4 = (C % 4 == 0) && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)add | (uintptr_t)z) % 16 == 0);
  if (y_bf16 && (!v4 || add || z)) return fail(MVH_ERR_UNSUPPORTED, "spmm: bf16 output needs 4-channel groups and no add/z terms");
  const int yb = y_bf16 ? 1 : 0;
  if (C == 3 && !exact && op->n_rows >= 4096) {  // big level, three channels: one lane per row
    const int wpm = cdiv(op->n_rows, 256);
    MVH_REQUIRE((long long)B * wpm < (1ll << 31), "spmm: grid too large");
    const int remap = (B % 8 == 0 && wpm >= 16 && !dbg().no_xcd_remap) ? 1 : 0;
    hipLaunchKernelGGL(k_spmm3, dim3(B * wpm), dim3(256), 0, st, op->rowptr, op->col, op->val, op->n_rows, op->n_cols,
                       x, y, add, z, alpha, beta, wpm, remap);
    MVH_LAUNCH_CHECK();
    return MVH_OK;
  }
  const long long per_mesh = (long long)op->n_rows * (v4 ? C / 4 : C);
  MVH_REQUIRE(per_mesh < (1ll << 31) - 256, "spmm: level too large");
  const int wg_per_mesh = (int)cdiv(per_mesh, 256);
  const long long grid_ll = (long long)B * wg_per_mesh;
  MVH_REQUIRE(grid_ll < (1ll << 31), "spmm: grid too large");
  const int grid = (int)grid_ll;
  // (only worth it when one mesh is big enough to thrash: small levels keep the plain order)
  const int xcd_remap = (B % 8 == 0 && wg_per_mesh >= 16 && !dbg().no_xcd_remap) ? 1 : 0;
#define MVH_SPMM(V, E)                                                                             \
  hipLaunchKernelGGL((k_spmm<V, E>), dim3(grid), dim3(256), 0, st, op->rowptr, op->col, op->val,   \
                     op->n_rows, op->n_cols, x, y, add, z, alpha, beta, C, wg_per_mesh, xcd_remap, yb)
  if (v4) {
    if (exact) MVH_SPMM(4, true); else MVH_SPMM(4, false);
  } else {
    if (exact) MVH_SPMM(1, true); else MVH_SPMM(1, false);
  }
#undef MVH_SPMM
  MVH_LAUNCH_CHECK();
  return MVH_OK;
}

}  // namespace mvh
