// Row P (nn/conv.py:242-331): the sparse propagate  y = add + alpha * (P x) + beta * z
// over [B, N, C] activations, without materialising the reference's [E, B, C] message
// tensor.  One lane owns one (mesh, output row, VEC-channel group): the row's CSR entries
// are walked in the reference's edge order and the neighbour rows are gathered with
// 16-byte loads; the 4..8 lanes of a row read the same col/val words (wave broadcast).
// HBM-bound; the gathers are served by L2 (one mesh level is <= 320 KB).
#include "common.hpp"

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
       long long total) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int CV = C / VEC;
  const int cv = (int)(idx % CV);
  const long long br = idx / CV;
  const int r = (int)(br % n_rows);
  const long long b = br / n_rows;
  const float* xb = x + b * (long long)n_cols * C + (long long)cv * VEC;
  const int e0 = rowptr[r], e1 = rowptr[r + 1];
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
      if (add) res += add[o + i];
      if (z) res = fmaf(beta, z[o + i], res);
    }
    acc[i] = res;
  }
  if constexpr (VEC == 4) {
    *reinterpret_cast<float4*>(y + o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  } else {
    y[o] = acc[0];
  }
}

int launch_spmm(hipStream_t st, const mvh_csr_t* op, const float* x, float* y, const float* add,
                const float* z, float alpha, float beta, int B, int C, bool exact) {
  if (B == 0 || op->n_rows == 0 || C == 0) return MVH_OK;
  const bool v4 = (C % 4 == 0) && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)add | (uintptr_t)z) % 16 == 0);
  const long long total = (long long)B * op->n_rows * (v4 ? C / 4 : C);
  const int grid = cdiv(total, 256);
#define MVH_SPMM(V, E)                                                                             \
  hipLaunchKernelGGL((k_spmm<V, E>), dim3(grid), dim3(256), 0, st, op->rowptr, op->col, op->val,   \
                     op->n_rows, op->n_cols, x, y, add, z, alpha, beta, C, total)
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
