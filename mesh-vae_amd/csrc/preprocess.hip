// Batch pre-processing around the step (input side): the Procrustes alignment the
// reference runs per mesh when a dataset is built (data.py:144 -> utils.py:58-157) and the per-item
// normalisation of data.py:103-111, as device passes over meshes that stay resident in HBM.
// All arithmetic is fp64, as in the reference (numpy double).  HBM-bound: each mesh is N*3 doubles
// (120 KB at 5k vertices, L2-resident between the passes of one workgroup).
#include "common.hpp"

namespace mvh {

constexpr int kPreThreads = 1024;

// wave-level then block-level sum of NV doubles per thread; result valid in every thread
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double* sh /* [NV][16] */) {
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v[i] += __shfl_xor(v[i], off, 64);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();  // (sh may still be read from a previous call)
  if (lane == 0)
#pragma unroll
    for (int i = 0; i < NV; ++i) sh[i * 16 + w] = v[i];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    double s = 0.0;
    for (int k = 0; k < nw; ++k) s += sh[i * 16 + k];  // fixed order: bitwise reproducible
    v[i] = s;
  }
}

// Pass 1 (utils.py:135-148): per mesh, the centroid, the Frobenius norm of the centred points and the
// 3x3 cross-covariance M = mtx1^T (mtx2 - mean) / norm2 against the standardised template mtx1.
// stats[b] = {mean[3], norm2, M[9] row-major (M[i][j] = sum_v mtx1[v][i] * mtx2n[v][j])}
__global__ void __launch_bounds__(kPreThreads)
k_procrustes_stats(const double* __restrict__ tmpl, const double* __restrict__ pts, double* __restrict__ stats, int N) {
  __shared__ double sh[10 * 16];
  const double* p = pts + (long long)blockIdx.x * N * 3;
  double m[3] = {0.0, 0.0, 0.0};
  for (int v = threadIdx.x; v < N; v += blockDim.x) {
    m[0] += p[v * 3 + 0];
    m[1] += p[v * 3 + 1];
    m[2] += p[v * 3 + 2];
  }
  block_sum<3>(m, sh);
  const double mean[3] = {m[0] / N, m[1] / N, m[2] / N};
  double acc[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) acc[i] = 0.0;
  for (int v = threadIdx.x; v < N; v += blockDim.x) {
    double c[3], t[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      c[j] = p[v * 3 + j] - mean[j];
      t[j] = tmpl[v * 3 + j];
      acc[9] += c[j] * c[j];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[i * 3 + j] += t[i] * c[j];
  }
  block_sum<10>(acc, sh);
  if (threadIdx.x == 0) {
    double* o = stats + (long long)blockIdx.x * 13;
    const double norm2 = sqrt(acc[9]);
    o[0] = mean[0]; o[1] = mean[1]; o[2] = mean[2];
    o[3] = norm2;
    for (int i = 0; i < 9; ++i) o[4 + i] = acc[i] / norm2;
  }
}

// Pass 2 (utils.py:147,152,155): aligned = ((pts - mean) / norm2) @ R^T * s and the disparity
// sum((mtx1 - aligned)^2).  R [B][9] row-major and s [B] come from the 3x3 SVD of M (host, LAPACK).
__global__ void __launch_bounds__(kPreThreads)
k_procrustes_apply(const double* __restrict__ tmpl, const double* __restrict__ pts, const double* __restrict__ stats,
                   const double* __restrict__ R, const double* __restrict__ s, double* __restrict__ aligned,
                   double* __restrict__ disparity, int N) {
  __shared__ double sh[16];
  const long long b = blockIdx.x;
  const double* p = pts + b * N * 3;
  double* o = aligned + b * N * 3;
  const double* st = stats + b * 13;
  const double mean[3] = {st[0], st[1], st[2]};
  const double norm2 = st[3], sc = s[b];
  double r[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) r[i] = R[b * 9 + i];
  double d[1] = {0.0};
  for (int v = threadIdx.x; v < N; v += blockDim.x) {
    double c[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) c[j] = (p[v * 3 + j] - mean[j]) / norm2;
#pragma unroll
    for (int i = 0; i < 3; ++i) {  // (c @ R^T)[i] = sum_j c[j] R[i][j]
      const double a = (c[0] * r[i * 3 + 0] + c[1] * r[i * 3 + 1] + c[2] * r[i * 3 + 2]) * sc;
      o[v * 3 + i] = a;
      const double e = tmpl[v * 3 + i] - a;
      d[0] += e * e;
    }
  }
  block_sum<1>(d, sh);
  if (threadIdx.x == 0 && disparity) disparity[b] = d[0];
}

// data.py:103-111: ori = (mesh[idx] - mean) / std in fp64 (x_gt), and its fp32 cast (the network input).
// Separate IEEE sub and div per element, round-to-nearest cast: bit-identical to the torch CPU ops.
__global__ void __launch_bounds__(256)
k_gather_normalize(const double* __restrict__ data, const long long* __restrict__ idx, const double* __restrict__ mean,
                   const double* __restrict__ stdv, float* __restrict__ x32, double* __restrict__ x64, long long n3,
                   long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const long long b = i / n3, e = i - b * n3;
  const double v = __ddiv_rn(__dsub_rn(data[idx[b] * n3 + e], mean[e]), stdv[e]);
  if (x64) x64[i] = v;
  if (x32) x32[i] = (float)v;
}

}  // namespace mvh

using namespace mvh;

extern "C" int mvh_procrustes_stats(mvh_stream_t stream, const double* tmpl, const double* pts, double* stats,
                                    int32_t B, int32_t N) {
  MVH_REQUIRE(B >= 0 && N > 0, "procrustes_stats: bad sizes B=%d N=%d", B, N);
  if (B == 0) return MVH_OK;  // (empty tensors have null storage)
  MVH_REQUIRE(tmpl && pts && stats, "procrustes_stats: null tensor");
  hipLaunchKernelGGL(k_procrustes_stats, dim3(B), dim3(kPreThreads), 0, (hipStream_t)stream, tmpl, pts, stats, N);
  MVH_LAUNCH_CHECK();
  return MVH_OK;
}

extern "C" int mvh_procrustes_apply(mvh_stream_t stream, const double* tmpl, const double* pts, const double* stats,
                                    const double* R, const double* s, double* aligned, double* disparity,
                                    int32_t B, int32_t N) {
  MVH_REQUIRE(B >= 0 && N > 0, "procrustes_apply: bad sizes B=%d N=%d", B, N);
  if (B == 0) return MVH_OK;
  MVH_REQUIRE(tmpl && pts && stats && R && s && aligned, "procrustes_apply: null tensor");
  hipLaunchKernelGGL(k_procrustes_apply, dim3(B), dim3(kPreThreads), 0, (hipStream_t)stream, tmpl, pts, stats, R, s,
                     aligned, disparity, N);
  MVH_LAUNCH_CHECK();
  return MVH_OK;
}

extern "C" int mvh_gather_normalize(mvh_stream_t stream, const double* data, int64_t n_meshes, const int64_t* idx,
                                    const double* mean, const double* stdv, float* x32, double* x64, int32_t B,
                                    int64_t n3) {
  MVH_REQUIRE(B >= 0 && n3 > 0 && n_meshes >= 0, "gather_normalize: bad sizes B=%d n3=%lld", B, (long long)n3);
  const long long total = (long long)B * n3;
  if (total == 0) return MVH_OK;
  MVH_REQUIRE(data && idx && mean && stdv, "gather_normalize: null tensor");
  MVH_REQUIRE(x32 || x64, "gather_normalize: nothing to compute");
  hipLaunchKernelGGL(k_gather_normalize, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, data,
                     (const long long*)idx, mean, stdv, x32, x64, (long long)n3, total);
  MVH_LAUNCH_CHECK();
  return MVH_OK;
}
