// Row L: cheb_VAE.loss_function (cheb_VAE.py:321-346) = KLD (logpdf.py:7-8) + Gaussian NLL with
// the soft-clipped constant log-sigma (logpdf.py:22-28) - 2 log q(y), fused into two launches
// forward (per-mesh partial sums in fp64, then a one-block finish) and one launch backward.
// x_gt arrives as fp64 from main.py (data.py:107) and fp32 from inference.py:87; the reference's
// type promotion makes rec/loss fp64 in the first case, which is reproduced here.
// (MEASURED, not kept: the finish pass inside k_loss_partial behind a last-block ticket -- __threadfence + atomicAdd
//  per block, the last one to arrive reduces -- to save a launch on the critical path: 31 us against 7.3 + 5.6 us.
//  A device-scope release writes the XCD's L2 back; 1024 blocks doing it is far dearer than a kernel boundary.)
#include "common.hpp"

namespace mvh {

constexpr int kLossSplit = 16;  // blocks per mesh for the reconstruction sum

// FUSE: the reconstruction itself is produced here (rows >= n_act of the final layer, cheb_VAE.py:288, are the per-vertex
// map x16[v] W_eff: element (v, o) = sum_c x16[v][c] W_eff[c][o], the fma chain of k_cheb_contract in the same order, so
// the values are bitwise the separate launch's; rows < n_act were written by the connected block's kernel before) --
// the element -> thread map and the order of the fp64 sums are those of the plain kernel, so loss and rec are too.
struct LossFuse {
  const float* x16;    // [B][N][cin] input of the final layer
  const float* weff;   // [cin][c3]
  float* recon_w;      // the reconstruction, written for rows >= n_act
  int cin, c3, n_act;
};

template <typename GT, bool FUSE>
__global__ void __launch_bounds__(256)
k_loss_partial(const float* __restrict__ recon, const GT* __restrict__ gt, double inv_sigma,
               double* __restrict__ partial, int NV, float* __restrict__ d_recon, double inv_var_over_B, LossFuse f) {
  const int b = blockIdx.y, s = blockIdx.x;
  const long long base = (long long)b * NV;
  double acc = 0.0;
  for (int i = s * blockDim.x + threadIdx.x; i < NV; i += gridDim.x * blockDim.x) {
    float rv;
    if constexpr (FUSE) {
      const int v = i / f.c3, o = i - v * f.c3;
      if (v >= f.n_act) {
        const float* xr = f.x16 + ((long long)b * (NV / f.c3) + v) * f.cin;
        float a = 0.f;
        for (int c4 = 0; c4 < f.cin; c4 += 4) {
          const float4 t = *reinterpret_cast<const float4*>(xr + c4);
          a = fmaf(t.x, f.weff[(c4 + 0) * f.c3 + o], a);
          a = fmaf(t.y, f.weff[(c4 + 1) * f.c3 + o], a);
          a = fmaf(t.z, f.weff[(c4 + 2) * f.c3 + o], a);
          a = fmaf(t.w, f.weff[(c4 + 3) * f.c3 + o], a);
        }
        rv = a;
        f.recon_w[base + i] = a;
      } else {
        rv = recon[base + i];
      }
    } else {
      rv = recon[base + i];
    }
    double d;
    if constexpr (sizeof(GT) == 8) {
      d = ((double)gt[base + i] - (double)rv) * inv_sigma;
    } else {  // fp32 path: the reference subtracts and divides in fp32
      const float df = (gt[base + i] - rv) / (float)(1.0 / inv_sigma);
      d = (double)df;
    }
    acc += 0.5 * d * d;
    // gradient seed for d_loss = 1 (what k_loss_bwd_recon would write): the step engine skips the
    // separate backward pass over recon / x_gt
    if (d_recon) d_recon[base + i] = (float)(inv_var_over_B * ((double)rv - (double)gt[base + i]));
  }
  __shared__ double red[256];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[(long long)b * gridDim.x + s] = red[0];
}

template <typename GT>
__global__ void __launch_bounds__(256)
k_loss_finish(const double* __restrict__ partial, const float* __restrict__ mu,
              const float* __restrict__ logvar, const float* __restrict__ y,
              const float* __restrict__ y_hat, double elem_const, GT* __restrict__ loss,
              GT* __restrict__ rec, float* __restrict__ kld, long long* __restrict__ correct, int B,
              int NV, int C, int Z, int S, float* __restrict__ d_mu, float* __restrict__ d_logvar,
              float* __restrict__ d_yhat) {
  __shared__ double red[256];
  __shared__ int redc[256];
  double tot = 0.0;
  int corr = 0;
  // four lanes per mesh: the S partial sums and the Z latent terms are split four ways and
  // combined with two xor-shuffles (a one-thread-per-mesh loop left 3/4 of the block idle and
  // made this single-block kernel a 10 us stop on the critical path)
  const int sub = threadIdx.x & 3;
  for (int b0 = 0; b0 < B; b0 += blockDim.x >> 2) {
    const int b = b0 + (threadIdx.x >> 2);
    const bool live = b < B;
    const int bb = live ? b : 0;
    double r = 0.0;
    for (int s = sub; s < S; s += 4) r += partial[(long long)bb * S + s];
    float k = 0.f;
    for (int t = sub; t < Z; t += 4) {
      const float m = mu[(long long)bb * Z + t], lv = logvar[(long long)bb * Z + t];
      k += 1.f + lv - m * m - expf(lv);
      if (d_mu && live) {  // seeds for d_loss = 1, as k_loss_bwd_small
        d_mu[(long long)b * Z + t] = (float)(1.0 / (double)B) * m;
        d_logvar[(long long)b * Z + t] = (float)(1.0 / (double)B) * (-0.5f) * (1.f - expf(lv));
      }
    }
    r += __shfl_xor(r, 1, 64);
    r += __shfl_xor(r, 2, 64);
    k += __shfl_xor(k, 1, 64);
    k += __shfl_xor(k, 2, 64);
    if (!live || sub != 0) continue;
    r += (double)NV * elem_const;
    k *= -0.5f;
    float q = 0.f;
    int am_hat = 0, am_y = 0;
    for (int c = 0; c < C; ++c) {
      const float yh = y_hat[(long long)b * C + c], yy = y[(long long)b * C + c];
      q = fmaf(yh, yy, q);
      if (yh > y_hat[(long long)b * C + am_hat]) am_hat = c;
      if (yy > y[(long long)b * C + am_y]) am_y = c;
    }
    if (d_yhat)
      for (int c = 0; c < C; ++c)
        d_yhat[(long long)b * C + c] = (float)(1.0 / (double)B) * (-2.f) * y[(long long)b * C + c] / q;
    const GT rec_t = (GT)r;
    rec[b] = rec_t;
    kld[b] = k;
    tot += (double)((GT)k + rec_t - (GT)2 * (GT)logf(q));
    corr += (am_hat == am_y) ? 1 : 0;
  }
  red[threadIdx.x] = tot;
  redc[threadIdx.x] = corr;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) {
      red[threadIdx.x] += red[threadIdx.x + off];
      redc[threadIdx.x] += redc[threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    loss[0] = (GT)(red[0] / (double)B);
    correct[0] = (long long)redc[0];
  }
}

template <typename GT>
__global__ void __launch_bounds__(256)
k_loss_bwd_recon(const float* __restrict__ recon, const GT* __restrict__ gt, const GT* __restrict__ d_loss,
                 double inv_var_over_B, float* __restrict__ d_recon, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double g = d_loss ? (double)d_loss[0] : 1.0;
  d_recon[i] = (float)(g * inv_var_over_B * ((double)recon[i] - (double)gt[i]));
}

template <typename GT>
__global__ void __launch_bounds__(256)
k_loss_bwd_small(const float* __restrict__ mu, const float* __restrict__ logvar, const float* __restrict__ y,
                 const float* __restrict__ y_hat, const GT* __restrict__ d_loss, float* __restrict__ d_mu,
                 float* __restrict__ d_logvar, float* __restrict__ d_yhat, int B, int C, int Z) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const float g = (float)((d_loss ? (double)d_loss[0] : 1.0) / (double)B);
  if (i < B * Z) {
    d_mu[i] = g * mu[i];
    d_logvar[i] = g * (-0.5f) * (1.f - expf(logvar[i]));
  }
  if (i < B) {
    float q = 0.f;
    for (int c = 0; c < C; ++c) q = fmaf(y_hat[(long long)i * C + c], y[(long long)i * C + c], q);
    for (int c = 0; c < C; ++c) d_yhat[(long long)i * C + c] = g * (-2.f) * y[(long long)i * C + c] / q;
  }
}

// main.py:88-93: mesh = bmm((recon * std + mean) * s, R) + m ; dist = ||mesh - gt||_2 per vertex
__global__ void __launch_bounds__(256)
k_recon_post(const float* __restrict__ recon, const float* __restrict__ std, const float* __restrict__ mean,
             const float* __restrict__ R, const float* __restrict__ m, const float* __restrict__ s,
             const float* __restrict__ gt, float* __restrict__ mesh_out, float* __restrict__ dist_out, long long rows,
             int N) {
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const long long b = r / N;
  const int v = (int)(r - b * N);
  const float sc = s[b];
  float p[3];
#pragma unroll
  for (int i = 0; i < 3; ++i)  // mul, add, mul with separate roundings, as the three torch ops
    p[i] = __fmul_rn(__fadd_rn(__fmul_rn(recon[r * 3 + i], std[v * 3 + i]), mean[v * 3 + i]), sc);
  const float* Rb = R + b * 9;
  float q[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    float a = __fmul_rn(p[0], Rb[j]);
    a = __fadd_rn(a, __fmul_rn(p[1], Rb[3 + j]));
    a = __fadd_rn(a, __fmul_rn(p[2], Rb[6 + j]));
    q[j] = a + m[b * 3 + j];
  }
  if (mesh_out) {
#pragma unroll
    for (int j = 0; j < 3; ++j) mesh_out[r * 3 + j] = q[j];
  }
  if (dist_out) {
    float d2 = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float d = gt[r * 3 + j] - q[j];
      d2 = __fadd_rn(d2, __fmul_rn(d, d));
    }
    dist_out[r] = sqrtf(d2);
  }
}

}  // namespace mvh

using namespace mvh;

extern "C" int mvh_recon_postprocess(mvh_stream_t stream, const float* recon, const float* std, const float* mean,
                                     const float* R, const float* m, const float* s, const float* gt,
                                     float* mesh_out, float* dist_out, int32_t B, int32_t N) {
  MVH_REQUIRE(recon && std && mean && R && m && s, "recon_postprocess: null tensor");
  MVH_REQUIRE(B >= 0 && N > 0, "recon_postprocess: bad sizes B=%d N=%d", B, N);
  MVH_REQUIRE(!dist_out || gt, "recon_postprocess: dist_out needs gt");
  MVH_REQUIRE(mesh_out || dist_out, "recon_postprocess: nothing to compute");
  const long long rows = (long long)B * N;
  if (rows == 0) return MVH_OK;
  hipLaunchKernelGGL(k_recon_post, dim3(cdiv(rows, 256)), dim3(256), 0, (hipStream_t)stream, recon, std, mean, R, m, s,
                     gt, mesh_out, dist_out, rows, N);
  MVH_LAUNCH_CHECK();
  return MVH_OK;
}

extern "C" size_t mvh_vae_loss_ws_bytes(int32_t B) { return (size_t)B * kLossSplit * sizeof(double) + 256; }

extern "C" int mvh_vae_loss_fwd(mvh_stream_t stream, const float* recon, const void* x_gt, int32_t gt_f64,
                                const float* mu, const float* logvar, const float* y, const float* y_hat,
                                float log_sigma, void* loss, void* rec, float* kld, int64_t* correct,
                                int32_t B, int32_t NV, int32_t C, int32_t Z, void* ws, size_t ws_bytes) {
  return loss_fwd_impl((hipStream_t)stream, recon, x_gt, gt_f64, mu, logvar, y, y_hat, log_sigma, loss, rec, kld,
                       correct, B, NV, C, Z, ws, ws_bytes, nullptr, nullptr, nullptr, nullptr);
}

int mvh::loss_fwd_impl(hipStream_t stream, const float* recon, const void* x_gt, int gt_f64, const float* mu,
                       const float* logvar, const float* y, const float* y_hat, float log_sigma, void* loss,
                       void* rec, float* kld, int64_t* correct, int B, int NV, int C, int Z, void* ws,
                       size_t ws_bytes, float* d_recon, float* d_mu, float* d_logvar, float* d_yhat,
                       const float* fuse_x16, const float* fuse_weff, float* fuse_recon, int fuse_cin, int fuse_c3,
                       int fuse_n_act) {
  MVH_REQUIRE(recon && x_gt && mu && logvar && y && y_hat && loss && rec && kld && correct, "loss_fwd: null tensor");
  LossFuse lf{fuse_x16, fuse_weff, fuse_recon, fuse_cin, fuse_c3, fuse_n_act};
  const bool fuse = fuse_x16 != nullptr;
  MVH_REQUIRE(!fuse || (fuse_weff && fuse_recon == recon && fuse_cin % 4 == 0 && fuse_c3 >= 1 && NV % fuse_c3 == 0 &&
                        fuse_n_act >= 0 && ((uintptr_t)fuse_x16 & 15) == 0),
              "loss_fwd: bad fused final-layer arguments");
  MVH_REQUIRE(B > 0 && NV > 0 && C > 0 && Z > 0, "loss_fwd: bad sizes");
  MVH_REQUIRE(ws && ws_bytes >= (size_t)B * kLossSplit * sizeof(double), "loss_fwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  double* partial = (double*)ws;
  // exp(log_sigma) in the precision the reference uses (fp32 tensor; cheb_VAE.py:329-330)
  const double sigma = (double)expf(log_sigma);
  const double elem_const = (double)(float)log_sigma + 0.5 * 1.8378770664093453 /* ln(2 pi) */;
  const double ivb = 1.0 / (sigma * sigma) / (double)B;
  MVH_REQUIRE(!d_mu || (d_logvar && d_yhat), "loss_fwd: incomplete gradient-seed outputs");
  if (gt_f64) {
    if (fuse)
      hipLaunchKernelGGL((k_loss_partial<double, true>), dim3(kLossSplit, B), dim3(256), 0, st, recon, (const double*)x_gt,
                         1.0 / sigma, partial, NV, d_recon, ivb, lf);
    else
      hipLaunchKernelGGL((k_loss_partial<double, false>), dim3(kLossSplit, B), dim3(256), 0, st, recon, (const double*)x_gt,
                         1.0 / sigma, partial, NV, d_recon, ivb, lf);
    MVH_LAUNCH_CHECK();
    hipLaunchKernelGGL((k_loss_finish<double>), dim3(1), dim3(256), 0, st, partial, mu, logvar, y, y_hat,
                       elem_const, (double*)loss, (double*)rec, kld, (long long*)correct, B, NV, C, Z, kLossSplit,
                       d_mu, d_logvar, d_yhat);
  } else {
    if (fuse)
      hipLaunchKernelGGL((k_loss_partial<float, true>), dim3(kLossSplit, B), dim3(256), 0, st, recon, (const float*)x_gt,
                         1.0 / sigma, partial, NV, d_recon, ivb, lf);
    else
      hipLaunchKernelGGL((k_loss_partial<float, false>), dim3(kLossSplit, B), dim3(256), 0, st, recon, (const float*)x_gt,
                         1.0 / sigma, partial, NV, d_recon, ivb, lf);
    MVH_LAUNCH_CHECK();
    hipLaunchKernelGGL((k_loss_finish<float>), dim3(1), dim3(256), 0, st, partial, mu, logvar, y, y_hat,
                       elem_const, (float*)loss, (float*)rec, kld, (long long*)correct, B, NV, C, Z, kLossSplit,
                       d_mu, d_logvar, d_yhat);
  }
  MVH_LAUNCH_CHECK();
  return MVH_OK;
}

extern "C" int mvh_vae_loss_bwd(mvh_stream_t stream, const float* recon, const void* x_gt, int32_t gt_f64,
                                const float* mu, const float* logvar, const float* y, const float* y_hat,
                                float log_sigma, const void* d_loss, float* d_recon, float* d_mu,
                                float* d_logvar, float* d_yhat, int32_t B, int32_t NV, int32_t C, int32_t Z) {
  MVH_REQUIRE(recon && x_gt && mu && logvar && y && y_hat && d_recon && d_mu && d_logvar && d_yhat, "loss_bwd: null tensor");
  MVH_REQUIRE(B > 0 && NV > 0 && C > 0 && Z > 0, "loss_bwd: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  const double sigma = (double)expf(log_sigma);
  const double ivb = 1.0 / (sigma * sigma) / (double)B;
  const long long n = (long long)B * NV;
  const int small = max(B * Z, B);
  if (gt_f64) {
    hipLaunchKernelGGL((k_loss_bwd_recon<double>), dim3(cdiv(n, 256)), dim3(256), 0, st, recon, (const double*)x_gt,
                       (const double*)d_loss, ivb, d_recon, n);
    MVH_LAUNCH_CHECK();
    hipLaunchKernelGGL((k_loss_bwd_small<double>), dim3(cdiv(small, 256)), dim3(256), 0, st, mu, logvar, y, y_hat,
                       (const double*)d_loss, d_mu, d_logvar, d_yhat, B, C, Z);
  } else {
    hipLaunchKernelGGL((k_loss_bwd_recon<float>), dim3(cdiv(n, 256)), dim3(256), 0, st, recon, (const float*)x_gt,
                       (const float*)d_loss, ivb, d_recon, n);
    MVH_LAUNCH_CHECK();
    hipLaunchKernelGGL((k_loss_bwd_small<float>), dim3(cdiv(small, 256)), dim3(256), 0, st, mu, logvar, y, y_hat,
                       (const float*)d_loss, d_mu, d_logvar, d_yhat, B, C, Z);
  }
  MVH_LAUNCH_CHECK();
  return MVH_OK;
}
