// Optimizer step of the train loop (reference main.py:251 torch.optim.Adam with coupled L2
// weight_decay, main.py:80-81): one fused elementwise pass over the FLAT parameter / gradient
// buffers (712,642 fp32 at default.cfg), graph-capturable: the step count lives on the device.
#include "common.hpp"

namespace mvh {

__global__ void k_adam_tick(int* step) { *step += 1; }

// host_step > 0: the caller counts the steps (eager launch sequences: no tick launch); the device counter is kept
// in sync by thread 0 so that a later graph capture continues from the right value.  host_step == 0: the
// counter on the device is authoritative (hipGraph replay: k_adam_tick ran just before).
__global__ void __launch_bounds__(256)
k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
       long long n, float lr, float b1, float b2, float eps, float wd, float grad_scale,
       int* __restrict__ step, int host_step, long long skip_lo, long long skip_hi) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (host_step > 0 && i == 0) *step = host_step;
  if (i >= n) return;
  // parameters that never receive a gradient (torch.optim.Adam skips `p.grad is None`: no decay, no state)
  if (i >= skip_lo && i < skip_hi) return;
  const float t = host_step > 0 ? (float)host_step : (float)(*step);
  const float bc1 = 1.f - powf(b1, t), bc2 = 1.f - powf(b2, t);
  float gi = g[i] * grad_scale;
  const float pi = p[i];
  gi = fmaf(wd, pi, gi);                       // coupled L2, as torch.optim.Adam(weight_decay=)
  const float mi = fmaf(b1, m[i], (1.f - b1) * gi);
  const float vi = fmaf(b2, v[i], (1.f - b2) * gi * gi);
  m[i] = mi;
  v[i] = vi;
  const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
  p[i] = pi - (lr / bc1) * (mi / denom);
}

}  // namespace mvh

using namespace mvh;

extern "C" int mvh_adam_step(mvh_stream_t stream, float* param, const float* grad, float* exp_avg,
                             float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2, float eps,
                             float weight_decay, float grad_scale, int32_t* step_count, int64_t skip_lo,
                             int64_t skip_hi) {
  MVH_REQUIRE(param && grad && exp_avg && exp_avg_sq && step_count, "adam_step: null tensor");
  MVH_REQUIRE(n >= 0, "adam_step: bad size");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_adam_tick, dim3(1), dim3(1), 0, st, step_count);
  MVH_LAUNCH_CHECK();
  if (n == 0) return MVH_OK;
  hipLaunchKernelGGL(k_adam, dim3(cdiv(n, 256)), dim3(256), 0, st, param, grad, exp_avg, exp_avg_sq,
                     (long long)n, lr, beta1, beta2, eps, weight_decay, grad_scale, step_count, 0, (long long)skip_lo,
                     (long long)skip_hi);
  MVH_LAUNCH_CHECK();
  return MVH_OK;
}

extern "C" int mvh_adam_step_counted(mvh_stream_t stream, float* param, const float* grad, float* exp_avg,
                                     float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2, float eps,
                                     float weight_decay, float grad_scale, int32_t* step_count, int32_t step,
                                     int64_t skip_lo, int64_t skip_hi) {
  MVH_REQUIRE(param && grad && exp_avg && exp_avg_sq && step_count, "adam_step: null tensor");
  MVH_REQUIRE(n > 0 && step > 0, "adam_step_counted: bad size or step");
  hipLaunchKernelGGL(k_adam, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq,
                     (long long)n, lr, beta1, beta2, eps, weight_decay, grad_scale, step_count, (int)step,
                     (long long)skip_lo, (long long)skip_hi);
  MVH_LAUNCH_CHECK();
  return MVH_OK;
}
