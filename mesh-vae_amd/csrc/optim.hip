// Optimizer step of the train loop (reference main.py:251 torch.optim.Adam with coupled L2
// weight_decay, main.py:80-81): one fused elementwise pass over the FLAT parameter / gradient
// buffers (712,642 fp32 at default.cfg), graph-capturable: the step count lives on the device.
#include "common.hpp"

namespace mvh {

__global__ void k_adam_tick(int* step) { *step += 1; }

// host_step > 0: the caller counts the steps (eager launch sequences: no tick launch); the device counter is kept
// in sync by thread 0 so that a later graph capture continues from the right value.  host_step == 0: the
// counter on the device is authoritative (hipGraph replay: k_adam_tick ran just before).
// One thread updates FOUR consecutive parameters (16-byte loads / stores of p, g, m, v; the two powf of the bias
// corrections once per four elements): 8.6 -> see DESIGN section 5 us at 712 642 parameters.  Per element the arithmetic
// is the scalar form's, operation for operation (results bitwise those of the one-element-per-thread kernel).
__device__ __forceinline__ void adam_one(float& pi, float gi, float& mi, float& vi, float lr_bc1, float rs_bc2, float b1,
                                         float b2, float eps, float wd, float grad_scale) {
  gi *= grad_scale;
  gi = fmaf(wd, pi, gi);                       // coupled L2, as torch.optim.Adam(weight_decay=)
  mi = fmaf(b1, mi, (1.f - b1) * gi);
  vi = fmaf(b2, vi, (1.f - b2) * gi * gi);
  const float denom = sqrtf(vi) / rs_bc2 + eps;
  pi = pi - lr_bc1 * (mi / denom);
}

__global__ void __launch_bounds__(256)
k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
       long long n, float lr, float b1, float b2, float eps, float wd, float grad_scale,
       int* __restrict__ step, int host_step, long long skip_lo, long long skip_hi, int vec_ok) {
  const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (host_step > 0 && i == 0) *step = host_step;
  if (i >= n) return;
  const float t = host_step > 0 ? (float)host_step : (float)(*step);
  const float bc1 = 1.f - powf(b1, t), bc2 = 1.f - powf(b2, t);
  const float lr_bc1 = lr / bc1, rs_bc2 = sqrtf(bc2);
  // parameters that never receive a gradient (torch.optim.Adam skips `p.grad is None`: no decay, no state)
  auto skipped = [&](long long j) { return j >= skip_lo && j < skip_hi; };
  if (vec_ok && i + 3 < n && (skip_hi <= i || skip_lo > i + 3)) {   // (vec_ok: all four buffers 16-byte aligned)
    float4 pp = *reinterpret_cast<const float4*>(p + i), mm = *reinterpret_cast<const float4*>(m + i);
    float4 vv = *reinterpret_cast<const float4*>(v + i);
    const float4 gg = *reinterpret_cast<const float4*>(g + i);
    adam_one(pp.x, gg.x, mm.x, vv.x, lr_bc1, rs_bc2, b1, b2, eps, wd, grad_scale);
    adam_one(pp.y, gg.y, mm.y, vv.y, lr_bc1, rs_bc2, b1, b2, eps, wd, grad_scale);
    adam_one(pp.z, gg.z, mm.z, vv.z, lr_bc1, rs_bc2, b1, b2, eps, wd, grad_scale);
    adam_one(pp.w, gg.w, mm.w, vv.w, lr_bc1, rs_bc2, b1, b2, eps, wd, grad_scale);
    *reinterpret_cast<float4*>(p + i) = pp;
    *reinterpret_cast<float4*>(m + i) = mm;
    *reinterpret_cast<float4*>(v + i) = vv;
    return;
  }
  const long long j_end = i + 4 < n ? i + 4 : n;
  for (long long j = i; j < j_end; ++j) {   // the tail of the buffer and the groups a skipped range cuts
    if (skipped(j)) continue;
    float pj = p[j], mj = m[j], vj = v[j];
    adam_one(pj, g[j], mj, vj, lr_bc1, rs_bc2, b1, b2, eps, wd, grad_scale);
    p[j] = pj;
    m[j] = mj;
    v[j] = vj;
  }
}

}  // namespace mvh

using namespace mvh;

static int adam_vec_ok(const void* a, const void* b, const void* c, const void* d) {
  return ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d) & 15) == 0) ? 1 : 0;
}

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
  hipLaunchKernelGGL(k_adam, dim3(cdiv(n, 1024)), dim3(256), 0, st, param, grad, exp_avg, exp_avg_sq,
                     (long long)n, lr, beta1, beta2, eps, weight_decay, grad_scale, step_count, 0, (long long)skip_lo,
                     (long long)skip_hi, adam_vec_ok(param, grad, exp_avg, exp_avg_sq));
  MVH_LAUNCH_CHECK();
  return MVH_OK;
}

extern "C" int mvh_adam_step_counted(mvh_stream_t stream, float* param, const float* grad, float* exp_avg,
                                     float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2, float eps,
                                     float weight_decay, float grad_scale, int32_t* step_count, int32_t step,
                                     int64_t skip_lo, int64_t skip_hi) {
  MVH_REQUIRE(param && grad && exp_avg && exp_avg_sq && step_count, "adam_step: null tensor");
  MVH_REQUIRE(n > 0 && step > 0, "adam_step_counted: bad size or step");
  hipLaunchKernelGGL(k_adam, dim3(cdiv(n, 1024)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq,
                     (long long)n, lr, beta1, beta2, eps, weight_decay, grad_scale, step_count, (int)step,
                     (long long)skip_lo, (long long)skip_hi, adam_vec_ok(param, grad, exp_avg, exp_avg_sq));
  MVH_LAUNCH_CHECK();
  return MVH_OK;
}
