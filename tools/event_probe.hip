// What does a cross-stream fork cost on the recording stream?  (build: hipcc --offload-arch=gfx950 -O2 -o gpurun_out/event_probe tools/event_probe.hip)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_small(float* p, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = p[i] * 1.0001f + 1.f;
}
__global__ void k_flag(float* p, int n, volatile unsigned* flag, unsigned seq) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) *flag = seq;
  if (i < n) p[i] = p[i] * 1.0001f + 1.f;
}

int main() {
  const int n = 1 << 16, N = 200;
  float *a, *b;
  CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4));
  CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4));
  hipStream_t s0, s1;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  std::vector<hipEvent_t> ev(N);
  for (auto& evi : ev) CK(hipEventCreateWithFlags(&evi, hipEventDisableTiming));
  unsigned* flag = nullptr;
  hipError_t fe = hipExtMallocWithFlags((void**)&flag, 64, hipMallocSignalMemory);
  if (fe != hipSuccess) { printf("signal memory alloc failed: %s\n", hipGetErrorString(fe)); flag = nullptr; (void)hipGetLastError(); }
  if (flag) CK(hipMemset(flag, 0, 64));
  hipEvent_t t0, t1;
  CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  auto run = [&](const char* name, int mode) -> int {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipDeviceSynchronize());
      if (flag) CK(hipMemset(flag, 0, 64));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(t0, s0));
      for (int i = 0; i < N; ++i) {
        if (mode == 4 || mode == 5) hipLaunchKernelGGL(k_flag, dim3(n / 256), dim3(256), 0, s0, a, n, flag, (unsigned)(i + 1));
        else hipLaunchKernelGGL(k_small, dim3(n / 256), dim3(256), 0, s0, a, n);
        if (mode == 1 || mode == 2) CK(hipEventRecord(ev[i], s0));
        if (mode == 2) { CK(hipStreamWaitEvent(s1, ev[i], 0)); hipLaunchKernelGGL(k_small, dim3(n / 256), dim3(256), 0, s1, b, n); }
        if (mode == 3) CK(hipStreamWriteValue32(s0, flag, (unsigned)(i + 1), 0));
        if (mode == 5) { CK(hipStreamWaitValue32(s1, flag, (unsigned)(i + 1), hipStreamWaitValueGte, 0xffffffffu)); hipLaunchKernelGGL(k_small, dim3(n / 256), dim3(256), 0, s1, b, n); }
      }
      CK(hipEventRecord(t1, s0));
      CK(hipEventSynchronize(t1));
      CK(hipDeviceSynchronize());
      float ms = 0;
      CK(hipEventElapsedTime(&ms, t0, t1));
      if (rep == 1) printf("%-58s %.2f us per kernel on the main stream\n", name, ms * 1e3 / N);
    }
    return 0;
  };
  if (run("plain chain", 0)) return 1;
  if (run("+ hipEventRecord after each", 1)) return 1;
  if (run("+ record, side stream waits and runs a kernel", 2)) return 1;
  if (flag) {
    if (run("+ hipStreamWriteValue32 after each", 3)) return 1;
    if (run("kernel writes a flag (no stream op)", 4)) return 1;
    if (run("kernel writes flag, side hipStreamWaitValue32 + kernel", 5)) return 1;
  }
  return 0;
}
