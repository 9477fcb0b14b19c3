// Micro-benchmark / feasibility probe of the asynchronous launcher's hand-over (DESIGN: "launcher thread"):
//   caller thread : hipEventRecord(ev, U) -> queue job k -> hipStreamWaitValue32(U, flag, k, >=) -> consumer kernel on U
//   worker thread : hipStreamWaitEvent(S, ev) -> producer kernels on S -> hipStreamWriteValue32(S, flag, k)
// U is a normal-priority stream (torch's current stream in the product), S a HIGH-priority non-blocking stream (a queue
// pool of its own in the runtime: S can never sit behind U's blocked wait packet in a shared hardware queue).
// The consumer checks that it sees the producer's data of ITS ticket (ordering), every wait is bounded by the process
// `timeout` of the caller.  Prints the per-iteration cost against the plain same-stream form.
//   hipcc --offload-arch=gfx950 -O2 -o tools/ubench/waitvalue tools/ubench/waitvalue.hip -lpthread
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <deque>
#include <mutex>
#include <thread>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

__global__ void k_produce(int* data, int n, int ticket) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) data[i] = ticket;
}
__global__ void k_consume(const int* data, int n, int ticket, int* bad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && data[i] != ticket) atomicAdd(bad, 1);
}

struct Job { int ticket; hipEvent_t ev; };

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  const int chain = argc > 2 ? atoi(argv[2]) : 40;      // producer launches per job (the native step: ~40 per call)
  const int null_u = argc > 3 ? atoi(argv[3]) : 0;      // 1: U is the legacy null stream (torch's default stream)
  const int high_s = argc > 4 ? atoi(argv[4]) : 1;      // 1: S at the highest priority
  const int wide = argc > 5 ? atoi(argv[5]) : 0;        // 1: 64-bit wait / write
  int can = 0;
  CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
  printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
  if (!can) return 2;
  int lo = 0, hi = 0;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  printf("stream priority range: least %d greatest %d\n", lo, hi);
  hipStream_t U, S;
  if (null_u) U = nullptr; else CK(hipStreamCreateWithFlags(&U, hipStreamNonBlocking));
  CK(hipStreamCreateWithPriority(&S, hipStreamNonBlocking, high_s ? hi : 0));
  printf("U = %s, S priority %s, %d-bit values\n", null_u ? "null stream" : "non-blocking stream", high_s ? "high" : "normal", wide ? 64 : 32);
  const int n = 1 << 16;
  int *data, *bad;
  CK(hipMalloc(&data, n * sizeof(int)));
  CK(hipMalloc(&bad, sizeof(int)));
  CK(hipMemset(bad, 0, sizeof(int)));
  uint32_t* flag = nullptr;
  CK(hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory));
  CK(hipMemset(flag, 0, 8));
  CK(hipDeviceSynchronize());
  hipEvent_t evs[64];
  for (auto& e : evs) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));

  // ---- reference: everything on U from one thread
  auto t0 = std::chrono::steady_clock::now();
  for (int k = 1; k <= iters; ++k) {
    for (int c = 0; c < chain; ++c) hipLaunchKernelGGL(k_produce, dim3(n / 256), dim3(256), 0, U, data, n, k);
    hipLaunchKernelGGL(k_consume, dim3(n / 256), dim3(256), 0, U, data, n, k, bad);
  }
  auto t1 = std::chrono::steady_clock::now();
  CK(hipStreamSynchronize(U));
  auto t2 = std::chrono::steady_clock::now();
  auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
  printf("same-stream form : host enqueue %.1f us/iter, wall %.1f us/iter\n", us(t0, t1) / iters, us(t0, t2) / iters);

  // ---- launcher form
  std::mutex mu;
  std::condition_variable cv;
  std::deque<Job> q;
  bool stop = false;
  std::atomic<int> werr{0};
  std::thread worker([&] {
    hipSetDevice(0);
    for (;;) {
      Job j;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return stop || !q.empty(); });
        if (q.empty()) return;
        j = q.front();
        q.pop_front();
      }
      if (hipStreamWaitEvent(S, j.ev, 0) != hipSuccess) werr = 1;
      for (int c = 0; c < chain; ++c) hipLaunchKernelGGL(k_produce, dim3(n / 256), dim3(256), 0, S, data, n, j.ticket);
      if ((wide ? hipStreamWriteValue64(S, flag, (uint64_t)j.ticket, 0) : hipStreamWriteValue32(S, flag, (uint32_t)j.ticket, 0)) != hipSuccess) werr = 2;   // ALWAYS written, whatever failed above
    }
  });
  t0 = std::chrono::steady_clock::now();
  for (int k = 1; k <= iters; ++k) {
    hipEvent_t ev = evs[k % 64];
    CK(hipEventRecord(ev, U));                           // the consumer of ticket k-1 has read `data` behind this point
    {
      std::lock_guard<std::mutex> lk(mu);
      q.push_back(Job{k, ev});
    }
    cv.notify_one();
    if (wide) CK(hipStreamWaitValue64(U, flag, (uint64_t)k, hipStreamWaitValueGte, ~0ull));
    else CK(hipStreamWaitValue32(U, flag, (uint32_t)k, hipStreamWaitValueGte, 0xffffffffu));
    hipLaunchKernelGGL(k_consume, dim3(n / 256), dim3(256), 0, U, data, n, k, bad);
    while (true) {                                       // bounded queue (the event ring has 64 slots)
      std::lock_guard<std::mutex> lk(mu);
      if (q.size() < 32) break;
    }
  }
  t1 = std::chrono::steady_clock::now();
  CK(hipStreamSynchronize(U));
  t2 = std::chrono::steady_clock::now();
  {
    std::lock_guard<std::mutex> lk(mu);
    stop = true;
  }
  cv.notify_one();
  worker.join();
  CK(hipDeviceSynchronize());
  int h_bad = -1;
  CK(hipMemcpy(&h_bad, bad, sizeof(int), hipMemcpyDeviceToHost));
  printf("launcher form    : caller enqueue %.1f us/iter, wall %.1f us/iter, ordering violations %d, worker err %d\n",
         us(t0, t1) / iters, us(t0, t2) / iters, h_bad, werr.load());
  return h_bad == 0 && werr == 0 ? 0 : 3;
}
