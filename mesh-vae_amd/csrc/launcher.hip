// The asynchronous launcher: mvh_vae_forward / mvh_vae_backward enqueued by a worker thread of the library on a stream of
// its own, while the caller's thread goes on.
//
// Why.  The reference's train loop (main.py:74-81: optimizer.zero_grad() -> model(...) -> loss.backward() ->
// optimizer.step()) is HOST-bound on this path: the native step costs the calling thread ~0.35 ms of launch calls
// (~85 launches at ~4 us each on ROCm 7.2) and torch.optim.Adam another ~0.35 ms of Python, one after the other, for
// ~0.45 ms of GPU work (profiles/r04_ref_loop_probe.txt).  A hipGraph would remove the launch calls but replays this
// step 25-40 % slower than the eager lanes (profiles/r04_graph_env_sweep.txt).  So the launch calls move to another
// thread -- C++, no GIL -- and the two host costs overlap.
//
// How the order is kept without the caller waiting for the worker.  A stream can only wait for an event that has been
// recorded already, and the worker records "job k done" long after the caller has returned.  The caller therefore leaves
// a VALUE wait on its own stream instead: hipStreamWaitValue64(user_stream, flag, k, >=), and the worker's last packet
// of job k is hipStreamWriteValue64(S, flag, k).  In the other direction the caller records an event on its stream
// when it hands the job over (everything the job reads was produced before that point) and the worker makes S wait for
// it before the job's first launch.  Every async call thus leaves the caller's stream ordered behind the job: later
// work on that stream -- the consumer of an output, the next allocation that reuses a freed input -- sees the job done.
//
// The one way this could hang is a stream the JOB uses sharing a hardware queue with the caller's stream (its packets
// would sit behind the blocked wait).  The runtime pools hardware queues per PRIORITY: S is created with the highest
// stream priority, and a step launched on a highest-priority stream runs its weight-gradient work INLINE on that stream
// instead of forking it onto the (default-priority) lanes (csrc/vae_step.hip, lanes_for), so nothing of a job ever queues
// behind a default-priority stream of the application -- whatever GPU_MAX_HW_QUEUES is and however many streams the
// application has (test: tests/test_gpu_engine.py::test_async_launcher_with_two_hardware_queues_and_many_streams).
// The value is ALWAYS written, whatever the job returned or threw; the first error of a job is kept and returned by the
// next call on the launcher (or by mvh_launcher_sync / at destruction).  Arguments are validated on the CALLER's thread
// before anything is queued, so a bad call fails synchronously like the plain entry points.
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "common.hpp"

struct mvh_launcher {
  int dev = 0;
  hipStream_t S = nullptr;
  uint64_t* flag = nullptr;      // hipMallocSignalMemory: the ticket of the last job whose work S has finished
  uint64_t ticket = 0;           // caller side: last ticket handed out
  hipEvent_t ev[64];
  struct Job { uint64_t ticket; hipEvent_t ev; std::function<int(hipStream_t)> fn; int kind; };
  std::mutex mu;
  std::condition_variable cv_job, cv_space;
  std::deque<Job> q;
  bool busy = false, stop = false;
  std::atomic<uint64_t> pushed{0};   // (the worker polls this for a short while before it sleeps)
  int err = MVH_OK;              // first failure of a job since the last report
  std::string errmsg;
  std::thread th;
  bool trace = false;
  double t_fn = 0, t_wr = 0;
  long n_tr = 0;
  bool high_prio = true;
  // The worker polls for the next job for a short while before it sleeps (a futex wake-up costs 30-60 us) -- but only where a
  // job usually follows soon: after a forward comes its backward (~0.25 ms), after a backward the caller's optimizer
  // (0.3-0.5 ms of Python) -- so the poll is ADAPTIVE: the idle gap that followed each kind of job is tracked (running
  // mean) and the worker polls at most 1.5 x that, never more than spin_us, and not at all when the gap is longer.
  int spin_us = 300;
  double gap_us[2] = {0.0, 1e9};     // [kind]: running mean of the idle time after a job of that kind (0 forward, 1 backward)
  std::atomic<bool> alive{false};
  static constexpr size_t kDepth = 4;   // jobs queued on the host side (a step is two; the binding keeps the tensors of the last 6 calls)

  void run() {
    (void)hipSetDevice(dev);
    alive.store(true, std::memory_order_release);
    uint64_t seen = 0;
    int last_kind = 1;
    for (;;) {
      Job j;
      {
        const auto t0 = std::chrono::steady_clock::now();
        const double budget = std::min((double)spin_us, 1.5 * gap_us[last_kind]);
        if (gap_us[last_kind] <= (double)spin_us) {
          while (pushed.load(std::memory_order_acquire) == seen &&
                 std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < budget) {
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
          }
        }
        std::unique_lock<std::mutex> lk(mu);
        cv_job.wait(lk, [&] { return stop || !q.empty(); });
        if (q.empty()) break;
        const double gap = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        gap_us[last_kind] = gap_us[last_kind] > 1e8 ? gap : 0.75 * gap_us[last_kind] + 0.25 * std::min(gap, 1e5);
        j = std::move(q.front());
        q.pop_front();
        ++seen;
        busy = true;
      }
      cv_space.notify_all();
      last_kind = j.kind & 1;
      int rc = MVH_OK;
      std::string msg;
      const auto tj0 = std::chrono::steady_clock::now();
      hipError_t e = hipStreamWaitEvent(S, j.ev, 0);
      if (e != hipSuccess) { rc = MVH_ERR_HIP; msg = std::string("launcher: hipStreamWaitEvent failed: ") + hipGetErrorString(e); }
      if (rc == MVH_OK) {
        // whatever the job does -- return an error, throw -- its ticket is written below: the caller's stream waits for it
        try {
          rc = j.fn(S);
          if (rc != MVH_OK) msg = mvh::last_error_buf();   // (thread-local: this thread's message)
        } catch (const std::exception& ex) {
          rc = MVH_ERR_INVALID;
          msg = std::string("launcher: the job threw: ") + ex.what();
        } catch (...) {
          rc = MVH_ERR_INVALID;
          msg = "launcher: the job threw an unknown exception";
        }
      }
      const auto tj1 = std::chrono::steady_clock::now();
      e = hipStreamWriteValue64(S, flag, j.ticket, 0);   // ALWAYS: the caller's stream is waiting for this value
      if (trace) {
        const auto tj2 = std::chrono::steady_clock::now();
        t_fn += std::chrono::duration<double, std::micro>(tj1 - tj0).count();
        t_wr += std::chrono::duration<double, std::micro>(tj2 - tj1).count();
        if (++n_tr % 128 == 0) {
          fprintf(stderr, "[mvh launcher] per job: wait+launches %.1f us, write-value %.1f us (128 jobs)\n", t_fn / 128, t_wr / 128);
          t_fn = t_wr = 0;
        }
      }
      if (e != hipSuccess && rc == MVH_OK) { rc = MVH_ERR_HIP; msg = std::string("launcher: hipStreamWriteValue64 failed: ") + hipGetErrorString(e); }
      {
        std::lock_guard<std::mutex> lk(mu);
        if (rc != MVH_OK && err == MVH_OK) { err = rc; errmsg = msg; }
        busy = false;
      }
      cv_space.notify_all();
    }
    alive.store(false, std::memory_order_release);
    cv_space.notify_all();
  }
};

using namespace mvh;

// the kept error of an earlier job, reported once (as this call's failure)
static int report_kept(mvh_launcher* L) {
  std::lock_guard<std::mutex> lk(L->mu);
  if (L->err == MVH_OK) return MVH_OK;
  const int rc = L->err;
  L->err = MVH_OK;
  return fail(rc, "asynchronous job failed: %s", L->errmsg.c_str());
}

static int submit(mvh_launcher* L, hipStream_t user, int kind, std::function<int(hipStream_t)> fn) {
  MVH_REQUIRE(L != nullptr, "launcher: null handle");
  if (int rc = report_kept(L)) return rc;
  MVH_REQUIRE(L->alive.load(std::memory_order_acquire), "launcher: the worker thread is gone (nothing was queued)");
  int dev = -1;
  MVH_HIP(hipGetDevice(&dev));
  MVH_REQUIRE(dev == L->dev, "launcher: created on device %d, called on device %d", L->dev, dev);
  uint64_t t = 0;
  {  // ticket, event and queue slot under one lock: callers may be different threads (the forward comes from the
     // application's thread, the backward from the autograd engine's), and tickets must enter the queue in order
    std::unique_lock<std::mutex> lk(L->mu);
    L->cv_space.wait(lk, [&] { return L->q.size() < mvh_launcher::kDepth; });
    t = L->ticket + 1;
    hipEvent_t ev = L->ev[t % 64];
    MVH_HIP(hipEventRecord(ev, user));
    L->ticket = t;
    L->q.push_back(mvh_launcher::Job{t, ev, std::move(fn), kind});
  }
  L->pushed.fetch_add(1, std::memory_order_release);
  L->cv_job.notify_one();
  // (from here on the job WILL write the value: a failure below leaves nothing blocked)
  MVH_HIP(hipStreamWaitValue64(user, L->flag, t, hipStreamWaitValueGte, ~0ull));
  return MVH_OK;
}

extern "C" int mvh_launcher_supported(void) {
  int dev = 0, can = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, dev) != hipSuccess) return 0;
  return can ? 1 : 0;
}

static void launcher_free(mvh_launcher* L) {     // (no worker thread running)
  if (!L) return;
  (void)hipSetDevice(L->dev);
  if (L->S) (void)hipStreamSynchronize(L->S);
  for (auto& e : L->ev)
    if (e) (void)hipEventDestroy(e);
  if (L->S) (void)hipStreamDestroy(L->S);
  if (L->flag) (void)hipFree(L->flag);
  delete L;
}

static int launcher_init(mvh_launcher* L) {
  MVH_HIP(hipGetDevice(&L->dev));
  int least = 0, greatest = 0;
  MVH_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
  if (const char* e = getenv("MESHVAE_ASYNC_PRIO")) L->high_prio = atoi(e) != 0;      // (A/B tooling)
  if (const char* e = getenv("MESHVAE_ASYNC_SPIN_US")) L->spin_us = atoi(e);
  if (const char* e = getenv("MESHVAE_ASYNC_TRACE")) L->trace = atoi(e) != 0;
  MVH_HIP(hipStreamCreateWithPriority(&L->S, hipStreamNonBlocking, L->high_prio ? greatest : 0));
  MVH_HIP(hipExtMallocWithFlags((void**)&L->flag, 8, hipMallocSignalMemory));
  MVH_HIP(hipMemset(L->flag, 0, 8));
  MVH_HIP(hipDeviceSynchronize());
  for (auto& e : L->ev) MVH_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence));
  return MVH_OK;
}

extern "C" int mvh_launcher_create(mvh_launcher_t** out) {
  MVH_REQUIRE(out != nullptr, "launcher_create: null argument");
  *out = nullptr;
  MVH_REQUIRE(mvh_launcher_supported(), "launcher_create: this device has no hipStreamWaitValue64");
  mvh_launcher* L = new mvh_launcher();
  for (auto& e : L->ev) e = nullptr;
  if (int rc = launcher_init(L)) {     // (whatever was created so far is released: stream, signal memory, events)
    launcher_free(L);
    return rc;
  }
  L->th = std::thread([L] { L->run(); });
  while (!L->alive.load(std::memory_order_acquire)) std::this_thread::yield();
  *out = L;
  return MVH_OK;
}

extern "C" int mvh_launcher_sync(mvh_launcher_t* L) {
  MVH_REQUIRE(L != nullptr, "launcher_sync: null handle");
  {
    // (a watchdog, not an unconditional wait: if the worker is gone with jobs still queued, say so instead of hanging)
    std::unique_lock<std::mutex> lk(L->mu);
    while (!(L->q.empty() && !L->busy)) {
      if (!L->alive.load(std::memory_order_acquire))
        return fail(MVH_ERR_INVALID, "launcher_sync: the worker thread is gone with %zu job(s) still queued", L->q.size());
      L->cv_space.wait_for(lk, std::chrono::milliseconds(50));
    }
  }
  return report_kept(L);
}

extern "C" int mvh_launcher_destroy(mvh_launcher_t* L) {
  if (!L) return MVH_OK;
  {
    std::lock_guard<std::mutex> lk(L->mu);
    L->stop = true;               // (the worker drains the queue first: every handed-over job still writes its value)
  }
  L->pushed.fetch_add(1, std::memory_order_release);
  L->cv_job.notify_all();
  if (L->th.joinable()) L->th.join();
  launcher_free(L);
  return MVH_OK;
}

// test aid (tests/test_gpu_engine.py): a job that throws / returns an error on the worker -- the ticket must still be written
extern "C" int mvh_launcher_test_job(mvh_launcher_t* L, mvh_stream_t user_stream, int32_t mode) {
  return submit(L, (hipStream_t)user_stream, 0, [mode](hipStream_t) -> int {
    if (mode == 1) throw std::runtime_error("test job: deliberate exception");
    if (mode == 2) return fail(MVH_ERR_INVALID, "test job: deliberate error code");
    return MVH_OK;
  });
}

extern "C" int mvh_vae_forward_async(mvh_launcher_t* L, mvh_stream_t user_stream, const mvh_vae_desc_t* desc,
                                     const float* const* params, const float* x, const float* y, const void* x_gt,
                                     int32_t gt_f64, const float* eps, const float* drop_u, int32_t B, float log_sigma,
                                     void* loss, int64_t* correct, float* recon, float* kld, void* rec, float* z,
                                     float* y_hat, float* mu, float* logvar, void* ws, size_t ws_bytes) {
  MVH_REQUIRE(desc && params, "vae_forward_async: null descriptor / parameter table");
  // (validated HERE, on the caller's thread: a bad call fails like the plain entry point, not on a later launcher call)
  MVH_REQUIRE(x && y && x_gt && loss && correct && recon && kld && rec && z && y_hat && mu && logvar, "vae_forward_async: null tensor");
  if (int rc = vae_step_precheck(desc, B, ws, ws_bytes)) return rc;
  const int np = mvh_vae_param_count(desc);
  // the caller may reuse its descriptor and pointer table as soon as this returns: the job owns copies
  auto d = std::make_shared<mvh_vae_desc_t>(*desc);
  auto P = std::make_shared<std::vector<const float*>>(params, params + np);
  return submit(L, (hipStream_t)user_stream, 0, [=](hipStream_t S) {
    return mvh_vae_forward((mvh_stream_t)S, d.get(), P->data(), x, y, x_gt, gt_f64, eps, drop_u, B, log_sigma, loss, correct,
                           recon, kld, rec, z, y_hat, mu, logvar, ws, ws_bytes);
  });
}

extern "C" int mvh_vae_backward_async(mvh_launcher_t* L, mvh_stream_t user_stream, const mvh_vae_desc_t* desc,
                                      const float* const* params, float* const* grads, const float* x, const float* y,
                                      const void* x_gt, int32_t gt_f64, const float* eps, const float* drop_u, int32_t B,
                                      float log_sigma, const void* d_loss, const float* recon, const float* y_hat,
                                      const float* mu, const float* logvar, void* ws, size_t ws_bytes) {
  MVH_REQUIRE(desc && params && grads, "vae_backward_async: null descriptor / pointer table");
  MVH_REQUIRE(x && y && x_gt && recon && y_hat && mu && logvar, "vae_backward_async: null tensor");
  if (int rc = vae_step_precheck(desc, B, ws, ws_bytes)) return rc;
  const int np = mvh_vae_param_count(desc);
  auto d = std::make_shared<mvh_vae_desc_t>(*desc);
  auto P = std::make_shared<std::vector<const float*>>(params, params + np);
  auto G = std::make_shared<std::vector<float*>>(grads, grads + np);
  return submit(L, (hipStream_t)user_stream, 1, [=](hipStream_t S) {
    return mvh_vae_backward((mvh_stream_t)S, d.get(), P->data(), G->data(), x, y, x_gt, gt_f64, eps, drop_u, B, log_sigma,
                            d_loss, recon, y_hat, mu, logvar, ws, ws_bytes, nullptr);
  });
}
