// Row F: cheb_VAE.forward (cheb_VAE.py:190-251) + loss.backward() (main.py:80) as ONE native
// launch sequence: every kernel of the forward and of the analytic backward is enqueued from
// C++ with activations / gradients in a caller-provided workspace and parameter gradients
// written straight into the caller's (flat) gradient buffer -- no framework glue kernels, no
// autograd graph.  The weight-gradient kernels depend only on (layer input, layer dout), so the
// backward forks them onto a side stream and the critical path is just the dX chain; the fork /
// join uses events only, so the whole step is capturable into one hipGraph.
#include <mutex>
#include <vector>

#include "common.hpp"

namespace mvh {

struct Buf {
  size_t off;  // byte offset into the workspace
};

constexpr size_t kNoBits = ~(size_t)0;

struct StepPlan {
  int n, B, H, C, Z, F0, flat;
  std::vector<int> Nn;            // vertices per level [n+1]
  std::vector<int> f;             // filters [n+2]
  // activations (floats): conv outputs / pooled, decoder unpooled / conv outputs
  std::vector<size_t> encA, encP, decU, decC, g_encA, g_encP, g_decU, g_decC;
  size_t h, zy, d1, d2, g_h, g_zy, g_d1, g_d2, g_recon, d_mu, d_lv, d_yhat, d_heads;
  size_t scratch_main, scratch_side, scratch_side2, scratch_bytes;   // (scratch_side2: the dense lane's conv weight gradients,
  size_t scratch2_bytes;                                               //  levels of the LDS-resident kernels only: scratch2_bytes)
  std::vector<size_t> pk_enc_f, pk_enc_b, pk_dec_f, pk_dec_b;  // slab-packed conv weights (fwd / W^T)
  std::vector<size_t> dwPartEnc, dwPartDec;  // per-layer dW partial tiles (reduced together after the join)
  std::vector<size_t> dwPartBytesEnc, dwPartBytesDec;
  size_t tstack;                         // T_k x of layer 0 at the rows its pooling selects (+ dW partials)
  size_t weff_final;                     // W_eff of the final layer (split path), built with the packs
  size_t dwPartFinal, dwPartBytesFinal, s_final;   // final layer (split path): partial tiles of its connected block, S
  size_t pk_h_f, pk_h_b;                 // bf16 weight slabs of the level-0 matrix-pipe kernel (cheb_l0h.hip), or kNoBits
  std::vector<size_t> encBits, decBits;  // ReLU sign bytes of the conv outputs (kNoBits when Cout % 4 != 0)
  std::vector<size_t> txEnc, txDec;      // saved T_1..T_{K-1} stacks of the levels too big for the LDS kernels
  size_t total;
};

static size_t take(size_t& cur, size_t floats) {
  const size_t o = cur;
  cur += align_up(floats * sizeof(float), 256);
  return o;
}

static int build_plan(const mvh_vae_desc_t* d, int B, StepPlan& p) {
  MVH_REQUIRE(d && d->n_layers >= 1 && d->n_layers <= MVH_VAE_MAX_LAYERS, "vae_step: bad n_layers");
  p.n = d->n_layers; p.B = B; p.H = d->num_hidden; p.C = d->num_classes; p.Z = d->num_style;
  p.F0 = d->filters[0];
  p.Nn.assign(d->num_nodes, d->num_nodes + p.n + 1);
  p.f.assign(d->filters, d->filters + p.n + 2);
  p.flat = p.Nn[p.n] * p.f[p.n + 1];
  const int n = p.n;
  size_t cur = 0;
  auto A = [&](std::vector<size_t>& v, int i, size_t floats) { v[i] = take(cur, floats); };
  p.encA.resize(n); p.encP.resize(n); p.decU.resize(n); p.decC.resize(n);
  p.g_encA.resize(n); p.g_encP.resize(n); p.g_decU.resize(n); p.g_decC.resize(n);
  size_t scratch = 0, scratch2 = 256;   // scratch2: backward calls of the levels of <= 5119 vertices (the dense lane takes only those)
  auto upd = [&](size_t b) { if (b > scratch) scratch = b; };
  auto upd2 = [&](size_t b, int N) { if (N + 1 <= 5120 && b > scratch2) scratch2 = b; };
  for (int i = 0; i < n; ++i) {
    const size_t a = (size_t)B * p.Nn[i] * p.f[i + 1], q = (size_t)B * p.Nn[i + 1] * p.f[i + 1];
    A(p.encA, i, a); A(p.g_encA, i, a); A(p.encP, i, q); A(p.g_encP, i, q);
    upd(mvh_cheb_conv_ws_bytes(B, p.Nn[i], p.f[i], p.f[i + 1], d->K[i]));
    upd(mvh_cheb_conv_bwd_ws_bytes(B, p.Nn[i], p.f[i], p.f[i + 1], d->K[i]));
    upd2(mvh_cheb_conv_bwd_ws_bytes(B, p.Nn[i], p.f[i], p.f[i + 1], d->K[i]), p.Nn[i]);
    // decoder stage i works at level n-i-1: filters[-i-1] -> filters[-i-2]
    const int lvl = n - i - 1, cin = p.f[n + 1 - i], cout = p.f[n - i];
    const size_t u = (size_t)B * p.Nn[lvl] * cin, c = (size_t)B * p.Nn[lvl] * cout;
    A(p.decU, i, u); A(p.g_decU, i, u); A(p.decC, i, c); A(p.g_decC, i, c);
    upd(mvh_cheb_conv_ws_bytes(B, p.Nn[lvl], cin, cout, d->K[i]));
    upd(mvh_cheb_conv_bwd_ws_bytes(B, p.Nn[lvl], cin, cout, d->K[i]));
    upd2(mvh_cheb_conv_bwd_ws_bytes(B, p.Nn[lvl], cin, cout, d->K[i]), p.Nn[lvl]);
  }
  upd(mvh_cheb_conv_ws_bytes(B, p.Nn[0], p.f[1], p.f[0], d->K[n]));
  upd(mvh_cheb_conv_bwd_ws_bytes(B, p.Nn[0], p.f[1], p.f[0], d->K[n]));
  upd((size_t)B * (p.flat > p.H ? p.flat : p.H) * sizeof(float) + 256);
  upd((size_t)B * (p.C + 2 * p.Z) * sizeof(float) + 256);
  upd(mvh_vae_loss_ws_bytes(B));
  p.h = take(cur, (size_t)B * p.H); p.g_h = take(cur, (size_t)B * p.H);
  p.zy = take(cur, (size_t)B * (p.C + p.Z)); p.g_zy = take(cur, (size_t)B * (p.C + p.Z));
  p.d1 = take(cur, (size_t)B * p.H); p.g_d1 = take(cur, (size_t)B * p.H);
  p.d2 = take(cur, (size_t)B * p.flat); p.g_d2 = take(cur, (size_t)B * p.flat);
  p.g_recon = take(cur, (size_t)B * p.Nn[0] * p.F0);
  p.d_mu = take(cur, (size_t)B * p.Z); p.d_lv = take(cur, (size_t)B * p.Z); p.d_yhat = take(cur, (size_t)B * p.C);
  p.d_heads = take(cur, (size_t)B * (p.C + 2 * p.Z));
  p.pk_enc_f.resize(n); p.pk_enc_b.resize(n); p.pk_dec_f.resize(n + 1); p.pk_dec_b.resize(n + 1);
  for (int i = 0; i < n; ++i) {
    p.pk_enc_f[i] = take(cur, pack_entry_floats(p.f[i], p.f[i + 1], d->K[i], false));
    p.pk_enc_b[i] = take(cur, pack_entry_floats(p.f[i], p.f[i + 1], d->K[i], true));
    p.pk_dec_f[i] = take(cur, pack_entry_floats(p.f[n + 1 - i], p.f[n - i], d->K[i], false));
    p.pk_dec_b[i] = take(cur, pack_entry_floats(p.f[n + 1 - i], p.f[n - i], d->K[i], true));
  }
  p.pk_dec_f[n] = take(cur, pack_entry_floats(p.f[1], p.f[0], d->K[n], false));
  p.pk_dec_b[n] = take(cur, pack_entry_floats(p.f[1], p.f[0], d->K[n], true));
  p.weff_final = take(cur, (size_t)p.f[1] * p.f[0]);
  p.pk_h_f = p.pk_h_b = kNoBits;
  if (d->storage == MVH_STORAGE_BF16 && p.f[2] == 16 && p.f[1] == 16) {   // last decoder stage: f[2] -> f[1] at level 0
    p.pk_h_f = take(cur, l0h_pack_dwords(d->K[n - 1]));
    p.pk_h_b = take(cur, l0h_pack_dwords(d->K[n - 1]));
  }
  p.tstack = take(cur, tstack_ws_floats(B, p.Nn[1], d->K[0], p.f[0], p.f[1]));   // (use_tstack requires down[0].n_rows == Nn[1])
  p.dwPartEnc.resize(n); p.dwPartDec.resize(n); p.dwPartBytesEnc.resize(n); p.dwPartBytesDec.resize(n);
  for (int i = 0; i < n; ++i) {
    p.dwPartBytesEnc[i] = cheb_dw_lds_ws_bytes(B, p.Nn[i], p.f[i], p.f[i + 1], d->K[i]);
    p.dwPartEnc[i] = take(cur, p.dwPartBytesEnc[i] / sizeof(float) + 1);
    p.dwPartBytesDec[i] = cheb_dw_lds_ws_bytes(B, p.Nn[n - i - 1], p.f[n + 1 - i], p.f[n - i], d->K[i]);
    p.dwPartDec[i] = take(cur, p.dwPartBytesDec[i] / sizeof(float) + 1);
  }
  p.dwPartBytesFinal = cheb_dw_lds_ws_bytes(B, p.Nn[0], p.f[1], p.f[0], d->K[n]);
  p.dwPartFinal = take(cur, p.dwPartBytesFinal / sizeof(float) + 1);
  p.s_final = take(cur, (size_t)p.f[1] * p.f[0]);
  p.encBits.assign(n, kNoBits); p.decBits.assign(n, kNoBits);
  for (int i = 0; i < n; ++i) {  // one byte per vertex and 4 output channels
    if (p.f[i + 1] % 4 == 0) p.encBits[i] = take(cur, ((size_t)B * p.Nn[i] * (p.f[i + 1] / 4) + 3) / 4);
    if (p.f[n - i] % 4 == 0) p.decBits[i] = take(cur, ((size_t)B * p.Nn[n - i - 1] * (p.f[n - i] / 4) + 3) / 4);
  }
  // A level beyond the LDS kernels' reach (> 5119 vertices: the 20k level of BASELINE configs[3]) runs the
  // stack pipeline, whose backward would otherwise rebuild the K-1 propagates of the forward (9 x 80 us at
  // 16 channels): with 288 GB of HBM the stack (737 MB at B = 64) is simply kept.
  p.txEnc.assign(n, kNoBits); p.txDec.assign(n, kNoBits);
  for (int i = 0; i < n; ++i) {
    if (p.Nn[i] + 1 > 5120 && d->K[i] > 1) p.txEnc[i] = take(cur, (size_t)(d->K[i] - 1) * B * p.Nn[i] * p.f[i]);
    const int lvl = n - i - 1;
    if (p.Nn[lvl] + 1 > 5120 && d->K[i] > 1) p.txDec[i] = take(cur, (size_t)(d->K[i] - 1) * B * p.Nn[lvl] * p.f[n + 1 - i]);
  }
  p.scratch_bytes = align_up(scratch, 256);
  p.scratch_main = cur; cur += p.scratch_bytes;
  p.scratch_side = cur; cur += p.scratch_bytes;
  p.scratch2_bytes = align_up(scratch2, 256);
  p.scratch_side2 = cur; cur += p.scratch2_bytes;
  p.total = cur + 256;
  return MVH_OK;
}

// parameter slots, state_dict order (cheb_VAE.py:121-165)
struct ParamIdx {
  int n;
  int encW(int i) const { return 2 * i; }
  int encB(int i) const { return 2 * i + 1; }
  int decW(int i) const { return 2 * n + 2 * i; }
  int decB(int i) const { return 2 * n + 2 * i + 1; }  // i < n only
  int base() const { return 4 * n + 1; }
  int clsW() const { return base(); }       int clsB() const { return base() + 1; }
  int zmW() const { return base() + 2; }    int zmB() const { return base() + 3; }
  int zvW() const { return base() + 4; }    int zvB() const { return base() + 5; }
  int encLW() const { return base() + 6; }  int encLB() const { return base() + 7; }
  int decLW() const { return base() + 8; }  int decLB() const { return base() + 9; }
  int dl1W() const { return base() + 10; }  int dl1B() const { return base() + 11; }
  int dl2W() const { return base() + 12; }  int dl2B() const { return base() + 13; }
  int count() const { return base() + 14; }
};

struct SideStream {
  hipStream_t stream = nullptr;
  hipStream_t dense = nullptr;  // second lane: weight gradients of the dense layers + latent heads
  hipStream_t lstream = nullptr, ldense = nullptr;  // the same two lanes at LOWEST queue priority, for the launcher's jobs (lanes_for)
  hipEvent_t ev[64];
  int n_ev = 0;
  int next_ev = 0;
  hipEvent_t dense_done = nullptr;  // recorded after the dense-layer weight gradients of the last backward
  bool dense_recorded = false;
  // mvh_vae_backward_prefetch: the first layer's Chebyshev stack is already being built on the side lane.  The state is
  // kept PER WORKSPACE (= per chain: TrainStep n_micro > 1 drives several from several host threads), so that one chain's
  // prefetch call or backward never disarms another's; kPrefetchSlots chains at a time, the least recently armed one is
  // recycled (a chain that loses its slot only loses the overlap: its backward builds the stack itself).
  static constexpr int kPrefetchSlots = 4;
  struct Prefetch {
    hipEvent_t done = nullptr;
    const void* x = nullptr;
    const void* ws = nullptr;
    hipStream_t stream = nullptr;   // the lane it was launched on; null: written by the forward's first layer itself
    hipStream_t fwd = nullptr;      // ... then: the forward's stream (a backward on the same stream needs no wait)
    bool pending = false;   // launched: mvh_vae_backward only waits for `done`
    bool armed = false;     // requested: the next mvh_vae_forward on (x, ws) launches it
    unsigned long long stamp = 0;
  } pf[kPrefetchSlots];
  unsigned long long pf_clock = 0;
  Prefetch* prefetch_of(const void* ws, bool take) {
    Prefetch* lru = &pf[0];
    for (Prefetch& q : pf) {
      if (q.ws == ws) return &q;
      if (q.stamp < lru->stamp) lru = &q;
    }
    if (!take) return nullptr;
    lru->ws = ws; lru->x = nullptr; lru->stream = nullptr; lru->pending = lru->armed = false;
    return lru;
  }
};

// ONE set of gradient lanes per DEVICE, shared by every host thread, behind a per-device lock that an entry point holds for
// the whole of its launch sequence.  (They used to be per host thread.  MEASURED, round 4, tools/stream_probe.py: a forward
// enqueued by one thread and its backward by another -- exactly what torch.autograd does with the module path, whose
// backward runs on the engine's device thread -- took 1.07-1.6 ms per step instead of 0.49 as soon as the second thread's
// lanes had hardware queues of their own: beyond four busy hardware queues this GPU / driver stalls single launches for
// ~10 ms.  GPU_MAX_HW_QUEUES = 3 made the same test 0.49 ms again.  With shared lanes a process has three busy queues
// however many threads drive it.)  Threads that enqueue concurrently (TrainStep n_micro > 1) take turns.
static std::mutex g_lane_mu[16];
struct LaneLock {
  std::unique_lock<std::mutex> lk;
  LaneLock() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    lk = std::unique_lock<std::mutex>(g_lane_mu[dev]);
  }
};
// (caller holds the device's LaneLock)
static SideStream* side_for_device() {
  static SideStream side[16];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  SideStream& s = side[dev];
  if (!s.stream) {
    // Fork / join events between the caller's stream and the two gradient lanes order work on ONE device: they do not need
    // the system-scope fence hipEventRecord performs by default (cache write-back and invalidation for the host and other
    // devices; hip_runtime_api.h: "may improve performance but device memory may not be visible to the host and other
    // devices").  Whatever leaves the device is behind the caller's own stream synchronisation / events.  MEASURED:
    // 524.2 against 535.0 us per step (profiles/r03_ab_evflags.txt); -DMVH_EV_FLAGS=0 restores the default events.
#ifndef MVH_EV_FLAGS
#define MVH_EV_FLAGS hipEventDisableSystemFence
#endif
    constexpr unsigned kEvFlags = MVH_EV_FLAGS;
    // debug switch side_prio = -1 / +1: queue priority of the two weight-gradient lanes relative to the caller's stream
    int lo = 0, hi = 0, prio = 0;
    const int pe = dbg().side_prio;
    if (pe != 0 && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess) prio = pe < 0 ? lo : hi;
    if (hipStreamCreateWithPriority(&s.stream, hipStreamNonBlocking, prio) != hipSuccess) return nullptr;
    if (hipStreamCreateWithPriority(&s.dense, hipStreamNonBlocking, prio) != hipSuccess) return nullptr;
    for (int i = 0; i < 64; ++i)
      if (hipEventCreateWithFlags(&s.ev[i], hipEventDisableTiming | kEvFlags) != hipSuccess) return nullptr;
    // dense_done is the one event whose data can leave the device (mvh_vae_wait_dense_grads hands the dense gradients to an
    // RCCL stream that peer GPUs read): it keeps the default system-scope fence
    if (hipEventCreateWithFlags(&s.dense_done, hipEventDisableTiming) != hipSuccess) return nullptr;
    for (SideStream::Prefetch& q : s.pf)
      if (hipEventCreateWithFlags(&q.done, hipEventDisableTiming | kEvFlags) != hipSuccess) return nullptr;
    s.n_ev = 64;
  }
  return &s;
}

// The two weight-gradient lanes for a step whose main chain runs on `main`.  A caller on a HIGHEST-priority stream -- the
// asynchronous launcher's (csrc/launcher.hip) -- must not get the default-priority lanes: its caller leaves a blocked
// hipStreamWaitValue64 on ITS stream until the job is done, and a lane that happened to share that stream's hardware queue
// (few queues, many application streams) would sit behind the blocked wait while the job waits for the lane: a deadlock.
// The runtime pools hardware queues PER PRIORITY, so a lane is safe when its priority differs from the waiting stream's.
// Lanes of a priority other than the default one were MEASURED at 2.3-2.5 ms per step instead of 0.46, both at the
// launcher's own (highest) priority and at the lowest (round 5, profiles/r05_ref_loop_probe.txt: this
// runtime does not run streams of a non-default pool side by side with the job's).  So a launcher job's gradient work runs
// INLINE on `main`: the step's kernels serialise (0.66 ms of GPU time, profiles/r05_ref_loop_trace.txt) and the reference loop
// lands at 0.75 ms, GPU-bound.  The debug switch launcher_lanes = 1 selects the lowest-priority pair (the measurement).
// (caller holds the device's LaneLock)
static int lanes_for(SideStream* s, hipStream_t main, hipStream_t* conv, hipStream_t* dense) {
  *conv = s->stream;
  *dense = s->dense;
  int lo = 0, hi = 0, pr = 0;
  if (!main || hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess || lo == hi) return MVH_OK;
  if (hipStreamGetPriority(main, &pr) != hipSuccess || pr != hi) return MVH_OK;
  *conv = main;
  *dense = main;
  if (!dbg().launcher_lanes || lo <= 0) return MVH_OK;      // no priority level below the default one: inline
  if (!s->lstream) {
    MVH_HIP(hipStreamCreateWithPriority(&s->lstream, hipStreamNonBlocking, lo));
    MVH_HIP(hipStreamCreateWithPriority(&s->ldense, hipStreamNonBlocking, lo));
  }
  *conv = s->lstream;
  *dense = s->ldense;
  return MVH_OK;
}

int vae_step_precheck(const mvh_vae_desc_t* d, int B, const void* ws, size_t ws_bytes) {
  StepPlan p;
  if (int rc = build_plan(d, B, p)) return rc;
  MVH_REQUIRE(B > 0, "vae step: empty batch");
  MVH_REQUIRE(ws && ws_bytes >= p.total, "vae step: workspace too small (%zu < %zu)", ws_bytes, p.total);
  return MVH_OK;
}

}  // namespace mvh

using namespace mvh;

extern "C" size_t mvh_vae_step_ws_bytes(const mvh_vae_desc_t* desc, int32_t B) {
  StepPlan p;
  if (build_plan(desc, B, p)) return 0;
  return p.total;
}

extern "C" int64_t mvh_vae_ws_offset(const mvh_vae_desc_t* desc, int32_t B, const char* name, int32_t index, int64_t* n_elems) {
  StepPlan p;
  if (!desc || !name || build_plan(desc, B, p)) return -1;
  const int n = p.n;
  auto lvl_of = [&](int i) { return n - i - 1; };
  struct V { const char* nm; const std::vector<size_t>* v; int kind; };   // kind: 0 encA 1 encP 2 decU 3 decC
  const V vs[] = {{"encA", &p.encA, 0}, {"g_encA", &p.g_encA, 0}, {"encP", &p.encP, 1}, {"g_encP", &p.g_encP, 1},
                  {"decU", &p.decU, 2}, {"g_decU", &p.g_decU, 2}, {"decC", &p.decC, 3}, {"g_decC", &p.g_decC, 3}};
  for (const V& v : vs)
    if (!strcmp(name, v.nm)) {
      if (index < 0 || index >= n) return -1;
      const int i = index;
      const int64_t cnt = v.kind == 0 ? (int64_t)B * p.Nn[i] * p.f[i + 1] : v.kind == 1 ? (int64_t)B * p.Nn[i + 1] * p.f[i + 1]
                          : v.kind == 2 ? (int64_t)B * p.Nn[lvl_of(i)] * p.f[n + 1 - i] : (int64_t)B * p.Nn[lvl_of(i)] * p.f[n - i];
      if (n_elems) *n_elems = cnt;
      return (int64_t)(*v.v)[i];
    }
  struct S { const char* nm; size_t off; int64_t cnt; };
  const S ss[] = {{"h", p.h, (int64_t)B * p.H}, {"g_h", p.g_h, (int64_t)B * p.H}, {"zy", p.zy, (int64_t)B * (p.C + p.Z)},
                  {"g_zy", p.g_zy, (int64_t)B * (p.C + p.Z)}, {"d1", p.d1, (int64_t)B * p.H}, {"g_d1", p.g_d1, (int64_t)B * p.H},
                  {"d2", p.d2, (int64_t)B * p.flat}, {"g_d2", p.g_d2, (int64_t)B * p.flat},
                  {"g_recon", p.g_recon, (int64_t)B * p.Nn[0] * p.F0}};
  for (const S& e : ss)
    if (!strcmp(name, e.nm)) {
      if (n_elems) *n_elems = e.cnt;
      return (int64_t)e.off;
    }
  return -1;
}

extern "C" int32_t mvh_vae_param_count(const mvh_vae_desc_t* desc) {
  if (!desc) return 0;
  ParamIdx ix{desc->n_layers};
  return ix.count();
}

#define F(off) ((float*)((char*)ws + (off)))
#define BITS(off) ((off) == kNoBits ? (uint8_t*)nullptr : (uint8_t*)((char*)ws + (off)))
#define TX(off) ((off) == kNoBits ? (float*)nullptr : (float*)((char*)ws + (off)))
#define TRY(expr) do { if (int rc__ = (expr)) return rc__; } while (0)

// phases of the forward: the whole step, or only the encoder (x -> h) / only the decoder (zy -> recon) for the
// inference-side callers (inference.py, crecon.py call net.encoder / net.sample on their own)
enum { kPhEnc = 1, kPhHead = 2, kPhDec = 4, kPhLoss = 8, kPhAll = 15 };

static int run_armed_prefetch(SideStream* side, hipStream_t main, const mvh_vae_desc_t* d, const StepPlan& p,
                              const float* x, void* ws, int B);

static int vae_forward_impl(mvh_stream_t stream, const mvh_vae_desc_t* d, const float* const* P, const float* x,
                            const float* y, const void* x_gt, int32_t gt_f64, const float* eps,
                            const float* drop_u, int32_t B, float log_sigma, void* loss, int64_t* correct,
                            float* recon, float* kld, void* rec, float* z, float* y_hat, float* mu,
                            float* logvar, void* ws, size_t ws_bytes, int phases, float* h_out, const float* zy_in) {
  StepPlan p;
  TRY(build_plan(d, B, p));
  if (phases == kPhAll)
  MVH_REQUIRE(P && x && y && x_gt && loss && correct && recon && kld && rec && z && y_hat && mu && logvar,
              "vae_forward: null tensor");
  MVH_REQUIRE(B > 0, "vae_forward: empty batch");
  MVH_REQUIRE(ws && ws_bytes >= p.total, "vae_forward: workspace too small (%zu < %zu)", ws_bytes, p.total);
  const int n = p.n;
  ParamIdx ix{n};
  void* sm = (char*)ws + p.scratch_main;
  const float pd = drop_u ? d->dropout_p : 0.f;
  // storage == MVH_STORAGE_BF16: every conv-level activation between two layers is a bf16 tensor (the buffers of
  // the plan keep their fp32 sizes); the network input, the reconstruction and the dense head stay fp32
  const bool bf = d->storage == MVH_STORAGE_BF16;
  // dropout uniforms: [B, H | H | H | flat] per row (encoder h, classifier, dec_lin, dec_lin_2)
  const int urow = 3 * p.H + p.flat;
  (void)urow;
  const float* u_enc = drop_u;
  const float* u_cls = drop_u ? drop_u + (size_t)B * p.H : nullptr;
  const float* u_d1 = drop_u ? drop_u + (size_t)2 * B * p.H : nullptr;
  const float* u_d2 = drop_u ? drop_u + (size_t)3 * B * p.H : nullptr;

  {  // slab-packed copies of every conv weight (forward order and W^T), one launch for the step
    PackTable t;
    t.n = 0;
    auto add = [&](const float* W, size_t off, int cin, int cout, int K, bool bwd) {
      PackEntry& e = t.e[t.n++];
      e.W = W; e.dst = F(off); e.K = K; e.Cin = cin; e.Cout = cout;
      e.CQ = bwd ? cout : cin; e.CO = bwd ? cin : cout; e.bwd = bwd ? 1 : 0;
    };
    for (int i = 0; i < n; ++i) {
      add(P[ix.encW(i)], p.pk_enc_f[i], p.f[i], p.f[i + 1], d->K[i], false);
      if (i > 0) add(P[ix.encW(i)], p.pk_enc_b[i], p.f[i], p.f[i + 1], d->K[i], true);
      add(P[ix.decW(i)], p.pk_dec_f[i], p.f[n + 1 - i], p.f[n - i], d->K[i], false);
      add(P[ix.decW(i)], p.pk_dec_b[i], p.f[n + 1 - i], p.f[n - i], d->K[i], true);
    }
    add(P[ix.decW(n)], p.pk_dec_f[n], p.f[1], p.f[0], d->K[n], false);
    add(P[ix.decW(n)], p.pk_dec_b[n], p.f[1], p.f[0], d->K[n], true);
    add(P[ix.decW(n)], p.weff_final, p.f[1], p.f[0], d->K[n], false);
    t.e[t.n - 1].bwd = 2;
    if (p.pk_h_f != kNoBits) {  // bf16 slabs for the matrix-pipe kernel of the last decoder stage (forward, W^T)
      add(P[ix.decW(n - 1)], p.pk_h_f, p.f[2], p.f[1], d->K[n - 1], false);
      t.e[t.n - 1].bwd = 3;
      add(P[ix.decW(n - 1)], p.pk_h_b, p.f[2], p.f[1], d->K[n - 1], true);
      t.e[t.n - 1].bwd = 4;
    }
    MVH_RANGE("fwd pack");
    TRY(launch_pack_all((hipStream_t)stream, t));
  }
  // ---- encoder (cheb_VAE.py:261-273)
  const float* cur = x;
  for (int i = 0; i < n && (phases & kPhEnc); ++i) {
    // conv + ReLU + one-hot downsampling in one launch (the pooled rows are extra stores of the epilogue)
    MVH_RANGE("fwd enc%d N=%d %d->%d", i, p.Nn[i], p.f[i], p.f[i + 1]);
    ConvIO io;
    io.x = bf && i > 0; io.out = bf; io.pooled = bf && i + 1 < n;   // (the last pooled level feeds the fp32 dense head)
    // the un-pooled rows have no reader (the backward takes the ReLU signs from the sign bytes, and only at the pooled rows)
    // (inference plans keep no sign bytes: only the first layer's patch kernel takes the hint without them)
    io.out_dead = (p.encBits[i] != kNoBits || i == 0) && !dbg().keep_enc_out;
    // the first layer on the patch kernel leaves its T_k stack for the backward's weight gradient (no k_cheb_tstack launch)
    bool stack_done = false;
    if (i == 0 && phases == kPhAll && d->down[0].n_rows == p.Nn[1] && !dbg().no_tstack &&
        tstack_eligible(&d->lap[0], &d->down[0], p.Nn[0], p.f[0], p.f[1], d->K[0])) {
      io.stack_out = F(p.tstack);
      io.stack_done = &stack_done;
    }
    TRY(cheb_conv_fwd_impl((hipStream_t)stream, &d->lap[i], cur, P[ix.encW(i)], P[ix.encB(i)], F(p.encA[i]),
                           bf ? nullptr : TX(p.txEnc[i]), B,
                           p.Nn[i], p.f[i], p.f[i + 1], d->K[i], MVH_ACT_RELU, sm, p.scratch_bytes, F(p.pk_enc_f[i]),
                           &d->down[i], F(p.encP[i]), BITS(p.encBits[i]), nullptr, io));
    cur = F(p.encP[i]);
    // the armed first-layer stack (mvh_vae_backward_prefetch) starts behind encoder stage `prefetch_at` (debug switch,
    // default 0; MEASURED 0 .. 3 on one box: 553 .. 555 us per step, no difference)
    if (stack_done) {        // (tells mvh_vae_backward on this workspace, and disarms a requested prefetch)
      LaneLock lanes;
      SideStream* side = side_for_device();
      MVH_REQUIRE(side != nullptr, "vae_forward: could not create the side stream");
      SideStream::Prefetch* pf = side->prefetch_of(ws, true);
      pf->x = x; pf->stream = nullptr; pf->fwd = (hipStream_t)stream; pf->armed = false; pf->pending = true;
      pf->stamp = ++side->pf_clock;
      MVH_HIP(hipEventRecord(pf->done, (hipStream_t)stream));
    }
    if (i == min(max(dbg().prefetch_at, 0), n - 1) && phases == kPhAll) {
      LaneLock lanes;
      TRY(run_armed_prefetch(side_for_device(), (hipStream_t)stream, d, p, x, ws, B));
    }
  }
  if (phases & kPhEnc) {
    MVH_RANGE("fwd enc_lin");
    TRY(mvh_linear_fwd(stream, cur, P[ix.encLW()], P[ix.encLB()], h_out ? h_out : F(p.h), B, p.flat, p.H, MVH_ACT_RELU,
                       u_enc, pd));
  }
  // ---- classifier + latent heads + reparameterisation (cheb_VAE.py:203-226)
  bool d1_done = false;   // dec_lin rides in the latent-head launch when the step runs both (one launch less)
  if (phases & kPhHead) {
    MVH_RANGE("fwd latent heads");
    const bool with_dec = (phases & kPhDec) && !zy_in;
    TRY(latent_fwd_impl((hipStream_t)stream, F(p.h), y, u_cls, pd, P[ix.clsW()], P[ix.clsB()], P[ix.zmW()], P[ix.zmB()],
                        P[ix.zvW()], P[ix.zvB()], eps, y_hat, mu, logvar, z, F(p.zy), B, p.H, p.C, p.Z,
                        with_dec ? P[ix.decLW()] : nullptr, P[ix.decLB()], u_d1, F(p.d1), &d1_done));
  }
  if (!(phases & kPhDec)) return MVH_OK;
  // ---- decoder (cheb_VAE.py:275-292)
  if (!d1_done)
    TRY(mvh_linear_fwd(stream, zy_in ? zy_in : F(p.zy), P[ix.decLW()], P[ix.decLB()], F(p.d1), B, p.C + p.Z, p.H,
                       MVH_ACT_RELU, u_d1, pd));
  TRY(mvh_linear_fwd(stream, F(p.d1), P[ix.dl2W()], P[ix.dl2B()], F(p.d2), B, p.H, p.flat, MVH_ACT_RELU, u_d2, pd));
  // the first upsampling takes the dense head's output; the later ones are produced by the previous
  // stage's conv kernel (pooled rows gathered from LDS in its epilogue, no pool launch)
  TRY(check_csr(&d->up[n - 1], "up"));
  TRY(launch_spmm((hipStream_t)stream, &d->up[n - 1], F(p.d2), F(p.decU[0]), nullptr, nullptr, 1.f, 0.f, B, p.f[n + 1], true, bf));
  bool unpool_in_next = false;      // this stage's input is still coarse: its vertex-patch kernel un-pools while it loads
  bool map_done = false;            // the last stage's kernel wrote the final layer's per-vertex map (rows off its connected block)
  const bool split_final = conv_split_eligible(&d->lap[n], p.Nn[0], p.f[1], p.f[0], d->K[n]);
  for (int i = 0; i < n; ++i) {
    const int lvl = n - i - 1, cin = p.f[n + 1 - i], cout = p.f[n - i];
    const bool more = i + 1 < n;
    MVH_RANGE("fwd dec%d N=%d %d->%d", i, p.Nn[lvl], cin, cout);
    ConvIO io;
    io.x = io.out = io.pooled = bf;
    if (i == n - 1 && p.pk_h_f != kNoBits) io.wh = reinterpret_cast<const uint32_t*>(F(p.pk_h_f));
    if (i == n - 1 && !bf && split_final && p.f[1] == 16 && p.f[0] <= 4 && recon) {
      // the final conv (cheb_VAE.py:288) off its connected block is the per-vertex map x16 W_eff: out of this stage's
      // epilogue when it runs the vertex-patch kernel (W_eff comes from the step's pack launch)
      io.map_w = F(p.weff_final); io.map_out = recon; io.map_c = p.f[0]; io.map_n0 = d->lap[n].n_active;
      io.map_done = &map_done;
    }
    const float* xin = F(p.decU[i]);
    if (unpool_in_next) {            // (decided by the stage before, which stored no un-pooled rows)
      xin = F(p.decC[i - 1]);
      io.x_unpool = true;
      io.x_store = ((phases & kPhLoss) && dbg().no_patch_unpool != 2) ? F(p.decU[i]) : nullptr;   // the backward's dW operand; a decode-only call has no use for it (no_patch_unpool = 2: TIMING ONLY, never stored)
    }
    // the NEXT stage on a vertex-patch plan that carries U's rows (fp32 storage; the 5k level): no un-pooled tensor is
    // written here (20 MB as 16-byte pieces of 64-byte rows from this kernel's 4-channel slabs), the next kernel gathers
    // three coarse rows per vertex from this stage's 5 MB output instead of reading it
    unpool_in_next = more && !bf && !p.txDec.empty() && TX(p.txDec[i + 1]) == nullptr &&
                     patch_unpool_eligible(&d->lap[lvl - 1], &d->up[lvl - 1], p.Nn[lvl - 1], p.f[n - i], p.f[n - i - 1], d->K[i + 1]) &&
                     ((((uintptr_t)F(p.decC[i])) | ((uintptr_t)F(p.decU[i + 1]))) & 15) == 0;
    const bool pool_here = more && !unpool_in_next;
    TRY(cheb_conv_fwd_impl((hipStream_t)stream, &d->lap[lvl], xin, P[ix.decW(i)], P[ix.decB(i)], F(p.decC[i]),
                           bf ? nullptr : TX(p.txDec[i]), B, p.Nn[lvl], cin, cout, d->K[i], MVH_ACT_RELU, sm, p.scratch_bytes,
                           F(p.pk_dec_f[i]), pool_here ? &d->up[lvl - 1] : nullptr, pool_here ? F(p.decU[i + 1]) : nullptr,
                           BITS(p.decBits[i]), nullptr, io));
    cur = F(p.decC[i]);
  }
  // final conv on the coarsest edge list (the reference's quirk, :288), no bias, no activation.  With the loss right
  // behind it (the train step) the per-vertex map of its isolated rows rides in the first loss launch: only the connected
  // block's kernel runs here (one streaming launch less on the main chain, the 20 MB input read once instead of twice)
  const bool fuse_final = (phases & kPhLoss) && !bf && !dbg().no_final_fuse && p.f[1] % 4 == 0 &&
                          conv_split_eligible(&d->lap[n], p.Nn[0], p.f[1], p.f[0], d->K[n]);
  {
    MVH_RANGE("fwd final conv N=%d %d->%d", p.Nn[0], p.f[1], p.f[0]);
    ConvIO io;
    io.x = bf;   // (the reconstruction itself is an fp32 tensor)
    io.out_lazy = fuse_final || map_done;      // (map_done: the rows off the connected block are already in `recon`)
    TRY(cheb_conv_fwd_impl((hipStream_t)stream, &d->lap[n], cur, P[ix.decW(n)], nullptr, recon, nullptr, B, p.Nn[0], p.f[1],
                           p.f[0], d->K[n], MVH_ACT_NONE, sm, p.scratch_bytes, F(p.pk_dec_f[n]), nullptr, nullptr, nullptr,
                           F(p.weff_final), io));
  }
  if (!(phases & kPhLoss)) return MVH_OK;
  // ---- loss (cheb_VAE.py:321-346)
  // (the gradient seeds of a d_loss = 1 backward come out of the same two launches)
  MVH_RANGE("fwd loss");
  return loss_fwd_impl((hipStream_t)stream, recon, x_gt, gt_f64, mu, logvar, y, y_hat, log_sigma, loss, rec, kld,
                       correct, B, p.Nn[0] * p.F0, p.C, p.Z, sm, p.scratch_bytes, F(p.g_recon), F(p.d_mu), F(p.d_lv),
                       F(p.d_yhat), (fuse_final && !map_done) ? cur : nullptr, F(p.weff_final), recon, p.f[1], p.F0,
                       d->lap[n].n_active);
}

extern "C" int mvh_vae_forward(mvh_stream_t stream, const mvh_vae_desc_t* d, const float* const* P, const float* x,
                               const float* y, const void* x_gt, int32_t gt_f64, const float* eps,
                               const float* drop_u, int32_t B, float log_sigma, void* loss, int64_t* correct,
                               float* recon, float* kld, void* rec, float* z, float* y_hat, float* mu,
                               float* logvar, void* ws, size_t ws_bytes) {
  return vae_forward_impl(stream, d, P, x, y, x_gt, gt_f64, eps, drop_u, B, log_sigma, loss, correct, recon, kld, rec, z,
                          y_hat, mu, logvar, ws, ws_bytes, kPhAll, nullptr, nullptr);
}

extern "C" int mvh_vae_encode(mvh_stream_t stream, const mvh_vae_desc_t* d, const float* const* P, const float* x,
                              const float* drop_u_enc, int32_t B, float* h, void* ws, size_t ws_bytes) {
  MVH_REQUIRE(P && x && h, "vae_encode: null tensor");
  return vae_forward_impl(stream, d, P, x, nullptr, nullptr, 0, nullptr, drop_u_enc, B, 0.f, nullptr, nullptr, nullptr,
                          nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, ws, ws_bytes, kPhEnc, h, nullptr);
}

extern "C" int mvh_vae_decode(mvh_stream_t stream, const mvh_vae_desc_t* d, const float* const* P, const float* zy,
                              const float* drop_u, int32_t B, float* recon, void* ws, size_t ws_bytes) {
  MVH_REQUIRE(P && zy && recon, "vae_decode: null tensor");
  return vae_forward_impl(stream, d, P, nullptr, nullptr, nullptr, 0, nullptr, drop_u, B, 0.f, nullptr, nullptr, recon,
                          nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, ws, ws_bytes, kPhDec, nullptr, zy);
}

// The part of the backward that depends on the INPUT only: T_k x of the first layer at the rows its pooling keeps
// (cheb_tstack.hip).  This call arms it; the forward then launches it on the weight-gradient lane right AFTER its
// first (chip-filling) convolution, so that it runs on 64 CUs underneath the forward's latency-bound small-level
// kernels instead of inside the backward's chip-filling window.  (Launched before the first convolution it
// delayed that kernel by its own 44 us: 611 vs 591 us per step.)
extern "C" int mvh_vae_backward_prefetch(mvh_stream_t stream, const mvh_vae_desc_t* d, const float* x, int32_t B,
                                         void* ws, size_t ws_bytes, mvh_stream_t side_stream) {
  StepPlan p;
  TRY(build_plan(d, B, p));
  MVH_REQUIRE(x && ws && ws_bytes >= p.total, "vae_backward_prefetch: null tensor or workspace too small");
  LaneLock lanes;
  SideStream* side = side_for_device();
  MVH_REQUIRE(side != nullptr, "vae_backward_prefetch: could not create the side stream");
  SideStream::Prefetch* pf = side->prefetch_of(ws, true);
  pf->pending = pf->armed = false;
  const bool use_tstack = tstack_eligible(&d->lap[0], &d->down[0], p.Nn[0], p.f[0], p.f[1], d->K[0]) &&
                          d->down[0].n_rows == p.Nn[1] && !dbg().no_tstack && !dbg().no_prefetch;
  hipStream_t main = (hipStream_t)stream;
  hipStream_t lane_conv = nullptr, lane_dense = nullptr;
  TRY(lanes_for(side, main, &lane_conv, &lane_dense));
  hipStream_t sstream = side_stream ? (hipStream_t)side_stream : lane_conv;
  if (!use_tstack || dbg().no_side || sstream == main) return MVH_OK;   // nothing to run ahead: the backward does it all
  pf->x = x; pf->stream = sstream; pf->armed = true; pf->stamp = ++side->pf_clock;
  return MVH_OK;
}

// (called by the forward after its first convolution)
static int run_armed_prefetch(SideStream* side, hipStream_t main, const mvh_vae_desc_t* d, const StepPlan& p,
                              const float* x, void* ws, int B) {
  SideStream::Prefetch* pf = side ? side->prefetch_of(ws, false) : nullptr;
  if (!pf || !pf->armed || pf->x != x) return MVH_OK;
  pf->armed = false;
  hipStream_t sstream = pf->stream;
  int& ev = side->next_ev;
  MVH_HIP(hipEventRecord(side->ev[ev], main));          // (the previous step's reader of the stack is behind this point)
  MVH_HIP(hipStreamWaitEvent(sstream, side->ev[ev], 0));
  ev = (ev + 1) % side->n_ev;
  TRY(launch_tstack(sstream, &d->lap[0], &d->down[0], x, F(p.tstack), B, p.Nn[0], p.f[0], d->K[0]));
  MVH_HIP(hipEventRecord(pf->done, sstream));
  pf->pending = true;
  return MVH_OK;
}

extern "C" int mvh_vae_backward(mvh_stream_t stream, const mvh_vae_desc_t* d, const float* const* P,
                                float* const* G, const float* x, const float* y, const void* x_gt,
                                int32_t gt_f64, const float* eps, const float* drop_u, int32_t B,
                                float log_sigma, const void* d_loss, const float* recon, const float* y_hat,
                                const float* mu, const float* logvar, void* ws, size_t ws_bytes,
                                mvh_stream_t side_stream) {
  StepPlan p;
  TRY(build_plan(d, B, p));
  MVH_REQUIRE(P && G && x && y && x_gt && recon && y_hat && mu && logvar, "vae_backward: null tensor");
  MVH_REQUIRE(ws && ws_bytes >= p.total, "vae_backward: workspace too small");
  const int n = p.n;
  ParamIdx ix{n};
  hipStream_t main = (hipStream_t)stream;
  LaneLock lanes;
  SideStream* side = side_for_device();
  MVH_REQUIRE(side != nullptr, "vae_backward: could not create the side stream");
  SideStream::Prefetch* pf = side->prefetch_of(ws, false);      // this chain's prefetch state, if it has any
  hipStream_t lane_conv = nullptr, lane_dense = nullptr;
  TRY(lanes_for(side, main, &lane_conv, &lane_dense));
  hipStream_t sstream = side_stream ? (hipStream_t)side_stream : lane_conv;
  if (dbg().no_side) sstream = main;  // debugging aid: weight gradients on the main chain
  // dense-layer weight gradients: their own lane (they would delay the conv dW chain on `sstream`)
  hipStream_t dstream = (sstream == main) ? main : (side_stream ? sstream : lane_dense);
  void* sm = (char*)ws + p.scratch_main;
  void* ss = (char*)ws + p.scratch_side;
  // debug switch dw_lane2: the small levels' conv weight gradients alternate between the conv lane and the dense lane
  const bool lane2 = dbg().dw_lane2 && dstream != main && dstream != sstream && p.scratch_side2 != kNoBits;
  void* ss2 = p.scratch_side2 != kNoBits ? (void*)((char*)ws + p.scratch_side2) : ss;   // (second lane's scratch)
  int lane_toggle = 0;
  const float pd = drop_u ? d->dropout_p : 0.f;  // eval mode: no mask was applied in the forward
  const bool bf = d->storage == MVH_STORAGE_BF16;   // bf16 activations and activation gradients (see the forward)
  const float* u_cls = drop_u ? drop_u + (size_t)B * p.H : nullptr;
  int& ev = side->next_ev;  // ring shared by every chain on this device (record/wait pairs are adjacent)
  const bool use_tstack = tstack_eligible(&d->lap[0], &d->down[0], p.Nn[0], p.f[0], p.f[1], d->K[0]) &&
                          d->down[0].n_rows == p.Nn[1] && !dbg().no_tstack;
  // lazy rows between the final layer (split path, cheb_VAE.py:288) and the last decoder stage: the final layer's dX is
  // dout W_eff^T off its 20-vertex block -- 3 numbers per vertex -- so it writes ONLY the block's rows and the stage's dX /
  // dW kernels rebuild the other rows while they load them (no k_gstack_rows launch on the main chain, 64 -> 12 bytes
  // per vertex read by each of the two chip-filling kernels, 20 MB less written).  5k-class fp32 models only.
  // (bf16 storage: when the stage's backward is the vertex-patch kernel, which reads lazy rows in either storage mode)
  const bool bf_lazy_ok = bf && !dbg().no_l0h && !dbg().no_patch_bf16 && !dbg().no_patch_bwd && n >= 1 &&
                          patch_eligible(&d->lap_t[0], p.Nn[0], p.f[2], p.f[1], d->K[n - 1]);
  const bool lazy3 = (!bf || bf_lazy_ok) && !dbg().no_src3 && !dbg().force_generic && !dbg().l0_wide && !dbg().dw_tie_x && !dbg().no_side &&
                     p.Nn[0] + 1 > 2048 && p.Nn[0] + 1 <= 5120 && p.f[1] == 16 && p.f[2] == 16 && p.F0 == 3 &&
                     conv_split_eligible(&d->lap[n], p.Nn[0], p.f[1], p.f[0], d->K[n]) &&
                     conv_split_eligible(&d->lap_t[n], p.Nn[0], p.f[1], p.f[0], d->K[n]) && d->lap[n].n_active <= 512 &&
                     d->lap_t[n].n_active == d->lap[n].n_active && !(d->lap[0].flags & MVH_CSR_ELL_OVERFLOW) &&
                     p.decBits[n - 1] != kNoBits && d->K[n - 1] >= 1;
  hipEvent_t ev_tstack = nullptr;
  const bool tail_on_main = use_tstack && sstream != main && dbg().tail_main != 0;
  DwReduceTable red;        // pending dW reductions: one launch after the join
  red.n = 0;
  // fork: the weight-gradient kernels of a conv layer run on the side stream once its dout exists.
  // The launches go through a small queue so that several layers can share one fork (event record
  // on the main stream + wait on the side stream): every queued item's inputs exist when it is
  // queued, so forking later is always safe.  MEASURED: sharing forks (debug switch fork_batch = 2..4) is
  // 3 % SLOWER than one fork per layer (the default, 1) -- the ~7 us gaps a rocprofv3 timeline
  // shows at the forks are tracing overhead, while starting a layer's dW later lengthens the tail.
  struct PendingDw {
    const mvh_csr_t *lap, *lap_t;
    const float *xin, *W, *out, *dout;
    float *dW, *db;
    int N, cin, cout, K, act;
    const uint8_t* bits;
    size_t part_off, part_bytes;
    const mvh_csr_t *dout_pool, *unpool_t;  // dout is the gradient of the POOLED output (fused un-pooling);
    float* unpooled;                        // fallback: un-pool with unpool_t into this buffer first
    const float* tx;                        // T_k stack kept by the forward (big levels), else null
    ConvIO io;
    bool to_dense;                          // run on the dense lane (behind whatever is queued there)
    int hold = 0;                           // schedule override (debug switch sched): forks of the chain to let pass first
  };
  PendingDw pending[6];
  int n_pending = 0;
  hipStream_t sstream_conv = sstream;
  void* ss_conv = ss;
  const int fork_batch = dbg().fork_batch < 1 ? 1 : (dbg().fork_batch > 4 ? 4 : dbg().fork_batch);
  // The level-0 lane.  The 5k level's weight-gradient kernel is one 160 KB workgroup per CU: as ONE launch it holds every CU
  // for its 54 us and the main chain's next small-level kernel waits for it (timeline: 64 us for a 7 us kernel).  It is
  // therefore held back until the fork BEHIND the level's dX kernel (chip-filling as well: side by side the two would only
  // take turns) and then runs on the dense lane in l0_lane launches of B / l0_lane meshes, one behind the other: at most
  // 256 / l0_lane CUs are taken at a time and the latency-bound small-level chain runs on the others.
  // MEASURED (B = 64, fp32): 489-496 against 512 us per step (profiles/r03_ab_l0_lane.txt); bf16 storage 459-468 against 483;
  // three launches 530, a lane of its own instead of the dense lane 502, behind one more fork 506-511.  It pays when the ONE
  // launch would hold (nearly) EVERY CU in one round: at B = 32 the kernel takes half the CUs as it is (451 against 433 us
  // with the cut), at B = 33 / 40 / 48 the single launch leaves 124 / 96 / 64 CUs to the chain and is the better form (441 /
  // 452 / 473 against 461 / 462 / 486), at B = 56 it is a draw (495), at B = 128 / 256 the kernel runs several rounds either
  // way (854 / 1547 against 844 / 1490) -- so the lane is taken for 56 < B <= 64 only (debug switch l0_lane_any lifts the
  // bound, tests).
  const bool l0_fits = (B > 56 && B <= 64) || (dbg().l0_lane_any && B >= 16);
  const int l0_split = (!lane2 && (!bf || dbg().l0_lane_bf) && dstream != main && dstream != sstream && p.scratch_side2 != kNoBits &&
                        l0_fits && dbg().l0_lane > 1) ? dbg().l0_lane : 0;
  bool next_to_dense = false;   // lane hint for the next conv_dw_side call
  PendingDw held;
  bool have_held = false;
  int held_forks = 0;
  // one queued weight-gradient item on its lane (the conv lane, or the dense lane with that lane's scratch; split > 1: the
  // 5k level's kernel in that many part-batch launches, ConvIO::dw_split)
  auto launch_one = [&](PendingDw& w, bool on_dense, int split) -> int {
    bool deferred = false, fused = false;
    hipStream_t sstream = on_dense ? dstream : sstream_conv;      // (shadows: this item's lane)
    void* ss = on_dense ? ss2 : ss_conv;
    const size_t ss_bytes = on_dense ? p.scratch2_bytes : p.scratch_bytes;
    if (split > 1) w.io.dw_split = split;
    const bool can = w.part_off != kNoBits && red.n < (int)(sizeof(red.e) / sizeof(red.e[0]));
    const float* dout = w.dout;
    if (w.dout_pool) {
      TRY(cheb_conv_bwd_impl(sstream, w.lap, w.lap_t, w.xin, w.W, w.out, w.dout, w.tx, nullptr, w.dW, w.db, B, w.N,
                             w.cin, w.cout, w.K, w.act, ss, ss_bytes, nullptr, w.dout_pool, &fused, w.bits,
                             nullptr, can ? &red.e[red.n] : nullptr, can ? F(w.part_off) : nullptr, w.part_bytes,
                             &deferred, nullptr, nullptr, w.io));
      MVH_REQUIRE(fused || !bf, "vae_backward: bf16 storage needs the fused un-pooling of the weight-gradient kernel");
      if (!fused) {  // not eligible: explicit un-pooling on this lane, then the plain call below
        TRY(mvh_pool_bwd((mvh_stream_t)sstream, w.unpool_t, w.dout, w.unpooled, B, w.cout));
        dout = w.unpooled;
      }
    }
    if (!fused)
      TRY(cheb_conv_bwd_impl(sstream, w.lap, w.lap_t, w.xin, w.W, w.out, dout, w.tx, nullptr, w.dW, w.db, B, w.N,
                             w.cin, w.cout, w.K, w.act, ss, ss_bytes, nullptr, nullptr, nullptr, w.bits, nullptr,
                             can ? &red.e[red.n] : nullptr, can ? F(w.part_off) : nullptr, w.part_bytes, &deferred,
                             nullptr, nullptr, w.io));
    if (deferred) ++red.n;
    return MVH_OK;
  };
  // ---- schedule override (debug switch sched = 1; A/B tooling, tools/sched_search.py): every conv weight-gradient item k
  // (call order: final layer, decoder stages last to first, encoder stages last to first) takes its lane from bit k of
  // sched_lane (1 = dense lane) and lets sched_hold's k-th base-4 digit forks of the chain pass before it is launched;
  // items that come due at the same fork share it.  Results are those of the default schedule (launch order only).
  const bool sched_on = dbg().sched != 0 && sstream != main && dstream != main && dstream != sstream && p.scratch_side2 != kNoBits;
  PendingDw heldv[12];
  int n_heldv = 0, sched_item = 0;
  auto sched_tick = [&](bool also_dense, bool final) -> int {
    int due[12], n_due = 0;
    bool any_conv = false, any_dense = also_dense;
    for (int h = 0; h < n_heldv; ++h) {
      if (final || heldv[h].hold <= 0) {
        due[n_due++] = h;
        const bool l0 = l0_fits && heldv[h].N + 1 > 2048 && heldv[h].N + 1 <= 5120 && heldv[h].cin == 16 && heldv[h].cout == 16;
        const bool dn = heldv[h].to_dense && (heldv[h].N + 1 <= 5120) && (l0 || heldv[h].N + 1 <= 2048 || heldv[h].cout > 4);
        heldv[h].to_dense = dn;
        (dn ? any_dense : any_conv) = true;
      } else {
        --heldv[h].hold;
      }
    }
    if (n_due == 0 && !also_dense) return MVH_OK;
    MVH_HIP(hipEventRecord(side->ev[ev], main));
    if (any_conv) MVH_HIP(hipStreamWaitEvent(sstream, side->ev[ev], 0));
    if (any_dense) MVH_HIP(hipStreamWaitEvent(dstream, side->ev[ev], 0));
    ev = (ev + 1) % side->n_ev;
    for (int d2 = 0; d2 < n_due; ++d2) {
      PendingDw& w = heldv[due[d2]];
      const bool l0 = l0_fits && w.N + 1 > 2048 && w.N + 1 <= 5120 && w.cin == 16 && w.cout == 16 && !w.dout_pool;
      TRY(launch_one(w, w.to_dense, (l0 && w.to_dense && dbg().l0_lane > 1) ? dbg().l0_lane : 0));
    }
    int keep = 0;
    for (int h = 0; h < n_heldv; ++h) {
      bool launched = false;
      for (int d2 = 0; d2 < n_due; ++d2) launched = launched || due[d2] == h;
      if (!launched) heldv[keep++] = heldv[h];
    }
    n_heldv = keep;
    return MVH_OK;
  };
  auto flush_dw = [&](bool also_dense, bool force_held = false) -> int {  // one event for everything queued (+ the dense lane)
    if (sched_on) return (also_dense || force_held) ? sched_tick(also_dense, force_held) : MVH_OK;
    if (n_pending == 0 && !also_dense && !have_held) return MVH_OK;
    const bool launch_held = have_held && (also_dense || force_held || n_pending == 0 || --held_forks <= 0);
    if (sstream != main) {
      MVH_HIP(hipEventRecord(side->ev[ev], main));
      if (n_pending > 0) MVH_HIP(hipStreamWaitEvent(sstream, side->ev[ev], 0));
      if ((also_dense && (dstream != sstream || n_pending == 0)) || (lane2 && n_pending > 0))
        MVH_HIP(hipStreamWaitEvent(dstream, side->ev[ev], 0));
      bool any_dense = launch_held;
      for (int q = 0; q < n_pending; ++q) any_dense = any_dense || pending[q].to_dense;
      if (any_dense && !also_dense) MVH_HIP(hipStreamWaitEvent(dstream, side->ev[ev], 0));
      ev = (ev + 1) % side->n_ev;
    }
    if (launch_held) { pending[n_pending++] = held; have_held = false; }   // (pending has room: a flush comes at two items at the latest)
    for (int q = 0; q < n_pending; ++q) {
      PendingDw& w = pending[q];
      const bool is_l0 = l0_split && w.N + 1 > 2048 && w.N + 1 <= 5120 && w.cin == 16 && w.cout == 16 && !w.dout_pool;
      if (is_l0 && !(launch_held && q == n_pending - 1)) {   // not yet: behind the next fork
        held = w;
        have_held = true;
        held_forks = dbg().l0_hold < 1 ? 1 : dbg().l0_hold;
        continue;
      }
      const bool lane2_item = lane2 && w.N + 1 <= 2048 && (lane_toggle++ & 1);
      TRY(launch_one(w, is_l0 || lane2_item || (w.to_dense && w.N + 1 <= 5120), is_l0 ? l0_split : 0));
    }
    n_pending = 0;
    return MVH_OK;
  };
  auto conv_dw_side = [&](const mvh_csr_t* lap, const mvh_csr_t* lap_t, const float* xin, const float* W,
                          const float* out, const float* dout, float* dW, float* db, int N, int cin, int cout,
                          int K, int act, const uint8_t* bits, const ConvIO& io, size_t part_off = kNoBits,
                          size_t part_bytes = 0, const mvh_csr_t* dout_pool = nullptr, const mvh_csr_t* unpool_t = nullptr,
                          float* unpooled = nullptr, const float* tx = nullptr) -> int {
    // TIMING ONLY (results invalid): 1 = the main chain without the weight-gradient lane; 2 = without the levels of <= 400
    // vertices; 3 = without the 5k level's; 4 = without the levels of 401 .. 2047 vertices
    {
      const int sk = dbg().skip_conv_dw;
      if (sk == 1 || (sk == 2 && N <= 400) || (sk == 3 && N > 2047) || (sk == 4 && N > 400 && N <= 2047)) return MVH_OK;
    }
    if (sched_on) {
      const int k = sched_item++;
      PendingDw w{lap, lap_t, xin, W, out, dout, dW, db, N, cin, cout, K, act, bits, part_off, part_bytes,
                  dout_pool, unpool_t, unpooled, tx, io, ((dbg().sched_lane >> k) & 1) != 0};
      w.hold = (dbg().sched_hold >> (2 * k)) & 3;
      MVH_REQUIRE(n_heldv < 12, "vae_backward: schedule override queue full");
      heldv[n_heldv++] = w;
      next_to_dense = false;
      return sched_tick(false, false);
    }
    pending[n_pending++] = PendingDw{lap, lap_t, xin, W, out, dout, dW, db, N, cin, cout, K, act, bits, part_off, part_bytes,
                                     dout_pool, unpool_t, unpooled, tx, io,
                                     next_to_dense && dstream != main && dstream != sstream && p.scratch_side2 != kNoBits};
    next_to_dense = false;
    // A fork is an event record between two kernels of the critical chain: 3.5 us of that chain.  The weight
    // gradients of the coarse levels (<= fork_small vertices, default 400: two 17 us kernels at 79 / 313 vertices) are short
    // enough to wait for the next layer's fork: two such layers share one.  MEASURED: 515-517 against 521-526 us per step;
    // with the 1250-vertex level in the scheme (fork_small = 1300) 526-531, with three layers per fork 541.
    // The final layer's three short launches wait for the last decoder stage's fork: 512.6-513.4 against 515.8-518.4 us.
    // (Letting the decoder's coarse layers wait for the dense head's fork behind them as well: 540-545 -- the lane IS the
    //  tail of the step.)
    if (fork_batch == 1 && N <= dbg().fork_small && n_pending < 2) return MVH_OK;
    if (fork_batch == 1 && dbg().fork_small > 0 && act == MVH_ACT_NONE && cout <= 4 && n_pending < 2) return MVH_OK;
    if (n_pending >= fork_batch) return flush_dw(false);
    return MVH_OK;
  };
  auto conv_dx_main = [&](const mvh_csr_t* lap, const mvh_csr_t* lap_t, const float* xin, const float* W,
                          const float* out, const float* dout, float* dx, int N, int cin, int cout, int K,
                          int act, size_t pk, const uint8_t* bits, const ConvIO& io, const float* weff = nullptr,
                          const mvh_csr_t* pool_t = nullptr, float* pooled = nullptr) -> int {
    return cheb_conv_bwd_impl(main, lap, lap_t, xin, W, out, dout, nullptr, dx, nullptr, nullptr, B, N, cin, cout, K,
                              act, sm, p.scratch_bytes, F(pk), nullptr, nullptr, bits, weff, nullptr, nullptr, 0,
                              nullptr, pool_t, pooled, io);
  };

  // ---- loss: mvh_vae_forward already left the d_loss = 1 seeds in the workspace
  if (d_loss)
    TRY(mvh_vae_loss_bwd(stream, recon, x_gt, gt_f64, mu, logvar, y, y_hat, log_sigma, d_loss, F(p.g_recon), F(p.d_mu),
                         F(p.d_lv), F(p.d_yhat), B, p.Nn[0] * p.F0, p.C, p.Z));
  // ---- final conv
  {
    MVH_RANGE("bwd final conv");
    const float* xin = F(p.decC[n - 1]);
    ConvIO io;
    io.x = bf; io.dx = bf;   // (g_recon is fp32)
    io.s_keep = F(p.s_final);
    io.dx_lazy = lazy3;
    // (MEASURED, not kept: queueing this weight gradient BEHIND the last decoder stage's, so that the chip-filling
    //  level-0 dW starts as soon as this layer's dX has produced its dout -- 587 vs 583 us per fp32 step, 534 vs 523
    //  in bf16: started that early it takes the CUs from the level-0 dX kernel of the main chain)
    TRY(conv_dw_side(&d->lap[n], &d->lap_t[n], xin, P[ix.decW(n)], nullptr, F(p.g_recon), G[ix.decW(n)], nullptr,
                     p.Nn[0], p.f[1], p.f[0], d->K[n], MVH_ACT_NONE, nullptr, io, p.dwPartFinal, p.dwPartBytesFinal));
    TRY(conv_dx_main(&d->lap[n], &d->lap_t[n], xin, P[ix.decW(n)], nullptr, F(p.g_recon), F(p.g_decC[n - 1]), p.Nn[0],
                     p.f[1], p.f[0], d->K[n], MVH_ACT_NONE, p.pk_dec_b[n], nullptr, io, F(p.weff_final)));
  }
  // ---- decoder stages, last to first
  for (int i = n - 1; i >= 0; --i) {
    const int lvl = n - i - 1, cin = p.f[n + 1 - i], cout = p.f[n - i];
    MVH_RANGE("bwd dec%d N=%d %d->%d", i, p.Nn[lvl], cin, cout);
    ConvIO io;
    io.x = io.dout = io.dx = bf;
    io.dx_pooled = bf && i > 0;   // (stage 0 hands its pooled gradient to the fp32 dense head)
    if (i == n - 1 && p.pk_h_b != kNoBits) io.wh = reinterpret_cast<const uint32_t*>(F(p.pk_h_b));
    if (i == n - 1 && lazy3) { io.src3_g = F(p.g_recon); io.src3_w = F(p.weff_final); io.src3_n = d->lap[n].n_active; io.src3_c = p.F0; }
    // dX and the upsampling backward (U^T) in one launch: the pooled gradient goes straight to the previous stage
    float* dst = (i > 0) ? F(p.g_decC[i - 1]) : F(p.g_d2);
    auto dx_this = [&]() -> int {
      return conv_dx_main(&d->lap[lvl], &d->lap_t[lvl], F(p.decU[i]), P[ix.decW(i)], F(p.decC[i]), F(p.g_decC[i]),
                          F(p.g_decU[i]), p.Nn[lvl], cin, cout, d->K[i], MVH_ACT_RELU, p.pk_dec_b[i], BITS(p.decBits[i]), io,
                          nullptr, &d->up_t[lvl], dst);
    };
    // Levels of the streaming kernels (> 5119 vertices, BASELINE configs[3]): this layer's dX and dW are both HBM
    // streams over ~1 GB; side by side they took 783 + 1044 us (218 + ~500 alone), so the dX goes first and the dW is
    // forked behind it, under the LDS-resident kernels of the coarser levels.
    // (At the 5k level the same order was MEASURED and not kept: 576 us per step against 573, and 594 with the dW
    //  launch cut into two 128-workgroup halves -- the weight-gradient lane then finishes last.)
    const bool dx_first = p.Nn[lvl] + 1 > 5120 && !dbg().no_dx_first;
    // (the first layer's stack: when the forward left it -- the streaming level's selected-rows form always does -- there is
    //  nothing to launch here; otherwise its launch belongs to the side-lane form below)
    const bool stack_ready = pf && pf->pending && pf->x == x && (pf->stream == sstream || !pf->stream);
    if (dx_first && cin == 16 && cout == 16 && !bf && !dbg().no_bwd_fused && !dbg().no_dx_tstack && !dbg().no_big &&
        !(i == n - 1 && use_tstack && !stack_ready)) {
      // ... and for 16 -> 16 channels both gradients come out of ONE pass over the T_k(dpre) stack (k_big_bwd16,
      // cheb_conv.hip): a single call on the main stream, nothing of this stage on the weight-gradient lane
      TRY(cheb_conv_bwd_impl(main, &d->lap[lvl], &d->lap_t[lvl], F(p.decU[i]), P[ix.decW(i)], F(p.decC[i]), F(p.g_decC[i]),
                             nullptr, F(p.g_decU[i]), G[ix.decW(i)], G[ix.decB(i)], B, p.Nn[lvl], cin, cout, d->K[i],
                             MVH_ACT_RELU, sm, p.scratch_bytes, F(p.pk_dec_b[i]), nullptr, nullptr, BITS(p.decBits[i]),
                             nullptr, nullptr, nullptr, 0, nullptr, &d->up_t[lvl], dst, io));
      if (i == n - 1 && use_tstack) {          // (stack_ready holds: see the condition)
        ev_tstack = (!pf->stream && pf->fwd == main) ? nullptr : pf->done;
        pf->pending = false;
      }
      continue;
    }
    // A level with a vertex-patch plan (cheb_patch.hip; the 5k level's 16 -> 16 stage): dX, its U^T pooling and the dW / db
    // partial tiles come out of ONE launch on the main chain -- nothing of this stage on the weight-gradient lanes (the
    // slab form: a 43 us dX launch on the chain + two 71 us half-batch dW launches that hold 128 CUs each on the dense lane)
    // (bf16 storage: the same kernel with bf16 loads / stores -- instead of k_cheb_l0h's dX on the chain + k_cheb_dw_l0h twice on a
    //  lane; the debug switch no_l0h keeps the general kernels there)
    if ((!bf || (!dbg().no_l0h && !dbg().no_patch_bf16)) && !dx_first && !dbg().no_patch_bwd && BITS(p.decBits[i]) && red.n < (int)(sizeof(red.e) / sizeof(red.e[0])) &&
        patch_eligible(&d->lap_t[lvl], p.Nn[lvl], cin, cout, d->K[i]) &&
        p.dwPartBytesDec[i] >= patch_part_bytes(&d->lap_t[lvl], B, d->K[i])) {
      if (dbg().patch_flush_first) TRY(flush_dw(false));   // (queued items of the final layer start beside this launch, not behind it)
      bool dfr = false;
      TRY(cheb_conv_bwd_impl(main, &d->lap[lvl], &d->lap_t[lvl], F(p.decU[i]), P[ix.decW(i)], F(p.decC[i]), F(p.g_decC[i]),
                             nullptr, F(p.g_decU[i]), G[ix.decW(i)], G[ix.decB(i)], B, p.Nn[lvl], cin, cout, d->K[i],
                             MVH_ACT_RELU, sm, p.scratch_bytes, F(p.pk_dec_b[i]), nullptr, nullptr, BITS(p.decBits[i]),
                             nullptr, &red.e[red.n], F(p.dwPartDec[i]), p.dwPartBytesDec[i], &dfr, &d->up_t[lvl], dst, io));
      MVH_REQUIRE(dfr, "vae_backward: the patch kernel did not defer its weight-gradient reduction");
      ++red.n;
      if (i == n - 1 && use_tstack) {
        if (pf && pf->pending && pf->x == x && (pf->stream == sstream || !pf->stream)) {
          ev_tstack = (!pf->stream && pf->fwd == main) ? nullptr : pf->done;   // built ahead (prefetch), or by the forward itself
        } else {
          TRY(flush_dw(false));
          MVH_HIP(hipEventRecord(side->ev[ev], main));
          MVH_HIP(hipStreamWaitEvent(sstream, side->ev[ev], 0));
          ev = (ev + 1) % side->n_ev;
          TRY(launch_tstack(sstream, &d->lap[0], &d->down[0], x, F(p.tstack), B, p.Nn[0], p.f[0], d->K[0]));
          if (sstream != main) {
            MVH_HIP(hipEventRecord(side->ev[ev], sstream));
            ev_tstack = side->ev[ev];
            ev = (ev + 1) % side->n_ev;
          }
        }
        if (pf) pf->pending = false;
      }
      continue;
    }
    if (dx_first) TRY(dx_this());
    TRY(conv_dw_side(&d->lap[lvl], &d->lap_t[lvl], F(p.decU[i]), P[ix.decW(i)], F(p.decC[i]), F(p.g_decC[i]),
                     G[ix.decW(i)], G[ix.decB(i)], p.Nn[lvl], cin, cout, d->K[i], MVH_ACT_RELU, BITS(p.decBits[i]), io,
                     p.dwPartDec[i], p.dwPartBytesDec[i], nullptr, nullptr, nullptr, bf ? nullptr : TX(p.txDec[i])));
    if (i == n - 1 && use_tstack) {
      if (pf && pf->pending && pf->x == x && (pf->stream == sstream || !pf->stream)) {
        ev_tstack = (!pf->stream && pf->fwd == main) ? nullptr : pf->done;   // built ahead (prefetch), or by the forward itself
      } else {
        // T_k x of encoder layer 0 at its pooled rows: 64 workgroups on the side lane behind the (chip-filling)
        // dW above, i.e. while the main chain runs its small-level kernels; consumed at the very end
        TRY(launch_tstack(sstream, &d->lap[0], &d->down[0], x, F(p.tstack), B, p.Nn[0], p.f[0], d->K[0]));
        if (sstream != main) {
          MVH_HIP(hipEventRecord(side->ev[ev], sstream));
          ev_tstack = side->ev[ev];
          ev = (ev + 1) % side->n_ev;
        }
      }
      if (pf) pf->pending = false;
    }
    if (!dx_first) TRY(dx_this());
  }
  // ---- dense decoder head, latent heads, dense encoder head: the dX chain stays on the main stream,
  //      every weight gradient (4 GEMMs + the head gradients) goes to the dense lane after ONE fork
  TRY(mvh_linear_bwd(stream, F(p.d1), P[ix.dl2W()], F(p.d2), F(p.g_d2), F(p.g_d1), nullptr, nullptr, B, p.H, p.flat,
                     MVH_ACT_RELU, pd, sm, p.scratch_bytes));
  {  // dec_lin's dX (g_d1 -> g_zy) inside the latent-head launch when eligible, else as its own GEMM first
    bool probe = false;
    const bool try_fuse = p.C + p.Z <= 32 && !dbg().force_generic && !dbg().no_head_fuse;
    if (!try_fuse)
      TRY(mvh_linear_bwd(stream, F(p.zy), P[ix.decLW()], F(p.d1), F(p.g_d1), F(p.g_zy), nullptr, nullptr, B, p.C + p.Z,
                         p.H, MVH_ACT_RELU, pd, sm, p.scratch_bytes));
    TRY(latent_bwd_heads(main, u_cls, pd, P[ix.clsW()], P[ix.zmW()], P[ix.zvW()], eps, y_hat, logvar, F(p.d_yhat),
                         F(p.d_mu), F(p.d_lv), F(p.g_zy), F(p.g_h), F(p.d_heads), B, p.H, p.C, p.Z,
                         try_fuse ? P[ix.decLW()] : nullptr, F(p.d1), F(p.g_d1), F(p.g_zy), &probe));
    MVH_REQUIRE(probe == try_fuse, "vae_backward: latent head fusion mismatch");
  }
  TRY(flush_dw(true));  // one fork: queued conv dW -> side lane, dense weight gradients -> dense lane
  TRY(mvh_linear_bwd(stream, F(p.encP[n - 1]), P[ix.encLW()], F(p.h), F(p.g_h), F(p.g_encP[n - 1]), nullptr, nullptr,
                     B, p.flat, p.H, MVH_ACT_RELU, pd, sm, p.scratch_bytes));
  {
    mvh_stream_t ds = (mvh_stream_t)dstream;
    TRY(mvh_linear_bwd(ds, F(p.d1), P[ix.dl2W()], F(p.d2), F(p.g_d2), nullptr, G[ix.dl2W()], G[ix.dl2B()], B, p.H,
                       p.flat, MVH_ACT_RELU, pd, nullptr, 0));
    TRY(mvh_linear_bwd(ds, F(p.zy), P[ix.decLW()], F(p.d1), F(p.g_d1), nullptr, G[ix.decLW()], G[ix.decLB()], B,
                       p.C + p.Z, p.H, MVH_ACT_RELU, pd, nullptr, 0));
    TRY(latent_bwd_wgrad(dstream, F(p.h), y, u_cls, pd, F(p.d_heads), G[ix.clsW()], G[ix.clsB()], G[ix.zmW()],
                         G[ix.zmB()], G[ix.zvW()], G[ix.zvB()], B, p.H, p.C, p.Z));
    TRY(mvh_linear_bwd(ds, F(p.encP[n - 1]), P[ix.encLW()], F(p.h), F(p.g_h), nullptr, G[ix.encLW()], G[ix.encLB()], B,
                       p.flat, p.H, MVH_ACT_RELU, pd, nullptr, 0));
    // dec_lin_1 is never used by the forward (cheb_VAE.py:165): it HAS no gradient (torch leaves .grad None and
    // Adam skips it), so G[dl1W] / G[dl1B] are not written -- the engine's flat buffer keeps its zeros there
    // every dense-layer gradient (98 % of the parameter bytes) is final from here on: a data-parallel caller
    // starts their all-reduce now, under the encoder half of the backward (mvh_vae_wait_dense_grads).
    // Not inside a stream capture: the event would become part of the graph.
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(main, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone) {
      MVH_HIP(hipEventRecord(side->dense_done, dstream));
      side->dense_recorded = true;
    }
  }
  // ---- encoder stages, last to first.  g_encP[i] is the gradient of the POOLED conv output; the
  // one-hot un-pooling is folded into the loads of the dW / dX kernels (no scatter launch, no
  // zero-filled [B, N_i, C] gradient tensor); if a layer is not eligible it is un-pooled explicitly.
  for (int i = n - 1; i >= 0; --i) {
    const float* xin = (i > 0) ? F(p.encP[i - 1]) : x;
    MVH_RANGE("bwd enc%d N=%d %d->%d", i, p.Nn[i], p.f[i], p.f[i + 1]);
    ConvIO io;
    io.x = bf && i > 0; io.dout = bf && i + 1 < n; io.dx = bf;   // (the last level's gradient comes from the fp32 dense head)
    if (i > 0) {
      // weight gradient: queued for the side lane (fused un-pooling, explicit un-pooling there if not eligible);
      // debug switch tail_main = 1 puts layer 1's dW on the main stream behind layer 0's instead (round 1's
      // arrangement, when the side lane still had a backlog at this point)
      // the coarsest encoder stage's weight gradient goes to the dense lane (behind the dense layers' there): with the
      // level-0 lane the conv lane is the one that finishes last.  MEASURED: 486 against 494 us per step (stage n - 2: 494)
      next_to_dense = !lane2 && i == (dbg().enc_dense == -2 ? (l0_split ? n - 1 : -1) : dbg().enc_dense);
      // (configs[3], MEASURED and not kept: the 5 041-vertex encoder stage's chip-filling weight gradient on the dense lane instead
      //  of the end of the conv lane's queue, in one or two launches: 2338-2348 against 2308-2311 us -- it then takes the CUs of
      //  that level's dX kernel on the main chain)
      if (i == 1 && tail_on_main) next_to_dense = false;   // (no call below to consume the hint: it must not leak to a later layer)
      else
      TRY(conv_dw_side(&d->lap[i], &d->lap_t[i], xin, P[ix.encW(i)], F(p.encA[i]), F(p.g_encP[i]), G[ix.encW(i)],
                       G[ix.encB(i)], p.Nn[i], p.f[i], p.f[i + 1], d->K[i], MVH_ACT_RELU, BITS(p.encBits[i]), io,
                       p.dwPartEnc[i], p.dwPartBytesEnc[i], &d->down[i], &d->down_t[i], F(p.g_encA[i]),
                       bf ? nullptr : TX(p.txEnc[i])));
      bool ok_dx = false;
      TRY(cheb_conv_bwd_impl(main, &d->lap[i], &d->lap_t[i], xin, P[ix.encW(i)], F(p.encA[i]), F(p.g_encP[i]), nullptr,
                             F(p.g_encP[i - 1]), nullptr, nullptr, B, p.Nn[i], p.f[i], p.f[i + 1], d->K[i], MVH_ACT_RELU,
                             sm, p.scratch_bytes, F(p.pk_enc_b[i]), &d->down[i], &ok_dx, BITS(p.encBits[i]), nullptr, nullptr,
                             nullptr, 0, nullptr, nullptr, nullptr, io));
      MVH_REQUIRE(ok_dx || !bf, "vae_backward: bf16 storage needs the fused un-pooling of the dX kernel");
      if (!ok_dx) {  // (its own un-pooled copy: the side lane may be writing g_encA for the dW fallback)
        // (decoder buffer of the same level, free by now, when it is wide enough)
        float* tmp = (p.f[i + 2] >= p.f[i + 1]) ? F(p.g_decU[n - 1 - i]) : F(p.g_encA[i]);
        TRY(mvh_pool_bwd(stream, &d->down_t[i], F(p.g_encP[i]), tmp, B, p.f[i + 1]));
        TRY(conv_dx_main(&d->lap[i], &d->lap_t[i], xin, P[ix.encW(i)], F(p.encA[i]), tmp, F(p.g_encP[i - 1]), p.Nn[i],
                         p.f[i], p.f[i + 1], d->K[i], MVH_ACT_RELU, p.pk_enc_b[i], BITS(p.encBits[i]), io));
      }
      continue;
    }
    // layer 0 has no dX: the main stream has nothing left to do, so its dW runs there (no fork
    // latency, no queueing behind the side lane).  Everything still queued is forked FIRST, or it
    // would wait behind this kernel.
    TRY(flush_dw(false));
    if (use_tstack) {  // dW_0 = stack^T dpre over the pooled rows only: a 5 us streaming reduction
      if (ev_tstack) MVH_HIP(hipStreamWaitEvent(main, ev_tstack, 0));
      TRY(launch_stack_dw(main, &d->down[0], F(p.tstack), F(p.g_encP[0]), BITS(p.encBits[0]), F(p.encA[0]),
                          G[ix.encW(0)], G[ix.encB(0)], F(p.tstack) + tstack_stack_floats(B, p.Nn[1], d->K[0]), B, p.Nn[0],
                          p.f[0], p.f[1], d->K[0], &red.e[red.n], io.dout));
      ++red.n;
      if (tail_on_main && n > 1) {
        bool fused = false, dfr = false;
        const float* xin1 = F(p.encP[0]);
        ConvIO io1;
        io1.x = bf; io1.dout = bf && n > 2;
        TRY(cheb_conv_bwd_impl(main, &d->lap[1], &d->lap_t[1], xin1, P[ix.encW(1)], F(p.encA[1]), F(p.g_encP[1]), nullptr,
                               nullptr, G[ix.encW(1)], G[ix.encB(1)], B, p.Nn[1], p.f[1], p.f[2], d->K[1], MVH_ACT_RELU, sm,
                               p.scratch_bytes, nullptr, &d->down[1], &fused, BITS(p.encBits[1]), nullptr, &red.e[red.n],
                               F(p.dwPartEnc[1]), p.dwPartBytesEnc[1], &dfr, nullptr, nullptr, io1));
        MVH_REQUIRE(fused || !bf, "vae_backward: bf16 storage needs the fused un-pooling of the weight-gradient kernel");
        if (!fused) {
          TRY(mvh_pool_bwd(stream, &d->down_t[1], F(p.g_encP[1]), F(p.g_encA[1]), B, p.f[2]));
          TRY(cheb_conv_bwd_impl(main, &d->lap[1], &d->lap_t[1], xin1, P[ix.encW(1)], F(p.encA[1]), F(p.g_encA[1]),
                                 nullptr, nullptr, G[ix.encW(1)], G[ix.encB(1)], B, p.Nn[1], p.f[1], p.f[2], d->K[1],
                                 MVH_ACT_RELU, sm, p.scratch_bytes, nullptr, nullptr, nullptr, BITS(p.encBits[1]), nullptr,
                                 &red.e[red.n], F(p.dwPartEnc[1]), p.dwPartBytesEnc[1], &dfr));
        }
        if (dfr) ++red.n;
      }
      continue;
    }
    bool ok_dw = false, deferred = false;
    TRY(cheb_conv_bwd_impl(main, &d->lap[0], &d->lap_t[0], xin, P[ix.encW(0)], F(p.encA[0]), F(p.g_encP[0]),
                           bf ? nullptr : TX(p.txEnc[0]),
                           nullptr, G[ix.encW(0)], G[ix.encB(0)], B, p.Nn[0], p.f[0], p.f[1], d->K[0], MVH_ACT_RELU, sm,
                           p.scratch_bytes, nullptr, &d->down[0], &ok_dw, BITS(p.encBits[0]), nullptr, &red.e[red.n],
                           F(p.dwPartEnc[0]), p.dwPartBytesEnc[0], &deferred, nullptr, nullptr, io));
    MVH_REQUIRE(ok_dw || !bf, "vae_backward: bf16 storage needs the fused un-pooling of the first layer's weight gradient");
    if (!ok_dw) {
      TRY(mvh_pool_bwd(stream, &d->down_t[0], F(p.g_encP[0]), F(p.g_encA[0]), B, p.f[1]));
      TRY(cheb_conv_bwd_impl(main, &d->lap[0], &d->lap_t[0], xin, P[ix.encW(0)], F(p.encA[0]), F(p.g_encA[0]), TX(p.txEnc[0]),
                             nullptr, G[ix.encW(0)], G[ix.encB(0)], B, p.Nn[0], p.f[0], p.f[1], d->K[0], MVH_ACT_RELU, sm,
                             p.scratch_bytes, nullptr, nullptr, nullptr, BITS(p.encBits[0]), nullptr, &red.e[red.n],
                             F(p.dwPartEnc[0]), p.dwPartBytesEnc[0], &deferred));
    }
    if (deferred) ++red.n;
  }
  TRY(flush_dw(false, true));   // the last fork also takes a level-0 item still held back (debug switch l0_hold >= 2)
  MVH_REQUIRE(!have_held && n_pending == 0 && n_heldv == 0, "vae_backward: a weight-gradient launch was still queued at the join");
  // join
  if (dstream != main && dstream != sstream) {
    MVH_HIP(hipEventRecord(side->ev[ev], dstream));
    MVH_HIP(hipStreamWaitEvent(main, side->ev[ev], 0));
    ev = (ev + 1) % side->n_ev;
  }
  if (sstream != main) {
    MVH_HIP(hipEventRecord(side->ev[ev], sstream));
    MVH_HIP(hipStreamWaitEvent(main, side->ev[ev], 0));
    ev = (ev + 1) % side->n_ev;
  }
  MVH_RANGE("bwd reduce weight-gradient partials");
  TRY(launch_dw_reduce_all(main, red));  // every deferred dW / db in one launch
  return MVH_OK;
}

extern "C" int mvh_vae_wait_dense_grads(mvh_stream_t stream) {
  LaneLock lanes;
  SideStream* side = side_for_device();
  MVH_REQUIRE(side != nullptr, "vae_wait_dense_grads: no device");
  MVH_REQUIRE(side->dense_recorded, "vae_wait_dense_grads: no mvh_vae_backward was issued on this device");
  MVH_HIP(hipStreamWaitEvent((hipStream_t)stream, side->dense_done, 0));
  return MVH_OK;
}

extern "C" size_t mvh_sizeof_vae_desc(void) { return sizeof(mvh_vae_desc_t); }
extern "C" size_t mvh_sizeof_csr(void) { return sizeof(mvh_csr_t); }
