// Shared host-side helpers for libmeshvae_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "meshvae_hip.h"

// In-kernel stamps of the diagnostic build (make STAMP=1): every wave's lane 0 writes s_memtime into
// g_mvh_stamp[(block * 16 + wave) * 32 + slot]; MVH_STAMP_READER(name) in the instrumented file copies them out.
// In the product build MVH_STAMPX expands to nothing.
#ifdef MVH_STAMP
static __device__ unsigned long long g_mvh_stamp[512 * 16 * 32];   // (no relocatable device code: one table per translation unit)
// the instrumented translation unit exports its reader under its own name (diagnostic build only, not in the header)
#define MVH_STAMP_READER(name)                                                                                  \
  extern "C" int name(unsigned long long* host, int clear) {                                                    \
    if (host && hipMemcpyFromSymbol(host, HIP_SYMBOL(g_mvh_stamp), sizeof(g_mvh_stamp)) != hipSuccess) return 1; \
    void* p__ = nullptr;                                                                                        \
    if (clear && (hipGetSymbolAddress(&p__, HIP_SYMBOL(g_mvh_stamp)) != hipSuccess ||                            \
                  hipMemset(p__, 0, sizeof(g_mvh_stamp)) != hipSuccess)) return 1;                               \
    return 0;                                                                                                   \
  }
#define MVH_STAMPX(slot)                                                                                      \
  do {                                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
    unsigned long long t__;                                                                                   \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");                                \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 512 && (slot) < 32)                                           \
      g_mvh_stamp[((long long)blockIdx.x * 16 + (threadIdx.x >> 6)) * 32 + (slot)] = t__;                     \
  } while (0)
#else
#define MVH_STAMPX(slot) do { } while (0)
#endif

namespace mvh {

constexpr int kWave = 64;  // CDNA4 wavefront

char* last_error_buf();
int fail(int code, const char* fmt, ...);

#define MVH_REQUIRE(cond, ...)                                   \
  do {                                                           \
    if (!(cond)) return ::mvh::fail(MVH_ERR_INVALID, __VA_ARGS__); \
  } while (0)

#define MVH_HIP(expr)                                                                   \
  do {                                                                                  \
    hipError_t e__ = (expr);                                                            \
    if (e__ != hipSuccess)                                                              \
      return ::mvh::fail(MVH_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                         __FILE__, __LINE__);                                           \
  } while (0)

#define MVH_LAUNCH_CHECK() MVH_HIP(hipGetLastError())

inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

int check_csr(const mvh_csr_t* op, const char* what);

// "The dynamic-LDS limit of this kernel is already >= bytes on the current device": one static instance per launch
// site (= per kernel instantiation), keyed by device and lock-free, so that launches from several host threads (the
// eager n_micro > 1 mode) and on several devices of one process are safe.  hipFuncSetAttribute itself is idempotent.
struct LdsAttr {
  std::atomic<size_t> have[16];
  LdsAttr() { for (auto& h : have) h.store(0, std::memory_order_relaxed); }
  int ensure(const void* kern, size_t bytes) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = -1;
    if (dev >= 0 && have[dev].load(std::memory_order_acquire) >= bytes) return MVH_OK;
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return fail(MVH_ERR_HIP, "hipFuncSetAttribute(dynamic LDS %zu) failed: %s", bytes, hipGetErrorString(e));
    if (dev >= 0) {
      size_t cur = have[dev].load(std::memory_order_relaxed);
      while (cur < bytes && !have[dev].compare_exchange_weak(cur, bytes, std::memory_order_release)) {}
    }
    return MVH_OK;
  }
};

// Debug / A-B switches of the whole library: ONE struct, filled once when the library is loaded from
// MESHVAE_DEBUG="key=value,key=value" (keys = the member names) and changed afterwards only through
// mvh_debug_set() (the tests flip force_generic / l0_wide inside one process).  No launcher reads the
// environment.
struct DebugCfg {
  int force_generic = 0;   // 1: never take the LDS-resident kernels (general stack pipeline everywhere)
  int l0_wide = 0;         // 1: 512 threads x 10 vertices at the 5k level instead of 1024 x 5
  int side_prio = 0;       // -1 / +1: low / high queue priority for the weight-gradient lanes
  int launcher_lanes = 0;  // launcher jobs: 0 = gradient work inline on the job's stream, 1 = lowest-priority gradient lanes (2.5 ms/step)
  int no_side = 0;         // 1: weight gradients inline on the main stream
  int no_tstack = 0;       // 1: first-layer dW through the recurrence kernel instead of the saved stack
  int tail_main = 0;       // 1: encoder layer 1's dW runs on the main stream after layer 0's (round 1's choice; since the
                           //    deferred reductions emptied the side lane's backlog the lane is the faster place: 570 vs 576 us)
  int fork_batch = 1;      // conv layers sharing one fork event (1..4)
  int l0_lane = 2;         // level-0 weight gradient on the dense lane, behind the level-0 dX, in this many launches (0: conv lane, one launch)
  int l0_lane_any = 0;     // 1: ... at any batch size >= 16 (default: 56 < B <= 64, where the one launch holds nearly every CU)
  int l0_lane_bf = 1;      // ... with bf16 storage as well
  int enc_dense = -2;      // encoder stage whose weight gradient runs on the dense lane instead of the conv lane (-1: none,
                           // -2: the coarsest one when the level-0 lane is in use)
  int l0_hold = 1;         // ... behind this many further forks of the main chain
  int fork_small = 400;    // with fork_batch = 1: layers of at most this many vertices share a fork in pairs (0: never)
  int no_gstack_mfma = 0;  // big-level fallbacks of cheb_conv.hip
  int no_dw_mfma = 0;
  int no_xcd_remap = 0;    // k_spmm tiles without the mesh -> XCD mapping
  int no_prefetch = 0;     // mvh_vae_backward_prefetch does nothing (the stack is built inside the backward)
  int no_l0h = 0;          // bf16 storage: keep the unpack-and-v_fma form at the 5k level (no cheb_l0h.hip kernel)
  int no_big = 0;          // levels above 5119 vertices keep the K - 1 SpMM launches (no cheb_big.hip kernel)
  int no_dx_tstack = 0;    // 16 -> 16 dX on a 5120 .. 20480-vertex level: G stack + Clenshaw kernel instead of T stack + contraction
  int no_dx_first = 0;     // streaming levels: fork a decoder stage's dW before (not behind) its dX, as at the small levels
  int no_bwd_fused = 0;    // ... and its dW / dX as two kernels reading two stacks instead of one pass over T_k(dpre)
  int no_dw_rows = 0;      // streaming levels: un-pool the gradient with its own launch, reduce the weight gradient over all rows
  int no_head_fuse = 0;    // dec_lin (forward and dX) as its own GEMM launch instead of inside the latent-head kernels
  int dw_lane2 = 0;        // 1: the small levels' conv weight gradients alternate between the two gradient lanes
  int tstack_tall = 0;     // 1: k_cheb_tstack as 1024 threads x 5 vertices instead of 512 x 10
  int prefetch_at = 0;     // encoder stage behind which the forward launches the armed first-layer stack (MEASURED 0..3: 553-555 us, no difference)
  int dw_tie_x = 0;        // LDS dW kernel, Cin == Cout: the recurrence runs on x and dout stays in registers (round 2's choice)
  int roctx = 0;           // 1: roctxRangePush / Pop around every layer of mvh_vae_forward / mvh_vae_backward (MVH_RANGE)
  int no_src3 = 0;         // 1: the final layer's dX writes all rows of its input gradient (no lazy rows in the 5k level's dX / dW)
  int no_final_fuse = 0;   // 1: the final layer's per-vertex map as its own launch (k_cheb_contract) instead of inside the loss launch
  int skip_conv_dw = 0;    // TIMING ONLY, results invalid: no conv weight-gradient launches on the side lane (how long is the main chain alone?)
  int sched = 0;           // 1: the conv weight-gradient items follow sched_lane / sched_hold instead of the built-in schedule (tools/sched_search.py)
  int sched_lane = 0;      // bit k: item k (final layer, dec stages last to first, enc stages last to first) on the dense lane
  int sched_hold = 0;      // base-4 digit k: forks of the chain item k lets pass before it is launched
  int skip_xty = 0;        // TIMING ONLY, results invalid: the final layer's S = x^T dout pass (k_xty_small) is not launched
  int big_half_ids = 0;    // TIMING ONLY, results invalid: k_cheb_big fetches 8 of the 16 id bytes per vertex and order (what would 1-byte ids buy?)
  int no_patch = 0;        // 1: never take the vertex-patch kernels (cheb_patch.hip): the slab kernels and their lanes everywhere
  int no_patch_bwd = 0;    // 1: the backward of such a layer stays on the slab kernels (forward on the patch kernel)
  int no_patch_bf16 = 0;   // 1: bf16 storage keeps the matrix-pipe slab kernels (cheb_l0h.hip) for the 5k level's backward
  int no_big_tstack = 0;   // 1: the first layer of a streaming level keeps the full T_k stack pipeline (no selected-rows stack)
  int no_contract_extras = 0;   // 1: the streaming levels' contraction writes no sign bytes / per-vertex map (k_relu_bits and the loss launch do)
  int no_patch_map = 0;    // 1: the final layer's per-vertex map stays in the loss launch / its own launch (not in the last decoder stage's epilogue)
  int no_patch_unpool = 0; // 1: the last decoder stage reads a stored un-pooled input (the stage before writes it) instead of un-pooling in its loads
  int no_enc0_patch = 0;   // 1: the first layer's forward stays on the slab kernel and the backward builds its stack (k_cheb_tstack)
  int patch_flush_first = 1;   // the step forks the weight-gradient items queued so far (the final layer's) BEFORE a patch backward
                               // launch, so that they run beside it (MEASURED: 458 against 485 us per step; 0: behind it)
  int keep_enc_out = 0;    // 1: the encoder convs store their whole output and every sign byte (ConvIO::out_dead off)
};
DebugCfg& dbg();

// roctx range around the enqueue of one layer / phase of the step (rocprofv3 --marker-trace attributes the kernels
// launched inside it): active only under the debug switch `roctx` (the marker library, librocprofiler-sdk-roctx.so, is
// loaded with dlopen on first use; without the switch the constructor is one predictable branch).
struct RoctxRange {
  bool on;
  RoctxRange(const char* fmt, ...);
  ~RoctxRange();
};
#define MVH_RANGE(...) ::mvh::RoctxRange mvh_range__(__VA_ARGS__)

// ---- internal launchers shared between translation units (all async on `st`)
int launch_spmm(hipStream_t st, const mvh_csr_t* op, const float* x, float* y, const float* add,
                const float* z, float alpha, float beta, int B, int C, bool exact,
                bool y_bf16 = false /* y is stored as bf16 (C % 4 == 0 only) */);
// C[M,N] = A (M x K, strides sam,sak) * Bm (K x N, strides sbk,sbn) (+bias[n]) -> act -> dropout
int launch_gemm(hipStream_t st, const float* A, long long sam, long long sak, const float* Bm,
                long long sbk, long long sbn, float* C, int M, int N, int K, const float* bias,
                int act, const float* drop_u, float p);

// deferred reduction of the dW partial tiles (one launch for all conv layers of a step)
struct DwReduceEntry {
  const float* part;
  int n_part, NS, K, CQ, CP, p_is_x, Cin, Cout, db_mode;
  float* dW;
  float* db;
  // split path (mostly-isolated Laplacian, cheb_conv.hip): `part` holds the tiles of the CONNECTED block only and
  // dW_k = dWsub_k + T_k(0) (S - dWsub_0) with S = sum over ALL rows of x^T dpre [Cin][Cout]; null = plain sum
  const float* S = nullptr;
};
struct DwReduceTable {
  DwReduceEntry e[2 * MVH_VAE_MAX_LAYERS + 1];
  int n;
};
int launch_dw_reduce_all(hipStream_t st, const DwReduceTable& t);
// LDS-resident fused ChebConv (cheb_lds.hip); *handled == false -> caller uses the general pipeline
struct LdsConvOpts {
  const float* prepacked = nullptr;   // slab-packed weights already built (launch_pack_all)
  int in_bs = 0, out_bs = 0, mask_bs = 0;  // rows per mesh of in / out / mask buffers (0 = N; mask: = in_bs)
  const int32_t* in_map = nullptr;    // input row v = in[in_map[v]], zero when < 0 (fused un-pooling of dout)
  const int32_t* pool_inv = nullptr;  // fused one-hot pooling of the output into `pooled`
  float* pooled = nullptr;
  int pooled_bs = 0;
  bool dry_run = false;               // only report eligibility through *handled
  // ReLU sign bytes [B][rows][C/4] (bit j of byte c/4 = out[v][c+j] > 0): written by the forward
  // kernel (bits_out) and read by the backward kernels instead of the fp32 output (mask_bits)
  const uint8_t* mask_bits = nullptr;
  uint8_t* bits_out = nullptr;
  // pool the result rows with this CSR (n_cols = N) inside the kernel.  Backward: `out` is then the pooled
  // buffer [B][n_rows][CO] and the un-pooled rows are never stored; forward: `out` as usual, pooled rows to `pooled`
  const mvh_csr_t* out_pool_t = nullptr;
  // bf16 STORAGE of the activation tensors (bf16.hpp): `in`, `out` (backward with out_pool_t: the pooled buffer) and
  // `pooled` are then 2-byte tensors behind the float pointers; arithmetic and the LDS state stay fp32
  bool in_bf16 = false, out_bf16 = false, pooled_bf16 = false;
  bool out_dead = false;                   // with pool_inv: store only the selected rows (pooled) and their sign bytes
  const float* src3_g = nullptr;           // backward: input rows >= src3_n are src3_g[v][0..src3_c) W^T (ConvIO::src3_*)
  const float* src3_w = nullptr;
  int src3_n = 0, src3_c = 0;
  const uint32_t* prepacked_h = nullptr;   // bf16 weight slabs of cheb_l0h.hip already built (launch_pack_all)
};
// level-0 16 -> 16 forward / dX on bf16 rows with the contraction on the matrix pipe (cheb_l0h.hip)
int try_cheb_l0h(hipStream_t st, const mvh_csr_t* lap, const float* in, const uint8_t* mask_bits, const float* W,
                 const float* bias, float* out, int B, int N, int Cin, int Cout, int K, int act, bool bwd, void* wpack,
                 bool* handled, const LdsConvOpts& o);
int l0h_pack_dwords(int K);
// ... and its weight gradient (cheb_dw_l0h.hip)
int try_cheb_dw_l0h(hipStream_t st, const mvh_csr_t* lap, const float* x, const float* dout, const uint8_t* out_bits,
                    float* dW, float* db, int B, int N, int Cin, int Cout, int K, float* part, size_t part_bytes,
                    bool* handled, bool dry_run, DwReduceEntry* defer, int dw_split = 1);
// levels too big for the slab kernels (5120 .. 20480 vertices): the whole recurrence of a (mesh, channel pair) in one
// launch (cheb_big.hip): the T_k stack of the forward / dW, and the Clenshaw sum of a stack G [K][B][N][C] (dX).
// pm: the stack planes are pair-major [B][C/2][N][2] instead of rows [B][N][C] (what the 16 -> 16 producers / consumers
// of cheb_conv.hip write / read when cheb_big_eligible holds: contiguous streams for the pair-workgroups)
bool cheb_big_eligible(const mvh_csr_t* lap, int B, int N, int C, int K);
int try_cheb_big_tx(hipStream_t st, const mvh_csr_t* lap, const float* x, float* tx, int B, int N, int C, int K, bool pm,
                    bool* handled, const float* mask = nullptr, bool with_t0 = false);
int try_cheb_big_clenshaw(hipStream_t st, const mvh_csr_t* lap, const float* G, float* out, int B, int N, int C, int K,
                          bool pm, bool* handled);
int try_cheb_lds(hipStream_t st, const mvh_csr_t* lap, const float* in, const float* mask, const float* W,
                 const float* bias, float* out, int B, int N, int Cin, int Cout, int K, int act, bool bwd,
                 float* wpack /* kLdsWpackBytes of scratch */, bool* handled, const LdsConvOpts& o = LdsConvOpts());
struct PackEntry {
  const float* W;
  float* dst;
  int K, Cin, Cout, CQ, CO, bwd;  // bwd: 0 forward slabs, 1 W^T slabs, 2 = W_eff [Cin*Cout] of the split path,
                                  //      3 / 4 = bf16 slabs of cheb_l0h.hip, forward / W^T (16 -> 16 only)
};
struct PackTable {
  PackEntry e[2 * (MVH_VAE_MAX_LAYERS * 2 + 1) + 3];
  int n;
};
int pack_entry_floats(int Cin, int Cout, int K, bool bwd);
int launch_pack_all(hipStream_t st, const PackTable& t);
// Storage type of the activation tensors of one conv call (bf16.hpp): true = 2-byte bf16 elements behind the float
// pointer.  bf16 tensors exist only on the LDS-resident / split paths; a layer that would need the general stack
// pipeline fails with MVH_ERR_UNSUPPORTED instead of falling back.
struct ConvIO {
  bool x = false, out = false, pooled = false;          // forward: input, output, fused-pooling output
  bool dout = false, dx = false, dx_pooled = false;     // backward: output gradient, input gradient, its pooled form
  const uint32_t* wh = nullptr;                         // bf16 weight slabs of cheb_l0h.hip for this call (forward
                                                        // layout in the forward, W^T layout in the backward) or null
  float* s_keep = nullptr;                              // split path + deferred reduction: where S [Cin*Cout] may stay
                                                        // until launch_dw_reduce_all (the scratch is reused before)
  // forward with a fused one-hot pooling: nobody reads the output rows the pooling does NOT select, nor their sign
  // bytes (the encoder: the next layer takes the pooled tensor, the backward masks the un-pooled gradient, which is
  // zero off the selected rows) -- the LDS-resident kernel then stores the pooled rows and their sign bytes only (a
  // quarter of the epilogue's stores, 20 MB less HBM traffic at the 5k level); every other path ignores the hint
  bool out_dead = false;
  // "lazy" output gradient of a mostly-isolated (split-path) layer, cheb_VAE.py:288: off its connected block the layer is
  // the per-vertex map x W_eff, so its dX row is dout_row W_eff^T -- three numbers per vertex at the final layer.
  //   dx_lazy (that layer's backward): write ONLY the connected block's rows of dx (no all-rows pass, no [B, N, Cin]
  //   tensor streamed through HBM); the consumer below rebuilds the other rows while it loads them.
  //   src3_* (the backward of the layer underneath, its dout = that dx): row v >= src3_n is src3_g[v][0..src3_c) W^T
  //   with W = src3_w [C][src3_c] (W_eff), rows < src3_n are read from `dout` as stored.  LDS-resident fp32 kernels
  //   of the 5k level only; a layer that cannot take the hint fails loudly (the step engine checks eligibility first).
  bool dx_lazy = false;
  bool out_lazy = false;   // forward of a split-path layer: write ONLY the connected block's rows of `out` (the loss launch
                           // rebuilds the others, loss_fwd_impl's fuse_* arguments)
  const float* src3_g = nullptr;
  const float* src3_w = nullptr;
  int src3_n = 0, src3_c = 0;
  // Strided input x of a module-level call (mvh_cheb_conv_*_strided, SURVEY 8(b): the reference hands [N, B, C]-physical
  // views around, nn/conv.py:560): row v of mesh b starts at  x + (b * x_bs + x_map[v]) * Cin  -- x_bs = mesh stride / Cin,
  // x_map[v] = v * row stride / Cin (device int32 [N]).  The LDS-resident kernels read it through the row map and the
  // rows-per-mesh parameter they already have (the fused un-pooling's), with no copy; every other path refuses
  // (MVH_ERR_UNSUPPORTED, the caller copies).  Not part of any(): the elements are fp32.
  const int32_t* x_map = nullptr;
  int x_bs = 0;
  // weight gradient of the 5k level (k_cheb_dw_lds<.., 10, 512, ..>: one 160 KB workgroup per CU): the batch in dw_split
  // launches one behind the other, so that the kernel holds 1 / dw_split of the CUs at a time (the step engine's level-0 lane)
  int dw_split = 1;
  // forward of the first layer (<= 4 -> 16 channels, ReLU, one-hot pooling, out_dead) on the vertex-patch kernel
  // (cheb_patch.hip, k_patch_enc0): it also leaves T_k(L) x at the pooled rows, the stack [B][K][n_sel + 1][4] of
  // cheb_tstack.hip, in stack_out, and says so in *stack_done (left untouched on every other path)
  float* stack_out = nullptr;
  bool* stack_done = nullptr;
  // forward of a 16 -> 16 layer on the vertex-patch kernel whose plan carries the level's un-pooling rows (urec): `x` is the
  // COARSE tensor [B][plan.u_rows][16] and the kernel un-pools it while it loads (the previous stage then stores no
  // un-pooled rows: 20 MB less written and read at the 5k level); x_store != NULL: the un-pooled rows [B][N][16] are
  // written there once (the backward's dW operand).  Patch kernel only (patch_unpool_eligible); every other path refuses.
  bool x_unpool = false;
  float* x_store = nullptr;
  // ... and a per-vertex map BEHIND the layer, out of the same kernel's epilogue: rows >= map_n0 of map_out [B][N][map_c]
  // = out x map_w [Cout = 16][map_c] (map_c <= 4), in k_cheb_contract's fma order.  The VAE's final conv off its connected
  // block (cheb_VAE.py:288): neither that layer's map launch nor the loss launch then reads this layer's output.
  const float* map_w = nullptr;
  float* map_out = nullptr;
  int map_c = 0, map_n0 = 0;
  bool* map_done = nullptr;     // set when the kernel that honours map_* ran (every other path leaves it alone)
  bool any() const { return x || out || pooled || dout || dx || dx_pooled; }
};
// does this layer take the split path of cheb_conv.hip (mostly-isolated Laplacian: per-vertex map + connected block)?
bool conv_split_eligible(const mvh_csr_t* lap, int N, int Cin, int Cout, int K);
// conv entry points with optional prepacked weights (the extern "C" functions pass nullptr)
int cheb_conv_fwd_impl(hipStream_t st, const mvh_csr_t* lap, const float* x, const float* W, const float* bias,
                       float* out, float* tx_saved, int B, int N, int Cin, int Cout, int K, int act, void* ws,
                       size_t ws_bytes, const float* prepacked, const mvh_csr_t* pool = nullptr,
                       float* pooled = nullptr /* fused one-hot pooling of the output (falls back to a launch) */,
                       uint8_t* bits_out = nullptr /* ReLU sign bytes [B][N][Cout/4] of the output (Cout % 4 == 0) */,
                       const float* weff_pre = nullptr /* W_eff already built by launch_pack_all (split path) */,
                       const ConvIO& io = ConvIO());
int cheb_conv_bwd_impl(hipStream_t st, const mvh_csr_t* lap, const mvh_csr_t* lap_t, const float* x, const float* W,
                       const float* out, const float* dout, const float* tx_saved, float* dx, float* dW, float* db,
                       int B, int N, int Cin, int Cout, int K, int act, void* ws, size_t ws_bytes,
                       const float* prepacked_bwd, const mvh_csr_t* dout_pool = nullptr /* dout is the gradient of
                       the POOLED output [B, dout_pool->n_rows, Cout]; un-pooling is fused into the loads */,
                       bool* fused_ok = nullptr /* set false (nothing launched) when that fusion is not available */,
                       const uint8_t* out_bits = nullptr /* sign bytes from the forward; `out` stays the fallback */,
                       const float* weff_pre = nullptr,
                       DwReduceEntry* defer = nullptr /* with defer_part: the LDS dW kernel writes its partial tiles */,
                       float* defer_part = nullptr    /* there and *defer describes the pending reduction         */,
                       size_t defer_bytes = 0, bool* deferred = nullptr,
                       const mvh_csr_t* dx_pool_t = nullptr /* store dx_pooled = dx_pool_t * dx instead of dx (the      */,
                       float* dx_pooled = nullptr           /* decoder's upsampling backward); falls back to dx + spmm */,
                       const ConvIO& io = ConvIO());
constexpr size_t kLdsWpackBytes = 64 * 1024;
// vertex-patch kernels of a level's 16 -> 16 layers (cheb_patch.hip; the plan hangs off the Laplacian: mvh_csr_t::patch)
bool patch_eligible(const mvh_csr_t* lap, int N, int Cin, int Cout, int K);
bool patch_enc0_eligible(const mvh_csr_t* lap, const mvh_csr_t* down, int N, int Cin, int Cout, int K);
int launch_patch_enc0(hipStream_t st, const mvh_csr_t* lap, const mvh_csr_t* down, const float* x, const float* W,
                      const float* bias, float* pooled, bool pooled_bf16, uint8_t* bits, float* stack, int B, int N,
                      int Cin, int K, int act);
size_t patch_part_bytes(const mvh_csr_t* lap, int B, int K);
int launch_patch_fwd(hipStream_t st, const mvh_csr_t* lap, const float* x, const float* W, const float* bias, float* out,
                     uint8_t* bits, int B, int N, int K, int act, const int32_t* x_map = nullptr /* ConvIO::x_map */,
                     int x_bs = 0, bool x_unpool = false /* ConvIO::x_unpool */, float* x_store = nullptr,
                     const float* map_w = nullptr /* ConvIO::map_* */, float* map_out = nullptr, int map_c = 0, int map_n0 = 0);
bool patch_unpool_eligible(const mvh_csr_t* lap, const mvh_csr_t* up, int N, int Cin, int Cout, int K);
int launch_patch_bwd(hipStream_t st, const mvh_csr_t* lap, const float* x, const float* W, const float* dout,
                     const uint8_t* mbits, const float* g3, const float* w3, int src3_n, float* dx, bool pooled,
                     float* part, size_t part_bytes, DwReduceEntry* defer, float* dW, float* db, int B, int N, int K,
                     const int32_t* x_map = nullptr /* ConvIO::x_map: strided layer input (the dW operand) */, int x_bs = 0,
                     bool x_bf16 = false, bool dout_bf16 = false, bool dx_bf16 = false /* bf16 STORAGE of x / dout / dx */);
// LDS-resident dW/db (cheb_dw_lds.hip): `part` is scratch of cheb_dw_lds_ws_bytes()
size_t cheb_dw_lds_ws_bytes(int B, int N, int Cin, int Cout, int K);
int try_cheb_dw_lds(hipStream_t st, const mvh_csr_t* lap, const float* x, const float* dout, const float* out_mask,
                    float* dW, float* db, int B, int N, int Cin, int Cout, int K, float* part, size_t part_bytes,
                    bool* handled, int bstride = 0 /* rows per mesh of x/dout/out_mask (0 = N) */,
                    const int32_t* dout_map = nullptr, int dout_rows = 0 /* dout row v = dout[map[v]] (zero if < 0),
                                                                            compact buffer of dout_rows per mesh */,
                    bool dry_run = false, const uint8_t* out_bits = nullptr /* replaces out_mask when given */,
                    DwReduceEntry* defer = nullptr /* skip the reduce launch and describe it here instead */,
                    bool x_bf16 = false, bool dout_bf16 = false /* storage type of x / dout (bf16.hpp) */,
                    const ConvIO* src3 = nullptr /* dout rows >= src3_n come from src3_g W^T (ConvIO::src3_*) */);

// first-layer weight gradient through a saved Chebyshev stack (cheb_tstack.hip)
size_t tstack_stack_floats(int B, int n_sel, int K);   // (n_sel = rows the pooling selects = pool->n_rows)
size_t tstack_ws_floats(int B, int n_sel, int K, int Cin, int Cout);
bool tstack_eligible(const mvh_csr_t* lap, const mvh_csr_t* pool, int N, int Cin, int Cout, int K);
int launch_big_tstack(hipStream_t st, const mvh_csr_t* lap, const mvh_csr_t* pool, const float* x, float* stack, int B,
                      int N, int Cin, int K);     // (cheb_big.hip: the same stack on a level of the streaming kernels)
int launch_stack_contract(hipStream_t st, const mvh_csr_t* pool, const float* stack, const float* W, const float* bias,
                          float* pooled, uint8_t* bits, int B, int N, int Cin, int K);
int launch_tstack(hipStream_t st, const mvh_csr_t* lap, const mvh_csr_t* pool, const float* x, float* stack, int B,
                  int N, int Cin, int K);
int launch_stack_dw(hipStream_t st, const mvh_csr_t* pool, const float* stack, const float* dout, const uint8_t* bits,
                    const float* out_mask, float* dW, float* db, float* partial, int B, int N, int Cin, int Cout, int K,
                    DwReduceEntry* defer = nullptr /* leave the final sum to launch_dw_reduce_all */,
                    bool dout_bf16 = false);
// mvh_vae_loss_fwd with optional gradient seeds for d_loss = 1 (d_recon [B*NV], d_mu/d_logvar [B*Z], d_yhat [B*C])
int loss_fwd_impl(hipStream_t stream, const float* recon, const void* x_gt, int gt_f64, const float* mu,
                  const float* logvar, const float* y, const float* y_hat, float log_sigma, void* loss, void* rec,
                  float* kld, int64_t* correct, int B, int NV, int C, int Z, void* ws, size_t ws_bytes,
                  float* d_recon, float* d_mu, float* d_logvar, float* d_yhat,
                  // optional: the final layer's per-vertex map fused into the first loss launch (recon rows >= fuse_n_act
                  // are computed as fuse_x16[v] fuse_weff and WRITTEN to fuse_recon == recon; the others are read)
                  const float* fuse_x16 = nullptr, const float* fuse_weff = nullptr, float* fuse_recon = nullptr,
                  int fuse_cin = 0, int fuse_c3 = 0, int fuse_n_act = 0);
// mvh_vae_latent_fwd with an optional fused dec_lin: d1 = dropout(relu(cat[y, z] Wd^T + bd)) (drop_d: its uniforms, same
// p), bit-identical to mvh_linear_fwd; *fused tells whether the kernel took it (else the caller launches the GEMM)
int latent_fwd_impl(hipStream_t st, const float* h, const float* y, const float* drop_u, float p, const float* Wc,
                    const float* bc, const float* Wm, const float* bm, const float* Wv, const float* bv,
                    const float* eps, float* y_hat, float* mu, float* logvar, float* z, float* zy, int B, int H, int C,
                    int Z, const float* Wd, const float* bd, const float* drop_d, float* d1, bool* fused);
// halves of mvh_vae_latent_bwd: dh + the head pre-activation gradients dpre [B, C + 2Z]; then the weight gradients.
// With Wd / d1 / g_d1 / g_zy the dX GEMM of dec_lin (mvh_linear_bwd, bit-identical) runs inside the first launch
// and d_zy is not read (g_zy is written); *fused as above.
int latent_bwd_heads(hipStream_t st, const float* drop_u, float p, const float* Wc, const float* Wm, const float* Wv,
                     const float* eps, const float* y_hat, const float* logvar, const float* d_yhat,
                     const float* d_mu, const float* d_logvar, const float* d_zy, float* dh, float* dpre, int B,
                     int H, int C, int Z, const float* Wd = nullptr, const float* d1 = nullptr,
                     const float* g_d1 = nullptr, float* g_zy = nullptr, bool* fused = nullptr);
int latent_bwd_wgrad(hipStream_t st, const float* h, const float* y, const float* drop_u, float p, const float* dpre,
                     float* dWc, float* dbc, float* dWm, float* dbm, float* dWv, float* dbv, int B, int H, int C,
                     int Z);

// the argument checks of mvh_vae_forward / mvh_vae_backward that do not touch the device (descriptor, batch, workspace size):
// the asynchronous entries run them on the caller's thread before they queue the job (csrc/launcher.hip)
int vae_step_precheck(const mvh_vae_desc_t* d, int B, const void* ws, size_t ws_bytes);

}  // namespace mvh
