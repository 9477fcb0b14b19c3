// C-ABI glue: error reporting, argument validation and the thin entry points
// (propagate / pool) that map 1:1 onto the sparse kernel.
#include <dlfcn.h>

#include "common.hpp"

namespace mvh {

char* last_error_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(last_error_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

int check_csr(const mvh_csr_t* op, const char* what) {
  MVH_REQUIRE(op != nullptr, "%s: null CSR descriptor", what);
  MVH_REQUIRE(op->n_rows >= 0 && op->n_cols >= 0 && op->nnz >= 0, "%s: negative CSR sizes", what);
  MVH_REQUIRE(op->rowptr != nullptr, "%s: null rowptr", what);
  MVH_REQUIRE(op->nnz == 0 || (op->col != nullptr && op->val != nullptr), "%s: null col/val", what);
  return MVH_OK;
}

struct DebugKey {
  const char* name;
  int DebugCfg::*field;
};
static const DebugKey kDebugKeys[] = {
    {"force_generic", &DebugCfg::force_generic}, {"l0_wide", &DebugCfg::l0_wide},
    {"side_prio", &DebugCfg::side_prio},         {"launcher_lanes", &DebugCfg::launcher_lanes},         {"no_side", &DebugCfg::no_side},
    {"no_tstack", &DebugCfg::no_tstack},         {"tail_main", &DebugCfg::tail_main},
    {"fork_batch", &DebugCfg::fork_batch},       {"no_gstack_mfma", &DebugCfg::no_gstack_mfma},
    {"no_dw_mfma", &DebugCfg::no_dw_mfma},       {"no_xcd_remap", &DebugCfg::no_xcd_remap},
    {"no_prefetch", &DebugCfg::no_prefetch},     {"no_l0h", &DebugCfg::no_l0h},
    {"no_head_fuse", &DebugCfg::no_head_fuse},   {"no_big", &DebugCfg::no_big},
    {"no_dx_tstack", &DebugCfg::no_dx_tstack},   {"no_dx_first", &DebugCfg::no_dx_first},
    {"no_bwd_fused", &DebugCfg::no_bwd_fused},   {"no_dw_rows", &DebugCfg::no_dw_rows},
    {"keep_enc_out", &DebugCfg::keep_enc_out},   {"dw_lane2", &DebugCfg::dw_lane2},
    {"tstack_tall", &DebugCfg::tstack_tall},     {"prefetch_at", &DebugCfg::prefetch_at},
    {"dw_tie_x", &DebugCfg::dw_tie_x},           {"roctx", &DebugCfg::roctx},
    {"no_src3", &DebugCfg::no_src3},             {"skip_conv_dw", &DebugCfg::skip_conv_dw},
    {"big_half_ids", &DebugCfg::big_half_ids},     {"skip_xty", &DebugCfg::skip_xty},
    {"sched", &DebugCfg::sched},                   {"sched_lane", &DebugCfg::sched_lane},
    {"sched_hold", &DebugCfg::sched_hold},
    {"no_final_fuse", &DebugCfg::no_final_fuse}, {"fork_small", &DebugCfg::fork_small},
    {"l0_lane", &DebugCfg::l0_lane},             {"l0_lane_any", &DebugCfg::l0_lane_any},
    {"l0_lane_bf", &DebugCfg::l0_lane_bf},       {"l0_hold", &DebugCfg::l0_hold},
    {"enc_dense", &DebugCfg::enc_dense},         {"no_patch", &DebugCfg::no_patch},
    {"patch_flush_first", &DebugCfg::patch_flush_first}, {"no_patch_bwd", &DebugCfg::no_patch_bwd},
    {"no_enc0_patch", &DebugCfg::no_enc0_patch}, {"no_patch_bf16", &DebugCfg::no_patch_bf16}, {"no_patch_unpool", &DebugCfg::no_patch_unpool}, {"no_patch_map", &DebugCfg::no_patch_map}, {"no_contract_extras", &DebugCfg::no_contract_extras}, {"no_big_tstack", &DebugCfg::no_big_tstack},
};

static int DebugCfg::*find_debug_key(const char* key, size_t len) {
  for (const DebugKey& k : kDebugKeys)
    if (strlen(k.name) == len && strncmp(k.name, key, len) == 0) return k.field;
  return nullptr;
}

static DebugCfg parse_debug_env() {
  DebugCfg c;
  const char* e = getenv("MESHVAE_DEBUG");
  while (e && *e) {
    const char* end = strchr(e, ',');
    const size_t len = end ? (size_t)(end - e) : strlen(e);
    const char* eq = (const char*)memchr(e, '=', len);
    if (eq) {
      if (int DebugCfg::*f = find_debug_key(e, (size_t)(eq - e))) c.*f = atoi(eq + 1);
      else fprintf(stderr, "libmeshvae_hip: unknown MESHVAE_DEBUG key '%.*s'\n", (int)(eq - e), e);
    }
    e = end ? end + 1 : nullptr;
  }
  if (c.fork_batch < 1) c.fork_batch = 1;
  if (c.fork_batch > 4) c.fork_batch = 4;
  return c;
}

DebugCfg& dbg() {
  static DebugCfg c = parse_debug_env();
  return c;
}
static const DebugCfg& dbg_at_load = dbg();  // parsed when the library is loaded, not at the first launch

}  // namespace mvh

using namespace mvh;

extern "C" int mvh_debug_set(const char* key, int32_t value) {
  MVH_REQUIRE(key != nullptr, "debug_set: null key");
  int DebugCfg::*f = find_debug_key(key, strlen(key));
  MVH_REQUIRE(f != nullptr, "debug_set: unknown key '%s'", key);
  dbg().*f = value;
  return MVH_OK;
}

extern "C" int32_t mvh_debug_get(const char* key) {
  int DebugCfg::*f = key ? find_debug_key(key, strlen(key)) : nullptr;
  return f ? dbg().*f : -1;
}

namespace mvh {
namespace {
typedef int (*roctx_push_t)(const char*);
typedef int (*roctx_pop_t)(void);
roctx_push_t g_roctx_push = nullptr;
roctx_pop_t g_roctx_pop = nullptr;
std::atomic<int> g_roctx_state{0};   // 0 not tried, 1 loaded, -1 unavailable
bool roctx_ready() {
  int s = g_roctx_state.load(std::memory_order_acquire);
  if (s == 0) {
    void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librocprofiler-sdk-roctx.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
    roctx_push_t pu = h ? (roctx_push_t)dlsym(h, "roctxRangePushA") : nullptr;
    roctx_pop_t po = h ? (roctx_pop_t)dlsym(h, "roctxRangePop") : nullptr;
    if (pu && po) { g_roctx_push = pu; g_roctx_pop = po; s = 1; } else s = -1;
    g_roctx_state.store(s, std::memory_order_release);
  }
  return s == 1;
}
}  // namespace
RoctxRange::RoctxRange(const char* fmt, ...) : on(false) {
  if (!dbg().roctx || !roctx_ready()) return;
  char buf[96];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_roctx_push(buf);
  on = true;
}
RoctxRange::~RoctxRange() {
  if (on) g_roctx_pop();
}
}  // namespace mvh

extern "C" int mvh_version(void) { return MVH_ABI_VERSION; }

extern "C" const char* mvh_last_error(void) { return last_error_buf(); }

extern "C" int mvh_device_info(int* n_cu, int* lds_bytes_per_cu, char* arch, int arch_len) {
  int dev = 0;
  MVH_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  MVH_HIP(hipGetDeviceProperties(&prop, dev));
  if (n_cu) *n_cu = prop.multiProcessorCount;
  if (lds_bytes_per_cu) *lds_bytes_per_cu = (int)prop.maxSharedMemoryPerMultiProcessor;
  if (arch && arch_len > 0) {
    strncpy(arch, prop.gcnArchName, (size_t)arch_len - 1);
    arch[arch_len - 1] = 0;
  }
  return MVH_OK;
}

extern "C" int mvh_spmm(mvh_stream_t stream, const mvh_csr_t* op, const float* x, float* y, const float* add,
                        const float* z, float alpha, float beta, int32_t B, int32_t C, int32_t exact) {
  if (int rc = check_csr(op, "spmm")) return rc;
  MVH_REQUIRE(x && y, "spmm: null tensor");
  MVH_REQUIRE(B >= 0 && C > 0, "spmm: bad sizes B=%d C=%d", B, C);
  return launch_spmm((hipStream_t)stream, op, x, y, add, z, alpha, beta, B, C, exact != 0);
}

extern "C" int mvh_pool_fwd(mvh_stream_t stream, const mvh_csr_t* pool, const float* x, float* y, int32_t B,
                            int32_t C) {
  if (int rc = check_csr(pool, "pool_fwd")) return rc;
  MVH_REQUIRE(x && y, "pool_fwd: null tensor");
  MVH_REQUIRE(B >= 0 && C > 0, "pool_fwd: bad sizes B=%d C=%d", B, C);
  return launch_spmm((hipStream_t)stream, pool, x, y, nullptr, nullptr, 1.f, 0.f, B, C, true);
}

// SurfacePool.forward on a strided [B, n_cols, C] view (element strides x_mesh_stride, x_row_stride, 1; nn/pool.py:18
// hands the transposed view of its input to propagate): one thread per (mesh, output row, channel), products and adds
// rounded separately in edge order -- bit for bit mvh_pool_fwd on the contiguous copy.  y is contiguous.
__global__ void __launch_bounds__(256)
k_pool_strided(const int* __restrict__ rowptr, const int* __restrict__ col, const float* __restrict__ val, int n_rows,
               const float* __restrict__ x, long long ms, long long rs, float* __restrict__ y, int C, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C);
  const long long br = i / C;
  const int r = (int)(br % n_rows);
  const long long b = br / n_rows;
  const float* xb = x + b * ms + c;
  float acc = 0.f;
  for (int e = rowptr[r]; e < rowptr[r + 1]; ++e) acc = __fadd_rn(acc, __fmul_rn(val[e], xb[(long long)col[e] * rs]));
  y[i] = acc;
}

extern "C" int mvh_pool_fwd_strided(mvh_stream_t stream, const mvh_csr_t* pool, const float* x, int64_t x_mesh_stride,
                                    int64_t x_row_stride, float* y, int32_t B, int32_t C) {
  if (int rc = check_csr(pool, "pool_fwd_strided")) return rc;
  MVH_REQUIRE(x && y, "pool_fwd_strided: null tensor");
  MVH_REQUIRE(B >= 0 && C > 0 && x_mesh_stride >= 0 && x_row_stride > 0, "pool_fwd_strided: bad sizes B=%d C=%d", B, C);
  const long long total = (long long)B * pool->n_rows * C;
  if (total == 0) return MVH_OK;
  const long long nblk = (total + 255) / 256;
  MVH_REQUIRE(nblk < (1ll << 31), "pool_fwd_strided: grid too large");
  hipLaunchKernelGGL(k_pool_strided, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, pool->rowptr, pool->col,
                     pool->val, pool->n_rows, x, (long long)x_mesh_stride, (long long)x_row_stride, y, C, total);
  MVH_LAUNCH_CHECK();
  return MVH_OK;
}

extern "C" int mvh_pool_bwd(mvh_stream_t stream, const mvh_csr_t* pool_t, const float* dy, float* dx,
                            int32_t B, int32_t C) {
  if (int rc = check_csr(pool_t, "pool_bwd")) return rc;
  MVH_REQUIRE(dy && dx, "pool_bwd: null tensor");
  MVH_REQUIRE(B >= 0 && C > 0, "pool_bwd: bad sizes B=%d C=%d", B, C);
  return launch_spmm((hipStream_t)stream, pool_t, dy, dx, nullptr, nullptr, 1.f, 0.f, B, C, true);
}
