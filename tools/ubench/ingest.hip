// Micro-benchmark (tooling, not product): what does a level-0 slab workgroup pay to INGEST its rows and to STORE its slab?
//   rows  : [B][N][16] fp32 (the module-boundary layout): thread-owns-vertex = 4 x 16-byte loads per vertex at 64-byte lane stride
//   planes: [B][4][N][4] fp32 ("plane-major"): the same bytes, every wave-load 1 KB contiguous
// 256 workgroups x 1024 threads x 5 vertices (the shape of k_cheb_lds<16,5,1024,4,*>), 4 workgroups per mesh on one XCD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int N = 4998, B = 64, T = 1024, VPT = 5;

__device__ __forceinline__ void wg_map(int& mesh, int& slab) {
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  mesh = (jj / 4) * 8 + xcd; slab = jj % 4;
}

// MODE 0 rows (4 loads/vertex), 1 planes, 2 rows but only the own slab (1 load/vertex), 3 planes own slab only
template <int MODE>
__global__ void __launch_bounds__(T) k_load(const float4* __restrict__ in, float4* __restrict__ out, const uint4* __restrict__ ell, int with_ell) {
  int mesh, slab; wg_map(mesh, slab);
  const int tid = threadIdx.x;
  float4 acc = make_float4(0, 0, 0, 0);
  float4 r[VPT][4];
  uint4 id[VPT];
#pragma unroll
  for (int vi = 0; vi < VPT; ++vi) {
    const int v = min(tid + vi * T, N - 1);
    if (with_ell) id[vi] = ell[v];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (MODE == 0) r[vi][g] = in[((long long)mesh * N + v) * 4 + g];
      if (MODE == 1) r[vi][g] = in[((long long)mesh * 4 + g) * N + v];
      if (MODE == 2) r[vi][g] = g == 0 ? in[((long long)mesh * N + v) * 4 + slab] : make_float4(0, 0, 0, 0);
      if (MODE == 3) r[vi][g] = g == 0 ? in[((long long)mesh * 4 + slab) * N + v] : make_float4(0, 0, 0, 0);
    }
  }
#pragma unroll
  for (int vi = 0; vi < VPT; ++vi) {
#pragma unroll
    for (int g = 0; g < 4; ++g) { acc.x += r[vi][g].x; acc.y += r[vi][g].y; acc.z += r[vi][g].z; acc.w += r[vi][g].w; }
    if (with_ell) acc.x += __uint_as_float(id[vi].x ^ id[vi].y ^ id[vi].z ^ id[vi].w);
  }
  out[(long long)blockIdx.x * T + tid] = acc;
}

// MODE 0: slab store into rows (16 bytes at 64-byte stride), 1: into the slab's plane (contiguous)
template <int MODE>
__global__ void __launch_bounds__(T) k_store(float4* __restrict__ out, float seed) {
  int mesh, slab; wg_map(mesh, slab);
  const int tid = threadIdx.x;
#pragma unroll
  for (int vi = 0; vi < VPT; ++vi) {
    const int v = tid + vi * T;
    if (v >= N) continue;
    const float4 val = make_float4(seed + v, seed, slab, mesh);
    if (MODE == 0) out[((long long)mesh * N + v) * 4 + slab] = val;
    else out[((long long)mesh * 4 + slab) * N + v] = val;
  }
}

template <typename F> float timeit(F f, int reps, hipStream_t st, float* scrub, size_t scrub_n) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e9, sum = 0;
  for (int i = 0; i < reps; ++i) {
    if (scrub) CK(hipMemsetAsync(scrub, i, scrub_n, st));   // push the operands out of L2 (not out of the 256 MB MALL unless scrub is big)
    CK(hipEventRecord(a, st)); f(); CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); best = ms < best ? ms : best; if (i) sum += ms;
  }
  printf("  min %.1f us  avg %.1f us", best * 1e3, sum / (reps - 1) * 1e3);
  return best;
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  const size_t n4 = (size_t)B * N * 4;
  float4 *in, *out, *big; uint4* ell; float* scrub;
  CK(hipMalloc(&in, n4 * 16)); CK(hipMalloc(&big, n4 * 16)); CK(hipMalloc(&out, (size_t)256 * T * 16)); CK(hipMalloc(&ell, (size_t)N * 16));
  const size_t scrub_small = 64u << 20, scrub_big = 600u << 20;
  CK(hipMalloc(&scrub, scrub_big));
  CK(hipMemset(in, 0, n4 * 16)); CK(hipMemset(ell, 1, (size_t)N * 16));
  for (int cold = 0; cold < 3; ++cold) {
    float* sc = cold == 0 ? nullptr : scrub; size_t sn = cold == 1 ? scrub_small : scrub_big;
    printf("== operands %s\n", cold == 0 ? "hot (back-to-back replays)" : cold == 1 ? "out of L2 (64 MB memset between), MALL-resident" : "out of MALL (600 MB memset between)");
    for (int e = 0; e < 2; ++e) {
      printf(" load rows   16ch ell=%d:", e); timeit([&] { hipLaunchKernelGGL(k_load<0>, dim3(256), dim3(T), 0, st, in, out, ell, e); }, 12, st, sc, sn); printf("\n");
      printf(" load planes 16ch ell=%d:", e); timeit([&] { hipLaunchKernelGGL(k_load<1>, dim3(256), dim3(T), 0, st, in, out, ell, e); }, 12, st, sc, sn); printf("\n");
    }
    printf(" load rows   own slab   :"); timeit([&] { hipLaunchKernelGGL(k_load<2>, dim3(256), dim3(T), 0, st, in, out, ell, 0); }, 12, st, sc, sn); printf("\n");
    printf(" load planes own slab   :"); timeit([&] { hipLaunchKernelGGL(k_load<3>, dim3(256), dim3(T), 0, st, in, out, ell, 0); }, 12, st, sc, sn); printf("\n");
    printf(" store slab into rows   :"); timeit([&] { hipLaunchKernelGGL(k_store<0>, dim3(256), dim3(T), 0, st, big, 1.f); }, 12, st, sc, sn); printf("\n");
    printf(" store slab into planes :"); timeit([&] { hipLaunchKernelGGL(k_store<1>, dim3(256), dim3(T), 0, st, big, 1.f); }, 12, st, sc, sn); printf("\n");
  }
  return 0;
}
