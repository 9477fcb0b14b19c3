// Micro-benchmark (tooling, not product): what does a layer boundary of the small-level chain cost as a KERNEL boundary,
// and what as an in-kernel hand-off between the slab-workgroups of one mesh?  ("what comes next".)
//
// Shape of enc2 -> enc3 of the 5k model at B = 64: a producer layer of 4 slab-workgroups per mesh, a consumer layer of 8,
// 320 threads each; the producer leaves ROWS rows x 16 floats per mesh (4 floats per slab), every consumer workgroup reads
// all of them.  Both layers run WORK rounds of {LDS write, barrier, 8 LDS gathers, barrier} as stand-in for a K = 6 layer.
//   A  two launches per boundary (what the step does today)
//   B  one launch: producers store, every wave waits for its stores, barrier, ONE lane agent-release fence + counter add;
//      every workgroup of the mesh polls the counter (relaxed), agent-acquire fence, barrier, plain loads
//      (MI355X_MICROARCH.md, the fence form)
//   C  one launch: sc1 stores and sc1 loads, counter add behind the waves' waits and a barrier, no fence (the sc1 form)
// Every poll is BOUNDED (an error flag is raised instead of spinning forever).  Prints microseconds per boundary.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int B = 64, NSA = 4, NSB = 8, T = 320;

__device__ __forceinline__ float work(float4* lds, float seed, int rounds) {
  float acc = seed;
  for (int r = 0; r < rounds; ++r) {
    lds[threadIdx.x] = make_float4(acc, acc + 1.f, acc + 2.f, acc + 3.f);
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) s += lds[(threadIdx.x * 7 + q * 37 + r) % T].x;
    acc = 0.125f * s;
    __syncthreads();
  }
  return acc;
}

typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_sc1(float4* p, float4 v) {
  const f4v x = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(x) : "memory");
}
// four sc1 loads in flight, then ONE wait -- inside the asm statement: the compiler does not know that an inline-asm load
// returns later and would read the registers at once
__device__ __forceinline__ void ld4_sc1(const float4* p0, const float4* p1, const float4* p2, const float4* p3, float4 (&v)[4]) {
  f4v a, b, c, d;
  asm volatile(
      "global_load_dwordx4 %0, %4, off sc0 sc1\n"
      "global_load_dwordx4 %1, %5, off sc0 sc1\n"
      "global_load_dwordx4 %2, %6, off sc0 sc1\n"
      "global_load_dwordx4 %3, %7, off sc0 sc1\n"
      "s_waitcnt vmcnt(0)"
      : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)
      : "v"(p0), "v"(p1), "v"(p2), "v"(p3)
      : "memory");
  v[0] = make_float4(a[0], a[1], a[2], a[3]);
  v[1] = make_float4(b[0], b[1], b[2], b[3]);
  v[2] = make_float4(c[0], c[1], c[2], c[3]);
  v[3] = make_float4(d[0], d[1], d[2], d[3]);
}

// MODE 0: plain stores / loads (kernel-boundary form and fence form); 1: sc1 stores / loads
template <int MODE>
__device__ __forceinline__ void produce(float4* mid, int mesh, int slab, int rows, int rounds, float4* lds) {
  const float a = work(lds, (float)(mesh + slab), rounds);
  for (int v = threadIdx.x; v < rows; v += T) {
    float4* p = mid + ((long long)mesh * rows + v) * 4 + slab;
    const float4 val = make_float4(a + v, a, slab, mesh);
    if (MODE == 1) st_sc1(p, val);
    else *p = val;
  }
}
template <int MODE>
__device__ __forceinline__ void consume(const float4* mid, float* out, int mesh, int slab, int rows, int rounds, float4* lds) {
  float s = 0.f;
  const float4* base = mid + (long long)mesh * rows * 4;
  for (int i = threadIdx.x; i < rows * 4; i += 4 * T) {   // four 16-byte loads in flight
    float4 v[4];
    if (MODE == 1) {
      const int last = rows * 4 - 1;
      ld4_sc1(base + min(i, last), base + min(i + T, last), base + min(i + 2 * T, last), base + min(i + 3 * T, last), v);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = base[min(i + j * T, rows * 4 - 1)];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) s += (i + j * T < rows * 4) ? v[j].x + v[j].y : 0.f;
  }
  const float a = work(lds, s, rounds);
  if (threadIdx.x == 0) out[mesh * NSB + slab] = a + s;
}

__global__ void __launch_bounds__(T) k_prod(float4* mid, int rows, int rounds) {
  __shared__ float4 lds[T];
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3, mesh = (jj / NSA) * 8 + xcd, slab = jj % NSA;
  produce<0>(mid, mesh, slab, rows, rounds, lds);
}
__global__ void __launch_bounds__(T) k_cons(const float4* mid, float* out, int rows, int rounds) {
  __shared__ float4 lds[T];
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3, mesh = (jj / NSB) * 8 + xcd, slab = jj % NSB;
  consume<0>(mid, out, mesh, slab, rows, rounds, lds);
}

template <int MODE>
__global__ void __launch_bounds__(T) k_fused(float4* mid, float* out, unsigned* counter, unsigned target, int* err, int rows, int rounds) {
  __shared__ float4 lds[T];
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3, mesh = (jj / NSB) * 8 + xcd, slab = jj % NSB;
  if (slab < NSA) {
    produce<MODE>(mid, mesh, slab, rows, rounds, lds);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every storing wave
    __syncthreads();
    if (threadIdx.x == 0) {
      if (MODE == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __hip_atomic_fetch_add(counter + mesh, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (threadIdx.x == 0) {
    int spins = 0;
    while (__hip_atomic_load(counter + mesh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (++spins > (1 << 22)) { atomicExch(err, 1); break; }   // bounded: never hangs
      __builtin_amdgcn_s_sleep(2);
    }
    if (MODE == 0) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  consume<MODE>(mid, out, mesh, slab, rows, rounds, lds);
}

int main(int argc, char** argv) {
  const int iters = 300;
  float4* mid; float* out; unsigned* counter; int* err;
  CK(hipMalloc(&mid, (size_t)B * 5120 * 4 * sizeof(float4)));
  CK(hipMalloc(&out, B * NSB * sizeof(float)));
  CK(hipMalloc(&counter, B * sizeof(unsigned)));
  CK(hipMalloc(&err, sizeof(int)));
  CK(hipMemset(err, 0, sizeof(int)));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipStream_t st; CK(hipStreamCreate(&st));
  for (int rows : {79, 313, 1250}) for (int rounds : {0, 6}) {
    float ms[3] = {0, 0, 0};
    std::vector<float> ref(B * NSB), got(B * NSB);
    int wrong[3] = {0, 0, 0};
    for (int variant = 0; variant < 3; ++variant) {
      CK(hipMemsetAsync(counter, 0, B * sizeof(unsigned), st));
      unsigned epoch = 0;
      auto once = [&]() {
        if (variant == 0) {
          hipLaunchKernelGGL(k_prod, dim3(B * NSA), dim3(T), 0, st, mid, rows, rounds);
          hipLaunchKernelGGL(k_cons, dim3(B * NSB), dim3(T), 0, st, (const float4*)mid, out, rows, rounds);
        } else {
          ++epoch;
          if (variant == 1) hipLaunchKernelGGL((k_fused<0>), dim3(B * NSB), dim3(T), 0, st, mid, out, counter, epoch * NSA, err, rows, rounds);
          else hipLaunchKernelGGL((k_fused<1>), dim3(B * NSB), dim3(T), 0, st, mid, out, counter, epoch * NSA, err, rows, rounds);
        }
      };
      for (int i = 0; i < 30; ++i) once();
      CK(hipStreamSynchronize(st));
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < iters; ++i) once();
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      CK(hipEventElapsedTime(&ms[variant], e0, e1));
      // the consumers' sums depend on every handed-off value: a stale read shows as a difference from the two-launch form
      CK(hipMemcpy(variant == 0 ? ref.data() : got.data(), out, B * NSB * sizeof(float), hipMemcpyDeviceToHost));
      if (variant > 0)
        for (int i = 0; i < B * NSB; ++i) wrong[variant] += got[i] != ref[i];
    }
    int herr = 0; CK(hipMemcpy(&herr, err, sizeof(int), hipMemcpyDeviceToHost));
    printf("rows %4d rounds %d: two launches %6.2f us | fused, fences %6.2f us | fused, sc1 %6.2f us | saving %5.2f / %5.2f us per boundary%s\n",
           rows, rounds, 1e3f * ms[0] / iters, 1e3f * ms[1] / iters, 1e3f * ms[2] / iters, 1e3f * (ms[0] - ms[1]) / iters,
           1e3f * (ms[0] - ms[2]) / iters, herr ? "  [POLL TIMED OUT]" : "");
    if (wrong[1] || wrong[2]) printf("    MISMATCH vs two launches: fences %d, sc1 %d of %d sums\n", wrong[1], wrong[2], B * NSB);
  }
  return 0;
}
