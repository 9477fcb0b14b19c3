// bf16 STORAGE helpers (BASELINE configs[1] "bf16": activations live in HBM as bf16, every kernel converts at its
// loads / stores and computes and accumulates in fp32).  Tensors keep their [B][N][C] layout, C innermost, two bytes
// per element; a group of 4 channels is one 8-byte word, 8 channels one 16-byte word.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace mvh {

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

// round-to-nearest-even, NaN stays NaN (v_cvt_pk_bf16_f32 on gfx950); `a` lands in the low half
__device__ __forceinline__ uint32_t bf16_pack2(float a, float b) {
  const bf16x2_t r = __builtin_convertvector((f32x2_t){a, b}, bf16x2_t);
  return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ float bf16_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ float4 bf16_unpack4(uint2 w) {
  return make_float4(bf16_lo(w.x), bf16_hi(w.x), bf16_lo(w.y), bf16_hi(w.y));
}
__device__ __forceinline__ uint2 bf16_pack4(float a, float b, float c, float d) {
  return make_uint2(bf16_pack2(a, b), bf16_pack2(c, d));
}

// element `idx` .. idx+3 of a tensor that is fp32 or bf16 (idx % 4 == 0, base 16-byte aligned)
__device__ __forceinline__ float4 load4_any(const float* base, long long idx, bool is_bf16) {
  if (is_bf16) return bf16_unpack4(*reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(base) + idx));
  return *reinterpret_cast<const float4*>(base + idx);
}
__device__ __forceinline__ void store4_any(float* base, long long idx, bool is_bf16, float a, float b, float c, float d) {
  if (is_bf16) *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(base) + idx) = bf16_pack4(a, b, c, d);
  else *reinterpret_cast<float4*>(base + idx) = make_float4(a, b, c, d);
}
__device__ __forceinline__ float load1_any(const float* base, long long idx, bool is_bf16) {
  if (is_bf16) return __uint_as_float((uint32_t)reinterpret_cast<const uint16_t*>(base)[idx] << 16);
  return base[idx];
}

__host__ __device__ inline int l0h_pack_dwords_hd(int K) { return 4 * K * 32; }
// dword i of the bf16 weight slabs of the level-0 matrix-pipe kernel (cheb_l0h.hip): Wh[slab][k][cg][i][d], one
// dword = the two input channels 4 cg + 2 d (low half) and + 1 (high half) of output channel 4 slab + i --
// W[k][c][o] forward, W[k][o][c] backward (W^T); 16 channels on both sides.
__device__ __forceinline__ uint32_t pack_l0h_dword(const float* __restrict__ W, int K, int bwd, int i) {
  const int d = i & 1, oi = (i >> 1) & 3, cg = (i >> 3) & 3, k = (i >> 5) % K, sl = (i >> 5) / K;
  const int o = sl * 4 + oi, c = cg * 4 + 2 * d;
  const float w0 = bwd ? W[((long long)k * 16 + o) * 16 + c] : W[((long long)k * 16 + c) * 16 + o];
  const float w1 = bwd ? W[((long long)k * 16 + o) * 16 + c + 1] : W[((long long)k * 16 + c + 1) * 16 + o];
  return bf16_pack2(w0, w1);
}

}  // namespace mvh
