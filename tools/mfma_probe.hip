#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void probe(float* out) {
  int l = threadIdx.x;
  float a = 100.f + l;      // A value of this lane
  float b = 1000.f * (l + 1);  // B value
  f4 c = {0, 0, 0, 0};
  f4 d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = d[r];
}
int main() {
  float* d; hipMalloc(&d, 256 * 4);
  probe<<<1, 64>>>(d);
  float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  // decode: value = a_lane * b_lane' ; find (la, lb) for each (lane, reg)
  for (int l = 0; l < 64; ++l) {
    if (l < 12 || l > 59) {
      printf("lane %2d:", l);
      for (int r = 0; r < 4; ++r) {
        double v = h[l * 4 + r]; int fa = -1, fb = -1;
        for (int la = 0; la < 64 && fa < 0; ++la) for (int lb = 0; lb < 64; ++lb)
          if (fabs((100.0 + la) * 1000.0 * (lb + 1) - v) < 0.5) { fa = la; fb = lb; break; }
        printf("  r%d=A[l%d]*B[l%d]", r, fa, fb);
      }
      printf("\n");
    }
  }
  return 0;
}
