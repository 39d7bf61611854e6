// Does v_mfma_f32_32x32x16_f16 keep f16 subnormal inputs?  (The hash-grid features of a freshly initialised model are
// +-1e-4, below the smallest normal f16 6.1e-5 for most of them.)
//   hipcc --offload-arch=gfx950 -O3 tools/hw/mfma_f16_denorm.hip -o /tmp/d && /tmp/d
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ void k(float a, float b, float* out) {
  f16x8 A, B;
  for (int e = 0; e < 8; ++e) { A[e] = (_Float16)a; B[e] = (_Float16)b; }
  f32x16 acc = {};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, acc, 0, 0, 0);
  if (threadIdx.x == 0) { out[0] = acc[0]; out[1] = (float)A[0]; }
}
int main() {
  float* d; (void)hipMalloc(&d, 8);
  const float vals[] = {1.0f, 1e-4f, 3e-5f, 9.5367431640625e-07f /* 2^-20 */, 5.9604644775390625e-08f /* 2^-24 */};
  for (float a : vals) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, 1.0f, d);
    float h[2]; (void)hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("a = %.9g (as f16 %.9g): sum over k=16 of a*1 = %.9g, expected %.9g\n", a, h[1], h[0], 16.0 * h[1]);
  }
  return 0;
}
