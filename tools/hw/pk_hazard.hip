// Micro-test for DESIGN 3b "reproducibility": is the result of a 32-bit VALU instruction visible to a packed-fp32
// instruction (v_pk_mul_f32 with op_sel, the splat the SLP vectoriser forms in the hash encoder's trilinear combine)
// issued right behind it, while the SIMD's other wave keeps the matrix pipe busy?
//   hipcc --offload-arch=gfx950 -O3 tools/hw/pk_hazard.hip -o gpurun_out/pk_hazard && gpurun_out/pk_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// 512 threads = 8 waves = 2 per SIMD.  Waves 0..3: the dependent pair in a loop.  Waves 4..7: MFMAs (or nothing).
// MODE 0: v_mul_f32 p ; v_pk_mul_f32 (q0,q1) = (a.hi, a.hi) * (p, p)     -- back to back, as in the kernel
// MODE 1: one s_nop 0 between the two
// MODE 2: v_mul_f32 p ; v_mul_f32 q0 = a.hi * p ; v_mul_f32 q1 = a.hi * p  -- no packed instruction
#define NA_BETWEEN_0 ""
#define NA_BETWEEN_1 "s_nop 0\n\t"
template <int MODE>
__global__ __launch_bounds__(512) void k(const float* __restrict__ in, unsigned* __restrict__ bad_per_lane, int iters, int with_mfma) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave >= 4) {
    if (!with_mfma) return;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(lane + e); b[e] = (__bf16)(float)(lane - e); }
    f32x16 acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
    for (int i = 0; i < iters; ++i) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2, 0, 0, 0);
      acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc3, 0, 0, 0);
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r] + acc2[r] + acc3[r];
    if (s == 12345.678f) bad_per_lane[0] = 1u << 30;  // keep the MFMAs alive
    return;
  }
  const int gid = (blockIdx.x * 4 + wave) * 64 + lane;
  const float x0 = in[gid];
  unsigned bad = 0;
  for (int i = 0; i < iters; ++i) {
    const float x = x0 + (float)(i & 15), y = 1.0f + 0.125f * (float)(i & 3), ahi = 3.0f + (float)(i & 1);
    float q0, q1;
    if constexpr (MODE == 2) {
      asm volatile(
          "v_mov_b32 v10, 0\n\t"
          "s_nop 7\n\t"
          "v_mul_f32 v10, %2, %3\n\t"
          "v_mul_f32 %0, %4, v10\n\t"
          "v_mul_f32 %1, %4, v10\n\t"
          : "=&v"(q0), "=&v"(q1) : "v"(x), "v"(y), "v"(ahi) : "v10");
    } else if constexpr (MODE == 0) {
#define NA_BETWEEN NA_BETWEEN_0
      asm volatile(
          "v_mov_b32 v10, 0\n\t"    // stale p
          "v_mov_b32 v11, 0\n\t"
          "v_mov_b32 v12, 0\n\t"
          "v_mov_b32 v13, %4\n\t"   // a.hi
          "s_nop 7\n\t"
          "v_mul_f32 v10, %2, %3\n\t"  // p = x * y
          NA_BETWEEN
          "v_pk_mul_f32 v[14:15], v[12:13], v[10:11] op_sel:[1,0] op_sel_hi:[1,0]\n\t"
          "s_nop 7\n\t"
          "v_mov_b32 %0, v14\n\t"
          "v_mov_b32 %1, v15\n\t"
          : "=&v"(q0), "=&v"(q1) : "v"(x), "v"(y), "v"(ahi) : "v10", "v11", "v12", "v13", "v14", "v15");
#undef NA_BETWEEN
    } else {
#define NA_BETWEEN NA_BETWEEN_1
      asm volatile(
          "v_mov_b32 v10, 0\n\t"    // stale p
          "v_mov_b32 v11, 0\n\t"
          "v_mov_b32 v12, 0\n\t"
          "v_mov_b32 v13, %4\n\t"   // a.hi
          "s_nop 7\n\t"
          "v_mul_f32 v10, %2, %3\n\t"  // p = x * y
          NA_BETWEEN
          "v_pk_mul_f32 v[14:15], v[12:13], v[10:11] op_sel:[1,0] op_sel_hi:[1,0]\n\t"
          "s_nop 7\n\t"
          "v_mov_b32 %0, v14\n\t"
          "v_mov_b32 %1, v15\n\t"
          : "=&v"(q0), "=&v"(q1) : "v"(x), "v"(y), "v"(ahi) : "v10", "v11", "v12", "v13", "v14", "v15");
#undef NA_BETWEEN
    }
    const float want = ahi * (x * y);
    if (q0 != want || q1 != want) ++bad;
  }
  if (bad) atomicAdd(&bad_per_lane[lane], bad);
}

template <int MODE>
static void run(const char* name, const float* in, unsigned* bad, int iters, int with_mfma) {
  hipMemset(bad, 0, 64 * 4);
  hipLaunchKernelGGL(k<MODE>, dim3(1024), dim3(512), 0, 0, in, bad, iters, with_mfma);
  hipDeviceSynchronize();
  unsigned h[64];
  hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost);
  unsigned long long tot = 0, q[4] = {0, 0, 0, 0};
  for (int l = 0; l < 64; ++l) { tot += h[l]; q[l / 16] += h[l]; }
  printf("%-34s mfma neighbour %d: mismatches %llu  (lanes 0-15: %llu, 16-31: %llu, 32-47: %llu, 48-63: %llu) of %.3g\n", name, with_mfma, tot,
         q[0], q[1], q[2], q[3], 1024.0 * 256 * iters);
}
int main() {
  const int n = 1024 * 256;
  std::vector<float> h(n);
  for (int i = 0; i < n; ++i) h[i] = 0.5f + (float)(i % 1013) * 0.001f;
  float* in; unsigned* bad;
  hipMalloc(&in, n * 4); hipMalloc(&bad, 64 * 4);
  hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 3; ++rep)
    for (int m = 0; m < 2; ++m) {
      run<0>("v_mul ; v_pk_mul (back to back)", in, bad, 20000, m);
      run<1>("v_mul ; s_nop 0 ; v_pk_mul", in, bad, 20000, m);
      run<2>("v_mul ; v_mul ; v_mul", in, bad, 20000, m);
    }
  return 0;
}
