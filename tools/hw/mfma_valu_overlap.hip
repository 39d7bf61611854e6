// Do the VALU instructions of one wave overlap the MFMAs of ANOTHER wave on the same SIMD?  (tools/hw, gfx950)
// One workgroup of 8 waves on one CU = 2 waves per SIMD.  Waves 0-3 (one per SIMD) issue NM back-to-back
// v_mfma_f32_32x32x16_bf16 on four independent accumulators; waves 4-7 (their SIMD partners) issue NV independent v_fma_f32 on
// eight registers.  Timed with s_memtime for: MFMA waves alone, VALU waves alone, both together.
//   overlap    -> both ~ max(alone_mfma, alone_valu)
//   serialised -> both ~ alone_mfma + alone_valu
//   hipcc --offload-arch=gfx950 -O3 tools/hw/mfma_valu_overlap.hip -o /tmp/mvo && /tmp/mvo
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// LDS: 0 pure MFMA / pure VALU; 1 the MFMA waves take their B operands from LDS (4 ds_read_b128 per 6 MFMAs, like a GEMM consumer)
// and the VALU waves store to LDS (one ds_write_b64 per 4 v_fma, like a conversion loop)
template <int MODE, int LDS = 0>  // MODE: 1 MFMA waves only, 2 VALU waves only, 3 both
__global__ __launch_bounds__(512) void probe(int nm, int nv, float* sink, long long* cycles) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __shared__ __attribute__((aligned(16))) char lds[65536];
  for (int i = threadIdx.x; i < 16384; i += 512) ((float*)lds)[i] = 1.0f / (1 + (i & 15));
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  if (wave < 4) {
    if (MODE & 1) {
      bf16x8 a, b;
      for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(lane + e); b[e] = (__bf16)(float)(lane - e); }
      f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
      if (LDS) {
        const char* base = lds + wave * 8192 + lane * 16;
        for (int i = 0; i < nm; i += 6) {
          const bf16x8 b0 = *(const bf16x8*)(base + (i & 3) * 1024), b1 = *(const bf16x8*)(base + 4096 + (i & 3) * 1024);
          const bf16x8 b2 = *(const bf16x8*)(base + 2048 + (i & 1) * 1024), b3 = *(const bf16x8*)(base + 6144 + (i & 1) * 1024);
          c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b0, c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b1, c1, 0, 0, 0);
          c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b2, c2, 0, 0, 0);
          c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b3, c3, 0, 0, 0);
          c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, b1, c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b2, b3, c1, 0, 0, 0);
        }
      } else
      for (int i = 0; i < nm; i += 4) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
      }
      sink[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
    }
  } else {
    if (MODE & 2) {
      float r0 = lane, r1 = lane + 1, r2 = lane + 2, r3 = lane + 3, r4 = lane + 4, r5 = lane + 5, r6 = lane + 6, r7 = lane + 7;
      const float k = 1.0001f, m = 0.5f;
      char* wbase = lds + 32768 + (wave - 4) * 8192 + lane * 8;
      for (int i = 0; i < nv; i += 8) {
        if (LDS) {
          asm volatile("ds_write_b64 %0, %1\n ds_write_b64 %0, %2 offset:4096" :: "v"((uint32_t)(uintptr_t)(wbase + (i & 7) * 512)),
                       "v"(make_float2(r0, r1)), "v"(make_float2(r2, r3)) : "memory");
        }
        asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                     "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                     : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(k), "v"(m));
      }
      sink[threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cycles[wave] = t1 - t0;
}

template <int MODE, int LDS = 0>
static void run(const char* what, int nm, int nv) {
  float* sink; long long* cyc; long long h[8];
  (void)hipMalloc(&sink, 512 * 4); (void)hipMalloc(&cyc, 64);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe<MODE, LDS>), dim3(1), dim3(512), 0, 0, nm, nv, sink, cyc);
  (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  long long mm = 0, vv = 0;
  for (int w = 0; w < 4; ++w) { if (h[w] > mm) mm = h[w]; if (h[4 + w] > vv) vv = h[4 + w]; }
  printf("%-28s MFMA waves %8lld cycles (%.1f per MFMA)   VALU waves %8lld cycles (%.2f per v_fma)\n", what, mm, (double)mm / nm, vv,
         (double)vv / nv);
  (void)hipFree(sink); (void)hipFree(cyc);
}
int main() {
  const int nm = 4096;
  for (int nv : {4096, 16384, 32768}) {
    printf("--- %d MFMAs (32x32x16 bf16) per MFMA wave, %d v_fma_f32 per VALU wave\n", nm, nv);
    run<1>("MFMA waves alone", nm, nv);
    run<2>("VALU waves alone", nm, nv);
    run<3>("both (2 waves per SIMD)", nm, nv);
  }
  printf("--- with LDS traffic: MFMA waves read their B operands (4 ds_read_b128 per 6 MFMAs), VALU waves store (2 ds_write_b64 per 8 v_fma)\n");
  run<1, 1>("MFMA waves alone", 4098, 16384);
  run<2, 1>("VALU waves alone", 4098, 16384);
  run<3, 1>("both (2 waves per SIMD)", 4098, 16384);
  return 0;
}
