// Probe (gfx950): do 16-byte raw-buffer loads / stores work at 4-byte alignment, and is the range check per dword?
//   hipcc --offload-arch=gfx950 -O3 tools/hw/unaligned_probe.hip -o unaligned_probe && ./unaligned_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const float* src, int nrec_bytes, float* out, float* dst, int dst_bytes) {
  const int i = threadIdx.x;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nrec_bytes, 0x00020000);
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, i * 4, 0, 0);
  *(f32x4*)(out + 4 * i) = __builtin_bit_cast(f32x4, v);
  // stores: lane i (i < 32) writes 4 floats at byte offset 20 i + 4 (never 16-byte aligned in general)
  const __amdgpu_buffer_rsrc_t w = __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, dst_bytes, 0x00020000);
  if (i < 32) {
    const u32x4 s = {__builtin_bit_cast(uint32_t, 1000.f + i), __builtin_bit_cast(uint32_t, 2000.f + i), __builtin_bit_cast(uint32_t, 3000.f + i), __builtin_bit_cast(uint32_t, 4000.f + i)};
    __builtin_amdgcn_raw_buffer_store_b128(s, w, 20 * i + 4, 0, 0);
  }
}

int main() {
  const int N = 100, T = 128;
  float h[N + 64], *d, *o, *dst, ho[4 * T], hd[256];
  for (int i = 0; i < N + 64; ++i) h[i] = 1.f + i;
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(ho)); hipMalloc(&dst, sizeof(hd));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipMemset(dst, 0, sizeof(hd));
  const int dst_bytes = 20 * 31 + 4 + 8;  // the last lane's store has 2 dwords inside, 2 outside
  hipLaunchKernelGGL(probe, dim3(1), dim3(T), 0, 0, d, N * 4, o, dst, dst_bytes);
  hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
  hipMemcpy(hd, dst, sizeof(hd), hipMemcpyDeviceToHost);
  int bad_in = 0, bad_oob = 0, partial_ok = 0, partial_zeroed = 0;
  for (int i = 0; i < T; ++i)
    for (int e = 0; e < 4; ++e) {
      const float got = ho[4 * i + e];
      if (i + e < N) { if (got != h[i + e]) ++bad_in; }
      else if (got != 0.f) ++bad_oob;
    }
  for (int i = N - 3; i < N; ++i) {  // lanes whose 16 bytes straddle the end of the buffer
    bool inside = true;
    for (int e = 0; e < 4 && i + e < N; ++e) inside = inside && ho[4 * i + e] == h[i + e];
    if (inside) ++partial_ok; else ++partial_zeroed;
  }
  printf("loads : in-range mismatches %d, out-of-range non-zeros %d; straddling lanes with their in-range dwords intact %d, zeroed %d\n", bad_in, bad_oob, partial_ok, partial_zeroed);
  for (int i = 0; i < 10; ++i) printf("  lane %d (byte offset %d): %.0f %.0f %.0f %.0f\n", i, 4 * i, ho[4 * i], ho[4 * i + 1], ho[4 * i + 2], ho[4 * i + 3]);
  for (int i = 94; i < 104; ++i) printf("  lane %d (byte offset %d): %.0f %.0f %.0f %.0f\n", i, 4 * i, ho[4 * i], ho[4 * i + 1], ho[4 * i + 2], ho[4 * i + 3]);
  int sbad = 0;
  for (int i = 0; i < 31; ++i)
    for (int e = 0; e < 4; ++e) if (hd[5 * i + 1 + e] != 1000.f * (e + 1) + i) ++sbad;
  printf("stores: mismatches in lanes 0..30 %d; last lane (2 dwords inside): %.0f %.0f | beyond: %.0f %.0f\n", sbad, hd[5 * 31 + 1], hd[5 * 31 + 2], hd[5 * 31 + 3], hd[5 * 31 + 4]);
  return 0;
}
