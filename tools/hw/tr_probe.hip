// Probe of ds_read_b64_tr_b16 (gfx950, tools/hw): every lane supplies its own 8-byte-aligned LDS address; which 16-bit elements
// come back?  LDS holds its own halfword index; lane l reads at byte address addr[l]; the four returned halfwords are printed.
//   hipcc --offload-arch=gfx950 tools/hw/tr_probe.hip -o /tmp/tr_probe && /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void probe(const int* addr, v4s* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  typedef __attribute__((address_space(3))) v4s* lp;
  out[threadIdx.x] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)((__attribute__((address_space(3))) char*)lds + addr[threadIdx.x]));
}
static void run(const char* what, const int* a) {
  int* d; v4s* o; v4s h[64];
  hipMalloc(&d, 256); hipMalloc(&o, 64 * 8);
  hipMemcpy(d, a, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o);
  hipMemcpy(h, o, 64 * 8, hipMemcpyDeviceToHost);
  printf("%s\n", what);
  for (int l = 0; l < 64; ++l) {
    printf("  lane %2d addr %5d (halfword %4d): %4d %4d %4d %4d\n", l, a[l], a[l] / 2, (unsigned short)h[l][0], (unsigned short)h[l][1], (unsigned short)h[l][2], (unsigned short)h[l][3]);
    if (l == 17 || l == 33 || l == 49) { printf("  ...\n"); l += 13; }
  }
  hipFree(d); hipFree(o);
}
int main() {
  int a[64];
  for (int l = 0; l < 64; ++l) a[l] = 8 * l;                     // contiguous: lane l at halfwords 4 l .. 4 l + 3
  run("A: lane l at byte 8 l", a);
  for (int l = 0; l < 64; ++l) a[l] = 1000 * (l & 15) + 4096 * (l >> 4) / 2 * 2 - (1000 * (l & 15)) % 8;  // scattered, 8-aligned
  run("B: scattered 8-byte-aligned addresses", a);
  // C: the layout the weight-gradient kernel wants: group g = l >> 4, lane i = l & 15 points at row (i >> 2), 4 columns (i & 3):
  // byte = row * 576 + 8 * (i & 3) + 32 * (g & 1) + 8 * 576 * (g >> 1)
  for (int l = 0; l < 64; ++l) { const int i = l & 15, g = l >> 4; a[l] = (i >> 2) * 576 + 8 * (i & 3) + 32 * (g & 1) + 8 * 576 * (g >> 1); }
  run("C: [4 rows][16 columns] blocks, row pitch 576 B", a);
  return 0;
}
