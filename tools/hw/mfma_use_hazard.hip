// Probe (gfx950): a VALU instruction reading the destination of an MFMA W wait states after it.  The hardware does not interlock
// this dependency; LLVM's hazard recogniser inserts s_nop for the instructions it selects (12 wait states behind the 8-pass
// v_mfma_f32_32x32x16_f16 in every listing of this library) but is blind to INLINE ASSEMBLY, and a non-volatile asm statement
// can be scheduled across a barrier right behind the MFMA that produces its operand.  Found in round 4 (csrc/render_ls.hip,
// x::store_block's asm v_max3_f32 on the accumulators of first.out, 3 wait states behind the last MFMA in one instance of the
// mip renderer): last-bit run-to-run differences.  build.check_mfma_use scans every listing for it.
//   hipcc --offload-arch=gfx950 -O3 tools/hw/mfma_use_hazard.hip -o mfma_use_hazard && ./mfma_use_hazard
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define ZERO "v_mov_b32 v64, %[z]\n\tv_mov_b32 v65, %[z]\n\tv_mov_b32 v66, %[z]\n\tv_mov_b32 v67, %[z]\n\tv_mov_b32 v68, %[z]\n\tv_mov_b32 v69, %[z]\n\t" \
             "v_mov_b32 v70, %[z]\n\tv_mov_b32 v71, %[z]\n\tv_mov_b32 v72, %[z]\n\tv_mov_b32 v73, %[z]\n\tv_mov_b32 v74, %[z]\n\tv_mov_b32 v75, %[z]\n\t" \
             "v_mov_b32 v76, %[z]\n\tv_mov_b32 v77, %[z]\n\tv_mov_b32 v78, %[z]\n\tv_mov_b32 v79, %[z]\n\ts_nop 7\n\ts_nop 7\n\t"
#define CLOB "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79"
#define VARIANT(ID, NOPS)                                                                                                  \
  {                                                                                                                        \
    float o0, o1;                                                                                                          \
    asm volatile(ZERO "v_mfma_f32_32x32x16_f16 v[64:79], %[a], %[b], v[64:79]\n\t" NOPS                                    \
                 "v_mov_b32 %[o0], v64\n\tv_mov_b32 %[o1], v79\n\t"                                                        \
                 : [o0] "=&v"(o0), [o1] "=&v"(o1) : [a] "v"(a), [b] "v"(b), [z] "v"(z) : CLOB);                            \
    if (o0 != 16.0f) atomicAdd(bad + 2 * (ID), 1);                                                                         \
    if (o1 != 16.0f) atomicAdd(bad + 2 * (ID) + 1, 1);                                                                     \
  }

__global__ void probe(int* __restrict__ bad) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)1.0f; b[i] = (_Float16)1.0f; }
  const float z = 0.f;
  VARIANT(0, "")
  VARIANT(1, "s_nop 1\n\t")
  VARIANT(2, "s_nop 3\n\t")
  VARIANT(3, "s_nop 5\n\t")
  VARIANT(4, "s_nop 7\n\t")
  VARIANT(5, "s_nop 7\n\ts_nop 1\n\t")
  VARIANT(6, "s_nop 7\n\ts_nop 3\n\t")
  VARIANT(7, "s_nop 7\n\ts_nop 7\n\t")
}

int main() {
  int* bad;
  hipMalloc(&bad, 64 * 4);
  hipMemset(bad, 0, 64 * 4);
  const int blocks = 2048, threads = 512;
  hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 0, 0, bad);
  int hb[64];
  hipMemcpy(hb, bad, sizeof(hb), hipMemcpyDeviceToHost);
  const int ws[8] = {0, 2, 4, 6, 8, 10, 12, 16};
  printf("v_mfma_f32_32x32x16_f16 (8 passes) -> v_mov_b32 of its destination, %d lanes per variant\n", blocks * threads);
  for (int id = 0; id < 8; ++id)
    printf("wait states %2d: first register stale in %8d lanes, last register stale in %8d lanes\n", ws[id], hb[2 * id], hb[2 * id + 1]);
  return 0;
}
