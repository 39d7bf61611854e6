// Probe (gfx950): do PACKED fp32 consumers (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32, what the SLP vectoriser forms) need more
// wait states behind a transcendental op or an MFMA than the scalar ones?  (Round 2's run-to-run events went away with
// -fno-slp-vectorize; LLVM gives packed consumers the same 1 / passes + 4 wait states -- is that enough for the hardware?)
//   hipcc --offload-arch=gfx950 -O3 tools/hw/pk_after_trans_mfma.hip -o pk_after_trans_mfma && ./pk_after_trans_mfma
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define TV(ID, NOPS, CONS, WANT)                                                                                          \
  {                                                                                                                        \
    float o0, o1;                                                                                                          \
    asm volatile("v_mov_b32 v70, %[old]\n\tv_mov_b32 v71, %[old]\n\tv_mov_b32 v72, %[two]\n\tv_mov_b32 v73, %[two]\n\ts_nop 7\n\t" \
                 "v_exp_f32 v70, %[x]\n\tv_exp_f32 v71, %[x]\n\t" NOPS CONS                                                \
                 "s_nop 7\n\tv_mov_b32 %[o0], v74\n\tv_mov_b32 %[o1], v75\n\t"                                             \
                 : [o0] "=&v"(o0), [o1] "=&v"(o1) : [x] "v"(x), [old] "v"(old), [two] "v"(two) : "v70", "v71", "v72", "v73", "v74", "v75"); \
    if (o0 != (WANT)) atomicAdd(bad + 2 * (ID), 1);                                                                        \
    if (o1 != (WANT)) atomicAdd(bad + 2 * (ID) + 1, 1);                                                                    \
  }
#define ZERO "v_mov_b32 v64, %[z]\n\tv_mov_b32 v65, %[z]\n\tv_mov_b32 v66, %[z]\n\tv_mov_b32 v67, %[z]\n\tv_mov_b32 v68, %[z]\n\tv_mov_b32 v69, %[z]\n\t" \
             "v_mov_b32 v70, %[z]\n\tv_mov_b32 v71, %[z]\n\tv_mov_b32 v72, %[z]\n\tv_mov_b32 v73, %[z]\n\tv_mov_b32 v74, %[z]\n\tv_mov_b32 v75, %[z]\n\t" \
             "v_mov_b32 v76, %[z]\n\tv_mov_b32 v77, %[z]\n\tv_mov_b32 v78, %[z]\n\tv_mov_b32 v79, %[z]\n\tv_mov_b32 v80, %[two]\n\tv_mov_b32 v81, %[two]\n\ts_nop 7\n\ts_nop 7\n\t"
#define MV(ID, NOPS)                                                                                                      \
  {                                                                                                                        \
    float o0, o1;                                                                                                          \
    asm volatile(ZERO "v_mfma_f32_32x32x16_f16 v[64:79], %[a], %[b], v[64:79]\n\t" NOPS                                    \
                 "v_pk_mul_f32 v[82:83], v[78:79], v[80:81]\n\ts_nop 7\n\tv_mov_b32 %[o0], v82\n\tv_mov_b32 %[o1], v83\n\t" \
                 : [o0] "=&v"(o0), [o1] "=&v"(o1) : [a] "v"(a), [b] "v"(b), [z] "v"(z), [two] "v"(two)                     \
                 : "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83"); \
    if (o0 != 32.0f) atomicAdd(bad + 2 * (ID), 1);                                                                         \
    if (o1 != 32.0f) atomicAdd(bad + 2 * (ID) + 1, 1);                                                                     \
  }

__global__ void probe(int* __restrict__ bad) {
  const float x = 3.0f, old = 1234.5f, two = 2.0f, z = 0.f;  // v_exp_f32(3) = 8
  TV(0, "", "v_pk_mul_f32 v[74:75], v[70:71], v[72:73]\n\t", 16.0f)
  TV(1, "s_nop 0\n\t", "v_pk_mul_f32 v[74:75], v[70:71], v[72:73]\n\t", 16.0f)
  TV(2, "s_nop 1\n\t", "v_pk_mul_f32 v[74:75], v[70:71], v[72:73]\n\t", 16.0f)
  TV(3, "", "v_pk_add_f32 v[74:75], v[70:71], v[72:73]\n\t", 10.0f)
  TV(4, "s_nop 0\n\t", "v_pk_add_f32 v[74:75], v[70:71], v[72:73]\n\t", 10.0f)
  TV(5, "s_nop 0\n\t", "v_pk_fma_f32 v[74:75], v[70:71], v[72:73], v[72:73]\n\t", 18.0f)
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)1.0f; b[i] = (_Float16)1.0f; }
  MV(8, "s_nop 7\n\ts_nop 1\n\t")   // 10
  MV(9, "s_nop 7\n\ts_nop 2\n\t")   // 11
  MV(10, "s_nop 7\n\ts_nop 3\n\t")  // 12: LLVM's number
  MV(11, "s_nop 7\n\ts_nop 4\n\t")  // 13
}

int main() {
  int* bad;
  hipMalloc(&bad, 64 * 4);
  hipMemset(bad, 0, 64 * 4);
  const int blocks = 2048, threads = 512;
  hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 0, 0, bad);
  int hb[64];
  hipMemcpy(hb, bad, sizeof(hb), hipMemcpyDeviceToHost);
  const char* nm[12] = {"v_exp -> v_pk_mul, 0 wait states", "v_exp -> v_pk_mul, 1", "v_exp -> v_pk_mul, 2", "v_exp -> v_pk_add, 0", "v_exp -> v_pk_add, 1",
                        "v_exp -> v_pk_fma, 1", "", "", "mfma -> v_pk_mul (last regs), 10", "mfma -> v_pk_mul, 11", "mfma -> v_pk_mul, 12 (LLVM)", "mfma -> v_pk_mul, 13"};
  printf("%d lanes per variant\n", blocks * threads);
  for (int id = 0; id < 12; ++id)
    if (nm[id][0]) printf("%-36s: low half wrong in %8d lanes, high half wrong in %8d lanes\n", nm[id], hb[2 * id], hb[2 * id + 1]);
  return 0;
}
