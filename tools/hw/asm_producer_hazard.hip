// Probe (gfx950): the INLINE-ASM instructions of csrc/render_ls.hip as PRODUCERS -- is their result safe to read in the very next
// instruction?  (The compiler does not know what an asm statement executes: for the multi-pass fp6 conversion and the op_sel
// forms LLVM has forwarding hazards of its own -- hasCvtScaleForwardingHazard, hasDstSelForwardingHazard -- that it can only
// apply to instructions it selected.)  Consumers as they occur in x::store_block: v_mov_b32 / ds_write_b128 behind the
// conversion, the conversion behind v_fma_mix_f32.
//   hipcc --offload-arch=gfx950 -O3 tools/hw/asm_producer_hazard.hip -o p && ./p
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x6 __attribute__((ext_vector_type(6)));

#define LOAD_SRC                                                                                                  \
  "v_mov_b32 v64, %[a0]\n\tv_mov_b32 v65, %[a1]\n\tv_mov_b32 v66, %[a2]\n\tv_mov_b32 v67, %[a3]\n\t"               \
  "v_mov_b32 v68, %[a4]\n\tv_mov_b32 v69, %[a5]\n\tv_mov_b32 v70, %[a6]\n\tv_mov_b32 v71, %[a7]\n\t"               \
  "v_mov_b32 v72, %[a8]\n\tv_mov_b32 v73, %[a9]\n\tv_mov_b32 v74, %[a10]\n\tv_mov_b32 v75, %[a11]\n\t"             \
  "v_mov_b32 v76, %[a12]\n\tv_mov_b32 v77, %[a13]\n\tv_mov_b32 v78, %[a14]\n\tv_mov_b32 v79, %[a15]\n\t"           \
  "v_mov_b32 v80, %[b0]\n\tv_mov_b32 v81, %[b1]\n\tv_mov_b32 v82, %[b2]\n\tv_mov_b32 v83, %[b3]\n\t"               \
  "v_mov_b32 v84, %[b4]\n\tv_mov_b32 v85, %[b5]\n\tv_mov_b32 v86, %[b6]\n\tv_mov_b32 v87, %[b7]\n\t"               \
  "v_mov_b32 v88, %[b8]\n\tv_mov_b32 v89, %[b9]\n\tv_mov_b32 v90, %[b10]\n\tv_mov_b32 v91, %[b11]\n\t"             \
  "v_mov_b32 v92, %[b12]\n\tv_mov_b32 v93, %[b13]\n\tv_mov_b32 v94, %[b14]\n\tv_mov_b32 v95, %[b15]\n\t"           \
  "v_mov_b32 v102, %[s]\n\tv_mov_b32 v96, %[old]\n\tv_mov_b32 v101, %[old]\n\ts_nop 7\n\t"
#define SRC_OPS                                                                                                          \
  [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]), [a3] "v"(a[3]), [a4] "v"(a[4]), [a5] "v"(a[5]), [a6] "v"(a[6]),          \
  [a7] "v"(a[7]), [a8] "v"(a[8]), [a9] "v"(a[9]), [a10] "v"(a[10]), [a11] "v"(a[11]), [a12] "v"(a[12]), [a13] "v"(a[13]),  \
  [a14] "v"(a[14]), [a15] "v"(a[15]), [b0] "v"(b[0]), [b1] "v"(b[1]), [b2] "v"(b[2]), [b3] "v"(b[3]), [b4] "v"(b[4]),      \
  [b5] "v"(b[5]), [b6] "v"(b[6]), [b7] "v"(b[7]), [b8] "v"(b[8]), [b9] "v"(b[9]), [b10] "v"(b[10]), [b11] "v"(b[11]),      \
  [b12] "v"(b[12]), [b13] "v"(b[13]), [b14] "v"(b[14]), [b15] "v"(b[15]), [s] "v"(sc), [old] "v"(old)
#define CLOB "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", \
  "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100",    \
  "v101", "v102", "v103", "memory"

__global__ void probe(const float* __restrict__ in, float scale, int* __restrict__ bad) {
  __shared__ __attribute__((aligned(16))) int lds[512 * 4];
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  f32x16 a, b;
  for (int r = 0; r < 16; ++r) { a[r] = in[(tid * 37 + r) & 4095]; b[r] = in[(tid * 37 + 16 + r) & 4095]; }
  const float sc = scale;
  const int old = 0x5a5a5a5a;
  const i32x6 want = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a, b, sc);
  int o0, o5;
  // 0: conversion -> v_mov of its first and last destination register, no wait state
  asm volatile(LOAD_SRC "v_cvt_scalef32_2xpk16_fp6_f32 v[96:101], v[64:79], v[80:95], v102\n\tv_mov_b32 %[o0], v96\n\tv_mov_b32 %[o5], v101\n\t"
               : [o0] "=&v"(o0), [o5] "=&v"(o5) : SRC_OPS : CLOB);
  if (o0 != want[0]) atomicAdd(bad + 0, 1);
  if (o5 != want[5]) atomicAdd(bad + 1, 1);
  // 1: conversion -> ds_write_b128 of v[96:99], no wait state
  const uint32_t addr = (uint32_t)(uintptr_t)(lds + threadIdx.x * 4);
  asm volatile(LOAD_SRC "v_cvt_scalef32_2xpk16_fp6_f32 v[96:101], v[64:79], v[80:95], v102\n\tds_write_b128 %[ad], v[96:99]\n\ts_waitcnt lgkmcnt(0)\n\t"
               : : SRC_OPS, [ad] "v"(addr) : CLOB);
  __syncthreads();
  if (lds[threadIdx.x * 4] != want[0]) atomicAdd(bad + 2, 1);
  if (lds[threadIdx.x * 4 + 3] != want[3]) atomicAdd(bad + 3, 1);
  // 2: v_fma_mix_f32 (op_sel) -> conversion reading it as src0[0], no wait state
  f32x16 a2 = a;
  const uint32_t pk = 0x3c003800u;  // halves 0.5 (lo), 1.0 (hi)
  a2[0] = a[0] - 1.0f;              // what the fma_mix below computes with op_sel:[0,0,1]
  const i32x6 want2 = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a2, b, sc);
  asm volatile(LOAD_SRC "v_mov_b32 v103, %[pk]\n\ts_nop 4\n\tv_fma_mix_f32 v64, %[a0], 1.0, -v103 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
               "v_cvt_scalef32_2xpk16_fp6_f32 v[96:101], v[64:79], v[80:95], v102\n\ts_nop 7\n\tv_mov_b32 %[o0], v96\n\tv_mov_b32 %[o5], v101\n\t"
               : [o0] "=&v"(o0), [o5] "=&v"(o5) : SRC_OPS, [pk] "v"(pk) : CLOB);
  if (o0 != want2[0]) atomicAdd(bad + 4, 1);
  if (o5 != want2[5]) atomicAdd(bad + 5, 1);
}

int main() {
  float* in; int* bad;
  hipMalloc(&in, 4096 * 4); hipMalloc(&bad, 64 * 4);
  float h[4096];
  for (int i = 0; i < 4096; ++i) h[i] = 0.05f + 0.0007f * (float)((i * 2654435761u) % 4096);
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  hipMemset(bad, 0, 64 * 4);
  const int blocks = 2048, threads = 512;
  hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 0, 0, in, 0.25f, bad);
  int hb[64];
  hipMemcpy(hb, bad, sizeof(hb), hipMemcpyDeviceToHost);
  printf("%d lanes per variant, no wait state between producer and consumer\n", blocks * threads);
  printf("v_cvt_scalef32_2xpk16_fp6_f32 -> v_mov_b32      : first dword wrong %8d, last dword wrong %8d\n", hb[0], hb[1]);
  printf("v_cvt_scalef32_2xpk16_fp6_f32 -> ds_write_b128  : first dword wrong %8d, fourth dword wrong %8d\n", hb[2], hb[3]);
  printf("v_fma_mix_f32 (op_sel) -> v_cvt_scalef32 src0[0]: first dword wrong %8d, last dword wrong %8d\n", hb[4], hb[5]);
  return 0;
}
