// Probe (gfx950): packed fp32 arithmetic IN PLACE with op_sel -- does the second (high) pass of v_pk_mul_f32 / v_pk_fma_f32 see the
// low result the first pass has just written when the destination pair is also a source pair and op_sel_hi picks that source's
// LOW half?  (The render kernel's round-2 irreproducibility disappeared with the compiler-formed packed ops; the forwarding
// probe tools/hw/pk_hazard.hip found nothing.  After the fp6 conversion's write-before-read overlap this is the analogous case.)
//   hipcc --offload-arch=gfx950 -O3 tools/hw/pk_inplace.hip -o pk_inplace && ./pk_inplace
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void probe(const float* __restrict__ in, int* __restrict__ bad) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const float a0 = in[(tid * 4) & 4095], a1 = in[(tid * 4 + 1) & 4095], b0 = in[(tid * 4 + 2) & 4095], b1 = in[(tid * 4 + 3) & 4095];
  float r0, r1;
  // 1: dst = src0 pair, high pass reads src0.LOW (op_sel_hi[0] = 0): expected {a0*b0, a0*b1}
  asm volatile("v_mov_b32 v10, %2\n\tv_mov_b32 v11, %3\n\tv_mov_b32 v12, %4\n\tv_mov_b32 v13, %5\n\ts_nop 4\n\t"
               "v_pk_mul_f32 v[10:11], v[10:11], v[12:13] op_sel_hi:[0,1]\n\ts_nop 4\n\tv_mov_b32 %0, v10\n\tv_mov_b32 %1, v11\n\t"
               : "=&v"(r0), "=&v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1) : "v10", "v11", "v12", "v13");
  if (r0 != a0 * b0) atomicAdd(bad + 0, 1);
  if (r1 != a0 * b1) atomicAdd(bad + 1, 1);
  // 2: dst = src0 pair, LOW pass reads src0.HIGH (op_sel[0] = 1): expected {a1*b0, a1*b1}
  asm volatile("v_mov_b32 v10, %2\n\tv_mov_b32 v11, %3\n\tv_mov_b32 v12, %4\n\tv_mov_b32 v13, %5\n\ts_nop 4\n\t"
               "v_pk_mul_f32 v[10:11], v[10:11], v[12:13] op_sel:[1,0] op_sel_hi:[1,1]\n\ts_nop 4\n\tv_mov_b32 %0, v10\n\tv_mov_b32 %1, v11\n\t"
               : "=&v"(r0), "=&v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1) : "v10", "v11", "v12", "v13");
  if (r0 != a1 * b0) atomicAdd(bad + 2, 1);
  if (r1 != a1 * b1) atomicAdd(bad + 3, 1);
  // 3: fma in place on the addend pair, high pass reads the addend's LOW half: expected {a0*b0 + a0, a1*b1 + a0}
  asm volatile("v_mov_b32 v10, %2\n\tv_mov_b32 v11, %3\n\tv_mov_b32 v12, %4\n\tv_mov_b32 v13, %5\n\tv_mov_b32 v14, %2\n\tv_mov_b32 v15, %3\n\ts_nop 4\n\t"
               "v_pk_fma_f32 v[10:11], v[14:15], v[12:13], v[10:11] op_sel_hi:[1,1,0]\n\ts_nop 4\n\tv_mov_b32 %0, v10\n\tv_mov_b32 %1, v11\n\t"
               : "=&v"(r0), "=&v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1) : "v10", "v11", "v12", "v13", "v14", "v15");
  if (r0 != __builtin_fmaf(a0, b0, a0)) atomicAdd(bad + 4, 1);
  if (r1 != __builtin_fmaf(a1, b1, a0)) atomicAdd(bad + 5, 1);
}

int main() {
  float h[4096];
  for (int i = 0; i < 4096; ++i) h[i] = (float)((i * 2654435761u) % 2001) / 250.0f - 4.0f;
  float* din; int* dbad;
  hipMalloc(&din, sizeof(h)); hipMalloc(&dbad, 8 * 4);
  hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
  hipMemset(dbad, 0, 32);
  const int threads = 256 * 4096;
  hipLaunchKernelGGL(probe, dim3(threads / 256), dim3(256), 0, 0, din, dbad);
  hipDeviceSynchronize();
  int hb[8];
  hipMemcpy(hb, dbad, 32, hipMemcpyDeviceToHost);
  printf("%d lanes.  v_pk_mul in place, op_sel_hi[0]=0: wrong lo %d hi %d;  op_sel[0]=1: wrong lo %d hi %d;  v_pk_fma addend in place, op_sel_hi[2]=0: wrong lo %d hi %d\n",
         threads, hb[0], hb[1], hb[2], hb[3], hb[4], hb[5]);
  return 0;
}
