// Probe (gfx950): a transcendental VALU result (v_sin_f32 / v_exp_f32 / v_rcp_f32) read by the NEXT non-transcendental VALU
// instruction.  CDNA3/4 need one wait state there ("trans forwarding hazard"); LLVM's hazard recogniser inserts it for the
// instructions it selects itself (GCNHazardRecognizer::checkVALUHazards, hasTransForwardingHazard) but does not look at the
// operands of INLINE ASSEMBLY, so an `asm("v_fma_mix_f32 ...")` / `asm("v_max3_f32 ...")` / asm fp6 conversion that the
// scheduler places directly behind the trans op that produces its operand reads the register's OLD contents.
// Found in round 4: csrc/render_ls.hip x::store_block<NA_ACT_SIN> (v_sin_f32 activations feeding asm v_fma_mix_f32 and the asm
// fp6 conversion) made one instance of the mip renderer (MODEL 6) differ from run to run in the last bit of a few pixels.
//   hipcc --offload-arch=gfx950 -O3 tools/hw/trans_use_hazard.hip -o trans_use_hazard && ./trans_use_hazard
// prints, per (trans op, consumer, wait states), how many of the lanes read a stale operand.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define VARIANT(ID, TRANS, NOPS, CONSUMER)                                                                      \
  {                                                                                                             \
    float o;                                                                                                    \
    asm volatile("v_mov_b32 v70, %[old]\n\ts_nop 7\n\t" TRANS " v70, %[x]\n\t" NOPS CONSUMER                    \
                 : [o] "=&v"(o) : [x] "v"(x), [old] "v"(old), [pk] "v"(pk) : "v70");                             \
    float t;                                                                                                    \
    asm volatile(TRANS " v71, %[x]\n\ts_nop 7\n\tv_mov_b32 %[t], v71" : [t] "=&v"(t) : [x] "v"(x) : "v71");      \
    float want_new, want_old;                                                                                   \
    want_new = ref(t, pk, ID);                                                                                  \
    want_old = ref(old, pk, ID);                                                                                \
    if (o != want_new) atomicAdd(bad + 2 * (ID), 1);                                                            \
    if (o != want_new && o == want_old) atomicAdd(bad + 2 * (ID) + 1, 1);                                       \
  }

__device__ float ref(float v, uint32_t pk, int id) {
  const int c = id % 3;
  if (c == 0) {  // v_fma_mix_f32 o, v, 1.0, -half(pk.lo)
    const _Float16 h = __builtin_bit_cast(_Float16, (uint16_t)(pk & 0xffff));
    return v - (float)h;
  }
  if (c == 1) return fmaxf(fmaxf(0.f, fabsf(v)), fabsf(v));  // v_max3_f32 o, 0, |v|, |v|
  return v;                                                    // v_mov_b32
}

#define FMA "v_fma_mix_f32 %[o], v70, 1.0, -%[pk] op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
#define MAX3 "v_mov_b32 %[o], 0\n\t"  /* placeholder replaced below: keeps the trans op adjacent to ITS consumer */
#define MOV "v_mov_b32 %[o], v70\n\t"

__global__ void probe(const float* __restrict__ in, int* __restrict__ bad) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const float x = in[tid & 4095] * 0.15f, old = 1234.5f + (float)(tid & 63);
  const uint32_t pk = 0x3c003800u;  // halves 0.5 (lo), 1.0 (hi)
  // ids: 3 * (trans * 2 + waits) + consumer;  trans 0 sin, 1 exp, 2 rcp;  waits 0 | 1 (s_nop 0)
  VARIANT(0, "v_sin_f32", "", FMA)
  VARIANT(1, "v_sin_f32", "", "v_max3_f32 %[o], 0, |v70|, |v70|\n\t")
  VARIANT(2, "v_sin_f32", "", MOV)
  VARIANT(3, "v_sin_f32", "s_nop 0\n\t", FMA)
  VARIANT(4, "v_sin_f32", "s_nop 0\n\t", "v_max3_f32 %[o], 0, |v70|, |v70|\n\t")
  VARIANT(5, "v_sin_f32", "s_nop 0\n\t", MOV)
  VARIANT(6, "v_exp_f32", "", FMA)
  VARIANT(7, "v_exp_f32", "", "v_max3_f32 %[o], 0, |v70|, |v70|\n\t")
  VARIANT(8, "v_exp_f32", "", MOV)
  VARIANT(9, "v_exp_f32", "s_nop 0\n\t", FMA)
  VARIANT(10, "v_exp_f32", "s_nop 0\n\t", "v_max3_f32 %[o], 0, |v70|, |v70|\n\t")
  VARIANT(11, "v_exp_f32", "s_nop 0\n\t", MOV)
  VARIANT(12, "v_rcp_f32", "", FMA)
  VARIANT(13, "v_rcp_f32", "", "v_max3_f32 %[o], 0, |v70|, |v70|\n\t")
  VARIANT(14, "v_rcp_f32", "", MOV)
  VARIANT(15, "v_rcp_f32", "s_nop 0\n\t", FMA)
  VARIANT(16, "v_rcp_f32", "s_nop 0\n\t", "v_max3_f32 %[o], 0, |v70|, |v70|\n\t")
  VARIANT(17, "v_rcp_f32", "s_nop 0\n\t", MOV)
}

int main() {
  float* in; int* bad;
  hipMalloc(&in, 4096 * 4); hipMalloc(&bad, 64 * 4);
  float h[4096];
  for (int i = 0; i < 4096; ++i) h[i] = 0.37f + 0.001f * (float)((i * 2654435761u) % 4096);
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  hipMemset(bad, 0, 64 * 4);
  const int blocks = 2048, threads = 512;
  hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 0, 0, in, bad);
  int hb[64];
  hipMemcpy(hb, bad, sizeof(hb), hipMemcpyDeviceToHost);
  const char* tn[3] = {"v_sin_f32", "v_exp_f32", "v_rcp_f32"};
  const char* cn[3] = {"v_fma_mix_f32", "v_max3_f32", "v_mov_b32"};
  printf("%d lanes per variant\n", blocks * threads);
  for (int id = 0; id < 18; ++id)
    printf("%-10s -> %-14s wait states %d: %8d wrong, %8d of them = the register's OLD value\n", tn[id / 6], cn[id % 3], (id / 3) % 2,
           hb[2 * id], hb[2 * id + 1]);
  return 0;
}
