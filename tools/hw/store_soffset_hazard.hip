// Probe (gfx950): a 16-byte buffer store whose data registers are overwritten by the NEXT VALU instruction(s).
// The store reads its data VGPRs after it has issued.  LLVM's hazard recogniser (GCNHazardRecognizer::createsVALUHazard /
// checkVALUHazards, gfx940+: 2 wait states) inserts the wait for stores of more than 8 bytes -- but only when the soffset field
// is NOT an SGPR; with an SGPR soffset it assumes there is no hazard and lets `v_lshlrev_b32 v182, ...` follow
// `buffer_store_dwordx4 v[182:185], v176, s[20:23], s33 offen` directly.
// Found in round 5: the g_x stores of csrc/train_bwd.hip (first build: row step in the soffset) wrote a just-converted bf16
// instead of the gradient in lanes 12-15 / 28-31 of some waves -- 272 of 2 M elements, timing dependent, one activation only
// (the other instantiations had the same instruction pairs and got away with it).
//   hipcc --offload-arch=gfx950 -O3 tools/hw/store_soffset_hazard.hip -o store_soffset_hazard && ./store_soffset_hazard
// prints, per (soffset kind, wait states between the store and the overwrite), how many stored dwords were the NEW value.
// The store is issued while other waves of the CU keep the vector memory path busy (the hazard needs the data read to be late).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define NV 8
// data in v[20:23] = good; store; NOPS; overwrite v20..v23 with bad
#define VARIANT(ID, SOFF, NOPS)                                                                                      \
  {                                                                                                                  \
    const uint32_t vo = (uint32_t)(((ID) * total + tid) * 16);                                                       \
    asm volatile("v_mov_b32 v20, %[g]\n\tv_mov_b32 v21, %[g]\n\tv_mov_b32 v22, %[g]\n\tv_mov_b32 v23, %[g]\n\ts_nop 7\n\t" \
                 "buffer_store_dwordx4 v[20:23], %[vo], %[rs], " SOFF " offen\n\t" NOPS                              \
                 "v_mov_b32 v20, %[b]\n\tv_mov_b32 v21, %[b]\n\tv_mov_b32 v22, %[b]\n\tv_mov_b32 v23, %[b]\n\t"      \
                 : : [g] "v"(good), [b] "v"(bad), [vo] "v"(vo), [rs] "s"(rs), [so] "s"(so) : "v20", "v21", "v22", "v23", "memory"); \
  }

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void probe(uint32_t* __restrict__ out, const u32x4* __restrict__ noise, u32x4* __restrict__ sink, int total, int so_in) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t good = 0x600d0000u + (uint32_t)(tid & 0xffff), bad = 0xbad00000u + (uint32_t)(tid & 0xffff);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, 0x7fffffff, 0x00020000);
  const int so = __builtin_amdgcn_readfirstlane(so_in);  // (an SGPR holding zero: the address is the same in every variant)
  // half of the waves only make traffic: 16-byte loads and stores through the same vector memory path
  if ((threadIdx.x >> 6) & 1) {
    u32x4 acc = {0, 0, 0, 0};
    for (int i = 0; i < 64; ++i) {
      const u32x4 v = noise[(tid + i * 8191) & 0xfffff];
      acc += v;
      sink[(tid + i * 4099) & 0xfffff] = acc;
    }
    if (acc[0] == 0x12345678u) out[0] = acc[1];
    return;
  }
  for (int rep = 0; rep < 4; ++rep) {
    VARIANT(0, "%[so]", "")
    VARIANT(1, "%[so]", "s_nop 0\n\t")
    VARIANT(2, "%[so]", "s_nop 1\n\t")
    VARIANT(3, "%[so]", "s_nop 3\n\t")
    VARIANT(4, "0", "")
    VARIANT(5, "0", "s_nop 0\n\t")
    VARIANT(6, "0", "s_nop 1\n\t")
    VARIANT(7, "0", "s_nop 3\n\t")
  }
}

int main() {
  const int blocks = 2048, threads = 512, total = blocks * threads;
  uint32_t* out;
  u32x4 *noise, *sink;
  hipMalloc(&out, (size_t)NV * total * 16);
  hipMalloc(&noise, (size_t)(1 << 20) * 16);
  hipMalloc(&sink, (size_t)(1 << 20) * 16);
  hipMemset(noise, 1, (size_t)(1 << 20) * 16);
  uint32_t* host = (uint32_t*)malloc((size_t)NV * total * 16);
  const char* name[NV] = {"SGPR soffset, 0 wait states", "SGPR soffset, 1", "SGPR soffset, 2", "SGPR soffset, 4",
                          "soffset 0 (immediate), 0 wait states", "immediate, 1", "immediate, 2", "immediate, 4"};
  long bad[NV] = {0}, lanes[NV][64] = {{0}};
  for (int round = 0; round < 8; ++round) {
    hipMemset(out, 0, (size_t)NV * total * 16);
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 0, 0, out, noise, sink, total, 0);
    hipDeviceSynchronize();
    hipMemcpy(host, out, (size_t)NV * total * 16, hipMemcpyDeviceToHost);
    for (int v = 0; v < NV; ++v)
      for (int t = 0; t < total; ++t) {
        if ((t >> 6) & 1) continue;  // traffic waves
        for (int e = 0; e < 4; ++e) {
          const uint32_t w = host[((size_t)v * total + t) * 4 + e];
          if ((w & 0xfff00000u) == 0xbad00000u) { ++bad[v]; ++lanes[v][t & 63]; }
        }
      }
  }
  for (int v = 0; v < NV; ++v) {
    printf("%-40s overwritten dwords stored: %ld of %ld", name[v], bad[v], (long)8 * total / 2 * 4);
    if (bad[v]) {
      printf("   lanes:");
      for (int l = 0; l < 64; ++l) if (lanes[v][l]) printf(" %d", l);
    }
    printf("\n");
  }
  return 0;
}
