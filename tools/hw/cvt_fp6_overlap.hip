// Probe (gfx950): v_cvt_scalef32_2xpk16_fp6_f32 vdst[6], src0[16], src1[16], scale -- which overlaps of the destination with its
// operands does the hardware tolerate?  LLVM (ROCm 7.2) does not mark the destination early-clobber, so the register allocator
// is free to put the scale or the first source registers inside it; found when a build of csrc/render_ls.hip under
// -mllvm -amdgpu-sched-strategy=max-ilp packed different fp6 words 2..5 for one conversion whose scale was v1 and whose
// destination was v[0:5].
//   hipcc --offload-arch=gfx950 -O3 tools/hw/cvt_fp6_overlap.hip -o cvt_fp6_overlap && ./cvt_fp6_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x6 __attribute__((ext_vector_type(6)));

// src0 lives in v[64:79], src1 in v[80:95], the scale and the destination as the variant says
#define LOAD_SRC                                                                                                  \
  "v_mov_b32 v64, %[a0]\n\tv_mov_b32 v65, %[a1]\n\tv_mov_b32 v66, %[a2]\n\tv_mov_b32 v67, %[a3]\n\t"               \
  "v_mov_b32 v68, %[a4]\n\tv_mov_b32 v69, %[a5]\n\tv_mov_b32 v70, %[a6]\n\tv_mov_b32 v71, %[a7]\n\t"               \
  "v_mov_b32 v72, %[a8]\n\tv_mov_b32 v73, %[a9]\n\tv_mov_b32 v74, %[a10]\n\tv_mov_b32 v75, %[a11]\n\t"             \
  "v_mov_b32 v76, %[a12]\n\tv_mov_b32 v77, %[a13]\n\tv_mov_b32 v78, %[a14]\n\tv_mov_b32 v79, %[a15]\n\t"           \
  "v_mov_b32 v80, %[b0]\n\tv_mov_b32 v81, %[b1]\n\tv_mov_b32 v82, %[b2]\n\tv_mov_b32 v83, %[b3]\n\t"               \
  "v_mov_b32 v84, %[b4]\n\tv_mov_b32 v85, %[b5]\n\tv_mov_b32 v86, %[b6]\n\tv_mov_b32 v87, %[b7]\n\t"               \
  "v_mov_b32 v88, %[b8]\n\tv_mov_b32 v89, %[b9]\n\tv_mov_b32 v90, %[b10]\n\tv_mov_b32 v91, %[b11]\n\t"             \
  "v_mov_b32 v92, %[b12]\n\tv_mov_b32 v93, %[b13]\n\tv_mov_b32 v94, %[b14]\n\tv_mov_b32 v95, %[b15]\n\t"
#define SRC_OPS                                                                                                          \
  [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]), [a3] "v"(a[3]), [a4] "v"(a[4]), [a5] "v"(a[5]), [a6] "v"(a[6]),          \
  [a7] "v"(a[7]), [a8] "v"(a[8]), [a9] "v"(a[9]), [a10] "v"(a[10]), [a11] "v"(a[11]), [a12] "v"(a[12]), [a13] "v"(a[13]),  \
  [a14] "v"(a[14]), [a15] "v"(a[15]), [b0] "v"(b[0]), [b1] "v"(b[1]), [b2] "v"(b[2]), [b3] "v"(b[3]), [b4] "v"(b[4]),      \
  [b5] "v"(b[5]), [b6] "v"(b[6]), [b7] "v"(b[7]), [b8] "v"(b[8]), [b9] "v"(b[9]), [b10] "v"(b[10]), [b11] "v"(b[11]),      \
  [b12] "v"(b[12]), [b13] "v"(b[13]), [b14] "v"(b[14]), [b15] "v"(b[15]), [s] "v"(sc)
#define CLOB "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", \
  "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100",    \
  "v101", "v102", "v103"
#define OUTS [o0] "=&v"(o[0]), [o1] "=&v"(o[1]), [o2] "=&v"(o[2]), [o3] "=&v"(o[3]), [o4] "=&v"(o[4]), [o5] "=&v"(o[5])
#define READ(D0, D1, D2, D3, D4, D5) \
  "s_nop 7\n\tv_mov_b32 %[o0], " D0 "\n\tv_mov_b32 %[o1], " D1 "\n\tv_mov_b32 %[o2], " D2 "\n\tv_mov_b32 %[o3], " D3 "\n\tv_mov_b32 %[o4], " D4 "\n\tv_mov_b32 %[o5], " D5 "\n\t"

__global__ void probe(const float* __restrict__ in, float scale, int* __restrict__ bad) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  f32x16 a, b;
  for (int r = 0; r < 16; ++r) { a[r] = in[(tid * 37 + r) & 4095]; b[r] = in[(tid * 37 + 16 + r) & 4095]; }
  const float sc = scale;
  const i32x6 want = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a, b, sc);
  int o[6];
  auto cmp = [&](int which) {
    bool ok = true;
    for (int k = 0; k < 6; ++k) ok = ok && o[k] == want[k];
    if (!ok) atomicAdd(bad + which, 1);
    for (int k = 0; k < 6; ++k) if (o[k] != want[k]) atomicAdd(bad + 8 + which * 6 + k, 1);
  };
  // 0: everything disjoint (destination v[96:101], scale v102)
  asm volatile(LOAD_SRC "v_mov_b32 v102, %[s]\n\ts_nop 4\n\tv_cvt_scalef32_2xpk16_fp6_f32 v[96:101], v[64:79], v[80:95], v102\n\t"
               READ("v96", "v97", "v98", "v99", "v100", "v101") : OUTS : SRC_OPS : CLOB);
  cmp(0);
  // 1: the scale inside the destination (second register)
  asm volatile(LOAD_SRC "v_mov_b32 v97, %[s]\n\ts_nop 4\n\tv_cvt_scalef32_2xpk16_fp6_f32 v[96:101], v[64:79], v[80:95], v97\n\t"
               READ("v96", "v97", "v98", "v99", "v100", "v101") : OUTS : SRC_OPS : CLOB);
  cmp(1);
  // 2: the scale in the LAST destination register
  asm volatile(LOAD_SRC "v_mov_b32 v101, %[s]\n\ts_nop 4\n\tv_cvt_scalef32_2xpk16_fp6_f32 v[96:101], v[64:79], v[80:95], v101\n\t"
               READ("v96", "v97", "v98", "v99", "v100", "v101") : OUTS : SRC_OPS : CLOB);
  cmp(2);
  // 3: destination = the first six registers of src0
  asm volatile(LOAD_SRC "v_mov_b32 v102, %[s]\n\ts_nop 4\n\tv_cvt_scalef32_2xpk16_fp6_f32 v[64:69], v[64:79], v[80:95], v102\n\t"
               READ("v64", "v65", "v66", "v67", "v68", "v69") : OUTS : SRC_OPS : CLOB);
  cmp(3);
  // 4: destination = the first six registers of src1
  asm volatile(LOAD_SRC "v_mov_b32 v102, %[s]\n\ts_nop 4\n\tv_cvt_scalef32_2xpk16_fp6_f32 v[80:85], v[64:79], v[80:95], v102\n\t"
               READ("v80", "v81", "v82", "v83", "v84", "v85") : OUTS : SRC_OPS : CLOB);
  cmp(4);
  // 5: destination = the LAST six registers of src1
  asm volatile(LOAD_SRC "v_mov_b32 v102, %[s]\n\ts_nop 4\n\tv_cvt_scalef32_2xpk16_fp6_f32 v[90:95], v[64:79], v[80:95], v102\n\t"
               READ("v90", "v91", "v92", "v93", "v94", "v95") : OUTS : SRC_OPS : CLOB);
  cmp(5);
}

int main() {
  float h[4096];
  for (int i = 0; i < 4096; ++i) h[i] = (float)((i * 2654435761u) % 2001) / 250.0f - 4.0f;
  float* din; int* dbad;
  hipMalloc(&din, sizeof(h)); hipMalloc(&dbad, 64 * 4);
  hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
  const char* names[6] = {"disjoint", "scale = vdst[1]", "scale = vdst[5]", "vdst = src0[0:5]", "vdst = src1[0:5]", "vdst = src1[10:15]"};
  for (int rep = 0; rep < 2; ++rep) {
    hipMemset(dbad, 0, 64 * 4);
    const int threads = 256 * 1024;
    hipLaunchKernelGGL(probe, dim3(threads / 256), dim3(256), 0, 0, din, 0.5f, dbad);
    hipDeviceSynchronize();
    int hb[64];
    hipMemcpy(hb, dbad, sizeof(hb), hipMemcpyDeviceToHost);
    for (int w = 0; w < 6; ++w)
      printf("run %d  %-20s wrong results %7d of %d   by destination dword: %d %d %d %d %d %d\n", rep, names[w], hb[w], threads, hb[8 + w * 6], hb[9 + w * 6],
             hb[10 + w * 6], hb[11 + w * 6], hb[12 + w * 6], hb[13 + w * 6]);
  }
  return 0;
}
