// Probe (gfx950): a 16-byte global store followed K wait states later by v_cvt_scalef32_2xpk16_fp6_f32 whose six destination
// registers include the store's four data registers.  Found through a build of csrc/render_ls.hip under another instruction
// scheduler (-mllvm -amdgpu-sched-strategy=max-ilp): the fp6 weight pack then emitted different words 2..5 of one conversion,
// exactly the registers a store two instructions earlier was still reading (the compiler had put `s_nop 0` between them).
//   hipcc --offload-arch=gfx950 -O3 tools/hw/store_cvt_hazard.hip -o store_cvt_hazard && ./store_cvt_hazard
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x6 __attribute__((ext_vector_type(6)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int K>
__global__ void probe(const float* __restrict__ in, u32x4* __restrict__ stored, int* __restrict__ out, int* __restrict__ ref, int rounds) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  f32x16 a, b;
  for (int r = 0; r < 16; ++r) { a[r] = in[(tid * 32 + r) & 4095]; b[r] = in[(tid * 32 + 16 + r) & 4095]; }
  const float sc = 1.0f;
  const i32x6 want = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a, b, sc);
  int bad_cvt = 0, bad_store = 0;
  for (int it = 0; it < rounds; ++it) {
    const uint32_t t0 = 0x11110000u + it, t1 = 0x22220000u + it, t2 = 0x33330000u + it, t3 = 0x44440000u + it;
    u32x4* dst = stored + (size_t)tid * rounds + it;
    int o0, o1, o2, o3, o4, o5;
    asm volatile(
        "v_mov_b32 v42, %[t0]\n\tv_mov_b32 v43, %[t1]\n\tv_mov_b32 v44, %[t2]\n\tv_mov_b32 v45, %[t3]\n\t"
        "s_nop 4\n\t"
        "global_store_dwordx4 %[p], v[42:45], off\n\t"
        "s_nop %[k]\n\t"
        "v_cvt_scalef32_2xpk16_fp6_f32 v[40:45], %[a], %[b], %[s]\n\t"
        "s_nop 7\n\t"
        "v_mov_b32 %[o0], v40\n\tv_mov_b32 %[o1], v41\n\tv_mov_b32 %[o2], v42\n\tv_mov_b32 %[o3], v43\n\tv_mov_b32 %[o4], v44\n\tv_mov_b32 %[o5], v45\n\t"
        : [o0] "=&v"(o0), [o1] "=&v"(o1), [o2] "=&v"(o2), [o3] "=&v"(o3), [o4] "=&v"(o4), [o5] "=&v"(o5)
        : [t0] "v"(t0), [t1] "v"(t1), [t2] "v"(t2), [t3] "v"(t3), [p] "v"(dst), [a] "v"(a), [b] "v"(b), [s] "v"(sc), [k] "n"(K)
        : "v40", "v41", "v42", "v43", "v44", "v45", "memory");
    if (o0 != want[0] || o1 != want[1] || o2 != want[2] || o3 != want[3] || o4 != want[4] || o5 != want[5]) ++bad_cvt;
  }
  __builtin_amdgcn_s_waitcnt(0);
  for (int it = 0; it < rounds; ++it) {
    const u32x4 v = stored[(size_t)tid * rounds + it];
    if (v[0] != 0x11110000u + it || v[1] != 0x22220000u + it || v[2] != 0x33330000u + it || v[3] != 0x44440000u + it) ++bad_store;
  }
  out[tid] = bad_cvt;
  ref[tid] = bad_store;
}

template <int K>
static void run(const float* din, u32x4* dst, int* dout, int* dref, int threads, int rounds) {
  hipMemset(dout, 0, threads * 4); hipMemset(dref, 0, threads * 4);
  hipLaunchKernelGGL(probe<K>, dim3(threads / 256), dim3(256), 0, 0, din, dst, dout, dref, rounds);
  hipDeviceSynchronize();
  static int h[1 << 18], g[1 << 18];
  hipMemcpy(h, dout, threads * 4, hipMemcpyDeviceToHost); hipMemcpy(g, dref, threads * 4, hipMemcpyDeviceToHost);
  long bc = 0, bs = 0;
  for (int i = 0; i < threads; ++i) { bc += h[i]; bs += g[i]; }
  printf("s_nop %d between the store and the conversion: wrong conversions %ld, wrong stored data %ld (of %ld each)\n", K, bc, bs, (long)threads * rounds);
}

int main() {
  const int threads = 256 * 1024, rounds = 16;
  float h[4096];
  for (int i = 0; i < 4096; ++i) h[i] = (float)((i * 2654435761u) % 2001) / 250.0f - 4.0f;
  float* din; u32x4* dst; int *dout, *dref;
  hipMalloc(&din, sizeof(h)); hipMalloc(&dst, (size_t)threads * rounds * 16); hipMalloc(&dout, threads * 4); hipMalloc(&dref, threads * 4);
  hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
  run<0>(din, dst, dout, dref, threads, rounds);
  run<1>(din, dst, dout, dref, threads, rounds);
  run<2>(din, dst, dout, dref, threads, rounds);
  run<3>(din, dst, dout, dref, threads, rounds);
  run<4>(din, dst, dout, dref, threads, rounds);
  run<7>(din, dst, dout, dref, threads, rounds);
  return 0;
}
