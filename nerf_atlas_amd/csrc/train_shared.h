// Pieces shared by the training-step GEMM units (train_gemm.hip, train_bwd.hip): the 2-way bf16 split, the activations, the
// branch-free tile buffers, the packed-operand geometry of lsnt::pack_kernel / pack_many_kernel.
#pragma once
#include <atomic>
#include "common.h"

namespace na {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

// sin / cos of the split-bf16 training GEMMs (round 6): the hardware v_sin_f32 / v_cos_f32 on a revolution count reduced with a
// two-constant 1 / (2 pi) -- the reduction of mlp_engine.h's sin_hw2, which the inference kernels have used since round 2 (same
// end-to-end L-inf as the Cody-Waite polynomial there).  One shared reduction for the pair: where a kernel needs act(x) AND act'(x)
// of the same element (lsbw: the weight gradient's operand and the input gradient's factor) the compiler keeps one.  ~1e-6
// absolute, far inside the 2^-16 relative error of the three bf16 products downstream; the polynomial pair cost 48 us per sin
// layer in lsbw (243 against 195 us for LeakyReLU).  NA_TRAIN_POLY_SIN=1 at build time restores the polynomials for A/B runs.
#ifndef NA_TRAIN_POLY_SIN
#define NA_TRAIN_POLY_SIN 0
#endif
__device__ __forceinline__ float trev(float x) {
  const float q = rintf(x * 0.15915493667125702f);
  float r = fmaf(x, 0.15915493667125702f, -q);
  return fmaf(x, 6.4206382432985265e-09f, r);
}
__device__ __forceinline__ float tact(float v, int act) {
  if (act == NA_ACT_LEAKY_RELU) return __builtin_amdgcn_fmed3f(v, v * 0.01f, 3.0e38f);
  if (act == NA_ACT_SIN) return NA_TRAIN_POLY_SIN ? sin_cw(v) : __builtin_amdgcn_sinf(trev(v));
  return v;
}
__device__ __forceinline__ float tact_grad(float v, int act) {
  if (act == NA_ACT_LEAKY_RELU) return v > 0.f ? 1.f : 0.01f;
  if (act == NA_ACT_SIN) return NA_TRAIN_POLY_SIN ? cos_cw(v) : __builtin_amdgcn_cosf(trev(v));
  return 1.f;
}

// v = hi + lo + O(2^-17 |v|), both halves rounded to nearest even.  Pairwise, so that each pair costs v_cvt_pk_bf16_f32, a shift,
// a mask, v_pk_add_f32 and v_cvt_pk_bf16_f32 (element by element the compiler converted every high half twice)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split4(const f32x4 v, bf16x4& hi, bf16x4& lo) {
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const f32x2 w = {v[2 * p], v[2 * p + 1]};
    const bf16x2 h = __builtin_convertvector(w, bf16x2);
    const uint32_t hb = __builtin_bit_cast(uint32_t, h);
    const f32x2 f = {__builtin_bit_cast(float, hb << 16), __builtin_bit_cast(float, hb & 0xffff0000u)};
    const bf16x2 l = __builtin_convertvector(w - f, bf16x2);
    hi[2 * p] = h[0]; hi[2 * p + 1] = h[1];
    lo[2 * p] = l[0]; lo[2 * p + 1] = l[1];
  }
}

namespace lsnt {
constexpr int KC = 128;             // k per LDS fill = 8 k steps = one segment of the weight stream per column tile
constexpr int SEG = 8 * 2048;       // stream bytes of one (column group, chunk, column tile): 8 k steps x (hi | lo) fragments
constexpr uint32_t OOB = 0x78000000u;  // a byte offset past every tile buffer: the hardware drops the access
constexpr int kNoScratch = 1;            // launch(): hipMallocAsync refused (returned to the dispatcher, never to the C ABI)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const float* base, int ld, int64_t m0, int64_t rows, const void* dummy) {
  int64_t bytes = base == nullptr ? 0 : (rows - m0) * ld * 4;
  if (bytes > 0x70000000ll) bytes = 0x70000000ll;
  if (bytes < 0) bytes = 0;
  return __builtin_amdgcn_make_buffer_rsrc((void*)(base == nullptr ? (const float*)dummy : base + m0 * ld), 0, (int)bytes, 0x00020000);
}

static int cu_count() {
  static int n = 0;
  if (n == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    n = v;
  }
  return n;
}

}  // namespace lsnt

// Scratch of the training GEMMs (partial gradients, a per-call packed operand): ONE grow-only device buffer per (device, stream),
// handed out without any driver call once it exists.  Uses on one stream are stream-ordered, so the next call may overwrite it;
// another stream gets another buffer.  (hipMallocAsync / hipFreeAsync around every launch, rounds 3-5, cost ~230 us of HOST time
// per call on ROCm 7.2 whatever the pool's release threshold: 14 calls made the training step host-bound at 5.5 ms with 5.15 ms
// of kernels -- tools/train_host_time.py.)  A buffer that has to grow is replaced, the old one is kept until process exit (kernels
// may still be reading it).  nullptr = the allocation failed (the caller falls back or reports).  Defined in train_gemm.hip.
void* train_scratch(hipStream_t st, size_t bytes);

// dW[row, col] (+)= the sum over `nwg` partials of lstn::PART floats each (dW 256 x 256 | 8 row groups of db), in a fixed order
// (lstn::reduce_kernel, defined in train_gemm.hip)
int train_reduce_many(int n, const float* const* part, const int* nwg, const int* out, const int* in, const int* ldw, float* const* dW,
                      float* const* db, hipStream_t st);
int train_reduce_partials(const float* part, int nwg, int out, int in, int ldw, float* dW, float* db, int overwrite, hipStream_t st);

// y[N, 256] = act([x0 (256) | x1 (in1)]) . W^T + b with W resident in registers (train_fwd.hip); w_packed: W [256, 256 + in1] as
// na_train_pack_many packs it
bool train_fwd_wanted(int64_t N, int out, int in0, int in1, int act);
int train_fwd_launch(const float* x0, int in0, const float* x1, int in1, int64_t N, const void* w_packed, const float* b, int out,
                     int pre_act, float* y, hipStream_t st, const char* what);

}  // namespace na
