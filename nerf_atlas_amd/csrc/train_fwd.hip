// Training step, round 5: the forward of a 256-wide Linear with the WEIGHTS resident in registers.
//
//   y[N, 256] = act([x0 (256) | x1 (<= 80)]) . W^T + b                 (src/neural_blocks.py:288-296, training mode)
//
// lsnt::kernel<0> (train_gemm.hip) streams the packed W from L2 once per 64-sample tile -- 256 KiB per tile, four times the tile's
// HBM bytes, through the same vector memory path as the row fetches and the stores; it runs a 256 x 256 layer at 3.7 TB/s of its
// 537 MB, and a skip layer [256 | 38] as two passes.  Same data flow as the input-gradient role of lsbw::kernel (train_bwd.hip)
// instead: a workgroup owns a sample slice and one HALF of the output columns; its eight waves are 4 column tiles x 2 halves of
// k, each holding ITS fragments of W for the whole launch (K = 256: 8 k steps x (hi | lo) = 64 registers; K = 325: 88), so the
// loop moves nothing but the rows: fetched TWO stages ahead as whole contiguous rows (x1's unaligned 38 / 69 columns as dwords),
// activated, split into bf16 hi | lo planes in LDS ([sample][k of the concatenation]: both sources in ONE image, so a skip layer
// is one pass), 24-33 MFMAs per wave and stage, the two k halves of a tile meet in an LDS tile that all threads carry out as
// whole rows (+ bias).  x0 is read by both column halves: workgroups b and b + 8, same XCD, same time -> the second read is L2.
// Same three bf16 products per k, fp32 accumulation; the k order of the additions differs from lsnt's (two halves): last-bit
// differences, pinned at 2e-6 of the largest value by tests/test_gpu_train_gemm.py like the other kernel pairs.
#include <atomic>
#include <type_traits>
#include <stdlib.h>
#include <string.h>
#include "common.h"
#include "train_shared.h"

namespace na {
#ifndef TFW_ABLATE
#define TFW_ABLATE 0  // timing experiments: 1 no row fetches, 2 no convert / LDS fill, 4 no MFMAs, 8 no stores
#endif
namespace lsfw {
constexpr int SS = 32;                  // samples per stage
constexpr int KMAX = 336;               // 256 + 80
constexpr int NKH = 11;                 // k steps a wave can hold: ceil(21 / 2)
constexpr int OP = 528;                 // row pitch of a partial output tile (128 floats + 4)
constexpr int OT = SS * OP;
template <bool X1> struct Geo {
  static constexpr int P = X1 ? KMAX * 2 + 16 : 528;   // row pitch of a plane: 172 / 132 dwords = 4 x odd mod 64: conflict-free b128 reads
  static constexpr int PLANE = SS * P;
  static constexpr int STAGE = 2 * PLANE;              // hi | lo
  static constexpr int LDS = 2 * STAGE + 4 * OT;       // two stage buffers + (two k halves) x (two stages) of output tiles: 134 / 154 KiB
};

struct Args {
  const float* x0;   // [N, 256]
  const float* x1;   // [N, in1] or null
  const char* wp;    // packed W [256, K] (layout of lsnt::pack_many_kernel)
  const float* bias; // [256] or null
  float* y;          // [N, 256]
  int in1, NCH, nks, nsl, xcd_map;
  int64_t N;
};

template <int ACT, bool X1>
__global__ __launch_bounds__(512) void kernel(Args g) {
  using G = Geo<X1>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int slice, h;
  if (g.xcd_map) { const int p = blockIdx.x >> 3; h = p & 1; slice = (p >> 1) * 8 + (blockIdx.x & 7); }
  else { h = blockIdx.x & 1; slice = blockIdx.x >> 1; }
  const int64_t nst_all = (g.N + SS - 1) / SS;
  const int64_t per = (nst_all + g.nsl - 1) / g.nsl;
  const int64_t st0 = slice * per;
  const int nst = (int)((st0 + per <= nst_all ? per : (nst_all > st0 ? nst_all - st0 : 0)));
  char* const otile = smem + 2 * G::STAGE;

  const int c4 = tid & 63, r0 = tid >> 6;    // x0: piece c4 (4 columns) of rows r0 + 8 j, j = 0..3
  const int xc = tid & 31, xr0 = tid >> 5;   // x1 (and the output half): piece xc of rows xr0 + 16 j, j = 0..1
  const uint32_t o0 = (uint32_t)((r0 * 256 + 4 * c4) * 4);
  uint32_t o1[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) o1[e] = (X1 && 4 * xc + e < g.in1) ? (uint32_t)((xr0 * g.in1 + 4 * xc + e) * 4) : lsnt::OOB;
  const uint32_t oy = (uint32_t)((xr0 * 256 + 128 * h + 4 * xc) * 4);
  f32x4 a0[4], a1[4], b0[2], b1[2];  // two stages in flight: the launch is bound by bytes in flight, and the registers are there
  auto stage_rsrc = [&](const float* base, int ld, int st) __attribute__((always_inline)) {
    const int64_t m0 = (st >= 0 && st < nst) ? (st0 + st) * SS : g.N;
    return lsnt::tile_rsrc(base, ld, m0, g.N, g.wp);
  };
  auto load = [&](f32x4 (&a)[4], f32x4 (&b)[2], int st) __attribute__((always_inline)) {
    if (TFW_ABLATE & 1) return;
    const __amdgpu_buffer_rsrc_t r0s = stage_rsrc(g.x0, 256, st);
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r0s, o0, 8 * j * 256 * 4, 0));
    if constexpr (X1) {
      const __amdgpu_buffer_rsrc_t r1s = stage_rsrc(g.x1, g.in1, st);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) b[j][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r1s, o1[e], 16 * j * g.in1 * 4, 0));
    }
  };
  const int ro0 = r0 * G::P + c4 * 8, ro1 = xr0 * G::P + (256 + 4 * xc) * 2;
  auto convert = [&](const f32x4 (&a)[4], const f32x4 (&b)[2], char* buf) __attribute__((always_inline)) {
    if (TFW_ABLATE & 2) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 v = a[j];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = tact(v[e], ACT);
      bf16x4 hi, lo;
      split4(v, hi, lo);
      char* p = buf + ro0 + 8 * j * G::P;
      *(bf16x4*)p = hi;
      *(bf16x4*)(p + G::PLANE) = lo;
    }
    if constexpr (X1) {
      if (4 * xc < KMAX - 256) {  // (columns 256 .. 335 of the image: 20 pieces per row)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          f32x4 v = b[j];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = tact(v[e], ACT);
          bf16x4 hi, lo;
          split4(v, hi, lo);
          char* p = buf + ro1 + 16 * j * G::P;
          *(bf16x4*)p = hi;
          *(bf16x4*)(p + G::PLANE) = lo;
        }
      }
    }
  };
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (g.bias != nullptr) {
#pragma unroll
    for (int e = 0; e < 4; ++e) bias4[e] = g.bias[128 * h + 4 * xc + e];
  }
  // y of stage st: the two k halves of every tile summed, + bias, as whole row pieces
  auto finish = [&](int st) __attribute__((always_inline)) {
    if (TFW_ABLATE & 8) return;
    const __amdgpu_buffer_rsrc_t ry = stage_rsrc(g.y, 256, st);
    const char* ot = otile + (st & 1) * 2 * OT + xr0 * OP + xc * 16;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const f32x4 u = *(const f32x4*)(ot + 16 * j * OP), w = *(const f32x4*)(ot + OT + 16 * j * OP);
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (u[e] + w[e]) + bias4[e];
      // (row step in the vector offset, soffset 0: build.check_store_data_overwrite, tools/hw/store_soffset_hazard.hip)
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ry, oy + (uint32_t)(16 * j * 256 * 4), 0, 0);
    }
  };

  load(a0, b0, 0);
  load(a1, b1, 1);
  // this wave's fragments of W: column tile 4 h + (wave & 3), k steps [k0, k0 + nk)
  const int T = 4 * h + (wave & 3), kh = wave >> 2;
  const int half = (g.nks + 1) >> 1;
  const int k0 = kh ? half : 0, nk = kh ? g.nks - half : half;
  bf16x8 wh[NKH], wl[NKH];
  {
    const char* base = g.wp + (size_t)(T >> 1) * g.NCH * (2 * lsnt::SEG) + lane * 16;
#pragma unroll
    for (int i = 0; i < NKH; ++i) {
      const int ks = k0 + i < g.nks ? k0 + i : g.nks - 1;  // (slots past nk: a valid address, never used)
      const char* f = base + (size_t)(((ks >> 3) * 2 + (T & 1)) * 8 + (ks & 7)) * 2048;
      wh[i] = *(const bf16x8*)f;
      wl[i] = *(const bf16x8*)(f + 1024);
    }
  }
  const int n = lane & 31, hh = lane >> 5;
  const int fro = n * G::P + hh * 16 + k0 * 32;
  const int oto = kh * OT + n * OP + (32 * (wave & 3) + 4 * hh) * 4;
  auto mma = [&](const char* buf, char* ot) __attribute__((always_inline)) {
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    const char* fr = buf + fro;
#pragma unroll
    for (int i = 0; i < NKH; ++i) {
      if (i < 8 || i < nk) {  // (wave-uniform; K = 256: exactly 8 per wave)
        const bf16x8 xh = *(const bf16x8*)(fr + i * 32);
        const bf16x8 xl = *(const bf16x8*)(fr + G::PLANE + i * 32);
        if (TFW_ABLATE & 4) { acc[i] += (float)xh[0] + (float)xl[1] + (float)wl[i][0] + (float)wh[i][1]; continue; }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[i], xh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[i], xl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[i], xh, acc, 0, 0, 0);
      }
    }
    // register q = column 32 (wave & 3) + 8 (q >> 2) + 4 hh + (q & 3) of sample n
#pragma unroll
    for (int k = 0; k < 4; ++k) *(f32x4*)(ot + oto + 32 * k) = f32x4{acc[4 * k], acc[4 * k + 1], acc[4 * k + 2], acc[4 * k + 3]};
  };
  // (X1: every stage's conversion writes ALL 80 columns behind the first source -- zeros past in1 -- so the k steps up to 336 read finite zeros)
  convert(a0, b0, smem);
  load(a0, b0, 2);
  __syncthreads();
  // The two waves of a SIMD (k halves 0 and 1 of one column tile) run in ANTIPHASE: one converts / stores while the other
  // multiplies (in lockstep the conversion -- 150 VALU per thread and stage, twice that with sin -- and the MFMAs added up:
  // 170 us per layer, 210 with sin)
  if (kh == 0) {
    for (int s = 0; s < nst; s += 2) {
      mma(smem, otile);
      __builtin_amdgcn_sched_barrier(0);
      finish(s - 1);
      convert(a1, b1, smem + G::STAGE);
      load(a1, b1, s + 3);
      __syncthreads();
      if (s + 1 < nst) {
        mma(smem + G::STAGE, otile + 2 * OT);
        __builtin_amdgcn_sched_barrier(0);
        finish(s);
        convert(a0, b0, smem);
        load(a0, b0, s + 4);
        __syncthreads();
      }
    }
  } else {
    for (int s = 0; s < nst; s += 2) {
      finish(s - 1);
      convert(a1, b1, smem + G::STAGE);
      load(a1, b1, s + 3);
      __builtin_amdgcn_sched_barrier(0);  // (the scheduler sinks the fetches below the MFMAs otherwise)
      mma(smem, otile);
      __syncthreads();
      if (s + 1 < nst) {
        finish(s);
        convert(a0, b0, smem);
        load(a0, b0, s + 4);
        __builtin_amdgcn_sched_barrier(0);
        mma(smem + G::STAGE, otile + 2 * OT);
        __syncthreads();
      }
    }
  }
  finish(nst - 1);
}

// Not for sin: both column halves convert ALL of x0 (each needs the whole k range in its LDS), so the activation is computed
// twice per element -- nothing for LeakyReLU, 40 VALU instructions per element for the Cody-Waite sin: measured in the training
// step 159 us against lsnt's 147 (skip layer 261 against 222), where LeakyReLU layers run 122 against 141-147 (171 against 182).
// NA_TRAIN_FUSED_FWD=0 | all: never | every activation.
static bool wanted(int64_t N, int out, int in0, int in1, int act) {
  static const int mode = [] { const char* e = getenv("NA_TRAIN_FUSED_FWD"); return e == nullptr ? 1 : strcmp(e, "0") == 0 ? 0 : strcmp(e, "all") == 0 ? 2 : 1; }();
  return mode != 0 && (mode == 2 || act != NA_ACT_SIN) && in0 == 256 && out == 256 && in1 >= 0 && in1 <= KMAX - 256 && N >= 8192;
}

template <bool X1>
static auto pick_act(int act) -> void (*)(Args) {
  if (act == NA_ACT_LEAKY_RELU) return kernel<NA_ACT_LEAKY_RELU, X1>;
  if (act == NA_ACT_SIN) return kernel<NA_ACT_SIN, X1>;
  return kernel<NA_ACT_NONE, X1>;
}

static int launch(Args a, int act, hipStream_t st, const char* what) {
  const int K = 256 + a.in1;
  a.NCH = (K + lsnt::KC - 1) / lsnt::KC;
  if (a.NCH < 2) a.NCH = 2;
  a.nks = (K + 15) / 16;
  const int64_t nst = (a.N + SS - 1) / SS;
  const int cus = lsnt::cu_count();
  int nsl = cus / 2;
  if (nst / 4 < nsl) nsl = (int)(nst / 4 > 0 ? nst / 4 : 1);
  a.nsl = nsl;
  a.xcd_map = (nsl % 8) == 0;
  const bool x1 = a.in1 > 0;
  auto k = x1 ? pick_act<true>(act) : pick_act<false>(act);
  const int lds = x1 ? Geo<true>::LDS : Geo<false>::LDS;
  const int which = (x1 ? 3 : 0) + (act == NA_ACT_LEAKY_RELU ? 1 : act == NA_ACT_SIN ? 2 : 0);
  static std::atomic<uint64_t> done[6];
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t bit = 1ull << (dev & 63);
  if (!(done[which].load(std::memory_order_acquire) & bit)) {
    hipError_t e2 = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e2 != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e2)); return NA_EHIP; }
    done[which].fetch_or(bit, std::memory_order_release);
  }
  hipLaunchKernelGGL(k, dim3(2 * nsl), dim3(512), lds, st, a);
  return check_launch(what);
}
}  // namespace lsfw

bool train_fwd_wanted(int64_t N, int out, int in0, int in1, int act) { return lsfw::wanted(N, out, in0, in1, act); }

int train_fwd_launch(const float* x0, const float* x1, int in1, int64_t N, const void* w_packed, const float* b, int pre_act, float* y,
                     hipStream_t st, const char* what) {
  lsfw::Args a{};
  a.x0 = x0; a.x1 = in1 > 0 ? x1 : nullptr; a.in1 = in1; a.wp = (const char*)w_packed; a.bias = b; a.y = y; a.N = N;
  return lsfw::launch(a, pre_act, st, what);
}
}  // namespace na
