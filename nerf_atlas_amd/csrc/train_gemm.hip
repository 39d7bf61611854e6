// Training-step GEMMs of SkipConnMLP (src/neural_blocks.py:288-296 differentiated) on the bf16 matrix
// core with the 2-way split of the fused renderer:  a*b ~= a_lo*b_hi + a_hi*b_lo + a_hi*b_hi  with  x_hi = bf16(x),
// x_lo = bf16(x - x_hi)  (products exact in the fp32 accumulator; relative error of a dot product ~2^-16, same class as
// NA_PREC_BF16X3).  Per k=16 slab a 32x32 tile costs 3 x 32 cycles here against 8 x 64 cycles on
// v_mfma_f32_32x32x2_f32, which turns the three GEMMs of a layer (forward, input gradient, weight gradient) from
// matrix-core-bound into HBM-bound on their [N,256] fp32 operands.
//
//   na_linear_bf16x3   y[N,out]       = act([x0|x1]) . W^T + b                      (forward, training mode)
//   na_linear_dgrad_bf16x3  g_x0|g_x1      = (dY . W) * act'([x0|x1])                    (act backward fused in the epilogue)
//   na_linear_wgrad_bf16x3  dW[out,in]    += dY^T . act([x0|x1]),  db += colsum(dY)      (split over sample slices, fp32 atomics)
//
// Workgroup = WM x WN waves, each a 64x64 output tile (2x2 MFMA 32x32x16); K is staged 32 at a time through a
// double-buffered LDS tile pair [rows][32 k] of hi and lo bf16 with an 80-byte row pitch (conflict-free ds_read_b128
// fragments), global loads for stage k+1 in flight under the MFMAs of stage k, one barrier per stage.  The activation
// (and the split) is applied once per element by the loader; with BN = 256 a sample's row is activated exactly once.
// These K-staged kernels serve small batches and very wide layers; since round 3 every launch with N >= 2048 and at most 1024
// output columns goes to the layer-synchronous kernels in the second half of this file (`lsnt`, `lstn`), which keep the
// arithmetic (the same three products per k, fp32 accumulation; the order of the additions differs: last-bit differences) and
// reorganise the data flow.
#include <atomic>
#include <mutex>
#include <vector>
#include <type_traits>
#include <stdlib.h>
#include <string.h>
#include "common.h"
#include "train_shared.h"

#include "train_gemm_tiled.h"

namespace na {

// ================================================================================ layer-synchronous NT kernel (round 3)
// The same GEMM  C[samples, M] = A[samples, K] . B[M, K]^T  organised like the renderer (render_ls.hip) instead of a
// K-staged tile pair: a workgroup keeps a 64-sample tile of A in LDS as split bf16 planes, 128 k at a time, and streams the
// pre-packed bf16 hi/lo fragments of B from L2 straight into registers (a fragment ring refilled in place), so B never
// passes through LDS or the split arithmetic again and A is read from HBM as whole contiguous rows (the K-staged kernel
// moved 48 KiB through the vector memory path per 24 MFMAs and was bound by that path).
// Three roles of four waves (twelve waves, three per SIMD, 168 registers each).  LOADERS (waves 4-7) fetch the rows two units
// ahead into registers, activate + split them and fill the other LDS buffer.  MOVERS (waves 8-11) carry the finished output tile from LDS to HBM, in parts spread over the units
// of the next tile, and for the input gradient with an activation bring the tile of forward inputs into LDS.  CONSUMERS
// (waves 0-3, one 64-column group of C each) run the MFMAs and the epilogue and touch HBM never: loads and stores of one wave
// retire in order, so a row fetch (HBM latency) or a tile store in front of a weight fragment (L2) would stall the matrix
// pipe; and a wave that mixes kinds of memory work loses the compiler its exact wait counts (two roles in one wave: vmcnt(0)
// in front of every conversion, the prefetch one unit deep whatever the code said).
// A unit = (tile, pass over 256 columns of C, k chunk), one barrier per unit.  Inside a unit a consumer finishes its column tile
// t = 0 before it starts t = 1.
#ifndef TGL_ABLATE
#define TGL_ABLATE 0  // timing experiments: 1 no row fetches, 2 no epilogue stores, 4 no MFMAs, 8 no weight refills, 16 no LDS fill, 32 fill from constants, 64 conversion without its LDS writes
#endif
#ifndef TGL_TRACE
#define TGL_TRACE 0  // experiment builds (tools/ls_variant.py): s_memtime stamps of workgroup 0, waves 0 (consumer) and 4 (loader)
#endif
#ifndef TGL_PRIO
#define TGL_PRIO 0   // s_setprio of the loader and mover waves
#endif
#ifndef TGL_CPRIO
#define TGL_CPRIO 0  // s_setprio of the consumer waves (1, 3: 147-151 against 153-160 us forward alone, 7.5-7.9 ms in the step either way)
#endif
#ifndef TGL_P0F
#define TGL_P0F 6   // of the 16 pieces a mover thread stores per tile, those that go out in the first of the tile's two units: forward
#endif
#ifndef TGL_P0D
#define TGL_P0D 2   // ... input gradient with an activation (the first unit also parks x: 8 / 8 -> 2 / 14: 210 -> 189 us)
#endif
// Cache policies (gfx950 buffer aux bits: 2 = nt), experiment switches.  Timed alone in a loop, non-temporal stores of the
// output tile take the forward from 151-156 to 136-137 us and, with the parked x fetched non-temporal too, the input gradient
// from 200-205 to 177-180 -- but that is the loop: the same 268-MB inputs are read again by the next iteration and survive in
// the 256-MB memory-side cache when the outputs do not pass through it.  In the training step, where every tensor is
// written once and read by the next kernel, the three policies give 7.49 / 7.56 / 7.45 ms (tools/train_bench_ab.py): nothing.
#ifndef TGL_ST_AUX
#define TGL_ST_AUX 0     // the output tile's stores
#endif
#ifndef TGL_X_AUX
#define TGL_X_AUX 0      // the fetches of the forward inputs x an input gradient parks
#endif
#ifndef TGL_RIDE
#define TGL_RIDE 0   // 1: the first column tile's epilogue rides under the second one's MFMAs, an item per k step (measured SLOWER:
                     // forward 151 -> 154 us, input gradient 191 -> 206: its LDS traffic and waits land inside the k steps)
#endif
#ifndef TGL_RD
#define TGL_RD 4     // weight fragment ring of a consumer wave, in k steps (8 = a whole segment, or 4)
#endif
namespace lsnt {
#if TGL_TRACE
__device__ unsigned long long tgl_trace[3][32][4];
#define TGL_STAMP(role, u, slot) do { if (blockIdx.x == 0 && lane == 0 && (u) < 32) tgl_trace[role][(u)][slot] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TGL_STAMP(role, u, slot) do {} while (0)
#endif
constexpr int TS = 64;              // samples per tile
constexpr int PITCH = KC * 2 + 16;  // row pitch of a plane: 68 dwords = 4 mod 64 -> conflict-free b128 fragment reads
constexpr int PLANE = TS * PITCH;
constexpr int BUF = 2 * PLANE;      // hi | lo
constexpr int LDS = 2 * BUF;        // two buffers: 68 KiB
constexpr int PT = 256;             // threads of a role
constexpr int NTHR = 256 + 2 * PT;
constexpr int NPF = TS * (KC / 4) / PT;  // 16-byte pieces per loader thread and unit (8)
constexpr int XP = 260;             // float pitch of the output tile (= the forward-input tile of an input gradient with an activation)
constexpr int XB = TS * XP * 4;     // 65 KiB behind the buffers
constexpr int NXF = TS * 64 / PT;   // its 16-byte pieces per mover thread (16)

struct Args {
  RowSrc a;          // [samples, K] (concat)
  const char* wp;    // packed B: [column group of 64][chunk][tile 2][k step 8][plane 2][lane 64][8 bf16]
  int M, NCH, act;   // NCH = chunks of 128 k (zero padded)
  const float* bias;
  float* y0;
  float* y1;
  const float* x0;
  const float* x1;
  int c0, c1;
  int64_t ntiles;
};

// one thread per (column group, chunk, tile, k step, lane): 8 consecutive k of row m of B, split
__global__ void pack_kernel(const float* __restrict__ W, int M, int K, int NCH, int nrg, char* __restrict__ dst) {
  const int64_t total = (int64_t)nrg * NCH * 2 * 8 * 64;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63), ks = (int)((i >> 6) & 7), t = (int)((i >> 9) & 1);
    const int c = (int)((i >> 10) % NCH), rg = (int)((i >> 10) / NCH);
    const int m = 64 * rg + 32 * t + (lane & 31), k0 = KC * c + 16 * ks + 8 * (lane >> 5);
    bf16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = (m < M && k0 + e < K) ? W[(int64_t)m * K + k0 + e] : 0.f;
      const __bf16 h = (__bf16)v;
      hi[e] = h;
      lo[e] = (__bf16)(v - (float)h);
    }
    char* o = dst + (int64_t)(i >> 6) * 2048 + lane * 16;
    *(bf16x8*)o = hi;
    *(bf16x8*)(o + 1024) = lo;
  }
}

// Round 5: every B operand of a training step packed by ONE launch (na_train_pack_many: the forward's W and the input gradient's
// W^T of all Linears of an MLP -- 24 pack launches, 12 transposing copies and their launch gaps per PlainNeRF step before).
// Entry e packs B_e[m, k] = trans ? W[k * ld + m] : W[m * ld + k] (M x K, zero padded) into dst_e; same layout, same rounding as
// pack_kernel.
constexpr int kPackMany = 32;
struct PackMany {
  const float* W[kPackMany];
  char* dst[kPackMany];
  int M[kPackMany], K[kPackMany], ld[kPackMany], trans[kPackMany], NCH[kPackMany];
  long long first[kPackMany + 1];  // thread index where entry e starts
  int n;
};
__global__ void pack_many_kernel(PackMany pm) {
  const long long total = pm.first[pm.n];
  for (long long g = blockIdx.x * (long long)blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
    int e = 0;
    while (e + 1 < pm.n && g >= pm.first[e + 1]) ++e;  // (wave-uniform but for the few waves that straddle two entries)
    const long long i = g - pm.first[e];
    const int M = pm.M[e], K = pm.K[e], NCH = pm.NCH[e], ld = pm.ld[e];
    const float* __restrict__ W = pm.W[e];
    const int lane = (int)(i & 63), ks = (int)((i >> 6) & 7), t = (int)((i >> 9) & 1);
    const int c = (int)((i >> 10) % NCH), rg = (int)((i >> 10) / NCH);
    const int m = 64 * rg + 32 * t + (lane & 31), k0 = KC * c + 16 * ks + 8 * (lane >> 5);
    bf16x8 hi, lo;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      float v = 0.f;
      if (m < M && k0 + q < K) v = pm.trans[e] ? W[(int64_t)(k0 + q) * ld + m] : W[(int64_t)m * ld + k0 + q];
      const __bf16 h = (__bf16)v;
      hi[q] = h;
      lo[q] = (__bf16)(v - (float)h);
    }
    char* o = pm.dst[e] + (int64_t)(i >> 6) * 2048 + lane * 16;
    *(bf16x8*)o = hi;
    *(bf16x8*)(o + 1024) = lo;
  }
}
static inline int pack_nch(int K) { int n = (K + KC - 1) / KC; return n < 2 ? 2 : n; }
static inline size_t packed_bytes(int M, int K) { return (size_t)((M + 63) / 64) * pack_nch(K) * 2 * SEG; }

// Four consecutive columns `col` .. `col + 3` of row `row` (tile-relative) of the concatenation [p0 (k0 columns) | p1 (k1)],
// zero outside, WITHOUT branches or waits between pieces: buffer loads whose offset lies outside the tile's buffer return
// zero, so every piece issues the same few loads and only the offsets differ (a branchy loader made the compiler wait for
// each piece before it issued the next).  The column counts decide (uniformly) between 16-byte and 4-byte loads.
struct Src2 {
  __amdgpu_buffer_rsrc_t r0, r1;
  int k0, k1;
};
// WHICH: 1 the first source only, 2 the second only -- a caller that knows where its columns lie (a whole k chunk inside one
// source): ONE 16-byte load per piece whatever the row length.  Raw-buffer loads of 16 bytes need only 4-byte alignment and
// are range-checked per dword (profiles/r03/unaligned_probe.log), so a 38-column source takes them too; a piece that crosses
// the end of its row brings the first elements of the NEXT row along, which the caller zeroes when it USES the piece (zeroing
// here would wait for the load).  WHICH 0: both sources, summed element by element on the spot: exact zeros outside, but it
// waits for its loads (only the STRADDLE instantiation uses it).
template <int WHICH = 0, int AUX = 0>
__device__ __forceinline__ f32x4 load_piece(const Src2& s, int row, int col) {
  if (WHICH == 1) {
    const uint32_t o = col < s.k0 ? (uint32_t)((row * s.k0 + col) * 4) : OOB;
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(s.r0, o, 0, AUX));
  }
  if (WHICH == 2) {
    const uint32_t o = (col >= s.k0 && col < s.k0 + s.k1) ? (uint32_t)((row * s.k1 + col - s.k0) * 4) : OOB;
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(s.r1, o, 0, AUX));
  }
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if ((s.k0 & 3) == 0) {
    const uint32_t o = col < s.k0 ? (uint32_t)((row * s.k0 + col) * 4) : OOB;
    v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(s.r0, o, 0, 0));
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t o = col + e < s.k0 ? (uint32_t)((row * s.k0 + col + e) * 4) : OOB;
      v[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(s.r0, o, 0, 0));
    }
  }
  if (s.k1 > 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int ce = col + e;
      const uint32_t o = (ce >= s.k0 && ce < s.k0 + s.k1) ? (uint32_t)((row * s.k1 + ce - s.k0) * 4) : OOB;
      v[e] += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(s.r1, o, 0, 0));
    }
  }
  return v;
}

// MODE 0 forward, 1 input gradient; DACT: the input gradient is scaled by act'(forward input) in the epilogue; STRADDLE: a
// 128-wide chunk of k may hold columns of BOTH concatenated sources (first source not a multiple of 128 wide).  That loader
// adds two fetches per element on the spot, and one copy of it inside the unit loop is enough for the compiler's wait-count
// pass to give up on the whole loop (vmcnt(0) in front of every conversion): it gets its own instantiation.
template <int MODE, bool DACT, bool STRADDLE>
__global__ __launch_bounds__(NTHR) void kernel(Args g) {
  constexpr int RD = TGL_RD;  // depth of the weight fragment ring (8 = the k steps of one segment)
  constexpr int KS = 8;       // k steps of a segment
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int NCH = g.NCH;
  const int nrg = (g.M + 63) >> 6;
  const int NP = (nrg + 3) >> 2;  // passes over the tile: 256 columns of C each
  const int UPT = NCH * NP;       // units per tile
  const int my_tiles = (int)((g.ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x);
  const int nunits = my_tiles * UPT;
  const int K = g.a.k0 + g.a.k1;
  // x goes through LDS (any column layout: the movers' loader handles unaligned rows); that needs a unit in which the
  // consumers do not read it, so the launcher gives a DACT launch at least two chunks (a zero chunk if K <= 128)
  constexpr bool xlds = DACT;
  // The output tile of a (tile, pass): [64 samples][256 columns] fp32.  The consumers write it in the last chunk's unit, the
  // MOVERS store it to HBM during the next tile's units: loads of a wave return in order behind its older stores, so a consumer
  // that stored its own results waited ~3 k cycles per column tile for the stores to complete before its next weight
  // fragments counted as arrived.  For an input gradient with an activation the same memory first holds the forward inputs x
  // (parked by the movers one unit earlier): a consumer reads x, scales its accumulators and overwrites x with the result.
  float* xbuf = (float*)(smem + LDS);
  // the bias vector (zero padded to whole column groups) lives in LDS: the epilogue reads it with LDS loads, which do not queue
  // behind the wave's outstanding stores the way a global load would
  float* lbias = (float*)(smem + LDS + XB);
  if (MODE == 0) {
    for (int i = tid; i < nrg * 64; i += NTHR) lbias[i] = (g.bias != nullptr && i < g.M) ? g.bias[i] : 0.f;
  }

  if (wave >= 4) {
    // ------------------------------------------------------------------------------------- loaders (4-7) and movers (8-11)
    if (TGL_PRIO) __builtin_amdgcn_s_setprio(TGL_PRIO);
    // Two roles of four waves each, so that every wave's vector memory queue holds ONE kind of long-latency work and the
    // compiler's s_waitcnt counts stay exact: vmcnt counts loads and stores in order, and with the tile's stores, the x fetches
    // and the row fetches of several uniform branches in one wave the pass fell back to vmcnt(0) in front of the conversion --
    // every unit waited for the rows it had just requested AND for the tile it had just stored (45 of 150 us at 256 -> 256).
    const int ptid = (tid - 256) & 255, c4 = ptid & 31, r0 = ptid >> 5;  // loader: piece c4 (4 k) of rows r0 + RS j
    constexpr int RS = PT / 32, XS = PT / 64;
    const int xc4 = ptid & 63, xr0 = ptid >> 6;                          // mover: piece xc4 (4 columns) of rows xr0 + XS j
    f32x4 pf0[NPF], pf1[NPF], xs[NXF];
    auto tile_of = [&](int u) __attribute__((always_inline)) -> int64_t { return u < nunits ? blockIdx.x + (int64_t)(u / UPT) * gridDim.x : g.ntiles; };
    // With two chunks per tile both stay in the two LDS buffers, at the same buffer parity, for every further pass over the
    // output columns (a skip layer's 38 extra gradient columns): those units need neither rows nor a conversion.  (Their row
    // fetches are still ISSUED, against an empty buffer: conditional fetches cost the wait-count pass its precision.)
    auto resident = [&](int u) __attribute__((always_inline)) { return NCH == 2 && u < nunits && (u / NCH) % NP >= 1; };
    auto load = [&](f32x4 (&pf)[NPF], int u) {
      if (TGL_ABLATE & 1) return;
      const int64_t m0 = (resident(u) ? g.ntiles : tile_of(u)) * TS;
      const int k = (u % NCH) * KC + c4 * 4;
      const Src2 src{tile_rsrc(g.a.p0, g.a.k0, m0, g.a.rows, g.wp), tile_rsrc(g.a.p1, g.a.k1, m0, g.a.rows, g.wp), g.a.k0, g.a.k1};
      const int k_lo = (u % NCH) * KC;
      if (k_lo + KC <= g.a.k0 || g.a.k1 == 0) {  // (uniform) the chunk lies inside the first source
#pragma unroll
        for (int j = 0; j < NPF; ++j) pf[j] = load_piece<1>(src, r0 + RS * j, k);
      } else if (!STRADDLE || k_lo >= g.a.k0) {  // inside the second
#pragma unroll
        for (int j = 0; j < NPF; ++j) pf[j] = load_piece<2>(src, r0 + RS * j, k);
      } else {
#pragma unroll
        for (int j = 0; j < NPF; ++j) pf[j] = load_piece<0>(src, r0 + RS * j, k);
      }
    };
    // (the activation is a runtime argument: one specialised copy of the loop per kind, chosen once per unit -- with the switch
    // inside, every element carried the sine polynomial next to the LeakyReLU select: 180 instructions per 4 values)
    // (kq: the first k of the thread's pieces in the unit being converted; with a row length that is not a multiple of four,
    // elements at k >= K came along from the next row.  One copy of the loop per (activation, ragged): decided once per unit)
    const bool ragged = (K & 3) != 0;
    auto convert_as = [&](const f32x4 (&pf)[NPF], char* buf, int kq, auto actc, auto raggedc) __attribute__((always_inline)) {
      constexpr int ACT = decltype(actc)::value;
      constexpr bool RAGGED = decltype(raggedc)::value;
      char* d = buf + r0 * PITCH + c4 * 8;
#pragma unroll
      for (int j = 0; j < NPF; ++j) {
        f32x4 v = pf[j];
        if (TGL_ABLATE & 32) v = f32x4{(float)c4, (float)(r0 + j), 1.5f, 2.5f};  // (no wait for the rows)
        if (RAGGED) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = kq + e < K ? v[e] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = tact(v[e], ACT);
        bf16x4 hi, lo;
        split4(v, hi, lo);
        if (TGL_ABLATE & 64) { asm volatile("" ::"v"(hi), "v"(lo)); continue; }  // (no LDS writes)
        *(bf16x4*)(d + RS * j * PITCH) = hi;
        *(bf16x4*)(d + RS * j * PITCH + PLANE) = lo;
      }
    };
    auto convert = [&](const f32x4 (&pf)[NPF], char* buf, int u) __attribute__((always_inline)) {
      if (TGL_ABLATE & 16) return;
      const int kq = (u % NCH) * KC + c4 * 4;
      const int act = MODE == 0 ? g.act : NA_ACT_NONE;
      if (ragged) {
        if (act == NA_ACT_LEAKY_RELU) convert_as(pf, buf, kq, std::integral_constant<int, NA_ACT_LEAKY_RELU>{}, std::true_type{});
        else if (act == NA_ACT_SIN) convert_as(pf, buf, kq, std::integral_constant<int, NA_ACT_SIN>{}, std::true_type{});
        else convert_as(pf, buf, kq, std::integral_constant<int, NA_ACT_NONE>{}, std::true_type{});
      } else {
        if (act == NA_ACT_LEAKY_RELU) convert_as(pf, buf, kq, std::integral_constant<int, NA_ACT_LEAKY_RELU>{}, std::false_type{});
        else if (act == NA_ACT_SIN) convert_as(pf, buf, kq, std::integral_constant<int, NA_ACT_SIN>{}, std::false_type{});
        else convert_as(pf, buf, kq, std::integral_constant<int, NA_ACT_NONE>{}, std::false_type{});
      }
    };
    // forward inputs of (tile, pass) = unit u's: columns 256 pass + 4 xc4 .. + 3 of the concatenated [x0 | x1]
    auto xload = [&](int u) __attribute__((always_inline)) {
      const int64_t m0 = tile_of(u) * TS;
      const int col = 256 * ((u / NCH) % NP) + 4 * xc4;
      const Src2 src{tile_rsrc(g.x0, g.c0, m0, g.a.rows, g.wp), tile_rsrc(g.x1, g.c1, m0, g.a.rows, g.wp), g.c0, g.c1};
      const int c_lo = 256 * ((u / NCH) % NP);
      if (c_lo + 256 <= g.c0 || g.c1 == 0) {
#pragma unroll
        for (int j = 0; j < NXF; ++j) xs[j] = load_piece<1, TGL_X_AUX>(src, xr0 + XS * j, col);
      } else if (c_lo >= g.c0) {
#pragma unroll
        for (int j = 0; j < NXF; ++j) xs[j] = load_piece<2, TGL_X_AUX>(src, xr0 + XS * j, col);
      } else {
#pragma unroll
        for (int j = 0; j < NXF; ++j) xs[j] = load_piece<0>(src, xr0 + XS * j, col);
      }
    };
    auto xstore = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < NXF; ++j) *(f32x4*)(xbuf + (xr0 + XS * j) * XP + 4 * xc4) = xs[j];
    };
    // the finished output tile of unit u's (tile, pass), from LDS to HBM: whole rows, 16 bytes per lane (the thread's pieces are the
    // ones it parks x in, so its own program order is all the synchronisation the shared buffer needs)
    const int ncols = g.c0 + g.c1;
    // (two steps: the tile leaves LDS for registers at once, then goes out in NCH parts, one per unit of the next (tile, pass) --
    // 64 KB in one burst kept every wave of the CU, the loaders' fetches and the consumers' weight refills included, queued
    // behind it for ~9 k cycles of every other unit)
    f32x4 ot[NXF];
    constexpr int P0 = DACT ? TGL_P0D : TGL_P0F;  // pieces stored in the first of a tile's two units, the rest in the second
    auto take_out = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < NXF; ++j) ot[j] = *(const f32x4*)(xbuf + (xr0 + XS * j) * XP + 4 * xc4);
    };
    auto store_out = [&](int u, int part) __attribute__((always_inline)) {  // part < 0: all of it
      if (TGL_ABLATE & 2) return;
      const int64_t m0 = tile_of(u) * TS;
      const int c_lo = 256 * ((u / NCH) % NP);
      const int col = c_lo + 4 * xc4;
      const __amdgpu_buffer_rsrc_t ry0 = tile_rsrc(g.y0, g.c0, m0, g.a.rows, g.wp), ry1 = tile_rsrc(g.y1, g.c1, m0, g.a.rows, g.wp);
      // (uniform) which outputs the pass's 256 columns touch, and whether their rows take aligned 16-byte pieces: a skip layer's
      // [256 | 38] gradient stores its first 256 columns as vectors and only the 38 element by element
      const bool side0 = c_lo < g.c0, side1 = g.c1 > 0 && c_lo + 256 > g.c0;
      const bool vec0 = (g.c0 & 3) == 0, vec1 = ((g.c0 | g.c1) & 3) == 0;
#pragma unroll
      for (int j = 0; j < NXF; ++j) {
        if (part >= 0 && (NCH == 2 ? (j < P0 ? 0 : 1) : j * NCH / NXF) != part) continue;  // (uniform)
        const f32x4 v = ot[j];
        if (side0) {
          if (vec0) {
            const uint32_t o0 = col < g.c0 ? (uint32_t)((xr0 * g.c0 + col) * 4) : OOB;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ry0, o0, XS * j * g.c0 * 4, TGL_ST_AUX);
          } else {
            // rows of a length that is no multiple of four: a piece that lies inside its row still leaves as ONE 16-byte store
            // (4-byte aligned: tools/hw/unaligned_probe.hip), only the piece across the row's end element by element --
            // 38 four-byte stores per row of a skip layer's second gradient kept the mover's queue busy for 10 k cycles
            const bool whole = col + 3 < g.c0;
            const uint32_t o0 = whole ? (uint32_t)((xr0 * g.c0 + col) * 4) : OOB;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ry0, o0, XS * j * g.c0 * 4, TGL_ST_AUX);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const uint32_t p0 = (!whole && col + e < g.c0) ? (uint32_t)((xr0 * g.c0 + col + e) * 4) : OOB;
              const float w = e == 0 ? v[0] : e == 1 ? v[1] : e == 2 ? v[2] : v[3];
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, w), ry0, p0, XS * j * g.c0 * 4, TGL_ST_AUX);
            }
          }
        }
        if (side1) {
          if (vec1) {
            const uint32_t o1 = (col >= g.c0 && col < ncols) ? (uint32_t)((xr0 * g.c1 + col - g.c0) * 4) : OOB;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ry1, o1, XS * j * g.c1 * 4, TGL_ST_AUX);
          } else {
            const bool whole = col >= g.c0 && col + 3 < ncols;
            const uint32_t o1 = whole ? (uint32_t)((xr0 * g.c1 + col - g.c0) * 4) : OOB;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ry1, o1, XS * j * g.c1 * 4, TGL_ST_AUX);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int ce = col + e;
              const uint32_t p1 = (!whole && ce >= g.c0 && ce < ncols) ? (uint32_t)((xr0 * g.c1 + ce - g.c0) * 4) : OOB;
              const float w = e == 0 ? v[0] : e == 1 ? v[1] : e == 2 ? v[2] : v[3];
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, w), ry1, p1, XS * j * g.c1 * 4, TGL_ST_AUX);
            }
          }
        }
      }
    };
    if (wave >= 8) {
      // movers, during unit u: the previous (tile, pass)'s finished output from LDS to HBM if this unit opens a new one; x of the
      // (tile, pass) whose last chunk comes next is parked in the same memory one unit before the consumers read it (its
      // registers are waited for BEFORE the stores are issued: behind them the in-order count would wait for the stores too),
      // and the following one's is requested
      if (xlds) xload(0);
      __syncthreads();
      for (int u = 0; u < nunits; ++u) {
        const int i = u % NCH;
        const bool park = xlds && i == NCH - 2;
        if (wave == 8) TGL_STAMP(2, u, 0);
        if (park) {
#pragma unroll
          for (int j = 0; j < NXF; ++j) asm volatile("" ::"v"(xs[j]));
        }
        if (wave == 8) TGL_STAMP(2, u, 1);
        if (u >= NCH) {
          if (i == 0) take_out();
          store_out(u - 1 - i, i);
        }
        if (wave == 8) TGL_STAMP(2, u, 2);
        if (park) xstore();
        // the next (tile, pass)'s x is requested one unit later, behind the last stores of the tile in flight: with both the tile
        // (16 registers x 4) and the parked x (another 64) alive through every unit the movers spilled
        if (xlds && i == NCH - 1) xload(u + 1);
        if (wave == 8) TGL_STAMP(2, u, 3);
        __syncthreads();
      }
      take_out();
      store_out(nunits - 1, -1);  // (behind the last unit's barrier)
      return;
    }
    // loaders, during unit u: fill the other buffer with unit u + 1, fetch unit u + 1 + PD into the registers this frees.
    // (no conditions around the fetches: units past the end read rows past the batch = zeros)
    // (The wait counts are the compiler's.  A version of this loop issued the fetches by inline asm and waited for them with a
    // hand-counted `s_waitcnt vmcnt(16)`, three units in flight: not faster in the final structure (151-154 against 154-155 us
    // forward, 185-188 against 191 input gradient) and WRONG under another instruction scheduler -- registers the compiler does
    // not know to be in flight get copied: `-mllvm -amdgpu-sched-strategy=max-ilp` produced 1e34.  With the fetches visible to
    // the compiler the three schedulers give the same bits.)
    {
      auto step = [&](f32x4 (&pf)[NPF], int u, char* other) __attribute__((always_inline)) {
        if (wave == 4) TGL_STAMP(1, u, 0);
        if (!resident(u + 1)) convert(pf, other, u + 1);
        if (wave == 4) TGL_STAMP(1, u, 1);
        load(pf, u + 3);
        if (wave == 4) TGL_STAMP(1, u, 2);
        __syncthreads();
        if (wave == 4) TGL_STAMP(1, u, 3);
      };
      load(pf0, 0);
      load(pf1, 1);
      convert(pf0, smem, 0);
      load(pf0, 2);
      __syncthreads();
      for (int u = 0; u < nunits; u += 2) {
        step(pf1, u, smem + BUF);
        if (u + 1 < nunits) step(pf0, u + 1, smem);
      }
    }
    return;
  }

  // -------------------------------------------------------------------------------------------------- consumers
  if (TGL_CPRIO) __builtin_amdgcn_s_setprio(TGL_CPRIO);
  const int ncols = g.c0 + g.c1;
  bf16x8 ring[RD][2];    // [k step of the segment][plane]
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)g.wp, 0, nrg * NCH * 2 * SEG, 0x00020000);
  // Work of this wave in pass p (ng = column groups of the pass, 1..4): with 3 or 4 groups a wave owns a whole group (2 column
  // tiles x 2 sample blocks); with 2 groups a PAIR of waves shares one (a sample block each); with a single group -- a narrow C,
  // or the 38 extra columns of a skip layer's gradient -- the four waves take one (column tile, sample block) each instead of
  // three of them idling through the pass.  The weight stream of a wave: per tile, pass 0 .. NP-1, chunk 0 .. NCH-1, its column
  // tiles in order: one segment of 8 k steps each.
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  struct Work { int R, tmask, bmask; };
  auto work_of = [&](int pass) __attribute__((always_inline)) -> Work {
    const int left = nrg - 4 * pass, ng = left > 4 ? 4 : left;
    if (ng >= 3) return Work{wave_s < ng ? 4 * pass + wave_s : -1, 3, 3};
    if (ng == 2) return Work{4 * pass + (wave_s >> 1), 3, 1 << (wave_s & 1)};
    return Work{4 * pass, 1 << (wave_s >> 1), 1 << (wave_s & 1)};
  };
  auto seg_off = [&](int R, int ch, int t) __attribute__((always_inline)) { return ((R * NCH + ch) * 2 + t) * SEG; };
  auto wfrag = [&](int soff, int i, int p) __attribute__((always_inline)) -> bf16x8 {  // (k step in the scalar offset: 12-bit immediates end at 4095)
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane * 16 + p * 1024, soff + i * 2048, 0));
  };
  // first segment of the wave's stream from pass p on (cyclically: the next tile starts at pass 0 again)
  auto first_seg = [&](int p0) __attribute__((always_inline)) -> int {
    for (int k = 0; k < NP; ++k) {
      const int p = p0 + k >= NP ? p0 + k - NP : p0 + k;
      const Work w = work_of(p);
      if (w.R >= 0) return seg_off(w.R, 0, (w.tmask & 1) ? 0 : 1);
    }
    return -1;
  };
  int cso = __builtin_amdgcn_readfirstlane(first_seg(0));  // the segment the ring's oldest entries belong to
  if (cso >= 0) {
#pragma unroll
    for (int i = 0; i < RD; ++i) { ring[i][0] = wfrag(cso, i, 0); ring[i][1] = wfrag(cso, i, 1); }
  }
  constexpr bool dact = DACT;
  __syncthreads();
  // Two copies of the unit loop: every pass with four whole column groups (C a multiple of 256 columns wide: the masks of
  // work_of() are constants), and the general case.  With the wave-uniform branches of the general case around every fragment
  // read and MFMA group, the compiler's s_waitcnt placement turns conservative at each join -- lgkmcnt(0) in front of the MFMAs,
  // i.e. the NEXT k step's LDS reads waited for at once: 54 instead of 32 cycles per MFMA.
  // One unit's work of a wave, NT column tiles x NB sample blocks of its group R: (2, 2) a whole group, (2, 1) both tiles of
  // block `bsel`, (1, 1) tile `tsel` of block `bsel` -- straight-line code each, accumulators acc[tt][bb] with constant indices;
  // which tile / block they stand for only moves addresses.  (Runtime masks around every fragment read and MFMA group, the
  // first version, cost the compiler's s_waitcnt placement its precision at each join -- lgkmcnt(0) in front of the MFMAs, i.e.
  // the NEXT k step's LDS reads waited for at once: 54 instead of 32 cycles per MFMA -- and spilled.)
  auto unit_body = [&](auto ntc, auto nbc, auto lastc, auto& acc, int u, int R, int tsel, int bsel) __attribute__((always_inline)) {
    constexpr int NT = decltype(ntc)::value, NB = decltype(nbc)::value;
    constexpr bool last = decltype(lastc)::value;  // the (tile, pass)'s last chunk: its own copy of the code, with the epilogue
    const int ch = u % NCH, pass = (u / NCH) % NP;
    const char* brow = smem + (u & 1) * BUF + (lane & 31) * PITCH + (lane >> 5) * 16 + (NB == 2 ? 0 : bsel * 32 * PITCH);
    const int t_first = NT == 2 ? 0 : tsel;
    // Epilogue of column tile t, one quad of columns of one sample block at a time: bias / activation derivative in the
    // accumulator layout -- register 4 q + e of acc[.][.] = column 64 R + 32 t + 8 q + 4 (lane >> 5) + e, sample 32 b +
    // (lane & 31) -- then the quad goes to the output tile in LDS (row-major; x, where it is needed, sits at the very same
    // place and is overwritten by the result).  (TGL_RIDE: with two column tiles the first one's eight items under the second
    // one's k steps, one per step.)
    auto epi_item = [&](const f32x16& a, int t, int q, int bk) __attribute__((always_inline)) {
      const int cl = 64 * (R & 3) + 32 * t + 8 * q + 4 * (lane >> 5);  // column inside the pass
      float* o = xbuf + (32 * bk + (lane & 31)) * XP + cl;
      f32x4 v = {a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]};
      if (MODE == 0) v += *(const f32x4*)(lbias + 256 * pass + cl);
      if (dact) {
        const f32x4 xv = *(const f32x4*)o;
        if (g.act == NA_ACT_SIN) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= tact_grad(xv[e], NA_ACT_SIN);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= tact_grad(xv[e], NA_ACT_LEAKY_RELU);
        }
      }
      *(f32x4*)o = v;
    };
    constexpr bool ride = last && NT == 2 && TGL_RIDE;  // tile 0's epilogue under tile 1's MFMAs
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
      const int t = NT == 2 ? tt : tsel;
      // the segment after this one: the wave's other column tile, the next chunk, the next pass it works in, the next tile
      int nso;
      if (NT == 2 && tt == 0) nso = seg_off(R, ch, 1);
      else if (!last) nso = seg_off(R, ch + 1, t_first);
      else nso = first_seg(pass + 1 == NP ? 0 : pass + 1);
      nso = __builtin_amdgcn_readfirstlane(nso);
      bf16x8 xh[2][NB], xl[2][NB];  // [buffer][block]
      auto xfrag = [&](int ks, int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int bb = 0; bb < NB; ++bb) {
          const char* p = brow + bb * 32 * PITCH + ks * 32;
          xh[slot][bb] = *(const bf16x8*)p;
          xl[slot][bb] = *(const bf16x8*)(p + PLANE);
        }
      };
      xfrag(0, 0);
#pragma unroll
      for (int i = 0; i < KS; ++i) {
        const int cur = i & 1, sl = i % RD;
        if (i + 1 < KS) xfrag(i + 1, cur ^ 1);  // the next k step's fragments under this one's MFMAs
        __builtin_amdgcn_sched_barrier(0);
        const bf16x8 wh = ring[sl][0], wl = ring[sl][1];
#pragma unroll
        for (int bb = 0; bb < NB; ++bb) {
          if (TGL_ABLATE & 4) { acc[tt][bb][0] += (float)wl[0] + (float)xh[cur][bb][0] + (float)wh[1] + (float)xl[cur][bb][1]; continue; }
          acc[tt][bb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, xh[cur][bb], acc[tt][bb], 0, 0, 0);
          acc[tt][bb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xl[cur][bb], acc[tt][bb], 0, 0, 0);
          acc[tt][bb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xh[cur][bb], acc[tt][bb], 0, 0, 0);
        }
        // refill in place, behind the slot's MFMAs: needed RD k steps from now
        if (!(TGL_ABLATE & 8)) {
          const int so = i + RD < KS ? cso : nso, j = (i + RD) % KS;
          ring[sl][0] = wfrag(so, j, 0); ring[sl][1] = wfrag(so, j, 1);
        }
        if (ride && tt == 1 && i < 4 * NB) epi_item(acc[0][NB == 2 ? (i & 1) : 0], 0, NB == 2 ? (i >> 1) : i, NB == 2 ? (i & 1) : bsel);
        __builtin_amdgcn_sched_barrier(0);
      }
      cso = nso;
      if (wave == 0 && tt == NT - 1) TGL_STAMP(0, u, 1);
      if (last && !(ride && tt == 0)) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int bb = 0; bb < NB; ++bb) epi_item(acc[tt][bb], t, q, NB == 2 ? bb : bsel);
      }
    }
  };
  // The chunks of one (tile, pass), with accumulators of their own: the three shapes of work never meet in one live range
  auto chunks = [&](auto ntc, auto nbc, int u0, int R, int tsel, int bsel) __attribute__((always_inline)) {
    constexpr int NT = decltype(ntc)::value, NB = decltype(nbc)::value;
    f32x16 acc[NT][NB];
#pragma unroll
    for (int tt = 0; tt < NT; ++tt)
#pragma unroll
      for (int bb = 0; bb < NB; ++bb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tt][bb][r] = 0.f;
    for (int u = u0; u < u0 + NCH - 1; ++u) {
      if (wave == 0) TGL_STAMP(0, u, 0);
      unit_body(ntc, nbc, std::false_type{}, acc, u, R, tsel, bsel);
      if (wave == 0) TGL_STAMP(0, u, 2);
      __syncthreads();
      if (wave == 0) TGL_STAMP(0, u, 3);
    }
    {
      const int u = u0 + NCH - 1;
      if (wave == 0) TGL_STAMP(0, u, 0);
      unit_body(ntc, nbc, std::true_type{}, acc, u, R, tsel, bsel);
      if (wave == 0) TGL_STAMP(0, u, 2);
      __syncthreads();
      if (wave == 0) TGL_STAMP(0, u, 3);
    }
  };
  // Two copies of the loop: every pass with four whole column groups (C a multiple of 256 columns wide), and the general case
  auto units = [&](auto full) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(full)::value;
    constexpr std::integral_constant<int, 1> one{};
    constexpr std::integral_constant<int, 2> two{};
    for (int u0 = 0; u0 < nunits; u0 += NCH) {
      const int pass = (u0 / NCH) % NP;
      if (FULL) {
        chunks(two, two, u0, 4 * pass + wave_s, 0, 0);
      } else {
        const Work wk = work_of(pass);
        if (wk.R < 0) {
          for (int ch = 0; ch < NCH; ++ch) __syncthreads();
        } else if (wk.bmask == 3) chunks(two, two, u0, wk.R, 0, 0);
        else if (wk.tmask == 3) chunks(two, one, u0, wk.R, 0, wk.bmask >> 1);
        else chunks(one, one, u0, wk.R, wk.tmask >> 1, wk.bmask >> 1);
      }
    }
  };
  if ((nrg & 3) == 0) units(std::true_type{});
  else units(std::false_type{});
}

// Bmat = the [M, K] row-major operand (weights, or their transpose for the input gradient)
// prepacked: the operand as na_train_pack_many left it (round 5) -- no scratch allocation, no pack launch
template <int MODE>
static int launch(Args a, const float* Bmat, int K, hipStream_t st, const char* what, const char* prepacked = nullptr) {
  const bool dact = MODE == 1 && a.act != NA_ACT_NONE;
  a.NCH = (K + KC - 1) / KC;
  if (a.NCH < 2) a.NCH = 2;  // the output tile in LDS is stored (and x parked) in a unit in which the consumers do not touch it
  const int nrg = (a.M + 63) / 64;
  if (nrg > 16) { set_error("%s: more than 1024 output columns", what); return NA_EUNSUPPORTED; }
  const size_t wbytes = (size_t)nrg * a.NCH * 2 * SEG;
  char* wp = nullptr;
  if (prepacked == nullptr) {
    wp = (char*)train_scratch(st, wbytes);
    if (wp == nullptr) return kNoScratch;  // no scratch here: the caller takes the K-staged kernel
    const int64_t nthr = (int64_t)nrg * a.NCH * 2 * 8 * 64;
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, st, Bmat, a.M, K, a.NCH, nrg, wp);
  }
  a.wp = prepacked != nullptr ? prepacked : wp;
  a.ntiles = (a.a.rows + TS - 1) / TS;
  const int grid = a.ntiles < cu_count() ? (int)a.ntiles : cu_count();
  const bool straddle = a.a.k1 > 0 && a.a.k0 % KC != 0;  // (forward only: the gradient's A operand is a single matrix)
  auto k = kernel<MODE, false, false>;
  if constexpr (MODE == 0) { if (straddle) k = kernel<MODE, false, true>; }
  else { if (straddle) { set_error("%s: two-source A operand", what); return NA_EUNSUPPORTED; } if (dact) k = kernel<MODE, true, false>; }
  const int lds = LDS + XB + nrg * 256;
  static std::atomic<uint64_t> done[4];
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t bit = 1ull << (dev & 63);
  int rc = NA_OK;
  if (!(done[2 * dact + straddle].load(std::memory_order_acquire) & bit)) {
    hipError_t e2 = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS + XB + 4096);
    if (e2 != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e2)); rc = NA_EHIP; }
    else done[2 * dact + straddle].fetch_or(bit, std::memory_order_release);
  }
  if (rc == NA_OK) hipLaunchKernelGGL(k, dim3(grid), dim3(NTHR), lds, st, a);
  if (rc != NA_OK) return rc;
  return check_launch(what);
}
}  // namespace lsnt
#if TGL_TRACE
extern "C" int na_debug_tgl_trace(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(na::lsnt::tgl_trace), sizeof(na::lsnt::tgl_trace)) == hipSuccess ? 0 : -1;
}
#endif

// ================================================================================ layer-synchronous TN kernel (round 3)
// Weight gradient  dW[out, in] += dY[N, out]^T . act(X)[N, in]  (+ db = column sums of dY) with the samples as the MFMA's K.
// One workgroup of eight waves owns the whole 256 x 256 gradient -- a wave 64 x 128 of it = 2 x 4 tiles = 128 accumulator
// registers, two waves per SIMD so that one's MFMAs cover the other's loads and conversions -- and every row of dY and X is
// read from HBM exactly once (the K-staged kernel
// splits dW over two workgroups: both read all of X, and 48 KiB per 24 MFMAs cross the vector memory path).  A stage = 32
// samples: their rows are fetched two stages ahead into registers as whole contiguous rows, activated, split into bf16 hi / lo
// and stored ROW-MAJOR ([sample][feature], pitch 576 B) into the other LDS buffer; the MFMA operands -- 8 consecutive samples
// of one feature per lane -- come out of that image through gfx950's transposing LDS read `ds_read_b64_tr_b16`
// (tools/hw/tr_probe.hip: lane i of a 16-lane group receives halfword i & 3 of the 8 bytes addressed by lane 4 j + (i >> 2),
// j = 0..3: with lane i pointing at row i >> 2, columns 4 (i & 3).. of a [4 samples][16 features] block it gets column i).
// Workgroup w reduces its slice of samples into a partial gradient in a workspace; a second kernel sums the partials in a
// fixed order (bit-reproducible without the fixed-point path, and 17 M float atomics cheaper).
#ifndef TGW_ABLATE
#define TGW_ABLATE 0  // timing experiments: 1 no row fetches, 2 no convert / LDS fill, 4 no MFMAs, 8 no fragment reads, 16 no partials
#endif
#ifndef TGW_G_AUX
#define TGW_G_AUX 0  // cache policy of the dY fetches
#endif
#ifndef TGW_X_AUX
#define TGW_X_AUX 0  // ... of the x fetches
#endif
namespace lstn {
constexpr int SS = 32;                // samples per stage
constexpr int PW = 576;               // row pitch of a plane: 144 dwords = 16 mod 64 -> the 4 rows of a transposing read do not collide
constexpr int PLANE = SS * PW;        // 18 KiB
constexpr int STAGE = 4 * PLANE;      // G hi | G lo | X hi | X lo
constexpr int LDS = 2 * STAGE;        // 144 KiB
constexpr int NPC = SS * 64 / 512;    // 16-byte pieces per thread, operand and stage (4)
constexpr int PART = 256 * 256 + 8 * 256;  // floats of one workgroup's partial: dW | 8 row groups of db

struct Args {
  const float* dY;   // [N, out]
  const float* x;    // [N, in]  (one source of the concatenation)
  int out, in, act;
  int64_t N;
  float* part;       // [grid][PART]
  int want_db;
};

typedef short v4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4s* lds_v4s;

// GA / XA: the column count of dY / x is a multiple of 4 (whole aligned 16-byte pieces); otherwise that operand is fetched as
// dwords (the unaligned ones are narrow: 3, 38, 65, 69 columns)
template <bool GA, bool XA>
__global__ __launch_bounds__(512) void kernel(Args g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;  // 64 rows x 128 columns of dW per wave
  const int64_t nst_all = (g.N + SS - 1) / SS;
  const int64_t per = (nst_all + gridDim.x - 1) / gridDim.x;
  const int64_t st0 = blockIdx.x * per;
  const int nst = (int)((st0 + per <= nst_all ? per : (nst_all > st0 ? nst_all - st0 : 0)));
  // 32-column tiles of this wave that hold anything: rows of dW past `out` / columns past `in` are skipped by the whole wave
  const int ni = (g.out - 64 * wm + 31) / 32, nj = (g.in - 128 * wn + 31) / 32;
  const int NI = ni < 0 ? 0 : ni > 2 ? 2 : ni, NJ = nj < 0 ? 0 : nj > 4 ? 4 : nj;

  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};

  const int c4 = tid & 63, r0 = tid >> 6;  // piece c4 (4 columns) of rows r0 + 8 j
  // lane offsets of the thread's piece in row r0 (pieces / elements past the last column read zeros)
  uint32_t og[GA ? 1 : 4], ox[XA ? 1 : 4];
  if (GA) og[0] = 4 * c4 < g.out ? (uint32_t)((r0 * g.out + 4 * c4) * 4) : lsnt::OOB;
  else {
#pragma unroll
    for (int e = 0; e < 4; ++e) og[e & (GA ? 0 : 3)] = 4 * c4 + e < g.out ? (uint32_t)((r0 * g.out + 4 * c4 + e) * 4) : lsnt::OOB;
  }
  if (XA) ox[0] = 4 * c4 < g.in ? (uint32_t)((r0 * g.in + 4 * c4) * 4) : lsnt::OOB;
  else {
#pragma unroll
    for (int e = 0; e < 4; ++e) ox[e & (XA ? 0 : 3)] = 4 * c4 + e < g.in ? (uint32_t)((r0 * g.in + 4 * c4 + e) * 4) : lsnt::OOB;
  }
  f32x4 gs0[NPC], xs0[NPC], gs1[NPC], xs1[NPC];
  auto load = [&](f32x4 (&gs)[NPC], f32x4 (&xs)[NPC], int st) __attribute__((always_inline)) {
    if (TGW_ABLATE & 1) return;
    const int64_t m0 = st < nst ? (st0 + st) * SS : g.N;  // past the slice: an empty buffer
    // (column counts are multiples of 4 here: whole 16-byte pieces; out-of-range pieces return zero)
    const __amdgpu_buffer_rsrc_t rg = lsnt::tile_rsrc(g.dY, g.out, m0, g.N, g.part), rx = lsnt::tile_rsrc(g.x, g.in, m0, g.N, g.part);
#pragma unroll
    for (int j = 0; j < NPC; ++j) {  // rows r0 + 8 j: one lane offset, the row step in the scalar offset
      if constexpr (GA) gs[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, og[0], 8 * j * g.out * 4, TGW_G_AUX));
      else {
#pragma unroll
        for (int e = 0; e < 4; ++e) gs[j][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg, og[e & (GA ? 0 : 3)], 8 * j * g.out * 4, TGW_G_AUX));
      }
      if constexpr (XA) xs[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, ox[0], 8 * j * g.in * 4, TGW_X_AUX));
      else {
#pragma unroll
        for (int e = 0; e < 4; ++e) xs[j][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, ox[e & (XA ? 0 : 3)], 8 * j * g.in * 4, TGW_X_AUX));
      }
    }
  };
  auto convert_as = [&](const f32x4 (&gs)[NPC], const f32x4 (&xs)[NPC], char* buf, auto actc) __attribute__((always_inline)) {
    constexpr int ACT = decltype(actc)::value;
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
      char* d = buf + (r0 + 8 * j) * PW + c4 * 8;
      bf16x4 hi, lo;
      const f32x4 gv = gs[j];
#pragma unroll
      for (int e = 0; e < 4; ++e) bsum[e] += gv[e];
      split4(gv, hi, lo);
      *(bf16x4*)d = hi;
      *(bf16x4*)(d + PLANE) = lo;
      f32x4 xv = xs[j];
#pragma unroll
      for (int e = 0; e < 4; ++e) xv[e] = tact(xv[e], ACT);
      split4(xv, hi, lo);
      *(bf16x4*)(d + 2 * PLANE) = hi;
      *(bf16x4*)(d + 3 * PLANE) = lo;
    }
  };
  auto convert = [&](const f32x4 (&gs)[NPC], const f32x4 (&xs)[NPC], char* buf) __attribute__((always_inline)) {
    if (TGW_ABLATE & 2) return;
    if (g.act == NA_ACT_LEAKY_RELU) convert_as(gs, xs, buf, std::integral_constant<int, NA_ACT_LEAKY_RELU>{});
    else if (g.act == NA_ACT_SIN) convert_as(gs, xs, buf, std::integral_constant<int, NA_ACT_SIN>{});
    else convert_as(gs, xs, buf, std::integral_constant<int, NA_ACT_NONE>{});
  };
  // one operand fragment: 8 consecutive samples (k = 16 ks + 8 (lane >> 5) + e) of feature 32 tile + (lane & 31)
  const int li = lane & 15;
  const int frag_off = (8 * (lane >> 5) + (li >> 2)) * PW + (16 * ((lane >> 4) & 1) + 4 * (li & 3)) * 2;
  auto frag = [&](const char* plane, int ks, int tile) __attribute__((always_inline)) -> bf16x8 {
    const char* p = plane + frag_off + ks * 16 * PW + tile * 64;
    const v4s a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(__attribute__((address_space(3))) char*)p);
    const v4s b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(__attribute__((address_space(3))) char*)(p + 4 * PW));
    typedef short v8s __attribute__((ext_vector_type(8)));
    return __builtin_bit_cast(bf16x8, v8s{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]});
  };
  auto mma = [&](const char* buf) __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 ah[2], al[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (TGW_ABLATE & 8) { ah[i] = al[i] = *(const bf16x8*)(buf + lane * 16); continue; }
        if (i < NI) { ah[i] = frag(buf, ks, 2 * wm + i); al[i] = frag(buf + PLANE, ks, 2 * wm + i); }
      }
#pragma unroll
      for (int jh = 0; jh < 2; ++jh) {  // the four column tiles in two halves: 16 fragment registers less
        bf16x8 bh[2], bl[2];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int j = 2 * jh + jj;
          if (TGW_ABLATE & 8) { bh[jj] = bl[jj] = *(const bf16x8*)(buf + lane * 16); continue; }
          if (j < NJ) { bh[jj] = frag(buf + 2 * PLANE, ks, 4 * wn + j); bl[jj] = frag(buf + 3 * PLANE, ks, 4 * wn + j); }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            const int j = 2 * jh + jj;
            if (i >= NI || j >= NJ) continue;
            if (TGW_ABLATE & 4) { acc[i][j][0] += (float)al[i][0] + (float)bh[jj][1] + (float)ah[i][2] + (float)bl[jj][3]; continue; }
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[jj], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[jj], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[jj], acc[i][j], 0, 0, 0);
          }
      }
    }
  };
  // (no conditions around the fetches: stages past the slice read rows past the batch, which the buffer bounds turn into
  // zeros; with `if (s + 3 < nst)` in front of them the compiler's wait-count pass fell back to vmcnt(0) before every
  // conversion, i.e. waited for the rows it had just requested)
  auto stage = [&](f32x4 (&gs)[NPC], f32x4 (&xs)[NPC], int s) __attribute__((always_inline)) {
    convert(gs, xs, smem + ((s + 1) & 1) * STAGE);
    load(gs, xs, s + 3);
    mma(smem + (s & 1) * STAGE);
    __syncthreads();
  };
  load(gs0, xs0, 0);
  load(gs1, xs1, 1);
  convert(gs0, xs0, smem);
  load(gs0, xs0, 2);
  __syncthreads();
  for (int s = 0; s < nst; s += 2) {
    stage(gs1, xs1, s);
    if (s + 1 < nst) stage(gs0, xs0, s + 1);
  }
  // partial gradient of this workgroup: register r of acc[i][j] = row 64 wm + 32 i + (r & 3) + 8 (r >> 2) + 4 (lane >> 5),
  // column 128 wn + 32 j + (lane & 31): 128 contiguous bytes per row
  float* part = g.part + (int64_t)blockIdx.x * PART;
  if ((TGW_ABLATE & 16) && acc[0][0][0] != 1.2345f) return;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (i >= NI || j >= NJ) continue;  // (the reduction reads rows < out, columns < in only)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        part[(64 * wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 256 + 128 * wn + 32 * j + (lane & 31)] = acc[i][j][r];
    }
  if (g.want_db) {
#pragma unroll
    for (int e = 0; e < 4; ++e) part[256 * 256 + r0 * 256 + 4 * c4 + e] = bsum[e];
  }
}

// dW[row, col] (+)= sum over workgroups; db likewise over workgroups x 8 row groups.  A block = 64 items x 4 quarters of the
// workgroup range: every thread sums its quarter in index order, the quarters are combined in LDS in a fixed order, so the result
// does not depend on timing (and 256 sequential 256-KiB-strided reads per element became 4 x 64 in parallel).
// Round 5: an item = FOUR consecutive columns of one row, read as one 16-byte load per partial (the partial rows have pitch 256:
// always aligned) with sixteen partials in flight -- the per-element order of additions is unchanged, so are the bits.  `overwrite`: dW / db are written, not accumulated into (no zero fill by the caller).
__device__ __forceinline__ void reduce_body(const float* __restrict__ part, int nwg, int out, int in, int ldw,
                                            float* __restrict__ dW, float* __restrict__ db, int overwrite, int block) {
  __shared__ f32x4 red[4][64];
  const int e = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int in4 = (in + 3) >> 2;
  const int nI = out * in4;                 // 4-column items of dW
  const int nB = db != nullptr ? (out + 3) >> 2 : 0;
  const int idx = block * 64 + e;
  const int w0 = (int)((int64_t)nwg * q / 4), w1 = (int)((int64_t)nwg * (q + 1) / 4);
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  int row = 0, col = 0;
  if (idx < nI) {
    row = idx / in4;
    col = 4 * (idx % in4);
    const float* p = part + row * 256 + col;
    int w = w0;
    // (a launch has one block per CU: what bounds it is bytes in flight -- sixteen 16-byte loads per lane before the first add)
    for (; w + 16 <= w1; w += 16) {
      f32x4 v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = *(const f32x4*)(p + (int64_t)(w + u) * PART);
#pragma unroll
      for (int u = 0; u < 16; ++u)
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] += v[u][k];
    }
    for (; w + 4 <= w1; w += 4) {
      const f32x4 a = *(const f32x4*)(p + (int64_t)w * PART), b = *(const f32x4*)(p + (int64_t)(w + 1) * PART);
      const f32x4 c = *(const f32x4*)(p + (int64_t)(w + 2) * PART), d = *(const f32x4*)(p + (int64_t)(w + 3) * PART);
#pragma unroll
      for (int k = 0; k < 4; ++k) s[k] = (((s[k] + a[k]) + b[k]) + c[k]) + d[k];
    }
    for (; w < w1; ++w) {
      const f32x4 a = *(const f32x4*)(p + (int64_t)w * PART);
#pragma unroll
      for (int k = 0; k < 4; ++k) s[k] += a[k];
    }
  } else if (idx < nI + nB) {
    col = 4 * (idx - nI);
    const float* p = part + 256 * 256 + col;
    for (int w = w0; w < w1; ++w) {
      f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const f32x4 a = *(const f32x4*)(p + (int64_t)w * PART + k * 256);
#pragma unroll
        for (int c = 0; c < 4; ++c) t[c] += a[c];
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) s[c] += t[c];
    }
  }
  red[q][e] = s;
  __syncthreads();
  if (q == 0) {
    f32x4 t;
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = ((red[0][e][k] + red[1][e][k]) + red[2][e][k]) + red[3][e][k];
    if (idx < nI) {
      float* o = dW + (int64_t)row * ldw + col;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (col + k < in) o[k] = overwrite ? t[k] : o[k] + t[k];
    } else if (idx < nI + nB) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (col + k < out) db[col + k] = overwrite ? t[k] : db[col + k] + t[k];
    }
  }
}


__global__ __launch_bounds__(256) void reduce_kernel(const float* __restrict__ part, int nwg, int out, int in, int ldw,
                                                     float* __restrict__ dW, float* __restrict__ db, int overwrite) {
  reduce_body(part, nwg, out, in, ldw, dW, db, overwrite, blockIdx.x);
}

// Round 5: the partial gradients of EVERY Linear of a network summed by ONE launch (na_train_reduce_many: the whole-network backward
// of autograd.MlpTrainFn keeps all partial buffers alive and reduces at the end -- 14 launches of 16 us + their gaps per PlainNeRF
// step before).  Entry e owns blocks first[e] .. first[e + 1); same per-element order of additions as reduce_kernel: same bits.
constexpr int kReduceMany = 32;
struct ReduceMany {
  const float* part[kReduceMany];
  float* dW[kReduceMany];
  float* db[kReduceMany];
  int nwg[kReduceMany], out[kReduceMany], in[kReduceMany], ldw[kReduceMany];
  int first[kReduceMany + 1];
  int n;
};
__global__ __launch_bounds__(256) void reduce_many_kernel(ReduceMany rm) {
  int e = 0;
  while (e + 1 < rm.n && (int)blockIdx.x >= rm.first[e + 1]) ++e;  // (block-uniform)
  reduce_body(rm.part[e], rm.nwg[e], rm.out[e], rm.in[e], rm.ldw[e], rm.dW[e], rm.db[e], 1, (int)blockIdx.x - rm.first[e]);
}

// one source of the concatenation: dW[:, 0 .. in) at leading dimension ldw
static int launch(const float* dY, int out, const float* x, int in, int act, int64_t N, float* dW, int ldw, float* db,
                  hipStream_t st, const char* what, int overwrite = 0) {
  Args a{};
  a.dY = dY; a.x = x; a.out = out; a.in = in; a.act = act; a.N = N; a.want_db = db != nullptr;
  const int64_t nst = (N + SS - 1) / SS;
  int grid = lsnt::cu_count();
  if (nst / 4 < grid) grid = (int)(nst / 4 > 0 ? nst / 4 : 1);  // at least 4 stages per workgroup
  const size_t bytes = (size_t)grid * PART * sizeof(float);
  float* part = (float*)train_scratch(st, bytes);
  if (part == nullptr) return lsnt::kNoScratch;
  a.part = part;
  const bool ga = (out & 3) == 0, xa = (in & 3) == 0;
  auto k = ga ? (xa ? kernel<true, true> : kernel<true, false>) : (xa ? kernel<false, true> : kernel<false, false>);
  const int which = 2 * ga + xa;
  static std::atomic<uint64_t> done[4];
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t bit = 1ull << (dev & 63);
  int rc = NA_OK;
  if (!(done[which].load(std::memory_order_acquire) & bit)) {
    hipError_t e2 = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e2 != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e2)); rc = NA_EHIP; }
    else done[which].fetch_or(bit, std::memory_order_release);
  }
  if (rc == NA_OK) {
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), LDS, st, a);
    const int n = out * ((in + 3) / 4) + (db != nullptr ? (out + 3) / 4 : 0);
    hipLaunchKernelGGL(reduce_kernel, dim3((n + 63) / 64), dim3(256), 0, st, part, grid, out, in, ldw, dW, db, overwrite);
  }
  if (rc != NA_OK) return rc;
  return check_launch(what);
}
}  // namespace lstn

// ================================================================================================ narrow outputs (round 5)
// C[N, M] = f(A)[N, 256] . B[M, 256]^T for M <= 128: the out Linears of the networks (256 -> 65, 256 -> 3), the input gradient
// of the encoder inputs (256 -> 38 / 69: the init Linears and the init part of a skip layer).  Their FLOPs are nothing; they are
// one pass over an [N, 256] fp32 tensor.  The three-role kernel above is built around a 256-wide output tile (its movers,
// its weight stream per 64-sample tile) and runs these shapes at 105-130 us per 268 MB, and a skip layer's input gradient
// [256 | 38] as a second pass that costs as much as the first (393 us against 178 for a plain layer).  Here the whole B
// operand of a wave's column tile lives in REGISTERS for the whole launch (16 k steps x (hi | lo) fragments = 128 VGPRs, read
// once from the packed stream of na_train_pack_many), so the kernel is nothing but the row stream: a workgroup of four waves
// fetches a 32-sample tile as whole contiguous rows one tile ahead, activates / splits it into padded bf16 planes in LDS
// (double-buffered, one barrier per tile), wave t multiplies it with column tile t, and the 32 x 32 result leaves in the
// accumulator layout (lane = sample, registers = columns): + bias (forward) or x act'(x) of the matching input (gradient).
// Same three bf16 products per k, fp32 accumulation.
namespace nrw {
constexpr int TS = 32, K = 256;
constexpr int PITCH = K * 2 + 16;      // 132 dwords = 4 mod 64: conflict-free b128 fragment reads
constexpr int PLANE = TS * PITCH;
constexpr int BUF = 2 * PLANE;         // hi | lo
constexpr int LDS = 2 * BUF;           // 66 KiB
constexpr int NP = TS * (K / 4) / 256; // 16-byte pieces per loader thread and tile (8)
#ifndef NRW_AHEAD
#define NRW_AHEAD 4
#endif
#ifndef NRW_ABLATE
#define NRW_ABLATE 0  // experiments only: 1 no MFMAs, 2 no conversion / LDS stash, 4 no result stores / derivative loads, 8 no row fetches
#endif
constexpr int AHEAD = NRW_AHEAD;       // tiles a loader keeps in flight (even; 32 KiB each: the launch is bound by bytes in flight)

struct Args {
  const float* a;      // [N, 256]
  int64_t N;
  const char* wp;      // packed B, at the column group that holds row 0 of this launch (layout of lsnt::pack_kernel, K = 256)
  int M;               // rows of B = output columns (<= 128)
  int act;             // MODE 0: activation applied to A; MODE 1: activation whose derivative (at xd) scales the result
  const float* bias;   // MODE 0 (nullable)
  const float* xd;     // MODE 1: [N, M] pre-activation inputs (unused with NA_ACT_NONE)
  float* y;            // [N, M]
  int64_t ntiles;
};

// Eight waves, two roles (the rule of the three-role kernel above: the loads and stores of one wave retire in order, so a wave
// that stores results must not be the wave whose next row fetch the pipeline waits for).  Waves 0-3 = LOADERS: whole contiguous
// rows of tile i + AHEAD requested (branch-free: out-of-range pieces are dropped by the buffer hardware), tile i activated, split
// and written into plane buffer i & 1.  Waves 4-7 = CONSUMERS of tile i - 1 in the other buffer: wave 4 + t multiplies it with
// column tile t.  One workgroup barrier per tile.
// The result of a tile, 32 rows of M floats, is ONE contiguous run of the output (and so are the derivative's inputs): the
// consumers leave it in an LDS tile in [row][column] order and carry it out together one step later as 16-byte pieces in
// address order -- sixteen 4-byte stores per lane scattered a row pitch apart (the accumulator layout) cost 25-50 us of the
// first version's 70-100 us per launch (ablation, profiles/r05/narrow_bench.log); the derivative's inputs come in the same way.
constexpr int OT = TS * 128 * 4;       // one output (or derivative-input) tile in LDS: 16 KiB
constexpr int LDS_ALL = LDS + 2 * OT + 2 * OT;  // planes | two output tiles | two derivative-input tiles = 130 KiB
template <int MODE>
__global__ __launch_bounds__(512) void kernel(Args g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t per = (g.ntiles + gridDim.x - 1) / gridDim.x;       // a contiguous run of tiles per workgroup
  const int64_t t0 = blockIdx.x * per;
  const int nt = (int)(t0 >= g.ntiles ? 0 : (t0 + per <= g.ntiles ? per : g.ntiles - t0));
  char* const otile = smem + LDS;            // [2][32 * M] floats: results of tile t in otile + (t & 1) * OT
  char* const xtile = smem + LDS + 2 * OT;   // [2][32 * M] floats: derivative inputs of tile t
  const bool deriv = MODE == 1 && g.act != NA_ACT_NONE;
  const int npc = (TS * g.M + 3) >> 2;       // 16-byte pieces of a tile's output run (32 M floats: a multiple of 4)
  if (wave < 4) {
    // ---------------------------------------------------------------- loaders
    const int lt = tid;  // 0..255
    f32x4 pre[AHEAD][NP];
    f32x4 xpre[AHEAD][4];   // the derivative's inputs of the same tile: <= 1024 pieces, 4 per loader thread
    auto fetch = [&](int slot, int i) __attribute__((always_inline)) {
      const int64_t m0 = (t0 + i) * TS;
      const __amdgpu_buffer_rsrc_t rs = lsnt::tile_rsrc(i < nt ? g.a : nullptr, K, m0 < g.N ? m0 : g.N, g.N, g.a);
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        const int p = lt + 256 * j;
        if (NRW_ABLATE & 8) pre[slot][j] = f32x4{1.f, 2.f, 3.f, (float)i};
        else pre[slot][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (uint32_t)(p * 16), 0, 0));
      }
      if (deriv && !(NRW_ABLATE & 4)) {
        const __amdgpu_buffer_rsrc_t rx = lsnt::tile_rsrc(i < nt ? g.xd : nullptr, g.M, m0 < g.N ? m0 : g.N, g.N, g.a);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int p = lt + 256 * j;
          xpre[slot][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, p < npc ? (uint32_t)(p * 16) : lsnt::OOB, 0, 0));
        }
      }
    };
    auto stash = [&](int slot, int i) __attribute__((always_inline)) {
      char* buf = smem + (i & 1) * BUF;
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        const int p = lt + 256 * j;
        f32x4 v = pre[slot][j];
        if (NRW_ABLATE & 2) { if (v[0] == 1.2345f) *(float*)buf = v[1] + v[2] + v[3]; continue; }
        if (MODE == 0 && g.act != NA_ACT_NONE) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = tact(v[e], g.act);
        }
        bf16x4 hi, lo;
        split4(v, hi, lo);
        char* o = buf + (p >> 6) * PITCH + (p & 63) * 8;
        *(bf16x4*)o = hi;
        *(bf16x4*)(o + PLANE) = lo;
      }
      if (deriv && !(NRW_ABLATE & 4)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int p = lt + 256 * j;
          if (p < npc) *(f32x4*)(xtile + (i & 1) * OT + p * 16) = xpre[slot][j];
        }
      }
    };
#pragma unroll
    for (int q = 0; q < AHEAD; ++q) fetch(q, q);
    // step t: tile t stashed, tile t + AHEAD requested (the consumers multiply tile t - 1 and carry out the results of tile t - 2:
    // two more steps at the end drain them)
    for (int i = 0; i < nt + 2; i += AHEAD) {  // (AHEAD tiles per trip: the register slots are compile-time)
#pragma unroll
      for (int q = 0; q < AHEAD; ++q) {
        if (i + q < nt) stash(q, i + q);
        fetch(q, i + q + AHEAD);
        __syncthreads();
      }
    }
    return;
  }
  // ------------------------------------------------------------------ consumers
  const int ct = wave - 4;
  const bool active = 32 * ct < g.M;
  // this wave's column tile: rows 32 ct .. of B = tile (ct & 1) of column group (ct >> 1); K = 256 = 2 chunks x 8 k steps
  bf16x8 bh[16], bl[16];
  if (active) {
    const char* base = g.wp + (size_t)(ct >> 1) * (2 * 2 * lsnt::SEG) + lane * 16;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const char* f = base + (size_t)(((ks >> 3) * 2 + (ct & 1)) * 8 + (ks & 7)) * 2048;
      bh[ks] = *(const bf16x8*)f;
      bl[ks] = *(const bf16x8*)(f + 1024);
    }
  }
  const int s = lane & 31, h = lane >> 5;
  float bias[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = 32 * ct + 8 * (i >> 2) + 4 * h + (i & 3);
    bias[i] = (MODE == 0 && g.bias != nullptr && c < g.M) ? g.bias[c] : 0.f;
  }
  auto consume = [&](int i) __attribute__((always_inline)) {
    if (!active || i < 0 || i >= nt) return;  // (wave-uniform)
    const char* buf = smem + (i & 1) * BUF;
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = bias[q];
    const char* fr = buf + s * PITCH + h * 16;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const bf16x8 xh = *(const bf16x8*)(fr + ks * 32);
      const bf16x8 xl = *(const bf16x8*)(fr + PLANE + ks * 32);
      if (NRW_ABLATE & 1) { acc[ks] += (float)xh[0] + (float)xl[1] + (float)bl[ks][0] + (float)bh[ks][1]; continue; }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[ks], xh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[ks], xl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[ks], xh, acc, 0, 0, 0);
    }
    if ((NRW_ABLATE & 4) && acc[0] != 1.2345f) return;
    float* ot = (float*)(otile + (i & 1) * OT) + s * g.M;
    const float* xt = (const float*)(xtile + (i & 1) * OT) + s * g.M;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int c = 32 * ct + 8 * (q >> 2) + 4 * h + (q & 3);
      if (c < g.M) {
        float v = acc[q];
        if (deriv) v *= tact_grad(xt[c], g.act);
        ot[c] = v;
      }
    }
  };
  // results of tile i (left in the output tile before the previous barrier) -> global, in address order, by all four consumer
  // waves (also those without a column tile): their memory queue holds nothing but these stores
  const int ctid = tid - 256;
  auto carry = [&](int i) __attribute__((always_inline)) {
    if (i < 0 || i >= nt || (NRW_ABLATE & 4)) return;
    const int64_t m0 = (t0 + i) * TS;
    const __amdgpu_buffer_rsrc_t ry = lsnt::tile_rsrc(g.y, g.M, m0, g.N, g.a);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int p = ctid + 256 * j;
      if (p < npc) {
        const u32x4 v = *(const u32x4*)(otile + (i & 1) * OT + p * 16);
        __builtin_amdgcn_raw_buffer_store_b128(v, ry, (uint32_t)(p * 16), 0, 0);
      }
    }
  };
  for (int i = 0; i < nt + 2; i += AHEAD) {
#pragma unroll
    for (int q = 0; q < AHEAD; ++q) {
      carry(i + q - 2);
      consume(i + q - 1);   // (tile t: planes and derivative inputs in buffer t & 1, results into output tile t & 1)
      __syncthreads();
    }
  }
}

static bool wanted(int64_t N, int M, int Kdim) {
  static const bool off = [] { const char* e = getenv("NA_TRAIN_NARROW"); return e != nullptr && strcmp(e, "0") == 0; }();
  return !off && Kdim == K && M >= 1 && M <= 128 && N >= 2048;
}

template <int MODE>
static int launch(Args a, hipStream_t st, const char* what) {
  a.ntiles = (a.N + TS - 1) / TS;
  const int cus = lsnt::cu_count();
  const int grid = a.ntiles < cus ? (int)a.ntiles : cus;
  auto k = kernel<MODE>;
  static std::atomic<uint64_t> done{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t bit = 1ull << (dev & 63);
  if (!(done.load(std::memory_order_acquire) & bit)) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_ALL);
    if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return NA_EHIP; }
    done.fetch_or(bit, std::memory_order_release);
  }
  hipLaunchKernelGGL(k, dim3(grid), dim3(512), LDS_ALL, st, a);
  return check_launch(what);
}
}  // namespace nrw

void* train_scratch(hipStream_t st, size_t bytes) {
  struct Slot { int dev; hipStream_t st; void* p; size_t n; };
  struct Retired { int dev; hipEvent_t ev; void* p; };
  static std::mutex mu;
  static std::vector<Slot> slots;
  static std::vector<Retired> retired;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  // buffers replaced earlier are freed once the kernels that were queued before their replacement have finished (an event
  // recorded on the stream at that moment: ADVICE r05 -- they used to stay alive until process exit)
  for (size_t i = 0; i < retired.size();) {
    if (retired[i].dev == dev && hipEventQuery(retired[i].ev) == hipSuccess) {
      (void)hipEventDestroy(retired[i].ev);
      (void)hipFree(retired[i].p);
      retired[i] = retired.back();
      retired.pop_back();
    } else {
      (void)hipGetLastError();  // (hipErrorNotReady is not an error)
      ++i;
    }
  }
  for (Slot& s : slots)
    if (s.dev == dev && s.st == st) {
      if (s.n >= bytes) return s.p;
      // grow GEOMETRICALLY: a run whose batch grows step by step (crop schedules, coarse then fine sizes) replaces the buffer
      // O(log) times, not once per step
      size_t want = bytes > 2 * s.n ? bytes : 2 * s.n;
      void* p = nullptr;
      if (hipMalloc(&p, want) != hipSuccess) {
        (void)hipGetLastError();
        want = bytes;
        if (hipMalloc(&p, want) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
      }
      hipEvent_t ev;
      if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess && hipEventRecord(ev, st) == hipSuccess)
        retired.push_back(Retired{dev, ev, s.p});
      else (void)hipGetLastError();  // (no event: the old buffer stays alive until process exit, as before)
      s.p = p; s.n = want;
      return p;
    }
  void* p = nullptr;
  const size_t n = bytes < (64u << 20) ? (64u << 20) : bytes;  // (one 256 x 256 layer's partials are 35-69 MB)
  if (hipMalloc(&p, n) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  slots.push_back(Slot{dev, st, p, n});
  return p;
}

int train_reduce_many(int n, const float* const* part, const int* nwg, const int* out, const int* in, const int* ldw, float* const* dW,
                      float* const* db, hipStream_t st) {
  for (int base = 0; base < n; base += lstn::kReduceMany) {
    lstn::ReduceMany rm{};
    rm.n = n - base < lstn::kReduceMany ? n - base : lstn::kReduceMany;
    int blocks = 0;
    for (int e = 0; e < rm.n; ++e) {
      const int i = base + e;
      rm.part[e] = part[i]; rm.dW[e] = dW[i]; rm.db[e] = db[i]; rm.nwg[e] = nwg[i]; rm.out[e] = out[i]; rm.in[e] = in[i]; rm.ldw[e] = ldw[i];
      rm.first[e] = blocks;
      const int items = out[i] * ((in[i] + 3) / 4) + (db[i] != nullptr ? (out[i] + 3) / 4 : 0);
      blocks += (items + 63) / 64;
    }
    rm.first[rm.n] = blocks;
    if (blocks > 0) hipLaunchKernelGGL(lstn::reduce_many_kernel, dim3(blocks), dim3(256), 0, st, rm);
  }
  return NA_OK;
}

int train_reduce_partials(const float* part, int nwg, int out, int in, int ldw, float* dW, float* db, int overwrite, hipStream_t st) {
  const int n = out * ((in + 3) / 4) + (db != nullptr ? (out + 3) / 4 : 0);
  hipLaunchKernelGGL(lstn::reduce_kernel, dim3((n + 63) / 64), dim3(256), 0, st, part, nwg, out, in, ldw, dW, db, overwrite);
  return NA_OK;
}

}  // namespace na

using namespace na;

// The layer-synchronous kernel takes every shape once the batch is worth a persistent launch; NA_TRAIN_GEMM=
// tiled (environment) keeps round 1's K-staged kernels for A/B measurements.
static bool lsnt_wanted(int64_t N, int M) {
  static const bool tiled = [] { const char* e = getenv("NA_TRAIN_GEMM"); return e != nullptr && strcmp(e, "tiled") == 0; }();
  return !tiled && N >= 2048 && M <= 1024;
}

extern "C" {

int na_linear_bf16x3(const float* x0, int in0, const float* x1, int in1, int64_t N, const float* W, const float* b,
                     int out, int pre_act, float* y, void* stream) {
  if (N == 0) return NA_OK;  // empty batch: nothing to read or write (zero-size tensors carry null pointers)
  NA_REQUIRE(x0 && W && y, NA_ENULL, "na_linear_bf16x3: null pointer");
  NA_REQUIRE(in0 >= 1 && in1 >= 0 && out >= 1 && N >= 0, NA_EINVAL, "na_linear_bf16x3: bad shape");
  NA_REQUIRE(in1 == 0 || x1 != nullptr, NA_ENULL, "na_linear_bf16x3: in1>0 needs x1");
  NA_REQUIRE(pre_act >= NA_ACT_NONE && pre_act <= NA_ACT_SIN, NA_EUNSUPPORTED, "na_linear_bf16x3: activation %d", pre_act);
  NtArgs a{};
  a.a = RowSrc{x0, x1, in0, in1, N};
  a.b = RowSrc{W, nullptr, in0 + in1, 0, out};
  a.act = pre_act;
  a.bias = b;
  a.y0 = y;
  a.c0 = out;
  if (lsnt_wanted(N, out)) {
    lsnt::Args l{};
    l.a = a.a; l.M = out; l.act = pre_act; l.bias = b; l.y0 = y; l.c0 = out;
    const int rc = lsnt::launch<0>(l, W, in0 + in1, (hipStream_t)stream, "na_linear_bf16x3");
    if (rc != lsnt::kNoScratch) return rc;
  }
  return dispatch_nt<0>(a, out, (hipStream_t)stream, "na_linear_bf16x3");
}

int na_linear_dgrad_bf16x3(const float* dY, int out, int64_t N, const float* Wt, const float* x0, int in0, const float* x1,
                    int in1, int pre_act, float* g_x0, float* g_x1, void* stream) {
  if (N == 0) return NA_OK;  // empty batch: nothing to read or write (zero-size tensors carry null pointers)
  NA_REQUIRE(dY && Wt && (g_x0 || g_x1), NA_ENULL, "na_linear_dgrad_bf16x3: null pointer");
  NA_REQUIRE(in0 >= 1 && in1 >= 0 && out >= 1 && N >= 0, NA_EINVAL, "na_linear_dgrad_bf16x3: bad shape");
  NA_REQUIRE(pre_act >= NA_ACT_NONE && pre_act <= NA_ACT_SIN, NA_EUNSUPPORTED, "na_linear_dgrad_bf16x3: activation %d", pre_act);
  NA_REQUIRE(pre_act == NA_ACT_NONE || ((g_x0 == nullptr || x0) && (g_x1 == nullptr || in1 == 0 || x1)), NA_ENULL,
             "na_linear_dgrad_bf16x3: the activation derivative needs the forward inputs");
  NtArgs a{};
  a.a = RowSrc{dY, nullptr, out, 0, N};
  a.b = RowSrc{Wt, nullptr, out, 0, in0 + in1};
  a.act = pre_act;
  a.y0 = g_x0;
  a.y1 = in1 > 0 ? g_x1 : nullptr;
  a.x0 = x0;
  a.x1 = x1;
  a.c0 = in0;
  a.c1 = in1;
  if (lsnt_wanted(N, in0 + in1)) {
    lsnt::Args l{};
    l.a = a.a; l.M = in0 + in1; l.act = pre_act; l.y0 = a.y0; l.y1 = a.y1; l.x0 = x0; l.x1 = x1; l.c0 = in0; l.c1 = in1;
    const int rc = lsnt::launch<1>(l, Wt, out, (hipStream_t)stream, "na_linear_dgrad_bf16x3");
    if (rc != lsnt::kNoScratch) return rc;
  }
  return dispatch_nt<1>(a, in0 + in1, (hipStream_t)stream, "na_linear_dgrad_bf16x3");
}

static int wgrad_bf16x3_impl(const float* x0, int in0, const float* x1, int in1, int64_t N, const float* dY, int out,
                             int pre_act, float* dW, float* db, void* stream, int overwrite) {
  if (N == 0) {  // empty batch: nothing to read (zero-size tensors carry null pointers); an overwriting call still defines its outputs
    if (overwrite && dW != nullptr && out >= 1 && in0 + in1 >= 1) {
      (void)hipMemsetAsync(dW, 0, (size_t)out * (in0 + in1) * sizeof(float), (hipStream_t)stream);
      if (db != nullptr) (void)hipMemsetAsync(db, 0, (size_t)out * sizeof(float), (hipStream_t)stream);
    }
    return NA_OK;
  }
  NA_REQUIRE(x0 && dY && dW, NA_ENULL, "na_linear_wgrad_bf16x3: null pointer");
  NA_REQUIRE(in0 >= 1 && in1 >= 0 && out >= 1 && N >= 0, NA_EINVAL, "na_linear_wgrad_bf16x3: bad shape");
  NA_REQUIRE(in1 == 0 || x1 != nullptr, NA_ENULL, "na_linear_wgrad_bf16x3: in1>0 needs x1");
  NA_REQUIRE(pre_act >= NA_ACT_NONE && pre_act <= NA_ACT_SIN, NA_EUNSUPPORTED, "na_linear_wgrad_bf16x3: activation %d", pre_act);
  // The layer-synchronous kernel, one launch per source of the concatenation (dY is read again for the second: the sources of
  // a skip layer are [hidden 256 | encoding 38 or 69]); wider shapes and small batches stay on the K-staged kernel.
  const bool ls0 = lsnt_wanted(N, out) && out <= 256 && in0 <= 256 && in1 <= 256;
  if (ls0) {
    int rc = lstn::launch(dY, out, x0, in0, pre_act, N, dW, in0 + in1, db, (hipStream_t)stream, "na_linear_wgrad_bf16x3", overwrite);
    if (rc != lsnt::kNoScratch) {
      if (rc != NA_OK || in1 == 0) return rc;
      rc = lstn::launch(dY, out, x1, in1, pre_act, N, dW + in0, in0 + in1, nullptr, (hipStream_t)stream, "na_linear_wgrad_bf16x3", overwrite);
      if (rc != lsnt::kNoScratch) return rc;
      // (no scratch for the second source only: the K-staged kernel below would add the first source's columns twice)
      set_error("na_linear_wgrad_bf16x3: stream-ordered scratch allocation failed between the two sources");
      return NA_EHIP;
    }
  }
  if (overwrite) {  // the K-staged kernel accumulates with atomics: define the outputs first
    if (hipMemsetAsync(dW, 0, (size_t)out * (in0 + in1) * sizeof(float), (hipStream_t)stream) != hipSuccess ||
        (db != nullptr && hipMemsetAsync(db, 0, (size_t)out * sizeof(float), (hipStream_t)stream) != hipSuccess)) {
      set_error("na_linear_wgrad_bf16x3_ow: hipMemsetAsync failed");
      return NA_EHIP;
    }
  }
  constexpr int WM = 2, WN = 4, BM = 64 * WM, BN = 64 * WN;
  const int ldw = in0 + in1, in = in0 + in1, col0 = 0;
  TnArgs a{};
  a.dY = dY;
  a.out = out;
  a.x = RowSrc{x0, x1, in0, in1, N};
  a.act = pre_act;
  a.dW = dW + col0;
  a.ldw = ldw;
  a.db = db;
  const int64_t tiles = (int64_t)((out + BM - 1) / BM) * ((in + BN - 1) / BN);
  // ~2 workgroups per CU, slices of at least 512 samples (the 32K-atomic epilogue must stay a small fraction)
  int64_t want = (512 + tiles - 1) / tiles;
  int64_t slice = (N + want - 1) / want;
  if (slice < 512) slice = 512;
  slice = (slice + TK - 1) / TK * TK;
  const int64_t nz = (N + slice - 1) / slice;
  NA_REQUIRE(nz <= 65535, NA_EINVAL, "na_linear_wgrad_bf16x3: N too large");
  a.slice = slice;
  auto k = linear_tn_kernel<WM, WN>;
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes<WM, WN>());
    if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return NA_EHIP; }
    configured = true;
  }
  dim3 grid((out + BM - 1) / BM, (in + BN - 1) / BN, (unsigned)nz);
  const int lds = smem_bytes<WM, WN>();
  int rc;
  const size_t nW = (size_t)out * ldw;
  long long* fix = det_begin(nW + (a.db ? out : 0), (hipStream_t)stream, "na_linear_wgrad_bf16x3", &rc);
  if (rc != NA_OK) return rc;
  a.fixW = fix ? fix + col0 : nullptr;
  a.fixb = fix ? fix + nW : nullptr;
  hipLaunchKernelGGL(k, grid, dim3(64 * WM * WN), lds, (hipStream_t)stream, a);
  if (fix != nullptr) {
    if ((rc = det_finish(fix, nW, dW, (hipStream_t)stream, "na_linear_wgrad_bf16x3")) != NA_OK) return rc;
    if (a.db != nullptr) return det_finish(fix + nW, (size_t)out, db, (hipStream_t)stream, "na_linear_wgrad_bf16x3");
    return NA_OK;
  }
  return check_launch("na_linear_wgrad_bf16x3");
}

int na_linear_wgrad_bf16x3(const float* x0, int in0, const float* x1, int in1, int64_t N, const float* dY, int out,
                           int pre_act, float* dW, float* db, void* stream) {
  return wgrad_bf16x3_impl(x0, in0, x1, in1, N, dY, out, pre_act, dW, db, stream, 0);
}

int na_linear_wgrad_bf16x3_ow(const float* x0, int in0, const float* x1, int in1, int64_t N, const float* dY, int out,
                              int pre_act, float* dW, float* db, void* stream) {
  return wgrad_bf16x3_impl(x0, in0, x1, in1, N, dY, out, pre_act, dW, db, stream, 1);
}

// One source of a concatenated input on its own: dW[:, 0:in) at leading dimension ldw WRITTEN (db too when given).  The wide source
// of a skip layer goes through na_linear_bwd_bf16x3_pk (train_bwd.hip), its narrow source [N, 38 / 69] through here.
int na_linear_wgrad_bf16x3_cols(const float* x, int in, int64_t N, const float* dY, int out, int pre_act, float* dW, int ldw,
                                float* db, void* stream) {
  NA_REQUIRE(in >= 1 && in <= 256 && out >= 1 && out <= 256 && N >= 0 && ldw >= in, NA_EINVAL, "na_linear_wgrad_bf16x3_cols: bad shape");
  NA_REQUIRE(pre_act >= NA_ACT_NONE && pre_act <= NA_ACT_SIN, NA_EUNSUPPORTED, "na_linear_wgrad_bf16x3_cols: activation %d", pre_act);
  NA_REQUIRE(dW != nullptr, NA_ENULL, "na_linear_wgrad_bf16x3_cols: null pointer");
  if (N == 0) {
    for (int r = 0; r < out; ++r) (void)hipMemsetAsync(dW + (size_t)r * ldw, 0, (size_t)in * sizeof(float), (hipStream_t)stream);
    if (db != nullptr) (void)hipMemsetAsync(db, 0, (size_t)out * sizeof(float), (hipStream_t)stream);
    return NA_OK;
  }
  NA_REQUIRE(x && dY, NA_ENULL, "na_linear_wgrad_bf16x3_cols: null pointer");
  NA_REQUIRE(lsnt_wanted(N, out), NA_EUNSUPPORTED, "na_linear_wgrad_bf16x3_cols: this batch runs the K-staged kernel (na_train_gemm_packed_ok)");
  const int rc = lstn::launch(dY, out, x, in, pre_act, N, dW, ldw, db, (hipStream_t)stream, "na_linear_wgrad_bf16x3_cols", 1);
  if (rc == lsnt::kNoScratch) { set_error("na_linear_wgrad_bf16x3_cols: stream-ordered scratch allocation failed"); return NA_EHIP; }
  return rc;
}

// ---- round 5: the B operands of a whole training step packed by one launch, and the GEMMs that take them --------------------
int na_train_gemm_packed_ok(int64_t N, int M) { return (M >= 1 && lsnt_wanted(N, M)) ? 1 : 0; }

size_t na_train_packed_bytes(int M, int K) { return (M >= 1 && M <= 1024 && K >= 1) ? lsnt::packed_bytes(M, K) : 0; }

int na_train_pack_many(int n, const float* const* W, const int* M, const int* K, const int* ld, const int* transposed,
                       void* const* dst, void* stream) {
  NA_REQUIRE(n >= 0, NA_EINVAL, "na_train_pack_many: n < 0");
  if (n == 0) return NA_OK;
  NA_REQUIRE(W && M && K && ld && transposed && dst, NA_ENULL, "na_train_pack_many: null pointer");
  for (int base = 0; base < n; base += lsnt::kPackMany) {
    lsnt::PackMany pm{};
    pm.n = n - base < lsnt::kPackMany ? n - base : lsnt::kPackMany;
    long long total = 0;
    for (int e = 0; e < pm.n; ++e) {
      const int i = base + e;
      NA_REQUIRE(W[i] && dst[i], NA_ENULL, "na_train_pack_many: null matrix %d", i);
      NA_REQUIRE(M[i] >= 1 && M[i] <= 1024 && K[i] >= 1 && ld[i] >= (transposed[i] ? M[i] : K[i]), NA_EINVAL,
                 "na_train_pack_many: bad shape of matrix %d (M=%d K=%d ld=%d)", i, M[i], K[i], ld[i]);
      NA_REQUIRE(((uintptr_t)dst[i] & 15) == 0, NA_EINVAL, "na_train_pack_many: destination %d is not 16-byte aligned", i);
      pm.W[e] = W[i]; pm.dst[e] = (char*)dst[i]; pm.M[e] = M[i]; pm.K[e] = K[i]; pm.ld[e] = ld[i]; pm.trans[e] = transposed[i] != 0;
      pm.NCH[e] = lsnt::pack_nch(K[i]);
      pm.first[e] = total;
      total += (long long)((M[i] + 63) / 64) * pm.NCH[e] * 2 * 8 * 64;
    }
    pm.first[pm.n] = total;
    hipLaunchKernelGGL(lsnt::pack_many_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pm);
  }
  return check_launch("na_train_pack_many");
}

int na_linear_bf16x3_pk(const float* x0, int in0, const float* x1, int in1, int64_t N, const void* w_packed, const float* b,
                        int out, int pre_act, float* y, void* stream) {
  if (N == 0) return NA_OK;
  NA_REQUIRE(x0 && w_packed && y, NA_ENULL, "na_linear_bf16x3_pk: null pointer");
  NA_REQUIRE(in0 >= 1 && in1 >= 0 && out >= 1 && N >= 0, NA_EINVAL, "na_linear_bf16x3_pk: bad shape");
  NA_REQUIRE(in1 == 0 || x1 != nullptr, NA_ENULL, "na_linear_bf16x3_pk: in1>0 needs x1");
  NA_REQUIRE(pre_act >= NA_ACT_NONE && pre_act <= NA_ACT_SIN, NA_EUNSUPPORTED, "na_linear_bf16x3_pk: activation %d", pre_act);
  NA_REQUIRE(lsnt_wanted(N, out), NA_EUNSUPPORTED, "na_linear_bf16x3_pk: this batch / width runs the K-staged kernel: call "
             "na_linear_bf16x3 (na_train_gemm_packed_ok says which)");
  if (train_fwd_wanted(N, out, in0, in1, pre_act))  // a 256 wide layer (hidden, skip [256 | 38 / 69]): W resident in registers, one pass (train_fwd.hip)
    return train_fwd_launch(x0, in0, x1, in1, N, w_packed, b, out, pre_act, y, (hipStream_t)stream, "na_linear_bf16x3_pk");
  if (in1 == 0 && nrw::wanted(N, out, in0)) {  // a narrow output (256 -> 65 / 3): the row-stream kernel
    nrw::Args n{};
    n.a = x0; n.N = N; n.wp = (const char*)w_packed; n.M = out; n.act = pre_act; n.bias = b; n.y = y;
    return nrw::launch<0>(n, (hipStream_t)stream, "na_linear_bf16x3_pk");
  }
  lsnt::Args l{};
  l.a = RowSrc{x0, x1, in0, in1, N};
  l.M = out; l.act = pre_act; l.bias = b; l.y0 = y; l.c0 = out;
  return lsnt::launch<0>(l, nullptr, in0 + in1, (hipStream_t)stream, "na_linear_bf16x3_pk", (const char*)w_packed);
}

int na_linear_dgrad_bf16x3_pk(const float* dY, int out, int64_t N, const void* wt_packed, const float* x0, int in0,
                              const float* x1, int in1, int pre_act, float* g_x0, float* g_x1, void* stream) {
  if (N == 0) return NA_OK;
  NA_REQUIRE(dY && wt_packed && (g_x0 || g_x1), NA_ENULL, "na_linear_dgrad_bf16x3_pk: null pointer");
  NA_REQUIRE(in0 >= 1 && in1 >= 0 && out >= 1 && N >= 0, NA_EINVAL, "na_linear_dgrad_bf16x3_pk: bad shape");
  NA_REQUIRE(pre_act >= NA_ACT_NONE && pre_act <= NA_ACT_SIN, NA_EUNSUPPORTED, "na_linear_dgrad_bf16x3_pk: activation %d", pre_act);
  NA_REQUIRE(pre_act == NA_ACT_NONE || ((g_x0 == nullptr || x0) && (g_x1 == nullptr || in1 == 0 || x1)), NA_ENULL,
             "na_linear_dgrad_bf16x3_pk: the activation derivative needs the forward inputs");
  NA_REQUIRE(lsnt_wanted(N, in0 + in1), NA_EUNSUPPORTED, "na_linear_dgrad_bf16x3_pk: this batch / width runs the K-staged kernel: call "
             "na_linear_dgrad_bf16x3 (na_train_gemm_packed_ok says which)");
  // narrow gradients on the row-stream kernel (round 5): a narrow single source (the init Linears: 256 -> 38 / 69), and the
  // narrow SECOND source of a skip layer [256 | 38], whose 256 wide columns then run as a plain single-pass layer (the
  // packed W^T is column-group major: rows 0..255 are its first four groups, the narrow rows start at group in0 / 64)
  if (in1 == 0 && nrw::wanted(N, in0, out)) {
    nrw::Args n{};
    n.a = dY; n.N = N; n.wp = (const char*)wt_packed; n.M = in0; n.act = pre_act; n.xd = x0; n.y = g_x0;
    return nrw::launch<1>(n, (hipStream_t)stream, "na_linear_dgrad_bf16x3_pk");
  }
  if (in1 > 0 && (in0 & 63) == 0 && nrw::wanted(N, in1, out)) {
    if (g_x1 != nullptr) {
      nrw::Args n{};
      n.a = dY; n.N = N; n.wp = (const char*)wt_packed + (size_t)(in0 / 64) * (2 * 2 * lsnt::SEG); n.M = in1; n.act = pre_act; n.xd = x1;
      n.y = g_x1;
      const int rc = nrw::launch<1>(n, (hipStream_t)stream, "na_linear_dgrad_bf16x3_pk");
      if (rc != NA_OK || g_x0 == nullptr) return rc;
    }
    lsnt::Args l{};
    l.a = RowSrc{dY, nullptr, out, 0, N};
    l.M = in0; l.act = pre_act; l.y0 = g_x0; l.y1 = nullptr; l.x0 = x0; l.x1 = nullptr; l.c0 = in0; l.c1 = 0;
    return lsnt::launch<1>(l, nullptr, out, (hipStream_t)stream, "na_linear_dgrad_bf16x3_pk", (const char*)wt_packed);
  }
  lsnt::Args l{};
  l.a = RowSrc{dY, nullptr, out, 0, N};
  l.M = in0 + in1; l.act = pre_act; l.y0 = g_x0; l.y1 = in1 > 0 ? g_x1 : nullptr; l.x0 = x0; l.x1 = x1; l.c0 = in0; l.c1 = in1;
  return lsnt::launch<1>(l, nullptr, out, (hipStream_t)stream, "na_linear_dgrad_bf16x3_pk", (const char*)wt_packed);
}

}  // extern "C"
