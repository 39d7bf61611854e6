// Training step, round 5: input gradient AND weight gradient of one Linear in ONE pass over its two tensors.
//
//   g_x[N, 256]  = (dY[N, out] . W[out, 0:256]) * act'(x)          (src/neural_blocks.py:288-296 differentiated, w.r.t. the input)
//   dW[out, 0:256] = dY^T . act(x),   db = column sums of dY         (... w.r.t. the parameters)
//
// Both read the same two [N, .] tensors -- dY and the forward input x.  As two launches (lsnt::kernel<1> + lstn::kernel in
// train_gemm.hip) a 256 x 256 layer moves 805 + 537 MB at N = 262 144; here 805 (dY twice through L2, once from HBM).
// What made one kernel impossible before: the weight gradient of a 256 x 256 layer is 256 KiB of
// accumulators, the packed W^T another 256 KiB -- the register file of a CU is 512 KiB.  The split that fits: a workgroup owns a
// sample slice AND one HALF of the input columns (128 of x / g_x / dW's columns).  Eight waves, two per SIMD:
//   waves 0-3 ("dgrad"): wave t holds W^T's fragments of column tile 4 h + t for ALL k (16 k steps x (hi | lo) = 128 registers,
//             read once per launch from the stream na_train_pack_many packed) and multiplies every stage with them: 48 MFMAs, the
//             32 x 32 result goes into an LDS tile;
//   waves 4-7 ("wgrad"): wave u owns rows 64 u .. 64 u + 63 of dW's half (2 x 4 tiles = 128 accumulator registers), the samples
//             are the MFMA's k: 48 MFMAs per stage, operands through the transposing LDS read like lstn::kernel.
// ALL eight waves fetch the stage ONE stage ahead (whole contiguous rows: dY 1 KiB, the x half 512 B; a stage lasts ~2.8 us, and a
// second prefetch set spilled: a scratch reload is a vmcnt(0)), activate / split it into
// bf16 hi | lo planes in LDS, and carry out the previous stage's g_x tile: the thread that fetched x[s, c..c+3] keeps act'(x) in
// registers and finishes exactly those four elements, so the forward input is read ONCE for both gradients and its derivative
// never touches LDS.  Neither role refills anything from L2 inside the loop (the weight stream of lsnt::kernel -- 256 KiB per
// 64-sample tile, four times the HBM bytes of the tile -- is what bounded the standalone input gradient).
// One LDS image serves both MFMA shapes: LDS row R of a stage holds sample rho(R), rho = swap of the bit fields [1:0] and [3:2]
// (an involution); row pitch = 4 mod 64 dwords.  The transposing reads of four consecutive samples then hit rows 4 apart = 16
// banks apart (conflict-free, and the next four samples are ONE row further: constant offsets), and the 16 lanes of a
// ds_read_b128 group read 16 rows that are distinct mod 16 (conflict-free).  The input gradient's MFMA sees the samples in the
// permuted order and un-permutes when it writes its tile.
// The two halves of a slice are workgroups b and b + 8: same XCD (round-robin dispatch), same time -> dY's second read is an L2 hit
// (842 MB of HBM traffic per launch measured against 805 + 34 of partials).  The two roles of a SIMD run in ANTIPHASE (the wgrad
// wave multiplies first, then converts).  A narrow source (<= 128 columns: an init Linear's 38 / 69, the second source of a skip
// layer) runs the same kernel with ONE workgroup per slice, dword accesses for unaligned rows and idle tiles past its columns.
// Same arithmetic as the two kernels it replaces (three bf16 products per k, fp32 accumulation); the input gradient's k order is
// unchanged, the weight gradient's partials are per slice (128 instead of 256 per layer) and summed by lstn::reduce_kernel in
// a fixed order: bit-reproducible, last-bit differences against the two-launch path.
#include <atomic>
#include <type_traits>
#include <stdlib.h>
#include <string.h>
#include "common.h"
#include "train_shared.h"

namespace na {
#ifndef TBW_EXP
#define TBW_EXP 0     // experiments: 1 non-temporal stores of g_x, 2 non-temporal row fetches
#endif
#ifndef TBW_ABLATE
#define TBW_ABLATE 0  // timing experiments: 1 no row fetches, 2 no convert / LDS fill, 4 no MFMAs, 8 no g_x stores, 16 no partials
#endif
namespace lsbw {
constexpr int SS = 32;                 // samples per stage
constexpr int GP = 528;                // row pitch of a dY plane: 132 dwords = 4 mod 64
constexpr int GPLANE = SS * GP;
constexpr int XP = 272;                // row pitch of an act(x) plane (128 columns): 68 dwords = 4 mod 64
constexpr int XPLANE = SS * XP;
constexpr int STAGE = 2 * GPLANE + 2 * XPLANE;  // G hi | G lo | X hi | X lo = 50 KiB
constexpr int OP = 528;                // row pitch of the g_x tile (128 floats + 4)
constexpr int OT = SS * OP;
constexpr int LDS = 2 * STAGE + 2 * OT;  // 133 KiB
constexpr int PART = 256 * 256 + 8 * 256;  // = lstn::PART (the reduction kernel's layout)

struct Args {
  const float* dY;   // [N, out]
  const float* x;    // [N, ldx]: the forward input (columns 0..255 used)
  const char* wp;    // packed W^T (rows = input columns, k = out), layout of lsnt::pack_many_kernel
  float* gx;         // [N, ldx]
  const float* add;  // [N, ldx] or null: g_x = (dY . W) * act'(x) + add (another consumer's gradient of the same tensor: a skip
                     // layer's second source and the init Linear both produce d/d init; not in the FULL instantiations)
  float* part;       // [nsl][PART]
  int out, act, ldx, nsl, xcd_map, want_db;
  int in, nhalf;     // columns of x a workgroup owns (128 = one half of a 256 wide source; a narrow source: all of its <= 128) | 2 or 1
  int64_t N;
};

typedef short v4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4s* lds_v4s;

__device__ __forceinline__ int rho(int r) { return (r & 16) | ((r & 3) << 2) | ((r >> 2) & 3); }

// GA: dY's column count is a multiple of 4 (whole aligned 16-byte pieces); otherwise it is fetched as dwords (65, 3 columns).
// ACT: the Linear's input activation (compile-time: a runtime switch around the conversion is a branch around memory waits).
// FULL: out == 256 and a 256 wide source (no guards around k steps / row and column tiles).
// XA: x / g_x rows are whole aligned 16-byte pieces; otherwise (a narrow source of 38 / 69 columns) dword accesses.
// No conditional around ANY global access, no spill in the loop: a scratch reload is a vmcnt(0), i.e. a wait for the rows that
// were just requested (the first version: 40 spilled registers, fetch time + MFMA time added up exactly: 381 us = 205 + 176).
template <bool GA, int ACT, bool FULL, bool XA>
__global__ __launch_bounds__(512) void kernel(Args g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int slice, h;
  if (g.nhalf == 1) { h = 0; slice = blockIdx.x; }
  else if (g.xcd_map) { const int p = blockIdx.x >> 3; h = p & 1; slice = (p >> 1) * 8 + (blockIdx.x & 7); }
  else { h = blockIdx.x & 1; slice = blockIdx.x >> 1; }
  const int64_t nst_all = (g.N + SS - 1) / SS;
  const int64_t per = (nst_all + g.nsl - 1) / g.nsl;
  const int64_t st0 = slice * per;
  const int nst = (int)((st0 + per <= nst_all ? per : (nst_all > st0 ? nst_all - st0 : 0)));
  char* const otile = smem + 2 * STAGE;

  // ---- fetch / convert / finish: every thread, both roles
  const int c4 = tid & 63, r0 = tid >> 6;    // dY: piece c4 (4 columns) of rows r0 + 8 j, j = 0..3
  const int xc = tid & 31, xr0 = tid >> 5;   // x half: piece xc of rows xr0 + 16 j, j = 0..1
  uint32_t og[GA ? 1 : 4];
  og[0] = 4 * c4 < g.out ? (uint32_t)((r0 * g.out + 4 * c4) * 4) : lsnt::OOB;
  if (!GA) {
#pragma unroll
    for (int e = 1; e < 4; ++e) og[e & (GA ? 0 : 3)] = 4 * c4 + e < g.out ? 1u : 0u;   // element e of the piece belongs to this row
  }
  const uint32_t ox = (uint32_t)((xr0 * g.ldx + 128 * h + 4 * xc) * 4);   // (XA: 4 xc < in always -- 128 columns, or whole pieces dropped below)
  uint32_t oxe[XA ? 1 : 4];
  if (XA) oxe[0] = 4 * xc < g.in ? ox : lsnt::OOB;
  else {
#pragma unroll
    for (int e = 0; e < 4; ++e) oxe[e & (XA ? 0 : 3)] = 4 * xc + e < g.in ? ox + 4 * e : lsnt::OOB;
  }
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  f32x4 gs[4], xs[2], d0[2], d1[2], as[2];
  // rows of stage st (past the slice: an empty buffer -- every piece reads zeros, every store is dropped)
  auto stage_rsrc = [&](const float* base, int ld, int st) __attribute__((always_inline)) {
    const int64_t m0 = (st >= 0 && st < nst) ? (st0 + st) * SS : g.N;
    return lsnt::tile_rsrc(base, ld, m0, g.N, g.part);
  };
  auto load = [&](int st) __attribute__((always_inline)) {
    if (TBW_ABLATE & 1) return;
    const __amdgpu_buffer_rsrc_t rg = stage_rsrc(g.dY, g.out, st), rx = stage_rsrc(g.x, g.ldx, st);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // ONE 16-byte load per piece whatever the row length: raw-buffer loads of 16 bytes need only 4-byte alignment and are
      // range-checked per dword (profiles/r03/unaligned_probe.log); a piece that crosses the end of its row (65, 3 columns) brings
      // elements of the NEXT row along, which the conversion zeroes (og[1..3] keep the per-element validity)
      gs[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, og[0], 8 * j * g.out * 4, (TBW_EXP & 2) ? 2 : 0));
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      // (one 16-byte load per piece here too; the elements of the next row a piece of a 38 / 69 wide source brings along are
      // zeroed at conversion.  The STORES of such a source stay dwords: a 16-byte store across the row end would clobber.)
      xs[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, oxe[0], 16 * j * g.ldx * 4, (TBW_EXP & 2) ? 2 : 0));
    }
    if constexpr (!FULL) {  // the addend of the stage that is FINISHED in the next step (this fetch is for stage st = step + 2, the
      // next step finishes stage step = st - 2); no pointer: an empty buffer, zeros
      const __amdgpu_buffer_rsrc_t ra = stage_rsrc(g.add, g.ldx, g.add != nullptr ? st - 2 : -1);
#pragma unroll
      for (int j = 0; j < 2; ++j) as[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, oxe[0], 16 * j * g.ldx * 4, 0));
    }
  };
  // LDS rows of this thread's pieces (sample s sits in row rho(s))
  const int gro = rho(r0) * GP + c4 * 8;            // sample r0 + 8 j -> row rho(r0) + 2 (j & 1) + 16 (j >> 1)
  const int xro = rho(xr0) * XP + xc * 8;           // sample xr0 + 16 j -> row rho(xr0) + 16 j
  auto convert = [&](f32x4 (&d)[2], char* buf) __attribute__((always_inline)) {
    if (TBW_ABLATE & 2) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      char* p = buf + gro + (2 * (j & 1) + 16 * (j >> 1)) * GP;
      bf16x4 hi, lo;
      f32x4 gv = gs[j];
      if constexpr (!GA) {
#pragma unroll
        for (int e = 1; e < 4; ++e) gv[e] = og[e & (GA ? 0 : 3)] ? gv[e] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) bsum[e] += gv[e];
      split4(gv, hi, lo);
      *(bf16x4*)p = hi;
      *(bf16x4*)(p + GPLANE) = lo;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      char* p = buf + 2 * GPLANE + xro + 16 * j * XP;
      f32x4 xv = xs[j], dv;
      if constexpr (!XA) {
#pragma unroll
        for (int e = 1; e < 4; ++e) xv[e] = oxe[e & (XA ? 0 : 3)] != lsnt::OOB ? xv[e] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) { dv[e] = tact_grad(xv[e], ACT); xv[e] = tact(xv[e], ACT); }
      d[j] = dv;
      bf16x4 hi, lo;
      split4(xv, hi, lo);
      *(bf16x4*)p = hi;
      *(bf16x4*)(p + XPLANE) = lo;
    }
  };
  // g_x of stage st: the tile the dgrad waves left (rows = samples, 128 columns) x act'(x) of the pieces this thread fetched
  auto finish = [&](const f32x4 (&d)[2], int st) __attribute__((always_inline)) {
    if (TBW_ABLATE & 8) return;
    const __amdgpu_buffer_rsrc_t ry = stage_rsrc(g.gx, g.ldx, st);
    const char* ot = otile + (st & 1) * OT + xr0 * OP + xc * 16;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      f32x4 v = *(const f32x4*)(ot + 16 * j * OP);
      if (ACT != NA_ACT_NONE) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= d[j][e];
      }
      if constexpr (!FULL) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += as[j][e];
      }
      // (the row step in the VECTOR offset, soffset 0: with an SGPR soffset the compiler inserts no wait between a 16-byte store
      // and a VALU write of its data registers, and gfx950 needs one -- build.check_store_data_overwrite, tools/hw/store_soffset_hazard.hip)
      if constexpr (XA) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), ry, oxe[0] + (uint32_t)(16 * j * g.ldx * 4), 0, (TBW_EXP & 1) ? 2 : 0);
      else {
        const u32x4 u = __builtin_bit_cast(u32x4, v);  // (the whole vector: a bit cast of v[e] in an unrolled loop reads element 0 four times)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          __builtin_amdgcn_raw_buffer_store_b32(u[e], ry, oxe[e & (XA ? 0 : 3)] + (uint32_t)(16 * j * g.ldx * 4), 0, 0);
      }
    }
  };

  load(0);
  convert(d0, smem);
  load(1);

  if (wave < 4) {
    // ------------------------------------------------------------------------------------------------ dgrad waves
    const int T = 4 * h + wave;  // 32-column tile of W^T's rows (= input columns)
    bf16x8 bh[16], bl[16];
    {
      // (round 6) a narrow source's W^T has ceil(in / 64) row groups in the packed stream: a wave whose 32-column tile lies past
      // the source (it idles below: nks = 0) must not fetch "its" fragments -- they are past the end of the stream, a plain
      // global read that faulted once the allocation behind the stream was unmapped (in0 = 64 / 38: waves 2, 3).  Such waves
      // read the first tile's fragments (valid memory, never used).
      const int Tl = (FULL || 32 * wave < g.in) ? T : 0;
      const char* base = g.wp + (size_t)(Tl >> 1) * (2 * 2 * lsnt::SEG) + lane * 16;
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const char* f = base + (size_t)(((ks >> 3) * 2 + (Tl & 1)) * 8 + (ks & 7)) * 2048;
        bh[ks] = *(const bf16x8*)f;
        bl[ks] = *(const bf16x8*)(f + 1024);
      }
    }
    const int nks = FULL ? 16 : (32 * wave < g.in ? (g.out + 15) >> 4 : 0);  // (a narrow source: tiles past its columns idle)
    const int n = lane & 31, hh = lane >> 5;
    const int fro = n * GP + hh * 16;                           // LDS row n = sample rho(n)
    const int oto = rho(n) * OP + (32 * wave + 4 * hh) * 4;     // its row of the g_x tile
    auto mma = [&](const char* buf, char* ot) __attribute__((always_inline)) {
      f32x16 acc;
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q] = 0.f;
      const char* fr = buf + fro;
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        if (FULL || ks < nks) {  // (wave-uniform: out = 65 / 3 have 5 / 1 k steps)
          const bf16x8 xh = *(const bf16x8*)(fr + ks * 32);
          const bf16x8 xl = *(const bf16x8*)(fr + GPLANE + ks * 32);
          if (TBW_ABLATE & 4) { acc[ks] += (float)xh[0] + (float)xl[1] + (float)bl[ks][0] + (float)bh[ks][1]; continue; }
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[ks], xh, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[ks], xl, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[ks], xh, acc, 0, 0, 0);
        }
      }
      // register q = column 32 wave + 8 (q >> 2) + 4 hh + (q & 3) of sample rho(n)
#pragma unroll
      for (int k = 0; k < 4; ++k) *(f32x4*)(ot + oto + 32 * k) = f32x4{acc[4 * k], acc[4 * k + 1], acc[4 * k + 2], acc[4 * k + 3]};
    };
    __syncthreads();
    for (int s = 0; s < nst; s += 2) {
      finish(d1, s - 1);
      convert(d1, smem + STAGE);
      load(s + 2);
      __builtin_amdgcn_sched_barrier(0);  // (the scheduler sank the fetches below the MFMAs: no latency hidden)
      mma(smem, otile);
      __syncthreads();
      if (s + 1 < nst) {
        finish(d0, s);
        convert(d0, smem);
        load(s + 3);
        __builtin_amdgcn_sched_barrier(0);
        mma(smem + STAGE, otile + OT);
        __syncthreads();
      }
    }
    if ((nst - 1) & 1) finish(d1, nst - 1); else finish(d0, nst - 1);
  } else {
    // ------------------------------------------------------------------------------------------------ wgrad waves
    const int wm = wave - 4;  // rows 64 wm .. of dW; all four column tiles of the half
    const int ni = (g.out - 64 * wm + 31) / 32;
    const int NI = FULL ? 2 : (ni < 0 ? 0 : ni > 2 ? 2 : ni);
    const int nj = (g.in + 31) / 32;
    const int NJ = FULL ? 4 : (nj > 4 ? 4 : nj);
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // one operand fragment: 8 consecutive samples (k = 16 ks + 8 (lane >> 5) + e) of feature 32 tile + (lane & 31): samples
    // 16 ks + 8 hh + 4 q + j sit in LDS row 16 ks + 4 j + 2 hh + q, so lane li of a 16-lane group points at row 4 (li >> 2) + 2 hh
    // (q = 0) and the second read is one row further
    const int li = lane & 15;
    const int rowl = 4 * (li >> 2) + 2 * (lane >> 5);
    const int colb = (16 * ((lane >> 4) & 1) + 4 * (li & 3)) * 2;
    const int goff = rowl * GP + colb + 2 * wm * 64, xoff = rowl * XP + colb;
    auto fragG = [&](const char* plane, int ks, int tile) __attribute__((always_inline)) -> bf16x8 {
      const char* p = plane + goff + ks * 16 * GP + tile * 64;
      const v4s a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(__attribute__((address_space(3))) char*)p);
      const v4s b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(__attribute__((address_space(3))) char*)(p + GP));
      typedef short v8s __attribute__((ext_vector_type(8)));
      return __builtin_bit_cast(bf16x8, v8s{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]});
    };
    auto fragX = [&](const char* plane, int ks, int tile) __attribute__((always_inline)) -> bf16x8 {
      const char* p = plane + xoff + ks * 16 * XP + tile * 64;
      const v4s a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(__attribute__((address_space(3))) char*)p);
      const v4s b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(__attribute__((address_space(3))) char*)(p + XP));
      typedef short v8s __attribute__((ext_vector_type(8)));
      return __builtin_bit_cast(bf16x8, v8s{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]});
    };
    auto mma = [&](const char* buf) __attribute__((always_inline)) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 ah[2], al[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
          if (FULL || i < NI) { ah[i] = fragG(buf, ks, i); al[i] = fragG(buf + GPLANE, ks, i); }
#pragma unroll
        for (int jh = 0; jh < 2; ++jh) {  // the four column tiles in two halves: 16 fragment registers less
          bf16x8 bh[2], bl[2];
#pragma unroll
          for (int jj = 0; jj < 2; ++jj)
            if (FULL || 2 * jh + jj < NJ) {
              bh[jj] = fragX(buf + 2 * GPLANE, ks, 2 * jh + jj);
              bl[jj] = fragX(buf + 2 * GPLANE + XPLANE, ks, 2 * jh + jj);
            }
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
              const int j = 2 * jh + jj;
              if (!FULL && (i >= NI || j >= NJ)) continue;
              if (TBW_ABLATE & 4) { acc[i][j][0] += (float)al[i][0] + (float)bh[jj][1] + (float)ah[i][2] + (float)bl[jj][3]; continue; }
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[jj], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[jj], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[jj], acc[i][j], 0, 0, 0);
            }
        }
      }
    };
    __syncthreads();
    for (int s = 0; s < nst; s += 2) {
      // (the two roles of a SIMD in antiphase: this wave's MFMAs cover the dgrad wave's fetch / conversion / stores and vice versa)
      mma(smem);
      __builtin_amdgcn_sched_barrier(0);
      finish(d1, s - 1);
      convert(d1, smem + STAGE);
      load(s + 2);
      __syncthreads();
      if (s + 1 < nst) {
        mma(smem + STAGE);
        __builtin_amdgcn_sched_barrier(0);
        finish(d0, s);
        convert(d0, smem);
        load(s + 3);
        __syncthreads();
      }
    }
    if ((nst - 1) & 1) finish(d1, nst - 1); else finish(d0, nst - 1);
    // partial gradient of this slice, columns of this half: register r of acc[i][j] = row 64 wm + 32 i + (r & 3) + 8 (r >> 2) +
    // 4 (lane >> 5), column 128 h + 32 j + (lane & 31)
    if (!((TBW_ABLATE & 16) && acc[0][0][0] != 1.2345f)) {
      float* part = g.part + (int64_t)slice * PART;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (!FULL && (i >= NI || j >= NJ)) continue;  // (the reduction reads rows < out, columns < in only)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            part[(64 * wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 256 + 128 * h + 32 * j + (lane & 31)] = acc[i][j][r];
        }
    }
  }
  if (g.want_db && h == 0) {
    float* part = g.part + (int64_t)slice * PART;
#pragma unroll
    for (int e = 0; e < 4; ++e) part[256 * 256 + r0 * 256 + 4 * c4 + e] = bsum[e];
  }
}

template <bool GA, bool FULL, bool XA>
static auto pick_act(int act) -> void (*)(Args) {
  if (act == NA_ACT_LEAKY_RELU) return kernel<GA, NA_ACT_LEAKY_RELU, FULL, XA>;
  if (act == NA_ACT_SIN) return kernel<GA, NA_ACT_SIN, FULL, XA>;
  return kernel<GA, NA_ACT_NONE, FULL, XA>;
}

static bool wanted(int64_t N, int out, int in0) {
  static const bool off = [] { const char* e = getenv("NA_TRAIN_FUSED_BWD"); return e != nullptr && strcmp(e, "0") == 0; }();
  return !off && (in0 == 256 || (in0 >= 1 && in0 <= 128)) && out >= 1 && out <= 256 && N >= 8192;
}

// sample slices of a launch (= partial gradients the reduction sums)
static int slices(int64_t N, bool wide) {
  const int64_t nst = (N + SS - 1) / SS;
  int nsl = wide ? lsnt::cu_count() / 2 : lsnt::cu_count();
  if (nst / 4 < nsl) nsl = (int)(nst / 4 > 0 ? nst / 4 : 1);  // at least 4 stages per slice
  return nsl;
}

// workspace: nsl x PART floats from the caller (torch's allocator: stream-ordered, microseconds), or null = the per-stream scratch
// of train_shared.h (hipMallocAsync here cost 230 us of HOST time per call: tools/train_host_time.py)
// reduce = false: the partial gradients stay in `workspace` (required then) for na_train_reduce_many
static int launch(Args a, float* dW, int ldw, float* db, int overwrite, float* workspace, hipStream_t st, const char* what, bool reduce = true) {
  const bool wide = a.ldx == 256;  // (a narrow source: ONE workgroup per slice owns all its <= 128 columns)
  a.nhalf = wide ? 2 : 1;
  a.in = wide ? 128 : a.ldx;
  const int nsl = slices(a.N, wide);
  a.nsl = nsl;
  a.xcd_map = (nsl % 8) == 0;
  a.want_db = db != nullptr;
  float* part = workspace;
  if (part == nullptr) {
    part = (float*)train_scratch(st, (size_t)nsl * PART * sizeof(float));
    if (part == nullptr) return lsnt::kNoScratch;
  }
  a.part = part;
  const bool ga = (a.out & 3) == 0, full = a.out == 256 && wide, xa = (a.ldx & 3) == 0;
  auto k = full ? pick_act<true, true, true>(a.act)
                : xa ? (ga ? pick_act<true, false, true>(a.act) : pick_act<false, false, true>(a.act))
                     : (ga ? pick_act<true, false, false>(a.act) : pick_act<false, false, false>(a.act));
  const int which = (full ? 0 : 1 + 2 * xa + ga) * 3 + (a.act == NA_ACT_LEAKY_RELU ? 1 : a.act == NA_ACT_SIN ? 2 : 0);
  static std::atomic<uint64_t> done[15];
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
    hipLaunchKernelGGL(k, dim3(a.nhalf * nsl), dim3(512), LDS, st, a);
    if (reduce) rc = train_reduce_partials(part, nsl, a.out, a.ldx, ldw, dW, db, overwrite, st);
  }
  if (rc != NA_OK) return rc;
  return check_launch(what);
}
}  // namespace lsbw
}  // namespace na

using namespace na;

extern "C" {

int na_linear_bwd_fused_ok(int64_t N, int out, int in0) { return lsbw::wanted(N, out, in0) ? 1 : 0; }

// g_x0[N, in0] = (dY . W[:, c0:c0+in0]) * act'(x0);  dW[out, 0:in0] (leading dimension ldw) and db WRITTEN (not accumulated).
// in0 = 256 (a wide source: two workgroups per sample slice, one per column half) or in0 <= 128 (a narrow source -- an init
// Linear's 38 / 69 columns, the second source of a skip layer: one workgroup per slice).  wt_packed: W^T ([in, out]) as
// na_train_pack_many packs it, AT the column group that holds the source's first row (row c0, a multiple of 64:
// na_train_packed_row_offset(c0, out) bytes into the stream).  The sources of a concatenation are separate calls.
size_t na_train_packed_row_offset(int row0, int K) { return (size_t)(row0 / 64) * (size_t)((K + lsnt::KC - 1) / lsnt::KC < 2 ? 2 : (K + lsnt::KC - 1) / lsnt::KC) * 2 * lsnt::SEG; }

size_t na_linear_bwd_workspace_bytes(int64_t N, int in0) {
  return N > 0 ? (size_t)lsbw::slices(N, in0 == 256) * lsbw::PART * sizeof(float) : 0;
}

// partials in that workspace (= its bytes / the size of one partial): what na_train_reduce_many takes as nwg_i
int na_linear_bwd_partial_count(int64_t N, int in0) { return N > 0 ? lsbw::slices(N, in0 == 256) : 0; }

int na_linear_bwd_bf16x3_pk(const float* dY, int out, int64_t N, const void* wt_packed, const float* x0, int in0, int pre_act,
                            float* g_x0, float* dW, int ldw, float* db, void* workspace, void* stream) {
  NA_REQUIRE((in0 == 256 || (in0 >= 1 && in0 <= 128)) && out >= 1 && out <= 256 && N >= 0 && ldw >= in0, NA_EINVAL,
             "na_linear_bwd_bf16x3_pk: bad shape (in0 = %d, out = %d)", in0, out);
  NA_REQUIRE(pre_act >= NA_ACT_NONE && pre_act <= NA_ACT_SIN, NA_EUNSUPPORTED, "na_linear_bwd_bf16x3_pk: activation %d", pre_act);
  NA_REQUIRE(dW != nullptr, NA_ENULL, "na_linear_bwd_bf16x3_pk: null pointer");
  if (N == 0) {  // empty batch: the outputs are still defined
    for (int r = 0; r < out; ++r) (void)hipMemsetAsync(dW + (size_t)r * ldw, 0, (size_t)in0 * sizeof(float), (hipStream_t)stream);
    if (db != nullptr) (void)hipMemsetAsync(db, 0, (size_t)out * sizeof(float), (hipStream_t)stream);
    return NA_OK;
  }
  NA_REQUIRE(dY && wt_packed && x0 && g_x0, NA_ENULL, "na_linear_bwd_bf16x3_pk: null pointer");
  NA_REQUIRE(lsbw::wanted(N, out, in0), NA_EUNSUPPORTED, "na_linear_bwd_bf16x3_pk: this batch runs the two-launch path "
             "(na_linear_bwd_fused_ok says which)");
  lsbw::Args a{};
  a.dY = dY; a.x = x0; a.wp = (const char*)wt_packed; a.gx = g_x0; a.out = out; a.act = pre_act; a.ldx = in0; a.N = N;
  const int rc = lsbw::launch(a, dW, ldw, db, 1, (float*)workspace, (hipStream_t)stream, "na_linear_bwd_bf16x3_pk");
  if (rc == lsnt::kNoScratch) { set_error("na_linear_bwd_bf16x3_pk: stream-ordered scratch allocation failed"); return NA_EHIP; }
  return rc;
}

// The same pass WITHOUT the reduction: the partial gradients (na_linear_bwd_workspace_bytes(N, in0) bytes = slices x 67 584 floats) stay
// in `workspace` (required); na_train_reduce_many sums the partials of many Linears in one launch.  want_db: the bias gradient's
// partial sums are produced too (reduce it by passing db there).  g_add (nullable, [N, in0]): added to the input gradient before
// it is stored -- another consumer's gradient of the same tensor (a skip layer's second source and the init Linear both produce
// d/d init: the sum costs no launch); narrow sources (in0 <= 128) only.
int na_linear_bwd_partials_bf16x3_pk(const float* dY, int out, int64_t N, const void* wt_packed, const float* x0, int in0, int pre_act,
                                     float* g_x0, const float* g_add, int want_db, void* workspace, void* stream) {
  NA_REQUIRE((in0 == 256 || (in0 >= 1 && in0 <= 128)) && out >= 1 && out <= 256 && N >= 1, NA_EINVAL,
             "na_linear_bwd_partials_bf16x3_pk: bad shape (in0 = %d, out = %d)", in0, out);
  NA_REQUIRE(pre_act >= NA_ACT_NONE && pre_act <= NA_ACT_SIN, NA_EUNSUPPORTED, "na_linear_bwd_partials_bf16x3_pk: activation %d", pre_act);
  NA_REQUIRE(dY && wt_packed && x0 && g_x0 && workspace, NA_ENULL, "na_linear_bwd_partials_bf16x3_pk: null pointer");
  NA_REQUIRE(lsbw::wanted(N, out, in0), NA_EUNSUPPORTED, "na_linear_bwd_partials_bf16x3_pk: this batch runs the two-launch path");
  lsbw::Args a{};
  NA_REQUIRE(g_add == nullptr || in0 <= 128, NA_EUNSUPPORTED, "na_linear_bwd_partials_bf16x3_pk: g_add is for narrow sources (in0 <= 128)");
  a.dY = dY; a.x = x0; a.wp = (const char*)wt_packed; a.gx = g_x0; a.add = g_add; a.out = out; a.act = pre_act; a.ldx = in0; a.N = N;
  float dummy_db = 0.f;  // (launch() reads db only as "wanted or not" when it does not reduce)
  return lsbw::launch(a, nullptr, in0, want_db ? &dummy_db : nullptr, 1, (float*)workspace, (hipStream_t)stream,
                      "na_linear_bwd_partials_bf16x3_pk", false);
}

// dW_i[out_i, 0:in_i) (leading dimension ldw_i) and db_i[out_i] (nullable) WRITTEN = the sum of nwg_i partials of 67 584 floats each
// (the layout of na_linear_bwd_partials_bf16x3_pk's workspace), i = 0 .. n-1, in one launch per 32 entries; fixed order of the
// additions (the bits of na_linear_bwd_bf16x3_pk's own reduction).
int na_train_reduce_many(int n, const float* const* part, const int* nwg, const int* out, const int* in, const int* ldw, float* const* dW,
                         float* const* db, void* stream) {
  NA_REQUIRE(n >= 0, NA_EINVAL, "na_train_reduce_many: n < 0");
  if (n == 0) return NA_OK;
  NA_REQUIRE(part && nwg && out && in && ldw && dW && db, NA_ENULL, "na_train_reduce_many: null pointer");
  for (int i = 0; i < n; ++i) {
    NA_REQUIRE(part[i] && dW[i], NA_ENULL, "na_train_reduce_many: null entry %d", i);
    NA_REQUIRE(nwg[i] >= 1 && out[i] >= 1 && out[i] <= 256 && in[i] >= 1 && in[i] <= 256 && ldw[i] >= in[i], NA_EINVAL,
               "na_train_reduce_many: bad shape of entry %d", i);
  }
  const int rc = train_reduce_many(n, part, nwg, out, in, ldw, dW, db, (hipStream_t)stream);
  if (rc != NA_OK) return rc;
  return check_launch("na_train_reduce_many");
}

}  // extern "C"
