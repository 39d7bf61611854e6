// Register-resident fused SkipConnMLP engine for gfx950 (CDNA4).
//
// Design (DESIGN.md "K5"):
//  * one wave owns 32 samples for the whole network; the product is computed TRANSPOSED,
//        C^T[feature, sample] = W[feature, k] . X^T[k, sample]
//    with v_mfma_f32_32x32x16_bf16, weights as the A operand and activations as the B operand.
//    The C layout of a 32x32 tile (lane = sample, regs = features) is, up to a fixed permutation of
//    k inside each 16-chunk, exactly the B-operand layout of the next layer, so activations never
//    leave the register file: no LDS round trip, no cross-lane shuffle.  The k permutation is folded
//    into the weight packing (mlp_pack.hip).
//  * weights are pre-packed into MFMA A-fragments (1 KiB = 64 lanes x 16 B each) and DMA-streamed
//    global -> LDS with global_load_lds_dwordx4, one "tile" (32 output features x all K) at a time,
//    double-buffered; every wave of the workgroup consumes the same tile for its own samples.
//  * precision: bf16 (1 product) or 2-way split bf16 (hi+lo, 3 products: hi*hi + hi*lo + lo*hi) with
//    fp32 accumulation -- the latter is fp32-class (SURVEY 7 "hard parts").
#pragma once
#include "common.h"

// Ablation switches for tools/ablate.py (timing experiments only; results are WRONG when any bit is set):
//  1 no barrier/wait   2 identity activation   4 no MFMA   8 one LDS read per tile   16 no weight DMA   32 no hash
#ifndef NA_ABLATE
#define NA_ABLATE 0
#endif
// NA_TRACE: waves 0 and NWAVES/2 of workgroup 0 log (s_memtime << 8 | event id) into LDS during their first two
// passes; the render kernel dumps the log to its workspace (tools/trace.py).  Timing experiments only.
#ifndef NA_TRACE
#define NA_TRACE 0
#endif

namespace na {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int kHidden = 256;
constexpr int kHC = kHidden / 16;  // hidden K-chunks per layer
constexpr int kMaxTiles = 112;

// A packed MLP = [1-KiB header][weight stream].  Header (written by na_mlp_pack): uint32 ntiles, then per
// tile {uint32 off_blocks, uint32 nblk}: tile t = `nblk` 1-KiB blocks at block offset `off` of the stream.
constexpr int kHeaderBytes = 1024;
constexpr int kMaxTilesPerMlp = 8 * 9 + 3;

// Up to two packed MLPs consumed back to back (tiles [0, split) from the first, the rest from the second).
struct TileTab {
  const uint32_t* hdr0;
  const uint32_t* hdr1;
  int32_t ntiles;  // tiles per pass (both MLPs)
  int32_t split;
};

template <int PREC>
struct Frag {
  bf16x8 hi;
  bf16x8 lo;  // dead (eliminated) when PREC == 0
};

// ------------------------------------------------------------------------------------------------ activations
// sin_cw (Cody-Waite + degree-9 polynomial, 1.6e-7) lives in common.h
// Hardware sine (v_sin_f32 takes revolutions, valid for |r| <= 256 -> reduce with v_fract first).  Used by the
// bf16 fast mode only, where its ~1e-6 absolute error is far below the bf16 operand rounding.
__device__ __forceinline__ float sin_hw(float x) {
  return __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(x * 0.15915494309189535f));
}

// Parity mode: the same hardware sine on a revolution count reduced with a two-constant 1/(2 pi) (the fma keeps the
// product exact before the subtraction: |r| <= 0.5 with ~1e-9 rev of reduction error for |x| up to 1e3).  Measured
// against the Cody-Waite + polynomial sin_cw it replaced here: same end-to-end L-inf (4.4e-6 vs 4.7e-6 on the bench
// tile), 10 fewer VALU instructions per activation, bf16x3 renderer 276 -> 290 Msamples/s.
__device__ __forceinline__ float sin_hw2(float x) {
  const float q = rintf(x * 0.15915493667125702f);
  float r = fmaf(x, 0.15915493667125702f, -q);
  r = fmaf(x, 6.4206382432985265e-09f, r);
  return __builtin_amdgcn_sinf(r);
}

#ifndef NA_F16X_FAST_SIN
#define NA_F16X_FAST_SIN 1  // the f16x mode takes the three-instruction sine (v_mul, v_fract, v_sin) of the fast modes: measured the
                            // same L-inf as the exact reduction (1.6e-5 golden / 9e-6 bench weights), 128 fewer VALU per sine epilogue
#endif
// precision traits: two operand planes (hi + lo) per fragment; IEEE-half elements in the 16-bit containers
template <int PREC> constexpr bool kTwoPlane = PREC == NA_PREC_BF16X3 || PREC == NA_PREC_F16X;
template <int PREC> constexpr bool kHalfElem = PREC == NA_PREC_F16 || PREC == NA_PREC_F16X;

template <int ACT, int PREC = NA_PREC_BF16X3>
__device__ __forceinline__ float act_apply(float v) {
  // leaky_relu(v) = max(v, 0.01 v) = median(v, 0.01 v, +big): v_med3_f32 needs no canonicalising v_max
  if constexpr ((NA_ABLATE & 2) != 0) return v;
  // (f16 operands: the upper bound doubles as the clamp to the largest finite half, so a large pre-activation becomes 65504
  // instead of +inf -> NaN downstream; the negative side is safe down to v = -6.5e6)
  if constexpr (ACT == NA_ACT_LEAKY_RELU) return __builtin_amdgcn_fmed3f(v, v * 0.01f, kHalfElem<PREC> ? 65504.0f : 3.0e38f);
  else if constexpr (ACT == NA_ACT_SIN) return (kTwoPlane<PREC> && !(PREC == NA_PREC_F16X && NA_F16X_FAST_SIN)) ? sin_hw2(v) : sin_hw(v);
  else return v;
}

// ------------------------------------------------------------------------------------------------ fragments
// One 16-bit operand element.  NA_PREC_F16 keeps IEEE half values in the same 16-byte containers (bit patterns in
// bf16x8); only these two conversions, the element pair packing of the epilogue and the MFMA builtin differ.
// CLAMP: values beyond the half range saturate at +-65504 instead of becoming +-inf (NaN after a sine or 0 * inf); bf16 has
// fp32's range and needs nothing.  The hidden-layer epilogues pass CLAMP = false: their activations bound the value already
// (sine; LeakyReLU through the med3 above).
template <int PREC, bool CLAMP = true>
__device__ __forceinline__ __bf16 to_elem(float v) {
  if constexpr (kHalfElem<PREC>) {
    if constexpr (CLAMP) v = __builtin_amdgcn_fmed3f(v, -65504.0f, 65504.0f);
    return __builtin_bit_cast(__bf16, (_Float16)v);
  } else return (__bf16)v;
}
template <int PREC>
__device__ __forceinline__ float from_elem(__bf16 h) {
  if constexpr (kHalfElem<PREC>) return (float)__builtin_bit_cast(_Float16, h);
  else return (float)h;
}

template <int PREC>
__device__ __forceinline__ Frag<PREC> make_frag(const float (&v)[8]) {
  Frag<PREC> f;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    __bf16 h = to_elem<PREC>(v[e]);
    f.hi[e] = h;
    if constexpr (kTwoPlane<PREC>) f.lo[e] = to_elem<PREC, false>(v[e] - from_elem<PREC>(h));
  }
  return f;
}

template <int PREC>
__device__ __forceinline__ float frag_value(const Frag<PREC>& f, int e) {
  float v = from_elem<PREC>(f.hi[e]);
  if constexpr (kTwoPlane<PREC>) v = v + from_elem<PREC>(f.lo[e]);
  return v;
}

template <int PREC>
__device__ __forceinline__ void pin_frag(Frag<PREC>& f) {
  asm volatile("" : "+v"(f.hi));
  if constexpr (kTwoPlane<PREC>) asm volatile("" : "+v"(f.lo));
}

// act() applied in place to an input fragment (the skip connection re-enters through the activation,
// src/neural_blocks.py:291-293).
template <int PREC, int ACT>
__device__ __forceinline__ void frag_activate(Frag<PREC>& f) {
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = act_apply<ACT, PREC>(frag_value<PREC>(f, e));
  f = make_frag<PREC>(v);
  pin_frag<PREC>(f);
}

// ------------------------------------------------------------------------------------------------ weight stream
__device__ __forceinline__ void glds16(const void* g, char* lds) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

// The stream is cyclic: every pass (one group of samples) consumes tiles 0..ntiles-1 in order, and the
// workgroup knows up front how many passes it will run, so the prefetcher manages itself.
//
// Three LDS slots, ONE barrier per tile placed in the MIDDLE of the tile's MFMA stream (mid_sync): at mid-tile t
// every wave waits for its own DMA pieces of tile t+1 (issued a full tile earlier), the barrier makes tile t+1
// resident for everybody and proves that everybody is done with tile t-1, whose slot then receives the DMA of
// tile t+2.  Tile boundaries therefore carry no barrier and no DMA issue: the MFMA stream runs across them.
// The (offset, size) table of the tiles is copied into LDS once so the main loop has no scalar-memory loads
// (an outstanding SMEM load would turn every LDS wait into lgkmcnt(0)).
// LDS slots that fit next to the tile table: 3 (mid-tile barrier) when possible, else 2 (every wave takes the
// barrier at the end of its tile; the DMA then reuses the slot that was just consumed).
constexpr int slots_for(int precision, int ni_max) {
  return 3 * (((kHC + ni_max) * (precision == NA_PREC_BF16X3 ? 2 : 1) + 1) * 1024) + 8 * kMaxTiles + 24 * 1024 <= 160 * 1024
             ? 3 : 2;
}

template <int NWAVES, int SLOTS = 3>
struct WeightStream {
  static constexpr int kSlots = SLOTS;
  static constexpr bool kSplitRoles = SLOTS == 3 && NWAVES >= 8;  // second half of the waves: late barrier, no DMA
  const char* base0;
  const char* base1;
  char* lds;            // kSlots buffers of buf_bytes each, then uint2 table[ntiles]
  uint32_t buf_bytes;
  int wave, lane;
  int ntiles;
  int issued, total;    // tiles issued so far / to issue over the whole kernel
  int next_t;           // tile index (within a pass) of the next tile to issue
  int cur_slot;
  const char* cur;      // LDS address of the tile being consumed
  bool late;            // this wave takes barrier t at the end of tile t (else in the middle of tile t)
#if NA_TRACE
  static constexpr int kTraceMax = 1400;
  unsigned long long* tlog;
  int tpos;
  __device__ __forceinline__ void mark(int id) {
    if (tlog != nullptr && tpos < kTraceMax) {
      tlog[tpos++] = (__builtin_amdgcn_s_memtime() << 8) | (unsigned long long)id;
    }
  }
#else
  __device__ __forceinline__ void mark(int) {}
#endif

  __device__ __forceinline__ void issue_next(int slot) {
    const uint2 e = ((const uint2*)(lds + kSlots * buf_bytes))[next_t];
    const uint32_t off = __builtin_amdgcn_readfirstlane(e.x);
    const int nblk = __builtin_amdgcn_readfirstlane(e.y);
    const char* src = ((off >> 31) ? base1 : base0) + (size_t)(off & 0x7fffffffu) * 1024 + lane * 16;
    char* dst = lds + slot * buf_bytes;
    // An LDS-DMA instruction costs ~100-200 issue cycles during which its wave cannot issue MFMAs.  With two
    // waves per SIMD (waves w and w + NWAVES/2) only the first one issues DMA; its partner (the `late` half, which
    // takes the per-tile barrier at the END of its tile instead of the middle and so runs half a tile ahead)
    // keeps the matrix pipe busy meanwhile.
    if ((NA_ABLATE & 16) == 0 || issued < 3) {
      if constexpr (kSplitRoles) {
        constexpr int HW = NWAVES / 2;
        if (!late)
          for (int b = wave; b < nblk; b += HW) glds16(src + (size_t)b * 1024, dst + b * 1024);
      } else {
        for (int b = wave; b < nblk; b += NWAVES) glds16(src + (size_t)b * 1024, dst + b * 1024);
      }
    }
    ++issued;
    next_t = next_t + 1 == ntiles ? 0 : next_t + 1;
  }
  __device__ __forceinline__ void start(const TileTab& tab, const char* b0, const char* b1, char* smem, uint32_t bufb,
                                        int npasses, int wave_, int lane_) {
    base0 = b0; base1 = b1;
    lds = smem; buf_bytes = bufb; wave = wave_; lane = lane_;
    ntiles = tab.ntiles; issued = 0; total = npasses * tab.ntiles; next_t = 0;
    late = SLOTS == 2 || (kSplitRoles && wave_ >= NWAVES / 2);
    uint2* table = (uint2*)(smem + kSlots * bufb);
#if NA_TRACE
    tlog = nullptr;
    tpos = 0;
    if (blockIdx.x == 0 && lane_ == 0 && (wave_ == 0 || wave_ == NWAVES / 2))
      tlog = (unsigned long long*)(smem + kSlots * bufb + 8 * kMaxTiles) + (wave_ ? kTraceMax : 0);
#endif
    for (int t = wave_ * 64 + lane_; t < tab.ntiles; t += NWAVES * 64) {
      const bool first = t < tab.split;
      const uint32_t* e = (first ? tab.hdr0 : tab.hdr1) + 1 + 2 * (first ? t : t - tab.split);
      table[t] = make_uint2(e[0] | (first ? 0u : 0x80000000u), e[1]);
    }
    __syncthreads();
    {
      const bool keep = late;
      late = false;  // every wave helps with the two start-up tiles
      if (total > 0) issue_next(0);
      if (total > 1) issue_next(1);
      late = keep;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur_slot = 0;
    cur = lds;
    // The DMA-issuing half is the critical path of a tile (MFMAs + ~500 cycles of DMA issue): give it static
    // priority on the shared matrix pipe; its partner fills the pipe while it issues DMA or waits for LDS.
    // (measured +4 %; a register-staged copy instead of LDS-DMA spills at 256 VGPRs and runs 0.58x)
    if (kSplitRoles && __builtin_amdgcn_readfirstlane(wave_) < NWAVES / 2) __builtin_amdgcn_s_setprio(1);
  }
  // Once per tile, between two MFMAs of the resident tile.
  __device__ __forceinline__ void mid_sync() {
    mark(2);
    if ((NA_ABLATE & 1) == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      mark(3);
      __builtin_amdgcn_s_barrier();
    }
    mark(4);
    if (issued < total) issue_next((cur_slot + 2) % SLOTS);
    mark(5);
  }
  // Once per tile, after its last LDS read: the next tile has been resident since this tile's barrier.
  __device__ __forceinline__ void next_tile() {
    if (late) mid_sync();
    mark(6);
    cur_slot = cur_slot == SLOTS - 1 ? 0 : cur_slot + 1;
    cur = lds + cur_slot * buf_bytes;
  }
};

// ------------------------------------------------------------------------------------------------ tile math
// A wave owns NB blocks of 32 samples (NB = 1 or 2): every A fragment read from LDS feeds NB MFMAs (one per
// block, independent accumulators), which halves LDS, DMA and barrier traffic per sample at NB = 2.
//
// Software pipeline inside a wave: while the MFMAs of tile j issue (each keeps the matrix pipe busy for 32
// cycles), the VALU slots in between carry the activation epilogue of tile j-1 (bias is already in the
// accumulator; activation; bf16 round; pack into next-layer B fragments).  The epilogue is cut into 8*NB
// two-value steps that are spread evenly over the tile's MFMAs in program order.
// A fragments are read from LDS one stage (kStage chunks) ahead of their MFMAs.  2 chunks in the 8-wave bf16
// kernels (256-VGPR budget: depth 4 spills ~11 registers, same speed), 4 in the 4-wave bf16x3 kernels.
#ifdef NA_KSTAGE
template <int PREC> constexpr int stage_depth() { return NA_KSTAGE; }
#else
template <int PREC> constexpr int stage_depth() { return PREC == NA_PREC_BF16X3 ? 4 : 2; }
#endif

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

template <int PREC = NA_PREC_BF16, bool CLAMP = true>
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
  bf16x2 v;
  v[0] = to_elem<PREC, CLAMP>(a);
  v[1] = to_elem<PREC, CLAMP>(b);
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float bf16_round(float a) { return (float)(__bf16)a; }

// Activation epilogue of NB finished accumulator tiles, two values (one packed dword) per step.
template <int PREC, int ACT, int NB>
struct Epilogue {
  f32x16 acc[NB];       // finished tiles (bias included)
  uint32_t hi[NB][8];   // packed bf16 pairs: dwords 0..3 -> fragment 2j, 4..7 -> fragment 2j+1
  uint32_t lo[NB][8];   // low halves (bf16x3 only)
  bool live;
  static constexpr int kSteps = 8 * NB;

  __device__ __forceinline__ void step(int u) {  // u in [0, 8*NB): block u / 8, dword u % 8
    if (!live) return;
    const int b = u >> 3, d = u & 7;
    const float x = act_apply<ACT, PREC>(acc[b][2 * d]);
    const float y = act_apply<ACT, PREC>(acc[b][2 * d + 1]);
    hi[b][d] = pack_bf16x2<PREC, ACT == NA_ACT_NONE>(x, y);
    // the volatile asm orders this dword with the surrounding scheduling fences, i.e. keeps its VALU work
    // between the two MFMAs it was written between (pure VALU would otherwise sink to the end of the tile)
    asm volatile("" : "+v"(hi[b][d]));
    if constexpr (PREC == NA_PREC_BF16X3) {
      lo[b][d] = pack_bf16x2(x - bf16_round(x), y - bf16_round(y));
      asm volatile("" : "+v"(lo[b][d]));
    } else if constexpr (kTwoPlane<PREC>) {  // f16 hi + f16 lo
      const bf16x2 hv = __builtin_bit_cast(bf16x2, hi[b][d]);
      lo[b][d] = pack_bf16x2<PREC, false>(x - from_elem<PREC>(hv[0]), y - from_elem<PREC>(hv[1]));
      asm volatile("" : "+v"(lo[b][d]));
    }
  }
  __device__ __forceinline__ void steps(int u0, int u1) {
#pragma unroll
    for (int u = 0; u < kSteps; ++u)
      if (u >= u0 && u < u1) step(u);
  }
  // fragments 2j and 2j+1 of block b
  __device__ __forceinline__ void finish(int b, Frag<PREC>& f0, Frag<PREC>& f1) {
    if (!live) return;
    u32x4 a = {hi[b][0], hi[b][1], hi[b][2], hi[b][3]}, c = {hi[b][4], hi[b][5], hi[b][6], hi[b][7]};
    f0.hi = __builtin_bit_cast(bf16x8, a);
    f1.hi = __builtin_bit_cast(bf16x8, c);
    if constexpr (kTwoPlane<PREC>) {
      u32x4 al = {lo[b][0], lo[b][1], lo[b][2], lo[b][3]}, cl = {lo[b][4], lo[b][5], lo[b][6], lo[b][7]};
      f0.lo = __builtin_bit_cast(bf16x8, al);
      f1.lo = __builtin_bit_cast(bf16x8, cl);
    }
  }
};

// acc[b](32 features x 32 samples) += A[frag0 .. frag0+NCH) . B[b][0..NCH) for the NB blocks of this wave.
// After MFMA group number m (counted from m0 over a tile total of mtot chunk-steps) the pending epilogue
// advances by its share of its steps.
template <int PREC, int ACT, int NB, int NCH, int BSTRIDE, class WS>
__device__ __forceinline__ void mma_chunks(WS& ws, f32x16 (&acc)[NB], const char* tile, int frag0,
                                           const Frag<PREC>* B, int lane, Epilogue<PREC, ACT, NB>& epi, int m0,
                                           int mtot) {
  constexpr int FB = PREC == NA_PREC_BF16X3 ? 2048 : 1024;
  constexpr int kStage = stage_depth<PREC>();
  constexpr int NS = (NCH + kStage - 1) / kStage;
  constexpr int ES = Epilogue<PREC, ACT, NB>::kSteps;
  bf16x8 ah[2][kStage], al[2][kStage];
  const char* a0 = tile + 1024 + frag0 * FB + lane * 16;  // fragments follow the 1-KiB bias block
#pragma unroll
  for (int c = 0; c < kStage && c < NCH; ++c) {
    ah[0][c] = *(const bf16x8*)(a0 + c * FB);
    if constexpr (PREC == NA_PREC_BF16X3) al[0][c] = *(const bf16x8*)(a0 + c * FB + 1024);
  }
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int cur = s & 1;
    if (s + 1 < NS) {
#pragma unroll
      for (int c = 0; c < kStage; ++c) {
        const int cc = (s + 1) * kStage + c;
        if (cc < NCH) {
          if constexpr ((NA_ABLATE & 8) != 0) { ah[cur ^ 1][c] = ah[0][0]; al[cur ^ 1][c] = ah[0][0]; continue; }
          ah[cur ^ 1][c] = *(const bf16x8*)(a0 + cc * FB);
          if constexpr (PREC == NA_PREC_BF16X3) al[cur ^ 1][c] = *(const bf16x8*)(a0 + cc * FB + 1024);
        }
      }
    }
#pragma unroll
    for (int c = 0; c < kStage; ++c) {
      const int cc = s * kStage + c;
      if (cc < NCH) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const Frag<PREC>& Bf = B[b * BSTRIDE + cc];
          if constexpr ((NA_ABLATE & 4) != 0) {
            asm volatile("" ::"v"(ah[cur][c]), "v"(Bf.hi));
          } else {
            if constexpr (PREC == NA_PREC_BF16X3) {
              acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[cur][c], Bf.hi, acc[b], 0, 0, 0);
              acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cur][c], Bf.lo, acc[b], 0, 0, 0);
            }
            if constexpr (PREC == NA_PREC_F16) {
              typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
              acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah[cur][c]), __builtin_bit_cast(f16x8, Bf.hi),
                                                              acc[b], 0, 0, 0);
            } else {
              acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cur][c], Bf.hi, acc[b], 0, 0, 0);
            }
          }
          const int m = (m0 + cc) * NB + b;
          if (m == 0) ws.mark(1);
          epi.steps(ES * m / (mtot * NB), ES * (m + 1) / (mtot * NB));
          __builtin_amdgcn_sched_barrier(0);
          if (m == (mtot * NB - 1) / 2 && !ws.late) {
            ws.mid_sync();
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
  }
}

// bias block of a tile (its first KiB): floats [hi(2)][16] in accumulator order
__device__ __forceinline__ f32x16 load_bias(const char* tile, int lane) {
  const f32x4* b = (const f32x4*)(tile + (lane >> 5) * 64);
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 v = b[q];
    acc[q * 4 + 0] = v[0]; acc[q * 4 + 1] = v[1]; acc[q * 4 + 2] = v[2]; acc[q * 4 + 3] = v[3];
  }
  return acc;
}

// accumulator tile -> activated input fragments of the next layer (chunks 2j and 2j+1), not overlapped
template <int PREC, int ACT>
__device__ __forceinline__ void acc_to_frags(const f32x16& acc, Frag<PREC>& f0, Frag<PREC>& f1) {
  Epilogue<PREC, ACT, 1> e;
  e.acc[0] = acc;
  e.live = true;
  e.steps(0, 8);
  e.finish(0, f0, f1);
}

// The eight 32-feature tiles of one Linear with 256 outputs: K = [H (NH chunks) | I (NI chunks)].
// Tile j's MFMAs are interleaved with the epilogue of tile j-1; the last epilogue is flushed at the end.
// Fragment arrays are [NB][...] flattened: block b's chunk c of H is H[b*kHC + c], of I is I[b*NIS + c].
template <int PREC, int ACT, int NB, int NH, int NI, int NIS, class WS>
__device__ __forceinline__ void linear256(WS& ws, const Frag<PREC>* H, const Frag<PREC>* I,
                                          Frag<PREC>* Hn, int lane) {
  Epilogue<PREC, ACT, NB> epi;
  epi.live = false;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    f32x16 acc[NB];
    ws.mark(0);
    acc[0] = load_bias(ws.cur, lane);
#pragma unroll
    for (int b = 1; b < NB; ++b) acc[b] = acc[0];
    if constexpr (NH > 0) mma_chunks<PREC, ACT, NB, NH, kHC>(ws, acc, ws.cur, 0, H, lane, epi, 0, NH + NI);
    if constexpr (NI > 0) mma_chunks<PREC, ACT, NB, NI, NIS>(ws, acc, ws.cur, NH, I, lane, epi, NH, NH + NI);
    if (j > 0) {
#pragma unroll
      for (int b = 0; b < NB; ++b) epi.finish(b, Hn[b * kHC + 2 * j - 2], Hn[b * kHC + 2 * j - 1]);
    }
    ws.next_tile();
#pragma unroll
    for (int b = 0; b < NB; ++b) epi.acc[b] = acc[b];
    epi.live = true;
  }
  epi.steps(0, 8 * NB);
#pragma unroll
  for (int b = 0; b < NB; ++b) epi.finish(b, Hn[b * kHC + 14], Hn[b * kHC + 15]);
}

// One SkipConnMLP up to (not including) the `out` Linear.  On entry I[] holds the raw init input
// fragments; on exit H[] holds act(last hidden) ready for the out layer, I[] holds act(init).
template <int PREC, int ACT, int NB, int NI, class WS>
__device__ __forceinline__ void mlp_hidden_layers(WS& ws, int num_layers, int skip,
                                                  Frag<PREC> (&I)[NB * NI], Frag<PREC> (&H)[NB * kHC], int lane) {
  Frag<PREC> Hn[NB * kHC];
  // ---- init Linear: dim_p -> 256
  linear256<PREC, ACT, NB, 0, NI, NI>(ws, nullptr, I, Hn, lane);
#pragma unroll
  for (int c = 0; c < NB * kHC; ++c) H[c] = Hn[c];
#pragma unroll
  for (int c = 0; c < NB * NI; ++c) frag_activate<PREC, ACT>(I[c]);
  // ---- hidden Linears
  for (int i = 0; i < num_layers; ++i) {
    const bool sk = (i % skip) == 0 && i != num_layers - 1;
    if (sk) linear256<PREC, ACT, NB, kHC, NI, NI>(ws, H, I, Hn, lane);
    else linear256<PREC, ACT, NB, kHC, 0, NI>(ws, H, nullptr, Hn, lane);
#pragma unroll
    for (int c = 0; c < NB * kHC; ++c) H[c] = Hn[c];
  }
}

// One 32-row tile of the `out` Linear for the NB blocks (no activation; the caller's epilogue follows the sync).
template <int PREC, int NB, class WS>
__device__ __forceinline__ void mlp_out_tile(WS& ws, const Frag<PREC> (&H)[NB * kHC], int lane,
                                             f32x16 (&acc)[NB]) {
  acc[0] = load_bias(ws.cur, lane);
#pragma unroll
  for (int b = 1; b < NB; ++b) acc[b] = acc[0];
  Epilogue<PREC, NA_ACT_NONE, NB> none;
  none.live = false;
  mma_chunks<PREC, NA_ACT_NONE, NB, kHC, kHC>(ws, acc, ws.cur, 0, H, lane, none, 0, kHC);
  ws.next_tile();
}

// feature index (within a 32-row tile) held by accumulator register r of this lane
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

}  // namespace na
