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
// header words are read with scalar loads (s_load_dwordx2 through the constant address space)
typedef const __attribute__((address_space(4))) uint32_t* hdr_ptr_t;

template <int PREC>
struct Frag {
  bf16x8 hi;
  bf16x8 lo;  // dead (eliminated) when PREC == 0
};

// ------------------------------------------------------------------------------------------------ activations
// sin with a 2-term Cody-Waite reduction by pi and a degree-9 odd polynomial (least-squares on
// Chebyshev nodes of [-pi/2,pi/2]): max |err| 1.6e-7 for |x| <= 3e3 (checked against fp64).
__device__ __forceinline__ float sin_cw(float x) {
  float q = rintf(x * 0.318309886183790672f);
  float r = fmaf(q, -3.140625f, x);
  r = fmaf(q, -9.67502593994140625e-4f, r);
  r = fmaf(q, -1.509957990978376432e-7f, r);
  float r2 = r * r;
  float p = fmaf(r2, 2.5962193818e-06f, -1.9804804431e-04f);
  p = fmaf(p, r2, 8.3329907333e-03f);
  p = fmaf(p, r2, -1.6666655917e-01f);
  float s = fmaf(p * r2, r, r);
  int qi = (int)q;
  return (qi & 1) ? -s : s;
}

// Hardware sine (v_sin_f32 takes revolutions, valid for |r| <= 256 -> reduce with v_fract first).  Used by the
// bf16 fast mode only, where its ~1e-6 absolute error is far below the bf16 operand rounding.
__device__ __forceinline__ float sin_hw(float x) {
  return __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(x * 0.15915494309189535f));
}

template <int ACT, int PREC = NA_PREC_BF16X3>
__device__ __forceinline__ float act_apply(float v) {
  // leaky_relu(v) = max(v, 0.01 v) = median(v, 0.01 v, +big): v_med3_f32 needs no canonicalising v_max
  if constexpr ((NA_ABLATE & 2) != 0) return v;
  if constexpr (ACT == NA_ACT_LEAKY_RELU) return __builtin_amdgcn_fmed3f(v, v * 0.01f, 3.0e38f);
  else if constexpr (ACT == NA_ACT_SIN) return PREC == NA_PREC_BF16 ? sin_hw(v) : sin_cw(v);
  else return v;
}

// ------------------------------------------------------------------------------------------------ fragments
template <int PREC>
__device__ __forceinline__ Frag<PREC> make_frag(const float (&v)[8]) {
  Frag<PREC> f;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    __bf16 h = (__bf16)v[e];
    f.hi[e] = h;
    if constexpr (PREC == NA_PREC_BF16X3) f.lo[e] = (__bf16)(v[e] - (float)h);
  }
  return f;
}

template <int PREC>
__device__ __forceinline__ float frag_value(const Frag<PREC>& f, int e) {
  float v = (float)f.hi[e];
  if constexpr (PREC == NA_PREC_BF16X3) v = v + (float)f.lo[e];
  return v;
}

template <int PREC>
__device__ __forceinline__ void pin_frag(Frag<PREC>& f) {
  asm volatile("" : "+v"(f.hi));
  if constexpr (PREC == NA_PREC_BF16X3) asm volatile("" : "+v"(f.lo));
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
// workgroup knows up front how many passes it will run, so the prefetcher manages itself: advance() makes the
// next tile resident and issues the DMA of the one after it (double buffer, one barrier per tile).  The header
// entry of the tile to be issued is fetched one tile ahead with a scalar load, off the critical path.
template <int NWAVES>
struct WeightStream {
  const char* base0;
  const char* base1;
  hdr_ptr_t hdr0, hdr1;
  char* lds;           // two buffers of buf_bytes each
  uint32_t buf_bytes;
  uint32_t parity;     // buffer that receives the NEXT advance()'s tile
  int wave, lane;
  int ntiles, split;
  int issued, total;   // tiles issued so far / to issue over the whole kernel
  int next_t;          // tile index (within a pass) of the preloaded entry
  uint32_t e_off, e_nblk;
  bool e_first;
  const char* cur;     // LDS address of the resident tile

  __device__ __forceinline__ void preload(int t) {
    e_first = t < split;
    hdr_ptr_t e = (e_first ? hdr0 : hdr1) + 1 + 2 * (e_first ? t : t - split);
    e_off = e[0];
    e_nblk = e[1];
    next_t = t;
  }
  __device__ __forceinline__ void issue_preloaded(uint32_t par) {
    const char* src = (e_first ? base0 : base1) + (size_t)e_off * 1024 + lane * 16;
    char* dst = lds + par * buf_bytes;
    if ((NA_ABLATE & 16) == 0 || issued < 2)
      for (int b = wave; b < (int)e_nblk; b += NWAVES) glds16(src + (size_t)b * 1024, dst + b * 1024);
    ++issued;
    int t = next_t + 1;
    preload(t == ntiles ? 0 : t);
  }
  __device__ __forceinline__ void start(const TileTab& tab, const char* b0, const char* b1, char* smem, uint32_t bufb,
                                        int npasses, int wave_, int lane_) {
    base0 = b0; base1 = b1; hdr0 = (hdr_ptr_t)tab.hdr0; hdr1 = (hdr_ptr_t)tab.hdr1;
    lds = smem; buf_bytes = bufb; wave = wave_; lane = lane_;
    ntiles = tab.ntiles; split = tab.split; issued = 0; total = npasses * tab.ntiles;
    parity = 0;
    preload(0);
    if (total > 0) issue_preloaded(0);
    advance();
  }
  // Called once per tile, after the tile's last LDS read: makes the next tile resident (its DMA was issued one
  // tile earlier) and starts the DMA of the one after it into the buffer that was just released.
  __device__ __forceinline__ void advance() {
    if ((NA_ABLATE & 1) == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    cur = lds + parity * buf_bytes;
    parity ^= 1u;
    if (issued < total) issue_preloaded(parity);
  }
};

// ------------------------------------------------------------------------------------------------ tile math
// Software pipeline inside a wave: while the MFMAs of tile j issue (a dependent accumulator chain that keeps the
// matrix pipe busy 32 cycles per instruction), the VALU slots in between carry the activation epilogue of tile
// j-1 (bias is already in the accumulator; activation; bf16 round; pack into next-layer B fragments).  The
// epilogue is cut into 16 single-value steps that are spread evenly over the tile's MFMAs in program order.
constexpr int kStage = 4;  // A fragments are read from LDS one stage (4 chunks) ahead of their MFMAs

// Activation epilogue of a finished accumulator tile, one value per step.
template <int PREC, int ACT>
struct Epilogue {
  f32x16 acc;          // finished tile (bias included)
  float v0[8], v1[8];  // activated values (become fragments 2j and 2j+1 of the next layer)
  bool live;
  __device__ __forceinline__ void step(int k) {
    if (!live) return;
    // the volatile asm orders this value with the surrounding scheduling fences, i.e. keeps it between the
    // two MFMAs it was written between (pure VALU would otherwise sink to the end of the tile)
    if (k < 8) { v0[k] = act_apply<ACT, PREC>(acc[k]); asm volatile("" : "+v"(v0[k])); }
    else { v1[k - 8] = act_apply<ACT, PREC>(acc[k]); asm volatile("" : "+v"(v1[k - 8])); }
  }
  __device__ __forceinline__ void steps(int k0, int k1) {
#pragma unroll
    for (int k = 0; k < 16; ++k)
      if (k >= k0 && k < k1) step(k);
  }
  __device__ __forceinline__ void finish(Frag<PREC>& f0, Frag<PREC>& f1) {
    if (!live) return;
    f0 = make_frag<PREC>(v0);
    f1 = make_frag<PREC>(v1);
    // the fragments are only consumed by the NEXT layer: without this pin instruction selection sinks every
    // epilogue of a layer behind its last tile (128 live fp32 accumulators -> spills, no overlap)
    pin_frag<PREC>(f0);
    pin_frag<PREC>(f1);
  }
};

// acc(32 features x 32 samples) += A[frag0 .. frag0+NCH) . B[0..NCH); after MFMA number m (counted from m0 over a
// tile total of mtot) the pending epilogue advances by its share of the 16 steps.
template <int PREC, int ACT, int NCH>
__device__ __forceinline__ void mma_chunks(f32x16& acc, const char* tile, int frag0, const Frag<PREC>* B, int lane,
                                           Epilogue<PREC, ACT>& epi, int m0, int mtot) {
  constexpr int FB = PREC == NA_PREC_BF16X3 ? 2048 : 1024;
  constexpr int NS = (NCH + kStage - 1) / kStage;
  bf16x8 ah[2][kStage], al[2][kStage];
  const char* a0 = tile + frag0 * FB + lane * 16;
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
        if constexpr ((NA_ABLATE & 4) != 0) {
          asm volatile("" ::"v"(ah[cur][c]), "v"(B[cc].hi));
        } else {
          if constexpr (PREC == NA_PREC_BF16X3) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[cur][c], B[cc].hi, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cur][c], B[cc].lo, acc, 0, 0, 0);
          }
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cur][c], B[cc].hi, acc, 0, 0, 0);
        }
        const int m = m0 + cc;
        epi.steps(16 * m / mtot, 16 * (m + 1) / mtot);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
}

// bias block of a tile: floats [hi(2)][16] right after its `nfrag` fragments
template <int PREC>
__device__ __forceinline__ f32x16 load_bias(const char* tile, int nfrag, int lane) {
  constexpr int FB = PREC == NA_PREC_BF16X3 ? 2048 : 1024;
  const f32x4* b = (const f32x4*)(tile + nfrag * FB + (lane >> 5) * 64);
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
  Epilogue<PREC, ACT> e;
  e.acc = acc;
  e.live = true;
  e.steps(0, 16);
  e.finish(f0, f1);
}

// The eight 32-feature tiles of one Linear with 256 outputs: K = [H (NH chunks) | I (NI chunks)].
// Tile j's MFMAs are interleaved with the epilogue of tile j-1; the last epilogue is flushed at the end.
template <int PREC, int ACT, int NH, int NI, int NWAVES>
__device__ __forceinline__ void linear256(WeightStream<NWAVES>& ws, const Frag<PREC>* H, const Frag<PREC>* I,
                                          Frag<PREC> (&Hn)[kHC], int lane) {
  Epilogue<PREC, ACT> epi;
  epi.live = false;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    f32x16 acc = load_bias<PREC>(ws.cur, NH + NI, lane);
    if constexpr (NH > 0) mma_chunks<PREC, ACT, NH>(acc, ws.cur, 0, H, lane, epi, 0, NH + NI);
    if constexpr (NI > 0) mma_chunks<PREC, ACT, NI>(acc, ws.cur, NH, I, lane, epi, NH, NH + NI);
    if (j > 0) epi.finish(Hn[2 * j - 2], Hn[2 * j - 1]);
    ws.advance();
    epi.acc = acc;
    epi.live = true;
  }
  epi.steps(0, 16);
  epi.finish(Hn[14], Hn[15]);
}

// One SkipConnMLP up to (not including) the `out` Linear.  On entry I[] holds the raw init input
// fragments; on exit H[] holds act(last hidden) ready for the out layer, I[] holds act(init).
template <int PREC, int ACT, int NI, int NWAVES>
__device__ __forceinline__ void mlp_hidden_layers(WeightStream<NWAVES>& ws, int num_layers, int skip,
                                                  Frag<PREC> (&I)[NI], Frag<PREC> (&H)[kHC], int lane) {
  Frag<PREC> Hn[kHC];
  // ---- init Linear: dim_p -> 256
  linear256<PREC, ACT, 0, NI, NWAVES>(ws, nullptr, I, Hn, lane);
#pragma unroll
  for (int c = 0; c < kHC; ++c) H[c] = Hn[c];
#pragma unroll
  for (int c = 0; c < NI; ++c) frag_activate<PREC, ACT>(I[c]);
  // ---- hidden Linears
  for (int i = 0; i < num_layers; ++i) {
    const bool sk = (i % skip) == 0 && i != num_layers - 1;
    if (sk) linear256<PREC, ACT, kHC, NI, NWAVES>(ws, H, I, Hn, lane);
    else linear256<PREC, ACT, kHC, 0, NWAVES>(ws, H, nullptr, Hn, lane);
#pragma unroll
    for (int c = 0; c < kHC; ++c) H[c] = Hn[c];
  }
}

// One 32-row tile of the `out` Linear (no activation on the result; the caller's epilogue follows the sync).
template <int PREC, int NWAVES>
__device__ __forceinline__ f32x16 mlp_out_tile(WeightStream<NWAVES>& ws, const Frag<PREC> (&H)[kHC], int lane) {
  f32x16 acc = load_bias<PREC>(ws.cur, kHC, lane);
  Epilogue<PREC, NA_ACT_NONE> none;
  none.live = false;
  mma_chunks<PREC, NA_ACT_NONE, kHC>(acc, ws.cur, 0, H, lane, none, 0, kHC);
  ws.advance();
  return acc;
}

// feature index (within a 32-row tile) held by accumulator register r of this lane
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

}  // namespace na
