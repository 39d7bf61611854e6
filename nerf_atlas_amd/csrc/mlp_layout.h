// Weight-stream layout shared by the packer (mlp_pack.hip), the generic fused MLP (mlp_fused.hip) and
// the fused PlainNeRF renderer (render_fused.hip).
//
// Stream = tiles in consumption order.  Tile (layer, j) = a 1-KiB bias block (floats [hi(2)][16]) followed by
// `nfrag` A-fragments (one per 16-wide K chunk; `planes` KiB each: bf16 hi [, bf16 lo]).
//   init layer   : 8 tiles x NI fragments              (K = init input, "slot map" order)
//   hidden layer : 8 tiles x 16 (+NI if skip) fragments (K = hidden in pi-order [, init slots])
//   out layer    : ceil(out/32) tiles x 16 fragments
// Fragment (j, c), lane l, element e holds  W[row(32j + (l&31))][col(c, 8*(l>>5) + e)].
#pragma once
#include "mlp_engine.h"

namespace na {

// hidden feature held in k-slot kappa of chunk c (see mlp_engine.h: C layout -> B layout)
__host__ __device__ inline int pi_perm(int kappa) {
  int hi = kappa >> 3, s = kappa & 7;
  return (s & 3) + 8 * (s >> 2) + 4 * hi;
}

// Reference init-feature index (column of init.weight / of the skip part of layers[i].weight) that lives
// in k-slot kappa of init chunk c; -1 = zero padding.  Reference order is [p | enc(p) | latent]
// (src/neural_blocks.py:283-287).
__host__ __device__ inline int init_slot_feature(const NaMlpDesc& d, int c, int kappa) {
  const int dim_p = d.in_size + d.enc_dims + d.latent_size;
  if (d.layout == NA_LAYOUT_PLAIN_VIEW) {
    // chunks 0..3: latent (= `first` out rows 1..64, arriving in accumulator order); chunk 4: x,y,z,elev,azim
    if (c < 4) return d.in_size + 16 * c + pi_perm(kappa);
    if (c == 4) return kappa < d.in_size ? kappa : -1;
    return -1;
  }
  if (d.enc_kind == NA_ENC_HASH) {
    // chunks 0,1: lane-half hi computes levels 4hi..4hi+3; chunk c slots s -> level 4hi + 2c + (s>>2)
    if (c < 2) {
      int hi = kappa >> 3, s = kappa & 7;
      int level = 4 * hi + 2 * c + (s >> 2);
      return d.in_size + 3 + 4 * level + (s & 3);  // enc = [x(3) | 8 levels x 4]
    }
    int rho = 16 * (c - 2) + kappa;  // [p(3) | x(3) | latent]
    if (rho < 6) return rho;
    rho -= 6;
    return rho < d.latent_size ? d.in_size + d.enc_dims + rho : -1;
  }
  if (d.enc_kind == NA_ENC_FOURIER) {
    const int F = d.enc_dims / 2;
    const int nfc = F / 8;  // chunks holding sin/cos
    if (c < nfc) {
      int freq = 16 * (c >> 1) + kappa;
      return d.in_size + ((c & 1) ? F : 0) + freq;
    }
    int rho = 16 * (c - nfc) + kappa;  // [p | latent]
    if (rho < d.in_size) return rho;
    rho -= d.in_size;
    return rho < d.latent_size ? d.in_size + d.enc_dims + rho : -1;
  }
  int f = 16 * c + kappa;
  return f < dim_p ? f : -1;
}

// Reference output row stored at stream row rho of the out layer; -1 = zero row.
__host__ __device__ inline int out_row_map(const NaMlpDesc& d, int rho) {
  if (d.layout == NA_LAYOUT_PLAIN_FIRST) {
    // rows 0..63 = intermediate (reference rows 1..64), row 64 = density (reference row 0)
    if (rho < d.out_size - 1) return rho + 1;
    if (rho == d.out_size - 1) return 0;
    return -1;
  }
  return rho < d.out_size ? rho : -1;
}

inline int init_chunks_needed(const NaMlpDesc& d) {
  if (d.layout == NA_LAYOUT_PLAIN_VIEW) return 5;
  if (d.enc_kind == NA_ENC_HASH) return 2 + (6 + d.latent_size + 15) / 16;
  if (d.enc_kind == NA_ENC_FOURIER) return d.enc_dims / 16 + (d.in_size + d.latent_size + 15) / 16;
  return (d.in_size + d.enc_dims + d.latent_size + 15) / 16;
}

// NI values that have a compiled kernel, per (activation, encoder).  0 = unsupported.
inline int effective_ni(const NaMlpDesc& d) {
  static const int leaky_none[] = {1, 3, 0};
  static const int leaky_hash[] = {3, 7, 9, 0};
  static const int leaky_fourier[] = {17, 0};
  static const int sine_none[] = {1, 5, 11, 0};
  static const int none[] = {0};
  const int need = init_chunks_needed(d);
  const int* set = none;
  if (d.activation == NA_ACT_SIN) set = d.enc_kind == NA_ENC_NONE ? sine_none : none;
  else if (d.enc_kind == NA_ENC_NONE) set = leaky_none;
  else if (d.enc_kind == NA_ENC_HASH) set = leaky_hash;
  else if (d.enc_kind == NA_ENC_FOURIER) set = leaky_fourier;
  for (int i = 0; set[i]; ++i)
    if (set[i] >= need) return set[i];
  return 0;
}

inline int out_tiles(const NaMlpDesc& d) { return (d.out_size + 31) / 32; }

inline bool layer_has_skip(const NaMlpDesc& d, int i) { return (i % d.skip) == 0 && i != d.num_layers - 1; }

inline int planes_of(int precision) { return precision == NA_PREC_BF16X3 ? 2 : 1; }

// why a desc cannot run on the MFMA path (nullptr = supported)
inline const char* mlp_unsupported_reason(const NaMlpDesc& d) {
  if (d.hidden != kHidden) return "hidden width must be 256";
  if (d.num_layers < 1 || d.num_layers > 8) return "num_layers must be 1..8";
  if (d.skip < 1) return "skip must be >= 1";
  if (d.activation != NA_ACT_LEAKY_RELU && d.activation != NA_ACT_SIN) return "activation must be leaky_relu or sin";
  if (d.out_size < 1 || d.out_size > 96) return "out_size must be 1..96";
  if (d.enc_kind == NA_ENC_HASH && (d.in_size != 3 || d.enc_dims != 35)) return "hash encoder needs in_size 3, enc_dims 35";
  if (d.enc_kind == NA_ENC_FOURIER && (d.enc_dims % 32 != 0 || d.in_size > 8)) return "fourier encoder needs 2F % 32 == 0";
  if (d.enc_kind == NA_ENC_NONE && d.enc_dims != 0) return "enc_dims must be 0 without an encoder";
  if (d.layout == NA_LAYOUT_PLAIN_VIEW && (d.in_size != 5 || d.latent_size != 64 || d.enc_kind != NA_ENC_NONE))
    return "PLAIN_VIEW layout needs in_size 5, latent 64, no encoder";
  if (d.layout == NA_LAYOUT_PLAIN_FIRST && d.out_size != 65) return "PLAIN_FIRST layout needs out_size 65";
  if (effective_ni(d) == 0) return "init width has no compiled kernel";
  return nullptr;
}

// Tile list of one MLP: entries[2t] = block offset in the stream, entries[2t+1] = block count.
// Returns the tile count (or -1 on overflow); *blocks = stream size in 1-KiB blocks.
inline int build_tiles(const NaMlpDesc& d, int precision, uint32_t* entries, uint32_t* blocks) {
  const int P = planes_of(precision), NI = effective_ni(d);
  uint32_t off = 0;
  int t = 0;
  auto push = [&](int nfrag) {
    if (t >= kMaxTilesPerMlp) return false;
    if (entries) { entries[2 * t] = off; entries[2 * t + 1] = (uint32_t)(nfrag * P + 1); }
    off += (uint32_t)(nfrag * P + 1);
    ++t;
    return true;
  };
  for (int j = 0; j < 8; ++j)
    if (!push(NI)) return -1;
  for (int i = 0; i < d.num_layers; ++i)
    for (int j = 0; j < 8; ++j)
      if (!push(kHC + (layer_has_skip(d, i) ? NI : 0))) return -1;
  for (int j = 0; j < out_tiles(d); ++j)
    if (!push(kHC)) return -1;
  if (blocks) *blocks = off;
  return t;
}

inline int tile_count(const NaMlpDesc& d) { return 8 * (1 + d.num_layers) + out_tiles(d); }

}  // namespace na
