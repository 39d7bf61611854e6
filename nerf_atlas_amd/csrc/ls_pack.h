// Weight-stream packing of the bf16 / bf16x3 / f16 schedules and the launch of render_ls_kernel (the NA_PREC_F16X streams are
// built from schedule tables: ls_xsched.h).
#pragma once
#include "ls_kernel.h"

namespace na {
namespace ls {

// ================================================================================================ pack
#if NA_PREC_INST == 0
// dir_to_elev_azim of every ray, once (src/utils.py:247-254): the View MLP's geometry chunk reads it per block
__global__ void ray_elaz_kernel(const float* __restrict__ rays, int64_t R, float* __restrict__ elaz) {
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < R; r += (int64_t)gridDim.x * blockDim.x) {
    float el, az;
    elev_azim(rays[r * 6 + 3], rays[r * 6 + 4], rays[r * 6 + 5], el, az);
    elaz[r * 2] = el;
    elaz[r * 2 + 1] = az;
  }
}

struct PackArgs {
  const float* w_first[6];  // init, layers.0..3, out   (nn.Linear layout [out,in])
  const float* b_first[6];
  const float* w_view[6];
  const float* b_view[6];
};

__host__ __device__ inline int phase_first_frag(int p) {
  int s = 0;
  for (int i = 0; i < p; ++i) s += 2 * phase_pairs(i);
  return s;
}

// One thread per bf16 element of the hi plane of every fragment, plus the bias blocks.
__global__ void pack_ls_kernel(PackArgs w, int planes, int f16, char* __restrict__ dst) {
  const NaMlpDesc d1 = {3, NA_ENC_HASH, 35, 0, 4, 256, 65, 3, NA_ACT_LEAKY_RELU, NA_LAYOUT_PLAIN_FIRST};
  const NaMlpDesc d2 = {5, NA_ENC_NONE, 0, 64, 4, 256, 3, 3, NA_ACT_SIN, NA_LAYOUT_PLAIN_VIEW};
  const int frag_bytes = 1024 * planes;
  const int64_t nfrag_rg = 2 * kPairsPerPass;
  const int64_t nelem = 4 * nfrag_rg * 512;
  const int64_t nbias = 4 * kNPhase * 256;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nelem + nbias; i += (int64_t)gridDim.x * blockDim.x) {
    if (i < nelem) {
      const int e = (int)(i & 7), l = (int)((i >> 3) & 63);
      const int64_t fg = i >> 9;
      const int rg = (int)(fg / nfrag_rg);
      int f = (int)(fg % nfrag_rg);
      int p = 0;
      while (f >= 2 * phase_pairs(p)) { f -= 2 * phase_pairs(p); ++p; }
      const int kappa = 8 * (l >> 5) + e;
      const bool view = p >= 6;
      const NaMlpDesc& d = view ? d2 : d1;
      const int lp = view ? p - 6 : p;  // 0 init, 1 skip layer, 2..4 hidden, 5 out
      const float* W = view ? w.w_view[lp] : w.w_first[lp];
      const int dim_p = d.in_size + d.enc_dims + d.latent_size;
      int row, col, in_dim, out_dim;
      if (lp == 5) {  // out layers.  view.out: fragment f = chunk c (one tile); first.out: row group rg holds tile min(rg, 2)
        const int c = f, j = view ? 0 : (rg < 2 ? rg : 2);
        row = out_row_map(d, 32 * j + (l & 31));
        col = 16 * c + pi_perm(kappa);
        in_dim = kHidden; out_dim = d.out_size;
      } else {
        const int q = f >> 1, t = f & 1;
        row = 32 * (2 * rg + t) + (l & 31);
        out_dim = kHidden;
        if (lp == 0) { col = init_slot_feature(d, q, kappa); in_dim = dim_p; }
        else if (lp == 1) {
          // chunk order of the skip layer: init chunks from LDS, the 16 hidden chunks, then (View) the geometry chunk
          const int nlds = view ? 4 : 3;
          if (q < nlds) { col = init_slot_feature(d, q, kappa); if (col >= 0) col += kHidden; }
          else if (q < nlds + kHC) col = 16 * (q - nlds) + pi_perm(kappa);
          else { col = init_slot_feature(d, 4, kappa); if (col >= 0) col += kHidden; }
          in_dim = kHidden + dim_p;
        } else { col = 16 * q + pi_perm(kappa); in_dim = kHidden; }
      }
      float v = 0.f;
      if (row >= 0 && row < out_dim && col >= 0 && col < in_dim) v = W[(int64_t)row * in_dim + col];
      char* o = dst + kHeaderBytes + kBiasBytes + ((int64_t)rg * nfrag_rg + (fg % nfrag_rg)) * frag_bytes + l * 16 + e * 2;
      const __bf16 h = f16 ? to_elem<NA_PREC_F16>(v) : (__bf16)v;
      *(uint16_t*)o = __builtin_bit_cast(uint16_t, h);
      if (planes == 2) {
        const __bf16 lo = (__bf16)(v - (float)h);
        *(uint16_t*)(o + 1024) = __builtin_bit_cast(uint16_t, lo);
      }
    } else {
      const int64_t q = i - nelem;
      const int k = (int)(q & 255), p = (int)((q >> 8) % kNPhase), rg = (int)((q >> 8) / kNPhase);
      const bool view = p >= 6;
      const NaMlpDesc& d = view ? d2 : d1;
      const int lp = view ? p - 6 : p;
      const float* B = p >= 12 ? nullptr : view ? w.b_view[lp] : w.b_first[lp];  // (bias blocks 12..kNPhase-1: other schedules)
      const int slot = k >> 5, hi = (k >> 4) & 1, r = k & 15;
      const int rin = (r & 3) + 8 * (r >> 2) + 4 * hi;
      float v = 0.f;
      if (lp == 5) {
        const int row = slot < (view ? 1 : 3) ? out_row_map(d, 32 * slot + rin) : -1;
        if (row >= 0 && row < d.out_size && B != nullptr) v = B[row];
      } else if (slot < 2 && B != nullptr) {
        v = B[32 * (2 * rg + slot) + rin];
      }
      *(float*)(dst + kHeaderBytes + ((int64_t)rg * kNPhase + p) * 1024 + k * 4) = v;
    }
  }
}

__global__ void pack_ls_header_kernel(uint32_t magic, uint32_t precision, uint32_t* __restrict__ dst, uint32_t pairs) {
  if (threadIdx.x == 0) { dst[0] = magic; dst[1] = precision; dst[2] = pairs; dst[3] = kNPhase; }
}

// TinyNeRF stream (MODEL 1): same element order as pack_ls_kernel, phases per tiny_phase_pairs
struct TinyPackArgs {
  const float* w[8];  // init, layers.0..5, out   (nn.Linear layout [out,in])
  const float* b[8];
};
__global__ void pack_ls_tiny_kernel(TinyPackArgs w, int planes, int f16, char* __restrict__ dst) {
  const NaMlpDesc d = {3, NA_ENC_NONE, 0, 0, 6, 256, 4, 3, NA_ACT_LEAKY_RELU, NA_LAYOUT_GENERIC};
  const int frag_bytes = 1024 * planes;
  const int64_t nfrag_rg = 2 * kTinyPairs;
  const int64_t nelem = 4 * nfrag_rg * 512;
  const int64_t nbias = 4 * kNPhase * 256;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nelem + nbias; i += (int64_t)gridDim.x * blockDim.x) {
    if (i < nelem) {
      const int e = (int)(i & 7), l = (int)((i >> 3) & 63);
      const int64_t fg = i >> 9;
      const int rg = (int)(fg / nfrag_rg);
      int f = (int)(fg % nfrag_rg);
      int p = 0;
      while (f >= 2 * tiny_phase_pairs(p)) { f -= 2 * tiny_phase_pairs(p); ++p; }
      const int kappa = 8 * (l >> 5) + e;
      const float* W = w.w[p];  // p: 0 init, 1..6 layers.0..5, 7 out
      int row, col, in_dim, out_dim;
      if (p == 7) {  // fragment f = chunk c of the single out tile
        row = out_row_map(d, l & 31);
        col = 16 * f + pi_perm(kappa);
        in_dim = kHidden; out_dim = d.out_size;
      } else {
        const int q = f >> 1, t = f & 1;
        row = 32 * (2 * rg + t) + (l & 31);
        out_dim = kHidden;
        if (p == 0) { col = q == 0 ? init_slot_feature(d, 0, kappa) : -1; in_dim = d.in_size; }
        else if (p == 1 || p == 4) {  // [hidden | init] in the reference's column order, init chunk first in the stream
          if (q == 0) { col = init_slot_feature(d, 0, kappa); if (col >= 0) col += kHidden; }
          else col = 16 * (q - 1) + pi_perm(kappa);
          in_dim = kHidden + d.in_size;
        } else { col = 16 * q + pi_perm(kappa); in_dim = kHidden; }
      }
      float v = 0.f;
      if (row >= 0 && row < out_dim && col >= 0 && col < in_dim) v = W[(int64_t)row * in_dim + col];
      char* o = dst + kHeaderBytes + kBiasBytes + ((int64_t)rg * nfrag_rg + (fg % nfrag_rg)) * frag_bytes + l * 16 + e * 2;
      const __bf16 h = f16 ? to_elem<NA_PREC_F16>(v) : (__bf16)v;
      *(uint16_t*)o = __builtin_bit_cast(uint16_t, h);
      if (planes == 2) {
        const __bf16 lo = (__bf16)(v - (float)h);
        *(uint16_t*)(o + 1024) = __builtin_bit_cast(uint16_t, lo);
      }
    } else {
      const int64_t q = i - nelem;
      const int k = (int)(q & 255), p = (int)((q >> 8) % kNPhase), rg = (int)((q >> 8) / kNPhase);
      const int slot = k >> 5, hi = (k >> 4) & 1, r = k & 15;
      const int rin = (r & 3) + 8 * (r >> 2) + 4 * hi;
      float v = 0.f;
      if (p < kTinyPhases && w.b[p] != nullptr) {
        if (p == 7) {
          const int row = slot < 1 ? out_row_map(d, rin) : -1;
          if (row >= 0 && row < d.out_size) v = w.b[p][row];
        } else if (slot < 2) {
          v = w.b[p][32 * (2 * rg + slot) + rin];
        }
      }
      *(float*)(dst + kHeaderBytes + ((int64_t)rg * kNPhase + p) * 1024 + k * 4) = v;
    }
  }
}

// Hash-encoded SkipConnMLP stream (MODEL 4 outside f16x, round 6): same element order, phases per hashmlp_phase_pairs; the out phase
// of row group rg holds ONE 32-row tile (rows 32 (rg >> 1) ..: fragment f = chunk c) and three zero pairs
struct HashMlpPackArgs {
  const float* w[7];  // init, layers.0..4, out   (nn.Linear layout [out,in])
  const float* b[7];
  int n_out;
};
__global__ void pack_ls_hashmlp_kernel(HashMlpPackArgs w, int planes, int f16, char* __restrict__ dst) {
  const NaMlpDesc d = {3, NA_ENC_HASH, 35, 0, 5, 256, w.n_out, 3, NA_ACT_LEAKY_RELU, NA_LAYOUT_GENERIC};
  const int dim_p = d.in_size + d.enc_dims + d.latent_size;  // 38
  const int frag_bytes = 1024 * planes;
  const int64_t nfrag_rg = 2 * kHashMlpPairs;
  const int64_t nelem = 4 * nfrag_rg * 512;
  const int64_t nbias = 4 * kNPhase * 256;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nelem + nbias; i += (int64_t)gridDim.x * blockDim.x) {
    if (i < nelem) {
      const int e = (int)(i & 7), l = (int)((i >> 3) & 63);
      const int64_t fg = i >> 9;
      const int rg = (int)(fg / nfrag_rg);
      int f = (int)(fg % nfrag_rg);
      int p = 0;
      while (f >= 2 * hashmlp_phase_pairs(p)) { f -= 2 * hashmlp_phase_pairs(p); ++p; }
      const int kappa = 8 * (l >> 5) + e;
      const float* W = w.w[p];  // p: 0 init, 1..5 layers.0..4, 6 out
      int row = -1, col = -1, in_dim = kHidden, out_dim = kHidden;
      if (p == 6) {
        if (f < 16) {  // fragment f = chunk c of this row group's tile
          row = out_row_map(d, 32 * (rg >> 1) + (l & 31));
          col = 16 * f + pi_perm(kappa);
        }
        in_dim = kHidden; out_dim = d.out_size;
      } else {
        const int q = f >> 1, t = f & 1;
        row = 32 * (2 * rg + t) + (l & 31);
        if (p == 0) { col = init_slot_feature(d, q, kappa); in_dim = dim_p; }
        else if (p == 1 || p == 4) {  // [hidden | init] in the reference's column order, the three init chunks first in the stream
          if (q < 3) { col = init_slot_feature(d, q, kappa); if (col >= 0) col += kHidden; }
          else col = 16 * (q - 3) + pi_perm(kappa);
          in_dim = kHidden + dim_p;
        } else { col = 16 * q + pi_perm(kappa); in_dim = kHidden; }
      }
      float v = 0.f;
      if (row >= 0 && row < out_dim && col >= 0 && col < in_dim) v = W[(int64_t)row * in_dim + col];
      char* o = dst + kHeaderBytes + kBiasBytes + ((int64_t)rg * nfrag_rg + (fg % nfrag_rg)) * frag_bytes + l * 16 + e * 2;
      const __bf16 h = f16 ? to_elem<NA_PREC_F16>(v) : (__bf16)v;
      *(uint16_t*)o = __builtin_bit_cast(uint16_t, h);
      if (planes == 2) {
        const __bf16 lo = (__bf16)(v - (float)h);
        *(uint16_t*)(o + 1024) = __builtin_bit_cast(uint16_t, lo);
      }
    } else {
      const int64_t q = i - nelem;
      const int k = (int)(q & 255), p = (int)((q >> 8) % kNPhase), rg = (int)((q >> 8) / kNPhase);
      const int slot = k >> 5, hi = (k >> 4) & 1, r = k & 15;
      const int rin = (r & 3) + 8 * (r >> 2) + 4 * hi;
      float v = 0.f;
      if (p < kHashMlpPhases && w.b[p] != nullptr) {
        if (p == 6) {
          const int row = slot < 1 ? out_row_map(d, 32 * (rg >> 1) + rin) : -1;   // (slot 0 = this row group's tile)
          if (row >= 0 && row < d.out_size) v = w.b[p][row];
        } else if (slot < 2) {
          v = w.b[p][32 * (2 * rg + slot) + rin];
        }
      }
      *(float*)(dst + kHeaderBytes + ((int64_t)rg * kNPhase + p) * 1024 + k * 4) = v;
    }
  }
}

// View head stream (MODEL 2): same element order, phases per view_phase_pairs; the last two pairs of the out phase are zero
struct ViewPackArgs {
  const float* w[6];  // init, layers.0..3, out
  const float* b[6];
};
__global__ void pack_ls_view_kernel(ViewPackArgs w, int planes, int f16, char* __restrict__ dst) {
  const NaMlpDesc d = {5, NA_ENC_NONE, 0, 64, 4, 256, 3, 3, NA_ACT_SIN, NA_LAYOUT_PLAIN_VIEW};
  const int frag_bytes = 1024 * planes;
  const int64_t nfrag_rg = 2 * kViewPairs;
  const int64_t nelem = 4 * nfrag_rg * 512;
  const int64_t nbias = 4 * kNPhase * 256;
  const int dim_p = d.in_size + d.latent_size;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nelem + nbias; i += (int64_t)gridDim.x * blockDim.x) {
    if (i < nelem) {
      const int e = (int)(i & 7), l = (int)((i >> 3) & 63);
      const int64_t fg = i >> 9;
      const int rg = (int)(fg / nfrag_rg);
      int f = (int)(fg % nfrag_rg);
      int p = 0;
      while (f >= 2 * view_phase_pairs(p)) { f -= 2 * view_phase_pairs(p); ++p; }
      const int kappa = 8 * (l >> 5) + e;
      const float* W = w.w[p];  // p: 0 init, 1..4 layers.0..3, 5 out
      int row = -1, col = -1, in_dim = 1, out_dim = 0;
      if (p == 5) {
        if (f < 16) {  // fragment f = chunk f of the single out tile; fragments 16..19 are padding
          row = out_row_map(d, l & 31);
          col = 16 * f + pi_perm(kappa);
          in_dim = kHidden; out_dim = d.out_size;
        }
      } else {
        const int q = f >> 1, t = f & 1;
        row = 32 * (2 * rg + t) + (l & 31);
        out_dim = kHidden;
        if (p == 0) { col = init_slot_feature(d, q, kappa); in_dim = dim_p; }  // q = 0..3 latent chunks, 4 = geometry chunk
        else if (p == 1) {  // init chunks from LDS, the 16 hidden chunks, then the geometry chunk
          if (q < 4) { col = init_slot_feature(d, q, kappa); if (col >= 0) col += kHidden; }
          else if (q < 4 + kHC) col = 16 * (q - 4) + pi_perm(kappa);
          else { col = init_slot_feature(d, 4, kappa); if (col >= 0) col += kHidden; }
          in_dim = kHidden + dim_p;
        } else { col = 16 * q + pi_perm(kappa); in_dim = kHidden; }
      }
      float v = 0.f;
      if (row >= 0 && row < out_dim && col >= 0 && col < in_dim) v = W[(int64_t)row * in_dim + col];
      char* o = dst + kHeaderBytes + kBiasBytes + ((int64_t)rg * nfrag_rg + (fg % nfrag_rg)) * frag_bytes + l * 16 + e * 2;
      const __bf16 h = f16 ? to_elem<NA_PREC_F16>(v) : (__bf16)v;
      *(uint16_t*)o = __builtin_bit_cast(uint16_t, h);
      if (planes == 2) {
        const __bf16 lo = (__bf16)(v - (float)h);
        *(uint16_t*)(o + 1024) = __builtin_bit_cast(uint16_t, lo);
      }
    } else {
      const int64_t q = i - nelem;
      const int k = (int)(q & 255), p = (int)((q >> 8) % kNPhase), rg = (int)((q >> 8) / kNPhase);
      const int slot = k >> 5, hi = (k >> 4) & 1, r = k & 15;
      const int rin = (r & 3) + 8 * (r >> 2) + 4 * hi;
      float v = 0.f;
      if (p < kViewPhases && w.b[p] != nullptr) {
        if (p == 5) {
          const int row = slot < 1 ? out_row_map(d, rin) : -1;
          if (row >= 0 && row < d.out_size) v = w.b[p][row];
        } else if (slot < 2) {
          v = w.b[p][32 * (2 * rg + slot) + rin];
        }
      }
      *(float*)(dst + kHeaderBytes + ((int64_t)rg * kNPhase + p) * 1024 + k * 4) = v;
    }
  }
}

// SIREN-VolSDF stream (MODEL 3): the SIREN SDF network (7 Linears) followed by the View head (6 Linears)
struct SirenPackArgs {
  const float* ws[7];  // sdf: init, layers.0..4, out
  const float* bs[7];
  const float* wv[6];  // view: init, layers.0..3, out
  const float* bv[6];
};
__global__ void pack_ls_siren_kernel(SirenPackArgs w, int planes, int f16, char* __restrict__ dst) {
  const NaMlpDesc d1 = {3, NA_ENC_NONE, 0, 0, 5, 256, 65, 3, NA_ACT_SIN, NA_LAYOUT_PLAIN_FIRST};
  const NaMlpDesc d2 = {5, NA_ENC_NONE, 0, 64, 4, 256, 3, 3, NA_ACT_SIN, NA_LAYOUT_PLAIN_VIEW};
  const int frag_bytes = 1024 * planes;
  const int64_t nfrag_rg = 2 * kSirenPairs;
  const int64_t nelem = 4 * nfrag_rg * 512;
  const int64_t nbias = 4 * kNPhase * 256;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nelem + nbias; i += (int64_t)gridDim.x * blockDim.x) {
    if (i < nelem) {
      const int e = (int)(i & 7), l = (int)((i >> 3) & 63);
      const int64_t fg = i >> 9;
      const int rg = (int)(fg / nfrag_rg);
      int f = (int)(fg % nfrag_rg);
      int p = 0;
      while (f >= 2 * siren_phase_pairs(p)) { f -= 2 * siren_phase_pairs(p); ++p; }
      const int kappa = 8 * (l >> 5) + e;
      const bool view = p >= 7;
      const int lp = view ? p - 7 : p;  // sdf: 0 init, 1..5 layers.0..4, 6 out;  view: 0 init, 1..4 layers.0..3, 5 out
      const float* W = view ? w.wv[lp] : w.ws[lp];
      int row = -1, col = -1, in_dim = 1, out_dim = 0;
      if (!view) {
        if (lp == 6) {  // row group rg holds tile min(rg, 2) of the 65 rows; fragment f = chunk f
          row = out_row_map(d1, 32 * (rg < 2 ? rg : 2) + (l & 31));
          col = 16 * f + pi_perm(kappa);
          in_dim = kHidden; out_dim = d1.out_size;
        } else {
          const int q = f >> 1, t = f & 1;
          row = 32 * (2 * rg + t) + (l & 31);
          out_dim = kHidden;
          if (lp == 0) { col = q == 0 ? init_slot_feature(d1, 0, kappa) : -1; in_dim = d1.in_size; }
          else if (lp == 1 || lp == 4) {
            if (q == 0) { col = init_slot_feature(d1, 0, kappa); if (col >= 0) col += kHidden; }
            else col = 16 * (q - 1) + pi_perm(kappa);
            in_dim = kHidden + d1.in_size;
          } else { col = 16 * q + pi_perm(kappa); in_dim = kHidden; }
        }
      } else {
        const int dim_p = d2.in_size + d2.latent_size;
        if (lp == 5) {
          if (f < 16) {
            row = out_row_map(d2, l & 31);
            col = 16 * f + pi_perm(kappa);
            in_dim = kHidden; out_dim = d2.out_size;
          }
        } else {
          const int q = f >> 1, t = f & 1;
          row = 32 * (2 * rg + t) + (l & 31);
          out_dim = kHidden;
          if (lp == 0) { col = init_slot_feature(d2, q, kappa); in_dim = dim_p; }
          else if (lp == 1) {
            if (q < 4) { col = init_slot_feature(d2, q, kappa); if (col >= 0) col += kHidden; }
            else if (q < 4 + kHC) col = 16 * (q - 4) + pi_perm(kappa);
            else { col = init_slot_feature(d2, 4, kappa); if (col >= 0) col += kHidden; }
            in_dim = kHidden + dim_p;
          } else { col = 16 * q + pi_perm(kappa); in_dim = kHidden; }
        }
      }
      float v = 0.f;
      if (row >= 0 && row < out_dim && col >= 0 && col < in_dim) v = W[(int64_t)row * in_dim + col];
      char* o = dst + kHeaderBytes + kBiasBytes + ((int64_t)rg * nfrag_rg + (fg % nfrag_rg)) * frag_bytes + l * 16 + e * 2;
      const __bf16 h = f16 ? to_elem<NA_PREC_F16>(v) : (__bf16)v;
      *(uint16_t*)o = __builtin_bit_cast(uint16_t, h);
      if (planes == 2) {
        const __bf16 lo = (__bf16)(v - (float)h);
        *(uint16_t*)(o + 1024) = __builtin_bit_cast(uint16_t, lo);
      }
    } else {
      const int64_t q = i - nelem;
      const int k = (int)(q & 255), p = (int)((q >> 8) % kNPhase), rg = (int)((q >> 8) / kNPhase);
      const int slot = k >> 5, hi = (k >> 4) & 1, r = k & 15;
      const int rin = (r & 3) + 8 * (r >> 2) + 4 * hi;
      float v = 0.f;
      if (p < kSirenPhases) {
        const bool view = p >= 7;
        const int lp = view ? p - 7 : p;
        const float* B = view ? w.bv[lp] : w.bs[lp];
        if (B != nullptr) {
          if (!view && lp == 6) {
            const int row = slot < 3 ? out_row_map(d1, 32 * slot + rin) : -1;
            if (row >= 0 && row < d1.out_size) v = B[row];
          } else if (view && lp == 5) {
            const int row = slot < 1 ? out_row_map(d2, rin) : -1;
            if (row >= 0 && row < d2.out_size) v = B[row];
          } else if (slot < 2) {
            v = B[32 * (2 * rg + slot) + rin];
          }
        }
      }
      *(float*)(dst + kHeaderBytes + ((int64_t)rg * kNPhase + p) * 1024 + k * 4) = v;
    }
  }
}

#endif  // NA_PREC_INST == 0

static std::atomic<uint32_t> g_lsx_launch_id{0};  // ids of the NA_PREC_F16X launches (range guard), shared by every schedule
// per-device hipFuncSetAttribute bookkeeping (the attribute is per device, not per thread)
template <int PREC, int MODEL = 0>
static int launch(Args& a, hipStream_t stream) {
  using C = Cfg<PREC>;
  auto kern = render_ls_kernel<PREC, MODEL>;
  static std::atomic<uint64_t> attr_done{0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { set_error("hipGetDevice failed"); return NA_EHIP; }
  const uint64_t bit = 1ull << (dev & 63);
  if (!(attr_done.load(std::memory_order_acquire) & bit)) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return NA_EHIP; }
    attr_done.fetch_or(bit, std::memory_order_release);
  }
  const int64_t wgs = (a.R + 1) / 2;  // at least one ray per sample group
  const int grid = wgs < 256 ? (int)wgs : 256;
  a.nG = 2 * grid;
  const int64_t rays_per_group = (a.R + a.nG - 1) / a.nG;
  a.npg = (int)((rays_per_group * a.nb + C::NBLK - 1) / C::NBLK);
  a.nb_magic = (1ull << 32) / (uint64_t)a.nb + 1;
  if ((int64_t)(a.npg + 1) * C::NBLK * a.nb >= (1ll << 32)) { set_error("na_render_plain_view_ls: batch too large"); return NA_EINVAL; }
  if constexpr (PREC == NA_PREC_F16X) {
    // ONE counter for all schedules: the flag ring is shared, and a stale id left by a saturated launch of one schedule
    // must never equal the id of a later launch of another (a per-instantiation counter did exactly that: the frame after
    // tests/test_gpu_range.py's saturated PlainNeRF launch came out poisoned in whichever VolSDF kernel reached the same count)
    uint32_t g = g_lsx_launch_id.fetch_add(1, std::memory_order_relaxed) + 1;
    if (g == 0) g = g_lsx_launch_id.fetch_add(1, std::memory_order_relaxed) + 1;  // (0 is the flag's initial value: never an id)
    a.sat_gen = g;
  } else {
    a.sat_gen = 0;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 2 * C::GROUP, stream, a);
  if constexpr (PREC == NA_PREC_F16X) {  // range guard: NaN output if any activation of this launch sat at the half clamp
    if constexpr (MODEL == 4 || MODEL == 5)
      hipLaunchKernelGGL(lsx_poison_kernel, dim3(grid_for((int64_t)a.T * a.R * a.y_ld, 256, 1024)), dim3(256), 0, stream, a.sat_gen, a.y,
                         (int64_t)a.T * a.R * a.y_ld);
    else
      hipLaunchKernelGGL(lsx_poison_kernel, dim3(grid_for(a.R * 3, 256, 256)), dim3(256), 0, stream, a.sat_gen, a.out, a.R * 3);
  }
  return check_launch("na_render_plain_view_ls");
}
}  // namespace ls
}  // namespace na
