// Generic fused SkipConnMLP forward kernel (instantiated per precision in mlp_fwd_inst.hip).
#pragma once
#include "mlp_layout.h"
#include "encoders.h"

namespace na {

// ================================================================================================ forward
struct MlpArgs {
  NaMlpDesc d;
  const char* packed;
  const float* p;
  const float* latent;
  int64_t p_ld, latent_ld;  // row pitches in floats (>= in_size / latent_size): inputs may be column slices of wider buffers
  const float* enc;
  float* y;
  int64_t N;
  int ngroups;  // ceil(N / (32*NWAVES))
  int out_tiles;
  uint32_t buf_bytes;
  HashRes res;
};

template <int PREC, int ACT, int ENC, int NI, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void mlp_forward_kernel(MlpArgs a, TileTab tab) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  WeightStream<NWAVES, slots_for(PREC, NI)> ws;
  const int npasses = ((int)blockIdx.x < a.ngroups) ? (a.ngroups - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
  ws.start(tab, a.packed + kHeaderBytes, a.packed + kHeaderBytes, smem, a.buf_bytes, npasses, wave, lane);

  for (int g = blockIdx.x; g < a.ngroups; g += gridDim.x) {
    const int64_t n_raw = ((int64_t)g * NWAVES + wave) * 32 + (lane & 31);
    const int64_t n = n_raw < a.N ? n_raw : a.N - 1;
    // `hi` is laundered once per group: everything the prologue derives from it (slot -> feature maps, basis
    // addresses) is otherwise loop-invariant, gets hoisted out of the group loop and lives in scratch (up to 344
    // spilled registers measured)
    int hi = lane >> 5;
    asm volatile("" : "+v"(hi));
    Frag<PREC> I[NI];
    // ---------------- init input fragments
    {
      const NaMlpDesc& d = a.d;
      int c0 = 0;  // first "rest" chunk
      float px = 0.f, py = 0.f, pz = 0.f;
      if constexpr (ENC == NA_ENC_HASH) {
        px = a.p[n * a.p_ld]; py = a.p[n * a.p_ld + 1]; pz = a.p[n * a.p_ld + 2];
        float f[16];
        hash_levels4(px, py, pz, (const float4*)a.enc, a.res, 4 * hi, f);
        float v0[8], v1[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { v0[e] = f[e]; v1[e] = f[8 + e]; }
        I[0] = make_frag<PREC>(v0);
        if constexpr (NI > 1) I[1] = make_frag<PREC>(v1);
        c0 = 2;
      } else if constexpr (ENC == NA_ENC_FOURIER) {
        const int F = d.enc_dims / 2, D = d.in_size;
        float xv[8];
        for (int q = 0; q < 8; ++q) xv[q] = q < D ? a.p[n * a.p_ld + q] : 0.f;
#pragma unroll
        for (int c = 0; c < NI; ++c) {
          if ((c & 1) == 0 && c + 1 < F / 8) {
            float sv[8], cv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              int freq = 16 * (c >> 1) + 8 * hi + e;
              float m = 0.f;
              for (int q = 0; q < D; ++q) m = q == 0 ? xv[0] * a.enc[freq] : fmaf(xv[q], a.enc[q * F + freq], m);
              // Cody-Waite + polynomial (1.6e-7 / 5e-7 for |m| <= 3e3; the fp32 argument itself carries 6e-5 at
              // |m| = 1e3) -- libm's sinf/cosf with their large-argument path cost 30-40 % of this kernel;
              // fast mode: hardware v_sin / v_cos on the fractional revolution
              if constexpr (PREC == NA_PREC_BF16) {
                const float rev = __builtin_amdgcn_fractf(m * 0.15915494309189535f);
                sv[e] = __builtin_amdgcn_sinf(rev);
                cv[e] = __builtin_amdgcn_cosf(rev);
              } else {
                sincos_cw(m, sv[e], cv[e]);
              }
            }
            I[c] = make_frag<PREC>(sv);
            if (c + 1 < NI) I[c + 1] = make_frag<PREC>(cv);
          }
        }
        c0 = F / 8;
      }
      // remaining chunks: slots hold consecutive features of the virtual row [p | (x again for hash) | latent].  The
      // slot -> feature map is written out per encoder with compile-time chunk/element indices and the loads are
      // select-based (clamped address, unconditional load, zero by select): the generic init_slot_feature() call
      // per element compiled to ~900 branches in this prologue.
      const int dim_rest = ENC == NA_ENC_HASH ? 6 + d.latent_size : d.in_size + d.latent_size;
      const float* prow = a.p + n * a.p_ld;
      const float* lrow = d.latent_size > 0 ? a.latent + n * a.latent_ld : prow;
      const int npos = ENC == NA_ENC_HASH ? 6 : d.in_size;  // leading position slots (hash: p then x, both = p)
#pragma unroll
      for (int c = 0; c < NI; ++c) {
        if (c >= c0) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int rho = 16 * (c - c0) + 8 * hi + e;  // index into the virtual row
            const bool ok = rho < dim_rest;
            const bool is_pos = rho < npos;
            int pi = ENC == NA_ENC_HASH ? (rho >= 3 ? rho - 3 : rho) : rho;
            pi = is_pos ? pi : 0;
            int li = rho - npos;
            li = (!is_pos && ok) ? li : 0;
            const float xp = prow[pi];
            const float xl = lrow[li];
            v[e] = ok ? (is_pos ? xp : xl) : 0.f;
          }
          I[c] = make_frag<PREC>(v);
        }
      }
    }
    // ---------------- network
    Frag<PREC> H[kHC];
    mlp_hidden_layers<PREC, ACT, 1, NI>(ws, a.d.num_layers, a.d.skip, I, H, lane);
    for (int j = 0; j < a.out_tiles; ++j) {
      f32x16 accs[1];
      mlp_out_tile<PREC, 1>(ws, H, lane, accs);
      const f32x16 acc = accs[0];
      if (n_raw < a.N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int f = 32 * j + acc_row(r, lane);
          if (f < a.d.out_size) a.y[n_raw * a.d.out_size + f] = acc[r];
        }
      }
    }
  }
}


int dispatch_forward_bf16(MlpArgs& a, const TileTab& tab, int NI, hipStream_t s);
int dispatch_forward_bf16x3(MlpArgs& a, const TileTab& tab, int NI, hipStream_t s);

}  // namespace na
