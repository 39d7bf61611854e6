// Generic fused SkipConnMLP forward kernel (instantiated per precision in mlp_fwd_inst.hip).
#pragma once
#include "mlp_layout.h"
#include "encoders.h"
#ifndef NA_MLP_OUT_NT
#define NA_MLP_OUT_NT 0  // 1: the network's output rows leave with non-temporal stores (measured: the chains of configs 3 / 4 / 5m,
                         // whose next kernel reads them at once, get 1-4 % SLOWER: 594 -> 568, 575 -> 568, 554 -> 542 Msamples/s in bf16)
#endif

namespace na {

// ================================================================================================ forward
// IPE latent generated in the prologue (SURVEY config 3; src/utils.py:83-140, hook src/nerf.py:256-261): the leading
// 6*nd latent columns of sample n = t * (B*H*W) + ray are the integrated positional encoding of that conical-frustum /
// cylinder sample, computed from the rays of the crop instead of being read from a [N, 6 nd] tensor in HBM
struct MipGen {
  const float* rays;  // [B,H,W,6]
  const float* ts;    // [T]
  int B, H, W, T, kind, min_deg, nd;
  float t_end;
};

struct MlpArgs {
  NaMlpDesc d;
  MipGen mip;
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

template <int PREC, int ACT, int ENC, int NI, int NWAVES, int GEN = 0>
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
        if constexpr (GEN != 0) {
          // two levels in flight (64 result + 16 offset registers instead of 128 + 64 address registers): with the IPE
          // state live next to it, the all-at-once gather of hash_levels4 spilled its results to scratch one by one
#pragma unroll
          for (int k2 = 0; k2 < 4; k2 += 2) {
            HashGather h0, h1;
            hash_level_issue(px, py, pz, (const float4*)a.enc, hi ? a.res.n[4 + k2] : a.res.n[k2], 4 * hi + k2, h0);
            hash_level_issue(px, py, pz, (const float4*)a.enc, hi ? a.res.n[5 + k2] : a.res.n[1 + k2], 4 * hi + k2 + 1, h1);
            float f0[4], f1[4];
            hash_level_finish(h0, f0);
            hash_level_finish(h1, f1);
#pragma unroll
            for (int q = 0; q < 4; ++q) { f[4 * k2 + q] = f0[q]; f[4 * k2 + 4 + q] = f1[q]; }
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
          hash_levels4(px, py, pz, (const float4*)a.enc, a.res, 4 * hi, f);
        }
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
            // this lane's 8 frequencies are consecutive columns of the basis: two 16-byte loads per input dimension
            // instead of eight 4-byte ones (same products and the same summation order per frequency)
            float mv[8];
            {
              const int freq0 = 16 * (c >> 1) + 8 * hi;
              for (int q = 0; q < D; ++q) {
                const f32x4 b0 = *(const f32x4*)(a.enc + q * F + freq0), b1 = *(const f32x4*)(a.enc + q * F + freq0 + 4);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                  const float be = e < 4 ? b0[e & 3] : b1[e & 3];
                  mv[e] = q == 0 ? xv[0] * be : fmaf(xv[q], be, mv[e]);
                }
              }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float m = mv[e];
              // Cody-Waite + polynomial (1.6e-7 / 5e-7 for |m| <= 3e3; the fp32 argument itself carries 6e-5 at
              // |m| = 1e3) -- libm's sinf/cosf with their large-argument path cost 30-40 % of this kernel;
              // fast mode: hardware v_sin / v_cos on the fractional revolution
              if constexpr (PREC != NA_PREC_BF16X3) {
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
      const int gen = GEN ? 6 * a.mip.nd : 0;  // latent columns produced here instead of being loaded
      const float* lrow = d.latent_size > gen ? a.latent + n * a.latent_ld : prow;
      float gm0 = 0.f, gm1 = 0.f, gm2 = 0.f, gc0 = 0.f, gc1 = 0.f, gc2 = 0.f;
      if constexpr (GEN != 0) {
        const MipGen& m = a.mip;
        const int64_t R = (int64_t)m.B * m.H * m.W;
        const int t = (int)(n / R);
        const int64_t r = n - (int64_t)t * R;
        const int wq = (int)(r % m.W), hq = (int)((r / m.W) % m.H), b = (int)(r / ((int64_t)m.W * m.H));
        const float rad = mip_radius(m.rays, m.H, m.W, b, hq, wq);
        const MipGauss gs = mip_gaussian(m.rays + r * 6, rad, m.ts[t], t < m.T - 1 ? m.ts[t + 1] : mip_last_edge(m.ts, m.T, m.t_end), m.kind);
        gm0 = gs.m0; gm1 = gs.m1; gm2 = gs.m2; gc0 = gs.c0; gc1 = gs.c1; gc2 = gs.c2;
      }
      // leading position slots (hash: p then x, both = p).  The IPE-prologue instantiation without an encoder is the View
      // head (x, y, z, elev, azim): a compile-time 5 lets every chunk behind the first drop its position loads
      const int npos = ENC == NA_ENC_HASH ? 6 : (GEN != 0 ? 5 : d.in_size);
#pragma unroll
      for (int c = 0; c < NI; ++c) {
        if (c >= c0) {
          float v[8];
          // (uniform) does this chunk hold any generated column?  The View MLP's trailing chunks carry only loaded ones.
          const bool chunk_gen = GEN != 0 && 16 * (c - c0) < npos + gen && 16 * (c - c0) + 16 > npos;
          // this lane's 8 slots are 8 consecutive columns of the virtual row: when all of them are STORED latent columns
          // they come in as two 16-byte loads (rows are only dword-aligned in general) instead of 8 + 8 four-byte loads a
          // row pitch apart across the lanes (mip `first` shape with the latent in HBM: 4.76 -> 4.07 ms per 5.12 M samples)
          const int rho0 = 16 * (c - c0) + 8 * hi;
          if (rho0 >= npos + gen && rho0 + 8 <= dim_rest) {
            typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
            const float* src = lrow + (rho0 - npos - gen);
            const f32x4u lo4 = *(const f32x4u*)src, hi4 = *(const f32x4u*)(src + 4);
            v[0] = lo4[0]; v[1] = lo4[1]; v[2] = lo4[2]; v[3] = lo4[3];
            v[4] = hi4[0]; v[5] = hi4[1]; v[6] = hi4[2]; v[7] = hi4[3];
          } else if (rho0 >= dim_rest) {  // padding behind the row: nothing to load
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.f;
          } else
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int rho = 16 * (c - c0) + 8 * hi + e;  // index into the virtual row
            const bool ok = rho < dim_rest;
            const bool is_pos = rho < npos;
            int pi = ENC == NA_ENC_HASH ? (rho >= 3 ? rho - 3 : rho) : rho;
            pi = is_pos ? pi : 0;
            int li = rho - npos;
            li = (!is_pos && ok) ? li : 0;
            float xp;
            if constexpr (ENC == NA_ENC_HASH) {
              // the six position slots (p, then x = p again) sit in the first rest chunk of the hi = 0 lanes: registers, not loads
              const float sel = (e % 3) == 0 ? px : (e % 3) == 1 ? py : pz;
              xp = (c == c0 && e < 6) ? sel : 0.f;  // (is_pos is false everywhere else)
            } else {
              xp = prow[pi];
            }
            float xl;
            if constexpr (GEN != 0) {
              const bool is_gen = li < gen;
              const float xg = chunk_gen ? mip_feature<PREC != NA_PREC_BF16X3>(gm0, gm1, gm2, gc0, gc1, gc2, is_gen ? li : 0,
                                                                             a.mip.nd, a.mip.min_deg) : 0.f;
              // (the hash instantiation is PlainNeRF.first: its whole latent is generated, there is no stored column to load)
              if constexpr (ENC == NA_ENC_HASH) xl = xg;
              else xl = is_gen ? xg : lrow[is_gen ? 0 : li - gen];
            } else {
              // (uniform) no stored latent at all: nothing to load (lrow would alias the position row)
              xl = d.latent_size > 0 ? lrow[li] : 0.f;
            }
            v[e] = ok ? (is_pos ? xp : xl) : 0.f;
          }
          I[c] = make_frag<PREC>(v);
          // one chunk at a time: interleaving the transcendental chains of several chunks costs more registers than the
          // 44 the finished fragments already hold
          if constexpr (GEN != 0) __builtin_amdgcn_sched_barrier(0);
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
        // registers 4q..4q+3 of a lane are 4 CONSECUTIVE output features of its sample (acc_row): one 16-byte store per
        // q instead of four 4-byte stores one row pitch apart across the lanes (rows are only 4-byte aligned when
        // out_size is odd, e.g. VolSDF's 257: dword-aligned vector stores)
        typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
        float* yrow = a.y + n_raw * a.d.out_size;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int f0 = 32 * j + acc_row(4 * q, lane);
          if (f0 + 4 <= a.d.out_size) {
            const f32x4u v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
            if (NA_MLP_OUT_NT) __builtin_nontemporal_store(v, (f32x4u*)(yrow + f0)); else *(f32x4u*)(yrow + f0) = v;
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (f0 + i < a.d.out_size) yrow[f0 + i] = acc[4 * q + i];
          }
        }
      }
    }
  }
}


int dispatch_forward_bf16(MlpArgs& a, const TileTab& tab, int NI, hipStream_t s);
int dispatch_forward_bf16x3(MlpArgs& a, const TileTab& tab, int NI, hipStream_t s);
int dispatch_forward_f16(MlpArgs& a, const TileTab& tab, int NI, hipStream_t s);

}  // namespace na
