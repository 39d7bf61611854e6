// na_mlp_packed_bytes / na_mlp_pack / na_mlp_forward: SkipConnMLP (src/neural_blocks.py:204-296) with the
// encoder fused in front, on the register-resident MFMA engine of mlp_engine.h.
#include "mlp_forward_kernel.h"

namespace na {

// ================================================================================================ pack
enum { LK_INIT = 0, LK_HIDDEN = 1, LK_HIDDEN_SKIP = 2, LK_OUT = 3 };

__device__ __forceinline__ uint16_t bf16_bits(float v) {
  __bf16 h = (__bf16)v;
  return __builtin_bit_cast(uint16_t, h);
}

// One launch per Linear.  W [out_dim, in_dim] fp32 (nn.Linear layout), bias [out_dim].
__global__ void pack_layer_kernel(NaMlpDesc d, int kind, const float* __restrict__ W, const float* __restrict__ bias,
                                  int in_dim, int out_dim, int ntile, int nfrag, int NI, int planes, int f16,
                                  char* __restrict__ dst) {
  const int tile_bytes = (nfrag * planes + 1) * 1024;
  const int64_t nelem = (int64_t)ntile * nfrag * 512;  // bf16 elements per plane
  const int64_t total = nelem + (int64_t)ntile * 256;  // + bias block floats (256 per tile, 32 used)
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    if (i < nelem) {
      int e = (int)(i & 7);
      int l = (int)((i >> 3) & 63);
      int c = (int)((i >> 9) % nfrag);
      int j = (int)((i >> 9) / nfrag);
      int kappa = 8 * (l >> 5) + e;
      int rho = 32 * j + (l & 31);
      int row = kind == LK_OUT ? out_row_map(d, rho) : rho;
      int col;
      if (kind == LK_INIT) col = init_slot_feature(d, c, kappa);
      else if (c < kHC) col = 16 * c + pi_perm(kappa);
      else { col = init_slot_feature(d, c - kHC, kappa); if (col >= 0) col += kHidden; }
      float w = 0.f;
      if (row >= 0 && row < out_dim && col >= 0 && col < in_dim) w = W[(int64_t)row * in_dim + col];
      char* p = dst + (int64_t)j * tile_bytes + 1024 + (int64_t)c * planes * 1024 + l * 16 + e * 2;
      __bf16 h = f16 ? to_elem<NA_PREC_F16>(w) : (__bf16)w;
      *(uint16_t*)p = __builtin_bit_cast(uint16_t, h);
      if (planes == 2) *(uint16_t*)(p + 1024) = bf16_bits(w - (float)h);
    } else {
      int64_t q = i - nelem;
      int j = (int)(q >> 8), k = (int)(q & 255);
      float v = 0.f;
      if (k < 32) {
        int hi = k >> 4, r = k & 15;
        int rho = 32 * j + (r & 3) + 8 * (r >> 2) + 4 * hi;
        int row = kind == LK_OUT ? out_row_map(d, rho) : rho;
        if (row >= 0 && row < out_dim && bias != nullptr) v = bias[row];
      }
      *(float*)(dst + (int64_t)j * tile_bytes + k * 4) = v;
    }
  }
}

struct HeaderWords { uint32_t w[2 * kMaxTilesPerMlp + 1]; };
__global__ void pack_header_kernel(HeaderWords h, int nwords, uint32_t* __restrict__ dst) {
  // static indexing only (a dynamically indexed by-value struct would be copied to scratch)
#pragma unroll
  for (int i = 0; i < 2 * kMaxTilesPerMlp + 1; ++i)
    if (i < nwords && threadIdx.x == 0) dst[i] = h.w[i];
}

}  // namespace na

// ================================================================================================ C ABI
using namespace na;

extern "C" {

size_t na_mlp_packed_bytes(const NaMlpDesc* desc, int precision) {
  if (desc == nullptr || (precision != NA_PREC_BF16 && precision != NA_PREC_BF16X3 && precision != NA_PREC_F16)) return 0;
  if (mlp_unsupported_reason(*desc) != nullptr) return 0;
  uint32_t blocks = 0;
  if (build_tiles(*desc, precision, nullptr, &blocks) < 0) return 0;
  return kHeaderBytes + (size_t)blocks * 1024;
}

int na_mlp_pack(const NaMlpDesc* desc, int precision, const float* const* weights, const float* const* biases,
                void* packed, void* stream) {
  NA_REQUIRE(desc && weights && biases && packed, NA_ENULL, "na_mlp_pack: null pointer");
  NA_REQUIRE(precision == NA_PREC_BF16 || precision == NA_PREC_BF16X3 || precision == NA_PREC_F16, NA_EUNSUPPORTED,
             "na_mlp_pack: precision %d", precision);
  const char* why = mlp_unsupported_reason(*desc);
  NA_REQUIRE(why == nullptr, NA_EUNSUPPORTED, "na_mlp_pack: %s", why);
  const NaMlpDesc& d = *desc;
  const int P = planes_of(precision), NI = effective_ni(d);
  const int dim_p = d.in_size + d.enc_dims + d.latent_size;
  const int L = d.num_layers;
  for (int i = 0; i < L + 2; ++i)
    NA_REQUIRE(weights[i] != nullptr, NA_ENULL, "na_mlp_pack: weights[%d] is null", i);
  {
    HeaderWords h;
    uint32_t blocks = 0;
    int nt = build_tiles(d, precision, h.w + 1, &blocks);
    NA_REQUIRE(nt > 0, NA_EUNSUPPORTED, "na_mlp_pack: too many tiles");
    h.w[0] = (uint32_t)nt;
    hipLaunchKernelGGL(pack_header_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, h, 2 * nt + 1, (uint32_t*)packed);
  }
  char* dst = (char*)packed + kHeaderBytes;
  auto launch = [&](int kind, const float* W, const float* b, int in_dim, int out_dim, int ntile, int nfrag) {
    int64_t total = (int64_t)ntile * nfrag * 512 + (int64_t)ntile * 256;
    hipLaunchKernelGGL(pack_layer_kernel, dim3(grid_for(total, 256, 2048)), dim3(256), 0, (hipStream_t)stream, d, kind, W,
                       b, in_dim, out_dim, ntile, nfrag, NI, P, precision == NA_PREC_F16 ? 1 : 0, dst);
    dst += (size_t)ntile * (nfrag * P + 1) * 1024;
  };
  launch(LK_INIT, weights[0], biases[0], dim_p, kHidden, 8, NI);
  for (int i = 0; i < L; ++i) {
    bool sk = layer_has_skip(d, i);
    launch(sk ? LK_HIDDEN_SKIP : LK_HIDDEN, weights[1 + i], biases[1 + i], sk ? kHidden + dim_p : kHidden, kHidden, 8,
           kHC + (sk ? NI : 0));
  }
  launch(LK_OUT, weights[L + 1], biases[L + 1], kHidden, d.out_size, out_tiles(d), kHC);
  return check_launch("na_mlp_pack");
}

int na_mlp_forward(const NaMlpDesc* desc, int precision, const void* packed, const float* p, const float* latent,
                   const float* enc_params, int64_t N, float* y, void* stream) {
  return na_mlp_forward_ld(desc, precision, packed, p, desc ? desc->in_size : 0, latent, desc ? desc->latent_size : 0,
                           enc_params, N, y, stream);
}

int na_mlp_forward_ld(const NaMlpDesc* desc, int precision, const void* packed, const float* p, int64_t p_ld,
                      const float* latent, int64_t latent_ld, const float* enc_params, int64_t N, float* y,
                      void* stream) {
  return na_mlp_forward_mip(desc, precision, packed, p, p_ld, latent, latent_ld, enc_params, nullptr, N, y, stream);
}

int na_mlp_forward_mip(const NaMlpDesc* desc, int precision, const void* packed, const float* p, int64_t p_ld,
                       const float* latent, int64_t latent_ld, const float* enc_params, const NaMipDesc* mip, int64_t N,
                       float* y, void* stream) {
  NA_REQUIRE(desc && packed && p && y, NA_ENULL, "na_mlp_forward: null pointer");
  int gen = 0;
  if (mip != nullptr) {
    NA_REQUIRE(mip->rays && mip->ts, NA_ENULL, "na_mlp_forward_mip: null rays / ts");
    NA_REQUIRE(mip->kind == 0 || mip->kind == 1, NA_EUNSUPPORTED, "na_mlp_forward_mip: kind %d", mip->kind);
    gen = 6 * (mip->max_deg - mip->min_deg);
    NA_REQUIRE(mip->max_deg > mip->min_deg && mip->max_deg - mip->min_deg <= 42 && gen <= desc->latent_size, NA_EINVAL,
               "na_mlp_forward_mip: %d IPE columns do not fit latent_size %d", gen, desc->latent_size);
    NA_REQUIRE(mip->B >= 1 && mip->H >= 2 && mip->W >= 1 && mip->T >= 1 && N == (int64_t)mip->T * mip->B * mip->H * mip->W,
               NA_EINVAL, "na_mlp_forward_mip: N must be T*B*H*W (radii need H >= 2 rows)");
  }
  NA_REQUIRE(p_ld >= desc->in_size && (desc->latent_size == gen || latent_ld >= desc->latent_size - gen), NA_EINVAL,
             "na_mlp_forward: row pitch smaller than the row (p_ld=%lld, latent_ld=%lld)", (long long)p_ld,
             (long long)latent_ld);
  NA_REQUIRE(precision == NA_PREC_BF16 || precision == NA_PREC_BF16X3 || precision == NA_PREC_F16, NA_EUNSUPPORTED,
             "na_mlp_forward: precision %d", precision);
  const char* why = mlp_unsupported_reason(*desc);
  NA_REQUIRE(why == nullptr, NA_EUNSUPPORTED, "na_mlp_forward: %s", why);
  NA_REQUIRE(desc->layout == NA_LAYOUT_GENERIC, NA_EUNSUPPORTED,
             "na_mlp_forward: PLAIN_* layouts are only consumed by na_render_plain_view");
  NA_REQUIRE(desc->latent_size == gen || latent != nullptr, NA_ENULL, "na_mlp_forward: latent_size>0 needs latent");
  NA_REQUIRE(desc->enc_kind == NA_ENC_NONE || enc_params != nullptr, NA_ENULL, "na_mlp_forward: encoder needs enc_params");
  NA_REQUIRE(N >= 0, NA_EINVAL, "na_mlp_forward: N=%lld", (long long)N);
  if (N == 0) return NA_OK;
  MlpArgs a;
  a.d = *desc; a.packed = (const char*)packed; a.p = p; a.latent = latent; a.enc = enc_params; a.y = y; a.N = N;
  a.p_ld = p_ld; a.latent_ld = latent_ld;
  a.mip = MipGen{nullptr, nullptr, 0, 0, 0, 0, 0, 0, 0, 0.f};
  if (mip != nullptr)
    a.mip = MipGen{mip->rays, mip->ts, mip->B, mip->H, mip->W, mip->T, mip->kind, mip->min_deg, mip->max_deg - mip->min_deg,
                   mip->t_end};
  a.out_tiles = out_tiles(*desc);
  a.res = hash_resolutions();
  TileTab tab;
  const int nt = tile_count(*desc);
  tab.hdr0 = (const uint32_t*)packed;
  tab.hdr1 = tab.hdr0;
  tab.ntiles = nt;
  tab.split = nt;
  const int NI = effective_ni(*desc);
  a.buf_bytes = (uint32_t)((kHC + NI) * planes_of(precision) + 1) * 1024;
  if (precision == NA_PREC_BF16) return dispatch_forward_bf16(a, tab, NI, (hipStream_t)stream);
  if (precision == NA_PREC_F16) return dispatch_forward_f16(a, tab, NI, (hipStream_t)stream);
  return dispatch_forward_bf16x3(a, tab, NI, (hipStream_t)stream);
}

}  // extern "C"
