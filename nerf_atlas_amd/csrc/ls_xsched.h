// NA_PREC_F16X weight streams built from schedule tables (XSched: which Linear every pair / record / bias block of a MODEL's
// phase sequence belongs to) -- one set of pack kernels for all schedules.  Included by the NA_PREC_INST == 3 unit only.
#pragma once
#include "ls_pack.h"

namespace na {
namespace ls {
// ---- NA_PREC_F16X weight streams (layout: namespace x above), built from a schedule table: which Linear every pair / record /
// bias block belongs to.  One set of kernels for the four schedules.
// column of Linear rd.lin's weight matrix that sits in k-slot kappa of chunk c of the record's K64 group; -1 = zero
__device__ __forceinline__ int xrec_col(const XSched& sc, const XRecD& rd, int c, int kappa) {
  if (rd.kind == 0) return 64 * rd.q + 16 * c + pi_perm(kappa);  // hidden feature (the skip layers store [hidden | init])
  if (rd.kind == 6) return rd.off + 16 * c + pi_perm(kappa);     // 64 columns from `off` on, in the slot order of accumulator rows
  if (rd.kind == 7) {
    // MODEL 8: [hash' | x | refl_latent] of PosLinearView.pos, weight columns [p 3 | x 3 + hash 32 | intermediate 64 | refl_latent]
    // (src/nerf.py:352-358, src/refl.py:277): the hash desc's slots, + refl_latent column j in slot 6 + j of chunk 2
    if (c == 2 && kappa >= 6 && kappa < 6 + sc.n_rl) return rd.off + 38 + 64 + (kappa - 6);
    const int col = init_slot_feature(sc.desc[sc.lin[rd.lin].desc], c, kappa);
    return col < 0 ? col : col + rd.off;
  }
  if (rd.kind == 3 || rd.kind == 4) {
    // Fourier group q as the MODEL 5 generator lays it out: slot s = 8 c + e of lane half h holds frequency f = 32 q + 16 h + s / 2,
    // its sine (s even) or cosine (s odd).  Reference columns: [p | sin(128) | cos(128)] (src/neural_blocks.py:36-55, 283-287)
    const NaMlpDesc& d = sc.desc[sc.lin[rd.lin].desc];
    const int F = d.enc_dims / 2, s = 8 * c + (kappa & 7), f = 32 * rd.q + 16 * (kappa >> 3) + (s >> 1);
    return (rd.kind == 4 ? kHidden : 0) + d.in_size + ((s & 1) ? F + f : f);
  }
  if (rd.kind == 5) {
    // IPE group q (0, 1) as the MODEL 6 generator lays it out: slot s = 8 c + e < 24 of lane half h holds the (degree, axis)
    // pair pidx = 24 q + 12 h + s / 2 (the fourth chunk of both groups is padding: the two generating waves do the same work),
    // its sine feature (s even: latent column pidx) or cosine feature (s odd: column 48 + pidx); src/utils.py:23-27 layout
    // [sin | cos], degree-major
    const int s = 8 * c + (kappa & 7), h = kappa >> 3;
    const int pidx = s < 24 ? 24 * rd.q + 12 * h + (s >> 1) : -1;
    return pidx < 0 ? -1 : rd.off + ((s & 1) ? 48 : 0) + pidx;
  }
  int col = init_slot_feature(sc.desc[sc.lin[rd.lin].desc], c, kappa);
  if (col >= 0 && rd.kind == 2) col += kHidden;
  return col < 0 ? col : col + rd.off;
}
// weight row held by lane l of tile t of row group rg; -1 = zero
__device__ __forceinline__ int xrec_row(const XSched& sc, const XRecD& rd, int rg, int t, int l) {
  const NaMlpDesc& d = sc.desc[sc.lin[rd.lin].desc];
  if (rd.out_mode == 0) return 32 * (2 * rg + t) + (l & 31);
  if (rd.out_mode == 5) return 32 * (2 * (rg & 1) + t) + (l & 31);  // a 128-wide Linear split by block (MODEL 8's view MLP)
  if (rd.out_mode == 3) return rg < 2 ? out_row_map(d, 32 * t + (l & 31)) : (t == 0 ? out_row_map(d, 64 + (l & 31)) : -1);
  // 4: PosLinearView.pos.out (src/refl.py:275-276: rows 0..2 colour, 3..66 intermediate), split by block like mode 3: row groups
  // 0, 1 hold the two intermediate tiles, row groups 2, 3 the colour rows
  if (rd.out_mode == 4) return rg < 2 ? 3 + 32 * t + (l & 31) : (t == 0 && (l & 31) < 3 ? (l & 31) : -1);
  return t == 0 ? out_row_map(d, (rd.out_mode == 1 ? 32 * (rg < 2 ? rg : 2) : 0) + (l & 31)) : -1;
}
__global__ void pack_lsx_f16_kernel(XSched sc, char* __restrict__ dst) {
  // one thread per 16-bit element of the f16 planes of the pairs and of the records' f16 fragments
  const int srg = sc.npair * x::PAIRB + sc.nrec * x::REC;
  const int64_t npair_e = 4ll * sc.npair * 2 * 512;  // [rg][pair][tile][lane][8]
  const int64_t nrec_e = 4ll * sc.nrec * 8 * 512;    // [rg][rec][tile*4+chunk][lane][8]
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < npair_e + nrec_e; i += (int64_t)gridDim.x * blockDim.x) {
    if (i < npair_e) {
      const int e = (int)(i & 7), l = (int)((i >> 3) & 63), t = (int)((i >> 9) & 1);
      const int pi = (int)((i >> 10) % sc.npair), rg = (int)((i >> 10) / sc.npair);
      const XPairD pd = sc.pair[pi];
      const XLin L = sc.lin[pd.lin];
      const int kap = 8 * (l >> 5) + e;
      int col = pd.q == 32 ? (kap < 6 ? kap : kap < 6 + sc.n_rl ? 6 + 64 + (kap - 6) : -1) : init_slot_feature(sc.desc[L.desc], pd.q, kap);
      if (col >= 0 && pd.skip) col += sc.desc[L.desc].hidden;
      const int row = 32 * (2 * (pd.q == 32 ? (rg & 1) : rg) + t) + (l & 31);  // (q = 32: a Linear split by block)
      float v = 0.f;
      if (col >= 0 && col < L.in_dim && row < L.out_dim) v = L.W[(int64_t)row * L.in_dim + col];
      const __bf16 h = to_elem<NA_PREC_F16X>(v);
      const __bf16 lo = to_elem<NA_PREC_F16X, false>(v - from_elem<NA_PREC_F16X>(h));
      char* o = dst + kHeaderBytes + kBiasBytes + (int64_t)rg * srg + pi * x::PAIRB + t * 2048 + l * 16 + e * 2;
      *(uint16_t*)o = __builtin_bit_cast(uint16_t, h);
      *(uint16_t*)(o + 1024) = __builtin_bit_cast(uint16_t, lo);
    } else {
      const int64_t k = i - npair_e;
      const int e = (int)(k & 7), l = (int)((k >> 3) & 63), f = (int)((k >> 9) & 7);
      const int ri = (int)((k >> 12) % sc.nrec), rg = (int)((k >> 12) / sc.nrec);
      const int t = f >> 2, c = f & 3;
      const XRecD rd = sc.rec[ri];
      const XLin L = sc.lin[rd.lin];
      const int col = xrec_col(sc, rd, c, 8 * (l >> 5) + e);
      const int row = xrec_row(sc, rd, rg, t, l);
      float v = 0.f;
      if (row >= 0 && row < L.out_dim && col >= 0 && col < L.in_dim) v = L.W[(int64_t)row * L.in_dim + col];
      char* o = dst + kHeaderBytes + kBiasBytes + (int64_t)rg * srg + sc.npair * x::PAIRB + (int64_t)ri * x::REC + f * 1024 + l * 16 + e * 2;
      *(uint16_t*)o = __builtin_bit_cast(uint16_t, to_elem<NA_PREC_F16X>(v));
    }
  }
}
// one thread per (row group, record, tile, lane): the lane's 32 weights of the K64 group -> WL6 and the scale bytes of WL6 and
// of WT6 (which the render kernel derives from the f16 fragments with that scale).  WL6 pairs with the activations' T plane:
// slot order = what v_cvt_scalef32_2xpk16_fp6_f32 gives it: slot 2 r <-> (producer tile 0, register r), slot 2 r + 1 <->
// (producer tile 1, register r), i.e. hidden feature 64 Q + 32 tt + (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
__global__ void pack_lsx_fp6_kernel(XSched sc, char* __restrict__ dst) {
  const int srg = sc.npair * x::PAIRB + sc.nrec * x::REC;
  const int64_t n = 4ll * sc.nrec * 2 * 64;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int l = (int)(i & 63), t = (int)((i >> 6) & 1);
    const int ri = (int)((i >> 7) % sc.nrec), rg = (int)((i >> 7) / sc.nrec);
    const int h = l >> 5;
    const XRecD rd = sc.rec[ri];
    const XLin L = sc.lin[rd.lin];
    const int row = xrec_row(sc, rd, rg, t, l);
    f32x16 wl0, wl1;
    float mt = 0.f, ml = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      // (hidden groups: columns 64 q + (r & 3) + 8 (r >> 2) + 4 h and + 32, the producer tiles' registers r)
      const int ca = xrec_col(sc, rd, r >> 3, 8 * h + (r & 7)), cb = xrec_col(sc, rd, 2 + (r >> 3), 8 * h + (r & 7));
      float a = 0.f, b = 0.f;
      if (row >= 0 && row < L.out_dim) {
        if (ca >= 0 && ca < L.in_dim) a = L.W[(int64_t)row * L.in_dim + ca];
        if (cb >= 0 && cb < L.in_dim) b = L.W[(int64_t)row * L.in_dim + cb];
      }
      const float ah = from_elem<NA_PREC_F16X>(to_elem<NA_PREC_F16X>(a)), bh = from_elem<NA_PREC_F16X>(to_elem<NA_PREC_F16X>(b));
      wl0[r] = a - ah;
      wl1[r] = b - bh;
      mt = fmaxf(mt, fmaxf(fabsf(ah), fabsf(bh)));  // (of the f16 values: that is what the kernel converts)
      ml = fmaxf(ml, fmaxf(fabsf(wl0[r]), fabsf(wl1[r])));
    }
    // block scale 2^(floor(log2 max) - 2): the largest element lands in [4, 8) (saturating at 7.5)
    auto scale_byte = [](float m) { const int ev = (int)(__builtin_bit_cast(uint32_t, m) >> 23); return ev > 3 ? ev - 2 : 1; };
    const int et = scale_byte(mt), el = scale_byte(ml);
    // (by construction here, where a few registers cost nothing: destination disjoint from every operand)
    const x::i32x6 L6 = x::cvt_fp6_disjoint(wl0, wl1, __builtin_bit_cast(float, (uint32_t)el << 23));
    char* rec = dst + kHeaderBytes + kBiasBytes + (int64_t)rg * srg + sc.npair * x::PAIRB + (int64_t)ri * x::REC;
    // {WL6 t0 | WL6 t1}: dword d of the lane's twelve sits in 16-byte part d >> 2 (three lane-linear parts)
#pragma unroll
    for (int d = 0; d < 6; ++d) {
      const int dd = 6 * t + d;
      *(uint32_t*)(rec + 8192 + (dd >> 2) * 1024 + l * 16 + (dd & 3) * 4) = (uint32_t)L6[d];
    }
    uint8_t* scb = (uint8_t*)(rec + 8192 + 3072 + l * 4);
    scb[2 * t] = (uint8_t)el;
    scb[2 * t + 1] = (uint8_t)et;
  }
}
// bias blocks: the layout of pack_ls_kernel ([row group][phase] 1-KiB blocks, floats [slot][hi(2)][16])
__global__ void pack_lsx_bias_kernel(XSched sc, char* __restrict__ dst) {
  const int64_t nbias = 4 * kNPhase * 256;
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < nbias; q += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(q & 255), p = (int)((q >> 8) % kNPhase), rg = (int)((q >> 8) / kNPhase);
    const int slot = k >> 5, hi = (k >> 4) & 1, r = k & 15;
    const int rin = (r & 3) + 8 * (r >> 2) + 4 * hi;
    float v = 0.f;
    if (p < sc.nphase) {
      const XLin L = sc.lin[sc.bias_lin[p]];
      const int mode = sc.bias_mode[p];  // 0 hidden rows, 1 / 3 out Linear with 3 tiles, 2 out Linear, one tile
      if (L.B != nullptr) {
        if (mode == 0) { if (slot < 2 && 32 * (2 * rg + slot) + rin < L.out_dim) v = L.B[32 * (2 * rg + slot) + rin]; }
        else if (mode == 5) { if (slot < 2) v = L.B[32 * (2 * (rg & 1) + slot) + rin]; }
        else if (mode == 4) {
          const int row = slot < 2 ? 3 + 32 * slot + rin : (slot == 2 && rin < 3 ? rin : -1);
          if (row >= 0 && row < L.out_dim) v = L.B[row];
        } else {
          const int row = slot < ((mode == 1 || mode == 3) ? 3 : 1) ? out_row_map(sc.desc[L.desc], 32 * slot + rin) : -1;
          if (row >= 0 && row < L.out_dim) v = L.B[row];
        }
      }
    }
    *(float*)(dst + kHeaderBytes + ((int64_t)rg * kNPhase + p) * 1024 + k * 4) = v;
  }
}
__global__ void pack_lsx_header_kernel(uint32_t* __restrict__ dst, uint32_t units) {
  if (threadIdx.x == 0) { dst[0] = kMagic; dst[1] = (uint32_t)NA_PREC_F16X; dst[2] = units; dst[3] = kNPhase; }
}

// `nl` Linears of one SkipConnMLP appended to the schedule: init (NI chunk pairs), hidden Linears (skip layers take the NI init
// chunks again, + kHidden), out.  geo: the View MLP's fifth init chunk is its own pair behind the init / skip chunks.
static void xs_add_mlp(XSched& sc, const NaMlpDesc& d, const float* const* w, const float* const* b, int nl, int ni, bool geo,
                       int out_mode, bool init_rec = false) {
  const int di = sc.ndesc++;
  sc.desc[di] = d;
  const int dim_p = d.in_size + d.enc_dims + d.latent_size;
  const int l0 = sc.nlin;
  for (int i = 0; i < nl; ++i) {
    const bool first = i == 0, last = i == nl - 1;
    const bool skip = !first && !last && ((i - 1) % d.skip) == 0 && (i - 1) != d.num_layers - 1;
    XLin L;
    L.W = w[i]; L.B = b[i]; L.desc = di;
    L.in_dim = first ? dim_p : skip ? kHidden + dim_p : kHidden;
    L.out_dim = last ? d.out_size : kHidden;
    sc.lin[sc.nlin++] = L;
    sc.bias_lin[sc.nphase] = (int8_t)(l0 + i);
    sc.bias_mode[sc.nphase++] = (int8_t)(last ? out_mode : 0);
    if (first || skip) {
      // init_rec: the (<= 4) init chunks as ONE record in front of the Linear's hidden records, consumed from the init region
      if (init_rec) sc.rec[sc.nrec++] = XRecD{(int8_t)(l0 + i), 0, 0, (int8_t)(skip ? 2 : 1), 0};
      else for (int q = 0; q < ni; ++q) sc.pair[sc.npair++] = XPairD{(int8_t)(l0 + i), (int8_t)q, (int8_t)(skip ? 1 : 0)};
    }
    if (!first) {
      for (int q = 0; q < 4; ++q) sc.rec[sc.nrec++] = XRecD{(int8_t)(l0 + i), (int8_t)q, (int8_t)(last ? out_mode : 0), 0, 0};
    }
    if ((first || skip) && geo) sc.pair[sc.npair++] = XPairD{(int8_t)(l0 + i), 4, (int8_t)(skip ? 1 : 0)};
  }
}

int render_lsx_pack(int model, const float* const* w0, const float* const* b0, const float* const* w1, const float* const* b1,
                    char* packed, hipStream_t stream, int n_out) {
  XSched sc;
  memset(&sc, 0, sizeof(sc));
  const NaMlpDesc view = {5, NA_ENC_NONE, 0, 64, 4, 256, 3, 3, NA_ACT_SIN, NA_LAYOUT_PLAIN_VIEW};
  if (model == 0) {
    const NaMlpDesc first = {3, NA_ENC_HASH, 35, 0, 4, 256, 65, 3, NA_ACT_LEAKY_RELU, NA_LAYOUT_PLAIN_FIRST};
    xs_add_mlp(sc, first, w0, b0, 6, 3, false, 3, true);
    xs_add_mlp(sc, view, w1, b1, 6, 4, true, 2, true);
  } else if (model == 1) {
    const NaMlpDesc tiny = {3, NA_ENC_NONE, 0, 0, 6, 256, 4, 3, NA_ACT_LEAKY_RELU, NA_LAYOUT_GENERIC};
    xs_add_mlp(sc, tiny, w0, b0, 8, 1, false, 2);
  } else if (model == 2) {
    xs_add_mlp(sc, view, w0, b0, 6, 4, true, 2, true);
  } else if (model == 6) {
    // PlainNeRF(view) + mip.  Column layouts (src/neural_blocks.py:283-287: [p | enc(p) | latent]): first [p 3 | x 3 + hash 32 |
    // IPE 96] (134), skip layer [hidden 256 | the same]; View [x y z elev azim | IPE 96 | intermediate 64] (165)
    // (src/nerf.py:352-358: latent = cat(mip, cat(intermediate, refl_latent))), skip layer [hidden 256 | the same].
    // The slot maps of the [hash | x] and latent groups come from descs WITHOUT the IPE columns; `off` puts them in place.
    const NaMlpDesc first = {3, NA_ENC_HASH, 35, 0, 4, 256, 65, 3, NA_ACT_LEAKY_RELU, NA_LAYOUT_PLAIN_FIRST};
    sc.desc[0] = first; sc.desc[1] = view; sc.ndesc = 2;
    for (int m = 0; m < 2; ++m) {
      const float* const* w = m == 0 ? w0 : w1;
      const float* const* b = m == 0 ? b0 : b1;
      const int dim_p = m == 0 ? 134 : 165, ipe0 = m == 0 ? 38 : 5, grp_off = m == 0 ? 0 : 96;
      for (int i = 0; i < 6; ++i) {  // init, layers.0..3, out
        const bool fst = i == 0, last = i == 5, skip = i == 1;
        XLin L;
        L.W = w[i]; L.B = b[i]; L.desc = m;
        L.in_dim = fst ? dim_p : skip ? kHidden + dim_p : kHidden;
        L.out_dim = last ? (m == 0 ? 65 : 3) : kHidden;
        const int li = sc.nlin;
        sc.lin[sc.nlin++] = L;
        sc.bias_lin[sc.nphase] = (int8_t)li;
        sc.bias_mode[sc.nphase++] = (int8_t)(last ? (m == 0 ? 3 : 2) : 0);
        if (fst || skip) {
          const int so = skip ? kHidden : 0;
          // consumption order: the [hash | x] / latent group (init region), then (skip layers) the four hidden groups, then the IPE groups
          sc.rec[sc.nrec++] = XRecD{(int8_t)li, 0, 0, (int8_t)(skip ? 2 : 1), (int16_t)grp_off};
          if (skip) for (int q = 0; q < 4; ++q) sc.rec[sc.nrec++] = XRecD{(int8_t)li, (int8_t)q, 0, 0, 0};
          for (int q = 0; q < 2; ++q) sc.rec[sc.nrec++] = XRecD{(int8_t)li, (int8_t)q, 0, 5, (int16_t)(so + ipe0)};
          if (m == 1) sc.pair[sc.npair++] = XPairD{(int8_t)li, 4, (int8_t)(skip ? 1 : 0)};
        } else {
          for (int q = 0; q < 4; ++q) sc.rec[sc.nrec++] = XRecD{(int8_t)li, (int8_t)q, (int8_t)(last ? (m == 0 ? 3 : 2) : 0), 0, 0};
        }
      }
    }
  } else if (model == 5) {
    const NaMlpDesc fmlp = {3, NA_ENC_FOURIER, 256, 0, 6, 256, 65, 3, NA_ACT_LEAKY_RELU, NA_LAYOUT_GENERIC};
    sc.desc[sc.ndesc++] = fmlp;
    const int dim_p = 3 + 256;
    for (int i = 0; i < 8; ++i) {  // init, layers.0..5, out
      const bool first = i == 0, last = i == 7, skip = i == 1 || i == 4;
      XLin L;
      L.W = w0[i]; L.B = b0[i]; L.desc = 0;
      L.in_dim = first ? dim_p : skip ? kHidden + dim_p : kHidden;
      L.out_dim = last ? 65 : kHidden;
      sc.lin[sc.nlin++] = L;
      sc.bias_lin[sc.nphase] = (int8_t)i;
      sc.bias_mode[sc.nphase++] = (int8_t)(last ? 1 : 0);
      if (!first) for (int q = 0; q < 4; ++q) sc.rec[sc.nrec++] = XRecD{(int8_t)i, (int8_t)q, (int8_t)(last ? 1 : 0), 0, 0};
      if (first || skip) {
        for (int q = 0; q < 4; ++q) sc.rec[sc.nrec++] = XRecD{(int8_t)i, (int8_t)q, 0, (int8_t)(skip ? 4 : 3), 0};
        sc.pair[sc.npair++] = XPairD{(int8_t)i, 16, (int8_t)(skip ? 1 : 0)};  // the position chunk: init chunk F / 8 of the Fourier layout
      }
    }
  } else if (model == 7 || model == 8) {
    // PlainNeRF + Positional (7) / PosLinearView (8): `first` exactly as in MODEL 0, then the head's Linears (w1 / b1: 7: init,
    // layers.0..4, out;  8: pos.init, pos.layers.0..1, pos.out, view.init, view.layers.0..1, view.out).  n_out carries n_rl (8).
    const NaMlpDesc first = {3, NA_ENC_HASH, 35, 0, 4, 256, 65, 3, NA_ACT_LEAKY_RELU, NA_LAYOUT_PLAIN_FIRST};
    xs_add_mlp(sc, first, w0, b0, 6, 3, false, 3, true);
    const int n_rl = model == 8 ? n_out : 0;
    sc.n_rl = n_rl;
    const int npos = model == 7 ? 7 : 4;
    const NaMlpDesc pos = {3, NA_ENC_HASH, 35, 0, npos - 2, 256, model == 7 ? 3 : 67, 3, NA_ACT_LEAKY_RELU, NA_LAYOUT_GENERIC};
    const int dpos = sc.ndesc++;
    sc.desc[dpos] = pos;
    const int dim_pos = 38 + 64 + n_rl;
    for (int i = 0; i < npos; ++i) {
      const bool fst = i == 0, last = i == npos - 1, skip = i == 1 || (model == 7 && i == 4);
      XLin L;
      L.W = w1[i]; L.B = b1[i]; L.desc = dpos;
      L.in_dim = fst ? dim_pos : skip ? kHidden + dim_pos : kHidden;
      L.out_dim = last ? pos.out_size : kHidden;
      const int li = sc.nlin;
      sc.lin[sc.nlin++] = L;
      const int om = last ? (model == 7 ? 2 : 4) : 0;
      sc.bias_lin[sc.nphase] = (int8_t)li;
      sc.bias_mode[sc.nphase++] = (int8_t)om;
      const int so = skip ? kHidden : 0;
      // consumption order: the [hash' | x] group (init region), the four hidden groups (skip layers), the latent group
      if (fst || skip) sc.rec[sc.nrec++] = XRecD{(int8_t)li, 0, 0, (int8_t)7, (int16_t)so};
      if (!fst) for (int q = 0; q < 4; ++q) sc.rec[sc.nrec++] = XRecD{(int8_t)li, (int8_t)q, (int8_t)om, 0, 0};
      if (fst || skip) sc.rec[sc.nrec++] = XRecD{(int8_t)li, 0, 0, (int8_t)6, (int16_t)(so + 38)};
    }
    if (model == 8) {
      // view: SkipConnMLP(in 6, latent 64 + n_rl + 64, 2 x 128, sin) -> 1; weight columns [x 3 | dir 3 | latent 64 | refl_latent | intermediate 64]
      const NaMlpDesc vw = {6, NA_ENC_NONE, 0, 128 + n_rl, 2, 128, 1, 3, NA_ACT_SIN, NA_LAYOUT_GENERIC};
      const int dv = sc.ndesc++;
      sc.desc[dv] = vw;
      const int dim_v = 6 + 128 + n_rl, H = 128;
      for (int i = 0; i < 4; ++i) {
        const bool fst = i == 0, last = i == 3, skip = i == 1;
        XLin L;
        L.W = w1[4 + i]; L.B = b1[4 + i]; L.desc = dv;
        L.in_dim = fst ? dim_v : skip ? H + dim_v : H;
        L.out_dim = last ? 1 : H;
        const int li = sc.nlin;
        sc.lin[sc.nlin++] = L;
        sc.bias_lin[sc.nphase] = (int8_t)li;
        sc.bias_mode[sc.nphase++] = (int8_t)(last ? 2 : 5);
        const int so = skip ? H : 0;
        // consumption order = the K64 groups 0..3 of a block as the kernel lays them out: view.init [latent, intermediate] (groups 2, 3);
        // view.L0 [sin(latent), sin(intermediate), hidden 0, hidden 1]; view.L1 / out [hidden 0, hidden 1]; then the geometry pair
        if (fst || skip) {
          sc.rec[sc.nrec++] = XRecD{(int8_t)li, 0, 5, (int8_t)6, (int16_t)(so + 6)};
          sc.rec[sc.nrec++] = XRecD{(int8_t)li, 0, 5, (int8_t)6, (int16_t)(so + 6 + 64 + n_rl)};
          sc.pair[sc.npair++] = XPairD{(int8_t)li, 32, (int8_t)(skip ? 1 : 0)};
        }
        if (!fst) for (int q = 0; q < 2; ++q) sc.rec[sc.nrec++] = XRecD{(int8_t)li, (int8_t)q, (int8_t)(last ? 2 : 5), 0, 0};
      }
    }
  } else if (model == 4) {
    const NaMlpDesc hashmlp = {3, NA_ENC_HASH, 35, 0, 5, 256, n_out, 3, NA_ACT_LEAKY_RELU, NA_LAYOUT_GENERIC};
    xs_add_mlp(sc, hashmlp, w0, b0, 7, 3, false, 2, true);
  } else {
    const NaMlpDesc siren = {3, NA_ENC_NONE, 0, 0, 5, 256, 65, 3, NA_ACT_SIN, NA_LAYOUT_PLAIN_FIRST};
    xs_add_mlp(sc, siren, w0, b0, 7, 1, false, 3);
    xs_add_mlp(sc, view, w1, b1, 6, 4, true, 2, true);
  }
  if (sc.npair != x::npair(model) || sc.nrec != x::nrec(model)) {
    set_error("render_lsx_pack: schedule of model %d has %d pairs / %d records, the kernel expects %d / %d", model, sc.npair,
              sc.nrec, x::npair(model), x::nrec(model));
    return NA_EINVAL;
  }
  hipLaunchKernelGGL(pack_lsx_header_kernel, dim3(1), dim3(64), 0, stream, (uint32_t*)packed, (uint32_t)x::hdr_units(model));
  const int64_t ne = 4ll * sc.npair * 2 * 512 + 4ll * sc.nrec * 8 * 512;
  hipLaunchKernelGGL(pack_lsx_f16_kernel, dim3(grid_for(ne, 256, 4096)), dim3(256), 0, stream, sc, packed);
  hipLaunchKernelGGL(pack_lsx_fp6_kernel, dim3(grid_for(4ll * sc.nrec * 2 * 64, 64, 4096)), dim3(64), 0, stream, sc, packed);
  hipLaunchKernelGGL(pack_lsx_bias_kernel, dim3(grid_for(4 * kNPhase * 256, 256, 4096)), dim3(256), 0, stream, sc, packed);
  return check_launch("na_render_*_ls_pack (f16x)");
}
}  // namespace ls
}  // namespace na
