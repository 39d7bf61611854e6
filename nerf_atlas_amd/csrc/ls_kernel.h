// render_ls_kernel<PREC, MODEL>: the layer-synchronous engine's ONE kernel -- prologue, pass loop, epilogue -- with the phase
// sequence of each MODEL (its "schedule") in a file of its own (ls_sched_*.inc, included inside the pass loop).  What the engine
// is and why: the header comment of render_ls.hip and DESIGN.md 3b / 3c; the building blocks (fragment I/O, the weight ring, the
// MFMA phases, the f16x namespace x) are in ls_engine.h.
#pragma once
#include "ls_engine.h"

#ifndef NA_LS_TRAIN_EXP
#define NA_LS_TRAIN_EXP 0  // timing experiments on MODEL 9 (tools/ls_variant.py; wrong or missing rows): 1 no plane stores, 2 the LDS staging
                           // alone, 4 the store instructions into a 64-KiB window (no HBM stream), 16 no bias loads.  1 048 576 samples, one
                           // box: 1: 3.20 ms, 2: 3.28, 4: 3.42, shipped 4.08 -- the staging costs 2.5 %, the instructions 4 %, the stream the rest
#endif
#ifndef NA_LS_TRAIN_AUX
// cache policy of MODEL 9's row stores: 2 = nt (non-temporal).  The 2.7 GB a step writes otherwise pass through the L2 that holds the
// weight stream and the hash tables: measured (one box, 262 144 / 1 048 576 samples) 0: 1.27 / 4.93 ms, 2: 1.12 / 4.00, 17 (sc0 sc1): 1.23 / 4.62
#define NA_LS_TRAIN_AUX 2
#endif
namespace na {
namespace ls {

template <int PREC, int MODEL = 0>
__global__ __launch_bounds__(512) void render_ls_kernel(Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using C = Cfg<PREC>;
  constexpr int NB = C::NBLK, FR = C::FRAG;
  constexpr bool M0 = MODEL == 0 || MODEL == 6 || MODEL == 7 || MODEL == 8 || MODEL == 9;  // the PlainNeRF schedules (0: View head, 6: + mip, 7: Positional, 8: PosLinearView, 9: 0 as the training forward)
  constexpr bool TRAIN = MODEL == 9;  // MODEL 0 that also leaves every Linear's output rows in HBM for the backward pass (train_store below)
  constexpr bool HEAD2 = MODEL == 7 || MODEL == 8;  // the reflectance head has a hash encoder of its own (src/refl.py:230-290)
  constexpr bool MIP = MODEL == 6;
  constexpr int PPP = PREC == NA_PREC_F16X ? x::hdr_units(MODEL)
                      : MODEL == 1 ? kTinyPairs : MODEL == 2 ? kViewPairs : MODEL == 3 ? kSirenPairs : MODEL == 4 ? kHashMlpPairs : kPairsPerPass;  // pairs per pass and row group
  // rays / elaz are read with scalar (SMEM) loads below; both were written by kernels that ran just before this one, into
  // buffers the allocator recycles from call to call: drop whatever the scalar cache still holds of those addresses
  __builtin_amdgcn_s_dcache_inv();
  {
    // a stream packed for another precision or schedule (or not a stream at all) would be consumed without any fault:
    // refuse it -- NaN colour for every ray -- instead of rendering garbage (header: na_render_*_ls_pack)
    const uint32_t* hdr = (const uint32_t*)a.packed;
    if (hdr[0] != kMagic || hdr[1] != (uint32_t)PREC || hdr[2] != (uint32_t)PPP) {
      const float nan = __builtin_nanf("");
      if constexpr (MODEL == 4 || MODEL == 5) {  // (rows to HBM: there is no colour buffer)
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)a.T * a.R * a.y_ld; i += (int64_t)gridDim.x * blockDim.x) a.y[i] = nan;
      } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.R * 3; i += (int64_t)gridDim.x * blockDim.x) a.out[i] = nan;
      }
      return;
    }
  }
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int rg = wv & 3, g = wv >> 2;
  const int hi = lane >> 5, ln = lane & 31;
  char* hb = smem + g * C::GROUP;
  char* ib = hb + C::HREG;
  const bool owner = rg < NB;           // this wave owns block rg of its group (encoder, out layers, compositing)
  const int blk = owner ? rg : rg - NB;  // non-owners (bf16x3: rg 2,3) shadow a block to keep their weight ring in step
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.packed, 0, a.packed_size, 0x00020000);
  const int wvoff = kHeaderBytes + kBiasBytes + rg * (PPP * C::PAIR) + lane * 16;
  const int bias_rg = kHeaderBytes + rg * (kNPhase * 1024);  // scalar offset of this row group's bias blocks
  // Fifth init chunk of the View MLP for block b of this group (x, y, z, elev, azim in the hi = 0 lanes; src/refl.py:
  // 190-207).  The ray of a block is wave-uniform: its origin, direction and elev/azim (from the pre-kernel) are read once
  // per pass (in the short epilogue of first.out) and kept in SGPRs (geo_u, readfirstlane), together with the block's step
  // offset; the two MFMA phases that consume the chunk only load t (or the explicit position) of their lane at their start.
  float geo_u[NB][8];   // ox oy oz dx dy dz elev azim of block b's ray (uniform)
  int geo_t0[NB];       // first step of block b
  int geo_ray[NB];
  float own_u[6];       // origin | direction of the ray of this wave's OWN block (encoder + compositing), per pass
  float own_dn = 0.f;   // |direction| of that ray
  float prev_dn = 0.f;  // the same for the wave's block of the previous pass (its compositing runs one pass later)
  // Work distribution: sample group G = 2 * workgroup + g renders the rays G, G + nG, G + 2 nG, ... one after the other,
  // each as its nb 32-step blocks in step order, NB blocks per pass.  At any moment the launch works on ~nG consecutive
  // rays (hash-table locality in L2 as before), and the blocks of one ray pass through one group in order, so the
  // transmittance is carried from block to block inside the kernel (no per-block partials, no second launch).
  // XCD-aware order: hardware workgroup w runs on XCD w % 8 (round-robin dispatch, each XCD has its own L2), so logical
  // workgroup lw = (w % 8) * (n / 8) + w / 8 puts CONSECUTIVE sample groups -- neighbouring rays, the same hash-table
  // lines and weight fragments -- on one XCD instead of on all eight: HBM traffic per launch 173 -> 93 MB (bf16),
  // 210 -> 150 MB (bf16x3) at equal speed (profiles/r02).  Grids that are not a multiple of 8 keep the identity.
#ifndef NA_LS_NO_XCD_MAP
  const int nwg = (int)gridDim.x;
  const int lw = (nwg % 8 == 0) ? ((int)blockIdx.x % 8) * (nwg / 8) + (int)blockIdx.x / 8 : (int)blockIdx.x;
#else
  const int lw = (int)blockIdx.x;
#endif
  const int G = __builtin_amdgcn_readfirstlane(lw * 2 + g);
  struct Loc { int ray, tb; bool ok; };
  auto locate = [&](int pl, int b) {
    const int sidx = pl * NB + b;
    const int k = (int)(((uint64_t)(uint32_t)sidx * a.nb_magic) >> 32);  // sidx / nb (exact: sidx * nb < 2^32), scalar ALU
    Loc L;
    L.tb = sidx - k * a.nb;
    const int64_t r = G + (int64_t)k * a.nG;
    L.ok = r < a.R;
    L.ray = L.ok ? (int)r : (int)a.R - 1;
    if (!L.ok) L.tb = a.nb - 1;
    return L;
  };
  auto geo_setup = [&](int pl) {
    // scalar (SMEM) loads: the addresses are wave-uniform and rays / elaz are read-only for the whole launch.  All loads
    // and their wait sit in ONE asm statement, so the compiler can neither read nor spill a destination in flight.
    // (ox oy) (oz dx) (dy dz) (elev azim): 8-byte loads (a ray is 24 bytes), each into a 64-bit scalar -- vector-typed
    // SGPR asm outputs are mis-split by the compiler (element 1 read from element 0's register)
    uint64_t ra[NB], rb[NB], rc[NB], e2[NB];
    const float* ry[NB];
    const float* ea[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const Loc L = locate(pl, b);
      const int ray = __builtin_amdgcn_readfirstlane(L.ray);
      ry[b] = a.rays + (int64_t)ray * 6;
      ea[b] = a.elaz + (int64_t)ray * 2;
      geo_t0[b] = __builtin_amdgcn_readfirstlane(L.tb) * 32;
      geo_ray[b] = ray;
    }
#define NA_LS_GEO_LOAD(oa, ob, oc, oe, pr, pe)                                                              \
  "s_load_dwordx2 " oa ", " pr ", 0x0\n\ts_load_dwordx2 " ob ", " pr ", 0x8\n\ts_load_dwordx2 " oc ", " pr ", 0x10\n\t" \
  "s_load_dwordx2 " oe ", " pe ", 0x0\n\t"
    if constexpr (NB == 4)
      asm volatile(NA_LS_GEO_LOAD("%0", "%1", "%2", "%3", "%16", "%17") NA_LS_GEO_LOAD("%4", "%5", "%6", "%7", "%18", "%19")
                   NA_LS_GEO_LOAD("%8", "%9", "%10", "%11", "%20", "%21") NA_LS_GEO_LOAD("%12", "%13", "%14", "%15", "%22", "%23")
                   "s_waitcnt lgkmcnt(0)"
                   : "=&s"(ra[0]), "=&s"(rb[0]), "=&s"(rc[0]), "=&s"(e2[0]), "=&s"(ra[1]), "=&s"(rb[1]), "=&s"(rc[1]), "=&s"(e2[1]),
                     "=&s"(ra[2]), "=&s"(rb[2]), "=&s"(rc[2]), "=&s"(e2[2]), "=&s"(ra[3]), "=&s"(rb[3]), "=&s"(rc[3]), "=&s"(e2[3])
                   : "s"(ry[0]), "s"(ea[0]), "s"(ry[1]), "s"(ea[1]), "s"(ry[2]), "s"(ea[2]), "s"(ry[3]), "s"(ea[3]));
    else
      asm volatile(NA_LS_GEO_LOAD("%0", "%1", "%2", "%3", "%8", "%9") NA_LS_GEO_LOAD("%4", "%5", "%6", "%7", "%10", "%11")
                   "s_waitcnt lgkmcnt(0)"
                   : "=&s"(ra[0]), "=&s"(rb[0]), "=&s"(rc[0]), "=&s"(e2[0]), "=&s"(ra[1 % NB]), "=&s"(rb[1 % NB]), "=&s"(rc[1 % NB]),
                     "=&s"(e2[1 % NB])
                   : "s"(ry[0]), "s"(ea[0]), "s"(ry[1 % NB]), "s"(ea[1 % NB]));
#undef NA_LS_GEO_LOAD
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      auto lo = [](uint64_t v) { return __builtin_bit_cast(float, (uint32_t)v); };
      auto hi32 = [](uint64_t v) { return __builtin_bit_cast(float, (uint32_t)(v >> 32)); };
      geo_u[b][0] = lo(ra[b]); geo_u[b][1] = hi32(ra[b]);
      geo_u[b][2] = lo(rb[b]); geo_u[b][3] = hi32(rb[b]);
      geo_u[b][4] = lo(rc[b]); geo_u[b][5] = hi32(rc[b]);
      geo_u[b][6] = lo(e2[b]); geo_u[b][7] = hi32(e2[b]);
    }
  };
  // the ray of this wave's own block of pass `pl`: three scalar loads at the top of EP (short-lived SGPRs; the group-wide
  // table above is filled later, in the epilogue of first.out, for the two View phases -- holding it for the whole pass
  // made the compiler park it in scratch memory and re-store it every pass)
  auto own_setup = [&](int pl) {
    const Loc L = locate(pl, blk);
    const float* ry = a.rays + (int64_t)__builtin_amdgcn_readfirstlane(L.ray) * 6;
    uint64_t ra, rb, rc;
    asm volatile("s_load_dwordx2 %0, %3, 0x0\n\ts_load_dwordx2 %1, %3, 0x8\n\ts_load_dwordx2 %2, %3, 0x10\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(ra), "=&s"(rb), "=&s"(rc) : "s"(ry));
    auto lo = [](uint64_t v) { return __builtin_bit_cast(float, (uint32_t)v); };
    auto hi32 = [](uint64_t v) { return __builtin_bit_cast(float, (uint32_t)(v >> 32)); };
    own_u[0] = lo(ra); own_u[1] = hi32(ra); own_u[2] = lo(rb); own_u[3] = hi32(rb); own_u[4] = lo(rc); own_u[5] = hi32(rc);
    const float dx = own_u[3], dy = own_u[4], dz = own_u[5];
    own_dn = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sqrtf((dx * dx + dy * dy) + dz * dz))));
  };
  struct GeoRaw { float x, y, z; };  // t (x) or the explicit position of this lane's sample
  auto geo_load = [&](int b) -> GeoRaw {
    const int t = geo_t0[b] + ln;
    const int tc = t < a.T ? t : a.T - 1;
    GeoRaw r;
    if (a.pts != nullptr) {
      const float* p = a.pts + ((int64_t)tc * a.R + geo_ray[b]) * 3;
      r.x = p[0]; r.y = p[1]; r.z = p[2];
    } else {
      r.x = a.ts[(int64_t)geo_ray[b] * a.ts_stride + tc]; r.y = r.z = 0.f;
    }
    return r;
  };
  auto geo_make = [&](int b, const GeoRaw& r, bool act) -> Frag<PREC> {
    float px = r.x, py = r.y, pz = r.z;
    if (a.pts == nullptr) {
      const float tt = r.x;
      px = geo_u[b][0] + tt * geo_u[b][3]; py = geo_u[b][1] + tt * geo_u[b][4]; pz = geo_u[b][2] + tt * geo_u[b][5];
    }
    float v4[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v4[e] = 0.f;
    if (hi == 0) { v4[0] = px; v4[1] = py; v4[2] = pz; v4[3] = geo_u[b][6]; v4[4] = geo_u[b][7]; }
    Frag<PREC> f = make_frag<PREC>(v4);
    if (act) frag_activate<PREC, NA_ACT_SIN>(f);
    return f;
  };
  struct Geom {
    int64_t ray;
    bool item_ok, t_ok;
    int t;
    float px, py, pz, dist, dx, dy, dz;
  };
  // geometry of this lane's sample of the wave's own block of the CURRENT pass (own_setup(pass) has run)
  // ts[t] and ts[t + 1] of this lane's step of block `blk` of pass pl: requested first thing in EP, before the scalar loads
  // of the ray (whose wait would otherwise sit in front of them)
  struct TsPair { float t0, t1; };
  auto ts_load = [&](int pl) {
    const Loc L = locate(pl, blk);
    const int t = L.tb * 32 + ln;
    const int tc = t < a.T ? t : a.T - 1;
    TsPair r;
    const float* tsr = a.ts + (int64_t)L.ray * a.ts_stride;
    r.t0 = tsr[tc];
    r.t1 = tsr[tc < a.T - 1 ? tc + 1 : tc];
    return r;
  };
  auto geom = [&](int pass, int b, const TsPair& tp) {
    Geom q;
    const Loc L = locate(pass, b);
    q.item_ok = L.ok;
    q.ray = L.ray;
    const int tb = L.tb;
    q.t = tb * 32 + ln;
    q.t_ok = q.t < a.T;
    const int tc = q.t_ok ? q.t : a.T - 1;
    const float (&u)[6] = own_u;
    q.dx = u[3]; q.dy = u[4]; q.dz = u[5];
    const float tt = tp.t0;
    if (a.pts != nullptr) {
      const float* p = a.pts + ((int64_t)tc * a.R + q.ray) * 3;
      q.px = p[0]; q.py = p[1]; q.pz = p[2];
    } else {
      q.px = u[0] + tt * q.dx; q.py = u[1] + tt * q.dy; q.pz = u[2] + tt * q.dz;
    }
    const float d = tc < a.T - 1 ? fmaxf(tp.t1 - tt, 1e-5f) : 1e10f;
    q.dist = d * own_dn;
    return q;
  };
  // what the compositing of block `blk` of the PREVIOUS pass needs: issued at the top of EP next to the loads above
  struct Prev { int64_t ray; int t; bool ok, t_ok; float dist; };
  auto prev_geom = [&](int pl, const TsPair& tp) {
    Prev q;
    const Loc L = locate(pl, blk);
    q.ok = L.ok;
    q.ray = L.ray;
    q.t = L.tb * 32 + ln;
    q.t_ok = q.t < a.T;
    const int tc = q.t_ok ? q.t : a.T - 1;
    const float tt = tp.t0;
    const float d = tc < a.T - 1 ? fmaxf(tp.t1 - tt, 1e-5f) : 1e10f;
    q.dist = d * prev_dn;
    return q;
  };

  float w_local = 0.f;  // block-local weight of this lane's sample, until `combine` knows the transmittance in front
  // Where block b's compositing partials (P, S_rgb, W_head) wait for `combine`.  MODEL 0 composites a block at the END of
  // its own pass, in the view.out phase (round 4), while other waves may still read the hidden region: the slots are the head
  // of the block's fourth init chunk (the last latent chunk: dead once view.L0 has run, not written by EP, rewritten by the
  // epilogue of first.out).  The other schedules composite in the next pass's exposed phase, into the idle hidden region.
  auto part_of = [&](int b) -> float* {
    if constexpr (M0 && PREC == NA_PREC_F16X) return (float*)(ib + b * x::KQ + 3 * 1024);  // (f16 chunk 3 of the block's init group)
    return M0 ? (float*)(ib + (b * 4 + 3) * FR) : (float*)hb + b * kPartialFloats;
  };
  // compositing of block rg of pass `pass` (src/nerf.py:22-27,60-80); the hi=0 half holds the samples
  auto composite = [&](const Prev& q, const f32x16& oc, float density) {
    // (MODEL 8: the colour arrives finished -- (sigmoid(linear) / 2 + 0.5) * act(pos), src/refl.py:284-290)
    const int sgk = MODEL == 8 ? (int)NA_SIG_IDENTITY : a.sigmoid_kind;
    const float cr = fast_sigmoid_kind(oc[0], sgk);
    const float cg = fast_sigmoid_kind(oc[1], sgk);
    const float cb = fast_sigmoid_kind(oc[2], sgk);
    // (MODEL 2: `density` is VolSDF's Laplace density, used as it is: src/nerf.py:1004-1006, softplus = False)
    const float sigma = (MODEL == 2 || MODEL == 3) ? fmaxf(density, 0.f) : fast_softplus(density - 1.0f);
    const float alpha = q.t_ok ? 1.0f - fast_exp(-sigma * q.dist) : 0.f;
    const float f = (1.0f - alpha) + 1e-10f;
    // exclusive product scan over the 32 steps of the block: shift by one lane (lane 0 of each half: 1), then scan
    float fs = NA_DPP(1.0f, f, 0x138, 0xF);  // wave_shr:1
    if (ln == 0) fs = 1.0f;
    const float excl = scan32_mul(fs);
    const float w = alpha * excl;
    const float P = excl * f;                 // lane 31: product of the whole block
    const float sr = scan32_add(w * cr), sg = scan32_add(w * cg), sb = scan32_add(w * cb);
    const float wh = scan32_add((q.t < a.T - 1) ? w : 0.f);
    if (owner && hi == 0) {
      if (ln == 31) {
        // block product and block-local sums -> the group's (idle) hidden region; combined by `combine` after the barrier
        float* o = part_of(blk);
        o[0] = P; o[1] = sr; o[2] = sg; o[3] = sb; o[4] = wh;
      }
      if (q.ok && q.t_ok && a.alpha != nullptr) a.alpha[(int64_t)q.t * a.R + q.ray] = alpha;
      // MODEL 0 composites at the end of a pass and combines in the next pass's exposed phase: the block-local weight waits in
      // the caller's weights array (scaled in place by `combine`) instead of in a register across the hash gathers
      if (M0 && q.ok && q.t_ok && a.weights != nullptr) a.weights[(int64_t)q.t * a.R + q.ray] = w;
    }
    if constexpr (!M0) w_local = w;
  };
  // Cross-block step of the compositing (the reference's cumprod runs over all T steps: src/nerf.py:22-27): every wave of
  // the group walks the NB blocks of pass `pl` in step order with the running transmittance / colour of the current ray
  // (uniform values, carried from pass to pass), scales its own block's weights by the transmittance in front of it and
  // wave 0 stores a ray's colour + background (src/nerf.py:96-98) after its last block.
  float cT = 1.f, cr0 = 0.f, cr1 = 0.f, cr2 = 0.f, cwh = 0.f;
  auto uni = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
  auto combine = [&](int pl) {
    float mine = 1.f;
    typedef __attribute__((ext_vector_type(4))) float f4;
    f4 pv[NB];
    float pw[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {  // all the group's partials in one batch of LDS reads (uniform addresses)
      pv[b] = *(const f4*)part_of(b);
      pw[b] = part_of(b)[4];
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const Loc L = locate(pl, b);
      if (!L.ok) continue;
      const float P = pv[b][0], s0 = pv[b][1], s1 = pv[b][2], s2 = pv[b][3], swh = pw[b];
      if (L.tb == 0) { cT = 1.f; cr0 = cr1 = cr2 = cwh = 0.f; }
      if (b == blk) mine = cT;
      cr0 = cr0 + cT * s0; cr1 = cr1 + cT * s1; cr2 = cr2 + cT * s2; cwh = cwh + cT * swh;
      cT = cT * P;
      if (L.tb == a.nb - 1 && rg == 0 && lane == 0) {
        const float sky = a.bg_kind == NA_BG_WHITE ? 1.0f - cwh : 0.f;
        float* o = a.out + (int64_t)L.ray * 3;
        o[0] = cr0 + sky; o[1] = cr1 + sky; o[2] = cr2 + sky;
      }
    }
    cT = uni(cT); cr0 = uni(cr0); cr1 = uni(cr1); cr2 = uni(cr2); cwh = uni(cwh);
    if (a.weights != nullptr && owner && hi == 0) {
      const Loc L = locate(pl, blk);
      const int t = L.tb * 32 + ln;
      if (L.ok && t < a.T) {
        float* wp = a.weights + (int64_t)t * a.R + L.ray;
        *wp = (M0 ? *wp : w_local) * mine;
      }
    }
  };

#if NA_LS_TRACE
  unsigned long long* tlog = (a.trace != nullptr && blockIdx.x == 0 && lane == 0 && rg == 0) ? a.trace + g * 128 : nullptr;
  int tpos = 0;
  bool ton = false;
#define SYNC()                                                                              \
  do {                                                                                      \
    if (tlog != nullptr && ton && tpos < 126) tlog[tpos++] = __builtin_amdgcn_s_memtime();  \
    __syncthreads();                                                                        \
    if (tlog != nullptr && ton && tpos < 126) tlog[tpos++] = __builtin_amdgcn_s_memtime();  \
  } while (0)
#define STAMP(i)                                                                              \
  do {                                                                                         \
    if (tlog != nullptr && ton) a.trace[256 + g * 16 + (i)] = __builtin_amdgcn_s_memtime();   \
  } while (0)
#else
#define SYNC() __syncthreads()
#define STAMP(i) \
  do {           \
  } while (0)
#endif
  f32x16 acc[2][NB];
  // One hash level (4*hi + k) of a sample -> bytes 8*(k&1)..+7 of this lane's 16 B of init chunk k>>1 (LDS).
  auto hash_finish = [&](int k, const HashGather& hg) {
    float f[4];
    hash_level_finish(hg, f);
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
    bf16x4 h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      h[e] = to_elem<PREC>(f[e]);
      l[e] = to_elem<PREC, false>(f[e] - from_elem<PREC>(h[e]));  // (two-plane precisions only)
    }
    char* dst = ib + (blk * 4 + (k >> 1)) * FR + lane * 16 + (k & 1) * 8;
    *(bf16x4*)dst = h;
    if constexpr (kTwoPlane<PREC>) *(bf16x4*)(dst + 1024) = l;
  };
  Frag<PREC> ring[kPF][2];
  x::Regs XR;  // (NA_PREC_F16X only)
  f32x16 bvx[2];
  // scalar bases of this row group's pair and record streams (F16X)
  const int xpair = kHeaderBytes + kBiasBytes + rg * x::stream_rg(MODEL);
  const int xrec = xpair + x::npair(MODEL) * x::PAIRB;
  constexpr int XNR = x::nrec(MODEL);
  if constexpr (PREC == NA_PREC_F16X) {
#pragma unroll
    for (int c = 0; c < 4; ++c) { x::a16set(XR.a16[0], c, x::wload16(wrs, lane, xrec, 0, c)); x::a16set(XR.a16[1], c, x::wload16(wrs, lane, xrec, 1, c)); }
    XR.a6 = x::wload6(wrs, lane, xrec);
    XR.asc[0] = x::wloadsc(wrs, lane, xrec);
    XR.asc[1] = 0;
  } else {
#pragma unroll
    for (int p = 0; p < kPF; ++p) {
      ring[p][0] = wload<PREC>(wrs, wvoff, p * C::PAIR);
      ring[p][1] = wload<PREC>(wrs, wvoff, p * C::PAIR + FR);
    }
  }

  f32x16 oc[1];
  float density = 0.f;
  int prev = -1;
  // Group 1 runs ONE phase behind group 0.  (A larger odd lag would put the ~10 k-cycle EP of either group opposite a full
  // hidden-layer MFMA phase of the other instead of its 1-k / 2.4-k-cycle view.out / first.init; measured with
  // -DNA_LS_LAG_OVERRIDE = 3 ... 11: the frame time is the same to 0.2 %.)
  constexpr int LAG = NA_LS_LAG_OVERRIDE > 0 ? NA_LS_LAG_OVERRIDE : 1;
  if (g == 1) {
#pragma unroll 1
    for (int i = 0; i < LAG; ++i) __syncthreads();
  }

  if constexpr (MODEL == 1 || MODEL == 3) {
    // the zero chunk behind (x, y, z): its weights are zero, its LDS words only have to be finite
    if (owner) {
      float z8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) z8[e] = 0.f;
      fwrite<PREC>(ib + (blk * 4 + 1) * FR + lane * 16, make_frag<PREC>(z8));
    }
  }

  // (NA_PREC_F16X) the bias of the NEXT phase waits in bvx and becomes the C operand of that phase's first MFMAs
  auto xbias = [&](int ph) {
#pragma unroll
    for (int t = 0; t < 2; ++t) bvx[t] = bias_tile(wrs, bias_rg + ph * 1024, t, lane);
  };
  // (MODEL 0) the end of a pass, inside its view.out phase: compositing of the wave's own block of THIS pass (its ts are
  // re-read: two loads that return under the MFMAs' tail), then everything the next pass's exposed phase would wait for --
  // its ts pair and the scalar loads of its ray
  TsPair tnext = {0.f, 0.f};
  auto pass_tail = [&](int pl) {
    if (NB == 4 || owner) {
      const TsPair tc = ts_load(pl);
      tnext = ts_load(pl + 1);
      prev_dn = own_dn;
      if constexpr (PREC == NA_PREC_F16X) density = *(const float*)(ib + blk * x::KQ + 6144 + 1024 + ln * 16 + 12);
      composite(prev_geom(pl, tc), oc[0], density);
      __builtin_amdgcn_sched_barrier(0);
      own_setup(pl + 1);
    } else {
      tnext = ts_load(pl + 1);
      own_setup(pl + 1);
    }
  };
  // ---- MODEL 6 (mip, NA_PREC_F16X): the 96 IPE features of a sample (src/utils.py:23-27, 83-140; hook src/nerf.py:256-261) as two
  // K64 groups in the hidden format.  Row group rg generates group g = rg & 1 of block rg >> 1: a lane (sample, k half h) computes
  // its 12 (degree, axis) pairs pidx = 24 g + 12 h + j -- sine and cosine feature of a pair from ONE reduced angle and one damping
  // factor, into slots 2 j and 2 j + 1 (three live chunks per group: the four generating waves of a sample group do equal work).
  // Block b's groups live at K64 groups 2 b, 2 b + 1 of block 0's hidden space (block 1's holds the raw [hash | x] / latent values that wait for the skip layer).
  float rad_u[NB];  // pixel radius of block b's ray (uniform)
  auto mip_setup = [&]() {
    if constexpr (MIP) {
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int r = geo_ray[b];
        const int W = a.mip_W, H = a.mip_H;
        const int row = r / W, wq = r - row * W, bi = row / H, hq = row - bi * H;
        rad_u[b] = uni(mip_radius(a.rays, H, W, bi, hq, wq));
      }
    }
  };
  auto gen_ipe = [&](auto act_tag) {
    if constexpr (MIP && PREC == NA_PREC_F16X && !(NA_LS_MIP_ABLATE & 1)) {
      constexpr int ACT = decltype(act_tag)::value;
      const int g = rg & 1, b = rg >> 1;  // this wave's unit (NB = 2)
      float ry[6], rad = 0.f;
      int t0i = 0;
#pragma unroll
      for (int bb = 0; bb < NB; ++bb)
        if (bb == b) {
#pragma unroll
          for (int e = 0; e < 6; ++e) ry[e] = geo_u[bb][e];
          rad = rad_u[bb];
          t0i = geo_t0[bb];
        }
      const int t = t0i + ln;
      const int tc = t < a.T ? t : a.T - 1;
      const float t0 = a.ts[tc];
      const float t1 = tc < a.T - 1 ? a.ts[tc + 1] : mip_last_edge(a.ts, a.T, a.mip_t_end);
      const MipGauss gs = mip_gaussian(ry, rad, t0, t1, a.mip_kind);
      // mip_feature's arithmetic (common.h) with the powers of two pulled out of the products -- bit-identical: scaling by
      // 2^deg commutes with every rounding here.  Per axis: the revolution count of the mean at degree 0 as a (rounded product,
      // recovered error) pair and the damping exponent; per (degree, axis) pair four v_ldexp, one reduction, two v_sin, one v_exp.
      float mm[3] = {gs.m0, gs.m1, gs.m2}, pr0[3], er0[3], ck[3];
      const float cc[3] = {gs.c0, gs.c1, gs.c2};
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) {
        pr0[ax] = mm[ax] * 0.15915494309189535f;
        er0[ax] = fmaf(mm[ax], 0.15915494309189535f, -pr0[ax]) + mm[ax] * 6.4206383e-9f;
        ck[ax] = cc[ax] * -0.7213475204444817f;
      }
      int min_deg = a.mip_min_deg;
      asm volatile("" : "+s"(min_deg));  // (not loop-invariant for the optimiser: 90 hoisted per-pair constants were spilled)
      f32x16 n0, n1;
#pragma unroll
      for (int e = 0; e < 16; ++e) { n0[e] = 0.f; n1[e] = 0.f; }
      auto pairs = [&](auto g_tag) {
        constexpr int G = decltype(g_tag)::value;
#pragma unroll
        for (int j = 0; j < 12; ++j) {
          // pair index of lane half 0 | 1 (compile-time: no per-lane division, one select per operand)
          const int pl = 24 * G + j, ph = 24 * G + 12 + j;
          const int kl = pl / 3, al = pl - 3 * kl, kh = ph / 3, ah = ph - 3 * kh;
          const float m = hi ? mm[ah] : mm[al], p0 = hi ? pr0[ah] : pr0[al], e0 = hi ? er0[ah] : er0[al];
          const float cK = hi ? ck[ah] : ck[al];
          const int deg = min_deg + (hi ? kh : kl);
          const float pr = ldexpf(p0, deg);
          const float rev = (pr - rintf(pr)) + ldexpf(e0, deg);
          const float y = ldexpf(m, deg);
          const float yc = y + 1.5707963267948966f;  // the cosine half is sin(fl(y + pi/2)): the rounded sum differs from y by an
          const float delta = yc - y;                // exactly representable delta
          const float damp = __builtin_amdgcn_exp2f(ldexpf(cK, 2 * deg));
          const float sn = damp * __builtin_amdgcn_sinf(rev);
          const float cs = damp * __builtin_amdgcn_sinf(fmaf(delta, 0.15915494309189535f, rev));
          if (j < 8) { n0[2 * j] = sn; n0[2 * j + 1] = cs; } else { n1[2 * (j - 8)] = sn; n1[2 * (j - 8) + 1] = cs; }
        }
      };
      if (g == 0) pairs(std::integral_constant<int, 0>{}); else pairs(std::integral_constant<int, 1>{});
      x::store_block<ACT>(hb + (2 * b + g) * x::KQ, n0, n1, lane, a.sat_gen);
    }
  };
  // ---- NA_PREC_F16X, schedules whose first MLP takes the hash encoder (MODEL 0, 4): the [hash | x] group
  auto hash_group_ep = [&](int pass) {
    if constexpr (PREC == NA_PREC_F16X) {
      // f16x (round 4): [hash | x] is ONE K64 group of the init region in the hidden format (f16 fragments | R | T), so the
      // lane that converts must hold all 32 values of an MFMA lane (sample, k half h): h = 0 (levels 0..3 + x, y, z twice)
      // comes from the block's owner wave, h = 1 (levels 4..7) from the helper wave rg + 2.  All 64 lanes gather -- lane
      // (sample, j) the levels 4 h + 2 j, 4 h + 2 j + 1 -- then the j = 1 half hands its eight features to the j = 0 half
      // (ds_bpermute: no memory), which converts and stores for MFMA lane (sample, h).  The raw values wait in the idle hidden
      // region for the skip connection (E1 re-enters them through the activation).
      const int part = owner ? 0 : 1;
      const Geom q = geom(pass, blk, tnext);
      float f8[8];
      HashGather hg;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int lvl0 = 4 * part + k;                      // (j = 0) | + 2 (j = 1)
        hash_level_issue(q.px, q.py, q.pz, a.tables, hi ? a.res.n[lvl0 + 2] : a.res.n[lvl0], lvl0 + 2 * hi, hg);
        float f[4];
        hash_level_finish(hg, f);
#pragma unroll
        for (int e = 0; e < 4; ++e) f8[4 * k + e] = f[e];
        __builtin_amdgcn_sched_barrier(0);
        STAMP(3 + k);
      }
      f32x16 n0, n1;  // the MFMA lane's 32 values in slot order: chunk 0 = n0[0..7], 1 = n0[8..15], 2 = n1[0..7], 3 = n1[8..15]
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        n0[e] = f8[e];
        n0[8 + e] = __shfl_down(f8[e], 32, 64);
        n1[e] = 0.f; n1[8 + e] = 0.f;
      }
      if (part == 0) { n1[0] = q.px; n1[1] = q.py; n1[2] = q.pz; n1[3] = q.px; n1[4] = q.py; n1[5] = q.pz; }
      if (hi == 0) {
        const int ml = ln + 32 * part;  // the MFMA lane these values belong to
        char* st = hb + x::BLKH + rg * x::KQ + ml * 16;   // raw values: this wave's K64 region of block 1 (idle until E1 stores it LAST)
#pragma unroll
        for (int c = 0; c < 4; ++c) *(f32x4*)(st + c * 1024) = f32x4{n0[4 * c], n0[4 * c + 1], n0[4 * c + 2], n0[4 * c + 3]};
        *(f32x4*)(st + 4096) = f32x4{n1[0], n1[1], n1[2], n1[3]};
        *(f32x4*)(st + 5120) = f32x4{n1[4], n1[5], n1[6], n1[7]};
        x::store_block<NA_ACT_NONE, 3>(ib + blk * x::KQ, n0, n1, ml, a.sat_gen);
      }
      STAMP(7);
    }
  };
      // raw values of this wave's half of an init group (written by EP / E6 into the wave's own K64 region of the hidden
      // space): read back before store_acts overwrites the region, re-entered through the activation (src/neural_blocks.py:291-293)
  auto reenter_hash = [&]() {
    if constexpr (PREC == NA_PREC_F16X) {
        if (hi == 0) {
          const int part = owner ? 0 : 1, ml = ln + 32 * part;
          const char* st = hb + x::BLKH + rg * x::KQ + ml * 16;
          f32x16 n0, n1;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const f32x4 v = *(const f32x4*)(st + c * 1024);
            n0[4 * c] = v[0]; n0[4 * c + 1] = v[1]; n0[4 * c + 2] = v[2]; n0[4 * c + 3] = v[3];
          }
          const f32x4 u = *(const f32x4*)(st + 4096), w = *(const f32x4*)(st + 5120);
#pragma unroll
          for (int e = 0; e < 16; ++e) n1[e] = 0.f;
          n1[0] = u[0]; n1[1] = u[1]; n1[2] = u[2]; n1[3] = u[3]; n1[4] = w[0]; n1[5] = w[1]; n1[6] = w[2]; n1[7] = w[3];
          x::store_block<NA_ACT_LEAKY_RELU, 3>(ib + blk * x::KQ, n0, n1, ml, a.sat_gen);
        }
    }
  };
  // ---- MODEL 7 / 8 (round 6): the reflectance head's OWN hash encoder (src/refl.py:233-237, 253-257: `enc=HashEncoder()`, a second set
  // of tables) at the same positions.  Gathered in the epilogue of first.out by the four waves exactly like hash_group_ep -- the init
  // region is free again (first.L0 consumed [hash | x]) and the raw values wait in the wave's K64 region of block 1 for the
  // re-entry through the activation -- with two differences: the T plane's spare dword is left alone (KEEP7: the block's density
  // waits there, as in MODEL 0's latent group) and MODEL 8 puts up to three refl_latent columns (--dyn-refl-latent, rows from
  // HBM) into the spare slots 6.. of chunk 2.
  // (both gather rounds are ISSUED first -- hash2_issue -- and combined after the epilogue's other work -- hash2_finish: the gathers of
  // this phase queue behind the partner group's weight stream, which keeps the vector memory path ~94 % busy during its MFMA phases)
  struct Hash2 { HashGather g[2]; float px, py, pz, r0, r1, r2; };
  auto hash2_issue = [&](int pass, Hash2& H) {
    if constexpr (PREC == NA_PREC_F16X && HEAD2) {
      const int part = owner ? 0 : 1;
      const TsPair tp = ts_load(pass);
      const Geom q = geom(pass, blk, tp);
      H.px = q.px; H.py = q.py; H.pz = q.pz;
      H.r0 = H.r1 = H.r2 = 0.f;
      if constexpr (MODEL == 8) {
        if (a.rl != nullptr) {
          const float* rr = a.rl + ((int64_t)(q.t_ok ? q.t : a.T - 1) * a.R + q.ray) * a.rl_ld;
          if (a.n_rl > 0) H.r0 = rr[0];
          if (a.n_rl > 1) H.r1 = rr[1];
          if (a.n_rl > 2) H.r2 = rr[2];
        }
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int lvl0 = 4 * part + k;                      // (j = 0) | + 2 (j = 1)
        hash_level_issue(q.px, q.py, q.pz, a.tables2, hi ? a.res.n[lvl0 + 2] : a.res.n[lvl0], lvl0 + 2 * hi, H.g[k]);
      }
    }
  };
  auto hash2_finish = [&](const Hash2& H) {
    if constexpr (PREC == NA_PREC_F16X && HEAD2) {
      const int part = owner ? 0 : 1;
      float f8[8];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        float f[4];
        hash_level_finish(H.g[k], f);
#pragma unroll
        for (int e = 0; e < 4; ++e) f8[4 * k + e] = f[e];
      }
      f32x16 n0, n1;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        n0[e] = f8[e];
        n0[8 + e] = __shfl_down(f8[e], 32, 64);
        n1[e] = 0.f; n1[8 + e] = 0.f;
      }
      if (part == 0) { n1[0] = H.px; n1[1] = H.py; n1[2] = H.pz; n1[3] = H.px; n1[4] = H.py; n1[5] = H.pz; n1[6] = H.r0; n1[7] = H.r1; }
      else n1[0] = H.r2;
      if (hi == 0) {
        const int ml = ln + 32 * part;
        char* st = hb + x::BLKH + rg * x::KQ + ml * 16;
#pragma unroll
        for (int c = 0; c < 4; ++c) *(f32x4*)(st + c * 1024) = f32x4{n0[4 * c], n0[4 * c + 1], n0[4 * c + 2], n0[4 * c + 3]};
        *(f32x4*)(st + 4096) = f32x4{n1[0], n1[1], n1[2], n1[3]};
        *(f32x4*)(st + 5120) = f32x4{n1[4], n1[5], n1[6], n1[7]};
        x::store_block<NA_ACT_NONE, 3, true>(ib + blk * x::KQ, n0, n1, ml, a.sat_gen);
      }
    }
  };
  auto reenter_hash2 = [&]() {  // reenter_hash with the spare dword kept
    if constexpr (PREC == NA_PREC_F16X && HEAD2) {
      if (hi == 0) {
        const int part = owner ? 0 : 1, ml = ln + 32 * part;
        const char* st = hb + x::BLKH + rg * x::KQ + ml * 16;
        f32x16 n0, n1;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const f32x4 v = *(const f32x4*)(st + c * 1024);
          n0[4 * c] = v[0]; n0[4 * c + 1] = v[1]; n0[4 * c + 2] = v[2]; n0[4 * c + 3] = v[3];
        }
        const f32x4 u = *(const f32x4*)(st + 4096), w = *(const f32x4*)(st + 5120);
#pragma unroll
        for (int e = 0; e < 16; ++e) n1[e] = 0.f;
        n1[0] = u[0]; n1[1] = u[1]; n1[2] = u[2]; n1[3] = u[3]; n1[4] = w[0]; n1[5] = w[1]; n1[6] = w[2]; n1[7] = w[3];
        x::store_block<NA_ACT_LEAKY_RELU, 3, true>(ib + blk * x::KQ, n0, n1, ml, a.sat_gen);
      }
    }
  };
  // raw rows of a K64 group (the lane's 32 accumulator values) parked in global memory: slot `slot` of block b of this sample group.
  // Written and read by waves of ONE workgroup with a workgroup barrier in between (L2-resident: 32 KiB per workgroup).
  // (round 6: through a buffer resource -- a uniform base, the slot's byte offset in an SGPR, 16 bytes x lane in the vector offset.
  // As plain global accesses every (block, slot) combination was a per-lane 64-bit pointer that the compiler hoisted out of the
  // pass loop and spilled: 10 of MODEL 8's reloads per pass)
  const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc((void*)a.park, 0, a.park != nullptr ? (int)kParkBytes : 0, 0x00020000);
  auto park_soff = [&](int b, int slot) -> uint32_t {
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)(((((uint32_t)blockIdx.x * 2 + g) * 2 + b) * kParkSlots + slot) * 8192u));
  };
  auto park_store = [&](int b, int slot, const f32x16& v0, const f32x16& v1) {
    // (the slot offset in the VECTOR offset, soffset 0: a 16-byte store with an SGPR soffset is the store-data hazard of DESIGN 3d)
    const uint32_t vo = (uint32_t)lane * 16 + park_soff(b, slot);
    typedef uint32_t u32x4p __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4p, f32x4{v0[4 * j], v0[4 * j + 1], v0[4 * j + 2], v0[4 * j + 3]}), prs, vo + j * 1024, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4p, f32x4{v1[4 * j], v1[4 * j + 1], v1[4 * j + 2], v1[4 * j + 3]}), prs, vo + (4 + j) * 1024, 0, 0);
    }
  };
  auto park_load = [&](int b, int slot, f32x16& v0, f32x16& v1) {
    const uint32_t vo = (uint32_t)lane * 16 + park_soff(b, slot);
    typedef uint32_t u32x4p __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 u = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prs, vo + j * 1024, 0, 0));
      const f32x4 w = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prs, vo + (4 + j) * 1024, 0, 0));
      v0[4 * j] = u[0]; v0[4 * j + 1] = u[1]; v0[4 * j + 2] = u[2]; v0[4 * j + 3] = u[3];
      v1[4 * j] = w[0]; v1[4 * j + 1] = w[1]; v1[4 * j + 2] = w[2]; v1[4 * j + 3] = w[3];
    }
  };
  // ---- MODEL 9 (round 6): the training step's forward of PlainNeRF(view) as ONE launch (src/neural_blocks.py:279-296 twice, between
  // them src/nerf.py:338-357).  The layer-by-layer training forward (csrc/train_fwd.hip) writes every Linear's output rows [N, 256]
  // and READS them back as the next Linear's input; here the rows stay in LDS from layer to layer as in inference, and the fp32
  // accumulators -- exactly what the backward kernel of the NEXT Linear wants as its input (it applies act and act' itself:
  // csrc/train_bwd.hip) -- are written once on the way: plane p of Args::y = output rows of Linear p (0..4 first.init, L0..L3;
  // 5..9 view.init, L0..L3), sample row t * R + ray like every [T, R, .] tensor of the path; first.out's 65 rows (train_store_first) and
  // view.out's 3 rows go to their own buffers.  A lane holds rows 8 j + 4 hi .. + 3 of a tile for
  // its sample in registers 4 j .. 4 j + 3: one 16-byte store per (tile, block, j), 128 contiguous bytes per sample and tile.
  // (buffer stores, the row offset in the VECTOR offset and soffset 0: DESIGN 3d's store-data hazard)
  auto train_row = [&](int pl, int b, bool& ok) -> uint32_t {
    const Loc L = locate(pl, b);
    const int t = L.tb * 32 + ln;
    ok = L.ok && t < a.T;
    return (uint32_t)((int64_t)t * a.R + L.ray);
  };
  auto train_store = [&](const f32x16 (&av)[2][NB], int pl, int plane) {
    if constexpr (TRAIN) {
      static_assert(!TRAIN || FR == 2048, "MODEL 9: the two-plane formats (a tile's two output fragments = 4 KiB of LDS)");
      const int64_t N = (int64_t)a.T * a.R;  // (< 2^22: checked by the host)
      const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.y + (int64_t)plane * N * 256), 0, (int)(N * 1024), 0x00020000);
      typedef uint32_t u32x4p __attribute__((ext_vector_type(4)));
      // Straight from the accumulators a store instruction would write 32-byte pieces of 32 different rows (measured: the write path
      // then takes 4 cycles per piece and CU -- 0.63 ms per 262 144 samples on top of 0.86); every tile goes through LDS instead and
      // leaves as WHOLE 128-byte lines, 8 lanes per row.  The staging area is the 4 KiB this wave is about to overwrite with the
      // tile's two activation fragments anyway (dead since the barrier that closed the MFMA phase; nobody else writes it): sample-major
      // [32][128 B], 16-byte piece q of sample s at slot q ^ (s & 7) -- conflict-free for the 8-lane groups of ds_write_b128 and the
      // 16-lane groups of ds_read_b128 (MI355X_MICROARCH "LDS").  LDS operations of one wave execute in order: no wait in between.
      const int s0 = lane >> 3, pc = lane & 7;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const Loc L = locate(pl, b);
        if (!L.ok || (NA_LS_TRAIN_EXP & 1)) continue;
        const int t0 = L.tb * 32 + s0;
        const uint32_t vo = (uint32_t)((int64_t)t0 * a.R + L.ray) * 1024u + (uint32_t)(rg * 256 + pc * 16);
        const uint32_t step = (uint32_t)a.R * 8192u;  // 8 samples on
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          char* stg = hb + (b * 16 + 4 * rg + 2 * t) * FR;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            *(f32x4*)(stg + ln * 128 + (((2 * j + hi) ^ (ln & 7)) * 16)) = f32x4{av[t][b][4 * j], av[t][b][4 * j + 1], av[t][b][4 * j + 2], av[t][b][4 * j + 3]};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const f32x4 v = *(const f32x4*)(stg + (8 * k + s0) * 128 + ((pc ^ (s0 & 7)) * 16));
#if NA_LS_TRAIN_EXP & 2   // (timing: the LDS staging alone -- the values are consumed, nothing is stored)
            asm volatile("" :: "v"(v));
#elif NA_LS_TRAIN_EXP & 4  // (timing: the same store instructions into a 64-KiB window per plane: no HBM stream behind them)
            if (t0 + 8 * k < a.T) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4p, v), trs, (vo + k * step + t * 128) & 0xFFFFu, 0, NA_LS_TRAIN_AUX);
#else
            if (t0 + 8 * k < a.T) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4p, v), trs, vo + k * step + t * 128, 0, NA_LS_TRAIN_AUX);
#endif
          }
        }
      }
    }
  };
  // first.out: row group rg < 2 holds the intermediate rows 32 rg .. + 31, row group 2 the density (register 0 of the hi = 0 lanes) of
  // the NB blocks.  They leave as what the step's NEXT consumers read (src/nerf.py:338-357, src/refl.py:190-207): the View MLP's init
  // rows [x, y, z, elev, azim | intermediate] -> Args::park as [N, 69] (the geometry columns by the density lanes: the explicit
  // position of the sample and the ray's angles from the pre-kernel, geo_setup(pl) has run) and the density -> Args::rl as [N];
  // first_out [N, 65] itself is never materialised (na_plain_head_rows built the same rows from it: one more pass over 68 MB)
  auto train_store_first = [&](const f32x16 (&oq)[NB], int pl) {
    if constexpr (TRAIN) {
      const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc((void*)a.park, 0, (int)((int64_t)a.T * a.R * 276), 0x00020000);
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        bool ok;
        const uint32_t row = train_row(pl, b, ok);
        if (ok && rg < 2) {
          const uint32_t vo = row * 276u + (uint32_t)(4 * (5 + 32 * rg + 4 * hi));
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const float v = oq[b][i];  // (__builtin_bit_cast of a vector ELEMENT expression reads element 0: through a scalar)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), frs, vo + 4 * (8 * (i >> 2) + (i & 3)), 0, 0);
          }
        } else if (ok && rg == 2 && hi == 0) {
          const_cast<float*>(a.rl)[row] = oq[b][0];
          const float* p = a.pts + (int64_t)row * 3;
          const float gx = p[0], gy = p[1], gz = p[2], el = geo_u[b][6], az = geo_u[b][7];
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, gx), frs, row * 276u, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, gy), frs, row * 276u + 4, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, gz), frs, row * 276u + 8, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, el), frs, row * 276u + 12, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, az), frs, row * 276u + 16, 0, 0);
        }
      }
    }
  };
  // view.out: rows 0..2 (registers 0..2 of the hi = 0 lanes) of the owner's block -> Args::feat as [N, 3] (before the sigmoid)
  auto train_store_rgb = [&](const f32x16& o, int pl) {
    if constexpr (TRAIN) {
      if (owner && hi == 0) {
        bool ok;
        const uint32_t row = train_row(pl, blk, ok);
        if (ok) {
          float* dst = const_cast<float*>(a.feat) + (int64_t)row * 3;
          dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2];
        }
      }
    }
  };
  if constexpr (M0 || MODEL == 4) {
    tnext = ts_load(0);
    own_setup(0);
  }
  for (int pass = 0; pass < a.npg; ++pass) {
    int cur = 0;
#if NA_LS_TRACE
    ton = pass == 1;
#endif
#include "ls_sched_fourier_mlp.inc"
#include "ls_sched_hash_mlp.inc"
#include "ls_sched_volsdf_siren.inc"
#include "ls_sched_view.inc"
#include "ls_sched_tiny.inc"
#include "ls_sched_plain_pos.inc"
#include "ls_sched_plain_plv.inc"
#include "ls_sched_plain.inc"
    prev = pass;
  }
  if (!M0 && MODEL < 4 && prev >= 0 && (NB == 4 || owner)) {  // (MODEL 0 composited its last pass in that pass's view.out phase)
    prev_dn = own_dn;
    if constexpr (MODEL == 1) {
      f32x16 rgbv = oc[0];
      rgbv[0] = oc[0][1]; rgbv[1] = oc[0][2]; rgbv[2] = oc[0][3];
      composite(prev_geom(prev, ts_load(prev)), rgbv, oc[0][0]);
    } else {
      composite(prev_geom(prev, ts_load(prev)), oc[0], density);
    }
  }
  __syncthreads();
  if ((MODEL < 4 || M0) && prev >= 0) combine(prev);
  if (g == 0) {  // group 0 takes its extra barriers at the end
#pragma unroll 1
    for (int i = 0; i < LAG; ++i) __syncthreads();
  }
}

}  // namespace ls
}  // namespace na
