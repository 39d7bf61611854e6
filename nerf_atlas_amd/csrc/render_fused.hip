// na_render_plain_view: PlainNeRF.forward with the View head (src/nerf.py:326-361, src/refl.py:190-207) as ONE
// kernel: stratified samples -> hash encode -> `first` MLP -> elev/azim -> View MLP -> sigmoid -> per-ray
// alpha compositing.  Nothing per-sample ever touches HBM: in = 24 B/ray, out = 12 B/ray (+ 32 B per
// 32-sample block of compositing partials).
//
// Work item = (ray, block of 32 consecutive steps).  One wave owns one item: lane&31 = step inside the block,
// so compositing is a 32-lane exclusive product scan in registers (wavefront shuffles).  Blocks of one ray are
// combined by a tiny second kernel (na_render_finalize), which also adds the background term.
//
// Compiled once per precision (-DNA_PREC_INST=0|1).
#include <atomic>
#include "mlp_layout.h"
#include "encoders.h"

#ifndef NA_PREC_INST
#error "compile with -DNA_PREC_INST=0 (bf16) or 1 (bf16x3)"
#endif

namespace na {

constexpr int kPartialFloats = 8;  // P, S_r, S_g, S_b, W_head, pad

struct RenderArgs {
  const float* rays;     // [R,6]
  const float* ts;       // [T]
  const float* pts;      // nullable [T,R,3]: explicit sample positions (deformed canonical points) instead of o + t d
  const float4* tables;  // [8,65536]
  const char* packed_first;
  const char* packed_view;
  float* alpha;    // nullable [T,R]
  float* weights;  // nullable [T,R] (block-local here; finalize applies the cross-block prefix)
  float* partials; // [R*nb, 8]
  int64_t R;
  int64_t nitems;  // R * nb
  int T, nb;
  int ngroups;
  int sigmoid_kind;
  int first_layers, view_layers;
  uint32_t buf_bytes;
  HashRes res;
};

template <int PREC, int NWAVES, int NB>
__global__ __launch_bounds__(NWAVES * 64) void render_plain_view_kernel(RenderArgs a, TileTab tab) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NI1 = 3, NI2 = 5;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hi = lane >> 5, ln = lane & 31;
  WeightStream<NWAVES, 3> ws;
  const int npasses = ((int)blockIdx.x < a.ngroups) ? (a.ngroups - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
  ws.start(tab, a.packed_first + kHeaderBytes, a.packed_view + kHeaderBytes, smem, a.buf_bytes, npasses, wave, lane);

  // Geometry of this wave's work item b of group g: recomputed where needed (prologue, View input, compositing)
  // instead of being carried in registers across both MLPs -- the kernel sits at the 256-VGPR limit and anything
  // long-lived would be spilled to scratch.
  struct Geom {
    int64_t item, ray;
    bool item_ok, t_ok;
    int t;
    float px, py, pz, dist, dx, dy, dz;
  };
  auto geom = [&](int g, int b) {
    Geom q;
    const int64_t item_raw = ((int64_t)g * NWAVES + wave) * NB + b;
    q.item_ok = item_raw < a.nitems;
    q.item = q.item_ok ? item_raw : a.nitems - 1;
    q.ray = q.item / a.nb;
    const int tb = (int)(q.item - q.ray * a.nb);
    q.t = tb * 32 + ln;
    q.t_ok = q.t < a.T;
    const int tc = q.t_ok ? q.t : a.T - 1;
    // sample position (src/nerf.py:53) and interval length (src/nerf.py:67-70)
    const float* ry = a.rays + q.ray * 6;
    q.dx = ry[3]; q.dy = ry[4]; q.dz = ry[5];
    const float tt = a.ts[tc];
    if (a.pts != nullptr) {
      const float* p = a.pts + ((int64_t)tc * a.R + q.ray) * 3;
      q.px = p[0]; q.py = p[1]; q.pz = p[2];
    } else {
      q.px = ry[0] + tt * q.dx; q.py = ry[1] + tt * q.dy; q.pz = ry[2] + tt * q.dz;
    }
    const float d = tc < a.T - 1 ? fmaxf(a.ts[tc + 1] - tt, 1e-5f) : 1e10f;
    q.dist = d * sqrtf((q.dx * q.dx + q.dy * q.dy) + q.dz * q.dz);
    return q;
  };

  for (int g = blockIdx.x; g < a.ngroups; g += gridDim.x) {
    ws.mark(10);
    Frag<PREC> I1[NB * NI1];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const Geom q = geom(g, b);
      // ---- `first` MLP input: [hash levels 4hi..4hi+3 | p, x]
      float f[16];
      if constexpr ((NA_ABLATE & 32) != 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) f[e] = q.px * (float)e;
      } else {
        hash_levels4(q.px, q.py, q.pz, a.tables, a.res, 4 * hi, f);
      }
      float v0[8], v1[8], v2[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { v0[e] = f[e]; v1[e] = f[8 + e]; v2[e] = 0.f; }
      if (hi == 0) { v2[0] = q.px; v2[1] = q.py; v2[2] = q.pz; v2[3] = q.px; v2[4] = q.py; v2[5] = q.pz; }
      I1[b * NI1 + 0] = make_frag<PREC>(v0);
      I1[b * NI1 + 1] = make_frag<PREC>(v1);
      I1[b * NI1 + 2] = make_frag<PREC>(v2);
    }
    Frag<PREC> H[NB * kHC];
    ws.mark(11);
    mlp_hidden_layers<PREC, NA_ACT_LEAKY_RELU, NB, NI1>(ws, a.first_layers, 3, I1, H, lane);

    // ---- `first` out: rows 0..63 = intermediate (-> View latent), row 64 = density
    Frag<PREC> I2[NB * NI2];
    float density[NB];
    {
      f32x16 o[NB];
      mlp_out_tile<PREC, NB>(ws, H, lane, o);
#pragma unroll
      for (int b = 0; b < NB; ++b) acc_to_frags<PREC, NA_ACT_NONE>(o[b], I2[b * NI2 + 0], I2[b * NI2 + 1]);
      mlp_out_tile<PREC, NB>(ws, H, lane, o);
#pragma unroll
      for (int b = 0; b < NB; ++b) acc_to_frags<PREC, NA_ACT_NONE>(o[b], I2[b * NI2 + 2], I2[b * NI2 + 3]);
      mlp_out_tile<PREC, NB>(ws, H, lane, o);
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        density[b] = o[b][0];  // row 64 lives in register 0 of the hi=0 lanes
        const Geom q = geom(g, b);
        float el, az;
        elev_azim(q.dx, q.dy, q.dz, el, az);
        float v4[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v4[e] = 0.f;
        if (hi == 0) { v4[0] = q.px; v4[1] = q.py; v4[2] = q.pz; v4[3] = el; v4[4] = az; }
        I2[b * NI2 + 4] = make_frag<PREC>(v4);
      }
    }
    // ---- View MLP (sin activations)
    mlp_hidden_layers<PREC, NA_ACT_SIN, NB, NI2>(ws, a.view_layers, 3, I2, H, lane);
    f32x16 oc[NB];
    mlp_out_tile<PREC, NB>(ws, H, lane, oc);

    ws.mark(12);
    // ---- compositing inside each block (src/nerf.py:22-27,60-80); the hi=0 half holds the samples
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const Geom q = geom(g, b);
      const float cr = apply_sigmoid_kind(oc[b][0], a.sigmoid_kind);
      const float cg = apply_sigmoid_kind(oc[b][1], a.sigmoid_kind);
      const float cb = apply_sigmoid_kind(oc[b][2], a.sigmoid_kind);
      const float sigma = softplusf_(density[b] - 1.0f);
      float alpha = q.t_ok ? 1.0f - expf(-sigma * q.dist) : 0.f;
      float f = (1.0f - alpha) + 1e-10f;
      float incl = f;  // inclusive product scan over the 32 lanes of this half
#pragma unroll
      for (int d = 1; d < 32; d <<= 1) {
        float up = __shfl_up(incl, d, 32);
        if (ln >= d) incl = incl * up;
      }
      float excl = __shfl_up(incl, 1, 32);
      if (ln == 0) excl = 1.0f;
      const float w = alpha * excl;
      float sr = w * cr, sg = w * cg, sb = w * cb;
      float wh = (q.t < a.T - 1) ? w : 0.f;
#pragma unroll
      for (int d = 16; d >= 1; d >>= 1) {
        sr += __shfl_xor(sr, d, 32);
        sg += __shfl_xor(sg, d, 32);
        sb += __shfl_xor(sb, d, 32);
        wh += __shfl_xor(wh, d, 32);
      }
      const float P = __shfl(incl, 31, 32);
      if (q.item_ok && hi == 0) {
        if (ln == 0) {
          float* o = a.partials + q.item * kPartialFloats;
          o[0] = P; o[1] = sr; o[2] = sg; o[3] = sb; o[4] = wh;
        }
        if (q.t_ok) {
          if (a.alpha != nullptr) a.alpha[(int64_t)q.t * a.R + q.ray] = alpha;
          if (a.weights != nullptr) a.weights[(int64_t)q.t * a.R + q.ray] = w;
        }
      }
    }
    ws.mark(13);
  }
#if NA_TRACE
  if (ws.tlog != nullptr) {  // dump behind the partials (the workspace is over-allocated by tools/trace.py)
    unsigned long long* dst = (unsigned long long*)(a.partials + a.nitems * kPartialFloats) + (wave ? 1 + ws.kTraceMax : 0);
    dst[0] = (unsigned long long)ws.tpos;
    for (int i = 0; i < ws.tpos; ++i) dst[1 + i] = ws.tlog[i];
  }
#endif
}

#if NA_PREC_INST == 0
// Combine the per-block partials of each ray in step order, add the background (src/nerf.py:96-98) and
// turn block-local weights into global ones.
__global__ void render_finalize_kernel(const float* __restrict__ partials, int64_t R, int nb, int T, int bg_kind,
                                       float* __restrict__ weights, float* __restrict__ out) {
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < R; r += (int64_t)gridDim.x * blockDim.x) {
    float prefix = 1.0f, c0 = 0.f, c1 = 0.f, c2 = 0.f, wh = 0.f;
    for (int b = 0; b < nb; ++b) {
      const float* p = partials + (r * nb + b) * kPartialFloats;
      c0 = c0 + prefix * p[1];
      c1 = c1 + prefix * p[2];
      c2 = c2 + prefix * p[3];
      wh = wh + prefix * p[4];
      if (weights != nullptr && b > 0) {
        const int t1 = min(T, (b + 1) * 32);
        for (int t = b * 32; t < t1; ++t) weights[(int64_t)t * R + r] *= prefix;
      }
      prefix = prefix * p[0];
    }
    const float sky = bg_kind == NA_BG_WHITE ? 1.0f - wh : 0.f;
    out[r * 3 + 0] = c0 + sky;
    out[r * 3 + 1] = c1 + sky;
    out[r * 3 + 2] = c2 + sky;
  }
}


int launch_render_finalize(const float* partials, int64_t R, int nb, int T, int bg_kind, float* weights, float* out,
                           hipStream_t stream) {
  hipLaunchKernelGGL(render_finalize_kernel, dim3(grid_for(R, 128, 1 << 16)), dim3(128), 0, stream, partials, R, nb, T,
                     bg_kind, weights, out);
  return check_launch("na_render_finalize");
}
#endif

// hipFuncSetAttribute is per DEVICE: remember which devices already carry the 160-KiB dynamic-LDS limit
inline int ensure_max_lds(const void* kern, std::atomic<uint64_t>& done) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { set_error("hipGetDevice failed"); return NA_EHIP; }
  const uint64_t bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return NA_OK;
  hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return NA_EHIP; }
  done.fetch_or(bit, std::memory_order_release);
  return NA_OK;
}

template <int PREC, int NWAVES, int NB>
static int launch_render(RenderArgs& a, const TileTab& tab, hipStream_t stream) {
  auto kern = render_plain_view_kernel<PREC, NWAVES, NB>;
  static std::atomic<uint64_t> attr_done{0};
  if (int rc = ensure_max_lds((const void*)kern, attr_done); rc != NA_OK) return rc;
  a.ngroups = (int)((a.nitems + NWAVES * NB - 1) / (NWAVES * NB));
  int grid = a.ngroups < 256 ? a.ngroups : 256;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NWAVES * 64), 3 * a.buf_bytes + 8 * kMaxTiles + (NA_TRACE ? 2 * 8 * 1400 : 0), stream, a, tab);
  return check_launch("na_render_plain_view");
}

#if NA_PREC_INST == 0
int render_dispatch_bf16(RenderArgs& a, const TileTab& tab, hipStream_t s) { return launch_render<NA_PREC_BF16, 8, 1>(a, tab, s); }
#else
int render_dispatch_bf16x3(RenderArgs& a, const TileTab& tab, hipStream_t s) { return launch_render<NA_PREC_BF16X3, 4, 1>(a, tab, s); }
#endif
int render_dispatch_bf16(RenderArgs& a, const TileTab& tab, hipStream_t s);
int render_dispatch_bf16x3(RenderArgs& a, const TileTab& tab, hipStream_t s);

}  // namespace na

#if NA_PREC_INST == 0
// the C ABI entry points live in the bf16 translation unit
using namespace na;

extern "C" size_t na_render_workspace_bytes(int T, int64_t R) {
  if (T < 1 || R < 0) return 0;
  int64_t nb = (T + 31) / 32;
  return (size_t)(R * nb * kPartialFloats * sizeof(float)) + 256;
}

static int render_plain_view_impl(const float* rays, const float* pts, int64_t R, const float* ts, int T,
                                  const float* hash_tables, const void* packed_first, const void* packed_view,
                                  int precision, int sigmoid_kind, int bg_kind, float* alpha, float* weights, float* out,
                                  void* workspace, size_t workspace_bytes, void* stream) {
  NA_REQUIRE(rays && ts && hash_tables && packed_first && packed_view && out && workspace, NA_ENULL,
             "na_render_plain_view: null pointer");
  NA_REQUIRE(T >= 1 && R >= 0, NA_EINVAL, "na_render_plain_view: bad shape T=%d R=%lld", T, (long long)R);
  NA_REQUIRE(precision == NA_PREC_BF16 || precision == NA_PREC_BF16X3, NA_EUNSUPPORTED,
             "na_render_plain_view: precision %d", precision);
  NA_REQUIRE(sigmoid_kind >= 0 && sigmoid_kind <= NA_SIG_IDENTITY, NA_EUNSUPPORTED, "na_render_plain_view: sigmoid %d",
             sigmoid_kind);
  NA_REQUIRE(bg_kind == NA_BG_BLACK || bg_kind == NA_BG_WHITE, NA_EUNSUPPORTED, "na_render_plain_view: bg %d", bg_kind);
  NA_REQUIRE(workspace_bytes >= na_render_workspace_bytes(T, R), NA_EWORKSPACE,
             "na_render_plain_view: workspace %zu < %zu bytes", workspace_bytes, na_render_workspace_bytes(T, R));
  if (R == 0) return NA_OK;
  // the two MLP shapes this kernel is specialised for (src/nerf.py:320-324, src/refl.py:201-204)
  NaMlpDesc d1 = {3, NA_ENC_HASH, 35, 0, 4, 256, 65, 3, NA_ACT_LEAKY_RELU, NA_LAYOUT_PLAIN_FIRST};
  NaMlpDesc d2 = {5, NA_ENC_NONE, 0, 64, 4, 256, 3, 3, NA_ACT_SIN, NA_LAYOUT_PLAIN_VIEW};
  RenderArgs a;
  a.rays = rays; a.ts = ts; a.pts = pts; a.tables = (const float4*)hash_tables;
  a.packed_first = (const char*)packed_first; a.packed_view = (const char*)packed_view;
  a.alpha = alpha; a.weights = weights;
  a.partials = (float*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  a.R = R; a.T = T; a.nb = (T + 31) / 32; a.nitems = R * a.nb;
  a.sigmoid_kind = sigmoid_kind;
  a.first_layers = d1.num_layers; a.view_layers = d2.num_layers;
  a.buf_bytes = (uint32_t)((kHC + 5) * planes_of(precision) + 1) * 1024;
  a.res = hash_resolutions();
  TileTab tab;
  tab.hdr0 = (const uint32_t*)packed_first;
  tab.hdr1 = (const uint32_t*)packed_view;
  tab.split = tile_count(d1);
  tab.ntiles = tab.split + tile_count(d2);
  int rc = precision == NA_PREC_BF16 ? render_dispatch_bf16(a, tab, (hipStream_t)stream)
                                     : render_dispatch_bf16x3(a, tab, (hipStream_t)stream);
  if (rc != NA_OK) return rc;
  return launch_render_finalize(a.partials, R, a.nb, T, bg_kind, weights, out, (hipStream_t)stream);
}

extern "C" int na_render_plain_view(const float* rays, int64_t R, const float* ts, int T, const float* hash_tables,
                                    const void* packed_first, const void* packed_view, int precision, int sigmoid_kind,
                                    int bg_kind, float* alpha, float* weights, float* out, void* workspace,
                                    size_t workspace_bytes, void* stream) {
  return render_plain_view_impl(rays, nullptr, R, ts, T, hash_tables, packed_first, packed_view, precision, sigmoid_kind,
                                bg_kind, alpha, weights, out, workspace, workspace_bytes, stream);
}

extern "C" int na_render_plain_view_pts(const float* rays, const float* pts, int64_t R, const float* ts, int T,
                                        const float* hash_tables, const void* packed_first, const void* packed_view,
                                        int precision, int sigmoid_kind, int bg_kind, float* alpha, float* weights,
                                        float* out, void* workspace, size_t workspace_bytes, void* stream) {
  NA_REQUIRE(pts, NA_ENULL, "na_render_plain_view_pts: null pts");
  return render_plain_view_impl(rays, pts, R, ts, T, hash_tables, packed_first, packed_view, precision, sigmoid_kind,
                                bg_kind, alpha, weights, out, workspace, workspace_bytes, stream);
}
#endif
