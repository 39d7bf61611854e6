// na_render_plain_view_ls: PlainNeRF.forward with the View head (src/nerf.py:326-361, src/refl.py:190-207) as ONE
// kernel on the LAYER-SYNCHRONOUS engine.  Same arithmetic and work items as render_fused.hip
// (a block = 32 consecutive steps of one ray, lane&31 = step, compositing = in-wave scan), different data flow:
//
//  * 8 waves = 4 row groups (64 output rows = two 32-row MFMA tiles) x 2 sample groups (NBLK blocks each).
//  * Activations live in LDS as ready-made MFMA B fragments, [group][block][chunk][lane] x 16 B: the lane that
//    produces 8 values of a chunk is the lane that consumes them, so ds_write_b128 / ds_read_b128 are lane-linear and
//    conflict-free; the implied k permutation is the engine's pi_perm, folded into the weight packing.
//  * A layer = an MFMA phase that is CHUNK-major: per 16-wide k chunk the wave streams its two A fragments straight
//    from global memory into registers (a 4-deep software-prefetched ring that runs across layer, MLP and pass
//    boundaries: no LDS-DMA, no weight ring in LDS) and feeds them to 2 x NBLK MFMAs whose B fragments come from LDS:
//    every LDS fragment read feeds TWO MFMAs (the one-read-per-MFMA structure of mlp_engine.h caps the matrix pipe
//    near 57 %).  All 2 x NBLK accumulator tiles stay in registers until the layer is complete; the activation
//    epilogue then overwrites the LDS fragments in place.
//  * The two sample groups run in ANTIPHASE, one workgroup barrier per phase: while one group streams MFMAs the
//    other runs its epilogue / encoder / compositing (VALU + LDS writes) on the same SIMDs, so VALU work never sits
//    between a wave's own MFMAs and each SIMD always has one MFMA-only wave.
//  * LDS: per group NBLK x (16 hidden + 4 init) chunks = 80 KiB, 160 KiB per workgroup (bf16: NBLK = 4; bf16x3 keeps
//    hi and lo planes: NBLK = 2).  The fifth init chunk of the View MLP (x, y, z, elev, azim) does not fit: every
//    consuming wave rebuilds its fragments in registers (scalar loads of the ray and of its per-ray elev/azim, which a
//    small pre-kernel writes once per ray; ~15 VALU per block) right before the two MFMAs that use them.
//  * `first.out` / `view.out` (65 and 3 rows) run block-per-wave (wave rg = block rg, all out tiles), which is also the
//    assignment of the hash-encoder prologue and of compositing, so density and colour never leave their wave.
//  * Sample group G = 2 * workgroup + g renders the rays G, G + nG, ... one after the other, block by block in step
//    order: the transmittance is carried across blocks (and passes) inside the group -- block partials meet in the
//    group's idle hidden region of LDS -- so there are no per-block partials in HBM and no finalize launch.
//
// Compiled once per precision (-DNA_PREC_INST=0|1|2).
#include <atomic>
#include "ls_pack.h"
#if NA_PREC_INST == 3
#include "ls_xsched.h"
#endif

namespace na {

#if NA_PREC_INST == 3
int render_ls_dispatch_f16x(ls::Args& a, hipStream_t s, int model) {
  return model == 1 ? ls::launch<NA_PREC_F16X, 1>(a, s) : model == 2 ? ls::launch<NA_PREC_F16X, 2>(a, s)
         : model == 3 ? ls::launch<NA_PREC_F16X, 3>(a, s) : model == 4 ? ls::launch<NA_PREC_F16X, 4>(a, s)
         : model == 5 ? ls::launch<NA_PREC_F16X, 5>(a, s) : model == 6 ? ls::launch<NA_PREC_F16X, 6>(a, s)
         : model == 7 ? ls::launch<NA_PREC_F16X, 7>(a, s) : model == 8 ? ls::launch<NA_PREC_F16X, 8>(a, s)
         : ls::launch<NA_PREC_F16X>(a, s);
}
int render_lsx_pack_head(int model, const float* const* w0, const float* const* b0, const float* const* w1, const float* const* b1,
                         int n_rl, char* packed, hipStream_t stream) {
  return ls::render_lsx_pack(model, w0, b0, w1, b1, packed, stream, n_rl);
}
int render_lsx_pack_hashmlp(const float* const* w, const float* const* b, int n_out, char* packed, hipStream_t stream) {
  return ls::render_lsx_pack(4, w, b, nullptr, nullptr, packed, stream, n_out);
}
int render_lsx_pack_fouriermlp(const float* const* w, const float* const* b, char* packed, hipStream_t stream) {
  return ls::render_lsx_pack(5, w, b, nullptr, nullptr, packed, stream);
}
int render_lsx_pack_mip(const float* const* w0, const float* const* b0, const float* const* w1, const float* const* b1, char* packed,
                        hipStream_t stream) {
  return ls::render_lsx_pack(6, w0, b0, w1, b1, packed, stream);
}
#elif NA_PREC_INST == 0
int render_ls_dispatch_bf16(ls::Args& a, hipStream_t s, int model) {
  return model == 1 ? ls::launch<NA_PREC_BF16, 1>(a, s) : model == 2 ? ls::launch<NA_PREC_BF16, 2>(a, s)
         : model == 3 ? ls::launch<NA_PREC_BF16, 3>(a, s) : ls::launch<NA_PREC_BF16>(a, s);
}
#elif NA_PREC_INST == 1
int render_ls_dispatch_bf16x3(ls::Args& a, hipStream_t s, int model) {
  return model == 1 ? ls::launch<NA_PREC_BF16X3, 1>(a, s) : model == 2 ? ls::launch<NA_PREC_BF16X3, 2>(a, s)
         : model == 3 ? ls::launch<NA_PREC_BF16X3, 3>(a, s) : model == 4 ? ls::launch<NA_PREC_BF16X3, 4>(a, s)
         : model == 9 ? ls::launch<NA_PREC_BF16X3, 9>(a, s) : ls::launch<NA_PREC_BF16X3>(a, s);
}
#elif NA_PREC_INST == 2
int render_ls_dispatch_f16(ls::Args& a, hipStream_t s, int model) {
  return model == 1 ? ls::launch<NA_PREC_F16, 1>(a, s) : model == 2 ? ls::launch<NA_PREC_F16, 2>(a, s)
         : model == 3 ? ls::launch<NA_PREC_F16, 3>(a, s) : ls::launch<NA_PREC_F16>(a, s);
}
#endif
int render_ls_dispatch_bf16(ls::Args& a, hipStream_t s, int model);
int render_ls_dispatch_bf16x3(ls::Args& a, hipStream_t s, int model);
int render_ls_dispatch_f16(ls::Args& a, hipStream_t s, int model);
int render_ls_dispatch_f16x(ls::Args& a, hipStream_t s, int model);
int render_lsx_pack_hashmlp(const float* const* w, const float* const* b, int n_out, char* packed, hipStream_t stream);
int render_lsx_pack_fouriermlp(const float* const* w, const float* const* b, char* packed, hipStream_t stream);
int render_lsx_pack_mip(const float* const* w0, const float* const* b0, const float* const* w1, const float* const* b1, char* packed,
                        hipStream_t stream);
int render_lsx_pack_head(int model, const float* const* w0, const float* const* b0, const float* const* w1, const float* const* b1,
                         int n_rl, char* packed, hipStream_t stream);

}  // namespace na

#if NA_PREC_INST == 0
using namespace na;

extern "C" size_t na_render_ls_packed_bytes(int precision) {
  if (precision != NA_PREC_BF16 && precision != NA_PREC_BF16X3 && precision != NA_PREC_F16 && precision != NA_PREC_F16X) return 0;
  return ls::packed_bytes(precision);
}

extern "C" int na_render_ls_pack(int precision, const float* const* w_first, const float* const* b_first,
                                 const float* const* w_view, const float* const* b_view, void* packed, void* stream) {
  NA_REQUIRE(w_first && b_first && w_view && b_view && packed, NA_ENULL, "na_render_ls_pack: null pointer");
  NA_REQUIRE(precision == NA_PREC_BF16 || precision == NA_PREC_BF16X3 || precision == NA_PREC_F16 || precision == NA_PREC_F16X,
             NA_EUNSUPPORTED, "na_render_ls_pack: precision %d", precision);
  if (precision == NA_PREC_F16X) {
    for (int i = 0; i < 6; ++i) NA_REQUIRE(w_first[i] && w_view[i], NA_ENULL, "na_render_ls_pack: weights[%d] is null", i);
    return ls::render_lsx_pack(0, w_first, b_first, w_view, b_view, (char*)packed, (hipStream_t)stream);
  }
  ls::PackArgs w;
  for (int i = 0; i < 6; ++i) {
    NA_REQUIRE(w_first[i] && w_view[i], NA_ENULL, "na_render_ls_pack: weights[%d] is null", i);
    w.w_first[i] = w_first[i]; w.b_first[i] = b_first[i];
    w.w_view[i] = w_view[i]; w.b_view[i] = b_view[i];
  }
  const int planes = precision == NA_PREC_BF16X3 ? 2 : 1;
  hipLaunchKernelGGL(ls::pack_ls_header_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ls::kMagic, (uint32_t)precision,
                     (uint32_t*)packed, (uint32_t)ls::kPairsPerPass);
  const int64_t total = 4 * 2 * (int64_t)ls::kPairsPerPass * 512 + 4 * ls::kNPhase * 256;
  hipLaunchKernelGGL(ls::pack_ls_kernel, dim3(grid_for(total, 256, 4096)), dim3(256), 0, (hipStream_t)stream, w, planes,
                     precision == NA_PREC_F16 ? 1 : 0, (char*)packed);
  return check_launch("na_render_ls_pack");
}

extern "C" size_t na_render_ls_workspace_bytes(int T, int64_t R) {
  if (T < 1 || R < 0) return 0;
  const int64_t nb = (T + 31) / 32;
  (void)nb;
  return (size_t)R * 2 * sizeof(float) + 256 + (NA_LS_TRACE ? 4096 + 256 : 0);
}

struct LsTrainOut { float* planes; float* view_rows; float* density; float* rgb; };
static int render_plain_view_ls_impl(const float* rays, const float* pts, int64_t R, const float* ts, int64_t ts_stride, int T,
                                     const float* hash_tables, const void* packed, int precision, int sigmoid_kind,
                                     int bg_kind, float* alpha, float* weights, float* out, void* workspace,
                                     size_t workspace_bytes, void* stream, const LsTrainOut* tr = nullptr) {
  NA_REQUIRE(T >= 1 && R >= 0, NA_EINVAL, "na_render_plain_view_ls: bad shape T=%d R=%lld", T, (long long)R);
  if (R == 0) return NA_OK;  // empty batch: a no-op before any pointer check (zero-size tensors carry null pointers)
  NA_REQUIRE(rays && ts && hash_tables && packed && out && workspace, NA_ENULL, "na_render_plain_view_ls: null pointer");
  NA_REQUIRE(precision == NA_PREC_BF16 || precision == NA_PREC_BF16X3 || precision == NA_PREC_F16 || precision == NA_PREC_F16X,
             NA_EUNSUPPORTED, "na_render_plain_view_ls: precision %d", precision);
  NA_REQUIRE(sigmoid_kind >= 0 && sigmoid_kind <= NA_SIG_IDENTITY, NA_EUNSUPPORTED, "na_render_plain_view_ls: sigmoid %d",
             sigmoid_kind);
  NA_REQUIRE(bg_kind == NA_BG_BLACK || bg_kind == NA_BG_WHITE, NA_EUNSUPPORTED, "na_render_plain_view_ls: bg %d", bg_kind);
  NA_REQUIRE(workspace_bytes >= na_render_ls_workspace_bytes(T, R), NA_EWORKSPACE,
             "na_render_plain_view_ls: workspace %zu < %zu bytes", workspace_bytes, na_render_ls_workspace_bytes(T, R));
  if (R == 0) return NA_OK;
  ls::Args a;
  a.rays = rays; a.ts = ts; a.ts_stride = ts_stride; a.pts = pts; a.tables = (const float4*)hash_tables;
  a.feat = nullptr; a.beta = nullptr; a.feat_ld = 0;
  a.packed = (const char*)packed; a.packed_size = (uint32_t)ls::packed_bytes(precision);
  a.alpha = alpha; a.weights = weights; a.out = out; a.bg_kind = bg_kind;
  a.R = R; a.T = T; a.nb = (T + 31) / 32;
  float* elaz = (float*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  a.elaz = elaz;
  hipLaunchKernelGGL(ls::ray_elaz_kernel, dim3(grid_for(R, 256, 4096)), dim3(256), 0, (hipStream_t)stream, rays, R, elaz);
  a.sigmoid_kind = sigmoid_kind;
  a.res = hash_resolutions();
  a.trace = NA_LS_TRACE ? (unsigned long long*)(((uintptr_t)(elaz + R * 2) + 255) & ~(uintptr_t)255) : nullptr;
  if (tr != nullptr) {  // MODEL 9: the training forward (bf16x3 only)
    const int64_t N = (int64_t)T * R;
    (void)N;
    a.y = tr->planes; a.park = tr->view_rows; a.rl = tr->density; a.feat = tr->rgb;  // (ls_engine.h Args: MODEL 9's outputs ride in fields MODEL 0 leaves alone)
    return render_ls_dispatch_bf16x3(a, (hipStream_t)stream, 9);
  }
  if (precision == NA_PREC_BF16) return render_ls_dispatch_bf16(a, (hipStream_t)stream, 0);
  if (precision == NA_PREC_F16) return render_ls_dispatch_f16(a, (hipStream_t)stream, 0);
  if (precision == NA_PREC_F16X) return render_ls_dispatch_f16x(a, (hipStream_t)stream, 0);
  return render_ls_dispatch_bf16x3(a, (hipStream_t)stream, 0);
}

// The training step's forward of PlainNeRF(view) as ONE launch (MODEL 9 = MODEL 0 in bf16x3 that also writes every Linear's output rows
// for the backward pass: ls_kernel.h train_store; what csrc/train_fwd.hip does layer by layer).  Replaces the twelve forward Linears of
// src/neural_blocks.py:279-296 (x 2) in a training iteration (runner.py:647-825).
extern "C" int na_train_plain_view_ls(const float* rays, const float* pts, int64_t R, const float* ts, int T, const float* hash_tables,
                                      const void* packed, int sigmoid_kind, float* planes, float* view_rows, float* density,
                                      float* rgb_pre, float* out, void* workspace, size_t workspace_bytes, void* stream) {
  NA_REQUIRE(T >= 1 && R >= 0, NA_EINVAL, "na_train_plain_view_ls: bad shape T=%d R=%lld", T, (long long)R);
  if (R == 0) return NA_OK;
  NA_REQUIRE(pts && planes && view_rows && density && rgb_pre, NA_ENULL, "na_train_plain_view_ls: null pointer");
  const int64_t N = (int64_t)T * R;
  NA_REQUIRE(N * 1024 < (1ll << 32), NA_EINVAL, "na_train_plain_view_ls: %lld samples (the row offsets are 32-bit: < 4 194 304)", (long long)N);
  const LsTrainOut tr = {planes, view_rows, density, rgb_pre};
  return render_plain_view_ls_impl(rays, pts, R, ts, 0, T, hash_tables, packed, NA_PREC_BF16X3, sigmoid_kind, NA_BG_BLACK, nullptr, nullptr,
                                   out, workspace, workspace_bytes, stream, &tr);
}

extern "C" int na_render_plain_view_ls(const float* rays, const float* pts, int64_t R, const float* ts, int T,
                                       const float* hash_tables, const void* packed, int precision, int sigmoid_kind,
                                       int bg_kind, float* alpha, float* weights, float* out, void* workspace,
                                       size_t workspace_bytes, void* stream) {
  return render_plain_view_ls_impl(rays, pts, R, ts, 0, T, hash_tables, packed, precision, sigmoid_kind, bg_kind, alpha, weights, out,
                                   workspace, workspace_bytes, stream);
}

// The same launch with PER-RAY steps ts_ray[R,T] (row r = the increasing sample distances of ray r): the fine pass of coarse ->
// fine rendering, whose steps come from na_resample_ts.  Positions are o + t d, interval lengths t[i + 1] - t[i] of the ray's own row.
extern "C" int na_render_plain_view_ls_rayts(const float* rays, int64_t R, const float* ts_ray, int T, const float* hash_tables,
                                             const void* packed, int precision, int sigmoid_kind, int bg_kind, float* alpha,
                                             float* weights, float* out, void* workspace, size_t workspace_bytes, void* stream) {
  return render_plain_view_ls_impl(rays, nullptr, R, ts_ray, T, T, hash_tables, packed, precision, sigmoid_kind, bg_kind, alpha, weights,
                                   out, workspace, workspace_bytes, stream);
}

// ---- PlainNeRF(view) + mip (config 3; src/nerf.py:256-261, 326-361, src/utils.py:23-140) as ONE launch, NA_PREC_F16X only:
// the 96 IPE features of every sample are generated in the kernel (MODEL 6) for the init and skip Linears of both MLPs
extern "C" size_t na_render_plain_mip_ls_packed_bytes(int precision) {
  return precision == NA_PREC_F16X ? ls::packed_bytes_x(6) : 0;
}

extern "C" int na_render_plain_mip_ls_pack(int precision, const float* const* w_first, const float* const* b_first,
                                           const float* const* w_view, const float* const* b_view, void* packed, void* stream) {
  NA_REQUIRE(w_first && b_first && w_view && b_view && packed, NA_ENULL, "na_render_plain_mip_ls_pack: null pointer");
  NA_REQUIRE(precision == NA_PREC_F16X, NA_EUNSUPPORTED, "na_render_plain_mip_ls_pack: precision %d (f16x only)", precision);
  for (int i = 0; i < 6; ++i)
    NA_REQUIRE(w_first[i] && w_view[i] && b_first[i] && b_view[i], NA_ENULL, "na_render_plain_mip_ls_pack: Linear %d is null", i);
  return render_lsx_pack_mip(w_first, b_first, w_view, b_view, (char*)packed, (hipStream_t)stream);
}

extern "C" int na_render_plain_mip_ls(const float* rays, int B, int H, int W, const float* ts, int T, const float* hash_tables,
                                      const void* packed, int precision, int mip_kind, int min_deg, int max_deg, float t_end,
                                      int sigmoid_kind, int bg_kind, float* alpha, float* weights, float* out, void* workspace,
                                      size_t workspace_bytes, void* stream) {
  NA_REQUIRE(T >= 1 && B >= 0 && H >= 0 && W >= 0, NA_EINVAL, "na_render_plain_mip_ls: bad shape T=%d B=%d H=%d W=%d", T, B, H, W);
  const int64_t R = (int64_t)B * H * W;
  if (R == 0) return NA_OK;
  NA_REQUIRE(rays && ts && hash_tables && packed && out && workspace, NA_ENULL, "na_render_plain_mip_ls: null pointer");
  NA_REQUIRE(precision == NA_PREC_F16X, NA_EUNSUPPORTED, "na_render_plain_mip_ls: precision %d (f16x only)", precision);
  NA_REQUIRE(mip_kind == 0 || mip_kind == 1, NA_EUNSUPPORTED, "na_render_plain_mip_ls: mip kind %d", mip_kind);
  NA_REQUIRE(max_deg - min_deg == 16, NA_EUNSUPPORTED, "na_render_plain_mip_ls: %d IPE degrees (the schedule is built for 16)",
             max_deg - min_deg);
  NA_REQUIRE(H >= 2, NA_EINVAL, "na_render_plain_mip_ls: pixel radii need H >= 2 rows per crop (H=%d)", H);
  NA_REQUIRE(sigmoid_kind >= 0 && sigmoid_kind <= NA_SIG_IDENTITY, NA_EUNSUPPORTED, "na_render_plain_mip_ls: sigmoid %d", sigmoid_kind);
  NA_REQUIRE(bg_kind == NA_BG_BLACK || bg_kind == NA_BG_WHITE, NA_EUNSUPPORTED, "na_render_plain_mip_ls: bg %d", bg_kind);
  NA_REQUIRE(workspace_bytes >= na_render_ls_workspace_bytes(T, R), NA_EWORKSPACE,
             "na_render_plain_mip_ls: workspace %zu < %zu bytes", workspace_bytes, na_render_ls_workspace_bytes(T, R));
  ls::Args a;
  a.rays = rays; a.ts = ts; a.pts = nullptr; a.tables = (const float4*)hash_tables;
  a.feat = nullptr; a.beta = nullptr; a.feat_ld = 0;
  a.packed = (const char*)packed; a.packed_size = (uint32_t)ls::packed_bytes_x(6);
  a.alpha = alpha; a.weights = weights; a.out = out; a.bg_kind = bg_kind;
  a.R = R; a.T = T; a.nb = (T + 31) / 32;
  a.mip_H = H; a.mip_W = W; a.mip_kind = mip_kind; a.mip_min_deg = min_deg; a.mip_nd = max_deg - min_deg; a.mip_t_end = t_end;
  float* elaz = (float*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  a.elaz = elaz;
  hipLaunchKernelGGL(ls::ray_elaz_kernel, dim3(grid_for(R, 256, 4096)), dim3(256), 0, (hipStream_t)stream, rays, R, elaz);
  a.sigmoid_kind = sigmoid_kind;
  a.res = hash_resolutions();
  a.trace = NA_LS_TRACE ? (unsigned long long*)(((uintptr_t)(elaz + R * 2) + 255) & ~(uintptr_t)255) : nullptr;
  return render_ls_dispatch_f16x(a, (hipStream_t)stream, 6);
}

// ---- PlainNeRF with the Positional head (`make original`: src/nerf.py:340-361 + src/refl.py:230-245) and with PosLinearView
// (`make dnerf`: src/refl.py:248-290, optional refl_latent rows of src/nerf.py:1272-1278) as ONE launch each (MODEL 7 / 8), NA_PREC_F16X only
extern "C" size_t na_render_plain_pos_ls_packed_bytes(int precision) { return precision == NA_PREC_F16X ? ls::packed_bytes_x(7) : 0; }
extern "C" size_t na_render_plain_plv_ls_packed_bytes(int precision) { return precision == NA_PREC_F16X ? ls::packed_bytes_x(8) : 0; }

extern "C" size_t na_render_head_ls_workspace_bytes(int T, int64_t R) {
  if (T < 1 || R < 0) return 0;
  return na_render_ls_workspace_bytes(T, R) + 256 + ls::kParkBytes;
}

extern "C" int na_render_plain_pos_ls_pack(int precision, const float* const* w_first, const float* const* b_first,
                                           const float* const* w_pos, const float* const* b_pos, void* packed, void* stream) {
  NA_REQUIRE(w_first && b_first && w_pos && b_pos && packed, NA_ENULL, "na_render_plain_pos_ls_pack: null pointer");
  NA_REQUIRE(precision == NA_PREC_F16X, NA_EUNSUPPORTED, "na_render_plain_pos_ls_pack: precision %d (f16x only)", precision);
  for (int i = 0; i < 6; ++i) NA_REQUIRE(w_first[i] && b_first[i], NA_ENULL, "na_render_plain_pos_ls_pack: first Linear %d is null", i);
  for (int i = 0; i < 7; ++i) NA_REQUIRE(w_pos[i] && b_pos[i], NA_ENULL, "na_render_plain_pos_ls_pack: pos Linear %d is null", i);
  return render_lsx_pack_head(7, w_first, b_first, w_pos, b_pos, 0, (char*)packed, (hipStream_t)stream);
}

extern "C" int na_render_plain_plv_ls_pack(int precision, const float* const* w_first, const float* const* b_first,
                                           const float* const* w_head, const float* const* b_head, int n_rl, void* packed, void* stream) {
  NA_REQUIRE(w_first && b_first && w_head && b_head && packed, NA_ENULL, "na_render_plain_plv_ls_pack: null pointer");
  NA_REQUIRE(precision == NA_PREC_F16X, NA_EUNSUPPORTED, "na_render_plain_plv_ls_pack: precision %d (f16x only)", precision);
  NA_REQUIRE(n_rl >= 0 && n_rl <= 3, NA_EUNSUPPORTED, "na_render_plain_plv_ls_pack: %d refl_latent columns (0..3)", n_rl);
  for (int i = 0; i < 6; ++i) NA_REQUIRE(w_first[i] && b_first[i], NA_ENULL, "na_render_plain_plv_ls_pack: first Linear %d is null", i);
  for (int i = 0; i < 8; ++i) NA_REQUIRE(w_head[i] && b_head[i], NA_ENULL, "na_render_plain_plv_ls_pack: head Linear %d is null", i);
  return render_lsx_pack_head(8, w_first, b_first, w_head, b_head, n_rl, (char*)packed, (hipStream_t)stream);
}

static int render_plain_head_ls_impl(const char* what, int model, const float* rays, const float* pts, int64_t R, const float* ts, int T,
                                     const float* hash_tables, const float* hash_tables_refl, const float* refl_latent, int64_t rl_ld,
                                     int n_rl, const void* packed, int precision, int sigmoid_kind, int bg_kind, float* alpha,
                                     float* weights, float* out, void* workspace, size_t workspace_bytes, void* stream) {
  NA_REQUIRE(T >= 1 && R >= 0, NA_EINVAL, "%s: bad shape T=%d R=%lld", what, T, (long long)R);
  if (R == 0) return NA_OK;
  NA_REQUIRE(rays && ts && hash_tables && hash_tables_refl && packed && out && workspace, NA_ENULL, "%s: null pointer", what);
  NA_REQUIRE(precision == NA_PREC_F16X, NA_EUNSUPPORTED, "%s: precision %d (f16x only)", what, precision);
  NA_REQUIRE(sigmoid_kind >= 0 && sigmoid_kind <= NA_SIG_IDENTITY, NA_EUNSUPPORTED, "%s: sigmoid %d", what, sigmoid_kind);
  // (MODEL 8 applies the activation to 67 rows per sample inside the kernel: the four sigmoid-shaped kinds only)
  NA_REQUIRE(model != 8 || sigmoid_kind == NA_SIG_NORMAL || sigmoid_kind == NA_SIG_THIN || sigmoid_kind == NA_SIG_FAT || sigmoid_kind == NA_SIG_UPSHIFTED,
             NA_EUNSUPPORTED, "%s: sigmoid kind %d (normal | thin | fat | upshifted)", what, sigmoid_kind);
  NA_REQUIRE(bg_kind == NA_BG_BLACK || bg_kind == NA_BG_WHITE, NA_EUNSUPPORTED, "%s: bg %d", what, bg_kind);
  NA_REQUIRE(n_rl >= 0 && n_rl <= 3 && (n_rl == 0 || (refl_latent && rl_ld >= n_rl && rl_ld < (1 << 20))), NA_EINVAL,
             "%s: refl_latent n=%d ld=%lld", what, n_rl, (long long)rl_ld);
  NA_REQUIRE(workspace_bytes >= na_render_head_ls_workspace_bytes(T, R), NA_EWORKSPACE, "%s: workspace %zu < %zu bytes", what,
             workspace_bytes, na_render_head_ls_workspace_bytes(T, R));
  ls::Args a;
  a.rays = rays; a.ts = ts; a.ts_stride = 0; a.pts = pts; a.tables = (const float4*)hash_tables;
  a.feat = nullptr; a.beta = nullptr; a.feat_ld = 0;
  a.packed = (const char*)packed; a.packed_size = (uint32_t)ls::packed_bytes_x(model);
  a.alpha = alpha; a.weights = weights; a.out = out; a.bg_kind = bg_kind;
  a.R = R; a.T = T; a.nb = (T + 31) / 32;
  a.elaz = rays;  // (MODEL 8's group-wide ray table also fetches two floats per ray from here: any readable 8 R bytes)
  a.sigmoid_kind = sigmoid_kind;
  a.res = hash_resolutions();
  a.trace = nullptr;
  a.tables2 = (const float4*)hash_tables_refl;
  a.park = (float*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  if (NA_LS_TRACE) a.trace = (unsigned long long*)((char*)a.park + ls::kParkBytes);  // (tools/head_trace.py; the base size's per-ray scratch is unused here)
  a.rl = n_rl > 0 ? refl_latent : nullptr; a.rl_ld = (int)rl_ld; a.n_rl = n_rl;
  return render_ls_dispatch_f16x(a, (hipStream_t)stream, model);
}

extern "C" int na_render_plain_pos_ls(const float* rays, const float* pts, int64_t R, const float* ts, int T, const float* hash_tables,
                                      const float* hash_tables_refl, const void* packed, int precision, int sigmoid_kind, int bg_kind,
                                      float* alpha, float* weights, float* out, void* workspace, size_t workspace_bytes, void* stream) {
  return render_plain_head_ls_impl("na_render_plain_pos_ls", 7, rays, pts, R, ts, T, hash_tables, hash_tables_refl, nullptr, 0, 0, packed,
                                   precision, sigmoid_kind, bg_kind, alpha, weights, out, workspace, workspace_bytes, stream);
}

extern "C" int na_render_plain_plv_ls(const float* rays, const float* pts, int64_t R, const float* ts, int T, const float* hash_tables,
                                      const float* hash_tables_refl, const float* refl_latent, int64_t rl_ld, int n_rl, const void* packed,
                                      int precision, int sigmoid_kind, int bg_kind, float* alpha, float* weights, float* out,
                                      void* workspace, size_t workspace_bytes, void* stream) {
  return render_plain_head_ls_impl("na_render_plain_plv_ls", 8, rays, pts, R, ts, T, hash_tables, hash_tables_refl, refl_latent, rl_ld,
                                   n_rl, packed, precision, sigmoid_kind, bg_kind, alpha, weights, out, workspace, workspace_bytes, stream);
}

// ---- TinyNeRF on the same engine (src/nerf.py:278-305)
extern "C" size_t na_render_tiny_ls_packed_bytes(int precision) {
  if (precision != NA_PREC_BF16 && precision != NA_PREC_BF16X3 && precision != NA_PREC_F16 && precision != NA_PREC_F16X) return 0;
  return ls::packed_bytes(precision, ls::kTinyPairs);
}

extern "C" int na_render_tiny_ls_pack(int precision, const float* const* w, const float* const* b, void* packed, void* stream) {
  NA_REQUIRE(w && b && packed, NA_ENULL, "na_render_tiny_ls_pack: null pointer");
  NA_REQUIRE(precision == NA_PREC_BF16 || precision == NA_PREC_BF16X3 || precision == NA_PREC_F16 || precision == NA_PREC_F16X, NA_EUNSUPPORTED,
             "na_render_tiny_ls_pack: precision %d", precision);
  ls::TinyPackArgs pa;
  for (int i = 0; i < 8; ++i) {
    NA_REQUIRE(w[i], NA_ENULL, "na_render_tiny_ls_pack: weights[%d] is null", i);
    pa.w[i] = w[i]; pa.b[i] = b[i];
  }
  if (precision == NA_PREC_F16X) return ls::render_lsx_pack(1, w, b, nullptr, nullptr, (char*)packed, (hipStream_t)stream);
  const int planes = precision == NA_PREC_BF16X3 ? 2 : 1;
  hipLaunchKernelGGL(ls::pack_ls_header_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ls::kMagic, (uint32_t)precision,
                     (uint32_t*)packed, (uint32_t)ls::kTinyPairs);
  const int64_t total = 4 * 2 * (int64_t)ls::kTinyPairs * 512 + 4 * ls::kNPhase * 256;
  hipLaunchKernelGGL(ls::pack_ls_tiny_kernel, dim3(grid_for(total, 256, 4096)), dim3(256), 0, (hipStream_t)stream, pa, planes,
                     precision == NA_PREC_F16 ? 1 : 0, (char*)packed);
  return check_launch("na_render_tiny_ls_pack");
}

extern "C" int na_render_tiny_ls(const float* rays, const float* pts, int64_t R, const float* ts, int T, const void* packed,
                                 int precision, int sigmoid_kind, int bg_kind, float* alpha, float* weights, float* out,
                                 void* stream) {
  NA_REQUIRE(T >= 1 && R >= 0, NA_EINVAL, "na_render_tiny_ls: bad shape T=%d R=%lld", T, (long long)R);
  if (R == 0) return NA_OK;
  NA_REQUIRE(rays && ts && packed && out, NA_ENULL, "na_render_tiny_ls: null pointer");
  NA_REQUIRE(precision == NA_PREC_BF16 || precision == NA_PREC_BF16X3 || precision == NA_PREC_F16 || precision == NA_PREC_F16X, NA_EUNSUPPORTED,
             "na_render_tiny_ls: precision %d", precision);
  NA_REQUIRE(sigmoid_kind >= 0 && sigmoid_kind <= NA_SIG_IDENTITY, NA_EUNSUPPORTED, "na_render_tiny_ls: sigmoid %d", sigmoid_kind);
  NA_REQUIRE(bg_kind == NA_BG_BLACK || bg_kind == NA_BG_WHITE, NA_EUNSUPPORTED, "na_render_tiny_ls: bg %d", bg_kind);
  ls::Args a;
  a.rays = rays; a.ts = ts; a.pts = pts; a.tables = nullptr;
  a.feat = nullptr; a.beta = nullptr; a.feat_ld = 0;
  a.packed = (const char*)packed; a.packed_size = (uint32_t)ls::packed_bytes(precision, ls::kTinyPairs);
  a.alpha = alpha; a.weights = weights; a.out = out; a.bg_kind = bg_kind;
  a.R = R; a.T = T; a.nb = (T + 31) / 32;
  a.elaz = nullptr;
  a.sigmoid_kind = sigmoid_kind;
  a.res = hash_resolutions();
  a.trace = nullptr;
  if (precision == NA_PREC_BF16) return render_ls_dispatch_bf16(a, (hipStream_t)stream, 1);
  if (precision == NA_PREC_F16) return render_ls_dispatch_f16(a, (hipStream_t)stream, 1);
  if (precision == NA_PREC_F16X) return render_ls_dispatch_f16x(a, (hipStream_t)stream, 1);
  return render_ls_dispatch_bf16x3(a, (hipStream_t)stream, 1);
}

// ---- View head + compositing on per-sample (density source, latent) rows: VolSDF's second half (src/nerf.py:981-1013)
extern "C" size_t na_render_view_ls_packed_bytes(int precision) {
  if (precision != NA_PREC_BF16 && precision != NA_PREC_BF16X3 && precision != NA_PREC_F16 && precision != NA_PREC_F16X) return 0;
  return ls::packed_bytes(precision, ls::kViewPairs);
}

extern "C" int na_render_view_ls_pack(int precision, const float* const* w, const float* const* b, void* packed, void* stream) {
  NA_REQUIRE(w && b && packed, NA_ENULL, "na_render_view_ls_pack: null pointer");
  NA_REQUIRE(precision == NA_PREC_BF16 || precision == NA_PREC_BF16X3 || precision == NA_PREC_F16 || precision == NA_PREC_F16X, NA_EUNSUPPORTED,
             "na_render_view_ls_pack: precision %d", precision);
  ls::ViewPackArgs pa;
  for (int i = 0; i < 6; ++i) {
    NA_REQUIRE(w[i], NA_ENULL, "na_render_view_ls_pack: weights[%d] is null", i);
    pa.w[i] = w[i]; pa.b[i] = b[i];
  }
  if (precision == NA_PREC_F16X) return ls::render_lsx_pack(2, w, b, nullptr, nullptr, (char*)packed, (hipStream_t)stream);
  const int planes = precision == NA_PREC_BF16X3 ? 2 : 1;
  hipLaunchKernelGGL(ls::pack_ls_header_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ls::kMagic, (uint32_t)precision,
                     (uint32_t*)packed, (uint32_t)ls::kViewPairs);
  const int64_t total = 4 * 2 * (int64_t)ls::kViewPairs * 512 + 4 * ls::kNPhase * 256;
  hipLaunchKernelGGL(ls::pack_ls_view_kernel, dim3(grid_for(total, 256, 4096)), dim3(256), 0, (hipStream_t)stream, pa, planes,
                     precision == NA_PREC_F16 ? 1 : 0, (char*)packed);
  return check_launch("na_render_view_ls_pack");
}

extern "C" int na_render_view_ls(const float* rays, const float* pts, int64_t R, const float* ts, int T, const float* feat,
                                 int feat_ld, const float* beta, const void* packed, int precision, int sigmoid_kind,
                                 int bg_kind, float* alpha, float* weights, float* out, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  NA_REQUIRE(T >= 1 && R >= 0, NA_EINVAL, "na_render_view_ls: bad shape T=%d R=%lld", T, (long long)R);
  if (R == 0) return NA_OK;
  NA_REQUIRE(rays && ts && feat && beta && packed && out && workspace, NA_ENULL, "na_render_view_ls: null pointer");
  NA_REQUIRE(feat_ld >= 65, NA_EINVAL, "na_render_view_ls: feat_ld %d < 65 (signed distance + 64 latent columns)", feat_ld);
  NA_REQUIRE(precision == NA_PREC_BF16 || precision == NA_PREC_BF16X3 || precision == NA_PREC_F16 || precision == NA_PREC_F16X, NA_EUNSUPPORTED,
             "na_render_view_ls: precision %d", precision);
  NA_REQUIRE(sigmoid_kind >= 0 && sigmoid_kind <= NA_SIG_IDENTITY, NA_EUNSUPPORTED, "na_render_view_ls: sigmoid %d", sigmoid_kind);
  NA_REQUIRE(bg_kind == NA_BG_BLACK || bg_kind == NA_BG_WHITE, NA_EUNSUPPORTED, "na_render_view_ls: bg %d", bg_kind);
  NA_REQUIRE(workspace_bytes >= na_render_ls_workspace_bytes(T, R), NA_EWORKSPACE, "na_render_view_ls: workspace %zu < %zu bytes",
             workspace_bytes, na_render_ls_workspace_bytes(T, R));
  ls::Args a;
  a.rays = rays; a.ts = ts; a.pts = pts; a.tables = nullptr;
  a.packed = (const char*)packed; a.packed_size = (uint32_t)ls::packed_bytes(precision, ls::kViewPairs);
  a.alpha = alpha; a.weights = weights; a.out = out; a.bg_kind = bg_kind;
  a.R = R; a.T = T; a.nb = (T + 31) / 32;
  a.feat = feat; a.feat_ld = feat_ld; a.beta = beta;
  float* elaz = (float*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  a.elaz = elaz;
  hipLaunchKernelGGL(ls::ray_elaz_kernel, dim3(grid_for(R, 256, 4096)), dim3(256), 0, (hipStream_t)stream, rays, R, elaz);
  a.sigmoid_kind = sigmoid_kind;
  a.res = hash_resolutions();
  a.trace = nullptr;
  if (precision == NA_PREC_BF16) return render_ls_dispatch_bf16(a, (hipStream_t)stream, 2);
  if (precision == NA_PREC_F16) return render_ls_dispatch_f16(a, (hipStream_t)stream, 2);
  if (precision == NA_PREC_F16X) return render_ls_dispatch_f16x(a, (hipStream_t)stream, 2);
  return render_ls_dispatch_bf16x3(a, (hipStream_t)stream, 2);
}

// ---- VolSDF with the SIREN SDF network as one kernel (src/sdf.py:278-287 + src/nerf.py:981-1013)
extern "C" size_t na_render_volsdf_siren_ls_packed_bytes(int precision) {
  if (precision != NA_PREC_BF16 && precision != NA_PREC_BF16X3 && precision != NA_PREC_F16 && precision != NA_PREC_F16X) return 0;
  return ls::packed_bytes(precision, ls::kSirenPairs);
}

extern "C" int na_render_volsdf_siren_ls_pack(int precision, const float* const* w_sdf, const float* const* b_sdf,
                                              const float* const* w_view, const float* const* b_view, void* packed, void* stream) {
  NA_REQUIRE(w_sdf && b_sdf && w_view && b_view && packed, NA_ENULL, "na_render_volsdf_siren_ls_pack: null pointer");
  NA_REQUIRE(precision == NA_PREC_BF16 || precision == NA_PREC_BF16X3 || precision == NA_PREC_F16 || precision == NA_PREC_F16X, NA_EUNSUPPORTED,
             "na_render_volsdf_siren_ls_pack: precision %d", precision);
  ls::SirenPackArgs pa;
  for (int i = 0; i < 7; ++i) {
    NA_REQUIRE(w_sdf[i], NA_ENULL, "na_render_volsdf_siren_ls_pack: sdf weights[%d] is null", i);
    pa.ws[i] = w_sdf[i]; pa.bs[i] = b_sdf[i];
  }
  for (int i = 0; i < 6; ++i) {
    NA_REQUIRE(w_view[i], NA_ENULL, "na_render_volsdf_siren_ls_pack: view weights[%d] is null", i);
    pa.wv[i] = w_view[i]; pa.bv[i] = b_view[i];
  }
  if (precision == NA_PREC_F16X) return ls::render_lsx_pack(3, w_sdf, b_sdf, w_view, b_view, (char*)packed, (hipStream_t)stream);
  const int planes = precision == NA_PREC_BF16X3 ? 2 : 1;
  hipLaunchKernelGGL(ls::pack_ls_header_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ls::kMagic, (uint32_t)precision,
                     (uint32_t*)packed, (uint32_t)ls::kSirenPairs);
  const int64_t total = 4 * 2 * (int64_t)ls::kSirenPairs * 512 + 4 * ls::kNPhase * 256;
  hipLaunchKernelGGL(ls::pack_ls_siren_kernel, dim3(grid_for(total, 256, 4096)), dim3(256), 0, (hipStream_t)stream, pa, planes,
                     precision == NA_PREC_F16 ? 1 : 0, (char*)packed);
  return check_launch("na_render_volsdf_siren_ls_pack");
}

extern "C" int na_render_volsdf_siren_ls(const float* rays, const float* pts, int64_t R, const float* ts, int T, const float* beta,
                                         const void* packed, int precision, int sigmoid_kind, int bg_kind, float* alpha,
                                         float* weights, float* out, void* workspace, size_t workspace_bytes, void* stream) {
  NA_REQUIRE(T >= 1 && R >= 0, NA_EINVAL, "na_render_volsdf_siren_ls: bad shape T=%d R=%lld", T, (long long)R);
  if (R == 0) return NA_OK;
  NA_REQUIRE(rays && ts && beta && packed && out && workspace, NA_ENULL, "na_render_volsdf_siren_ls: null pointer");
  NA_REQUIRE(precision == NA_PREC_BF16 || precision == NA_PREC_BF16X3 || precision == NA_PREC_F16 || precision == NA_PREC_F16X, NA_EUNSUPPORTED,
             "na_render_volsdf_siren_ls: precision %d", precision);
  NA_REQUIRE(sigmoid_kind >= 0 && sigmoid_kind <= NA_SIG_IDENTITY, NA_EUNSUPPORTED, "na_render_volsdf_siren_ls: sigmoid %d", sigmoid_kind);
  NA_REQUIRE(bg_kind == NA_BG_BLACK || bg_kind == NA_BG_WHITE, NA_EUNSUPPORTED, "na_render_volsdf_siren_ls: bg %d", bg_kind);
  NA_REQUIRE(workspace_bytes >= na_render_ls_workspace_bytes(T, R), NA_EWORKSPACE,
             "na_render_volsdf_siren_ls: workspace %zu < %zu bytes", workspace_bytes, na_render_ls_workspace_bytes(T, R));
  ls::Args a;
  a.rays = rays; a.ts = ts; a.pts = pts; a.tables = nullptr;
  a.packed = (const char*)packed; a.packed_size = (uint32_t)ls::packed_bytes(precision, ls::kSirenPairs);
  a.alpha = alpha; a.weights = weights; a.out = out; a.bg_kind = bg_kind;
  a.R = R; a.T = T; a.nb = (T + 31) / 32;
  a.feat = nullptr; a.feat_ld = 0; a.beta = beta;
  float* elaz = (float*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  a.elaz = elaz;
  hipLaunchKernelGGL(ls::ray_elaz_kernel, dim3(grid_for(R, 256, 4096)), dim3(256), 0, (hipStream_t)stream, rays, R, elaz);
  a.sigmoid_kind = sigmoid_kind;
  a.res = hash_resolutions();
  a.trace = nullptr;
  if (precision == NA_PREC_BF16) return render_ls_dispatch_bf16(a, (hipStream_t)stream, 3);
  if (precision == NA_PREC_F16) return render_ls_dispatch_f16(a, (hipStream_t)stream, 3);
  if (precision == NA_PREC_F16X) return render_ls_dispatch_f16x(a, (hipStream_t)stream, 3);
  return render_ls_dispatch_bf16x3(a, (hipStream_t)stream, 3);
}
// ---- a hash-encoded SkipConnMLP on the layer-synchronous engine, rows to HBM (D-NeRF's deformation network, src/nerf.py:1250-1257,
// 1267-1270): NA_PREC_F16X only -- the other precisions run it through na_mlp_forward
extern "C" size_t na_mlp_hash_ls_packed_bytes(int precision) {
  if (precision == NA_PREC_BF16X3) return ls::packed_bytes(precision, ls::kHashMlpPairs);
  return precision == NA_PREC_F16X ? ls::packed_bytes_x(4) : 0;
}

extern "C" int na_mlp_hash_ls_pack(int precision, const float* const* w, const float* const* b, int n_out, void* packed, void* stream) {
  NA_REQUIRE(w && b && packed, NA_ENULL, "na_mlp_hash_ls_pack: null pointer");
  NA_REQUIRE(precision == NA_PREC_F16X || precision == NA_PREC_BF16X3, NA_EUNSUPPORTED, "na_mlp_hash_ls_pack: precision %d (f16x | bf16x3)", precision);
  for (int i = 0; i < 7; ++i) NA_REQUIRE(w[i], NA_ENULL, "na_mlp_hash_ls_pack: weights[%d] is null", i);
  if (precision == NA_PREC_BF16X3) {  // round 6: the three-product stream (two output tiles: up to 64 rows)
    NA_REQUIRE(n_out >= 1 && n_out <= 64, NA_EUNSUPPORTED, "na_mlp_hash_ls_pack: n_out %d (bf16x3: 1..64, two output tiles)", n_out);
    ls::HashMlpPackArgs pw;
    for (int i = 0; i < 7; ++i) { pw.w[i] = w[i]; pw.b[i] = b[i]; }
    pw.n_out = n_out;
    hipLaunchKernelGGL(ls::pack_ls_header_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ls::kMagic, (uint32_t)precision,
                       (uint32_t*)packed, (uint32_t)ls::kHashMlpPairs);
    const int64_t total = 4 * 2 * (int64_t)ls::kHashMlpPairs * 512 + 4 * ls::kNPhase * 256;
    hipLaunchKernelGGL(ls::pack_ls_hashmlp_kernel, dim3(grid_for(total, 256, 4096)), dim3(256), 0, (hipStream_t)stream, pw, 2, 0, (char*)packed);
    return check_launch("na_mlp_hash_ls_pack");
  }
  NA_REQUIRE(n_out >= 1 && n_out <= 32, NA_EUNSUPPORTED, "na_mlp_hash_ls_pack: n_out %d (f16x: 1..32, one output tile)", n_out);
  return render_lsx_pack_hashmlp(w, b, n_out, (char*)packed, (hipStream_t)stream);
}

extern "C" int na_mlp_hash_ls(const float* rays, const float* pts, int64_t R, const float* ts, int T, const float* hash_tables,
                              const void* packed, int precision, int n_out, float* y, int64_t y_ld, void* stream) {
  NA_REQUIRE(T >= 1 && R >= 0, NA_EINVAL, "na_mlp_hash_ls: bad shape T=%d R=%lld", T, (long long)R);
  if (R == 0) return NA_OK;
  NA_REQUIRE(rays && ts && hash_tables && packed && y, NA_ENULL, "na_mlp_hash_ls: null pointer");
  NA_REQUIRE(precision == NA_PREC_F16X || precision == NA_PREC_BF16X3, NA_EUNSUPPORTED, "na_mlp_hash_ls: precision %d (f16x | bf16x3)", precision);
  NA_REQUIRE(n_out >= 1 && n_out <= (precision == NA_PREC_BF16X3 ? 64 : 32) && y_ld >= n_out && y_ld < (1 << 20), NA_EINVAL,
             "na_mlp_hash_ls: n_out %d, y_ld %lld", n_out, (long long)y_ld);
  ls::Args a;
  a.rays = rays; a.ts = ts; a.pts = pts; a.tables = (const float4*)hash_tables;
  a.feat = nullptr; a.beta = nullptr; a.feat_ld = 0;
  a.packed = (const char*)packed;
  a.packed_size = (uint32_t)(precision == NA_PREC_BF16X3 ? ls::packed_bytes(precision, ls::kHashMlpPairs) : ls::packed_bytes_x(4));
  a.alpha = nullptr; a.weights = nullptr; a.out = nullptr; a.bg_kind = NA_BG_BLACK;
  a.R = R; a.T = T; a.nb = (T + 31) / 32;
  a.elaz = nullptr; a.sigmoid_kind = 0;
  a.res = hash_resolutions();
  a.trace = nullptr;
  a.y = y; a.y_ld = (int)y_ld; a.n_out = n_out;
  if (precision == NA_PREC_BF16X3) return render_ls_dispatch_bf16x3(a, (hipStream_t)stream, 4);
  return render_ls_dispatch_f16x(a, (hipStream_t)stream, 4);
}
// ---- a Fourier-encoded SkipConnMLP on the layer-synchronous engine, rows to HBM (VolSDF's MLP SDF network, src/sdf.py:250-258):
// NA_PREC_F16X only
extern "C" size_t na_mlp_fourier_ls_packed_bytes(int precision) {
  return precision == NA_PREC_F16X ? ls::packed_bytes_x(5) : 0;
}

extern "C" int na_mlp_fourier_ls_pack(int precision, const float* const* w, const float* const* b, void* packed, void* stream) {
  NA_REQUIRE(w && b && packed, NA_ENULL, "na_mlp_fourier_ls_pack: null pointer");
  NA_REQUIRE(precision == NA_PREC_F16X, NA_EUNSUPPORTED, "na_mlp_fourier_ls_pack: precision %d (f16x only)", precision);
  for (int i = 0; i < 8; ++i) NA_REQUIRE(w[i], NA_ENULL, "na_mlp_fourier_ls_pack: weights[%d] is null", i);
  return render_lsx_pack_fouriermlp(w, b, (char*)packed, (hipStream_t)stream);
}

extern "C" int na_mlp_fourier_ls(const float* rays, const float* pts, int64_t R, const float* ts, int T, const float* basis,
                                 const void* packed, int precision, float* y, int64_t y_ld, void* stream) {
  NA_REQUIRE(T >= 1 && R >= 0, NA_EINVAL, "na_mlp_fourier_ls: bad shape T=%d R=%lld", T, (long long)R);
  if (R == 0) return NA_OK;
  NA_REQUIRE(rays && ts && basis && packed && y, NA_ENULL, "na_mlp_fourier_ls: null pointer");
  NA_REQUIRE(precision == NA_PREC_F16X, NA_EUNSUPPORTED, "na_mlp_fourier_ls: precision %d (f16x only)", precision);
  NA_REQUIRE(y_ld >= 65 && y_ld < (1 << 20), NA_EINVAL, "na_mlp_fourier_ls: y_ld %lld < 65", (long long)y_ld);
  NA_REQUIRE(((uintptr_t)basis & 15) == 0, NA_EINVAL, "na_mlp_fourier_ls: basis must be 16-byte aligned");
  ls::Args a;
  a.rays = rays; a.ts = ts; a.pts = pts; a.tables = (const float4*)basis;
  a.feat = nullptr; a.beta = nullptr; a.feat_ld = 0;
  a.packed = (const char*)packed; a.packed_size = (uint32_t)ls::packed_bytes_x(5);
  a.alpha = nullptr; a.weights = nullptr; a.out = nullptr; a.bg_kind = NA_BG_BLACK;
  a.R = R; a.T = T; a.nb = (T + 31) / 32;
  a.elaz = rays;  // (the group-wide ray table also fetches two floats per ray from here: any readable 8 R bytes)
  a.sigmoid_kind = 0;
  a.res = hash_resolutions();
  a.trace = nullptr;
  a.y = y; a.y_ld = (int)y_ld; a.n_out = 65;
  return render_ls_dispatch_f16x(a, (hipStream_t)stream, 5);
}
#endif
