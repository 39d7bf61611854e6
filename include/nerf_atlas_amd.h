/*
 * nerf_atlas_amd.h -- C ABI of the MI355X (gfx950) NeRF volume-rendering hot path.
 *
 * The reference (JulianKnodt/nerf_atlas) has no FFI: its "operator API" is the Python
 * protocol of src/cameras.py, src/nerf.py, src/neural_blocks.py.  Each
 * entry point below replaces one reference operator (file:line cited per function) and is
 * what a reference-side ctypes stub would bind.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes.  Every pointer is a DEVICE pointer owned by the
 *    caller (PyTorch allocates); the library never allocates or frees caller memory.
 *  - Every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = the
 *    default stream).  No hidden synchronisation.
 *  - Return 0 on success, a negative NA_E* code otherwise; na_last_error() gives a
 *    thread-local message.
 *  - Layouts follow the reference: sample tensors are T-major ([T, R, ...], R = B*H*W rays).
 */
#ifndef NERF_ATLAS_AMD_H
#define NERF_ATLAS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NA_VERSION 100 /* 0.1.0 */

enum {
  NA_OK = 0,
  NA_EINVAL = -1,      /* bad shape / argument */
  NA_ENULL = -2,       /* required pointer is NULL */
  NA_EUNSUPPORTED = -3,/* shape or kind has no kernel */
  NA_EHIP = -4,        /* HIP runtime error (launch failed) */
  NA_EWORKSPACE = -5   /* workspace too small */
};

/* activation applied to the INPUT of every hidden Linear and of `out` (src/neural_blocks.py:290-296) */
enum { NA_ACT_NONE = 0, NA_ACT_LEAKY_RELU = 1, NA_ACT_SIN = 2 };
/* encoder fused in front of a SkipConnMLP */
enum { NA_ENC_NONE = 0, NA_ENC_HASH = 1, NA_ENC_FOURIER = 2 };
/* MLP arithmetic */
enum {
  NA_PREC_BF16 = 0,   /* bf16 operands, fp32 accumulate (1 MFMA product)            */
  NA_PREC_BF16X3 = 1, /* 2-way split bf16, 3 MFMA products, fp32-class accuracy     */
  NA_PREC_F16 = 2,    /* f16 operands, fp32 accumulate (1 MFMA product, 11-bit operands): the layer-synchronous
                         renderers (na_render_*_ls) and na_mlp_pack / na_mlp_forward; not na_render_plain_view */
  NA_PREC_F16X = 3    /* f16 main product + two MX-fp6 (e2m3, E8M0 scale per 32 k) correction products
                         fp6(W - f16 W) x fp6(x) + fp6(W) x fp6(x - f16 x) on v_mfma_scale_f32_32x32x64_f8f6f4: 1.5 MFMA
                         products per k, ~15-bit operands (init / geometry chunks: f16 hi + lo, 3 products).
                         The layer-synchronous renderers only (na_render_*_ls_pack / na_render_*_ls) */
};
/* weight-stream layouts */
enum { NA_LAYOUT_GENERIC = 0, NA_LAYOUT_PLAIN_FIRST = 1, NA_LAYOUT_PLAIN_VIEW = 2 };
/* density -> sigma (src/nerf.py:64-65) */
enum { NA_DENSITY_SOFTPLUS_M1 = 0, NA_DENSITY_RELU = 1 };
/* background (src/nerf.py:96-109) */
enum { NA_BG_BLACK = 0, NA_BG_WHITE = 1, NA_BG_RANDOM = 2 /* only through the *_random_bg entry points: needs the draw */ };
/* colour-head activations, src/utils.py:484-518 sigmoid_kinds */
enum {
  NA_SIG_NORMAL = 0, NA_SIG_THIN = 1, NA_SIG_FAT = 2, NA_SIG_TANH = 3, NA_SIG_UPSHIFTED = 4,
  NA_SIG_RELU = 5, NA_SIG_SIN = 6, NA_SIG_LEAKY_RELU = 7, NA_SIG_UPSHIFTED_SOFTPLUS = 8,
  NA_SIG_UPSHIFTED_RELU = 9, NA_SIG_CYCLIC = 10, NA_SIG_IDENTITY = 11
};

int na_version(void);
const char* na_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * A1+A2  pixel grid + NeRFCamera.sample_positions   (runner.py:490-503, src/cameras.py:45-66)
 * rays[b, r, c, :] for the crop rows crop_t..crop_t+crop_h, cols crop_l..crop_l+crop_w of a
 * size x size image (the crop is clipped to the image like the reference's slicing, so the
 * caller passes the clipped h,w).  noise: NULL, or [h,w,2] uniform[0,1) draws (u then v)
 * used as (noise-0.5)*with_noise  (replaces the reference's global-RNG rand_like).
 * rays: [B, h, w, 6] = origin | un-normalised direction.                                      */
int na_raygen(const float* c2w /*[B,3,4]*/, int B, float focal, int size,
              int crop_t, int crop_l, int crop_h, int crop_w,
              const float* noise, float with_noise, float* rays, void* stream);

/* A2' DTUCamera.sample_positions + lift  (src/cameras.py:159-223); rays [B, h, w, 6] with
 * normalised directions; pose/intrinsic [B,4,4].                                              */
int na_raygen_dtu(const float* pose, const float* intrinsic, int B, int size,
                  int crop_t, int crop_l, int crop_h, int crop_w, float* rays, void* stream);

/* A3 compute_ts (src/nerf.py:29-47).  rand: NULL or device [T] uniform draws (perturb>0).
 * ts: [T]; mids: NULL or [T-1].                                                               */
int na_compute_ts(float near, float far, int T, int lindisp, float perturb, const float* rand,
                  float* ts, float* mids, void* stream);

/* A3 compute_pts_ts (src/nerf.py:50-55): pts[t, r, :] = r_o[r] + ts[t] * r_d[r].
 * rays [R,6]; pts [T,R,3].                                                                    */
int na_compute_pts(const float* rays, const float* ts, int T, int64_t R, float* pts, void* stream);

/* ---------------------------------------------------------------------------------------------
 * A5 HashEncoder.forward (src/neural_blocks.py:139-193).  x [N,3]; tables [8,65536,4] fp32;
 * out [N, 32 + 3*include_input]; idx_out: NULL or int64 [8 levels, 8 corners, N] table rows
 * (bit-exact int64 hashing).                                                                  
 * `out` must be 16-byte aligned (the rows are written with 16-byte stores; NA_EINVAL otherwise). */
int na_hash_encode(const float* x, int64_t N, const float* tables, int include_input,
                   float* out, int64_t* idx_out, void* stream);

/* A5' FourierEncoder.forward / fourier (src/neural_blocks.py:52, src/utils.py:14-17).
 * basis [D,F] (already multiplied by extra_scale by the caller or pass scale); out [N,2F].    */
int na_fourier_encode(const float* x, int64_t N, int D, const float* basis, int F, float scale,
                      float* out, void* stream);

/* PositionalEncoder.forward (src/neural_blocks.py:30-34): out [N, 2*D*NB].                     */
int na_positional_encode(const float* x, int64_t N, int D, const float* bands, int NB,
                         float* out, void* stream);

/* A7 dir_to_elev_azim (src/utils.py:247-254): dirs [N,3] -> out [N,2].                         */
int na_view_elaz(const float* dirs, int64_t N, float* out, void* stream);
/* rows [x, y, z, elev, azim] [N, 5] of the View reflectance's input: pts [N = T x R, 3] with the per-RAY directions dirs [R, 3]
 * broadcast along the samples (n = t R + r) -- src/refl.py:190-207's cat([x, dir_to_elev_azim(view)]) in one launch. */
int na_view_rows(const float* pts, const float* dirs, int64_t N, int64_t R, float* out, void* stream);
/* training step, the torch glue around the two networks of PlainNeRF(view) as kernels (round 6):
 * na_hash_encode_rows   out [N, 32 + 3 (include_input + lead)] = [x (lead = 1) | x (include_input) | features]: with lead = 1 the
 *                       init rows cat([p, enc(p)]) of a hash-encoded SkipConnMLP (src/neural_blocks.py:139-193, 283-287) written by
 *                       the encoder itself.
 * na_plain_head_rows    PlainNeRF.from_pts between its networks (src/nerf.py:338-357, src/refl.py:190-207): first_out [N, 1 + C] ->
 *                       density [N] = first_out[:, 0] and the View MLP's init rows [N, 5 + C] = [x, y, z, elev, azim | first_out[:, 1:]]
 *                       (pts [N = T x R, 3], per-RAY directions dirs [R, 3], n = t R + r).                                            */
int na_hash_encode_rows(const float* x, int64_t N, const float* tables, int include_input, int lead, float* out, void* stream);
int na_plain_head_rows(const float* first_out, const float* pts, const float* dirs, int64_t N, int64_t R, int C, float* density,
                       float* rows, void* stream);

/* sigmoid_kinds (src/utils.py:484-518) elementwise, in place allowed.                          */
int na_sigmoid(const float* x, int64_t N, int kind, float* out, void* stream);

/* A6 mip integrated positional encoding, intended layout: rays [B*H*W,6] of an
 * H x W crop (radii_x differences rows), ts [T]; kind 0 = cylinder, 1 = cone; t_end = upper
 * bound of the last interval (NaN: the kernel uses 2 ts[T-1] - ts[T-2], the intended closing, so
 * the host need not read ts back); out [T, B*H*W, 6*(max_deg-min_deg)].                      */
int na_mip_encode(const float* rays, int B, int H, int W, const float* ts, int T, int kind,
                  float t_end, int min_deg, int max_deg, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * A8 alpha_from_density + volumetric_integrate + sky (src/nerf.py:22-27,60-80,96-109).
 * density [T,R]; feat [T,R,C]; ts [T]; rays [R,6] (direction norm scales the intervals, Q1);
 * alpha/weights: NULL or [T,R]; out [R,C].                                                    */
int na_composite(const float* density, const float* feat, const float* ts, const float* rays,
                 int T, int64_t R, int C, int density_kind, int bg_kind,
                 float* alpha, float* weights, float* out, void* stream);

/* The same with `--bg random` (src/nerf.py:99-103 random_color): out += rand[r] * (1 - sum(weights[:-1])), ONE uniform draw
 * per ray broadcast over the channels.  The draw is an explicit input rand [R] (the reference calls torch.rand_like inside).
 * na_sky_random adds that term to the out [R,C] of a renderer that composited against black and kept weights [T,R]
 * (the fused one-kernel renderers).                                                              */
int na_composite_random_bg(const float* density, const float* feat, const float* ts, const float* rays,
                           int T, int64_t R, int C, int density_kind, const float* rand,
                           float* alpha, float* weights, float* out, void* stream);
int na_sky_random(const float* weights, const float* rand, int T, int64_t R, int C, float* out, void* stream);

/* volumetric_integrate(weights, other) alone (src/nerf.py:79-80; depth/flow maps,
 * runner.py:894-920): weights [T,R], other [T,R,C] -> out [R,C].                              */
int na_integrate(const float* weights, const float* other, int T, int64_t R, int C, float* out,
                 void* stream);

/* PosLinearView pieces (src/refl.py:248-290): unit view directions (F.normalize, eps 1e-12), and the final
 * out[N,C] = (sigmoid(lin[N]) / 2 + 0.5) * pos[N rows of pitch pos_ld, first C columns].                            */
int na_normalize3(const float* v, int64_t N, float* out, void* stream);
int na_pos_linear_combine(const float* lin, const float* pos, int64_t pos_ld, int64_t N, int C, float* out,
                          void* stream);

/* A12 VolSDF density = 1/beta * laplace_cdf(-sdf, beta) (src/utils.py:50-58, src/nerf.py:1000-1003);
 * beta is a device scalar (learned parameter).                                                */
int na_laplace_density(const float* sdf, int64_t N, const float* beta, float* density, void* stream);

/* A11 spline warp (src/nerf.py:1173-1178,1201-1206,1267-1278): est [N, 1+3n] = rigidity | n
 * control points; t [N] (or per-view broadcast via t_stride 0... caller expands); pts [N,3].
 * out_pts = pts + bezier(t) * sigmoid(rigidity/2); optional dp, rigidity_out [N,3],[N].       */
int na_bezier_warp(const float* est, int est_stride, const float* pts, const float* t, int64_t N,
                   int n_ctrl, float* out_pts, float* dp, float* rigidity_out, void* stream);
/* the same with DynamicNeRF's reflectance latent (--dyn-refl-latent n_rl > 0; src/nerf.py:1246-1248, 1272-1278):
 * est [N, >= 2 + (3 + n_rl) n] = rigidity | n control points | enc_rigidity | n latent control rows of n_rl;
 * refl_latent [N, n_rl] = bezier(latent control rows, t) * sigmoid(enc_rigidity) -- what from_pts receives as
 * `refl_latent` (src/nerf.py:1303).  n_rl 1..16.                                                               */
int na_bezier_warp_latent(const float* est, int est_stride, const float* pts, const float* t, int64_t N,
                          int n_ctrl, int n_rl, float* out_pts, float* dp, float* rigidity_out,
                          float* refl_latent, void* stream);

/* ---------------------------------------------------------------------------------------------
 * A4 SkipConnMLP (src/neural_blocks.py:204-296).
 *
 * Exact-fp32 Linear used for any layer shape: y[N,out] = W[out,in] . act(x[N,in]) + b
 * (f32 MFMA, v_mfma_f32_32x32x2_f32 -- bitwise an fp32 fma chain).  x may be the concatenation
 * of two buffers (skip connection): x = [x0 (in0 cols) | x1 (in1 cols)].                      */
int na_linear_f32(const float* x0, int in0, const float* x1, int in1, int64_t N,
                  const float* W, const float* b, int out, int pre_act, float* y, void* stream);

typedef struct NaMlpDesc {
  int32_t in_size;     /* raw input width (p)                                         */
  int32_t enc_kind;    /* NA_ENC_*                                                    */
  int32_t enc_dims;    /* encoder output width (35 hash, 2F fourier, 0 none)          */
  int32_t latent_size; /* extra per-sample latent columns                             */
  int32_t num_layers;  /* hidden Linear count L (src/neural_blocks.py:240-244)        */
  int32_t hidden;      /* hidden width; the MFMA path requires 256                    */
  int32_t out_size;
  int32_t skip;        /* 3                                                           */
  int32_t activation;  /* NA_ACT_LEAKY_RELU | NA_ACT_SIN                              */
  int32_t layout;      /* NA_LAYOUT_*: GENERIC for na_mlp_forward; the PLAIN_* values
                          re-order rows/columns for na_render_plain_view               */
} NaMlpDesc;

/* Fused MFMA SkipConnMLP: weights are re-laid out once into the MFMA A-fragment stream
 * (bf16 hi [+ lo]) that the kernel DMA-streams through LDS.
 *   na_mlp_packed_bytes : size of that stream for (desc, precision); 0 if unsupported.
 *   na_mlp_pack         : device-side pack.  weights/biases: HOST arrays of L+2 DEVICE pointers
 *                         in the order init, layers[0..L-1], out  (nn.Linear layout [out,in]).
 *   na_mlp_forward      : p [N,in_size]; latent [N,latent_size] or NULL; enc_params: hash tables
 *                         [8,65536,4] or Fourier basis [D,F] or NULL; y [N,out_size].          */
size_t na_mlp_packed_bytes(const NaMlpDesc* desc, int precision);
int na_mlp_pack(const NaMlpDesc* desc, int precision, const float* const* weights,
                const float* const* biases, void* packed, void* stream);
int na_mlp_forward(const NaMlpDesc* desc, int precision, const void* packed,
                   const float* p, const float* latent, const float* enc_params, int64_t N,
                   float* y, void* stream);
/* Same with explicit row pitches (in floats) for p and latent: the inputs may be column slices of wider buffers, e.g.
 * latent = first_out + 1 with pitch 65 (src/nerf.py:349-352 `intermediate = first_out[..., 1:]`) -- no copy.          */
int na_mlp_forward_ld(const NaMlpDesc* desc, int precision, const void* packed,
                      const float* p, int64_t p_ld, const float* latent, int64_t latent_ld,
                      const float* enc_params, int64_t N, float* y, void* stream);
/* IPE latent generated in the MLP prologue (config 3: src/utils.py:83-140 cylinder / conic Gaussians, hook
 * src/nerf.py:256-261).  `mip` describes the crop the N = T*B*H*W samples come from (sample n = t*B*H*W + ray); the
 * leading 6*(max_deg-min_deg) latent columns are computed from it inside the kernel (same arithmetic as na_mip_encode,
 * intended layout), the remaining latent_size - 6 nd columns are read from `latent` (pitch latent_ld) as usual.
 * mip == NULL is na_mlp_forward_ld.                                                                               */
typedef struct NaMipDesc {
  const float* rays;   /* [B,H,W,6] rays of ONE crop (pixel radii difference neighbouring rows)   */
  const float* ts;     /* [T]                                                                     */
  int32_t B, H, W, T;
  int32_t kind;        /* 0 cylinder, 1 cone                                                      */
  int32_t min_deg, max_deg;
  float t_end;         /* closes the last interval (see na_mip_encode); NaN = 2 ts[T-1] - ts[T-2] */
} NaMipDesc;
int na_mlp_forward_mip(const NaMlpDesc* desc, int precision, const void* packed, const float* p, int64_t p_ld,
                       const float* latent, int64_t latent_ld, const float* enc_params, const NaMipDesc* mip, int64_t N,
                       float* y, void* stream);

/* ---------------------------------------------------------------------------------------------
 * A10 PlainNeRF.forward with the View head (src/nerf.py:326-361, src/refl.py:190-207), fully
 * fused: sample -> hash encode -> first MLP -> elaz -> View MLP -> sigmoid -> composite.
 *   rays [R,6]; ts [T]; hash tables [8,65536,4]; packed_first / packed_view from na_mlp_pack
 *   (descs: first = {3,HASH,35,0,4,256,1+64,3,LEAKY,PLAIN_FIRST}, view = {5,NONE,0,64,4,256,3,3,SIN,PLAIN_VIEW}).
 *   workspace: na_render_workspace_bytes(T,R) bytes of device scratch (per-(ray,block) partials).
 *   alpha/weights: NULL or [T,R].  out [R,3].                                                 */
size_t na_render_workspace_bytes(int T, int64_t R);
int na_render_plain_view(const float* rays, int64_t R, const float* ts, int T,
                         const float* hash_tables, const void* packed_first, const void* packed_view,
                         int precision, int sigmoid_kind, int bg_kind,
                         float* alpha, float* weights, float* out,
                         void* workspace, size_t workspace_bytes, void* stream);
/* Same kernel with explicit sample positions pts[T,R,3] (PlainNeRF.from_pts, src/nerf.py:340-361, as called by
 * DynamicNeRF with the spline-warped canonical points, src/nerf.py:1301-1303); directions and interval lengths still
 * come from rays / ts.                                                                                              */
int na_render_plain_view_pts(const float* rays, const float* pts, int64_t R, const float* ts, int T,
                             const float* hash_tables, const void* packed_first, const void* packed_view,
                             int precision, int sigmoid_kind, int bg_kind, float* alpha, float* weights, float* out,
                             void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Backward operators (training step, runner.py:609-850: loss.backward() differentiates exactly these).
 * fp32; every gradient buffer that is ACCUMULATED into (dW, db, tables_grad) must be zero-initialised
 * (or hold the running .grad) by the caller.
 *
 * na_act_backward      g_x = g_act * act'(x) for the pre-activation input x of a Linear
 *                      (src/neural_blocks.py:293).
 * na_sigmoid_backward  g_x = g_y * d/dx sigmoid_kind(x) (src/utils.py:484-518).
 * na_pos_linear_combine_backward  gradient of na_pos_linear_combine (src/refl.py:288-290) w.r.t. lin [N] (g_lin,
 *                      nullable) and pos (g_pos [N, gpos_ld], nullable: columns < C receive g * (sigmoid/2 + 0.5),
 *                      columns C..gpos_ld-1 are zeroed) given g [N, C].
 * na_linear_wgrad      dW[out,in] += dY^T . act([x0|x1]);  db[out] += sum_n dY (db may be NULL), exact fp32
 *                      (the exact input gradient is na_linear_f32 with W^T followed by na_act_backward).
 * Fast training precision: the three GEMMs of a layer on the bf16 matrix core with a 2-way operand split
 * (hi = bf16(x), lo = bf16(x - hi); lo*hi + hi*lo + hi*hi accumulated in fp32: relative error ~2^-16):
 * na_linear_bf16x3        same contract as na_linear_f32 (y = W . act([x0|x1]) + b).
 * na_linear_dgrad_bf16x3  g_x0[N,in0] | g_x1[N,in1] = (dY[N,out] . W) * act'([x0|x1]); Wt = W^T as
 *                         [in0+in1, out] row-major; either output may be NULL; x0/x1 = the forward inputs.
 * na_linear_wgrad_bf16x3  same contract as na_linear_wgrad.
 * na_linear_wgrad_bf16x3_ow  the same gradient WRITTEN to dW / db instead of accumulated (no zero fill by the caller).
 * Round 5 -- one pack launch per training step instead of one per GEMM (the reference's loop, runner.py:647-825, re-reads
 * nn.Linear.weight in every F.linear; here the bf16 hi / lo MFMA fragments of every Linear are built once per step):
 * na_train_packed_bytes   bytes of the packed form of an [M, K] B operand (0: no packed form, M > 1024).
 * na_train_pack_many      n operands in ONE launch: operand i is the M[i] x K[i] matrix B[m,k] = W[i][m * ld[i] + k], or with
 *                         transposed[i] B[m,k] = W[i][k * ld[i] + m] (the input gradient's W^T read straight from
 *                         nn.Linear.weight: no transposing copy); dst[i]: na_train_packed_bytes(M[i], K[i]) bytes, 16-byte aligned.
 * na_train_gemm_packed_ok 1 if a batch of N rows with M output columns runs the kernels that take packed operands
 *                         (N >= 2048, M <= 1024, NA_TRAIN_GEMM != tiled), else 0: call the unpacked entry points.
 * na_linear_bf16x3_pk / na_linear_dgrad_bf16x3_pk   na_linear_bf16x3 / na_linear_dgrad_bf16x3 with the packed W [out, in0+in1] /
 *                         packed W^T [in0+in1, out] in place of the matrix (NA_EUNSUPPORTED where ..._packed_ok says 0).
 * na_hash_encode_backward  tables_grad[8,65536,4] += trilinear weights x g_out[N, 32(+3)]
 *                      (src/neural_blocks.py:166-190).
 * na_hash_encode_backward_input  g_x[N,3] = d(features)/d(position) . g_out (+ g_out[:, :3] with
 *                      include_input); floor() carries no gradient, as in torch.autograd of the same lines.
 * na_laplace_density_backward    g_sdf[N] and (optional, ACCUMULATED into one float the caller zeroed) g_beta of
 *                      na_laplace_density (src/utils.py:50-58, src/nerf.py:985-990).
 * na_bezier_warp_backward        g_est[N, est_stride] of na_bezier_warp given any of g_out_pts [N,3], g_dp [N,3],
 *                      g_rigidity [N] (null = zero) (src/nerf.py:1173-1178,1201-1206,1267-1278).
 * na_composite_backward    gradient of na_composite w.r.t. density [T,R] and feat [T,R,C] (C = 1 or 3)
 *                      given g_out [R,C] (src/nerf.py:60-80,96-98).                              */
 /* Deterministic accumulation.  The gradients summed across workgroups (dW/db of the two na_linear_wgrad* kernels,
 * the hash-table scatter, d/dbeta of the Laplace density) use fp32 atomics by default: fast, but the rounding depends
 * on the arrival order, so two runs differ in the last bits (and a chaotic training trajectory such as D-NeRF's
 * diverges from there).  After na_set_deterministic(ws, bytes) -- a caller-owned device buffer of at least
 * 8 * (largest accumulated output) bytes = 16 MiB for the hash tables -- those operators accumulate in 64-bit fixed
 * point (2^-40 resolution): integer addition is associative, so results are bitwise reproducible.  The setting is
 * process-wide (autograd engines call the backward operators from their own threads); the workspace is used by one
 * operator at a time, in stream order, so all deterministic work must share one stream.  NULL switches it off.       */
int na_set_deterministic(void* workspace, size_t bytes);

/* Forward-mode tangents for SDF normals and the eikonal regulariser (src/sdf.py:43,108, runner.py:685-692,
 * src/utils.py:31): d sdf / d x is propagated forward through the MLP as three tangent rows per sample,
 * t_{l+1} = W_l . (act'(z_l) * t_l), so the loss on the normals is a first-order graph of the operators below plus
 * na_linear_*; no double backward is needed.
 * na_act_deriv          out = act'(x) (order 1) or act''(x) (order 2) for NA_ACT_* (leaky: 0.01|1, 0; sin: cos, -sin).
 * na_mul_bcast          out[j,i] = a[i] * b[j,i], a [n], b/out [J,n]: one multiplier row shared by J tangent rows.
 * na_mul_reduce         out[i] = sum_j g[j,i] * b[j,i] (gradient of na_mul_bcast w.r.t. a).
 * na_eikonal_loss       loss[0] += mean_n (|normals[:,n]| - 1)^2, normals [3,N] tangent-major; loss zeroed by caller.
 * na_eikonal_loss_backward  g_normals [3,N] = g[0] * d loss / d normals.                                          */
int na_act_deriv(const float* x, int64_t n, int act, int order, float* out, void* stream);
int na_mul_bcast(const float* a, const float* b, int64_t n, int J, float* out, void* stream);
int na_mul_reduce(const float* g, const float* b, int64_t n, int J, float* out, void* stream);
int na_eikonal_loss(const float* normals, int64_t N, float* loss, void* stream);
int na_eikonal_loss_backward(const float* normals, int64_t N, const float* g, float* g_normals, void* stream);
int na_act_backward(const float* x, const float* g, int64_t n, int act, float* out, void* stream);
int na_sigmoid_backward(const float* x, const float* g, int64_t n, int kind, float* out, void* stream);
int na_pos_linear_combine_backward(const float* lin, const float* pos, int64_t pos_ld, const float* g, int64_t N, int C,
                                   float* g_lin, float* g_pos, int64_t gpos_ld, void* stream);
int na_linear_bf16x3(const float* x0, int in0, const float* x1, int in1, int64_t N, const float* W, const float* b,
                     int out, int pre_act, float* y, void* stream);
int na_linear_dgrad_bf16x3(const float* dY, int out, int64_t N, const float* Wt, const float* x0, int in0,
                           const float* x1, int in1, int pre_act, float* g_x0, float* g_x1, void* stream);
int na_linear_wgrad_bf16x3(const float* x0, int in0, const float* x1, int in1, int64_t N, const float* dY, int out,
                           int pre_act, float* dW, float* db, void* stream);
int na_linear_wgrad_bf16x3_ow(const float* x0, int in0, const float* x1, int in1, int64_t N, const float* dY, int out,
                              int pre_act, float* dW, float* db, void* stream);
size_t na_train_packed_bytes(int M, int K);
int na_train_gemm_packed_ok(int64_t N, int M);
int na_train_pack_many(int n, const float* const* W, const int* M, const int* K, const int* ld, const int* transposed,
                       void* const* dst, void* stream);
int na_linear_bf16x3_pk(const float* x0, int in0, const float* x1, int in1, int64_t N, const void* w_packed, const float* b,
                        int out, int pre_act, float* y, void* stream);
int na_linear_dgrad_bf16x3_pk(const float* dY, int out, int64_t N, const void* wt_packed, const float* x0, int in0,
                              const float* x1, int in1, int pre_act, float* g_x0, float* g_x1, void* stream);
/* Round 5: input gradient AND weight gradient of one Linear in ONE pass over dY and the forward input (they read the same two
 * [N, .] tensors; src/neural_blocks.py:288-296 differentiated, runner.py:647-825's loss.backward()):
 *   g_x0[N,in0] = (dY . W[:, c0:c0+in0]) * act'(x0);  dW[out, 0:in0] (leading dimension ldw >= in0) and db[out] (if given) WRITTEN.
 * in0 = 256 (a wide source) or in0 <= 128 (a narrow one: an init Linear's 38 / 69 columns, the second source of a skip layer);
 * out <= 256; wt_packed = W^T [in, out] as na_train_pack_many packs it, at the column group of the source's first row c0 (a
 * multiple of 64): na_train_packed_row_offset(c0, out) bytes into the packed stream.  The sources of a concatenated input are
 * separate calls (dW + c0 with the full leading dimension).  na_linear_bwd_fused_ok: 1 if (N, out, in0) runs this kernel
 * (N >= 8192, NA_TRAIN_FUSED_BWD != 0), else the entry point returns NA_EUNSUPPORTED. */
size_t na_train_packed_row_offset(int row0, int K);
int na_linear_bwd_fused_ok(int64_t N, int out, int in0);
/* na_linear_wgrad_bf16x3_cols: the weight gradient of ONE source of a concatenated input: dW[:, 0:in) at leading dimension ldw and
 * (if given) db WRITTEN -- the narrow source [N, 38 / 69] of a skip layer whose wide source went through na_linear_bwd_bf16x3_pk. */
int na_linear_wgrad_bf16x3_cols(const float* x, int in, int64_t N, const float* dY, int out, int pre_act, float* dW, int ldw,
                                float* db, void* stream);
/* workspace: na_linear_bwd_workspace_bytes(N, in0) bytes of device memory for the partial gradients (16-byte aligned; the caller's
 * allocator -- torch's is stream-ordered and costs microseconds), or NULL: the call then takes them from the library's grow-only
 * scratch of this (device, stream) -- no driver call once it exists; it grows geometrically and a replaced buffer is freed when
 * the work queued before its replacement has finished. */
size_t na_linear_bwd_workspace_bytes(int64_t N, int in0);
int na_linear_bwd_partial_count(int64_t N, int in0);   /* partial gradients in that workspace (the nwg_i of na_train_reduce_many) */
/* The pass without its reduction (the whole-network backward of a SkipConnMLP: every Linear's partial gradients stay in a
 * workspace of its own -- na_linear_bwd_workspace_bytes(N, in0) bytes = slices x 67 584 floats -- and ONE na_train_reduce_many
 * sums them all: dW_i[out_i, 0:in_i) at leading dimension ldw_i and db_i (nullable) WRITTEN from nwg_i = slices partials).
 * g_add (nullable, [N, in0]; narrow sources, in0 <= 128, only): added to the input gradient before it is stored. */
int na_linear_bwd_partials_bf16x3_pk(const float* dY, int out, int64_t N, const void* wt_packed, const float* x0, int in0, int pre_act,
                                     float* g_x0, const float* g_add, int want_db, void* workspace, void* stream);
int na_train_reduce_many(int n, const float* const* part, const int* nwg, const int* out, const int* in, const int* ldw, float* const* dW,
                         float* const* db, void* stream);
int na_linear_bwd_bf16x3_pk(const float* dY, int out, int64_t N, const void* wt_packed, const float* x0, int in0, int pre_act,
                            float* g_x0, float* dW, int ldw, float* db, void* workspace, void* stream);
int na_linear_wgrad(const float* x0, int in0, const float* x1, int in1, int64_t N, const float* dY, int out,
                    int pre_act, float* dW, float* db, void* stream);
int na_hash_encode_backward(const float* x, int64_t N, const float* g_out, int include_input,
                            float* tables_grad, void* stream);
int na_hash_encode_backward_input(const float* x, int64_t N, const float* tables, const float* g_out,
                                  int include_input, float* g_x, void* stream);
/* the same two on a gradient read IN PLACE from wider rows (the gradient of a network's init rows [x | x | features | ...] of pitch
 * g_ld: no slice copy): _backward_rows takes the 32 feature columns at g_col0; _backward_input_rows takes rows
 * [x (lead) | x (include_input) | features] and adds the leading copy's gradient.
 * na_plain_head_rows_backward: g_first_out [N, 1 + C] = [g_density (nullable: 0) | g_rows[:, 5:]], g_pts [N, 3] = g_rows[:, :3] (nullable). */
int na_hash_encode_backward_rows(const float* x, int64_t N, const float* g_rows, int g_ld, int g_col0, float* tables_grad,
                                 void* stream);
int na_hash_encode_backward_input_rows(const float* x, int64_t N, const float* tables, const float* g_rows, int g_ld,
                                       int include_input, int lead, float* g_x, void* stream);
int na_plain_head_rows_backward(const float* g_density, const float* g_rows, int64_t N, int C, float* g_first_out, float* g_pts,
                                void* stream);
/* The optimiser step of runner.py:448-458 (optim.Adam(params, lr, eps = 1e-7); amsgrad / maximize / weight decay off) for n tensors
 * by ONE launch: per element the operations of torch's foreach Adam in its order and rounding (fma_mask: which of its three
 * multiply-adds ATen contracts -- bit 0 lerp, bit 1 addcmul, bit 2 addcdiv; nerf_atlas_amd/train.py holds the value pinned against
 * torch bit for bit).  params / grads / exp_avg / exp_avg_sq: HOST arrays of n device pointers, numel[n]; the scalars as torch's
 * Python computes them in double: 1 - beta1, beta2, 1 - beta2, sqrt(1 - beta2^t), eps, -lr / (1 - beta1^t). */
int na_adam_step(int n, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                 const int64_t* numel, double one_minus_beta1, double beta2, double one_minus_beta2, double bias_correction2_sqrt,
                 double eps, double neg_step_size, int fma_mask, void* stream);
/* Forward-mode derivative of the hash features along a per-point direction e (the deformation network's input
 * Jacobian-vector product that the FFJORD divergence estimate needs: runner.py:697-700, src/utils.py:467-478;
 * src/neural_blocks.py:166-190 is what is differentiated).
 * na_hash_encode_jvp           t_out[N, 32 (+3)] = d hash_encode(x)/dx . e   (rows [e | per-level features])
 * na_hash_encode_jvp_backward  tables_grad += d <g_t, J(x).e> / d tables (the adjoint of the above in the tables)
 * na_ffjord_div                div[N] = <e, d(rigid_dp)/dx . e> from the deformation MLP's outputs est[N, stride] =
 *                              [rigidity | n_ctrl control points | ...] and their directional derivatives est_tangent:
 *                              rigid_dp = bezier(P, t) * sigmoid(est0 / 2)  (src/nerf.py:1267-1278).               */
int na_hash_encode_jvp(const float* x, int64_t N, const float* tables, const float* tangent, int include_input,
                       float* t_out, void* stream);
int na_hash_encode_jvp_backward(const float* x, const float* tangent, int64_t N, const float* g_t, int include_input,
                                float* tables_grad, void* stream);
int na_ffjord_div(const float* est, const float* est_tangent, int est_stride, const float* t, const float* e, int64_t N,
                  int n_ctrl, float* div, void* stream);
int na_laplace_density_backward(const float* sdf, int64_t N, const float* beta, const float* g, float* g_sdf,
                                float* g_beta, void* stream);
int na_bezier_warp_backward(const float* est, int est_stride, const float* t, int64_t N, int n_ctrl,
                            const float* g_out_pts, const float* g_dp, const float* g_rigidity, float* g_est,
                            void* stream);
/* gradient of na_bezier_warp_latent: + g_refl_latent [N, n_rl] (nullable) into the enc_rigidity / latent control columns */
int na_bezier_warp_latent_backward(const float* est, int est_stride, const float* t, int64_t N, int n_ctrl, int n_rl,
                                   const float* g_out_pts, const float* g_dp, const float* g_rigidity,
                                   const float* g_refl_latent, float* g_est, void* stream);
int na_composite_backward(const float* density, const float* feat, const float* ts, const float* rays,
                          int T, int64_t R, int C, int density_kind, int bg_kind, const float* g_out,
                          float* g_density, float* g_feat, void* stream);
/* the same for na_composite_random_bg (src/nerf.py:99-103): d sky / d w_t = -rand[r] for t < T-1; rand carries no gradient */
int na_composite_random_bg_backward(const float* density, const float* feat, const float* ts, const float* rays,
                                    int T, int64_t R, int C, int density_kind, const float* rand, const float* g_out,
                                    float* g_density, float* g_feat, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Layer-synchronous fused renderer (csrc/render_ls.hip): the same operator as na_render_plain_view
 * (PlainNeRF.forward with the View head, src/nerf.py:326-361, src/refl.py:190-207) on the engine that keeps
 * activations in LDS and streams weights straight into registers.  Both MLPs are packed into ONE stream:
 *   w_first / b_first = {init, layers.0..3, out} of PlainNeRF.first  (hash encoder, 4x256, out 65; src/nerf.py:320-324)
 *   w_view  / b_view  = {init, layers.0..3, out} of View.mlp         (5 + 64 latent -> 4x256 -> 3; src/refl.py:201-204)
 * in nn.Linear layout (fp32, device pointers; biases may be NULL).  `pts` ([T,R,3], nullable) replaces o + t d by
 * explicit sample positions (PlainNeRF.from_pts with deformed points, src/nerf.py:337-361).  A ray's 32-step blocks
 * pass through one sample group in step order, so the transmittance product of src/nerf.py:22-27 is carried inside the
 * kernel: `out`, `alpha`, `weights` are final when it returns (no second launch).  Workspace: 8 B per ray (elev / azim
 * of the ray, written by a pre-kernel and read with scalar loads).  The stream's header names its precision and
 * schedule; a kernel handed a stream packed for anything else writes NaN colours instead of consuming it.          */
size_t na_render_ls_packed_bytes(int precision);
int na_render_ls_pack(int precision, const float* const* w_first, const float* const* b_first,
                      const float* const* w_view, const float* const* b_view, void* packed, void* stream);
size_t na_render_ls_workspace_bytes(int T, int64_t R);
int na_render_plain_view_ls(const float* rays, const float* pts, int64_t R, const float* ts, int T,
                            const float* hash_tables, const void* packed, int precision, int sigmoid_kind, int bg_kind,
                            float* alpha, float* weights, float* out, void* workspace, size_t workspace_bytes,
                            void* stream);

/* TinyNeRF (/root/reference/src/nerf.py:278-305 with the intended density = estim[..., 0]) on the same
 * layer-synchronous engine, one kernel per call: sample -> SkipConnMLP(3 -> 6 x 256 -> 4, skip 3, LeakyReLU; src/
 * neural_blocks.py:279-296) -> density | sigmoid_kind(colour) -> alpha compositing (src/nerf.py:22-27,60-80,96-98).
 * na_render_tiny_ls_pack takes the 8 Linears {init, layers.0..5, out} (nn.Linear layout, fp32 device pointers, biases may
 * be NULL).  Same ray / sample indexing, `pts`, alpha / weights outputs and precisions as na_render_plain_view_ls; no
 * workspace.                                                                                                         */
size_t na_render_tiny_ls_packed_bytes(int precision);
int na_render_tiny_ls_pack(int precision, const float* const* w, const float* const* b, void* packed, void* stream);
int na_render_tiny_ls(const float* rays, const float* pts, int64_t R, const float* ts, int T, const void* packed,
                      int precision, int sigmoid_kind, int bg_kind, float* alpha, float* weights, float* out,
                      void* stream);

/* The View head + alpha compositing alone on the same engine: VolSDF's second half (/root/reference/
 * src/nerf.py:981-1013 with src/refl.py:190-207 and src/utils.py:50-58).  `feat` holds one row per sample (sample n =
 * t * R + ray, row pitch feat_ld >= 65 floats): column 0 the signed distance, columns 1..64 the latent the SDF network
 * produced; the kernel turns the distance into the Laplace density with scale beta[0] (device pointer), evaluates
 * refl.View (x, elev / azim of the ray, latent -> sigmoid_kind(rgb)) and composites with the density used as it is (no
 * softplus).  na_render_view_ls_pack takes View.mlp's 6 Linears {init, layers.0..3, out}.  Workspace and `pts` as for
 * na_render_plain_view_ls.                                                                                          */
size_t na_render_view_ls_packed_bytes(int precision);
int na_render_view_ls_pack(int precision, const float* const* w, const float* const* b, void* packed, void* stream);
int na_render_view_ls(const float* rays, const float* pts, int64_t R, const float* ts, int T, const float* feat, int feat_ld,
                      const float* beta, const void* packed, int precision, int sigmoid_kind, int bg_kind, float* alpha,
                      float* weights, float* out, void* workspace, size_t workspace_bytes, void* stream);

/* VolSDF with the SIREN SDF network (/root/reference/src/sdf.py:278-287: SkipConnMLP 3 -> 5 x 256 -> 1 + 64, sin, skip 3)
 * as ONE kernel: sample -> SIREN -> Laplace density | latent -> refl.View -> compositing (src/nerf.py:981-1013).  The pack
 * call takes the SIREN's 7 Linears {init, layers.0..4, out} and View.mlp's 6; `beta` is the Laplace scale on the device.
 * Workspace, `pts`, precisions and outputs as for na_render_plain_view_ls.                                          */
size_t na_render_volsdf_siren_ls_packed_bytes(int precision);
int na_render_volsdf_siren_ls_pack(int precision, const float* const* w_sdf, const float* const* b_sdf,
                                   const float* const* w_view, const float* const* b_view, void* packed, void* stream);
int na_render_volsdf_siren_ls(const float* rays, const float* pts, int64_t R, const float* ts, int T, const float* beta,
                              const void* packed, int precision, int sigmoid_kind, int bg_kind, float* alpha, float* weights,
                              float* out, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * SDF ray marching (src/march.py).  Per-ray state lives in caller-owned device arrays; the SDF network
 * is evaluated for ALL rays by na_mlp_forward between the updates (no mask compaction, no host sync) and the updates
 * touch only the rays the reference's boolean-mask code would have touched.  `sdf` is column 0 of an [R, stride] MLP
 * output.  hits/rem/todo are 0/1 bytes.
 *
 * Compacted iterations (src/march.py:37-45, 164-179: the reference evaluates the SDF only for the rays its boolean masks keep):
 * na_compact_rays         idx[0..count) = the rays with live[r] != 0, ASCENDING (deterministic); *count = their number (device
 *                         int32).  idx has room for R + 256 entries (the tail is scratch).
 * na_ray_points_indexed / na_sphere_march_update_indexed / na_bisection_update_indexed: the kernels below on n compacted
 *                         rows -- row i of pts / sdf belongs to ray idx[i]; per-ray state arrays stay full size.
 * na_ray_points           pts[R,3] = r_o + r_d * t, t per ray (t_ray[R]) or t_scalar when t_ray is NULL.
 * na_sphere_march_update  one iteration of sphere_march (src/march.py:39-45): for rem rays  hits |= (sdf < eps) &
 *                         (dist <= far);  dist += sdf;  rem &= !(hits | dist > far).
 * na_sign_change_update   one uniform step `step` of throughput_with_sign_change (src/march.py:96-103): running
 *                         minimum + its step index, first sign change (last_pos / first_neg step indices, -1 = none).
 * na_bisection_update     bisection (src/march.py:159-179): sdf_mid NULL initialises todo and z = (low+high)/2 from
 *                         low/high/sdf_low/sdf_high; otherwise one iteration with the SDF at z.                        */
int na_compact_rays(const uint8_t* live, int64_t R, int32_t* idx, int32_t* count, void* stream);
int na_ray_points_indexed(const float* r_o, const float* r_d, const float* t_ray, const int32_t* idx, int64_t n, float* pts,
                          void* stream);
int na_sphere_march_update_indexed(const float* sdf, int stride, const int32_t* idx, int64_t n, float eps, float far,
                                   float* dist, uint8_t* hits, uint8_t* rem, void* stream);
int na_bisection_update_indexed(const float* sdf_mid, int stride, const int32_t* idx, int64_t n, float eps, float* low,
                                float* high, float* sdf_low, float* sdf_high, float* z, uint8_t* todo, void* stream);
int na_ray_points(const float* r_o, const float* r_d, const float* t_ray, float t_scalar, int64_t R, float* pts,
                  void* stream);
int na_sphere_march_update(const float* sdf, int stride, int64_t R, float eps, float far, float* dist, uint8_t* hits,
                           uint8_t* rem, void* stream);
int na_sign_change_update(const float* sdf, int stride, int64_t R, int step, float* curr_min, int32_t* idxs,
                          int32_t* last_pos, int32_t* first_neg, void* stream);
int na_bisection_update(const float* sdf_mid, int stride, int64_t R, float eps, float* low, float* high, float* sdf_low,
                        float* sdf_high, float* z, uint8_t* todo, void* stream);

/* ---- N4: lights and occlusion (src/lights.py:69-132 Point; src/renderers.py:29-163 occlusion kinds) ----------------
 * na_point_light      Point.forward for N points: dir[N,3] = normalize(center - x) (eps 1e-6), dist[N] = |center - x|,
 *                     spectrum[N,3] = intensity / (4 pi dist^2) (or intensity when distance_decay == 0).  center and
 *                     intensity are one vector (stride 0) or one per point (stride 3: `loc.expand(mask.shape)[mask]`).
 * na_occlusion_apply  out = spectrum * a * v:  att_mode 0 a = 1; 1 a = sigmoid(raw_att) on hidden points only
 *                     (LearnedLighting :65-67); 2 a = sigmoid(raw_att) + 1e-2 (AllLearnedOcc :111-121);  v = 1 where
 *                     visible (or visible == NULL), hidden_value elsewhere (0: LightingWIsect :40-45; sigmoid(alpha):
 *                     LearnedConstantSoftLighting :82-83, JointLearnedConstOcc :143-146).                           */
int na_point_light(const float* x, const float* center, int center_stride, const float* intensity, int intensity_stride,
                   int distance_decay, int64_t N, float* dir, float* dist, float* spectrum, void* stream);
int na_occlusion_apply(const float* spectrum, const uint8_t* visible, const float* raw_att, int att_mode,
                       float hidden_value, int64_t N, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * A hash-encoded SkipConnMLP alone on the layer-synchronous engine (round 4): D-NeRF's deformation network
 * (src/nerf.py:1250-1257 delta_estim = SkipConnMLP(in 3, HashEncoder, 5 x 256, skip 3, out 3 n + 1), evaluated at
 * :1267-1270) at the samples of rays x ts (or explicit pts [T,R,3]); rows y[(t * R + ray) * y_ld + 0..n_out).
 * weights / biases: init, layers.0..4, out (nn.Linear layout).  NA_PREC_F16X (n_out <= 32; the output turns NaN as a whole if an
 * activation saturates the half range, see na_render_plain_view_ls) or, round 6, NA_PREC_BF16X3 -- the three-product bf16 split, the
 * accuracy class of na_mlp_forward's rows, n_out <= 64 (with `--dyn-refl-latent`, src/nerf.py:1246-1248: 3 n + 1 + n rl + 1 rows).
 * The other precisions use na_mlp_forward. */
size_t na_mlp_hash_ls_packed_bytes(int precision);
int na_mlp_hash_ls_pack(int precision, const float* const* weights, const float* const* biases, int n_out, void* packed,
                        void* stream);
int na_mlp_hash_ls(const float* rays, const float* pts, int64_t R, const float* ts, int T, const float* hash_tables,
                   const void* packed, int precision, int n_out, float* y, int64_t y_ld, void* stream);

/* The same for a Fourier-encoded SkipConnMLP: VolSDF's MLP SDF network (src/sdf.py:250-258: SkipConnMLP(in 3, FourierEncoder
 * with 128 frequencies, 6 x 256, skip 3, out 1 + 64), evaluated at src/sdf.py:109-112).  weights / biases: init, layers.0..5,
 * out; basis [3,128] fp32, 16-byte aligned, any extra_scale already multiplied in; rows y[(t * R + ray) * y_ld + 0..65) =
 * (signed distance | 64 latent columns), the layout na_render_view_ls takes as `feat`.  The 256 Fourier features are generated
 * inside the kernel wherever a Linear consumes them; they never exist in HBM.  NA_PREC_F16X only.                         */
size_t na_mlp_fourier_ls_packed_bytes(int precision);
int na_mlp_fourier_ls_pack(int precision, const float* const* weights, const float* const* biases, void* packed, void* stream);
int na_mlp_fourier_ls(const float* rays, const float* pts, int64_t R, const float* ts, int T, const float* basis,
                      const void* packed, int precision, float* y, int64_t y_ld, void* stream);

/* Coarse -> fine rendering (BASELINE config 2 "64 + 128"; the reference's sample_pdf, src/nerf.py:1745-1779 with its call site
 * :572-578, is dead code -- INTENDED reading, pinned against an fp64 restatement only: see csrc/basic_ops.hip).
 * na_resample_ts: ts [T] shared coarse steps, weights [T,R] of the coarse pass (rows 0..T-2 are used: weights[:-1]),
 * u NULL (linspace(0,1,N), the reference's `uniform`) or [N,R] draws in [0,1); fine [R,N] (nullable) = the N inverse-cdf
 * positions per ray; merged [R,T+N] (nullable) = coarse and new positions of a ray in increasing order (stable).
 * na_render_plain_view_ls_rayts: na_render_plain_view_ls with per-ray steps ts_ray [R,T] (e.g. `merged`).              */
size_t na_resample_ts_lds_bytes(int T, int N, int with_u);
int na_resample_ts(const float* ts, const float* weights, int64_t R, int T, const float* u, int N, float* fine, float* merged,
                   void* stream);
int na_render_plain_view_ls_rayts(const float* rays, int64_t R, const float* ts_ray, int T, const float* hash_tables,
                                  const void* packed, int precision, int sigmoid_kind, int bg_kind, float* alpha,
                                  float* weights, float* out, void* workspace, size_t workspace_bytes, void* stream);

/* The training iteration's FORWARD of PlainNeRF(view) as ONE launch (round 6; runner.py:647-825 drives src/nerf.py:326-361, whose two
 * SkipConnMLPs are src/neural_blocks.py:279-296): na_render_plain_view_ls's kernel in NA_PREC_BF16X3 -- the three-product bf16 split
 * of the training GEMMs -- with explicit sample positions pts [T,R,3], which also leaves in HBM what the backward pass reads: the output
 * rows (bias added, BEFORE the next layer's activation: what na_linear_bwd_partials takes as the next Linear's forward input) of the ten
 * 256-wide Linears, planes[(p * T * R + t * R + ray) * 256 + c], p = 0..4 `first`.init, layers.0..3; 5..9 the View MLP's;
 * view_rows [T*R, 69] = the View MLP's init rows [x, y, z, elev, azim | intermediate] (src/nerf.py:338-357, src/refl.py:190-207: what
 * na_plain_head_rows builds from `first`'s output, which is not materialised), density [T*R] (first.out's column 0) and rgb_pre [T*R, 3]
 * (before the sigmoid).  The layer-by-layer forward (na_linear_bf16x3_pk per Linear) writes every one of those rows AND reads it back as
 * the next layer's input; here the activations stay in LDS and the rows leave as whole 128-byte lines, non-temporal.  packed: na_render_ls_pack(NA_PREC_BF16X3) of the CURRENT weights; out [R,3] receives the kernel's own
 * composited colour (black background; callers that composite with noise or a random background ignore it); workspace:
 * na_render_ls_workspace_bytes.  T * R < 4 194 304 (32-bit row offsets). */
int na_train_plain_view_ls(const float* rays, const float* pts, int64_t R, const float* ts, int T, const float* hash_tables,
                           const void* packed, int sigmoid_kind, float* planes, float* view_rows, float* density, float* rgb_pre,
                           float* out, void* workspace, size_t workspace_bytes, void* stream);

/* PlainNeRF(view) with mip's integrated positional encoding (config 3: src/nerf.py:256-261 hook, :326-361 forward,
 * src/utils.py:23-27, 60-140 cylinder / conic Gaussians) as ONE launch of the layer-synchronous engine, NA_PREC_F16X only.
 * rays [B,H,W,6] of whole crops (pixel radii difference neighbouring rows: H >= 2); ts [T]; weights of `first`
 * ([256,134], [256,390], 3 x [256,256], [65,256]) and of the View MLP ([256,165], [256,421], 3 x [256,256], [3,256]) in the
 * reference's column order [p | enc(p) | IPE 96 | ...]; max_deg - min_deg must be 16.  The 96 IPE features of a sample are
 * generated in the kernel for each of the four Linears that consume them (same arithmetic as na_mip_encode, intended
 * layout; t_end as there).  Outputs, workspace (na_render_ls_workspace_bytes) and the range guard as na_render_plain_view_ls. */
size_t na_render_plain_mip_ls_packed_bytes(int precision);
int na_render_plain_mip_ls_pack(int precision, const float* const* w_first, const float* const* b_first,
                                const float* const* w_view, const float* const* b_view, void* packed, void* stream);
int na_render_plain_mip_ls(const float* rays, int B, int H, int W, const float* ts, int T, const float* hash_tables,
                           const void* packed, int precision, int mip_kind, int min_deg, int max_deg, float t_end,
                           int sigmoid_kind, int bg_kind, float* alpha, float* weights, float* out, void* workspace,
                           size_t workspace_bytes, void* stream);

/* PlainNeRF with the reference's OTHER two colour heads, each as ONE launch of the layer-synchronous engine (NA_PREC_F16X only):
 *   na_render_plain_pos_ls   `--refl-kind pos` (`make original`, makefile:8-13): PlainNeRF.from_pts (src/nerf.py:340-361) with
 *                            refl.Positional (src/refl.py:230-245) -- a second hash-encoded SkipConnMLP (5 x 256, skip 3, its OWN
 *                            HashEncoder tables, latent = the 64 intermediate rows of `first`) -> 3, act, compositing.
 *                            w_pos / b_pos: {init [256,102], layers.0 [256,358], layers.1, layers.2 [256,256], layers.3 [256,358],
 *                            layers.4 [256,256], out [3,256]}, reference column order [p 3 | x 3 + hash 32 | latent 64].
 *   na_render_plain_plv_ls   `--refl-kind pos-linear-view` (`make dnerf`, makefile:106-114): refl.PosLinearView (src/refl.py:248-290):
 *                            pos = SkipConnMLP(hash, 2 x 256) -> act -> [colour 3 | intermediate 64]; view = SkipConnMLP(in
 *                            [x | normalize(r_d)], latent [latent | intermediate], 2 x 128, sin) -> 1; colour * (sigmoid(view) / 2 + 0.5).
 *                            w_head / b_head: {pos.init [256,102+n], pos.layers.0 [256,358+n], pos.layers.1 [256,256], pos.out [67,256],
 *                            view.init [128,134+n], view.layers.0 [128,262+n], view.layers.1 [128,128], view.out [1,128]}.
 *                            n = n_rl in 0..3: DynamicNeRF's refl_latent columns (src/nerf.py:1245-1248, 1272-1278, 1303:
 *                            `--dyn-refl-latent`), rows refl_latent[T * R, rl_ld] (sample t * R + ray), appended to the latent of
 *                            both MLPs as the reference does (src/nerf.py:352-358).  sigmoid_kind: normal | thin | fat | upshifted
 *                            (the activation covers all 67 rows of `pos` inside the kernel: NA_EUNSUPPORTED for the other kinds,
 *                            which the host layer renders through the unfused operators).
 * rays / pts / ts / hash_tables (of `first`) / outputs / range guard as na_render_plain_view_ls; hash_tables_refl [8,65536,4] = the
 * head's own encoder.  workspace: na_render_head_ls_workspace_bytes(T, R) bytes (per-ray scratch + an 8-MiB L2-resident park
 * where a workgroup keeps the raw latent rows of its blocks between the Linears that consume them).                            */
size_t na_render_plain_pos_ls_packed_bytes(int precision);
size_t na_render_plain_plv_ls_packed_bytes(int precision);
size_t na_render_head_ls_workspace_bytes(int T, int64_t R);
int na_render_plain_pos_ls_pack(int precision, const float* const* w_first, const float* const* b_first,
                                const float* const* w_pos, const float* const* b_pos, void* packed, void* stream);
int na_render_plain_plv_ls_pack(int precision, const float* const* w_first, const float* const* b_first,
                                const float* const* w_head, const float* const* b_head, int n_rl, void* packed, void* stream);
int na_render_plain_pos_ls(const float* rays, const float* pts, int64_t R, const float* ts, int T, const float* hash_tables,
                           const float* hash_tables_refl, const void* packed, int precision, int sigmoid_kind, int bg_kind,
                           float* alpha, float* weights, float* out, void* workspace, size_t workspace_bytes, void* stream);
int na_render_plain_plv_ls(const float* rays, const float* pts, int64_t R, const float* ts, int T, const float* hash_tables,
                           const float* hash_tables_refl, const float* refl_latent, int64_t rl_ld, int n_rl, const void* packed,
                           int precision, int sigmoid_kind, int bg_kind, float* alpha, float* weights, float* out,
                           void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NERF_ATLAS_AMD_H */
