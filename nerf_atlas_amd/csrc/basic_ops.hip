// Memory-bound operators of the hot path (ray generation, sampling, encoders, compositing).
// Built with -ffp-contract=off: every rounding matches the reference's separate fp32 ops unless a
// fused multiply-add is written explicitly (torch's CPU linspace uses one).
#include <atomic>
#include "common.h"
#include <math.h>
#include <string.h>

namespace na {

static thread_local char g_err[512] = "";

static DetWs g_det = {nullptr, 0};  // process-wide: autograd runs backward kernels on its own threads
static int g_det_device = -1;        // device the workspace lives on
DetWs det_workspace() { return g_det; }

__global__ void det_fold_kernel(const long long* __restrict__ fix, int64_t n, float* __restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const long long v = fix[i];
    if (v != 0) out[i] = out[i] + (float)((double)v * (1.0 / 1099511627776.0));
  }
}

// zeroed int64 accumulator for n outputs, or nullptr (rc = NA_OK: mode off; rc != NA_OK: workspace too small)
long long* det_begin(size_t n, hipStream_t stream, const char* who, int* rc) {
  *rc = NA_OK;
  if (g_det.ptr == nullptr) return nullptr;
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev != g_det_device) {
    set_error("%s: the deterministic workspace lives on device %d, the current device is %d (na_set_deterministic)", who,
              g_det_device, dev);
    *rc = NA_EINVAL;
    return nullptr;
  }
  if (g_det.bytes < n * sizeof(long long)) {
    set_error("%s: deterministic workspace %zu < %zu bytes (na_set_deterministic)", who, g_det.bytes, n * sizeof(long long));
    *rc = NA_EWORKSPACE;
    return nullptr;
  }
  if (hipMemsetAsync(g_det.ptr, 0, n * sizeof(long long), stream) != hipSuccess) {
    set_error("%s: hipMemsetAsync failed", who);
    *rc = NA_EHIP;
    return nullptr;
  }
  return g_det.ptr;
}

int det_finish(const long long* fix, size_t n, float* out, hipStream_t stream, const char* who) {
  hipLaunchKernelGGL(det_fold_kernel, dim3(grid_for((int64_t)n, 256, 4096)), dim3(256), 0, stream, fix, (int64_t)n, out);
  return check_launch(who);
}

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

HashRes hash_resolutions() {
  HashRes r;
  const double scale = exp((log(16384.0) - log(16.0)) / 8.0 - 1.0);
  double p = 1.0;
  for (int i = 0; i < 8; ++i) {
    r.n[i] = (float)(16.0 * pow(scale, (double)i));
    (void)p;
  }
  return r;
}

// ------------------------------------------------------------------------------------ raygen
// runner.py:490-503 (positions[r,c] = (u=c, v=r)) + src/cameras.py:45-66.
__global__ void raygen_kernel(const float* __restrict__ c2w, int B, float focal, float half, int t0, int l0,
                              int h, int w, const float* __restrict__ noise, float with_noise,
                              float* __restrict__ rays) {
  int64_t total = (int64_t)B * h * w;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c = (int)(i % w);
    int r = (int)((i / w) % h);
    int b = (int)(i / ((int64_t)w * h));
    float u = (float)(l0 + c), v = (float)(t0 + r);
    if (noise != nullptr) {
      const float* nz = noise + ((int64_t)r * w + c) * 2;
      u = u + (nz[0] - 0.5f) * with_noise;
      v = v + (nz[1] - 0.5f) * with_noise;
    }
    float d0 = (u - half) / focal;
    float d1 = -(v - half) / focal;
    float d2 = -1.0f;
    const float* M = c2w + (int64_t)b * 12;
    float* o = rays + i * 6;
    o[0] = M[3];
    o[1] = M[7];
    o[2] = M[11];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      // torch.sum accumulates from +0.0: keep that first add so an all-(-0) row sums to +0 like the reference.
      // (The sign of a zero y decides azim = +pi vs -pi in dir_to_elev_azim for rays on the image's centre row.)
      float p0 = d0 * M[k * 4 + 0], p1 = d1 * M[k * 4 + 1], p2 = d2 * M[k * 4 + 2];
      o[3 + k] = ((0.0f + p0) + p1) + p2;
    }
  }
}

// src/cameras.py:159-223 DTUCamera (pose-matrix branch).  Output layout [B, h, w, 6] with the crop's
// first axis as "W" exactly like the reference's reshape(N, W, H, 6).
__global__ void raygen_dtu_kernel(const float* __restrict__ pose, const float* __restrict__ intr, int B, int size,
                                  int t0, int l0, int h, int w, float* __restrict__ rays) {
  int64_t total = (int64_t)B * h * w;
  const float nx = 1600.0f / (float)size, ny = 1200.0f / (float)size;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c = (int)(i % w);
    int r = (int)((i / w) % h);
    int b = (int)(i / ((int64_t)w * h));
    float u = (float)(l0 + c) * nx, v = (float)(t0 + r) * ny;
    const float* K = intr + (int64_t)b * 16;
    const float* P = pose + (int64_t)b * 16;
    float fx = K[0], fy = K[5], cx = K[2], cy = K[6], sk = K[1];
    float z = 1.0f;
    float xl = (((u - cx) + cy * sk / fy) - sk * v / fy) / fx * z;
    float yl = (v - cy) / fy * z;
    float pt[4] = {xl, yl, z, 1.0f};
    float wc[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      // torch.bmm row . column, accumulated in index order
      float acc = P[k * 4 + 0] * pt[0];
      acc = acc + P[k * 4 + 1] * pt[1];
      acc = acc + P[k * 4 + 2] * pt[2];
      acc = acc + P[k * 4 + 3] * pt[3];
      wc[k] = acc;
    }
    float ox = P[3], oy = P[7], oz = P[11];
    float dx = wc[0] - ox, dy = wc[1] - oy, dz = wc[2] - oz;
    float nrm = fmaxf(sqrtf((dx * dx + dy * dy) + dz * dz), 1e-12f);
    float* o = rays + i * 6;
    o[0] = ox; o[1] = oy; o[2] = oz;
    o[3] = dx / nrm; o[4] = dy / nrm; o[5] = dz / nrm;
  }
}

// ------------------------------------------------------------------------------------ sampling
// src/nerf.py:29-47.  torch's CPU linspace: step=(end-start)/(steps-1); first half start+step*i,
// second half end-step*(steps-1-i), each with ONE rounding (fma) -- probed, see DESIGN.md.
__device__ __forceinline__ float linspace_at(float start, float end, int steps, int i) {
  if (steps == 1) return start;
  float step = (end - start) / (float)(steps - 1);
  if (i < steps / 2) return fmaf(step, (float)i, start);
  return fmaf(-step, (float)(steps - 1 - i), end);
}

__device__ __forceinline__ float ts_base(float near, float far, float inv_near, float inv_far, int T, int lindisp, int i) {
  if (lindisp) {
    float tv = linspace_at(0.f, 1.f, T, i);
    return 1.0f / (inv_near * (1.0f - tv) + inv_far * tv);
  }
  return linspace_at(near, far, T, i);
}

__global__ void compute_ts_kernel(float near, float far, float inv_near, float inv_far, int T, int lindisp,
                                  float perturb, const float* __restrict__ rand, float* __restrict__ ts,
                                  float* __restrict__ mids) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= T) return;
  float t = ts_base(near, far, inv_near, inv_far, T, lindisp, i);
  if (perturb > 0.f) {
    float tp = i > 0 ? ts_base(near, far, inv_near, inv_far, T, lindisp, i - 1) : t;
    float tn = i < T - 1 ? ts_base(near, far, inv_near, inv_far, T, lindisp, i + 1) : t;
    float mid_lo = 0.5f * (tp + t);  // mids[i-1]
    float mid_hi = 0.5f * (t + tn);  // mids[i]
    float lower = i < T - 1 ? mid_hi : t;  // cat([mids, ts[-1:]])
    float upper = i > 0 ? mid_lo : t;      // cat([ts[:1], mids])
    if (mids != nullptr && i < T - 1) mids[i] = mid_hi;
    t = lower + (upper - lower) * (rand[i] * perturb);
  }
  ts[i] = t;
}

// src/nerf.py:53: pts = r_o + ts (x) r_d, T-major [T,R,3]
__global__ void compute_pts_kernel(const float* __restrict__ rays, const float* __restrict__ ts, int T, int64_t R,
                                   float* __restrict__ pts) {
  int64_t total = (int64_t)T * R;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = i % R;
    int t = (int)(i / R);
    const float* ry = rays + r * 6;
    float tt = ts[t];
    float* o = pts + i * 3;
    o[0] = ry[0] + tt * ry[3];
    o[1] = ry[1] + tt * ry[4];
    o[2] = ry[2] + tt * ry[5];
  }
}

// ------------------------------------------------------------------------------------ hash encoder
// src/neural_blocks.py:139-193.  One thread per (sample, level): 8 float4 gathers from a 1 MiB table.
// Streaming stores for outputs that are written once, are far larger than any cache and are read by ANOTHER kernel (encoder
// rows: 140-256 bytes per sample, gigabytes per frame tile): non-temporal stores leave the L2 / memory-side cache alone and
// took hash_encode from 1.74 ms to 1.05-1.19 ms per 20 M samples (1.8 -> 2.6-3.0 TB/s).  NA_STREAM_NT=0 switches them off.
#ifndef NA_STREAM_NT
#define NA_STREAM_NT 1
#endif
typedef float stream_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void stream_store4(float* dst, float a, float b, float c, float d) {
  const stream_f32x4 v = {a, b, c, d};
  if (NA_STREAM_NT) __builtin_nontemporal_store(v, (stream_f32x4*)dst);
  else *(stream_f32x4*)dst = v;
}
__device__ __forceinline__ void stream_store1(float* dst, float a) {
  if (NA_STREAM_NT) __builtin_nontemporal_store(a, dst);
  else *dst = a;
}
// lead = 1: the row starts with one more copy of x -- [x | x | features], the init row cat([p, enc(p)]) of a hash-encoded
// SkipConnMLP (src/neural_blocks.py:283-287) written by the encoder itself (training: no cat launch, round 6)
__global__ void hash_encode_kernel(const float* __restrict__ x, int64_t N, const float4* __restrict__ tables,
                                   HashRes res, int include_input, float* __restrict__ out,
                                   int64_t* __restrict__ idx_out, int lead = 0) {
  const int odim = 32 + 3 * include_input + 3 * lead;
  // A wave = 64 CONSECUTIVE SAMPLES at ONE level (8 waves of a 512-thread workgroup = the 8 levels of the same 64 samples):
  // neighbouring samples of a ray share grid cells on the coarse levels, so a gather instruction's lanes fall into few cache
  // lines (with the 8 levels of one sample in adjacent lanes every lane of an instruction hit a different table).
  // The rows of the 64 samples (64 x 35 floats, contiguous in memory) are assembled in LDS and leave as one coalesced sweep:
  // 4-byte pieces 140 bytes apart, straight from the registers, cost as much as the gathers.
  __shared__ __attribute__((aligned(16))) float rows[64 * 38];
  const int lvl = (int)(threadIdx.x >> 6);
  const int64_t nblocks = (N + 63) >> 6;
  for (int64_t blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
    const int64_t n_raw = blk * 64 + (threadIdx.x & 63);
    const int64_t n = n_raw < N ? n_raw : N - 1;  // (tail lanes repeat the last sample: every thread reaches the barriers)
    float px = x[n * 3 + 0], py = x[n * 3 + 1], pz = x[n * 3 + 2];
    float Nl = res.n[lvl];
    float vx = px * Nl, vy = py * Nl, vz = pz * Nl;
    float fx = floorf(vx), fy = floorf(vy), fz = floorf(vz);
    int lx = (int)fx, ly = (int)fy, lz = (int)fz;
    float wx = vx - fx, wy = vy - fy, wz = vz - fz;
    float iwx = 1.f - wx, iwy = 1.f - wy, iwz = 1.f - wz;
    const float4* tab = tables + (int64_t)lvl * 65536;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      // corner order of the reference: bit2 = x high, bit1 = y high, bit0 = z high
      int cx = lx + ((c >> 2) & 1), cy = ly + ((c >> 1) & 1), cz = lz + (c & 1);
      uint32_t id = hash_index(cx, cy, cz);
      if (idx_out != nullptr && n_raw < N) idx_out[((int64_t)lvl * 8 + c) * N + n] = (int64_t)id;
      float w = (((c >> 2) & 1) ? wx : iwx) * (((c >> 1) & 1) ? wy : iwy) * ((c & 1) ? wz : iwz);
      float4 e = tab[id];
      if (c == 0) {
        acc.x = e.x * w; acc.y = e.y * w; acc.z = e.z * w; acc.w = e.w * w;
      } else {
        acc.x = acc.x + e.x * w; acc.y = acc.y + e.y * w; acc.z = acc.z + e.z * w; acc.w = acc.w + e.w * w;
      }
    }
    float* o = rows + (threadIdx.x & 63) * odim + 3 * (include_input + lead) + lvl * 4;
    o[0] = acc.x; o[1] = acc.y; o[2] = acc.z; o[3] = acc.w;
    if (lvl == 0) {
      float* r = rows + (threadIdx.x & 63) * odim;
      for (int k = 0; k < include_input + lead; ++k) { r[3 * k] = px; r[3 * k + 1] = py; r[3 * k + 2] = pz; }
    }
    __syncthreads();
    const int64_t left = N - blk * 64;
    const int nval = (int)(left < 64 ? left : 64) * odim;
    float* dst = out + blk * 64 * odim;
    // (64 rows of 32 or 35 floats start on a 16-byte boundary: 16-byte stores, a quarter of the store instructions)
    const int nv4 = nval >> 2;
    for (int i = threadIdx.x; i < nv4; i += 512) {
      const float4 v = *(const float4*)(rows + 4 * i);
      stream_store4(dst + 4 * i, v.x, v.y, v.z, v.w);
    }
    for (int i = 4 * nv4 + threadIdx.x; i < nval; i += 512) stream_store1(dst + i, rows[i]);
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------ fourier / positional
// src/utils.py:14-17: [sin(x@B) | cos(x@B)], accurate sinf/cosf (arguments reach 1e3).
__device__ __forceinline__ void fourier_sincos(float m, float& sn, float& cs) {
  // Cody-Waite reduction + polynomials (common.h: 1.6e-7 / 5e-7 for |m| <= 3e3, the same pair the fused prologues use in the
  // parity mode); larger arguments (a basis far beyond the reference's sigma 16 / 32) take libm's large-argument path
  if (fabsf(m) <= 3.0e3f) sincos_cw(m, sn, cs);
  else { sn = sinf(m); cs = cosf(m); }
}
// VEC: F is a multiple of 4 -- a thread owns 4 consecutive frequencies of one sample: 16-byte stores, a quarter of the store
// instructions (the one-frequency-per-thread form ran at 30 % of the HBM rate whatever the sine cost)
template <bool VEC>
__global__ void fourier_kernel(const float* __restrict__ x, int64_t N, int D, const float* __restrict__ basis, int F,
                               float scale, float* __restrict__ out) {
  if constexpr (VEC) {
    const int F4 = F >> 2;
    const int64_t total = N * F4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
      const int j = (int)(i % F4) * 4;
      const int64_t n = i / F4;
      float m[4];
      for (int d = 0; d < D; ++d) {
        const float xd = x[n * D + d];
        const float4 bv = *(const float4*)(basis + d * F + j);
        const float b[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float be = scale == 1.0f ? b[e] : scale * b[e];
          m[e] = d == 0 ? xd * be : fmaf(xd, be, m[e]);
        }
      }
      float sn[4], cs[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) fourier_sincos(m[e], sn[e], cs[e]);
      stream_store4(out + n * 2 * F + j, sn[0], sn[1], sn[2], sn[3]);
      stream_store4(out + n * 2 * F + F + j, cs[0], cs[1], cs[2], cs[3]);
    }
  } else {
    int64_t total = N * F;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
      int j = (int)(i % F);
      int64_t n = i / F;
      float m = 0.f;
      for (int d = 0; d < D; ++d) {
        float b = scale == 1.0f ? basis[d * F + j] : scale * basis[d * F + j];
        m = d == 0 ? x[n * D + d] * b : fmaf(x[n * D + d], b, m);
      }
      float sn, cs;
      fourier_sincos(m, sn, cs);
      stream_store1(out + n * 2 * F + j, sn);
      stream_store1(out + n * 2 * F + F + j, cs);
    }
  }
}

// src/neural_blocks.py:30-34: raw[n, d*NB + k] = x[n,d]*bands[k]; out = [sin(raw) | cos(raw)]
template <bool VEC>
__global__ void positional_kernel(const float* __restrict__ x, int64_t N, int D, const float* __restrict__ bands, int NB,
                                  float* __restrict__ out) {
  int W = D * NB;
  if constexpr (VEC) {  // NB a multiple of 4: a thread owns 4 consecutive bands of one input dimension, 16-byte stores
    const int W4 = W >> 2;
    const int64_t total = N * W4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
      const int j = (int)(i % W4) * 4;
      const int64_t n = i / W4;
      const float xv = x[n * D + j / NB];
      const float4 bv = *(const float4*)(bands + j % NB);
      const float raw[4] = {xv * bv.x, xv * bv.y, xv * bv.z, xv * bv.w};
      stream_store4(out + n * 2 * W + j, sinf(raw[0]), sinf(raw[1]), sinf(raw[2]), sinf(raw[3]));
      stream_store4(out + n * 2 * W + W + j, cosf(raw[0]), cosf(raw[1]), cosf(raw[2]), cosf(raw[3]));
    }
  } else {
    int64_t total = N * W;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
      int j = (int)(i % W);
      int64_t n = i / W;
      float raw = x[n * D + j / NB] * bands[j % NB];
      stream_store1(out + n * 2 * W + j, sinf(raw));
      stream_store1(out + n * 2 * W + W + j, cosf(raw));
    }
  }
}

__global__ void elaz_kernel(const float* __restrict__ d, int64_t N, float* __restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
    float e, a;
    elev_azim(d[i * 3], d[i * 3 + 1], d[i * 3 + 2], e, a);
    out[i * 2] = e;
    out[i * 2 + 1] = a;
  }
}

// rows [x, y, z, elev, azim] of the View reflectance's input (src/refl.py:190-207: cat([x, dir_to_elev_azim(view)]) with the
// direction of a ray broadcast along its samples): one launch instead of elaz + expand + cat (five small kernels, 48 us per
// training step of 262 144 samples).  n = t * R + r.
__global__ void view_rows_kernel(const float* __restrict__ pts, const float* __restrict__ dirs, int64_t N, int64_t R,
                                 float* __restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i % R;
    float e, a;
    elev_azim(dirs[r * 3], dirs[r * 3 + 1], dirs[r * 3 + 2], e, a);
    float* o = out + i * 5;
    o[0] = pts[i * 3]; o[1] = pts[i * 3 + 1]; o[2] = pts[i * 3 + 2]; o[3] = e; o[4] = a;
  }
}

// PlainNeRF's step between its two networks in training (src/nerf.py:338-357, src/refl.py:190-207): density = first_out[..., 0]
// (contiguous) and the View MLP's init rows [x, y, z, elev, azim | first_out[..., 1:]] by ONE kernel (round 6: slice copy + elaz +
// expand + two cats + the latent's contiguous copy were seven launches, 85 us per step of 262 144 samples).  A workgroup owns 32
// consecutive rows: its output tile is one contiguous run of 32 (5 + C) floats, written by consecutive lanes; the row of an
// element comes from a float reciprocal (exact for the tile's < 2^16 elements), elevation / azimuth of the tile's rays once per
// row by the first 32 threads (a flat one-element-per-thread form made every wave walk through acosf / atan2f: 83 us).  n = t R + r.
__global__ __launch_bounds__(256) void plain_head_rows_kernel(const float* __restrict__ first_out, const float* __restrict__ pts,
                                                              const float* __restrict__ dirs, int64_t N, int64_t R, int C,
                                                              float* __restrict__ density, float* __restrict__ rows) {
  __shared__ float ea[32][2];
  const int W = 5 + C;
  const float invW = 1.0f / (float)W;
  const int64_t ntile = (N + 31) >> 5;
  for (int64_t tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const int64_t n0 = tile << 5;
    const int nrow = (int)(N - n0 < 32 ? N - n0 : 32);
    if (threadIdx.x < nrow) {
      const int64_t n = n0 + threadIdx.x;
      const int64_t r = n % R;
      float e, a;
      elev_azim(dirs[r * 3], dirs[r * 3 + 1], dirs[r * 3 + 2], e, a);
      ea[threadIdx.x][0] = e; ea[threadIdx.x][1] = a;
      density[n] = first_out[n * (1 + C)];
    }
    __syncthreads();
    const int total = nrow * W;
    float* dst = rows + n0 * W;
    for (int i = threadIdx.x; i < total; i += 256) {
      int row = (int)(((float)i + 0.5f) * invW);
      int c = i - row * W;
      if (c < 0) { --row; c += W; } else if (c >= W) { ++row; c -= W; }
      const int64_t n = n0 + row;
      dst[i] = c < 3 ? pts[n * 3 + c] : c < 5 ? ea[row][c - 3] : first_out[n * (1 + C) + 1 + (c - 5)];
    }
    __syncthreads();
  }
}

__global__ void sigmoid_kernel(const float* __restrict__ x, int64_t N, int kind, float* __restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = apply_sigmoid_kind(x[i], kind);
}

// ------------------------------------------------------------------------------------ mip IPE (intended layout)
// src/utils.py:23-27,39-48,60-101 with cov laid out like mean ([T,B,H,W,3]); radii_x differences rows of
// the crop (src/utils.py:77-81; the appended last row repeats the second-to-last difference).
// One thread per (sample, 4 consecutive features): the [.., 6*nd] rows leave as coalesced float4 stores (a thread per
// sample wrote 96 floats 384 bytes apart from its neighbour: 0.45 TB/s); the few dozen flops of cone/cylinder geometry
// are recomputed per thread.
template <int VEC>
__global__ void mip_kernel(const float* __restrict__ rays, int B, int H, int W, const float* __restrict__ ts, int T,
                           int kind, float t_end, int min_deg, int max_deg, float* __restrict__ out) {
  const int nd = max_deg - min_deg;
  const int F = 6 * nd;
  const int per = F / VEC;
  int64_t R = (int64_t)B * H * W;
  int64_t total = (int64_t)T * R * per;
  for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < total; j += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = j / per;
    const int f0 = (int)(j % per) * VEC;
    int64_t r = i % R;
    int t = (int)(i / R);
    int wq = (int)(r % W);
    int hq = (int)((r / W) % H);
    int b = (int)(r / ((int64_t)W * H));
    const float rad = mip_radius(rays, H, W, b, hq, wq);
    const float t0 = ts[t], t1 = t < T - 1 ? ts[t + 1] : mip_last_edge(ts, T, t_end);
    const MipGauss gs = mip_gaussian(rays + r * 6, rad, t0, t1, kind);
    float v[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) v[e] = mip_feature(gs.m0, gs.m1, gs.m2, gs.c0, gs.c1, gs.c2, f0 + e, nd, min_deg);
    float* o = out + i * F + f0;
    if constexpr (VEC == 4) stream_store4(o, v[0], v[1], v[2], v[3]);
    else stream_store1(o, v[0]);
  }
}

// ------------------------------------------------------------------------------------ compositing
// src/nerf.py:22-27,60-80,96-98.  One thread per ray walks T in order (T-major => every step is a
// coalesced row read); the running product is the reference's cumprod association exactly.
template <int C>
__global__ void composite_kernel(const float* __restrict__ density, const float* __restrict__ feat,
                                 const float* __restrict__ ts, const float* __restrict__ rays, int T, int64_t R,
                                 int density_kind, int bg_kind, float* __restrict__ alpha_out,
                                 float* __restrict__ weights_out, float* __restrict__ out, int Crt,
                                 const float* __restrict__ sky_rand = nullptr) {
  const int CC = C > 0 ? C : Crt;
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < R; r += (int64_t)gridDim.x * blockDim.x) {
    const float* ry = rays + r * 6 + 3;
    float nrm = sqrtf((ry[0] * ry[0] + ry[1] * ry[1]) + ry[2] * ry[2]);
    float trans = 1.0f;
    float acc[C > 0 ? C : 8];
    for (int c = 0; c < CC; ++c) acc[c] = 0.f;
    float wsum_head = 0.f;  // sum of weights[:-1] for the white background (Q4)
    // The walk is a dependent chain of T steps, and a step that waits for its own loads costs one HBM latency (165 us for
    // 128 steps = 1.3 us per step with ~10 waves per CU: latency-bound, 37 % of the HBM rate).  Rows are therefore fetched
    // U = 8 steps at a time -- 8 x (1 + C) independent loads in flight per thread -- before the 8 dependent updates run.
    constexpr int U = 8;
    for (int t0 = 0; t0 < T; t0 += U) {
      float dv[U], fv[U][C > 0 ? C : 8];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int t = t0 + u < T ? t0 + u : T - 1;  // (clamped: the tail re-reads the last row and ignores it)
        dv[u] = density[(int64_t)t * R + r];
        const float* f = feat + ((int64_t)t * R + r) * CC;
        for (int c = 0; c < CC; ++c) fv[u][c] = f[c];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int t = t0 + u;
        if (t < T) {
          const float d = dv[u];
          // hardware transcendentals (common.h): with libm's log1pf(expf()) / expf() this kernel was ALU-bound at ~300
          // instructions per sample.  One thread per (ray, 32-step segment) with the segments meeting in LDS -- 4x the
          // threads -- measured slower.
          float sigma = density_kind == NA_DENSITY_SOFTPLUS_M1 ? fast_softplus(d - 1.0f) : fmaxf(d, 0.f);
          float dist = t < T - 1 ? fmaxf(ts[t + 1] - ts[t], 1e-5f) : 1e10f;
          dist = dist * nrm;
          float a = 1.0f - fast_exp(-sigma * dist);
          float w = a * trans;
          trans = trans * ((1.0f - a) + 1e-10f);
          if (alpha_out != nullptr) alpha_out[(int64_t)t * R + r] = a;
          if (weights_out != nullptr) weights_out[(int64_t)t * R + r] = w;
          for (int c = 0; c < CC; ++c) acc[c] = acc[c] + w * fv[u][c];
          if (t < T - 1) wsum_head = wsum_head + w;
        }
      }
    }
    // white: the remainder 1 - sum(weights[:-1]) (src/nerf.py:98); random: one uniform draw per ray times it (src/nerf.py:101-103)
    float sky = bg_kind == NA_BG_WHITE ? 1.0f - wsum_head : bg_kind == NA_BG_RANDOM ? sky_rand[r] * (1.0f - wsum_head) : 0.f;
    for (int c = 0; c < CC; ++c) out[r * CC + c] = acc[c] + sky;
  }
}

// out[r, :] += rand[r] * (1 - sum_{t < T-1} weights[t, r]): the random background (src/nerf.py:101-103) behind a renderer that
// composited against black and kept its weights (the fused one-kernel renderers)
__global__ void sky_random_kernel(const float* __restrict__ weights, const float* __restrict__ rand, int T, int64_t R, int C,
                                  float* __restrict__ out) {
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < R; r += (int64_t)gridDim.x * blockDim.x) {
    float head = 0.f;
    for (int t = 0; t < T - 1; ++t) head = head + weights[(int64_t)t * R + r];
    const float sky = rand[r] * (1.0f - head);
    for (int c = 0; c < C; ++c) out[r * C + c] = out[r * C + c] + sky;
  }
}

__global__ void integrate_kernel(const float* __restrict__ weights, const float* __restrict__ other, int T, int64_t R,
                                 int C, float* __restrict__ out) {
  int64_t total = R * C;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = i / C;
    int c = (int)(i % C);
    float acc = 0.f;
    for (int t = 0; t < T; ++t) acc = acc + weights[(int64_t)t * R + r] * other[((int64_t)t * R + r) * C + c];
    out[i] = acc;
  }
}

// src/utils.py:50-58 + src/nerf.py:1000-1003
__device__ __forceinline__ float laplace_density_of(float sdf, float sc, float inv) {
  const float scaled = (-sdf) / sc;
  // (fast_exp: v_exp_f32 with the product's rounding error recovered, 2e-7 relative -- libm's expf made this elementwise
  // kernel ALU-bound at 30 % of the HBM rate)
  const float cdf = scaled <= 0.f ? fast_exp(fminf(scaled, 0.f)) / 2.f : 1.f - fast_exp(-fmaxf(scaled, 0.f)) / 2.f;
  return inv * cdf;
}
__global__ void laplace_density_kernel(const float* __restrict__ sdf, int64_t N, const float* __restrict__ beta,
                                       float* __restrict__ density) {
  const float sc = beta[0], inv = 1.0f / sc;
  const int64_t n4 = (((uintptr_t)sdf | (uintptr_t)density) & 15) == 0 ? N / 4 : 0;  // 16-byte rows when both are aligned
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 v = ((const float4*)sdf)[i];
    ((float4*)density)[i] = make_float4(laplace_density_of(v.x, sc, inv), laplace_density_of(v.y, sc, inv),
                                        laplace_density_of(v.z, sc, inv), laplace_density_of(v.w, sc, inv));
  }
  for (int64_t i = n4 * 4 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x)
    density[i] = laplace_density_of(sdf[i], sc, inv);
}

// src/nerf.py:1173-1178 (de Casteljau), 1201-1206 (cubic), 1267-1278 (warp)
// + the reflectance latent of --dyn-refl-latent (src/nerf.py:1246-1248, 1272-1278): n_rl more columns ride through the same
// spline (control point k of column j at est[3n + 2 + k * n_rl + j]) and leave scaled by sigmoid(est[3n + 1]).
__global__ __launch_bounds__(256) void bezier_warp_kernel(const float* __restrict__ est, int est_stride, const float* __restrict__ pts,
                                   const float* __restrict__ tt, int64_t N, int n, float* __restrict__ out_pts,
                                   float* __restrict__ dp_out, float* __restrict__ rig_out, int n_rl = 0,
                                   float* __restrict__ enc_out = nullptr) {
  // (round 6) the rows of a workgroup's consecutive samples (256; 64 for rows wider than 63 floats: 64 KiB of LDS) are ONE contiguous run: fetched by consecutive lanes
  // into LDS, then every thread walks its own row there (pitch est_stride | 1: odd, conflict-free).  One thread reading its 76-byte
  // row straight from memory ran the kernel at 1.3 TB/s (0.95 ms per 10 M samples of config 4's shard).
  // Only for rows of 32 floats and more (`staged`, a launch constant: `make dnerf`'s 38 columns 2.23 -> 1.25 ms per 10 M samples);
  // a thread walking its own 76-byte row straight from memory is faster for the narrow rows (19 columns: 0.33 against 0.57 ms).
  extern __shared__ float rowbuf[];
  const int P = est_stride | 1;
  const bool staged = est_stride >= 32;
  for (int64_t base = (int64_t)blockIdx.x * blockDim.x; base < N; base += (int64_t)gridDim.x * blockDim.x) {
    const int nrow = (int)(N - base < (int64_t)blockDim.x ? N - base : (int64_t)blockDim.x);
    if (staged) {
      const int total = nrow * est_stride;
      const float* src = est + base * est_stride;
      for (int k = threadIdx.x; k < total; k += blockDim.x) {
        const int r = k / est_stride, c = k - r * est_stride;
        rowbuf[r * P + c] = src[k];
      }
      __syncthreads();
    }
    const int64_t i = base + threadIdx.x;
    if ((int)threadIdx.x < nrow) {
    const float* e = staged ? rowbuf + threadIdx.x * P : est + i * est_stride;
    float rig = sigmoidf_(e[0] / 2.f);
    float t = tt[i], m1t = 1.f - t;
    float dp[3];
    if (n == 4) {
      float m2 = m1t * m1t, t2 = t * t;
      float k0 = m2 * m1t, k1 = 3.f * m2 * t, k2 = 3.f * t2 * m1t, k3 = t2 * t;
      for (int a = 0; a < 3; ++a)
        dp[a] = ((k0 * e[1 + a] + k1 * e[4 + a]) + k2 * e[7 + a]) + k3 * e[10 + a];
    } else {
      float b[8][3];
      for (int k = 0; k < n; ++k)
        for (int a = 0; a < 3; ++a) b[k][a] = e[1 + 3 * k + a];
      for (int it = 1; it < n; ++it)
        for (int k = 0; k < n - it; ++k)
          for (int a = 0; a < 3; ++a) b[k][a] = b[k][a] * m1t + b[k + 1][a] * t;
      for (int a = 0; a < 3; ++a) dp[a] = b[0][a];
    }
    for (int a = 0; a < 3; ++a) {
      out_pts[i * 3 + a] = pts[i * 3 + a] + dp[a] * rig;
      if (dp_out != nullptr) dp_out[i * 3 + a] = dp[a];
    }
    if (rig_out != nullptr) rig_out[i] = rig;
    if (n_rl > 0) {
      const float* c = e + 3 * n + 2;
      const float enc_rig = sigmoidf_(e[3 * n + 1]);
      for (int j = 0; j < n_rl; ++j) {
        float v;
        if (n == 4) {
          float m2 = m1t * m1t, t2 = t * t;
          float k0 = m2 * m1t, k1 = 3.f * m2 * t, k2 = 3.f * t2 * m1t, k3 = t2 * t;
          v = ((k0 * c[j] + k1 * c[n_rl + j]) + k2 * c[2 * n_rl + j]) + k3 * c[3 * n_rl + j];
        } else {
          float b[8];
          for (int k = 0; k < n; ++k) b[k] = c[k * n_rl + j];
          for (int it = 1; it < n; ++it)
            for (int k = 0; k < n - it; ++k) b[k] = b[k] * m1t + b[k + 1] * t;
          v = b[0];
        }
        enc_out[i * n_rl + j] = v * enc_rig;
      }
    }
    }
    if (staged) __syncthreads();
  }
}

// F.normalize(v, dim=-1) (eps 1e-12), src/refl.py:281: the raw-view input of PosLinearView
__global__ void normalize3_kernel(const float* __restrict__ v, int64_t N, float* __restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
    const float x = v[i * 3], y = v[i * 3 + 1], z = v[i * 3 + 2];
    const float n = fmaxf(sqrtf((x * x + y * y) + z * z), 1e-12f);
    out[i * 3] = x / n; out[i * 3 + 1] = y / n; out[i * 3 + 2] = z / n;
  }
}

// PosLinearView combine (src/refl.py:288-290): out[n,c] = (sigmoid(lin[n]) / 2 + 0.5) * pos[n * pos_ld + c]
__global__ void pos_linear_combine_kernel(const float* __restrict__ lin, const float* __restrict__ pos, int64_t pos_ld,
                                          int64_t N, int C, float* __restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N * C; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i / C;
    const int c = (int)(i - n * C);
    out[i] = (sigmoidf_(lin[n]) / 2.f + 0.5f) * pos[n * pos_ld + c];
  }
}


// ---- coarse -> fine resampling (BASELINE config 2 "64 + 128").  The reference's sample_pdf (src/nerf.py:1745-1779, called at
// :572-578 with (mids, weights[:-1], steps_fine)) is dead code: it prints, gathers bins with cdf indices (one too many for its
// `mids`) and calls exit().  INTENDED reading, pinned only against an fp64 restatement (oracle.sample_pdf_intended): weight i of
// the coarse pass is the mass of the interval [ts[i], ts[i + 1]] -- exactly how alpha_from_density defines it -- so the T - 1
// weights weights[:-1] and the T bin edges `ts` (the call site's own "TODO see if ts works ok here?") give
//     w' = w + 1e-5;  cdf = [0, cumsum(w' / sum w')]  (T entries);  u = linspace(0, 1, N) | rand(N, ...)
//     inds = searchsorted(cdf, u, right=True);  below = max(inds - 1, 0);  above = min(inds, T - 1)
//     denom = cdf[above] - cdf[below] (1 where < 1e-5);  sample = ts[below] + (u - cdf[below]) / denom * (ts[above] - ts[below])
// and the fine pass evaluates the union of the coarse and the new positions in order (NeRF's z_vals = sort(cat(...))).
// One workgroup = 64 consecutive rays (weight / u tiles read coalesced: lane = ray), then one wave per ray: cdf by a wave scan in
// fp64, N binary searches in LDS, and a stable rank sort of the T + N positions (coarse before fine on ties), written as one
// contiguous row per ray -- the layout the renderers take per-ray steps in.
// (lanes of ONE wave exchange data through LDS: order the wave's own writes before its reads, for the compiler and the memory model)
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__global__ __launch_bounds__(256) void resample_ts_kernel(const float* __restrict__ ts, const float* __restrict__ w, int64_t R,
                                                          int T, const float* __restrict__ u, int N, float* __restrict__ fine,
                                                          float* __restrict__ merged) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t ray0 = (int64_t)blockIdx.x * 64;
  const int M = T + N, Mp = (M + 3) & ~3;
  float* wt = (float*)smem;                                  // [T - 1][64]
  float* ut = wt + (size_t)(T - 1) * 64;                     // [N][64] (only with u)
  char* pw = (char*)(ut + (u != nullptr ? (size_t)N * 64 : 0));
  pw += (size_t)wave * ((size_t)T * 8 + (size_t)Mp * 4);
  double* cdf = (double*)pw;                                 // [T]
  float* vals = (float*)(pw + (size_t)T * 8);                // [T + N]: coarse steps, then the new ones
  for (int i = tid; i < (T - 1) * 64; i += 256) {
    const int64_t r = ray0 + (i & 63);
    wt[i] = r < R ? w[(int64_t)(i >> 6) * R + r] : 0.f;
  }
  if (u != nullptr)
    for (int i = tid; i < N * 64; i += 256) {
      const int64_t r = ray0 + (i & 63);
      ut[i] = r < R ? u[(int64_t)(i >> 6) * R + r] : 0.f;
    }
  __syncthreads();
  for (int q = 0; q < 16; ++q) {
    const int rl = wave * 16 + q;
    const int64_t ray = ray0 + rl;
    if (ray >= R) break;  // (wave-uniform)
    // ---- cdf: inclusive scan of w' over the T - 1 intervals, 64 at a time
    double total = 0.0;
    for (int i0 = 0; i0 < T - 1; i0 += 64) {
      const int i = i0 + lane;
      double v = i < T - 1 ? (double)wt[i * 64 + rl] + 1e-5 : 0.0;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const double o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
      }
      if (i < T - 1) cdf[i + 1] = total + v;
      total += __shfl(v, 63, 64);
    }
    if (lane == 0) cdf[0] = 0.0;
    for (int i = lane; i < T; i += 64) vals[i] = ts[i];
    wave_sync();
    for (int i = 1 + lane; i < T; i += 64) cdf[i] = cdf[i] / total;
    wave_sync();
    // ---- N inverse-cdf samples
    const float step = 1.0f / (float)(N > 1 ? N - 1 : 1);
    for (int j = lane; j < N; j += 64) {
      float uf;
      if (u != nullptr) uf = ut[j * 64 + rl];
      else uf = (N == 1 || j < N / 2) ? step * (float)j : 1.0f - step * (float)(N - 1 - j);  // torch.linspace(0, 1, N) in fp32 ([0] for N = 1)
      const double uj = (double)uf;
      int lo = 0, hi = T;  // first index with cdf > u
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cdf[mid] <= uj) lo = mid + 1; else hi = mid;
      }
      const int below = lo - 1 > 0 ? lo - 1 : 0, above = lo < T - 1 ? lo : T - 1;
      double den = cdf[above] - cdf[below];
      if (den < 1e-5) den = 1.0;
      const double tt = (uj - cdf[below]) / den;
      const double b0 = (double)vals[below], b1 = (double)vals[above];
      const float sj = (float)(b0 + tt * (b1 - b0));
      vals[T + j] = sj;
      if (fine != nullptr) fine[ray * N + j] = sj;
    }
    wave_sync();
    // ---- stable rank sort of the T + N positions
    if (merged != nullptr) {
      // (keys, not float compares: a strict total order -- NaN positions from non-finite weights sort last instead of all
      // taking the same rank and leaving slots of the row unwritten; -0 never occurs, the steps are positive)
      auto key = [](float f) -> uint32_t {
        const uint32_t b = __float_as_uint(f);
        if ((b & 0x7fffffffu) > 0x7f800000u) return 0xffffffffu;
        return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
      };
      for (int i = lane; i < M; i += 64) {
        const float v = vals[i];
        const uint32_t kv = key(v);
        int rank = 0;
        for (int k = 0; k < M; ++k) {
          const uint32_t ko = key(vals[k]);
          rank += (ko < kv || (ko == kv && k < i)) ? 1 : 0;
        }
        merged[ray * M + rank] = v;
      }
    }
    wave_sync();
  }
}

}  // namespace na

// ================================================================================================ C ABI
using namespace na;

extern "C" {

int na_version(void) { return NA_VERSION; }

int na_set_deterministic(void* workspace, size_t bytes) {
  NA_REQUIRE(workspace == nullptr || ((uintptr_t)workspace % 8 == 0 && bytes >= 8), NA_EINVAL,
             "na_set_deterministic: workspace must be 8-byte aligned");
  int dev = -1;
  if (workspace != nullptr) {
    hipPointerAttribute_t attr;
    NA_REQUIRE(hipPointerGetAttributes(&attr, workspace) == hipSuccess && attr.type == hipMemoryTypeDevice, NA_EINVAL,
               "na_set_deterministic: workspace is not device memory");
    dev = attr.device;
  }
  na::g_det = {(long long*)workspace, workspace ? bytes : 0};
  na::g_det_device = dev;
  return NA_OK;
}
const char* na_last_error(void) { return na::g_err; }

int na_raygen(const float* c2w, int B, float focal, int size, int crop_t, int crop_l, int crop_h, int crop_w,
              const float* noise, float with_noise, float* rays, void* stream) {
  NA_REQUIRE(B > 0 && size > 0 && crop_h >= 0 && crop_w >= 0 && crop_t >= 0 && crop_l >= 0, NA_EINVAL,
             "na_raygen: bad shape B=%d size=%d crop=(%d,%d,%d,%d)", B, size, crop_t, crop_l, crop_h, crop_w);
  NA_REQUIRE(crop_t + crop_h <= size && crop_l + crop_w <= size, NA_EINVAL,
             "na_raygen: crop (%d,%d,%d,%d) exceeds image %d (clip it like the reference's slicing)", crop_t, crop_l,
             crop_h, crop_w, size);
  int64_t total = (int64_t)B * crop_h * crop_w;
  if (total == 0) return NA_OK;  // empty crop: nothing to write (the output pointer may be NULL)
  NA_REQUIRE(c2w && rays, NA_ENULL, "na_raygen: null pointer");
  const float* nz = (noise != nullptr && with_noise != 0.f) ? noise : nullptr;
  hipLaunchKernelGGL(raygen_kernel, dim3(grid_for(total, 256, 4096)), dim3(256), 0, (hipStream_t)stream, c2w, B, focal,
                     (float)(size * 0.5), crop_t, crop_l, crop_h, crop_w, nz, with_noise, rays);
  return check_launch("na_raygen");
}

int na_raygen_dtu(const float* pose, const float* intrinsic, int B, int size, int crop_t, int crop_l, int crop_h,
                  int crop_w, float* rays, void* stream) {
  NA_REQUIRE(B > 0 && size > 0 && crop_h >= 0 && crop_w >= 0, NA_EINVAL, "na_raygen_dtu: bad shape");
  int64_t total = (int64_t)B * crop_h * crop_w;
  if (total == 0) return NA_OK;
  NA_REQUIRE(pose && intrinsic && rays, NA_ENULL, "na_raygen_dtu: null pointer");
  hipLaunchKernelGGL(raygen_dtu_kernel, dim3(grid_for(total, 256, 4096)), dim3(256), 0, (hipStream_t)stream, pose,
                     intrinsic, B, size, crop_t, crop_l, crop_h, crop_w, rays);
  return check_launch("na_raygen_dtu");
}

int na_compute_ts(float near, float far, int T, int lindisp, float perturb, const float* rand, float* ts, float* mids,
                  void* stream) {
  NA_REQUIRE(ts, NA_ENULL, "na_compute_ts: null ts");
  NA_REQUIRE(T >= 1, NA_EINVAL, "na_compute_ts: T=%d", T);
  NA_REQUIRE(!(perturb > 0.f) || rand != nullptr, NA_ENULL, "na_compute_ts: perturb>0 needs rand[T]");
  float inv_near = (float)(1.0 / fmax((double)near, 1e-10));
  float inv_far = (float)(1.0 / (double)far);
  hipLaunchKernelGGL(compute_ts_kernel, dim3((T + 255) / 256), dim3(256), 0, (hipStream_t)stream, near, far, inv_near,
                     inv_far, T, lindisp, perturb, rand, ts, mids);
  return check_launch("na_compute_ts");
}

int na_compute_pts(const float* rays, const float* ts, int T, int64_t R, float* pts, void* stream) {
  NA_REQUIRE(rays && ts && pts, NA_ENULL, "na_compute_pts: null pointer");
  NA_REQUIRE(T >= 1 && R >= 0, NA_EINVAL, "na_compute_pts: bad shape");
  if (R == 0) return NA_OK;
  hipLaunchKernelGGL(compute_pts_kernel, dim3(grid_for((int64_t)T * R, 256, 8192)), dim3(256), 0, (hipStream_t)stream,
                     rays, ts, T, R, pts);
  return check_launch("na_compute_pts");
}

int na_hash_encode(const float* x, int64_t N, const float* tables, int include_input, float* out, int64_t* idx_out,
                   void* stream) {
  NA_REQUIRE(x && tables && out, NA_ENULL, "na_hash_encode: null pointer");
  NA_REQUIRE(N >= 0, NA_EINVAL, "na_hash_encode: N=%lld", (long long)N);
  if (N == 0) return NA_OK;
  // the rows leave the kernel as 16-byte non-temporal stores (a row is 35 or 32 floats: every 64-sample block starts on a
  // 16-byte boundary of `out` exactly when `out` itself is aligned); torch allocations are 256-byte aligned
  NA_REQUIRE(((uintptr_t)out & 15) == 0, NA_EINVAL, "na_hash_encode: out must be 16-byte aligned (got %p)", (void*)out);
  hipLaunchKernelGGL(hash_encode_kernel, dim3(grid_for((N + 63) / 64 * 512, 512, 16384)), dim3(512), 0, (hipStream_t)stream, x, N,
                     (const float4*)tables, hash_resolutions(), include_input ? 1 : 0, out, idx_out, 0);
  return check_launch("na_hash_encode");
}

int na_hash_encode_rows(const float* x, int64_t N, const float* tables, int include_input, int lead, float* out, void* stream) {
  NA_REQUIRE(N >= 0 && (lead == 0 || lead == 1), NA_EINVAL, "na_hash_encode_rows: N=%lld lead=%d (0 | 1)", (long long)N, lead);
  if (N == 0) return NA_OK;  // empty: zero-size tensors carry null pointers
  NA_REQUIRE(x && tables && out, NA_ENULL, "na_hash_encode_rows: null pointer");
  NA_REQUIRE(((uintptr_t)out & 15) == 0, NA_EINVAL, "na_hash_encode_rows: out must be 16-byte aligned (got %p)", (void*)out);
  hipLaunchKernelGGL(hash_encode_kernel, dim3(grid_for((N + 63) / 64 * 512, 512, 16384)), dim3(512), 0, (hipStream_t)stream, x, N,
                     (const float4*)tables, hash_resolutions(), include_input ? 1 : 0, out, (int64_t*)nullptr, lead);
  return check_launch("na_hash_encode_rows");
}

int na_fourier_encode(const float* x, int64_t N, int D, const float* basis, int F, float scale, float* out,
                      void* stream) {
  NA_REQUIRE(x && basis && out, NA_ENULL, "na_fourier_encode: null pointer");
  NA_REQUIRE(N >= 0 && D >= 1 && F >= 1, NA_EINVAL, "na_fourier_encode: bad shape");
  if (N == 0) return NA_OK;
  if ((F & 3) == 0 && ((uintptr_t)basis & 15) == 0 && ((uintptr_t)out & 15) == 0)
    hipLaunchKernelGGL(fourier_kernel<true>, dim3(grid_for(N * (F / 4), 256, 16384)), dim3(256), 0, (hipStream_t)stream, x, N,
                       D, basis, F, scale, out);
  else
    hipLaunchKernelGGL(fourier_kernel<false>, dim3(grid_for(N * F, 256, 16384)), dim3(256), 0, (hipStream_t)stream, x, N, D,
                       basis, F, scale, out);
  return check_launch("na_fourier_encode");
}

int na_positional_encode(const float* x, int64_t N, int D, const float* bands, int NB, float* out, void* stream) {
  NA_REQUIRE(x && bands && out, NA_ENULL, "na_positional_encode: null pointer");
  NA_REQUIRE(N >= 0 && D >= 1 && NB >= 1, NA_EINVAL, "na_positional_encode: bad shape");
  if (N == 0) return NA_OK;
  if ((NB & 3) == 0 && ((uintptr_t)bands & 15) == 0 && ((uintptr_t)out & 15) == 0)
    hipLaunchKernelGGL(positional_kernel<true>, dim3(grid_for(N * (D * NB / 4), 256, 16384)), dim3(256), 0, (hipStream_t)stream,
                       x, N, D, bands, NB, out);
  else
    hipLaunchKernelGGL(positional_kernel<false>, dim3(grid_for(N * D * NB, 256, 16384)), dim3(256), 0, (hipStream_t)stream, x,
                       N, D, bands, NB, out);
  return check_launch("na_positional_encode");
}

int na_view_elaz(const float* dirs, int64_t N, float* out, void* stream) {
  NA_REQUIRE(dirs && out, NA_ENULL, "na_view_elaz: null pointer");
  if (N <= 0) return N == 0 ? NA_OK : NA_EINVAL;
  hipLaunchKernelGGL(elaz_kernel, dim3(grid_for(N, 256, 8192)), dim3(256), 0, (hipStream_t)stream, dirs, N, out);
  return check_launch("na_view_elaz");
}

int na_view_rows(const float* pts, const float* dirs, int64_t N, int64_t R, float* out, void* stream) {
  NA_REQUIRE(pts && dirs && out, NA_ENULL, "na_view_rows: null pointer");
  NA_REQUIRE(N >= 0 && R >= 1 && N % R == 0, NA_EINVAL, "na_view_rows: N %lld is not a multiple of R %lld", (long long)N, (long long)R);
  if (N == 0) return NA_OK;
  hipLaunchKernelGGL(view_rows_kernel, dim3(grid_for(N, 256, 8192)), dim3(256), 0, (hipStream_t)stream, pts, dirs, N, R, out);
  return check_launch("na_view_rows");
}

int na_plain_head_rows(const float* first_out, const float* pts, const float* dirs, int64_t N, int64_t R, int C, float* density,
                       float* rows, void* stream) {
  NA_REQUIRE(N >= 0 && R >= 1 && C >= 1 && N % R == 0, NA_EINVAL, "na_plain_head_rows: N %lld R %lld C %d", (long long)N, (long long)R, C);
  if (N == 0) return NA_OK;
  NA_REQUIRE(first_out && pts && dirs && density && rows, NA_ENULL, "na_plain_head_rows: null pointer");
  NA_REQUIRE(C <= 1024, NA_EINVAL, "na_plain_head_rows: C %d (<= 1024)", C);
  hipLaunchKernelGGL(plain_head_rows_kernel, dim3(grid_for((N + 31) / 32 * 256, 256, 16384)), dim3(256), 0, (hipStream_t)stream, first_out,
                     pts, dirs, N, R, C, density, rows);
  return check_launch("na_plain_head_rows");
}

int na_sigmoid(const float* x, int64_t N, int kind, float* out, void* stream) {
  NA_REQUIRE(x && out, NA_ENULL, "na_sigmoid: null pointer");
  NA_REQUIRE(kind >= 0 && kind <= NA_SIG_IDENTITY, NA_EUNSUPPORTED, "na_sigmoid: unknown kind %d", kind);
  if (N <= 0) return N == 0 ? NA_OK : NA_EINVAL;
  hipLaunchKernelGGL(sigmoid_kernel, dim3(grid_for(N, 256, 8192)), dim3(256), 0, (hipStream_t)stream, x, N, kind, out);
  return check_launch("na_sigmoid");
}

int na_mip_encode(const float* rays, int B, int H, int W, const float* ts, int T, int kind, float t_end, int min_deg,
                  int max_deg, float* out, void* stream) {
  NA_REQUIRE(rays && ts && out, NA_ENULL, "na_mip_encode: null pointer");
  NA_REQUIRE(B >= 1 && H >= 2 && W >= 1 && T >= 1 && max_deg > min_deg, NA_EINVAL,
             "na_mip_encode: bad shape (radii_x needs H>=2 rows)");
  NA_REQUIRE(kind == 0 || kind == 1, NA_EUNSUPPORTED, "na_mip_encode: kind %d", kind);
  const int F = 6 * (max_deg - min_deg);
  int64_t total = (int64_t)T * B * H * W;
  if (F % 4 == 0)
    hipLaunchKernelGGL(mip_kernel<4>, dim3(grid_for(total * (F / 4), 256, 1 << 18)), dim3(256), 0, (hipStream_t)stream,
                       rays, B, H, W, ts, T, kind, t_end, min_deg, max_deg, out);
  else
    hipLaunchKernelGGL(mip_kernel<1>, dim3(grid_for(total * F, 256, 1 << 18)), dim3(256), 0, (hipStream_t)stream, rays,
                       B, H, W, ts, T, kind, t_end, min_deg, max_deg, out);
  return check_launch("na_mip_encode");
}

int na_composite(const float* density, const float* feat, const float* ts, const float* rays, int T, int64_t R, int C,
                 int density_kind, int bg_kind, float* alpha, float* weights, float* out, void* stream) {
  NA_REQUIRE(density && feat && ts && rays && out, NA_ENULL, "na_composite: null pointer");
  NA_REQUIRE(T >= 1 && R >= 0 && C >= 1 && C <= 8, NA_EINVAL, "na_composite: bad shape T=%d R=%lld C=%d (C<=8)", T,
             (long long)R, C);
  NA_REQUIRE(density_kind == 0 || density_kind == 1, NA_EUNSUPPORTED, "na_composite: density kind %d", density_kind);
  NA_REQUIRE(bg_kind == NA_BG_BLACK || bg_kind == NA_BG_WHITE, NA_EUNSUPPORTED, "na_composite: bg kind %d", bg_kind);
  if (R == 0) return NA_OK;
  dim3 g(grid_for(R, 128, 1 << 16)), b(128);
  if (C == 3)
    hipLaunchKernelGGL(composite_kernel<3>, g, b, 0, (hipStream_t)stream, density, feat, ts, rays, T, R, density_kind,
                       bg_kind, alpha, weights, out, C);
  else
    hipLaunchKernelGGL(composite_kernel<0>, g, b, 0, (hipStream_t)stream, density, feat, ts, rays, T, R, density_kind,
                       bg_kind, alpha, weights, out, C);
  return check_launch("na_composite");
}

int na_composite_random_bg(const float* density, const float* feat, const float* ts, const float* rays, int T, int64_t R, int C,
                           int density_kind, const float* rand, float* alpha, float* weights, float* out, void* stream) {
  NA_REQUIRE(density && feat && ts && rays && rand && out, NA_ENULL, "na_composite_random_bg: null pointer");
  NA_REQUIRE(T >= 1 && R >= 0 && C >= 1 && C <= 8, NA_EINVAL, "na_composite_random_bg: bad shape T=%d R=%lld C=%d (C<=8)", T,
             (long long)R, C);
  NA_REQUIRE(density_kind == 0 || density_kind == 1, NA_EUNSUPPORTED, "na_composite_random_bg: density kind %d", density_kind);
  if (R == 0) return NA_OK;
  dim3 g(grid_for(R, 128, 1 << 16)), b(128);
  if (C == 3)
    hipLaunchKernelGGL(composite_kernel<3>, g, b, 0, (hipStream_t)stream, density, feat, ts, rays, T, R, density_kind,
                       (int)NA_BG_RANDOM, alpha, weights, out, C, rand);
  else
    hipLaunchKernelGGL(composite_kernel<0>, g, b, 0, (hipStream_t)stream, density, feat, ts, rays, T, R, density_kind,
                       (int)NA_BG_RANDOM, alpha, weights, out, C, rand);
  return check_launch("na_composite_random_bg");
}

int na_sky_random(const float* weights, const float* rand, int T, int64_t R, int C, float* out, void* stream) {
  NA_REQUIRE(T >= 1 && R >= 0 && C >= 1, NA_EINVAL, "na_sky_random: bad shape");
  if (R == 0) return NA_OK;
  NA_REQUIRE(weights && rand && out, NA_ENULL, "na_sky_random: null pointer");
  hipLaunchKernelGGL(sky_random_kernel, dim3(grid_for(R, 256, 1 << 16)), dim3(256), 0, (hipStream_t)stream, weights, rand, T, R, C, out);
  return check_launch("na_sky_random");
}

int na_integrate(const float* weights, const float* other, int T, int64_t R, int C, float* out, void* stream) {
  NA_REQUIRE(weights && other && out, NA_ENULL, "na_integrate: null pointer");
  NA_REQUIRE(T >= 1 && R >= 0 && C >= 1, NA_EINVAL, "na_integrate: bad shape");
  if (R == 0) return NA_OK;
  hipLaunchKernelGGL(integrate_kernel, dim3(grid_for(R * C, 256, 1 << 16)), dim3(256), 0, (hipStream_t)stream, weights,
                     other, T, R, C, out);
  return check_launch("na_integrate");
}

int na_laplace_density(const float* sdf, int64_t N, const float* beta, float* density, void* stream) {
  NA_REQUIRE(sdf && beta && density, NA_ENULL, "na_laplace_density: null pointer");
  if (N <= 0) return N == 0 ? NA_OK : NA_EINVAL;
  hipLaunchKernelGGL(laplace_density_kernel, dim3(grid_for(N, 256, 8192)), dim3(256), 0, (hipStream_t)stream, sdf, N,
                     beta, density);
  return check_launch("na_laplace_density");
}

int na_bezier_warp(const float* est, int est_stride, const float* pts, const float* t, int64_t N, int n_ctrl,
                   float* out_pts, float* dp, float* rigidity_out, void* stream) {
  NA_REQUIRE(est && pts && t && out_pts, NA_ENULL, "na_bezier_warp: null pointer");
  NA_REQUIRE(n_ctrl >= 2 && n_ctrl <= 8 && est_stride >= 1 + 3 * n_ctrl, NA_EINVAL,
             "na_bezier_warp: n_ctrl=%d (2..8) stride=%d", n_ctrl, est_stride);
  if (N <= 0) return N == 0 ? NA_OK : NA_EINVAL;
  NA_REQUIRE(est_stride <= 255, NA_EINVAL, "na_bezier_warp: est_stride %d (<= 255)", est_stride);
  const int bw = est_stride <= 63 ? 256 : 64;
  hipLaunchKernelGGL(bezier_warp_kernel, dim3(grid_for(N, bw, 32768)), dim3(bw), est_stride >= 32 ? bw * (est_stride | 1) * sizeof(float) : 0, (hipStream_t)stream, est,
                     est_stride, pts, t, N, n_ctrl, out_pts, dp, rigidity_out, 0, (float*)nullptr);
  return check_launch("na_bezier_warp");
}

int na_bezier_warp_latent(const float* est, int est_stride, const float* pts, const float* t, int64_t N, int n_ctrl,
                          int n_rl, float* out_pts, float* dp, float* rigidity_out, float* refl_latent, void* stream) {
  NA_REQUIRE(est && pts && t && out_pts && refl_latent, NA_ENULL, "na_bezier_warp_latent: null pointer");
  NA_REQUIRE(n_ctrl >= 2 && n_ctrl <= 8 && n_rl >= 1 && n_rl <= 16 && est_stride >= 2 + (3 + n_rl) * n_ctrl, NA_EINVAL,
             "na_bezier_warp_latent: n_ctrl=%d (2..8) n_rl=%d (1..16) stride=%d (>= 2 + (3 + n_rl) * n_ctrl)", n_ctrl, n_rl,
             est_stride);
  if (N <= 0) return N == 0 ? NA_OK : NA_EINVAL;
  NA_REQUIRE(est_stride <= 255, NA_EINVAL, "na_bezier_warp_latent: est_stride %d (<= 255)", est_stride);
  const int bw = est_stride <= 63 ? 256 : 64;
  hipLaunchKernelGGL(bezier_warp_kernel, dim3(grid_for(N, bw, 32768)), dim3(bw), est_stride >= 32 ? bw * (est_stride | 1) * sizeof(float) : 0, (hipStream_t)stream, est,
                     est_stride, pts, t, N, n_ctrl, out_pts, dp, rigidity_out, n_rl, refl_latent);
  return check_launch("na_bezier_warp_latent");
}

int na_normalize3(const float* v, int64_t N, float* out, void* stream) {
  if (N == 0) return NA_OK;
  NA_REQUIRE(v && out, NA_ENULL, "na_normalize3: null pointer");
  NA_REQUIRE(N > 0, NA_EINVAL, "na_normalize3: N=%lld", (long long)N);
  hipLaunchKernelGGL(normalize3_kernel, dim3(grid_for(N, 256, 8192)), dim3(256), 0, (hipStream_t)stream, v, N, out);
  return check_launch("na_normalize3");
}

int na_pos_linear_combine(const float* lin, const float* pos, int64_t pos_ld, int64_t N, int C, float* out,
                          void* stream) {
  if (N == 0) return NA_OK;
  NA_REQUIRE(lin && pos && out, NA_ENULL, "na_pos_linear_combine: null pointer");
  NA_REQUIRE(N > 0 && C >= 1 && pos_ld >= C, NA_EINVAL, "na_pos_linear_combine: bad shape");
  hipLaunchKernelGGL(pos_linear_combine_kernel, dim3(grid_for(N * C, 256, 8192)), dim3(256), 0, (hipStream_t)stream, lin,
                     pos, pos_ld, N, C, out);
  return check_launch("na_pos_linear_combine");
}

size_t na_resample_ts_lds_bytes(int T, int N, int with_u) {
  if (T < 2 || N < 1) return 0;
  return (size_t)(T - 1) * 256 + (with_u ? (size_t)N * 256 : 0) + 4 * ((size_t)T * 8 + (size_t)((T + N + 3) & ~3) * 4);
}

int na_resample_ts(const float* ts, const float* weights, int64_t R, int T, const float* u, int N, float* fine, float* merged,
                   void* stream) {
  NA_REQUIRE(T >= 2 && N >= 1 && R >= 0, NA_EINVAL, "na_resample_ts: bad shape T=%d N=%d R=%lld", T, N, (long long)R);
  if (R == 0) return NA_OK;
  NA_REQUIRE(ts && weights && (fine || merged), NA_ENULL, "na_resample_ts: null pointer");
  const size_t lds = na_resample_ts_lds_bytes(T, N, u != nullptr);
  NA_REQUIRE(lds <= 160 * 1024, NA_EUNSUPPORTED, "na_resample_ts: T=%d, N=%d need %zu bytes of LDS (160 KiB per workgroup)", T, N, lds);
  static std::atomic<uint64_t> attr_done{0};
  int dev = 0;
  NA_REQUIRE(hipGetDevice(&dev) == hipSuccess, NA_EHIP, "na_resample_ts: hipGetDevice failed");
  const uint64_t bit = 1ull << (dev & 63);
  if (!(attr_done.load(std::memory_order_acquire) & bit)) {
    hipError_t e = hipFuncSetAttribute((const void*)resample_ts_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    NA_REQUIRE(e == hipSuccess, NA_EHIP, "na_resample_ts: hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_done.fetch_or(bit, std::memory_order_release);
  }
  hipLaunchKernelGGL(resample_ts_kernel, dim3((unsigned)((R + 63) / 64)), dim3(256), lds, (hipStream_t)stream, ts, weights, R, T, u,
                     N, fine, merged);
  return check_launch("na_resample_ts");
}

}  // extern "C"
