// SDF ray marching state updates (src/march.py:27-47 sphere_march, :78-110
// throughput_with_sign_change, :147-180 bisection).  The reference gathers the still-active rays with boolean masks on
// every iteration (a host-visible compaction); here every ray keeps a persistent state in HBM, the SDF network is
// evaluated for all rays by the fused MLP kernel (no host sync, no gather) and these elementwise kernels apply the
// update only where the reference would have: inactive rays are left untouched, so the results are identical.
#include "common.h"

namespace na {

// pts[r] = r_o[r] + r_d[r] * t with t = t_ray[r] (per ray) or t_scalar
__global__ void ray_points_kernel(const float* __restrict__ r_o, const float* __restrict__ r_d,
                                  const float* __restrict__ t_ray, float t_scalar, int64_t R, float* __restrict__ pts) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < R * 3; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / 3;
    const float t = t_ray != nullptr ? t_ray[r] : t_scalar;
    pts[i] = r_o[i] + r_d[i] * t;   // r_o + r_d * t: product first, then the sum, like torch
  }
}

// src/march.py:39-45 for the rays with rem != 0
__global__ void sphere_march_update_kernel(const float* __restrict__ sdf, int stride, int64_t R, float eps, float far,
                                           float* __restrict__ dist, uint8_t* __restrict__ hits,
                                           uint8_t* __restrict__ rem) {
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < R; r += (int64_t)gridDim.x * blockDim.x) {
    if (!rem[r]) continue;
    const float d = sdf[r * stride];
    float cd = dist[r];
    uint8_t h = hits[r] | (uint8_t)((d < eps) && (cd <= far));
    cd += d;
    dist[r] = cd;
    hits[r] = h;
    if (h || cd > far) rem[r] = 0;
  }
}

// ---- compaction of the live rays (src/march.py:37-45, :164-179: the reference indexes its state with boolean masks, so every
// iteration evaluates the SDF network for the rays that are still marching only).  ORDERED and deterministic: block b owns the
// contiguous segment [b * seg, (b + 1) * seg) of the rays; pass 1 counts its live rays, pass 2 places them behind the live rays
// of the segments in front of it.  idx comes out ascending, so the gathers / scatters of the indexed kernels stay coalesced
// where rays are live in runs (they are: neighbouring pixels).
constexpr int kCompactBlocks = 256;
__global__ __launch_bounds__(256) void compact_count_kernel(const uint8_t* __restrict__ live, int64_t R, int64_t seg,
                                                            int32_t* __restrict__ counts) {
  __shared__ int wsum[4];
  const int64_t r0 = blockIdx.x * seg, r1 = r0 + seg < R ? r0 + seg : R;
  int c = 0;
  for (int64_t r = r0 + threadIdx.x; r < r1; r += 256) c += live[r] != 0;
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
__global__ __launch_bounds__(256) void compact_place_kernel(const uint8_t* __restrict__ live, int64_t R, int64_t seg,
                                                            const int32_t* __restrict__ counts, int32_t* __restrict__ idx,
                                                            int32_t* __restrict__ count) {
  __shared__ int wsum[4];
  __shared__ int base_s;
  if (threadIdx.x == 0) {
    int b = 0, total = 0;
    for (int i = 0; i < kCompactBlocks; ++i) { if (i == (int)blockIdx.x) b = total; total += counts[i]; }
    base_s = b;
    if (blockIdx.x == 0) *count = total;
  }
  __syncthreads();
  int base = base_s;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t r0 = blockIdx.x * seg, r1 = r0 + seg < R ? r0 + seg : R;
  for (int64_t c0 = r0; c0 < r1; c0 += 256) {
    const int64_t r = c0 + threadIdx.x;
    const bool on = r < r1 && live[r] != 0;
    const unsigned long long m = __ballot(on);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wv] = __popcll(m);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wv; ++w) off += wsum[w];
    if (on) idx[off + before] = (int32_t)r;
    base += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
}

// pts[i] = r_o[idx[i]] + r_d[idx[i]] * t_ray[idx[i]] for the n compacted rays
__global__ void ray_points_indexed_kernel(const float* __restrict__ r_o, const float* __restrict__ r_d,
                                          const float* __restrict__ t_ray, const int32_t* __restrict__ idx, int64_t n,
                                          float* __restrict__ pts) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n * 3; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t k = i / 3, r = idx[k];
    const int c = (int)(i - k * 3);
    pts[i] = r_o[r * 3 + c] + r_d[r * 3 + c] * t_ray[r];
  }
}

// sphere_march_update_kernel on compacted rows: row i of `sdf` belongs to ray idx[i] (all of them have rem != 0)
__global__ void sphere_march_update_indexed_kernel(const float* __restrict__ sdf, int stride, const int32_t* __restrict__ idx,
                                                   int64_t n, float eps, float far, float* __restrict__ dist,
                                                   uint8_t* __restrict__ hits, uint8_t* __restrict__ rem) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = idx[i];
    if (!rem[r]) continue;
    const float d = sdf[i * stride];
    float cd = dist[r];
    uint8_t h = hits[r] | (uint8_t)((d < eps) && (cd <= far));
    cd += d;
    dist[r] = cd;
    hits[r] = h;
    if (h || cd > far) rem[r] = 0;
  }
}

// src/march.py:96-103, one uniform step i (0-based)
__global__ void sign_change_update_kernel(const float* __restrict__ sdf, int stride, int64_t R, int step,
                                          float* __restrict__ curr_min, int32_t* __restrict__ idxs,
                                          int32_t* __restrict__ last_pos, int32_t* __restrict__ first_neg) {
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < R; r += (int64_t)gridDim.x * blockDim.x) {
    const float sd = sdf[r * stride];
    const float cm = curr_min[r];
    if (sd < cm) idxs[r] = step + 1;
    curr_min[r] = fminf(cm, sd);
    if (first_neg[r] == -1 && sd < 0.f) {
      last_pos[r] = step;
      first_neg[r] = step + 1;
    }
  }
}

__device__ __forceinline__ bool bisect_todo(float low, float high, float sl, float sh, float eps) {
  return ((high - low) > eps) && (sl > 0.f) && (sh < 0.f) && (high > low);
}

// src/march.py:159-162 (init: sdf_mid == nullptr) and :166-179 (one iteration)
__global__ void bisection_update_kernel(const float* __restrict__ sdf_mid, int stride, int64_t R, float eps,
                                        float* __restrict__ low, float* __restrict__ high, float* __restrict__ sdf_low,
                                        float* __restrict__ sdf_high, float* __restrict__ z, uint8_t* __restrict__ todo) {
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < R; r += (int64_t)gridDim.x * blockDim.x) {
    float lo = low[r], hi = high[r], sl = sdf_low[r], sh = sdf_high[r];
    bool td;
    if (sdf_mid == nullptr) {
      td = bisect_todo(lo, hi, sl, sh, eps);
    } else {
      td = todo[r];
      const float sm = sdf_mid[r * stride], zp = z[r];
      if (sm > 0.f && td) { lo = zp; sl = sm; }
      if (sm < 0.f && td) { hi = zp; sh = sm; }
      td = td && bisect_todo(lo, hi, sl, sh, eps);
      low[r] = lo; high[r] = hi; sdf_low[r] = sl; sdf_high[r] = sh;
    }
    z[r] = (lo + hi) / 2.f;
    todo[r] = td;
  }
}

// bisection_update_kernel (one iteration) on compacted rows: row i of sdf_mid belongs to ray idx[i] (todo != 0)
__global__ void bisection_update_indexed_kernel(const float* __restrict__ sdf_mid, int stride, const int32_t* __restrict__ idx,
                                                int64_t n, float eps, float* __restrict__ low, float* __restrict__ high,
                                                float* __restrict__ sdf_low, float* __restrict__ sdf_high, float* __restrict__ z,
                                                uint8_t* __restrict__ todo) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = idx[i];
    float lo = low[r], hi = high[r], sl = sdf_low[r], sh = sdf_high[r];
    bool td = todo[r];
    const float sm = sdf_mid[i * stride], zp = z[r];
    if (sm > 0.f && td) { lo = zp; sl = sm; }
    if (sm < 0.f && td) { hi = zp; sh = sm; }
    td = td && bisect_todo(lo, hi, sl, sh, eps);
    low[r] = lo; high[r] = hi; sdf_low[r] = sl; sdf_high[r] = sh;
    z[r] = (lo + hi) / 2.f;
    todo[r] = td;
  }
}

// src/lights.py:118-132 Point.forward for N points: direction to the light (F.normalize, eps 1e-6), distance and
// inverse-square spectrum.  center / intensity: one [3] vector (stride 0) or one per point (stride 3).
__global__ void point_light_kernel(const float* __restrict__ x, const float* __restrict__ center, int c_stride,
                                   const float* __restrict__ intensity, int i_stride, int decay, int64_t N,
                                   float* __restrict__ dir, float* __restrict__ dist, float* __restrict__ spectrum) {
#pragma clang fp contract(off)
  for (int64_t n = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x) {
    const float* c = center + n * c_stride;
    const float* in = intensity + n * i_stride;
    const float dx = c[0] - x[3 * n], dy = c[1] - x[3 * n + 1], dz = c[2] - x[3 * n + 2];
    const float len = sqrtf(dx * dx + dy * dy + dz * dz);
    const float den = fmaxf(len, 1e-6f);
    dir[3 * n] = dx / den; dir[3 * n + 1] = dy / den; dir[3 * n + 2] = dz / den;
    dist[n] = len;
    const float att = (len * len) * 12.566370614359172f;  // 4 * pi * dist^2 in the reference's order
    for (int k = 0; k < 3; ++k) spectrum[3 * n + k] = decay ? in[k] / att : in[k];
  }
}

// src/renderers.py:40-45, :65-67, :82-83, :118-121, :143-146: spectrum * a(raw_att) * v(visible)
//   att_mode 0: a = 1;  1: a = sigmoid(raw) only where hidden (learned);  2: a = sigmoid(raw) + 1e-2 everywhere
//   visible == nullptr: v = 1; else v = visible ? 1 : hidden_value (0 hard, sigmoid(alpha) learned-const; mode 1: 1)
__global__ void occlusion_apply_kernel(const float* __restrict__ spectrum, const uint8_t* __restrict__ visible,
                                       const float* __restrict__ raw_att, int att_mode, float hidden_value, int64_t N,
                                       float* __restrict__ out) {
#pragma clang fp contract(off)
  for (int64_t n = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x) {
    const bool vis = visible == nullptr || visible[n] != 0;
    float s0 = spectrum[3 * n], s1 = spectrum[3 * n + 1], s2 = spectrum[3 * n + 2];
    if (att_mode == 2) {
      const float a = sigmoidf_(raw_att[n]) + 1e-2f;
      s0 *= a; s1 *= a; s2 *= a;
    }
    if (!vis) {
      const float h = att_mode == 1 ? sigmoidf_(raw_att[n]) : hidden_value;
      s0 *= h; s1 *= h; s2 *= h;
    }
    out[3 * n] = s0; out[3 * n + 1] = s1; out[3 * n + 2] = s2;
  }
}

}  // namespace na

using namespace na;

extern "C" {

int na_ray_points(const float* r_o, const float* r_d, const float* t_ray, float t_scalar, int64_t R, float* pts,
                  void* stream) {
  if (R == 0) return NA_OK;  // empty: zero-size tensors carry null pointers
  NA_REQUIRE(r_o && r_d && pts, NA_ENULL, "na_ray_points: null pointer");
  if (R <= 0) return R == 0 ? NA_OK : NA_EINVAL;
  hipLaunchKernelGGL(ray_points_kernel, dim3(grid_for(R * 3, 256, 8192)), dim3(256), 0, (hipStream_t)stream, r_o, r_d,
                     t_ray, t_scalar, R, pts);
  return check_launch("na_ray_points");
}

int na_sphere_march_update(const float* sdf, int stride, int64_t R, float eps, float far, float* dist, uint8_t* hits,
                           uint8_t* rem, void* stream) {
  if (R == 0) return NA_OK;  // empty: zero-size tensors carry null pointers
  NA_REQUIRE(sdf && dist && hits && rem, NA_ENULL, "na_sphere_march_update: null pointer");
  NA_REQUIRE(stride >= 1, NA_EINVAL, "na_sphere_march_update: stride %d", stride);
  if (R <= 0) return R == 0 ? NA_OK : NA_EINVAL;
  hipLaunchKernelGGL(sphere_march_update_kernel, dim3(grid_for(R, 256, 8192)), dim3(256), 0, (hipStream_t)stream, sdf,
                     stride, R, eps, far, dist, hits, rem);
  return check_launch("na_sphere_march_update");
}

int na_compact_rays(const uint8_t* live, int64_t R, int32_t* idx, int32_t* count, void* stream) {
  NA_REQUIRE(R >= 0 && R < (1ll << 31), NA_EINVAL, "na_compact_rays: R %lld", (long long)R);
  NA_REQUIRE(idx && count, NA_ENULL, "na_compact_rays: null pointer");
  if (R == 0) { (void)hipMemsetAsync(count, 0, sizeof(int32_t), (hipStream_t)stream); return check_launch("na_compact_rays"); }
  NA_REQUIRE(live, NA_ENULL, "na_compact_rays: null pointer");
  const int64_t seg = (R + kCompactBlocks - 1) / kCompactBlocks;
  int32_t* counts = idx + R;  // scratch behind the indices: the caller allocates R + 256 entries
  hipLaunchKernelGGL(compact_count_kernel, dim3(kCompactBlocks), dim3(256), 0, (hipStream_t)stream, live, R, seg, counts);
  hipLaunchKernelGGL(compact_place_kernel, dim3(kCompactBlocks), dim3(256), 0, (hipStream_t)stream, live, R, seg, counts, idx, count);
  return check_launch("na_compact_rays");
}

int na_ray_points_indexed(const float* r_o, const float* r_d, const float* t_ray, const int32_t* idx, int64_t n, float* pts,
                          void* stream) {
  if (n == 0) return NA_OK;
  NA_REQUIRE(n > 0, NA_EINVAL, "na_ray_points_indexed: n %lld", (long long)n);
  NA_REQUIRE(r_o && r_d && t_ray && idx && pts, NA_ENULL, "na_ray_points_indexed: null pointer");
  hipLaunchKernelGGL(ray_points_indexed_kernel, dim3(grid_for(n * 3, 256, 8192)), dim3(256), 0, (hipStream_t)stream, r_o, r_d,
                     t_ray, idx, n, pts);
  return check_launch("na_ray_points_indexed");
}

int na_sphere_march_update_indexed(const float* sdf, int stride, const int32_t* idx, int64_t n, float eps, float far, float* dist,
                                   uint8_t* hits, uint8_t* rem, void* stream) {
  if (n == 0) return NA_OK;
  NA_REQUIRE(n > 0 && stride >= 1, NA_EINVAL, "na_sphere_march_update_indexed: n %lld stride %d", (long long)n, stride);
  NA_REQUIRE(sdf && idx && dist && hits && rem, NA_ENULL, "na_sphere_march_update_indexed: null pointer");
  hipLaunchKernelGGL(sphere_march_update_indexed_kernel, dim3(grid_for(n, 256, 8192)), dim3(256), 0, (hipStream_t)stream, sdf,
                     stride, idx, n, eps, far, dist, hits, rem);
  return check_launch("na_sphere_march_update_indexed");
}

int na_bisection_update_indexed(const float* sdf_mid, int stride, const int32_t* idx, int64_t n, float eps, float* low,
                                float* high, float* sdf_low, float* sdf_high, float* z, uint8_t* todo, void* stream) {
  if (n == 0) return NA_OK;
  NA_REQUIRE(n > 0 && stride >= 1, NA_EINVAL, "na_bisection_update_indexed: n %lld stride %d", (long long)n, stride);
  NA_REQUIRE(sdf_mid && idx && low && high && sdf_low && sdf_high && z && todo, NA_ENULL, "na_bisection_update_indexed: null pointer");
  hipLaunchKernelGGL(bisection_update_indexed_kernel, dim3(grid_for(n, 256, 8192)), dim3(256), 0, (hipStream_t)stream, sdf_mid,
                     stride, idx, n, eps, low, high, sdf_low, sdf_high, z, todo);
  return check_launch("na_bisection_update_indexed");
}

int na_sign_change_update(const float* sdf, int stride, int64_t R, int step, float* curr_min, int32_t* idxs,
                          int32_t* last_pos, int32_t* first_neg, void* stream) {
  if (R == 0) return NA_OK;  // empty: zero-size tensors carry null pointers
  NA_REQUIRE(sdf && curr_min && idxs && last_pos && first_neg, NA_ENULL, "na_sign_change_update: null pointer");
  NA_REQUIRE(stride >= 1 && step >= 0, NA_EINVAL, "na_sign_change_update: stride %d step %d", stride, step);
  if (R <= 0) return R == 0 ? NA_OK : NA_EINVAL;
  hipLaunchKernelGGL(sign_change_update_kernel, dim3(grid_for(R, 256, 8192)), dim3(256), 0, (hipStream_t)stream, sdf,
                     stride, R, step, curr_min, idxs, last_pos, first_neg);
  return check_launch("na_sign_change_update");
}

int na_bisection_update(const float* sdf_mid, int stride, int64_t R, float eps, float* low, float* high, float* sdf_low,
                        float* sdf_high, float* z, uint8_t* todo, void* stream) {
  if (R == 0) return NA_OK;  // empty: zero-size tensors carry null pointers
  NA_REQUIRE(low && high && sdf_low && sdf_high && z && todo, NA_ENULL, "na_bisection_update: null pointer");
  NA_REQUIRE(stride >= 1, NA_EINVAL, "na_bisection_update: stride %d", stride);
  if (R <= 0) return R == 0 ? NA_OK : NA_EINVAL;
  hipLaunchKernelGGL(bisection_update_kernel, dim3(grid_for(R, 256, 8192)), dim3(256), 0, (hipStream_t)stream, sdf_mid,
                     stride, R, eps, low, high, sdf_low, sdf_high, z, todo);
  return check_launch("na_bisection_update");
}

int na_point_light(const float* x, const float* center, int center_stride, const float* intensity, int intensity_stride,
                   int distance_decay, int64_t N, float* dir, float* dist, float* spectrum, void* stream) {
  if (N == 0) return NA_OK;
  NA_REQUIRE(N > 0, NA_EINVAL, "na_point_light: N %lld", (long long)N);
  NA_REQUIRE(x && center && intensity && dir && dist && spectrum, NA_ENULL, "na_point_light: null pointer");
  NA_REQUIRE((center_stride == 0 || center_stride == 3) && (intensity_stride == 0 || intensity_stride == 3), NA_EINVAL,
             "na_point_light: strides must be 0 (one light) or 3 (one per point)");
  hipLaunchKernelGGL(point_light_kernel, dim3(grid_for(N, 256, 8192)), dim3(256), 0, (hipStream_t)stream, x, center,
                     center_stride, intensity, intensity_stride, distance_decay, N, dir, dist, spectrum);
  return check_launch("na_point_light");
}

int na_occlusion_apply(const float* spectrum, const uint8_t* visible, const float* raw_att, int att_mode,
                       float hidden_value, int64_t N, float* out, void* stream) {
  if (N == 0) return NA_OK;
  NA_REQUIRE(N > 0, NA_EINVAL, "na_occlusion_apply: N %lld", (long long)N);
  NA_REQUIRE(spectrum && out, NA_ENULL, "na_occlusion_apply: null pointer");
  NA_REQUIRE(att_mode >= 0 && att_mode <= 2, NA_EINVAL, "na_occlusion_apply: att_mode %d", att_mode);
  NA_REQUIRE(att_mode == 0 || raw_att, NA_ENULL, "na_occlusion_apply: att_mode %d needs raw_att", att_mode);
  NA_REQUIRE(att_mode != 1 || visible, NA_ENULL, "na_occlusion_apply: att_mode 1 needs the visibility mask");
  hipLaunchKernelGGL(occlusion_apply_kernel, dim3(grid_for(N, 256, 8192)), dim3(256), 0, (hipStream_t)stream, spectrum,
                     visible, raw_att, att_mode, hidden_value, N, out);
  return check_launch("na_occlusion_apply");
}

}  // extern "C"
