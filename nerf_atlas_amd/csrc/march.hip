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

}  // extern "C"
