// Shared helpers for the gfx950 kernels of nerf_atlas_amd (C-ABI: include/nerf_atlas_amd.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/nerf_atlas_amd.h"

namespace na {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return NA_EHIP;
  }
  return NA_OK;
}

#define NA_REQUIRE(cond, code, ...)      \
  do {                                   \
    if (!(cond)) {                       \
      na::set_error(__VA_ARGS__);        \
      return (code);                     \
    }                                    \
  } while (0)

inline unsigned grid_for(int64_t n, int block, int64_t cap = 1 << 20) {
  int64_t g = (n + block - 1) / block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (unsigned)g;
}

// Deterministic accumulation (na_set_deterministic): gradients that are summed across workgroups go through 64-bit
// FIXED-POINT atomics instead of fp32 atomics: integer addition is associative, so the result is bitwise independent of
// the order in which workgroups arrive.  The int64 accumulator lives in a caller workspace that the entry point zeroes,
// fills and folds into the fp32 output (det_begin / det_finish).
//   resolution  2^-40 (9.1e-13) per addend: addends below 4.5e-13 in magnitude vanish (fp32 atomics keep ~6e-8 RELATIVE; the
//               1e-4-scale hash-table gradients of a fresh model sit 8 decades above the step, Adam's scale invariance does
//               not reach below it);
//   range       |sum| < 2^23 (8.4e6) before the int64 wraps; addends of 2^20 and more, and every non-finite addend, bypass
//               the accumulator and go to the fp32 output with a plain atomic (order-dependent rounding for those, but a
//               diverged step still shows up as Inf / NaN instead of as finite garbage from __float2ll_rn);
//   scope       ONE workspace per process, bound to the device it was allocated on (det_begin refuses another current
//               device); the entry points that use it must not run concurrently on several streams.
constexpr float kFixScale = 1099511627776.0f;  // 2^40
struct DetWs { long long* ptr; size_t bytes; };
DetWs det_workspace();  // {nullptr, 0} when the deterministic mode is off (basic_ops.hip)
long long* det_begin(size_t n, hipStream_t stream, const char* who, int* rc);
int det_finish(const long long* fix, size_t n, float* out, hipStream_t stream, const char* who);

// Python-double level resolutions of the reference hash encoder (src/neural_blocks.py:126-128,146):
// N_l = 16 * exp((ln 16384 - ln 16)/8 - 1)^l, then cast to fp32 when multiplied with the fp32 input.
struct HashRes { float n[8]; };
HashRes hash_resolutions();

// ---- device helpers ---------------------------------------------------------------------------
// out[idx] += v: fp32 atomic (fast, order-dependent rounding) or fixed-point atomic (deterministic)
__device__ __forceinline__ void accumulate(float* out, long long* fix, int64_t idx, float v) {
  // (!(|v| < 2^20) is true for NaN as well)
  if (fix != nullptr && fabsf(v) < 1048576.0f) atomicAdd((unsigned long long*)(fix + idx), (unsigned long long)__float2ll_rn(v * kFixScale));
  else atomicAdd(out + idx, v);
}

__device__ __forceinline__ float leaky_relu(float v) { return v > 0.f ? v : v * 0.01f; }

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

__device__ __forceinline__ float softplusf_(float v) { return v > 20.f ? v : log1pf(expf(v)); }

// ---- compositing arithmetic on the hardware transcendentals (v_exp_f32 / v_log_f32 / v_rcp_f32, 1 ulp each).  The
// libm calls they replace made the compositing of a block the longest piece of the exposed EP phase (~5 k cycles).
// exp(x): x log2(e) with the product's rounding error recovered by an fma (2e-7 relative for |x| <= 88)
__device__ __forceinline__ float fast_exp(float x) {
  x = fminf(x, 88.f);
  const float t = x * 1.4426950408889634f;
  const float r = fmaf(x, 1.4426950408889634f, -t) + x * 1.925963033500235e-8f;
  const float e = __builtin_amdgcn_exp2f(t);
  return fmaf(e, r * 0.6931471805599453f, e);
}
__device__ __forceinline__ float fast_sigmoid(float v) { return __builtin_amdgcn_rcpf(1.0f + fast_exp(-v)); }
// log1p(exp(v)) with the RELATIVE accuracy the 1e10-long last interval needs (src/nerf.py:60-68: sigma * 1e10): series
// below 2^-6, log(u) * e / (u - 1) above (u = fl(1 + e); the quotient undoes the rounding of the sum)
__device__ __forceinline__ float fast_softplus(float v) {
  if (v > 20.f) return v;
  const float e = fast_exp(v);
  const float series = e * fmaf(e, fmaf(e, fmaf(e, -0.25f, 0.33333334f), -0.5f), 1.0f);
  const float u = 1.0f + e;
  const float lg = __builtin_amdgcn_logf(u) * 0.6931471805599453f * (e * __builtin_amdgcn_rcpf(u - 1.0f));
  return e < 0.015625f ? series : lg;
}
__device__ __forceinline__ float apply_sigmoid_kind(float v, int kind) {
  switch (kind) {
    case NA_SIG_NORMAL: return sigmoidf_(v);
    case NA_SIG_THIN: return (sigmoidf_(v) * (1.f + 2.f * -1e-2f) - -1e-2f) + 1e-2f;
    case NA_SIG_FAT: return sigmoidf_(v) * (1.f + 2.f * 1e-2f) - 1e-2f;
    case NA_SIG_TANH: return tanhf(v);
    case NA_SIG_UPSHIFTED: return sigmoidf_(v) + 1e-2f;
    case NA_SIG_RELU: return fmaxf(v, 0.f);
    case NA_SIG_SIN: return sinf(v);
    case NA_SIG_LEAKY_RELU: return leaky_relu(v);
    case NA_SIG_UPSHIFTED_SOFTPLUS: return softplusf_(v) + 1e-2f;
    case NA_SIG_UPSHIFTED_RELU: return fmaxf(v, 0.f) + 1e-2f;
    case NA_SIG_CYCLIC: return (sinf(v / 5.f) + 1.f) / 2.f * (1.f + 2.f * -1e-2f) - -1e-2f;
    default: return v;
  }
}

// sin with a 2-term Cody-Waite reduction by pi and a degree-9 odd polynomial (least-squares on
// Chebyshev nodes of [-pi/2,pi/2]): max |err| 1.6e-7 for |x| <= 3e3 (checked against fp64).
__device__ __forceinline__ float sin_cw(float x) {
  float q = rintf(x * 0.318309886183790672f);
  float r = fmaf(q, -3.140625f, x);
  r = fmaf(q, -9.67502593994140625e-4f, r);
  r = fmaf(q, -1.509957990978376432e-7f, r);
  float r2 = r * r;
  float p = fmaf(r2, 2.5962193818e-06f, -1.9804804431e-04f);
  p = fmaf(p, r2, 8.3329907333e-03f);
  p = fmaf(p, r2, -1.6666655917e-01f);
  float s = fmaf(p * r2, r, r);
  int qi = (int)q;
  return (qi & 1) ? -s : s;
}

// Gaussian of one conical-frustum / cylinder sample (src/utils.py:39-48 lift, :60-101 cylinder / cone moments, cov laid
// out like mean): mean[3], diagonal cov[3] from the ray, its pixel radius and the interval [t0, t1].  kind 0 = cylinder.
// (scalar members: arrays in this struct are indexed through scratch memory by the per-axis selects of mip_feature)
struct MipGauss { float m0, m1, m2, c0, c1, c2; };
__device__ __forceinline__ MipGauss mip_gaussian(const float* ry, float rad, float t0, float t1, int kind) {
  float t_mean, t_var, r_var;
  if (kind == 0) {
    t_mean = (t1 + t0) / 2.f;
    r_var = rad * rad / 4.f;
    float dt = t1 - t0;
    t_var = dt * dt / 12.f;
  } else {
    float mu = (t1 + t0) / 2.f, hw = (t1 - t0) / 2.f;
    float mu2 = mu * mu, hw2 = hw * hw, hw4 = hw2 * hw2;
    float den = 3.f * mu2 + hw2;
    t_mean = mu + (2.f * mu * hw2) / den;
    t_var = hw / 3.f - (4.f / 15.f) * ((hw4 * (12.f * mu2 - hw2)) / (den * den));
    r_var = rad * rad * (mu2 / 4.f + (5.f / 12.f) * hw2 - 4.f / 15.f * hw4 / den);
  }
  const float dsq[3] = {ry[3] * ry[3], ry[4] * ry[4], ry[5] * ry[5]};
  const float magn = fmaxf((dsq[0] + dsq[1]) + dsq[2], 1e-10f);
  MipGauss g;
  g.m0 = ry[3] * t_mean + ry[0]; g.m1 = ry[4] * t_mean + ry[1]; g.m2 = ry[5] * t_mean + ry[2];
  g.c0 = t_var * dsq[0] + r_var * (1.f - dsq[0] / magn);
  g.c1 = t_var * dsq[1] + r_var * (1.f - dsq[1] / magn);
  g.c2 = t_var * dsq[2] + r_var * (1.f - dsq[2] / magn);
  return g;
}

// upper edge of the last sample interval: the caller's t_end, or -- t_end = NaN -- the intended closing 2 ts[T-1] - ts[T-2]
// (ts[T-1] + 1 for a single step) evaluated here in fp32 exactly like torch evaluates `2 * ts[-1] - ts[-2]`, so that the
// host does not have to read ts back (a device synchronisation per forward)
__device__ __forceinline__ float mip_last_edge(const float* ts, int T, float t_end) {
  if (t_end == t_end) return t_end;
  return T > 1 ? 2.0f * ts[T - 1] - ts[T - 2] : ts[T - 1] + 1.0f;
}

// pixel radius of ray (b, hq, wq) of a [B,H,W,6] crop (src/utils.py:77-81: difference of neighbouring rows' directions;
// the appended last row repeats the second-to-last difference)
__device__ __forceinline__ float mip_radius(const float* rays, int H, int W, int b, int hq, int wq) {
  int h0 = hq < H - 1 ? hq : H - 3;
  if (h0 < 0) h0 = 0;
  const float* ra = rays + (((int64_t)b * H + h0) * W + wq) * 6 + 3;
  const float* rb = rays + (((int64_t)b * H + h0 + 1) * W + wq) * 6 + 3;
  float e0 = ra[0] - rb[0], e1 = ra[1] - rb[1], e2 = ra[2] - rb[2];
  return sqrtf((e0 * e0 + e1 * e1) + e2 * e2) * 2.0f / 3.4641016151377544f;
}

// feature f of the 6*nd-wide IPE row [sin(mean 2^k) damped | sin(mean 2^k + pi/2) damped], k-major, axis-minor
// (src/utils.py:23-27).  Branch-free and libm-free: y = mean * 2^k is an exact scaling, so its revolution count is
// p = mean * (2^k / 2pi) with the product's rounding error recovered by an fma (two-term constant), the integer part
// drops out exactly and the sine sees an angle in [-pi, pi] for every degree (the reference's fp32 sin(y) reduces
// y = 2^15 mean the same way inside libm).  The cosine half is sin(fl(y + pi/2)) like the reference: the rounded sum
// differs from y by an exactly representable delta, which is added to the reduced angle.  damp = exp(-cov 4^k / 2)
// through v_exp_f32.  Max deviation from libm's sinf / expf: 6e-7 (angle) and 2 ulp (damp).  FAST (bf16 operands
// downstream): hardware v_sin_f32 on the revolution count instead of the polynomial.
template <bool FAST = false>
__device__ __forceinline__ float mip_feature(float m0, float m1, float m2, float c0, float c1, float c2, int f, int nd,
                                             int min_deg) {
  const int part = f >= 3 * nd;            // 0: sin(y), 1: sin(y + pi/2)
  const int rem = f - part * 3 * nd;
  const int k = (rem * 43) >> 7;           // rem / 3 for rem < 128
  const int a = rem - 3 * k;
  // (the six moments arrive as scalars: selects over members of a struct in memory become an indexed scratch load)
  const float m = a == 0 ? m0 : (a == 1 ? m1 : m2);
  const float c = a == 0 ? c0 : (a == 1 ? c1 : c2);
  const int deg = min_deg + k;
  const float chi = ldexpf(0.15915494309189535f, deg), clo = ldexpf(6.4206383e-9f, deg);  // 1/2pi = chi + clo
  const float p = m * chi;
  const float e = fmaf(m, chi, -p) + m * clo;
  const float rev = (p - rintf(p)) + e;
  const float y = ldexpf(m, deg);
  const float yc = y + 1.5707963267948966f;
  const float delta = part ? yc - y : 0.f;    // exact: yc ~ y
  const float damp = __builtin_amdgcn_exp2f(c * ldexpf(-0.7213475204444817f, 2 * deg));  // exp(-0.5 c 4^deg)
  if constexpr (FAST) return damp * __builtin_amdgcn_sinf(rev + delta * 0.15915494309189535f);
  return damp * sin_cw(6.283185307179586f * rev + delta);
}

// cos on the same reduction: even Taylor polynomial to r^10 on [-pi/2,pi/2] (max |err| 5e-7), sign by parity.
__device__ __forceinline__ float cos_cw(float x) {
  float q = rintf(x * 0.318309886183790672f);
  float r = fmaf(q, -3.140625f, x);
  r = fmaf(q, -9.67502593994140625e-4f, r);
  r = fmaf(q, -1.509957990978376432e-7f, r);
  float r2 = r * r;
  float p = fmaf(r2, -2.7557319224e-07f, 2.4801587302e-05f);
  p = fmaf(p, r2, -1.3888888889e-03f);
  p = fmaf(p, r2, 4.1666666667e-02f);
  p = fmaf(p, r2, -0.5f);
  float c = fmaf(p, r2, 1.0f);
  int qi = (int)q;
  return (qi & 1) ? -c : c;
}

// sin and cos on one shared reduction (Fourier-feature prologue: 2 x 128 of them per sample)
__device__ __forceinline__ void sincos_cw(float x, float& sn, float& cs) {
  float q = rintf(x * 0.318309886183790672f);
  float r = fmaf(q, -3.140625f, x);
  r = fmaf(q, -9.67502593994140625e-4f, r);
  r = fmaf(q, -1.509957990978376432e-7f, r);
  float r2 = r * r;
  float p = fmaf(r2, 2.5962193818e-06f, -1.9804804431e-04f);
  p = fmaf(p, r2, 8.3329907333e-03f);
  p = fmaf(p, r2, -1.6666655917e-01f);
  float s = fmaf(p * r2, r, r);
  float c = fmaf(r2, -2.7557319224e-07f, 2.4801587302e-05f);
  c = fmaf(c, r2, -1.3888888889e-03f);
  c = fmaf(c, r2, 4.1666666667e-02f);
  c = fmaf(c, r2, -0.5f);
  c = fmaf(c, r2, 1.0f);
  const bool odd = ((int)q) & 1;
  sn = odd ? -s : s;
  cs = odd ? -c : c;
}

// Reference hash (src/neural_blocks.py:135-139,166): ((x*1) ^ (y*2654435761) ^ (z*805459861)) mod 2^16
// in int64 with a non-negative remainder.  Only the low 16 bits survive the mod and the low bits of
// a two's-complement product/xor depend only on the low bits of the operands, so uint32 arithmetic
// reproduces the int64 result bit for bit.
__device__ __forceinline__ uint32_t hash_index(int lx, int ly, int lz) {
  return (((uint32_t)lx) ^ ((uint32_t)ly * 2654435761u) ^ ((uint32_t)lz * 805459861u)) & 0xFFFFu;
}

// dir_to_elev_azim (src/utils.py:247-254)
__device__ __forceinline__ void elev_azim(float dx, float dy, float dz, float& elev, float& azim) {
  const float lim = 1.f - 1e-6f;
  float nrm = sqrtf(dx * dx + dy * dy + dz * dz);
  nrm = fmaxf(nrm, 1e-12f);
  float x = fminf(fmaxf(dx / nrm, -lim), lim);
  float y = fminf(fmaxf(dy / nrm, -lim), lim);
  float z = fminf(fmaxf(dz / nrm, -lim), lim);
  elev = acosf(z);
  azim = atan2f(y, x);
}

}  // namespace na
