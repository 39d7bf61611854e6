// Backward operators of the hot path (the training step of runner.py:609-850 differentiates
// exactly these): activation, exact-fp32 Linear weight gradient (the split-bf16 GEMMs live in train_gemm.hip), hash-table scatter, colour-head activations and
// alpha compositing.  fp32 throughout; matrix work on the exact f32 MFMA.  Gradients are pinned against
// torch.autograd of the CPU oracle (tests/test_gpu_backward.py).
#include "common.h"
#include <cstdlib>
#include <cstring>

namespace na {

typedef __attribute__((ext_vector_type(16))) float f32x16;

// ------------------------------------------------------------------------------------ activations
__device__ __forceinline__ float act_fwd(float v, int act) {
  if (act == NA_ACT_LEAKY_RELU) return leaky_relu(v);
  if (act == NA_ACT_SIN) return sinf(v);
  return v;
}
__device__ __forceinline__ float act_grad(float v, int act) {
  if (act == NA_ACT_LEAKY_RELU) return v > 0.f ? 1.f : 0.01f;
  if (act == NA_ACT_SIN) return cosf(v);
  return 1.f;
}

// g_x = g_act * act'(x)   (x = the pre-activation input of a Linear, src/neural_blocks.py:293)
__global__ void act_backward_kernel(const float* __restrict__ x, const float* __restrict__ g, int64_t n, int act,
                                    float* __restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = g[i] * act_grad(x[i], act);
}

// ------------------------------------------------------------------------------------ forward-mode tangents
// SDF normals d sdf / d x (src/sdf.py:43,108) are propagated FORWARD through the MLP, three tangent rows per sample
// next to the value row: t_{l+1} = W_l . (act'(z_l) * t_l).  That keeps the eikonal term (runner.py:685-692) a
// first-order graph of Linear / multiply / act' nodes, so its gradient w.r.t. the weights needs no double backward.
__device__ __forceinline__ float act_deriv(float v, int act, int order) {  // order 1: act', order 2: act''
  if (act == NA_ACT_SIN) return order == 1 ? cosf(v) : -sinf(v);
  if (act == NA_ACT_LEAKY_RELU) return order == 1 ? (v > 0.f ? 1.f : 0.01f) : 0.f;
  return order == 1 ? 1.f : 0.f;
}
__global__ void act_deriv_kernel(const float* __restrict__ x, int64_t n, int act, int order, float* __restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = act_deriv(x[i], act, order);
}
// out[j, i] = a[i] * b[j, i]   (a [n], b / out [J, n]: one multiplier row shared by the J tangent rows)
__global__ void mul_bcast_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n, int J,
                                 float* __restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n * J; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = a[i % n] * b[i];
}
// out[i] = sum_j g[j, i] * b[j, i]   (gradient of mul_bcast w.r.t. the shared row; j summed in order)
__global__ void mul_reduce_kernel(const float* __restrict__ g, const float* __restrict__ b, int64_t n, int J,
                                  float* __restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int j = 0; j < J; ++j) s = s + g[(int64_t)j * n + i] * b[(int64_t)j * n + i];
    out[i] = s;
  }
}
// eikonal_loss(normals) = mean((|n| - 1)^2) over N rows of 3 (src/utils.py:31); n stored [3, N] (tangent-major)
__global__ __launch_bounds__(256) void eikonal_kernel(const float* __restrict__ nrm, int64_t N, float inv_n,
                                                      float* __restrict__ loss, long long* __restrict__ fix) {
  __shared__ float part[4];
  float acc = 0.f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
    const float x = nrm[i], y = nrm[N + i], z = nrm[2 * N + i];
    const float d = sqrtf((x * x + y * y) + z * z) - 1.f;
    acc += d * d;
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) accumulate(loss, fix, 0, ((part[0] + part[1]) + (part[2] + part[3])) * inv_n);
}
// d loss / d n = g * 2 (|n| - 1) / |n| * n / N
__global__ void eikonal_backward_kernel(const float* __restrict__ nrm, int64_t N, const float* __restrict__ g,
                                        float inv_n, float* __restrict__ g_n) {
  const float gs = g[0];
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
    const float x = nrm[i], y = nrm[N + i], z = nrm[2 * N + i];
    const float len = sqrtf((x * x + y * y) + z * z);
    const float c = len > 0.f ? gs * 2.f * (len - 1.f) / len * inv_n : 0.f;
    g_n[i] = c * x; g_n[N + i] = c * y; g_n[2 * N + i] = c * z;
  }
}

__device__ __forceinline__ float sigmoid_kind_grad(float v, int kind) {
  const float s = sigmoidf_(v);
  switch (kind) {
    case NA_SIG_NORMAL: return s * (1.f - s);
    case NA_SIG_THIN: return s * (1.f - s) * (1.f + 2.f * -1e-2f);
    case NA_SIG_FAT: return s * (1.f - s) * (1.f + 2.f * 1e-2f);
    case NA_SIG_TANH: { float t = tanhf(v); return 1.f - t * t; }
    case NA_SIG_UPSHIFTED: return s * (1.f - s);
    case NA_SIG_RELU: return v > 0.f ? 1.f : 0.f;
    case NA_SIG_SIN: return cosf(v);
    case NA_SIG_LEAKY_RELU: return v > 0.f ? 1.f : 0.01f;
    case NA_SIG_UPSHIFTED_SOFTPLUS: return s;
    case NA_SIG_UPSHIFTED_RELU: return v > 0.f ? 1.f : 0.f;
    case NA_SIG_CYCLIC: return cosf(v / 5.f) / 5.f / 2.f * (1.f + 2.f * -1e-2f);
    default: return 1.f;
  }
}

__global__ void sigmoid_backward_kernel(const float* __restrict__ x, const float* __restrict__ g, int64_t n, int kind,
                                        float* __restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = g[i] * sigmoid_kind_grad(x[i], kind);
}

// ------------------------------------------------------------------------------------ Linear weight gradient
// dW[out,in] += dY^T[out,N] . act([x0|x1])[N,in],  db[out] += sum_n dY[n,:]
// Workgroup = 64 (out) x 64 (in) tile over one slice of N; 4 waves, each a 32x32 tile on v_mfma_f32_32x32x2_f32
// (A[i=o][k=n], B[k=n][j=in]).  Slices are combined with fp32 atomics.
constexpr int WG_O = 64, WG_I = 64, WG_N = 16, WG_LD = 65;

__global__ __launch_bounds__(256) void linear_wgrad_kernel(const float* __restrict__ x0, int in0,
                                                           const float* __restrict__ x1, int in1, int64_t N,
                                                           const float* __restrict__ dY, int out, int act,
                                                           int64_t slice, float* __restrict__ dW,
                                                           float* __restrict__ db, long long* __restrict__ fixW,
                                                           long long* __restrict__ fixb) {
  __shared__ float Ys[WG_N * WG_LD];  // [n][o]
  __shared__ float Xs[WG_N * WG_LD];  // [n][k]
  const int in = in0 + in1;
  const int o0 = blockIdx.x * WG_O, k0 = blockIdx.y * WG_I;
  const int64_t n_begin = (int64_t)blockIdx.z * slice;
  const int64_t n_end = n_begin + slice < N ? n_begin + slice : N;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wo = wave >> 1, wk = wave & 1;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float bsum = 0.f;  // thread t < 64 sums column o0 + t of dY (only blocks with blockIdx.y == 0)
  const int ln = tid >> 4, lc = (tid & 15) * 4;  // this thread stages row ln, columns lc..lc+3 of both tiles
  for (int64_t nb = n_begin; nb < n_end; nb += WG_N) {
    const int64_t n = nb + ln;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float yv = 0.f, xv = 0.f;
      if (n < n_end) {
        const int o = o0 + lc + q, k = k0 + lc + q;
        if (o < out) yv = dY[n * out + o];
        if (k < in) {
          const float raw = k < in0 ? x0[n * in0 + k] : x1[n * in1 + (k - in0)];
          xv = act_fwd(raw, act);
        }
      }
      Ys[ln * WG_LD + lc + q] = yv;
      Xs[ln * WG_LD + lc + q] = xv;
    }
    __syncthreads();
    if (db != nullptr && blockIdx.y == 0 && tid < WG_O) {
#pragma unroll
      for (int r = 0; r < WG_N; ++r) bsum += Ys[r * WG_LD + tid];
    }
    const float* ya = Ys + (lane >> 5) * WG_LD + wo * 32 + (lane & 31);
    const float* xb = Xs + (lane >> 5) * WG_LD + wk * 32 + (lane & 31);
#pragma unroll
    for (int kk = 0; kk < WG_N; kk += 2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ya[kk * WG_LD], xb[kk * WG_LD], acc, 0, 0, 0);
    __syncthreads();
  }
  const int k = k0 + wk * 32 + (lane & 31);
  if (k < in) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = o0 + wo * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (o < out) accumulate(dW, fixW, (int64_t)o * in + k, acc[r]);
    }
  }
  if (db != nullptr && blockIdx.y == 0 && tid < WG_O && o0 + tid < out) accumulate(db, fixb, o0 + tid, bsum);
}

// ------------------------------------------------------------------------------------ hash encoder backward
// tables_grad[lvl][idx][:] += w_corner * g_feat[n, lvl, :]   (src/neural_blocks.py:166,190: gradient of the gather)
//
// The grids are coarse (16 ... 6.3 cells per unit) and consecutive samples are neighbouring pixels at one depth, so
// most of a wave lands in the same few cells: plain per-lane atomics serialise on a handful of addresses (measured
// 10.2 ms for 262 144 samples).  One wave handles 64 consecutive samples of ONE level; per corner the lanes that share
// a table row are combined first (leader loop: readfirstlane -> match mask -> wave reduction) and one lane adds the
// 4 sums.
// wave total on DPP (row shifts inside rows of 16, row_bcast:15 / :31 across them): 6 VALU instructions instead of the 6
// dependent ds_bpermute round trips of a __shfl_xor butterfly (402 -> 170 us for the kernel below); the total arrives in lane 63 only
#define NA_DPP_ADD(X, CTRL, ROWS) \
  X += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, X), CTRL, ROWS, 0xF, false))
__device__ __forceinline__ float wave_total_lane63(float x) {
  NA_DPP_ADD(x, 0x111, 0xF);
  NA_DPP_ADD(x, 0x112, 0xF);
  NA_DPP_ADD(x, 0x114, 0xF);
  NA_DPP_ADD(x, 0x118, 0xF);
  NA_DPP_ADD(x, 0x142, 0xA);
  NA_DPP_ADD(x, 0x143, 0xC);
  return x;
}

// Round 3: the wave leaders no longer go to memory.  A workgroup takes a CONTIGUOUS run of samples of one level and sums
// into a small table in LDS (1024 direct-mapped rows keyed by the table index; a row taken by another index falls through
// to the global atomic), flushed once at the end: the 4096 waves of a coarse level used to queue on the same few dozen
// addresses in L2 (613 us for 262 144 samples; per-lane atomics 8.9 ms).  In deterministic mode the LDS rows hold the
// same 2^-40 fixed-point integers as the global accumulator, so the result stays independent of the order.
#ifndef HB_MAXWG
#define HB_MAXWG 512  // workgroups per level (tools/hash_bwd_run.py, N = 262 144: 128 -> 173.9 us, 256 -> 157.0, 512 -> 144.6, 1024 -> 146.6; 64 -> 280)
#endif
#ifndef HB_ABLATE
#define HB_ABLATE 0   // timing experiments: 1 no main loop, 2 no flush of the LDS table, 4 no LDS adds (leader rounds only)
#endif
constexpr int HB_ROWS = 1024;
__global__ __launch_bounds__(256) void hash_backward_kernel(const float* __restrict__ x, int64_t N,
                                                            const float* __restrict__ g_out, int include_input,
                                                            HashRes res, float* __restrict__ tables_grad,
                                                            long long* __restrict__ fix,
                                                            const float* __restrict__ tangent, int g_ld = 0, int g_col0 = -1) {
  // tangent != nullptr: g_out is the gradient of the directional derivative J(x).e of hash_jvp_kernel, whose corner
  // weights are N_l * <grad w_corner, e> instead of w_corner
  __shared__ uint32_t tags[HB_ROWS];
  __shared__ unsigned long long vals[HB_ROWS * 4];  // fp32 sums (low word) or fixed-point sums
  const int odim = 32 + 3 * include_input;
  const int lvl = blockIdx.y;
  const float Nl = res.n[lvl];
  float* tab = tables_grad + (int64_t)lvl * 65536 * 4;
  long long* ftab = fix != nullptr ? fix + (int64_t)lvl * 65536 * 4 : nullptr;
  for (int i = threadIdx.x; i < HB_ROWS; i += 256) tags[i] = 0xffffffffu;
  for (int i = threadIdx.x; i < HB_ROWS * 4; i += 256) vals[i] = 0ull;
  __syncthreads();
  // one row of four sums; `who` is the only lane of its wave that calls this for the row right now
  auto add_row = [&](uint32_t id, float s0, float s1, float s2, float s3) {
    const uint32_t row = id & (HB_ROWS - 1);
    const uint32_t old = atomicCAS(&tags[row], 0xffffffffu, id);
    const float sv[4] = {s0, s1, s2, s3};
    if (old == 0xffffffffu || old == id) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (ftab == nullptr) atomicAdd((float*)&vals[row * 4 + e], sv[e]);
        else if (fabsf(sv[e]) < 1048576.0f) atomicAdd(&vals[row * 4 + e], (unsigned long long)__float2ll_rn(sv[e] * kFixScale));
        else accumulate(tab, ftab, (int64_t)id * 4 + e, sv[e]);  // (out of the fixed-point range, NaN: the fp32 path, as accumulate() does)
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) accumulate(tab, ftab, (int64_t)id * 4 + e, sv[e]);
    }
  };
  const int64_t per = ((N + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;  // samples of this workgroup: whole rounds
  const int64_t n_lo = blockIdx.x * per, n_hi = n_lo + per < N ? n_lo + per : N;
  for (int64_t base = n_lo; base < ((HB_ABLATE & 1) ? n_lo : n_hi); base += 256) {
    const int64_t n = base + threadIdx.x;
    const bool live = n < n_hi;
    float wx = 0.f, wy = 0.f, wz = 0.f, g0 = 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f;
    float ex = 0.f, ey = 0.f, ez = 0.f;
    int lx = 0, ly = 0, lz = 0;
    if (live && tangent != nullptr) {
      ex = tangent[n * 3] * Nl; ey = tangent[n * 3 + 1] * Nl; ez = tangent[n * 3 + 2] * Nl;
    }
    if (live) {
      const float vx = x[n * 3] * Nl, vy = x[n * 3 + 1] * Nl, vz = x[n * 3 + 2] * Nl;
      const float fx = floorf(vx), fy = floorf(vy), fz = floorf(vz);
      lx = (int)fx; ly = (int)fy; lz = (int)fz;
      wx = vx - fx; wy = vy - fy; wz = vz - fz;
      // (g_col0 >= 0: the features' gradient sits in columns g_col0 .. g_col0 + 31 of rows of pitch g_ld -- a slice of a wider
      // gradient, e.g. of a network's init rows, read in place)
      const float* g = g_col0 >= 0 ? g_out + n * g_ld + g_col0 + lvl * 4 : g_out + n * odim + 3 * include_input + lvl * 4;
      g0 = g[0]; g1 = g[1]; g2 = g[2]; g3 = g[3];
    }
    const float iwx = 1.f - wx, iwy = 1.f - wy, iwz = 1.f - wz;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const uint32_t id = hash_index(lx + ((c >> 2) & 1), ly + ((c >> 1) & 1), lz + (c & 1));
      const float ux = ((c >> 2) & 1) ? wx : iwx, uy = ((c >> 1) & 1) ? wy : iwy, uz = (c & 1) ? wz : iwz;
      float w = ux * uy * uz;
      if (tangent != nullptr) {
        const float sx = ((c >> 2) & 1) ? ex : -ex, sy = ((c >> 1) & 1) ? ey : -ey, sz = (c & 1) ? ez : -ez;
        w = (sx * (uy * uz) + sy * (ux * uz)) + sz * (ux * uy);
      }
      const float a0 = w * g0, a1 = w * g1, a2 = w * g2, a3 = w * g3;
      bool todo = live;
      // at most 4 leader rounds (covers the common case of a wave straddling a cell face), then lane by lane
      for (int round = 0; round < 4; ++round) {
        const uint64_t pending = __ballot(todo);
        if (pending == 0) break;
        const int leader = __ffsll((unsigned long long)pending) - 1;
        const uint32_t lid = __shfl(id, leader);
        const bool mine = todo && id == lid;
        const float s0 = wave_total_lane63(mine ? a0 : 0.f), s1 = wave_total_lane63(mine ? a1 : 0.f);
        const float s2 = wave_total_lane63(mine ? a2 : 0.f), s3 = wave_total_lane63(mine ? a3 : 0.f);
        if ((threadIdx.x & 63) == 63 && !((HB_ABLATE & 4) && s0 != 1.2345f)) add_row(lid, s0, s1, s2, s3);
        todo = todo && !mine;
      }
      if (todo) add_row(id, a0, a1, a2, a3);
    }
  }
  __syncthreads();
  for (int row = threadIdx.x; row < ((HB_ABLATE & 2) ? 0 : HB_ROWS); row += 256) {
    const uint32_t id = tags[row];
    if (id == 0xffffffffu) continue;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const unsigned long long v = vals[row * 4 + e];
      if (ftab == nullptr) atomicAdd(tab + (int64_t)id * 4 + e, __uint_as_float((uint32_t)v));
      else atomicAdd((unsigned long long*)(ftab + (int64_t)id * 4 + e), v);
    }
  }
}

// d(hash features)/d(position): the trilinear weights are linear in each fractional coordinate, floor() has zero
// gradient (torch.autograd of src/neural_blocks.py:166-190 sees exactly this), so
//   g_x[a] = g_in[a] (include_input) + sum_lvl N_lvl * sum_corner dW_corner/dw_a * <emb_corner, g_lvl>.
// One thread per (sample, level); the 8 levels of a sample sit in 8 adjacent lanes and are summed with DPP-free
// shuffles before a single store.
// g_ld > 0: rows of pitch g_ld = [`lead` more copies of x | x (include_input) | features] -- the gradient of a network's init rows
// read in place; the leading copies' gradient is added last (the order autograd's accumulation of the two slices had)
__global__ void hash_backward_input_kernel(const float* __restrict__ x, int64_t N, const float* __restrict__ tables,
                                           const float* __restrict__ g_out_, int include_input, HashRes res,
                                           float* __restrict__ g_x, int g_ld = 0, int lead = 0) {
  const int odim = g_ld > 0 ? g_ld : 32 + 3 * include_input;
  const float* __restrict__ g_out = g_out_ + 3 * lead;
  const int64_t total = (N * 8 + 63) / 64 * 64;  // whole waves, so the shuffles below are convergent
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int lvl = (int)(i & 7);
    const int64_t n = i >> 3;
    float gx = 0.f, gy = 0.f, gz = 0.f;
    if (n < N) {
      const float px = x[n * 3], py = x[n * 3 + 1], pz = x[n * 3 + 2];
      const float Nl = res.n[lvl];
      const float vx = px * Nl, vy = py * Nl, vz = pz * Nl;
      const float fx = floorf(vx), fy = floorf(vy), fz = floorf(vz);
      const int lx = (int)fx, ly = (int)fy, lz = (int)fz;
      const float wx = vx - fx, wy = vy - fy, wz = vz - fz;
      const float iwx = 1.f - wx, iwy = 1.f - wy, iwz = 1.f - wz;
      const float* g = g_out + n * odim + 3 * include_input + lvl * 4;
      const float g0 = g[0], g1 = g[1], g2 = g[2], g3 = g[3];
      const float* tab = tables + (int64_t)lvl * 65536 * 4;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int bx = (c >> 2) & 1, by = (c >> 1) & 1, bz = c & 1;
        const uint32_t id = hash_index(lx + bx, ly + by, lz + bz);
        const float4 e = *(const float4*)(tab + (int64_t)id * 4);
        const float d = ((e.x * g0 + e.y * g1) + e.z * g2) + e.w * g3;
        const float sx = bx ? 1.f : -1.f, sy = by ? 1.f : -1.f, sz = bz ? 1.f : -1.f;
        const float ux = bx ? wx : iwx, uy = by ? wy : iwy, uz = bz ? wz : iwz;
        gx += d * (sx * uy * uz);
        gy += d * (ux * sy * uz);
        gz += d * (ux * uy * sz);
      }
      gx *= Nl; gy *= Nl; gz *= Nl;
    }
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) {
      gx += __shfl_xor(gx, m);
      gy += __shfl_xor(gy, m);
      gz += __shfl_xor(gz, m);
    }
    if (lvl == 0 && n < N) {
      if (include_input) {
        gx += g_out[n * odim]; gy += g_out[n * odim + 1]; gz += g_out[n * odim + 2];
      }
      if (lead) {
        gx = g_out_[n * odim] + gx; gy = g_out_[n * odim + 1] + gy; gz = g_out_[n * odim + 2] + gz;
      }
      g_x[n * 3] = gx; g_x[n * 3 + 1] = gy; g_x[n * 3 + 2] = gz;
    }
  }
}

// Directional derivative of the hash features along e (forward mode; the reference's FFJORD estimate
// src/utils.py:467-478 obtains e^T J e with a vector-Jacobian product, which is the same number):
//   t[n, 0:3] = e (include_input),   t[n, lvl, :] = N_lvl * sum_corner <grad w_corner, e> * emb_corner.
// One thread per (sample, level), like hash_backward_input_kernel.
__global__ void hash_jvp_kernel(const float* __restrict__ x, int64_t N, const float* __restrict__ tables,
                                const float* __restrict__ tangent, int include_input, HashRes res,
                                float* __restrict__ t_out) {
  const int odim = 32 + 3 * include_input;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N * 8; i += (int64_t)gridDim.x * blockDim.x) {
    const int lvl = (int)(i & 7);
    const int64_t n = i >> 3;
    const float Nl = res.n[lvl];
    const float vx = x[n * 3] * Nl, vy = x[n * 3 + 1] * Nl, vz = x[n * 3 + 2] * Nl;
    const float fx = floorf(vx), fy = floorf(vy), fz = floorf(vz);
    const int lx = (int)fx, ly = (int)fy, lz = (int)fz;
    const float wx = vx - fx, wy = vy - fy, wz = vz - fz;
    const float iwx = 1.f - wx, iwy = 1.f - wy, iwz = 1.f - wz;
    const float e0 = tangent[n * 3], e1 = tangent[n * 3 + 1], e2 = tangent[n * 3 + 2];
    const float ex = e0 * Nl, ey = e1 * Nl, ez = e2 * Nl;
    const float* tab = tables + (int64_t)lvl * 65536 * 4;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int bx = (c >> 2) & 1, by = (c >> 1) & 1, bz = c & 1;
      const uint32_t id = hash_index(lx + bx, ly + by, lz + bz);
      const float4 emb = *(const float4*)(tab + (int64_t)id * 4);
      const float ux = bx ? wx : iwx, uy = by ? wy : iwy, uz = bz ? wz : iwz;
      const float sx = bx ? ex : -ex, sy = by ? ey : -ey, sz = bz ? ez : -ez;
      const float w = (sx * (uy * uz) + sy * (ux * uz)) + sz * (ux * uy);
      a0 += w * emb.x; a1 += w * emb.y; a2 += w * emb.z; a3 += w * emb.w;
    }
    float* o = t_out + n * odim + 3 * include_input + lvl * 4;
    o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3;
    if (lvl == 0 && include_input) {
      t_out[n * odim] = e0; t_out[n * odim + 1] = e1; t_out[n * odim + 2] = e2;
    }
  }
}

// FFJORD divergence estimate of the rigid deformation field (runner.py:697-700, src/utils.py:467-478) from the
// deformation network's outputs `est` = [rigidity | control points] and their directional derivatives `tan` along e:
//   rig = sigmoid(z0/2), d rig = rig (1-rig)/2 * t0, dp = sum_k B_k(t) P_k, d dp = sum_k B_k(t) tP_k,
//   div = <e, d dp * rig + dp * d rig>  (= e^T J e),   out[n] = div.
__global__ void ffjord_div_kernel(const float* __restrict__ est, const float* __restrict__ tan, int stride,
                                  const float* __restrict__ tt, const float* __restrict__ e, int64_t N, int n,
                                  float* __restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
    const float* z = est + i * stride;
    const float* dz = tan + i * stride;
    const float rig = sigmoidf_(z[0] / 2.f);
    const float drig = rig * (1.f - rig) * 0.5f * dz[0];
    const float t = tt[i], m1t = 1.f - t;
    float B[8];
    B[0] = 1.f;
    for (int it = 1; it < n; ++it) {
      B[it] = B[it - 1] * t;
      for (int k = it - 1; k >= 1; --k) B[k] = B[k] * m1t + B[k - 1] * t;
      B[0] *= m1t;
    }
    float div = 0.f;
    for (int a = 0; a < 3; ++a) {
      float dp = 0.f, ddp = 0.f;
      for (int k = 0; k < n; ++k) { dp += B[k] * z[1 + 3 * k + a]; ddp += B[k] * dz[1 + 3 * k + a]; }
      div += e[i * 3 + a] * (ddp * rig + dp * drig);
    }
    out[i] = div;
  }
}

// ------------------------------------------------------------------------------------ VolSDF Laplace density
// density = cdf(s)/beta, s = -sdf/beta (src/utils.py:50-58, src/nerf.py:985-990); pdf = exp(-|s|)/2.
//   d/dsdf = -pdf/beta^2,   d/dbeta = -cdf/beta^2 + pdf*sdf/beta^3   (block-reduced, one atomic per workgroup)
__global__ __launch_bounds__(256) void laplace_density_backward_kernel(const float* __restrict__ sdf, int64_t N,
                                                                       const float* __restrict__ beta,
                                                                       const float* __restrict__ g,
                                                                       float* __restrict__ g_sdf,
                                                                       float* __restrict__ g_beta,
                                                                       long long* __restrict__ fix) {
  __shared__ float part[4];
  const float sc = beta[0];
  const float r = 1.0f / sc, r2 = r * r;
  float acc = 0.f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = sdf[i], gi = g[i];
    const float s = -v / sc;
    const float e = expf(-fabsf(s)) * 0.5f;
    const float cdf = s <= 0.f ? e : 1.f - e;
    g_sdf[i] = -gi * e * r2;
    acc += gi * (e * v * r2 * r - cdf * r2);
  }
  if (g_beta == nullptr) return;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) accumulate(g_beta, fix, 0, (part[0] + part[1]) + (part[2] + part[3]));
}

// ------------------------------------------------------------------------------------ spline warp
// Forward (src/nerf.py:1173-1178,1201-1206,1267-1278): rig = sigmoid(e0/2), dp = sum_k B_k(t) P_k,
// out = pts + dp*rig.  Bezier curves are linear in their control points with the Bernstein weights B_k(t).
__global__ void bezier_warp_backward_kernel(const float* __restrict__ est, int est_stride,
                                            const float* __restrict__ tt, int64_t N, int n,
                                            const float* __restrict__ g_pts, const float* __restrict__ g_dp,
                                            const float* __restrict__ g_rig, float* __restrict__ g_est, int n_rl = 0,
                                            const float* __restrict__ g_enc = nullptr) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
    const float* e = est + i * est_stride;
    float* ge = g_est + i * est_stride;
    const float rig = sigmoidf_(e[0] / 2.f);
    const float t = tt[i], m1t = 1.f - t;
    float B[8];
    B[0] = 1.f;
    for (int it = 1; it < n; ++it) {
      B[it] = B[it - 1] * t;
      for (int k = it - 1; k >= 1; --k) B[k] = B[k] * m1t + B[k - 1] * t;
      B[0] *= m1t;
    }
    float dp[3] = {0.f, 0.f, 0.f};
    for (int k = 0; k < n; ++k)
      for (int a = 0; a < 3; ++a) dp[a] += B[k] * e[1 + 3 * k + a];
    float d_rig = g_rig != nullptr ? g_rig[i] : 0.f;
    float d_dp[3];
    for (int a = 0; a < 3; ++a) {
      const float gw = g_pts != nullptr ? g_pts[i * 3 + a] : 0.f;
      d_rig += gw * dp[a];
      d_dp[a] = gw * rig + (g_dp != nullptr ? g_dp[i * 3 + a] : 0.f);
    }
    ge[0] = d_rig * rig * (1.f - rig) * 0.5f;
    for (int k = 0; k < n; ++k)
      for (int a = 0; a < 3; ++a) ge[1 + 3 * k + a] = B[k] * d_dp[a];
    int c0 = 1 + 3 * n;
    if (n_rl > 0 && g_enc != nullptr) {
      // enc_j = (sum_k B_k(t) C_kj) * s, s = sigmoid(est[3n + 1])  (src/nerf.py:1272-1278)
      const float* cp = e + 3 * n + 2;
      const float s = sigmoidf_(e[3 * n + 1]);
      float d_s = 0.f;
      for (int j = 0; j < n_rl; ++j) {
        const float g = g_enc[i * n_rl + j];
        float v = 0.f;
        for (int k = 0; k < n; ++k) {
          v += B[k] * cp[k * n_rl + j];
          ge[3 * n + 2 + k * n_rl + j] = B[k] * (g * s);
        }
        d_s += g * v;
      }
      ge[3 * n + 1] = d_s * s * (1.f - s);
      c0 = 2 + (3 + n_rl) * n;
    }
    for (int c = c0; c < est_stride; ++c) ge[c] = 0.f;
  }
}

// gradient of na_plain_head_rows: g_first_out[n] = [g_density[n] | g_rows[n, 5:]], g_pts[n] = g_rows[n, :3] (nullable)
template <typename I>
__global__ void plain_head_rows_backward_kernel(const float* __restrict__ g_density, const float* __restrict__ g_rows, int64_t N, int C,
                                                float* __restrict__ g_first_out, float* __restrict__ g_pts) {
  const I W = 1 + C, total = (I)N * W;
  for (I i = blockIdx.x * (I)blockDim.x + threadIdx.x; i < total; i += (I)gridDim.x * blockDim.x) {
    const I n = i / W;
    const int c = (int)(i - n * W);
    g_first_out[i] = c == 0 ? (g_density != nullptr ? g_density[n] : 0.f) : g_rows[n * (5 + C) + 4 + c];
    if (g_pts != nullptr && i < (I)N * 3) {
      const I m = i / 3;
      g_pts[i] = g_rows[m * (5 + C) + (i - m * 3)];
    }
  }
}

// ------------------------------------------------------------------------------------ Adam (round 6)
// torch.optim.Adam's default (foreach) update -- runner.py:448-458 builds optim.Adam(params, lr, eps = 1e-7, weight_decay) -- is
// seven multi-tensor launches per step (lerp, mul, addcmul, sqrt, div, add, addcdiv: ~0.13 ms for PlainNeRF's 2.9 M parameters, a
// read-modify-write of the same 35 MB each).  Here ONE launch over all tensors with the SAME per-element operations in the same
// order, each rounded like the corresponding ATen kernel rounds it -- which of them contract a multiply-add (the ATen kernels are
// compiled with contraction on) is the `fma` mask, pinned bit for bit against torch by tests/test_gpu_train.py:
//   m = m + w (g - m)                  w = 1 - beta1              (bit 0: one fma)
//   v = v beta2;  v = v + c (g g)      c = 1 - beta2              (bit 1: one fma; g g rounded first -- tools/adam_probe.py)
//   d = sqrt(v) / sqrt(1 - beta2^t) + eps
//   p = p + a (m / d)                  a = -lr / (1 - beta1^t)     (bit 2: one fma)
// The scalars arrive as the doubles torch's Python computes and are rounded to float like ATen's opmath conversion does.
constexpr int kAdamMany = 48;
struct AdamMany {
  float* p[kAdamMany];
  const float* g[kAdamMany];
  float* m[kAdamMany];
  float* v[kAdamMany];
  int first[kAdamMany + 1];   // first block of tensor e (a block = 1024 elements)
  int64_t numel[kAdamMany];
  int n, fma;
  float w, beta2, c, bc2_sqrt, eps, a;
};
__global__ __launch_bounds__(256) void adam_many_kernel(AdamMany t) {
  int e = 0;
  while (e + 1 < t.n && (int)blockIdx.x >= t.first[e + 1]) ++e;   // (<= 48 entries: a scalar walk)
  const int64_t i0 = ((int64_t)blockIdx.x - t.first[e]) * 1024 + threadIdx.x * 4;
  const int64_t n = t.numel[e];
  float* __restrict__ P = t.p[e];
  const float* __restrict__ G = t.g[e];
  float* __restrict__ M = t.m[e];
  float* __restrict__ V = t.v[e];
  auto one = [&](float& p, float g, float& m, float& v) __attribute__((always_inline)) {
    const float diff = g - m;
    m = (t.fma & 1) ? fmaf(t.w, diff, m) : m + t.w * diff;
    v = v * t.beta2;
    const float gg = g * g;
    v = (t.fma & 2) ? fmaf(t.c, gg, v) : v + t.c * gg;
    const float d = sqrtf(v) / t.bc2_sqrt + t.eps;
    const float q = m / d;
    p = (t.fma & 4) ? fmaf(t.a, q, p) : p + t.a * q;
  };
  if (i0 + 4 <= n && ((((uintptr_t)P | (uintptr_t)G | (uintptr_t)M | (uintptr_t)V) & 15) == 0)) {
    float4 p = *(const float4*)(P + i0), m = *(const float4*)(M + i0), v = *(const float4*)(V + i0);
    const float4 g = *(const float4*)(G + i0);
    one(p.x, g.x, m.x, v.x); one(p.y, g.y, m.y, v.y); one(p.z, g.z, m.z, v.z); one(p.w, g.w, m.w, v.w);
    *(float4*)(P + i0) = p; *(float4*)(M + i0) = m; *(float4*)(V + i0) = v;
  } else {
    for (int64_t i = i0; i < i0 + 4 && i < n; ++i) {
      float p = P[i], m = M[i], v = V[i];
      one(p, G[i], m, v);
      P[i] = p; M[i] = m; V[i] = v;
    }
  }
}

// ------------------------------------------------------------------------------------ compositing backward
// Forward (src/nerf.py:60-80,96-98): a_t = 1-exp(-sigma_t*dist_t), f_t = (1-a_t)+1e-10, T_t = prod_{s<t} f_s,
// w_t = a_t*T_t, out_c = sum_t w_t*c_tc + sky.  With G_t = dL/dw_t = sum_c g_c*c_tc (- sum_c g_c for the white sky,
// t < T-1):  dL/da_t = G_t*T_t - (sum_{s>t} G_s*w_s)/f_t,  dL/dc_tc = w_t*g_c.
// One thread per ray: forward sweep writes T_t into g_density (used as scratch), backward sweep consumes it.
template <int C>
__global__ void composite_backward_kernel(const float* __restrict__ density, const float* __restrict__ feat,
                                          const float* __restrict__ ts, const float* __restrict__ rays, int T,
                                          int64_t R, int density_kind, int bg_kind, const float* __restrict__ g_out,
                                          float* __restrict__ g_density, float* __restrict__ g_feat,
                                          const float* __restrict__ sky_rand = nullptr) {
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < R; r += (int64_t)gridDim.x * blockDim.x) {
    const float* ry = rays + r * 6 + 3;
    const float nrm = sqrtf((ry[0] * ry[0] + ry[1] * ry[1]) + ry[2] * ry[2]);
    float g[C];
    float gsum = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) { g[c] = g_out[r * C + c]; gsum += g[c]; }
    // d sky / d w_t (t < T-1): -1 per channel for white, -rand[r] for the random background (src/nerf.py:98,101-103)
    const float gsky = bg_kind == NA_BG_WHITE ? gsum : bg_kind == NA_BG_RANDOM ? gsum * sky_rand[r] : 0.f;
    // Both sweeps are dependent chains over T; a step that waits for its own loads costs one memory latency (round 5: 94 us for
    // 4 096 rays x 64 steps, one thread per ray = 64 waves on the chip).  Rows are fetched U = 8 steps at a time, like the forward
    // kernel does -- 8 (first sweep) / 8 x (2 + C) (second) independent loads in flight per thread; the arithmetic and its order
    // are unchanged (bit-identical gradients).
    constexpr int U = 8;
    float trans = 1.0f;
    for (int t0 = 0; t0 < T; t0 += U) {
      float dv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) dv[u] = density[(int64_t)(t0 + u < T ? t0 + u : T - 1) * R + r];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int t = t0 + u;
        if (t < T) {
          const float d = dv[u];
          const float sigma = density_kind == NA_DENSITY_SOFTPLUS_M1 ? softplusf_(d - 1.0f) : fmaxf(d, 0.f);
          float dist = t < T - 1 ? fmaxf(ts[t + 1] - ts[t], 1e-5f) : 1e10f;
          dist *= nrm;
          const float a = 1.0f - expf(-sigma * dist);
          g_density[(int64_t)t * R + r] = trans;  // scratch: T_t
          trans = trans * ((1.0f - a) + 1e-10f);
        }
      }
    }
    float suffix = 0.f;  // sum_{s>t} G_s * w_s
    for (int t1 = T - 1; t1 >= 0; t1 -= U) {
      float dv[U], tv[U], cv[U][C];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int t = t1 - u >= 0 ? t1 - u : 0;  // (clamped: the head re-reads row 0 and ignores it)
        dv[u] = density[(int64_t)t * R + r];
        tv[u] = g_density[(int64_t)t * R + r];
        const float* ct = feat + ((int64_t)t * R + r) * C;
#pragma unroll
        for (int c = 0; c < C; ++c) cv[u][c] = ct[c];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int t = t1 - u;
        if (t >= 0) {
          const float d = dv[u];
          const float sigma = density_kind == NA_DENSITY_SOFTPLUS_M1 ? softplusf_(d - 1.0f) : fmaxf(d, 0.f);
          float dist = t < T - 1 ? fmaxf(ts[t + 1] - ts[t], 1e-5f) : 1e10f;
          dist *= nrm;
          const float e = expf(-sigma * dist);
          const float a = 1.0f - e;
          const float f = (1.0f - a) + 1e-10f;
          const float Tt = tv[u];
          const float w = a * Tt;
          float G = 0.f;
#pragma unroll
          for (int c = 0; c < C; ++c) {
            G += g[c] * cv[u][c];
            g_feat[((int64_t)t * R + r) * C + c] = w * g[c];
          }
          if (t < T - 1) G -= gsky;
          const float dLda = G * Tt - suffix / f;
          suffix += G * w;
          const float dsig = density_kind == NA_DENSITY_SOFTPLUS_M1 ? sigmoidf_(d - 1.0f) : (d > 0.f ? 1.f : 0.f);
          g_density[(int64_t)t * R + r] = dLda * dist * e * dsig;
        }
      }
    }
  }
}

// The same gradients with the T steps of a ray split into segments of CB_SEG (round 6): one thread per (ray, segment), a workgroup =
// 64 consecutive rays x all S segments (wave s = segment s: every load and store stays coalesced along the rays).  The chain of a
// thread is CB_SEG steps instead of T and there are S times as many waves (4 096 rays x 64 steps: 256 instead of 64; the one-thread-
// per-ray kernel takes 63 us whether the batch has 4 096 or 16 384 rays -- it is bound by its dependent exp / log chain per step).
// Segment-local prefix products / suffix sums meet in LDS: T_t = (prod of the earlier segments' products) * (local prefix), the suffix
// sum likewise -- the same terms in another association, so the last bits differ from the sequential kernel (tests: both against the
// oracle's autograd).
constexpr int CB_SEG = 16;
template <int C>
__global__ __launch_bounds__(512) void composite_backward_seg_kernel(const float* __restrict__ density, const float* __restrict__ feat,
                                              const float* __restrict__ ts, const float* __restrict__ rays, int T,
                                              int64_t R, int density_kind, int bg_kind, const float* __restrict__ g_out,
                                              float* __restrict__ g_density, float* __restrict__ g_feat,
                                              const float* __restrict__ sky_rand) {
  extern __shared__ float cb_lds[];  // [2][S][64]
  const int tx = threadIdx.x, seg = threadIdx.y, S = blockDim.y;
  const int64_t r = (int64_t)blockIdx.x * 64 + tx;
  const bool live = r < R;
  const int64_t rc = live ? r : R - 1;
  const int t_lo = seg * CB_SEG;
  const float* ry = rays + rc * 6 + 3;
  const float nrm = sqrtf((ry[0] * ry[0] + ry[1] * ry[1]) + ry[2] * ry[2]);
  float g[C];
  float gsum = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c) { g[c] = g_out[rc * C + c]; gsum += g[c]; }
  const float gsky = bg_kind == NA_BG_WHITE ? gsum : bg_kind == NA_BG_RANDOM ? gsum * sky_rand[rc] : 0.f;
  float dv[CB_SEG], cv[CB_SEG][C];
#pragma unroll
  for (int u = 0; u < CB_SEG; ++u) {
    const int t = t_lo + u < T ? t_lo + u : T - 1;
    dv[u] = density[(int64_t)t * R + rc];
    const float* ct = feat + ((int64_t)t * R + rc) * C;
#pragma unroll
    for (int c = 0; c < C; ++c) cv[u][c] = ct[c];
  }
  float ev[CB_SEG], fv[CB_SEG], lp[CB_SEG], dist[CB_SEG];
  float run = 1.0f;
#pragma unroll
  for (int u = 0; u < CB_SEG; ++u) {
    const int t = t_lo + u;
    const float d = dv[u];
    const float sigma = density_kind == NA_DENSITY_SOFTPLUS_M1 ? softplusf_(d - 1.0f) : fmaxf(d, 0.f);
    float di = t < T - 1 ? fmaxf(ts[t + 1] - ts[t < T ? t : T - 1], 1e-5f) : 1e10f;
    di *= nrm;
    dist[u] = di;
    const float e = expf(-sigma * di);
    ev[u] = e;
    const float a = 1.0f - e;
    fv[u] = (1.0f - a) + 1e-10f;
    lp[u] = run;
    if (t < T) run = run * fv[u];
  }
  float* P = cb_lds;
  float* Q = cb_lds + S * 64;
  P[seg * 64 + tx] = run;
  __syncthreads();
  float t_start = 1.0f;
  for (int s2 = 0; s2 < seg; ++s2) t_start = t_start * P[s2 * 64 + tx];
  float Gv[CB_SEG], Tv[CB_SEG], sl[CB_SEG];
  float run2 = 0.f;
#pragma unroll
  for (int u = CB_SEG - 1; u >= 0; --u) {
    const int t = t_lo + u;
    const float Tt = t_start * lp[u];
    const float w = (1.0f - ev[u]) * Tt;
    float G = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      G += g[c] * cv[u][c];
      if (live && t < T) g_feat[((int64_t)t * R + r) * C + c] = w * g[c];
    }
    if (t < T - 1) G -= gsky;
    Gv[u] = G; Tv[u] = Tt; sl[u] = run2;
    if (t < T) run2 += G * w;
  }
  Q[seg * 64 + tx] = run2;
  __syncthreads();
  float suffix0 = 0.f;
  for (int s2 = S - 1; s2 > seg; --s2) suffix0 += Q[s2 * 64 + tx];
#pragma unroll
  for (int u = 0; u < CB_SEG; ++u) {
    const int t = t_lo + u;
    if (live && t < T) {
      const float d = dv[u];
      const float dLda = Gv[u] * Tv[u] - (suffix0 + sl[u]) / fv[u];
      const float dsig = density_kind == NA_DENSITY_SOFTPLUS_M1 ? sigmoidf_(d - 1.0f) : (d > 0.f ? 1.f : 0.f);
      g_density[(int64_t)t * R + r] = dLda * dist[u] * ev[u] * dsig;
    }
  }
}

template <int C>
static void launch_composite_backward(const float* density, const float* feat, const float* ts, const float* rays, int T, int64_t R,
                                      int density_kind, int bg_kind, const float* g_out, float* g_density, float* g_feat,
                                      const float* sky_rand, hipStream_t stream) {
  static const bool seq = [] { const char* e = getenv("NA_COMPOSITE_BWD"); return e != nullptr && strcmp(e, "seq") == 0; }();
  const int S = (T + CB_SEG - 1) / CB_SEG;
  if (S >= 2 && S <= 8 && !seq) {  // (T <= 128: 512 threads, the register budget of a segment held in registers)
    hipLaunchKernelGGL(composite_backward_seg_kernel<C>, dim3((unsigned)((R + 63) / 64)), dim3(64, S), 2 * S * 64 * sizeof(float), stream,
                       density, feat, ts, rays, T, R, density_kind, bg_kind, g_out, g_density, g_feat, sky_rand);
  } else {
    dim3 g(grid_for(R, 64, 1 << 16)), b(64);  // (one wave per workgroup: 4 096 rays reach 64 CUs instead of 32)
    hipLaunchKernelGGL(composite_backward_kernel<C>, g, b, 0, stream, density, feat, ts, rays, T, R, density_kind, bg_kind, g_out,
                       g_density, g_feat, sky_rand);
  }
}

// out = (sigmoid(lin)/2 + 0.5) * pos[:, :C]  (src/refl.py:288-290): one thread per row.
//   g_lin[n]    = sum_c g[n,c] * pos[n,c] * s (1 - s) / 2
//   g_pos[n, c] = g[n,c] * (s/2 + 0.5) for c < C, 0 for the remaining (pass-through) columns
__global__ void pos_linear_combine_backward_kernel(const float* __restrict__ lin, const float* __restrict__ pos,
                                                   int64_t pos_ld, const float* __restrict__ g, int64_t N, int C,
                                                   float* __restrict__ g_lin, float* __restrict__ g_pos, int64_t gpos_ld) {
  for (int64_t n = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x) {
    const float s = sigmoidf_(lin[n]);
    const float scale = s / 2.f + 0.5f;
    float acc = 0.f;
    for (int c = 0; c < C; ++c) {
      const float gc = g[n * C + c];
      acc = acc + gc * pos[n * pos_ld + c];
      if (g_pos != nullptr) g_pos[n * gpos_ld + c] = gc * scale;
    }
    if (g_pos != nullptr)
      for (int64_t c = C; c < gpos_ld; ++c) g_pos[n * gpos_ld + c] = 0.f;
    if (g_lin != nullptr) g_lin[n] = acc * (s * (1.f - s) / 2.f);
  }
}

}  // namespace na

using namespace na;

extern "C" {

int na_act_backward(const float* x, const float* g, int64_t n, int act, float* out, void* stream) {
  if (n == 0) return NA_OK;  // empty: zero-size tensors carry null pointers
  NA_REQUIRE(x && g && out, NA_ENULL, "na_act_backward: null pointer");
  NA_REQUIRE(act >= NA_ACT_NONE && act <= NA_ACT_SIN, NA_EUNSUPPORTED, "na_act_backward: activation %d", act);
  if (n <= 0) return n == 0 ? NA_OK : NA_EINVAL;
  hipLaunchKernelGGL(act_backward_kernel, dim3(grid_for(n, 256, 16384)), dim3(256), 0, (hipStream_t)stream, x, g, n, act,
                     out);
  return check_launch("na_act_backward");
}

int na_sigmoid_backward(const float* x, const float* g, int64_t n, int kind, float* out, void* stream) {
  if (n == 0) return NA_OK;  // empty: zero-size tensors carry null pointers
  NA_REQUIRE(x && g && out, NA_ENULL, "na_sigmoid_backward: null pointer");
  NA_REQUIRE(kind >= 0 && kind <= NA_SIG_IDENTITY, NA_EUNSUPPORTED, "na_sigmoid_backward: kind %d", kind);
  if (n <= 0) return n == 0 ? NA_OK : NA_EINVAL;
  hipLaunchKernelGGL(sigmoid_backward_kernel, dim3(grid_for(n, 256, 16384)), dim3(256), 0, (hipStream_t)stream, x, g, n,
                     kind, out);
  return check_launch("na_sigmoid_backward");
}

int na_act_deriv(const float* x, int64_t n, int act, int order, float* out, void* stream) {
  if (n == 0) return NA_OK;
  NA_REQUIRE(x && out, NA_ENULL, "na_act_deriv: null pointer");
  NA_REQUIRE(n > 0 && (order == 1 || order == 2) && act >= NA_ACT_NONE && act <= NA_ACT_SIN, NA_EINVAL, "na_act_deriv: bad argument");
  hipLaunchKernelGGL(act_deriv_kernel, dim3(grid_for(n, 256, 16384)), dim3(256), 0, (hipStream_t)stream, x, n, act, order, out);
  return check_launch("na_act_deriv");
}

int na_mul_bcast(const float* a, const float* b, int64_t n, int J, float* out, void* stream) {
  if (n == 0 || J == 0) return NA_OK;
  NA_REQUIRE(a && b && out, NA_ENULL, "na_mul_bcast: null pointer");
  NA_REQUIRE(n > 0 && J > 0, NA_EINVAL, "na_mul_bcast: bad shape");
  hipLaunchKernelGGL(mul_bcast_kernel, dim3(grid_for(n * J, 256, 16384)), dim3(256), 0, (hipStream_t)stream, a, b, n, J, out);
  return check_launch("na_mul_bcast");
}

int na_mul_reduce(const float* g, const float* b, int64_t n, int J, float* out, void* stream) {
  if (n == 0) return NA_OK;
  NA_REQUIRE(g && b && out, NA_ENULL, "na_mul_reduce: null pointer");
  NA_REQUIRE(n > 0 && J > 0, NA_EINVAL, "na_mul_reduce: bad shape");
  hipLaunchKernelGGL(mul_reduce_kernel, dim3(grid_for(n, 256, 16384)), dim3(256), 0, (hipStream_t)stream, g, b, n, J, out);
  return check_launch("na_mul_reduce");
}

int na_eikonal_loss(const float* normals, int64_t N, float* loss, void* stream) {
  NA_REQUIRE(normals && loss, NA_ENULL, "na_eikonal_loss: null pointer");
  NA_REQUIRE(N > 0, NA_EINVAL, "na_eikonal_loss: N=%lld", (long long)N);
  int rc;
  long long* fix = det_begin(1, (hipStream_t)stream, "na_eikonal_loss", &rc);
  if (rc != NA_OK) return rc;
  hipLaunchKernelGGL(eikonal_kernel, dim3(grid_for(N, 256, 1024)), dim3(256), 0, (hipStream_t)stream, normals, N,
                     1.0f / (float)N, loss, fix);
  if (fix != nullptr) return det_finish(fix, 1, loss, (hipStream_t)stream, "na_eikonal_loss");
  return check_launch("na_eikonal_loss");
}

int na_eikonal_loss_backward(const float* normals, int64_t N, const float* g, float* g_normals, void* stream) {
  NA_REQUIRE(normals && g && g_normals, NA_ENULL, "na_eikonal_loss_backward: null pointer");
  NA_REQUIRE(N > 0, NA_EINVAL, "na_eikonal_loss_backward: N=%lld", (long long)N);
  hipLaunchKernelGGL(eikonal_backward_kernel, dim3(grid_for(N, 256, 4096)), dim3(256), 0, (hipStream_t)stream, normals, N, g,
                     1.0f / (float)N, g_normals);
  return check_launch("na_eikonal_loss_backward");
}

int na_pos_linear_combine_backward(const float* lin, const float* pos, int64_t pos_ld, const float* g, int64_t N, int C,
                                   float* g_lin, float* g_pos, int64_t gpos_ld, void* stream) {
  if (N == 0) return NA_OK;
  NA_REQUIRE(lin && pos && g, NA_ENULL, "na_pos_linear_combine_backward: null pointer");
  NA_REQUIRE(N > 0 && C >= 1 && pos_ld >= C && (g_pos == nullptr || gpos_ld >= C), NA_EINVAL,
             "na_pos_linear_combine_backward: bad shape");
  hipLaunchKernelGGL(pos_linear_combine_backward_kernel, dim3(grid_for(N, 256, 8192)), dim3(256), 0, (hipStream_t)stream,
                     lin, pos, pos_ld, g, N, C, g_lin, g_pos, gpos_ld);
  return check_launch("na_pos_linear_combine_backward");
}

int na_linear_wgrad(const float* x0, int in0, const float* x1, int in1, int64_t N, const float* dY, int out,
                    int pre_act, float* dW, float* db, void* stream) {
  NA_REQUIRE(x0 && dY && dW, NA_ENULL, "na_linear_wgrad: null pointer");
  NA_REQUIRE(in0 >= 1 && in1 >= 0 && out >= 1 && N >= 0, NA_EINVAL, "na_linear_wgrad: bad shape");
  NA_REQUIRE(in1 == 0 || x1 != nullptr, NA_ENULL, "na_linear_wgrad: in1>0 needs x1");
  NA_REQUIRE(pre_act >= NA_ACT_NONE && pre_act <= NA_ACT_SIN, NA_EUNSUPPORTED, "na_linear_wgrad: activation %d", pre_act);
  if (N == 0) return NA_OK;
  const int in = in0 + in1;
  // enough N-slices to fill the chip, each at least 1024 rows
  int64_t tiles = (int64_t)((out + WG_O - 1) / WG_O) * ((in + WG_I - 1) / WG_I);
  int64_t want = (2048 + tiles - 1) / tiles;
  int64_t slice = (N + want - 1) / want;
  if (slice < 1024) slice = 1024;
  slice = (slice + WG_N - 1) / WG_N * WG_N;
  int64_t nz = (N + slice - 1) / slice;
  NA_REQUIRE(nz <= 65535, NA_EINVAL, "na_linear_wgrad: N too large");
  dim3 grid((out + WG_O - 1) / WG_O, (in + WG_I - 1) / WG_I, (unsigned)nz);
  int rc;
  const size_t nW = (size_t)out * in, nfix = nW + (db ? out : 0);
  long long* fix = det_begin(nfix, (hipStream_t)stream, "na_linear_wgrad", &rc);
  if (rc != NA_OK) return rc;
  hipLaunchKernelGGL(linear_wgrad_kernel, grid, dim3(256), 0, (hipStream_t)stream, x0, in0, x1, in1, N, dY, out, pre_act,
                     slice, dW, db, fix, fix ? fix + nW : nullptr);
  if (fix != nullptr) {
    if ((rc = det_finish(fix, nW, dW, (hipStream_t)stream, "na_linear_wgrad")) != NA_OK) return rc;
    if (db != nullptr) return det_finish(fix + nW, (size_t)out, db, (hipStream_t)stream, "na_linear_wgrad");
    return NA_OK;
  }
  return check_launch("na_linear_wgrad");
}

int na_hash_encode_backward(const float* x, int64_t N, const float* g_out, int include_input, float* tables_grad,
                            void* stream) {
  if (N == 0) return NA_OK;  // empty: zero-size tensors carry null pointers
  NA_REQUIRE(x && g_out && tables_grad, NA_ENULL, "na_hash_encode_backward: null pointer");
  if (N <= 0) return N == 0 ? NA_OK : NA_EINVAL;
  int rc;
  const size_t ntab = (size_t)8 * 65536 * 4;
  long long* fix = det_begin(ntab, (hipStream_t)stream, "na_hash_encode_backward", &rc);
  if (rc != NA_OK) return rc;
  hipLaunchKernelGGL(hash_backward_kernel, dim3(grid_for(N, 256, HB_MAXWG), 8), dim3(256), 0, (hipStream_t)stream, x, N,
                     g_out, include_input ? 1 : 0, hash_resolutions(), tables_grad, fix, (const float*)nullptr);
  if (fix != nullptr) return det_finish(fix, ntab, tables_grad, (hipStream_t)stream, "na_hash_encode_backward");
  return check_launch("na_hash_encode_backward");
}

int na_hash_encode_backward_rows(const float* x, int64_t N, const float* g_rows, int g_ld, int g_col0, float* tables_grad,
                                 void* stream) {
  if (N == 0) return NA_OK;
  NA_REQUIRE(x && g_rows && tables_grad, NA_ENULL, "na_hash_encode_backward_rows: null pointer");
  NA_REQUIRE(N > 0 && g_col0 >= 0 && g_ld >= g_col0 + 32, NA_EINVAL, "na_hash_encode_backward_rows: N %lld g_ld %d g_col0 %d", (long long)N,
             g_ld, g_col0);
  int rc;
  const size_t ntab = (size_t)8 * 65536 * 4;
  long long* fix = det_begin(ntab, (hipStream_t)stream, "na_hash_encode_backward_rows", &rc);
  if (rc != NA_OK) return rc;
  hipLaunchKernelGGL(hash_backward_kernel, dim3(grid_for(N, 256, HB_MAXWG), 8), dim3(256), 0, (hipStream_t)stream, x, N,
                     g_rows, 0, hash_resolutions(), tables_grad, fix, (const float*)nullptr, g_ld, g_col0);
  if (fix != nullptr) return det_finish(fix, ntab, tables_grad, (hipStream_t)stream, "na_hash_encode_backward_rows");
  return check_launch("na_hash_encode_backward_rows");
}

int na_hash_encode_backward_input_rows(const float* x, int64_t N, const float* tables, const float* g_rows, int g_ld,
                                       int include_input, int lead, float* g_x, void* stream) {
  if (N == 0) return NA_OK;
  NA_REQUIRE(x && tables && g_rows && g_x, NA_ENULL, "na_hash_encode_backward_input_rows: null pointer");
  NA_REQUIRE(N > 0 && (lead == 0 || lead == 1) && g_ld >= 32 + 3 * ((include_input ? 1 : 0) + lead), NA_EINVAL,
             "na_hash_encode_backward_input_rows: N %lld g_ld %d lead %d", (long long)N, g_ld, lead);
  hipLaunchKernelGGL(hash_backward_input_kernel, dim3(grid_for(N * 8, 256, 16384)), dim3(256), 0, (hipStream_t)stream,
                     x, N, tables, g_rows, include_input ? 1 : 0, hash_resolutions(), g_x, g_ld, lead);
  return check_launch("na_hash_encode_backward_input_rows");
}

int na_plain_head_rows_backward(const float* g_density, const float* g_rows, int64_t N, int C, float* g_first_out, float* g_pts,
                                void* stream) {
  if (N == 0) return NA_OK;
  NA_REQUIRE(N > 0 && C >= 3, NA_EINVAL, "na_plain_head_rows_backward: N %lld C %d", (long long)N, C);
  NA_REQUIRE(g_rows && g_first_out, NA_ENULL, "na_plain_head_rows_backward: null pointer");
  const int64_t total = N * (1 + C);
  if (total < (1ll << 31))
    hipLaunchKernelGGL(plain_head_rows_backward_kernel<int>, dim3(grid_for(total, 256, 16384)), dim3(256), 0, (hipStream_t)stream,
                       g_density, g_rows, N, C, g_first_out, g_pts);
  else
    hipLaunchKernelGGL(plain_head_rows_backward_kernel<int64_t>, dim3(grid_for(total, 256, 16384)), dim3(256), 0, (hipStream_t)stream,
                       g_density, g_rows, N, C, g_first_out, g_pts);
  return check_launch("na_plain_head_rows_backward");
}

int na_adam_step(int n, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                 const int64_t* numel, double one_minus_beta1, double beta2, double one_minus_beta2, double bias_correction2_sqrt,
                 double eps, double neg_step_size, int fma_mask, void* stream) {
  NA_REQUIRE(n >= 0, NA_EINVAL, "na_adam_step: n %d", n);
  if (n == 0) return NA_OK;
  NA_REQUIRE(params && grads && exp_avg && exp_avg_sq && numel, NA_ENULL, "na_adam_step: null pointer");
  for (int base = 0; base < n; base += kAdamMany) {
    AdamMany t{};
    t.n = 0;
    int blocks = 0;
    for (int i = base; i < n && t.n < kAdamMany; ++i) {
      NA_REQUIRE(numel[i] >= 0, NA_EINVAL, "na_adam_step: numel[%d] = %lld", i, (long long)numel[i]);
      if (numel[i] == 0) continue;
      NA_REQUIRE(params[i] && grads[i] && exp_avg[i] && exp_avg_sq[i], NA_ENULL, "na_adam_step: tensor %d has a null pointer", i);
      const int e = t.n++;
      t.p[e] = params[i]; t.g[e] = grads[i]; t.m[e] = exp_avg[i]; t.v[e] = exp_avg_sq[i]; t.numel[e] = numel[i];
      t.first[e] = blocks;
      const int64_t nb = (numel[i] + 1023) / 1024;
      NA_REQUIRE(blocks + nb < (1ll << 30), NA_EINVAL, "na_adam_step: too many elements");
      blocks += (int)nb;
    }
    t.first[t.n] = blocks;
    if (t.n == 0) continue;
    t.fma = fma_mask;
    t.w = (float)one_minus_beta1; t.beta2 = (float)beta2; t.c = (float)one_minus_beta2; t.bc2_sqrt = (float)bias_correction2_sqrt;
    t.eps = (float)eps; t.a = (float)neg_step_size;
    hipLaunchKernelGGL(adam_many_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t);
  }
  return check_launch("na_adam_step");
}

int na_hash_encode_backward_input(const float* x, int64_t N, const float* tables, const float* g_out,
                                  int include_input, float* g_x, void* stream) {
  if (N == 0) return NA_OK;  // empty: zero-size tensors carry null pointers
  NA_REQUIRE(x && tables && g_out && g_x, NA_ENULL, "na_hash_encode_backward_input: null pointer");
  if (N <= 0) return N == 0 ? NA_OK : NA_EINVAL;
  hipLaunchKernelGGL(hash_backward_input_kernel, dim3(grid_for(N * 8, 256, 16384)), dim3(256), 0, (hipStream_t)stream,
                     x, N, tables, g_out, include_input ? 1 : 0, hash_resolutions(), g_x);
  return check_launch("na_hash_encode_backward_input");
}

int na_hash_encode_jvp(const float* x, int64_t N, const float* tables, const float* tangent, int include_input,
                       float* t_out, void* stream) {
  if (N == 0) return NA_OK;
  NA_REQUIRE(N > 0, NA_EINVAL, "na_hash_encode_jvp: N %lld", (long long)N);
  NA_REQUIRE(x && tables && tangent && t_out, NA_ENULL, "na_hash_encode_jvp: null pointer");
  hipLaunchKernelGGL(hash_jvp_kernel, dim3(grid_for(N * 8, 256, 16384)), dim3(256), 0, (hipStream_t)stream, x, N, tables,
                     tangent, include_input ? 1 : 0, hash_resolutions(), t_out);
  return check_launch("na_hash_encode_jvp");
}

int na_hash_encode_jvp_backward(const float* x, const float* tangent, int64_t N, const float* g_t, int include_input,
                                float* tables_grad, void* stream) {
  if (N == 0) return NA_OK;
  NA_REQUIRE(N > 0, NA_EINVAL, "na_hash_encode_jvp_backward: N %lld", (long long)N);
  NA_REQUIRE(x && tangent && g_t && tables_grad, NA_ENULL, "na_hash_encode_jvp_backward: null pointer");
  const int64_t ntab = 8LL * 65536 * 4;
  int rc;
  long long* fix = det_begin(ntab, (hipStream_t)stream, "na_hash_encode_jvp_backward", &rc);
  if (rc != NA_OK) return rc;
  hipLaunchKernelGGL(hash_backward_kernel, dim3(grid_for(N, 256, HB_MAXWG), 8), dim3(256), 0, (hipStream_t)stream, x, N,
                     g_t, include_input ? 1 : 0, hash_resolutions(), tables_grad, fix, tangent);
  if (fix != nullptr) return det_finish(fix, ntab, tables_grad, (hipStream_t)stream, "na_hash_encode_jvp_backward");
  return check_launch("na_hash_encode_jvp_backward");
}

int na_ffjord_div(const float* est, const float* est_tangent, int est_stride, const float* t, const float* e, int64_t N,
                  int n_ctrl, float* div, void* stream) {
  if (N == 0) return NA_OK;
  NA_REQUIRE(N > 0, NA_EINVAL, "na_ffjord_div: N %lld", (long long)N);
  NA_REQUIRE(est && est_tangent && t && e && div, NA_ENULL, "na_ffjord_div: null pointer");
  NA_REQUIRE(n_ctrl >= 2 && n_ctrl <= 8 && est_stride >= 1 + 3 * n_ctrl, NA_EINVAL,
             "na_ffjord_div: n_ctrl=%d (2..8) stride=%d", n_ctrl, est_stride);
  hipLaunchKernelGGL(ffjord_div_kernel, dim3(grid_for(N, 256, 8192)), dim3(256), 0, (hipStream_t)stream, est, est_tangent,
                     est_stride, t, e, N, n_ctrl, div);
  return check_launch("na_ffjord_div");
}

int na_laplace_density_backward(const float* sdf, int64_t N, const float* beta, const float* g, float* g_sdf,
                                float* g_beta, void* stream) {
  if (N == 0) return NA_OK;  // empty: zero-size tensors carry null pointers
  NA_REQUIRE(sdf && beta && g && g_sdf, NA_ENULL, "na_laplace_density_backward: null pointer");
  if (N <= 0) return N == 0 ? NA_OK : NA_EINVAL;
  int rc;
  long long* fix = g_beta != nullptr ? det_begin(1, (hipStream_t)stream, "na_laplace_density_backward", &rc) : (rc = NA_OK, nullptr);
  if (rc != NA_OK) return rc;
  hipLaunchKernelGGL(laplace_density_backward_kernel, dim3(grid_for(N, 256, 2048)), dim3(256), 0, (hipStream_t)stream,
                     sdf, N, beta, g, g_sdf, g_beta, fix);
  if (fix != nullptr) return det_finish(fix, 1, g_beta, (hipStream_t)stream, "na_laplace_density_backward");
  return check_launch("na_laplace_density_backward");
}

int na_bezier_warp_backward(const float* est, int est_stride, const float* t, int64_t N, int n_ctrl,
                            const float* g_out_pts, const float* g_dp, const float* g_rigidity, float* g_est,
                            void* stream) {
  if (N == 0) return NA_OK;  // empty: zero-size tensors carry null pointers
  NA_REQUIRE(est && t && g_est, NA_ENULL, "na_bezier_warp_backward: null pointer");
  NA_REQUIRE(n_ctrl >= 2 && n_ctrl <= 8 && est_stride >= 1 + 3 * n_ctrl, NA_EINVAL,
             "na_bezier_warp_backward: n_ctrl=%d (2..8) stride=%d", n_ctrl, est_stride);
  if (N <= 0) return N == 0 ? NA_OK : NA_EINVAL;
  hipLaunchKernelGGL(bezier_warp_backward_kernel, dim3(grid_for(N, 256, 8192)), dim3(256), 0, (hipStream_t)stream, est,
                     est_stride, t, N, n_ctrl, g_out_pts, g_dp, g_rigidity, g_est, 0, (const float*)nullptr);
  return check_launch("na_bezier_warp_backward");
}

int na_bezier_warp_latent_backward(const float* est, int est_stride, const float* t, int64_t N, int n_ctrl, int n_rl,
                                   const float* g_out_pts, const float* g_dp, const float* g_rigidity,
                                   const float* g_refl_latent, float* g_est, void* stream) {
  if (N == 0) return NA_OK;
  NA_REQUIRE(est && t && g_est, NA_ENULL, "na_bezier_warp_latent_backward: null pointer");
  NA_REQUIRE(n_ctrl >= 2 && n_ctrl <= 8 && n_rl >= 1 && n_rl <= 16 && est_stride >= 2 + (3 + n_rl) * n_ctrl, NA_EINVAL,
             "na_bezier_warp_latent_backward: n_ctrl=%d (2..8) n_rl=%d (1..16) stride=%d", n_ctrl, n_rl, est_stride);
  if (N <= 0) return N == 0 ? NA_OK : NA_EINVAL;
  hipLaunchKernelGGL(bezier_warp_backward_kernel, dim3(grid_for(N, 256, 8192)), dim3(256), 0, (hipStream_t)stream, est,
                     est_stride, t, N, n_ctrl, g_out_pts, g_dp, g_rigidity, g_est, n_rl, g_refl_latent);
  return check_launch("na_bezier_warp_latent_backward");
}

int na_composite_backward(const float* density, const float* feat, const float* ts, const float* rays, int T,
                          int64_t R, int C, int density_kind, int bg_kind, const float* g_out, float* g_density,
                          float* g_feat, void* stream) {
  NA_REQUIRE(density && feat && ts && rays && g_out && g_density && g_feat, NA_ENULL, "na_composite_backward: null pointer");
  NA_REQUIRE(T >= 1 && R >= 0, NA_EINVAL, "na_composite_backward: bad shape");
  NA_REQUIRE(C == 3 || C == 1, NA_EUNSUPPORTED, "na_composite_backward: C=%d (1 or 3)", C);
  if (R == 0) return NA_OK;
  if (C == 3) launch_composite_backward<3>(density, feat, ts, rays, T, R, density_kind, bg_kind, g_out, g_density, g_feat, nullptr, (hipStream_t)stream);
  else launch_composite_backward<1>(density, feat, ts, rays, T, R, density_kind, bg_kind, g_out, g_density, g_feat, nullptr, (hipStream_t)stream);
  return check_launch("na_composite_backward");
}

int na_composite_random_bg_backward(const float* density, const float* feat, const float* ts, const float* rays, int T,
                                    int64_t R, int C, int density_kind, const float* rand, const float* g_out,
                                    float* g_density, float* g_feat, void* stream) {
  NA_REQUIRE(density && feat && ts && rays && rand && g_out && g_density && g_feat, NA_ENULL,
             "na_composite_random_bg_backward: null pointer");
  NA_REQUIRE(T >= 1 && R >= 0, NA_EINVAL, "na_composite_random_bg_backward: bad shape");
  NA_REQUIRE(C == 3 || C == 1, NA_EUNSUPPORTED, "na_composite_random_bg_backward: C=%d (1 or 3)", C);
  if (R == 0) return NA_OK;
  if (C == 3) launch_composite_backward<3>(density, feat, ts, rays, T, R, density_kind, (int)NA_BG_RANDOM, g_out, g_density, g_feat, rand, (hipStream_t)stream);
  else launch_composite_backward<1>(density, feat, ts, rays, T, R, density_kind, (int)NA_BG_RANDOM, g_out, g_density, g_feat, rand, (hipStream_t)stream);
  return check_launch("na_composite_random_bg_backward");
}

}  // extern "C"
