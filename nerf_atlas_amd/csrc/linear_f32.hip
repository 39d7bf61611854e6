// Exact-fp32 Linear with fused pre-activation and skip-concat input:
//     y[N,out] = W[out,in] . act([x0 | x1])[N,in] + b            (src/neural_blocks.py:288-296)
// on the f32-input matrix core (v_mfma_f32_32x32x2_f32: bitwise an fp32 fma chain, 157 TFLOP/s peak).
// This is the any-shape path (hidden 64/128 heads, training-time parity checks); the 256-wide hot MLPs
// run in mlp_fused.hip.
#include "common.h"

namespace na {

typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int LBM = 64;   // rows (samples) per workgroup
constexpr int LBN = 64;   // output columns per workgroup
constexpr int LBK = 16;   // K chunk
constexpr int LLD = LBK + 1;

__device__ __forceinline__ float act_in(float v, int act) {
  if (act == NA_ACT_LEAKY_RELU) return leaky_relu(v);
  if (act == NA_ACT_SIN) return sinf(v);
  return v;
}

__global__ __launch_bounds__(256) void linear_f32_kernel(const float* __restrict__ x0, int in0,
                                                         const float* __restrict__ x1, int in1, int64_t N,
                                                         const float* __restrict__ W, const float* __restrict__ b,
                                                         int out, int act, float* __restrict__ y) {
  __shared__ float Xs[LBM * LLD];
  __shared__ float Ws[LBN * LLD];
  const int in = in0 + in1;
  const int64_t n0 = (int64_t)blockIdx.x * LBM;
  const int o0 = blockIdx.y * LBN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;  // 2x2 waves, each a 32x32 output tile
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  const int lrow = tid >> 2, lk = (tid & 3) * 4;
  for (int k0 = 0; k0 < in; k0 += LBK) {
    // stage X (activated) and W chunks
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      int k = k0 + lk + q;
      int64_t n = n0 + lrow;
      float xv = 0.f, wv = 0.f;
      if (k < in) {
        if (n < N) {
          float raw = k < in0 ? x0[n * in0 + k] : x1[n * in1 + (k - in0)];
          xv = act_in(raw, act);
        }
        int o = o0 + lrow;
        if (o < out) wv = W[(int64_t)o * in + k];
      }
      Xs[lrow * LLD + lk + q] = xv;
      Ws[lrow * LLD + lk + q] = wv;
    }
    __syncthreads();
    const float* xa = Xs + (wr * 32 + (lane & 31)) * LLD + (lane >> 5);
    const float* wb = Ws + (wc * 32 + (lane & 31)) * LLD + (lane >> 5);
#pragma unroll
    for (int kk = 0; kk < LBK; kk += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[kk], wb[kk], acc, 0, 0, 0);
    __syncthreads();
  }
  const int j = o0 + wc * 32 + (lane & 31);
  if (j < out) {
    float bj = b != nullptr ? b[j] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int64_t n = n0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (n < N) y[n * out + j] = acc[r] + bj;
    }
  }
}

}  // namespace na

extern "C" int na_linear_f32(const float* x0, int in0, const float* x1, int in1, int64_t N, const float* W,
                             const float* b, int out, int pre_act, float* y, void* stream) {
  using namespace na;
  NA_REQUIRE(x0 && W && y, NA_ENULL, "na_linear_f32: null pointer");
  NA_REQUIRE(in0 >= 1 && in1 >= 0 && out >= 1 && N >= 0, NA_EINVAL, "na_linear_f32: bad shape in0=%d in1=%d out=%d", in0,
             in1, out);
  NA_REQUIRE(in1 == 0 || x1 != nullptr, NA_ENULL, "na_linear_f32: in1>0 needs x1");
  NA_REQUIRE(pre_act >= NA_ACT_NONE && pre_act <= NA_ACT_SIN, NA_EUNSUPPORTED, "na_linear_f32: activation %d", pre_act);
  if (N == 0) return NA_OK;
  int64_t gx = (N + LBM - 1) / LBM;
  NA_REQUIRE(gx < (1ll << 31), NA_EINVAL, "na_linear_f32: N too large");
  dim3 grid((unsigned)gx, (unsigned)((out + LBN - 1) / LBN));
  hipLaunchKernelGGL(linear_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, x0, in0, x1, in1, N, W, b, out, pre_act,
                     y);
  return check_launch("na_linear_f32");
}
