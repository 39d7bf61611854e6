// PROTOTYPE (not part of libnerf_atlas_amd.so): layer-synchronous MLP forward.  See DESIGN.md section 10, item 1.
//
// A workgroup (4 waves, one per SIMD; variant with 8 waves x 32 rows: ls_mlp_w8.hip) takes S = 128 samples through the network one layer at a time.  Wave w keeps rows 64w..64w+63 (two 32-row tiles: every B fragment read from LDS feeds TWO MFMAs) of
// the current layer's weights in registers (16 MFMA A fragments fetched with plain global loads -- no LDS-DMA), the
// activations of all 128 samples live in two LDS buffers (bf16, [sample][feature], 528-byte pitch), every wave reads all
// of them as MFMA B fragments and writes its 32 output features back with ds_write_b64: ONE barrier per layer.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#ifndef LS_ABLATE
#define LS_ABLATE 0  // 1: no weight refetch per layer, 2: no epilogue/stores, 4: no barrier per layer
#endif
constexpr int S = 128;          // samples per pass
constexpr int PITCH = 528;      // bytes per sample row in LDS (256 bf16 + 16 pad)
constexpr int H = 256;

__device__ __forceinline__ float leaky(float v) { return __builtin_amdgcn_fmed3f(v, v * 0.01f, 3.0e38f); }

// Packed weights: layer l, wave w, chunk c -> 1 KiB fragment (lane-major 16 B).  Hidden layers: 8 waves x 16 chunks.
// init layer (K = 16): 8 waves x 1 chunk.  out layer (32 rows): 1 wave x 16 chunks.  Biases fp32 per layer [rows].
struct Args {
  const char* w_init;   // [8][1] KiB
  const char* w_hid;    // [L][8][16] KiB
  const char* w_out;    // [16] KiB
  const float* b_init;  // [256]
  const float* b_hid;   // [L][256]
  const float* b_out;   // [32]
  const float* x;       // [N,16]
  float* y;             // [N,32]
  int64_t N;
  int L;
  int npass;
};

__device__ __forceinline__ f32x16 bias_acc(const float* b, int row0, int lane) {
  f32x16 a;
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = b[row0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)];
  return a;
}

// activation + bf16 + store the wave's 32 features of 32 samples (block b) into the LDS buffer
__device__ __forceinline__ void store_block(char* buf, const f32x16& acc, int b, int wave, int lane) {
  char* row = buf + (b * 32 + (lane & 31)) * PITCH + (wave * 32 + 4 * (lane >> 5)) * 2;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    bf16x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (__bf16)leaky(acc[4 * q + e]);
    *(bf16x4*)(row + q * 16) = v;
  }
}

__global__ __launch_bounds__(256) void ls_mlp_kernel(Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* buf0 = smem;
  char* buf1 = smem + S * PITCH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int boff = (lane & 31) * PITCH + (lane >> 5) * 16;

  for (int pass = blockIdx.x; pass < a.npass; pass += gridDim.x) {
    const int64_t s0 = (int64_t)pass * S;
    // ---- stage the 16 input features of the 128 samples (fp32 -> bf16) into buf0[:, 0..15]
    {
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const int s = (tid >> 2) + 64 * rr, q = tid & 3;  // 128 samples x 4 float4
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (s0 + s < a.N) v = *(const f32x4*)(a.x + (s0 + s) * 16 + q * 4);
        bf16x4 h;
#pragma unroll
        for (int e = 0; e < 4; ++e) h[e] = (__bf16)v[e];
        *(bf16x4*)(buf0 + s * PITCH + q * 8) = h;
      }
    }
    // first hidden layer's weights in flight during the init layer
    bf16x8 A[2][16];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const char* wp = a.w_hid + ((size_t)(2 * wave + t) * 16) * 1024 + lane * 16;
#pragma unroll
      for (int c = 0; c < 16; ++c) A[t][c] = *(const bf16x8*)(wp + c * 1024);
    }
    __syncthreads();
    // ---- init layer: K = 16
    {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const bf16x8 Ai = *(const bf16x8*)(a.w_init + (size_t)(2 * wave + t) * 1024 + lane * 16);
        const f32x16 bias = bias_acc(a.b_init, (2 * wave + t) * 32, lane);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const bf16x8 B = *(const bf16x8*)(buf0 + b * 32 * PITCH + boff);
          f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ai, B, bias, 0, 0, 0);
          store_block(buf1, acc, b, 2 * wave + t, lane);
        }
      }
    }
    __syncthreads();
    // ---- hidden layers
    char* cur = buf1;
    char* nxt = buf0;
    for (int l = 0; l < a.L; ++l) {
      f32x16 acc[2][4];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const f32x16 bias = bias_acc(a.b_hid + l * H, (2 * wave + t) * 32, lane);
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[t][b] = bias;
      }
      bf16x8 An[2][16];
      const bool more = l + 1 < a.L;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const char* wp = more ? a.w_hid + (((size_t)(l + 1) * 8 + 2 * wave + t) * 16) * 1024 + lane * 16 : a.w_out + lane * 16;
#pragma unroll
        for (int c = 0; c < 16; ++c) An[t][c] = (LS_ABLATE & 1) ? A[t][c] : *(const bf16x8*)(wp + c * 1024);
      }
      constexpr int PF = 4;
      bf16x8 Bq[PF];
      auto bsrc = [&](int i) { return (const bf16x8*)(cur + (i >> 4) * 32 * PITCH + boff + (i & 15) * 32); };
#pragma unroll
      for (int i = 0; i < PF; ++i) Bq[i] = *bsrc(i);
      uint32_t pk[2][8];
#pragma unroll
      for (int i = 0; i < 64; ++i) {
        const int b = i >> 4, c = i & 15;
        const bf16x8 B = Bq[i % PF];
        if (i + PF < 64) Bq[i % PF] = *bsrc(i + PF);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          acc[t][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[t][c], B, acc[t][b], 0, 0, 0);
          if (!(LS_ABLATE & 2) && b > 0 && t == 0) {
            const int u = c >> 1, tt = (c & 1);  // dword u of tile tt, block b-1: one step per B fragment
            typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
            bf16x2 v;
            if (LS_ABLATE & 16) { pk[tt][u] = (uint32_t)lane; } else {
            v[0] = (__bf16)leaky(acc[tt][b - 1][2 * u]);
            v[1] = (__bf16)leaky(acc[tt][b - 1][2 * u + 1]);
            pk[tt][u] = __builtin_bit_cast(uint32_t, v); }
            asm volatile("" : "+v"(pk[tt][u]));
            if ((u & 1) && !(LS_ABLATE & 8)) {
              typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
              u32x2 w = {pk[tt][u - 1], pk[tt][u]};
              *(u32x2*)(nxt + ((b - 1) * 32 + (lane & 31)) * PITCH + ((2 * wave + tt) * 32 + 4 * (lane >> 5)) * 2 + (u >> 1) * 16) = w;
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (!(LS_ABLATE & 2)) {
        store_block(nxt, acc[0][3], 3, 2 * wave, lane);
        store_block(nxt, acc[1][3], 3, 2 * wave + 1, lane);
      } else if (acc[0][0][0] + acc[1][1][0] + acc[0][2][0] + acc[1][3][0] == 1.2345f) nxt[tid] = 1;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int c = 0; c < 16; ++c) A[t][c] = An[t][c];
      if (!(LS_ABLATE & 4)) __syncthreads();
      char* tswap = cur; cur = nxt; nxt = tswap;
    }
    // ---- out layer (32 rows): wave b handles sample block b (waves 0..3), all with the same 16 fragments
    if (wave < 4) {
      // A currently holds the out-layer fragments only for ... every wave fetched w_out in the last iteration
      const f32x16 bias = bias_acc(a.b_out, 0, lane);
      f32x16 acc = bias;
      const int b = wave;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const bf16x8 B = *(const bf16x8*)(cur + b * 32 * PITCH + boff + c * 32);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0][c], B, acc, 0, 0, 0);
      }
      const int64_t s = s0 + b * 32 + (lane & 31);
      if (s < a.N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) a.y[s * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)] = acc[r];
      }
    }
    __syncthreads();
  }
}

extern "C" int ls_mlp_forward(const void* w_init, const void* w_hid, const void* w_out, const float* b_init,
                              const float* b_hid, const float* b_out, const float* x, float* y, int64_t N, int L,
                              void* stream) {
  Args a;
  a.w_init = (const char*)w_init; a.w_hid = (const char*)w_hid; a.w_out = (const char*)w_out;
  a.b_init = b_init; a.b_hid = b_hid; a.b_out = b_out; a.x = x; a.y = y; a.N = N; a.L = L;
  a.npass = (int)((N + S - 1) / S);
  const int lds = 2 * S * PITCH;
  static bool done = false;
  if (!done) {
    if (hipFuncSetAttribute((const void*)ls_mlp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -1;
    done = true;
  }
  int grid = a.npass < 256 ? a.npass : 256;
  hipLaunchKernelGGL(ls_mlp_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
