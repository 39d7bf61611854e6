// PROTOTYPE (not part of libnerf_atlas_amd.so): layer-synchronous MLP forward, round-2 structure.
//
// 8 waves = 4 row groups (64 output rows = two 32-row tiles each) x 2 sample groups (SG = 32*NBLK samples each).
//  * activations live in LDS as ready-made MFMA B fragments, [group][block][chunk][lane] x 16 B (lane-linear: the
//    producer lane writes exactly the 16 B the consumer lane reads -> conflict-free ds_write_b128 / ds_read_b128; the
//    k permutation this implies is folded into the weight packing, like mlp_pack.hip does for the register engine);
//  * the MFMA phase of a layer is CHUNK-major: for every 16-wide k chunk the wave loads its two A fragments straight
//    from global memory into registers (software-prefetched PF chunks ahead, across layer boundaries: no LDS-DMA, no
//    weight ring, no exposed refetch) and feeds them to 2 x NBLK MFMAs whose B fragments come from LDS -- every LDS
//    fragment read feeds TWO MFMAs; all 2 x NBLK accumulator tiles of the layer stay in registers until the layer is
//    complete, then the activation epilogue overwrites the LDS fragments IN PLACE;
//  * the two sample groups run in ANTIPHASE (one workgroup barrier per phase): while group 0 streams MFMAs, group 1
//    runs its epilogue (VALU + ds_write) on the same SIMDs, and vice versa, so VALU work is never slipped between a
//    wave's own MFMAs and the matrix pipe always has one MFMA-only wave per SIMD.
//
// LS_ABLATE bits: 2 no epilogue math/stores, 4 lockstep groups (both in the same phase), 8 no weight loads (reuse),
// 16 no barriers (wrong), 32 no activation math, 64 no LDS stores (wrong)
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

#ifndef LS_ABLATE
#define LS_ABLATE 0
#endif
#ifndef NBLK
#define NBLK 4  // 32-sample blocks per sample group
#endif
#ifndef PF
#define PF 4    // weight prefetch depth in chunks
#endif
#ifndef BIAS_MODE
#define BIAS_MODE 0  // 0: bias loaded after the epilogue stores, 1: into its own registers before them, 2: constant
#endif
#ifndef BEARLY
#define BEARLY 0     // 1: the LDS reads of chunk c+1 are issued before the MFMAs of chunk c
#endif
constexpr int SG = 32 * NBLK;
constexpr int S = 2 * SG;  // samples per pass
constexpr int H = 256;

__device__ __forceinline__ float leaky(float v) { return __builtin_amdgcn_fmed3f(v, v * 0.01f, 3.0e38f); }

struct Args {
  const char* w_init;   // [8][1] KiB   (standard k order)
  const char* w_hid;    // [L+1][8][16] KiB (k order = producer register order, see ls_mlp.py pack_rows_perm); layer L,
                        // tile 0 = the out layer
  const char* w_out;    // unused
  const float* b_init;  // [256]
  const float* b_hid;   // [L][256]
  const float* b_out;   // [32]
  const float* x;       // [N,16]
  float* y;             // [N,32]
  int64_t N;
  int L;
  int npass;
  unsigned long long* trace;  // nullable: [2 waves][5 events][64 layers] s_memtime stamps of workgroup 0, pass 0
};

__device__ __forceinline__ f32x16 bias_acc(const float* b, int row0, int lane) {
  f32x16 a;
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = b[row0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)];
  return a;
}

// activation + bf16 of one accumulator tile -> the two B fragments (chunks 2*tile, 2*tile+1) of block b
__device__ __forceinline__ void store_tile(char* act_g, const f32x16& acc, int b, int tile, int lane) {
  uint32_t pk[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    bf16x2 v;
    v[0] = (__bf16)((LS_ABLATE & 32) ? acc[2 * u] : leaky(acc[2 * u]));
    v[1] = (__bf16)((LS_ABLATE & 32) ? acc[2 * u + 1] : leaky(acc[2 * u + 1]));
    pk[u] = __builtin_bit_cast(uint32_t, v);
  }
  char* dst = act_g + ((b * 16 + 2 * tile) * 64 + lane) * 16;
  if (LS_ABLATE & 64) {  // keep the VALU work, drop the LDS stores
#pragma unroll
    for (int u = 0; u < 8; ++u) asm volatile("" ::"v"(pk[u]));
    return;
  }
  *(u32x4*)dst = u32x4{pk[0], pk[1], pk[2], pk[3]};
  *(u32x4*)(dst + 1024) = u32x4{pk[4], pk[5], pk[6], pk[7]};
}

__global__ __launch_bounds__(512) void ls_mlp_kernel(Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rg = wv & 3, g = wv >> 2;
  char* act_g = smem + g * (NBLK * 16 * 1024);
  const bool lag = (LS_ABLATE & 4) ? false : (g == 1);  // group 1 runs one phase behind group 0

  // weight stream of this wave: layer l (the out layer is stored as layer L, tile 0), chunk c, tile t -> fragment
  // ((l*8 + 2rg+t)*16 + c) of ONE buffer; wave-uniform part in the scalar offset of a buffer load
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.w_hid, 0, (a.L + 1) * 8 * 16 * 1024, 0x00020000);
  const int wvoff = (2 * rg * 16) * 1024 + lane * 16;
  auto wload = [&](int l, int c, int t) -> bf16x8 {
    const int so = __builtin_amdgcn_readfirstlane((l * 8 + t) * 16 * 1024) + c * 1024;
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff, so, 0));
  };
  auto bload = [&](int b, int c) -> bf16x8 { return *(const bf16x8*)(act_g + ((b * 16 + c) * 64 + lane) * 16); };

  const bool tr = a.trace != nullptr && blockIdx.x == 0 && lane == 0 && rg == 0;
  auto stamp = [&](int ev, int l, int pass) {
    if (tr && pass == 0 && l < 64) a.trace[(g * 5 + ev) * 64 + l] = __builtin_amdgcn_s_memtime();
  };
  for (int pass = blockIdx.x; pass < a.npass; pass += gridDim.x) {
    const int64_t s0 = (int64_t)pass * S + g * SG;
    bf16x8 Aq[PF][2];
#pragma unroll
    for (int c = 0; c < PF; ++c)
#pragma unroll
      for (int t = 0; t < 2; ++t) Aq[c][t] = wload(0, c, t);

    f32x16 acc[2][NBLK];
    // ---- init layer (K = 16): B fragments straight from global memory
    {
      bf16x8 Ai[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) Ai[t] = *(const bf16x8*)(a.w_init + (size_t)(2 * rg + t) * 1024 + lane * 16);
#pragma unroll
      for (int b = 0; b < NBLK; ++b) {
        const int64_t s = s0 + b * 32 + (lane & 31);
        f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
        if (s < a.N) {
          v0 = *(const f32x4*)(a.x + s * 16 + 8 * (lane >> 5));
          v1 = *(const f32x4*)(a.x + s * 16 + 8 * (lane >> 5) + 4);
        }
        bf16x8 B;
#pragma unroll
        for (int e = 0; e < 4; ++e) { B[e] = (__bf16)v0[e]; B[4 + e] = (__bf16)v1[e]; }
#pragma unroll
        for (int t = 0; t < 2; ++t)
          acc[t][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ai[t], B, bias_acc(a.b_init, (2 * rg + t) * 32, lane), 0, 0, 0);
      }
    }
    if (lag) __syncthreads();  // phase offset of group 1
    // ---- hidden layers: E(l-1) | barrier | M(l) | barrier
    for (int l = 0; l <= a.L; ++l) {
      // epilogue of the previous layer (or of the init layer): activations -> LDS fragments (in place)
      stamp(0, l, pass);
#if BIAS_MODE == 1
      f32x16 bias_r[2];
      if (l < a.L) {
#pragma unroll
        for (int t = 0; t < 2; ++t) bias_r[t] = bias_acc(a.b_hid + l * H, (2 * rg + t) * 32, lane);
      }
      __builtin_amdgcn_sched_barrier(0);
#endif
      if (!(LS_ABLATE & 2)) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int b = 0; b < NBLK; ++b) store_tile(act_g, acc[t][b], b, 2 * rg + t, lane);
      }
      if (l == a.L) break;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#if BIAS_MODE == 1
        const f32x16 bias = bias_r[t];
#elif BIAS_MODE == 2
        f32x16 bias;
#pragma unroll
        for (int r = 0; r < 16; ++r) bias[r] = 0.01f * (float)r;
#else
        const f32x16 bias = bias_acc(a.b_hid + l * H, (2 * rg + t) * 32, lane);
#endif
#pragma unroll
        for (int b = 0; b < NBLK; ++b) acc[t][b] = bias;
      }
      stamp(1, l, pass);
      if (!(LS_ABLATE & 16)) __syncthreads();
      stamp(2, l, pass);
      // MFMA phase, chunk-major; scheduling fences keep the prefetch distances the source states
      bf16x8 Bq[2][NBLK];
#pragma unroll
      for (int b = 0; b < NBLK; ++b) Bq[0][b] = bload(b, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const bf16x8 A0 = Aq[c % PF][0], A1 = Aq[c % PF][1];
        if (c + 1 < 16) {
#pragma unroll
          for (int b = 0; b < NBLK; ++b) Bq[(c + 1) & 1][b] = bload(b, c + 1);
        }
#if BEARLY
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int b = 0; b < NBLK; ++b) {
          acc[0][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, Bq[c & 1][b], acc[0][b], 0, 0, 0);
          acc[1][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, Bq[c & 1][b], acc[1][b], 0, 0, 0);
          if (b == 0 && !(LS_ABLATE & 8)) {
            const int cn = c + PF;
            Aq[c % PF][0] = wload(cn < 16 ? l : l + 1, cn & 15, 0);
            Aq[c % PF][1] = wload(cn < 16 ? l : l + 1, cn & 15, 1);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      stamp(3, l, pass);
      if (!(LS_ABLATE & 16)) __syncthreads();
      stamp(4, l, pass);
    }
    __syncthreads();
    // ---- out layer (32 rows): row group 0 of each sample group (Aq holds the out fragments 0..PF-1 by now)
    if (rg == 0) {
      f32x16 o[NBLK];
      const f32x16 bias = bias_acc(a.b_out, 0, lane);
#pragma unroll
      for (int b = 0; b < NBLK; ++b) o[b] = bias;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const bf16x8 A0 = wload(a.L, c, 0);
#pragma unroll
        for (int b = 0; b < NBLK; ++b) {
          const bf16x8 B = bload(b, c);
          o[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B, o[b], 0, 0, 0);
        }
      }
#pragma unroll
      for (int b = 0; b < NBLK; ++b) {
        const int64_t s = s0 + b * 32 + (lane & 31);
        if (s < a.N) {
#pragma unroll
          for (int r = 0; r < 16; ++r) a.y[s * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)] = o[b][r];
        }
      }
    }
    if (!lag) __syncthreads();  // group 0 takes its extra barrier at the end
    __syncthreads();
  }
}

extern "C" int ls_mlp_forward(const void* w_init, const void* w_hid, const void* w_out, const float* b_init,
                              const float* b_hid, const float* b_out, const float* x, float* y, int64_t N, int L,
                              void* stream);
extern "C" int ls_mlp_forward_trace(const void* w_init, const void* w_hid, const void* w_out, const float* b_init,
                                    const float* b_hid, const float* b_out, const float* x, float* y, int64_t N, int L,
                                    void* stream, unsigned long long* trace);
extern "C" int ls_mlp_forward(const void* w_init, const void* w_hid, const void* w_out, const float* b_init,
                              const float* b_hid, const float* b_out, const float* x, float* y, int64_t N, int L,
                              void* stream) {
  return ls_mlp_forward_trace(w_init, w_hid, w_out, b_init, b_hid, b_out, x, y, N, L, stream, nullptr);
}
extern "C" int ls_mlp_forward_trace(const void* w_init, const void* w_hid, const void* w_out, const float* b_init,
                                    const float* b_hid, const float* b_out, const float* x, float* y, int64_t N, int L,
                                    void* stream, unsigned long long* trace) {
  Args a;
  a.trace = trace;
  a.w_init = (const char*)w_init; a.w_hid = (const char*)w_hid; a.w_out = (const char*)w_out;
  a.b_init = b_init; a.b_hid = b_hid; a.b_out = b_out; a.x = x; a.y = y; a.N = N; a.L = L;
  a.npass = (int)((N + S - 1) / S);
  const int lds = 2 * NBLK * 16 * 1024;
  static bool done = false;
  if (!done) {
    if (hipFuncSetAttribute((const void*)ls_mlp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -1;
    done = true;
  }
  int grid = a.npass < 256 ? a.npass : 256;
  hipLaunchKernelGGL(ls_mlp_kernel, dim3(grid), dim3(512), lds, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
