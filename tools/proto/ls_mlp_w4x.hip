// PROTOTYPE (not part of libnerf_atlas_amd.so): f16x (f16 + 2 x MX-fp6, tools/proto/ls_mlp_f16x.hip) on a data flow that
// fetches a layer's weights ONCE per 128 samples instead of once per 64.
//
// What bounds render_ls_kernel<NA_PREC_F16X>: every sample group of 64 samples streams the layer's 57 KiB per
// row group through the vector memory path (64 B/clk per CU); the two groups of a workgroup are one phase apart, on different
// waves, and cannot share a fetch.  Here ONE wave per SIMD (4 waves x 64 rows, 512 registers) works for BOTH groups:
//
//   phase 2l+1:  MFMAs of group A, layer l   ||  epilogue of group B, layer l-1   (same instruction stream)
//   phase 2l+2:  MFMAs of group B, layer l   ||  epilogue of group A, layer l
//
// The layer's f16 fragments (128 registers) stay resident for both phases and are refilled in place with the next layer's
// behind their last use; the fp6 operands (6.25 KiB per record) are streamed in both phases: 20.5 KiB per record pair instead
// of 28.5.  The epilogue of the other group is cut into 96 slots, one behind every MFMA (3 instructions on average), and
// pinned there with scheduling fences.
//
// LDS per (group, block, K64 group Q): 4 f16 fragments | R | T (fp6 operands of 32 B per lane: dwords 0..5 values, 6 scale) =
// 8 KiB; 2 groups x 2 blocks x 4 Q = 128 KiB.  Weight stream: ls_f16x.py pack_weights(layout "v2": per tile {WL6 | WT6} as
// three 16-byte parts).
// W4X_ABLATE bits: 1 no epilogue work (slots empty; wrong), 2 no weight loads after the first layer (wrong), 4 no barriers (wrong)
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(12))) uint32_t u32x12;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(6))) int i32x6;

#ifndef W4X_ABLATE
#define W4X_ABLATE 0
#endif
#ifndef W4X_PIPE
#define W4X_PIPE 0
#endif
constexpr int NB = 2;
constexpr int S = 2 * NB * 32;                 // samples per pass
constexpr int H = 256;
constexpr int KQ = 8192;                       // LDS bytes per (block, K64 group)
constexpr int BLK = 4 * KQ;
constexpr int GRP = NB * BLK;
constexpr int REC = 8192 + 2 * 3072 + 256;     // weight stream bytes per (layer, row group, K64 group)

__device__ __forceinline__ float leaky(float v) { return __builtin_amdgcn_fmed3f(v, v * 0.01f, 3.0e38f); }
__device__ __forceinline__ i32x8 lo6(const u32x12& v) { return i32x8{(int)v[0], (int)v[1], (int)v[2], (int)v[3], (int)v[4], (int)v[5], 0, 0}; }
__device__ __forceinline__ i32x8 hi6(const u32x12& v) { return i32x8{(int)v[6], (int)v[7], (int)v[8], (int)v[9], (int)v[10], (int)v[11], 0, 0}; }
template <int SA>
__device__ __forceinline__ void mma6(f32x16& acc, const i32x8& A, int sa, const i32x8& B) {
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc, 2, 2, SA, sa, 0, B[6]);
}

struct Args {
  const char* w_init;   // [8 tiles][1 KiB] f16 fragments, standard k order
  const char* w_hid;    // [L][4 rg][4 Q][REC]
  const float* b_init;  // [256]
  const float* b_pack;  // [L][4 rg][2 t][2 h][16]
  const float* x;       // [N,16]
  float* y;             // [N,32]: rows 0..31 of the LAST hidden Linear, before the activation
  int64_t N;
  int L;
  int npass;
  unsigned long long* trace;
};

// ---- the epilogue of one group (2 blocks x 2 tiles of finished accumulators -> LDS operands), cut into 96 slots
typedef __attribute__((ext_vector_type(16))) uint32_t u32x16;
struct Epi {
  u32x16 pk;   // (clang vectors, constant element indices: plain arrays of this struct stayed in scratch memory)
  f32x16 r0, r1;
  float m;
  int eT, eR;
  f32x4 t0, t1;  // (W4X_PIPE) products in flight (slots s % 3)
};
// slot s of block bb (0..47): acc0 / acc1 = the block's two accumulator tiles (activated in place), kq = its K64 group in LDS
__device__ __forceinline__ void epi_slot(int s, Epi& e, f32x16& a0, f32x16& a1, char* kq, int lane) {
  if (W4X_ABLATE & 1) return;
#if W4X_PIPE
  // software-pipelined over three slots, so that the instructions of one slot are independent of each other (a single wave has
  // nobody to fill the latency of a dependent VALU chain): slot s multiplies pair s, clamps pair s - 1, packs pair s - 2
  if (s < 18) {
    if (s < 16) {
      const int u = s & 7;
      const float x = s < 8 ? a0[2 * u] : a1[2 * u], y = s < 8 ? a0[2 * u + 1] : a1[2 * u + 1];
      float p = x * 0.01f, q = y * 0.01f;
      asm volatile("" : "+v"(p), "+v"(q));
      e.t0[s % 3] = p; e.t1[s % 3] = q;
    }
    if (s >= 1 && s < 17) {
      const int z = s - 1, u = z & 7;
      const float x = z < 8 ? a0[2 * u] : a1[2 * u], y = z < 8 ? a0[2 * u + 1] : a1[2 * u + 1];
      float p = __builtin_amdgcn_fmed3f(x, e.t0[z % 3], 3.0e38f), q = __builtin_amdgcn_fmed3f(y, e.t1[z % 3], 3.0e38f);
      asm volatile("" : "+v"(p), "+v"(q));
      if (z < 8) { a0[2 * u] = p; a0[2 * u + 1] = q; } else { a1[2 * u] = p; a1[2 * u + 1] = q; }
    }
    if (s >= 2) {
      const int z = s - 2, u = z & 7;
      const float x = z < 8 ? a0[2 * u] : a1[2 * u], y = z < 8 ? a0[2 * u + 1] : a1[2 * u + 1];
      uint32_t d = __builtin_bit_cast(uint32_t, f16x2{(_Float16)x, (_Float16)y});
      if (z == 0) e.m = 0.f;
      asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(e.m) : "v"(x), "v"(y));
      asm volatile("" : "+v"(d));
      e.pk[z] = d;
    }
  } else if (s == 18 || s == 19) {
    const int c0 = 2 * (s - 18);
#pragma unroll
    for (int c = c0; c < c0 + 2; ++c) *(u32x4*)(kq + c * 1024 + lane * 16) = u32x4{e.pk[4 * c], e.pk[4 * c + 1], e.pk[4 * c + 2], e.pk[4 * c + 3]};
    if (s == 18) {
      const int ev = (int)(__builtin_bit_cast(uint32_t, e.m) >> 23);
      e.eT = ev > 3 ? ev - 2 : 1;
      e.eR = ev > 14 ? ev - 13 : 1;
      asm volatile("" : "+v"(e.eT), "+v"(e.eR));
    }
  } else if (s < 36) {
    const int u = s - 20;
    const int k = u & 7;
    const float va = u < 8 ? a0[2 * k] : a1[2 * k], vb = u < 8 ? a0[2 * k + 1] : a1[2 * k + 1];
    const uint32_t d = e.pk[u];
    float p, q;
    asm volatile("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(p) : "v"(va), "v"(d));
    asm volatile("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(q) : "v"(vb), "v"(d));
    if (u < 8) { e.r0[2 * k] = p; e.r0[2 * k + 1] = q; } else { e.r1[2 * k] = p; e.r1[2 * k + 1] = q; }
  } else if (s == 36) {
    const i32x6 R = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(e.r0, e.r1, __builtin_bit_cast(float, (uint32_t)e.eR << 23));
    char* p = kq + 4096 + lane * 16;
    *(u32x4*)p = u32x4{(uint32_t)R[0], (uint32_t)R[1], (uint32_t)R[2], (uint32_t)R[3]};
    *(u32x4*)(p + 1024) = u32x4{(uint32_t)R[4], (uint32_t)R[5], (uint32_t)e.eR, 0u};
  } else if (s == 37) {
    const i32x6 T = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a0, a1, __builtin_bit_cast(float, (uint32_t)e.eT << 23));
    char* p = kq + 4096 + 2048 + lane * 16;
    *(u32x4*)p = u32x4{(uint32_t)T[0], (uint32_t)T[1], (uint32_t)T[2], (uint32_t)T[3]};
    *(u32x4*)(p + 1024) = u32x4{(uint32_t)T[4], (uint32_t)T[5], (uint32_t)e.eT, 0u};
  }
  return;
#endif
  if (s < 16) {
    const int u = s & 7;
    const float x = leaky(s < 8 ? a0[2 * u] : a1[2 * u]), y = leaky(s < 8 ? a0[2 * u + 1] : a1[2 * u + 1]);
    if (s < 8) { a0[2 * u] = x; a0[2 * u + 1] = y; } else { a1[2 * u] = x; a1[2 * u + 1] = y; }
    uint32_t d = __builtin_bit_cast(uint32_t, f16x2{(_Float16)x, (_Float16)y});
    if (s == 0) e.m = 0.f;
    asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(e.m) : "v"(x), "v"(y));
    asm volatile("" : "+v"(d));
    e.pk[s] = d;
  } else if (s == 16 || s == 17) {
    const int c0 = 2 * (s - 16);
#pragma unroll
    for (int c = c0; c < c0 + 2; ++c) *(u32x4*)(kq + c * 1024 + lane * 16) = u32x4{e.pk[4 * c], e.pk[4 * c + 1], e.pk[4 * c + 2], e.pk[4 * c + 3]};
  } else if (s == 18) {
    const int ev = (int)(__builtin_bit_cast(uint32_t, e.m) >> 23);
    e.eT = ev > 3 ? ev - 2 : 1;
    e.eR = ev > 14 ? ev - 13 : 1;
    asm volatile("" : "+v"(e.eT), "+v"(e.eR));
  } else if (s < 35) {
    const int u = s - 19;  // 0..15: packed dword u <-> tile u >> 3, registers 2 (u & 7), +1
    const int k = u & 7;
    const float va = u < 8 ? a0[2 * k] : a1[2 * k], vb = u < 8 ? a0[2 * k + 1] : a1[2 * k + 1];
    const uint32_t d = e.pk[u];
    float p, q;
    asm volatile("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(p) : "v"(va), "v"(d));
    asm volatile("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(q) : "v"(vb), "v"(d));
    if (u < 8) { e.r0[2 * k] = p; e.r0[2 * k + 1] = q; } else { e.r1[2 * k] = p; e.r1[2 * k + 1] = q; }
  } else if (s == 35) {
    const i32x6 R = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(e.r0, e.r1, __builtin_bit_cast(float, (uint32_t)e.eR << 23));
    char* p = kq + 4096 + lane * 16;
    *(u32x4*)p = u32x4{(uint32_t)R[0], (uint32_t)R[1], (uint32_t)R[2], (uint32_t)R[3]};
    *(u32x4*)(p + 1024) = u32x4{(uint32_t)R[4], (uint32_t)R[5], (uint32_t)e.eR, 0u};
  } else if (s == 36) {
    const i32x6 T = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a0, a1, __builtin_bit_cast(float, (uint32_t)e.eT << 23));
    char* p = kq + 4096 + 2048 + lane * 16;
    *(u32x4*)p = u32x4{(uint32_t)T[0], (uint32_t)T[1], (uint32_t)T[2], (uint32_t)T[3]};
    *(u32x4*)(p + 1024) = u32x4{(uint32_t)T[4], (uint32_t)T[5], (uint32_t)e.eT, 0u};
  }
}

__global__ __launch_bounds__(256) void ls_w4x_kernel(Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int rg = __builtin_amdgcn_readfirstlane(tid >> 6);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.w_hid, 0, a.L * 16 * REC, 0x00020000);
  const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc((void*)a.b_pack, 0, a.L * 4 * 64 * 4, 0x00020000);
  auto rec_off = [&](int l, int Q) { return __builtin_amdgcn_readfirstlane(((l * 4 + rg) * 4 + Q) * REC); };
  auto wload16 = [&](int off, int t, int c) -> f16x8 {
    return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane * 16 + c * 1024, off + t * 4096, 0));
  };
  auto wload6 = [&](int off, int t) -> u32x12 {
    const u32x4 p = __builtin_amdgcn_raw_buffer_load_b128(wrs, lane * 16, off + 8192 + t * 3072, 0);
    const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(wrs, lane * 16 + 1024, off + 8192 + t * 3072, 0);
    const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(wrs, lane * 16 + 2048, off + 8192 + t * 3072, 0);
    return u32x12{p[0], p[1], p[2], p[3], q[0], q[1], q[2], q[3], r[0], r[1], r[2], r[3]};
  };
  auto wloadsc = [&](int off) -> int { return (int)__builtin_amdgcn_raw_buffer_load_b32(wrs, lane * 4, off + 8192 + 6144, 0); };
  auto bias_tile = [&](int l, int t) -> f32x16 {
    f32x16 b;
    const int so = __builtin_amdgcn_readfirstlane(((l * 4 + rg) * 2 + t) * 128);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brs, (lane >> 5) * 64 + q * 16, so, 0));
      b[4 * q] = v[0]; b[4 * q + 1] = v[1]; b[4 * q + 2] = v[2]; b[4 * q + 3] = v[3];
    }
    return b;
  };
  const bool tr = a.trace != nullptr && blockIdx.x == 0 && lane == 0 && rg == 0;

  // ---- weights of layer 0: f16 fragments resident, fp6 operands of record 0 (every pass ends with them in place again)
  f16x8 A16[4][4][2];
  u32x12 A6[2][2];
  int Asc[2];
#pragma unroll
  for (int Q = 0; Q < 4; ++Q)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int t = 0; t < 2; ++t) A16[Q][c][t] = wload16(rec_off(0, Q), t, c);
  A6[0][0] = wload6(rec_off(0, 0), 0); A6[0][1] = wload6(rec_off(0, 0), 1); Asc[0] = wloadsc(rec_off(0, 0));
  for (int pass = blockIdx.x; pass < a.npass; pass += gridDim.x) {

    f32x16 acc[2][2][NB];  // [group][tile][block]
    {  // ---- init layer (K = 16, f16): both groups
      f16x8 Ai[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) Ai[t] = *(const f16x8*)(a.w_init + (size_t)(2 * rg + t) * 1024 + lane * 16);
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const int64_t s = (int64_t)pass * S + (g * NB + b) * 32 + (lane & 31);
          f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
          if (s < a.N) {
            v0 = *(const f32x4*)(a.x + s * 16 + 8 * (lane >> 5));
            v1 = *(const f32x4*)(a.x + s * 16 + 8 * (lane >> 5) + 4);
          }
          f16x8 B;
#pragma unroll
          for (int e = 0; e < 4; ++e) { B[e] = (_Float16)v0[e]; B[4 + e] = (_Float16)v1[e]; }
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            f32x16 bi;
#pragma unroll
            for (int r = 0; r < 16; ++r) bi[r] = a.b_init[(2 * rg + t) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)];
            acc[g][t][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ai[t], B, bi, 0, 0, 0);
          }
        }
    }
    Epi ep;
    // ---- phase 0: the epilogue of group 0's init layer, alone
#pragma unroll
    for (int bb = 0; bb < NB; ++bb)
#pragma unroll
      for (int s = 0; s < 38; ++s) epi_slot(s, ep, acc[0][0][bb], acc[0][1][bb], smem + bb * BLK + rg * KQ, lane);
    if (!(W4X_ABLATE & 4)) __syncthreads();

    // ---- phases 1 .. 2L: MFMAs of group g at layer l  ||  epilogue of group g ^ 1 (its previous Linear)
    for (int l = 0; l < a.L; ++l) {
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        if (tr && pass == 0 && l < 32) a.trace[l * 2 + g] = __builtin_amdgcn_s_memtime();
        const int eg = g ^ 1;
        char* mbase = smem + g * GRP;    // activations this phase's MFMAs read
        char* ebase = smem + eg * GRP;   // activations the epilogue slots overwrite
        f32x16 bv[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) bv[t] = bias_tile(l, t);
        auto b16 = [&](int b, int Q, int c) -> f16x8 { return *(const f16x8*)(mbase + b * BLK + Q * KQ + c * 1024 + lane * 16); };
        auto b6 = [&](int b, int Q, int k) -> i32x8 {
          const char* p = mbase + b * BLK + Q * KQ + 4096 + k * 2048 + lane * 16;
          const u32x4 x = *(const u32x4*)p, y = *(const u32x4*)(p + 1024);
          return i32x8{(int)x[0], (int)x[1], (int)x[2], (int)x[3], (int)y[0], (int)y[1], (int)y[2], (int)y[3]};
        };
        f16x8 Bq[2][NB];
        i32x8 B6[NB][2];
#pragma unroll
        for (int b = 0; b < NB; ++b) Bq[0][b] = b16(b, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        int slot = 0;
#pragma unroll
        for (int Q = 0; Q < 4; ++Q) {
          // next record's fp6 operands: same layer next Q, or (Q == 3) record 0 of the layer the NEXT phase works on
          const int nl = Q < 3 ? l : (g == 0 ? l : (l + 1 < a.L ? l + 1 : 0));
          const int noff = rec_off(nl, (Q + 1) & 3);
          // refill target for the resident f16 fragments (second phase of the layer only): layer l + 1, same Q
          const int roff = rec_off(l + 1 < a.L ? l + 1 : 0, Q);
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int ci = Q * 4 + c;
            if (ci + 1 < 16) {
#pragma unroll
              for (int b = 0; b < NB; ++b) Bq[(ci + 1) & 1][b] = b16(b, (ci + 1) >> 2, (ci + 1) & 3);
            }
            if (c == 1) {
#pragma unroll
              for (int b = 0; b < NB; ++b) { B6[b][0] = b6(b, Q, 0); B6[b][1] = b6(b, Q, 1); }
            }
            if (c == 0) {
              A6[(Q + 1) & 1][0] = wload6(noff, 0); A6[(Q + 1) & 1][1] = wload6(noff, 1); Asc[(Q + 1) & 1] = wloadsc(noff);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
              for (int t = 0; t < 2; ++t) {
                acc[g][t][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A16[Q][c][t], Bq[ci & 1][b], ci == 0 ? bv[t] : acc[g][t][b], 0, 0, 0);
                epi_slot(slot % 48, ep, acc[eg][0][slot / 48], acc[eg][1][slot / 48], ebase + (slot / 48) * BLK + rg * KQ, lane);
                ++slot;
                __builtin_amdgcn_sched_barrier(0);
              }
            if (g == 1 && !(W4X_ABLATE & 2)) {  // last use of this chunk's fragments: refill with the next layer's
              A16[Q][c][0] = wload16(roff, 0, c);
              A16[Q][c][1] = wload16(roff, 1, c);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
#pragma unroll
          for (int b = 0; b < NB; ++b) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              if (k == 0) mma6<0>(acc[g][0][b], lo6(A6[Q & 1][0]), Asc[Q & 1], B6[b][1]);
              if (k == 1) mma6<1>(acc[g][0][b], hi6(A6[Q & 1][0]), Asc[Q & 1], B6[b][0]);
              if (k == 2) mma6<2>(acc[g][1][b], lo6(A6[Q & 1][1]), Asc[Q & 1], B6[b][1]);
              if (k == 3) mma6<3>(acc[g][1][b], hi6(A6[Q & 1][1]), Asc[Q & 1], B6[b][0]);
              epi_slot(slot % 48, ep, acc[eg][0][slot / 48], acc[eg][1][slot / 48], ebase + (slot / 48) * BLK + rg * KQ, lane);
              ++slot;
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
        if (!(W4X_ABLATE & 4)) __syncthreads();
      }
    }
    // ---- rows 0..31 of the last Linear, before the activation: group 1's come straight from its accumulators, group 0's
    // were taken through one more epilogue (in place: activated) -- the prototype reports group 1 only and group 0 via a copy
    // made before that epilogue ran; simpler: the host compares group 1's samples only
    if (rg == 0) {
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int64_t s = (int64_t)pass * S + (NB + b) * 32 + (lane & 31);
        if (s < a.N) {
#pragma unroll
          for (int r = 0; r < 16; ++r) a.y[s * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)] = acc[1][0][b][r];
        }
      }
    }
    if (!(W4X_ABLATE & 4)) __syncthreads();
  }
}

extern "C" int ls_w4x_samples_per_pass() { return S; }

extern "C" int ls_w4x_forward(const void* w_init, const void* w_hid, const float* b_init, const float* b_pack, const float* x, float* y,
                              int64_t N, int L, void* stream, unsigned long long* trace) {
  Args a;
  a.trace = trace;
  a.w_init = (const char*)w_init; a.w_hid = (const char*)w_hid;
  a.b_init = b_init; a.b_pack = b_pack; a.x = x; a.y = y; a.N = N; a.L = L;
  a.npass = (int)((N + S - 1) / S);
  const int lds = 2 * GRP;
  static bool done = false;
  if (!done) {
    if (hipFuncSetAttribute((const void*)ls_w4x_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -1;
    done = true;
  }
  int grid = a.npass < 256 ? a.npass : 256;
  hipLaunchKernelGGL(ls_w4x_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
