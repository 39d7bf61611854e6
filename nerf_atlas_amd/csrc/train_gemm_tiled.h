// The K-staged training GEMMs of round 1 (workgroup = WM x WN waves, each a 64x64 output tile; K staged 32 at a time through
// double-buffered LDS planes): small batches (N < 2048) and very wide layers; everything else runs the layer-synchronous kernels
// of train_gemm.hip (`lsnt`, `lstn`, `nrw`), train_bwd.hip and train_fwd.hip.  Moved out of train_gemm.hip in round 5 (text only).
#pragma once
#include "common.h"
#include "train_shared.h"

namespace na {

#ifndef TG_ABLATE
#define TG_ABLATE 0  // experiments only: 1 no MFMA, 2 no global loads, 4 no LDS stash, 8 no fragment reads
#endif
constexpr int TK = 32;   // K per stage
constexpr int TLD = 80;  // LDS row pitch, bytes

// One operand source: rows x K fp32, K contiguous, optionally the concatenation [p0 (k0 cols) | p1 (k1 cols)].
struct RowSrc {
  const float* p0;
  const float* p1;
  int k0, k1;
  int64_t rows;
};

// 4 consecutive k of one row (zero outside), vectorised when the 16-byte alignment is provable
__device__ __forceinline__ f32x4 load_k4(const RowSrc& s, int64_t row, int k) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (TG_ABLATE & 2) return v;
  if (row >= s.rows) return v;
  if (k + 4 <= s.k0 && (s.k0 & 3) == 0) return *(const f32x4*)(s.p0 + row * s.k0 + k);
  if (k >= s.k0 && k + 4 <= s.k0 + s.k1 && ((s.k0 | s.k1) & 3) == 0) return *(const f32x4*)(s.p1 + row * s.k1 + (k - s.k0));
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int kk = k + e;
    if (kk < s.k0) v[e] = s.p0[row * s.k0 + kk];
    else if (kk < s.k0 + s.k1) v[e] = s.p1[row * s.k1 + (kk - s.k0)];
  }
  return v;
}

// ---------------------------------------------------------------------------------------------- MFMA stage
// acc[mi][ni] += A(64 x 32) . B(64 x 32)^T from the LDS tiles of one stage (hi/lo planes `plane` bytes apart)
__device__ __forceinline__ void mma_stage(const char* At, const char* Bt, int a_plane, int b_plane, int lane,
                                          f32x16 (&acc)[2][2]) {
  const int off = (lane & 31) * TLD + (lane >> 5) * 16;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    bf16x8 ah[2], al[2], bh[2], bl[2];
    if (TG_ABLATE & 8) {
      for (int t = 0; t < 2; ++t) { ah[t] = al[t] = bh[t] = bl[t] = *(const bf16x8*)(At + off); }
    } else
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      ah[t] = *(const bf16x8*)(At + t * 32 * TLD + off + ks * 32);
      al[t] = *(const bf16x8*)(At + a_plane + t * 32 * TLD + off + ks * 32);
      bh[t] = *(const bf16x8*)(Bt + t * 32 * TLD + off + ks * 32);
      bl[t] = *(const bf16x8*)(Bt + b_plane + t * 32 * TLD + off + ks * 32);
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        if (TG_ABLATE & 1) { acc[mi][ni][0] += (float)al[mi][0] + (float)bh[ni][0] + (float)ah[mi][1] + (float)bl[ni][1]; continue; }
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mi], bh[ni], acc[mi][ni], 0, 0, 0);
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bl[ni], acc[mi][ni], 0, 0, 0);
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bh[ni], acc[mi][ni], 0, 0, 0);
      }
  }
}

// ---------------------------------------------------------------------------------------------- NT kernel
// C[M rows (samples), N cols] = A[M,K] . B[N,K]^T; both operands K-contiguous.  MODE 0: forward (A activated, +bias);
// MODE 1: input gradient (A = dY plain, epilogue multiplies by act'(x) and scatters into the two halves of the concat).
struct NtArgs {
  RowSrc a;        // [M, K]
  RowSrc b;        // [Ncols, K]
  int act;         // forward: activation on A; dgrad: activation whose derivative scales the result
  const float* bias;
  float* y0;       // forward: y [M, ncols];  dgrad: g_x0 [M, c0] (nullable)
  float* y1;       // dgrad: g_x1 [M, c1] (nullable)
  const float* x0; // dgrad: pre-activation inputs matching y0 / y1
  const float* x1;
  int c0, c1;      // column split (forward: c0 = ncols, c1 = 0)
};

template <int WM, int WN, int MODE>
__global__ __launch_bounds__(64 * WM * WN) void linear_nt_kernel(NtArgs g) {
  constexpr int BM = 64 * WM, BN = 64 * WN, NT = 64 * WM * WN;
  constexpr int A_PLANE = BM * TLD, B_PLANE = BN * TLD;
  constexpr int STAGE = 2 * A_PLANE + 2 * B_PLANE;
  constexpr int NA4 = BM * (TK / 4) / NT, NB4 = BN * (TK / 4) / NT;
  static_assert(NA4 >= 1 && NB4 >= 1, "tile too small for the workgroup");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int64_t m0 = (int64_t)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int K = g.a.k0 + g.a.k1;
  const int nk = (K + TK - 1) / TK;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x4 ra[1][NA4], rb[1][NB4];
  auto fetch = [&](const int slot, int kt) {
#pragma unroll
    for (int j = 0; j < NA4; ++j) {
      const int idx = tid + j * NT;
      ra[slot][j] = load_k4(g.a, m0 + (idx >> 3), kt * TK + (idx & 7) * 4);
    }
#pragma unroll
    for (int j = 0; j < NB4; ++j) {
      const int idx = tid + j * NT;
      rb[slot][j] = load_k4(g.b, n0 + (idx >> 3), kt * TK + (idx & 7) * 4);
    }
  };
  auto stash = [&](const int slot, int stage) {
    char* base = smem + stage * STAGE;
    if (TG_ABLATE & 4) { if (ra[slot][0][0] == 1.2345f && rb[slot][0][0] == 2.345f) base[tid] = 1; return; }
#pragma unroll
    for (int j = 0; j < NA4; ++j) {
      const int idx = tid + j * NT;
      f32x4 v = ra[slot][j];
      if (MODE == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = tact(v[e], g.act);
      }
      bf16x4 hi, lo;
      split4(v, hi, lo);
      char* d = base + (idx >> 3) * TLD + (idx & 7) * 8;
      *(bf16x4*)d = hi;
      *(bf16x4*)(d + A_PLANE) = lo;
    }
#pragma unroll
    for (int j = 0; j < NB4; ++j) {
      const int idx = tid + j * NT;
      bf16x4 hi, lo;
      split4(rb[slot][j], hi, lo);
      char* d = base + 2 * A_PLANE + (idx >> 3) * TLD + (idx & 7) * 8;
      *(bf16x4*)d = hi;
      *(bf16x4*)(d + B_PLANE) = lo;
    }
  };

  auto mma = [&](int stage) {
    const char* base = smem + stage * STAGE;
    mma_stage(base + wm * 64 * TLD, base + 2 * A_PLANE + wn * 64 * TLD, A_PLANE, B_PLANE, lane, acc);
  };
  fetch(0, 0);
  stash(0, 0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) fetch(0, kt + 1);  // in flight under the MFMAs of this stage
    mma(cur);
    if (kt + 1 < nk) stash(0, cur ^ 1);
    __syncthreads();
  }

  // Epilogue.  The MFMA layout is C[row = (r&3) + 8(r>>2) + 4(lane>>5)][col = lane&31]: memory is touched in 128-byte
  // pieces one row apart.  The input gradient also has to READ the forward input at those places (activation
  // derivative); there the tile goes through the (now free) LDS in two halves of BM/2 rows and moves as whole-row
  // float4 bursts (dgrad with sin: 461 -> 339 us).
  const int ncols = g.c0 + g.c1;
  if constexpr (MODE == 0) {
    // forward: plain stores straight from the MFMA layout measured faster (265 vs 289 us at 262144 x 256 x 256)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int col = n0 + wn * 64 + ni * 32 + (lane & 31);
      if (col >= ncols) continue;
      const float bj = g.bias != nullptr ? g.bias[col] : 0.f;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (row < g.a.rows) g.y0[row * g.c0 + col] = acc[mi][ni][r] + bj;
        }
    }
    return;
  }
  constexpr int CP = BN + 4;  // fp32 row pitch of the staging tile
  float* Cs = (float*)smem;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    if (mi) __syncthreads();
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        Cs[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * CP + wn * 64 + ni * 32 + (lane & 31)] = acc[mi][ni][r];
    __syncthreads();
    constexpr int SLOTS = (BM / 2) * (BN / 4);
#pragma unroll 4
    for (int idx = tid; idx < SLOTS; idx += NT) {
      const int lrow = idx / (BN / 4), c4 = idx % (BN / 4);
      const int64_t row = m0 + (lrow >> 5) * 64 + mi * 32 + (lrow & 31);
      const int col = n0 + c4 * 4;
      if (row >= g.a.rows || col >= ncols) continue;
      f32x4 v = *(const f32x4*)(Cs + lrow * CP + c4 * 4);
      const bool first = col < g.c0;
      float* dst = first ? g.y0 : g.y1;
      if (dst == nullptr) continue;
      const int ld = first ? g.c0 : g.c1;
      const int cc = first ? col : col - g.c0;
      const float* xin = MODE == 1 ? (first ? g.x0 : g.x1) : nullptr;
      const bool whole = (ld & 3) == 0 && (cc & 3) == 0 && cc + 4 <= ld && (first || (g.c0 & 3) == 0);
      if (whole) {
        if (MODE == 0 && g.bias != nullptr) {
          const f32x4 bj = *(const f32x4*)(g.bias + col);
          v += bj;
        }
        if (MODE == 1 && g.act != NA_ACT_NONE) {
          const f32x4 xv = *(const f32x4*)(xin + row * ld + cc);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= tact_grad(xv[e], g.act);
        }
        *(f32x4*)(dst + row * ld + cc) = v;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int ce = col + e;
          if (ce >= ncols) break;
          const bool f1 = ce < g.c0;
          float* d1 = f1 ? g.y0 : g.y1;
          if (d1 == nullptr) continue;
          const int l1 = f1 ? g.c0 : g.c1, c1 = f1 ? ce : ce - g.c0;
          float w = v[e];
          if (MODE == 0 && g.bias != nullptr) w += g.bias[ce];
          if (MODE == 1 && g.act != NA_ACT_NONE) w *= tact_grad((f1 ? g.x0 : g.x1)[row * l1 + c1], g.act);
          d1[row * l1 + c1] = w;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------- TN kernel (wgrad)
// dW[o, i] += sum_n dY[n, o] * act(X)[n, i] over one slice of samples.  Both operands are K(sample)-major in memory, so
// the loader transposes 4x4 micro-blocks in registers and writes [row][4 k] b64 words into the same LDS layout.
struct TnArgs {
  const float* dY;  // [N, out]
  int out;
  RowSrc x;         // [N, in] (concat) -- RowSrc.rows = N
  int act;
  int64_t slice;    // samples per workgroup (multiple of TK)
  float* dW;        // [out, in] (or a column block of a wider matrix: ldw)
  int ldw;          // leading dimension of dW and of the fixed-point accumulators
  float* db;        // [out] or null
  long long* fixW;  // deterministic mode: int64 fixed-point accumulators parallel to dW / db (else null)
  long long* fixb;
};

template <int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void linear_tn_kernel(TnArgs g) {
  constexpr int BM = 64 * WM, BN = 64 * WN, NT = 64 * WM * WN;
  constexpr int A_PLANE = BM * TLD, B_PLANE = BN * TLD;
  constexpr int STAGE = 2 * A_PLANE + 2 * B_PLANE;
  constexpr int A_BLOCKS = BM / 4 * (TK / 4), B_BLOCKS = BN / 4 * (TK / 4);  // 4x4 micro-blocks per tile
  constexpr int NA = (A_BLOCKS + NT - 1) / NT, NB = (B_BLOCKS + NT - 1) / NT;  // ... per thread
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int o0 = blockIdx.x * BM, i0 = blockIdx.y * BN;
  const int in = g.x.k0 + g.x.k1;
  const int64_t N = g.x.rows;
  const int64_t s0 = (int64_t)blockIdx.z * g.slice;
  const int64_t s1 = s0 + g.slice < N ? s0 + g.slice : N;
  const int nk = (int)((s1 - s0 + TK - 1) / TK);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // micro-block id -> (ng = id & 7: 4 samples, cg = id >> 3: 4 columns); 8 lanes cover one 128-byte row segment... x4 rows
  f32x4 ra[1][NA][4], rb[1][NB][4];
  float bsum[NA][4];
#pragma unroll
  for (int j = 0; j < NA; ++j)
#pragma unroll
    for (int c = 0; c < 4; ++c) bsum[j][c] = 0.f;
  const bool want_db = g.db != nullptr && blockIdx.y == 0;

  auto fetch = [&](const int slot, int kt) {
    const int64_t nb = s0 + (int64_t)kt * TK;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int id = tid + j * NT;
      const int col = o0 + (id >> 3) * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t n = nb + (id & 7) * 4 + r;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (n < s1 && id < A_BLOCKS) {
          if (col + 4 <= g.out && (g.out & 3) == 0) v = *(const f32x4*)(g.dY + n * g.out + col);
          else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (col + e < g.out) v[e] = g.dY[n * g.out + col + e];
          }
        }
        ra[slot][j][r] = v;
      }
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int id = tid + j * NT;
      const int col = i0 + (id >> 3) * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t n = nb + (id & 7) * 4 + r;
        rb[slot][j][r] = load_k4(g.x, (n < s1 && id < B_BLOCKS) ? n : N, col);  // row N -> zeros
      }
    }
  };
  auto stash = [&](const int slot, int stage) {
    char* base = smem + stage * STAGE;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int id = tid + j * NT;
      if (id >= A_BLOCKS) continue;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        f32x4 v = {ra[slot][j][0][c], ra[slot][j][1][c], ra[slot][j][2][c], ra[slot][j][3][c]};  // 4 consecutive samples of column c
        bsum[j][c] += (v[0] + v[1]) + (v[2] + v[3]);
        bf16x4 hi, lo;
        split4(v, hi, lo);
        char* d = base + ((id >> 3) * 4 + c) * TLD + (id & 7) * 8;
        *(bf16x4*)d = hi;
        *(bf16x4*)(d + A_PLANE) = lo;
      }
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int id = tid + j * NT;
      if (id >= B_BLOCKS) continue;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        f32x4 v = {tact(rb[slot][j][0][c], g.act), tact(rb[slot][j][1][c], g.act), tact(rb[slot][j][2][c], g.act),
                   tact(rb[slot][j][3][c], g.act)};
        bf16x4 hi, lo;
        split4(v, hi, lo);
        char* d = base + 2 * A_PLANE + ((id >> 3) * 4 + c) * TLD + (id & 7) * 8;
        *(bf16x4*)d = hi;
        *(bf16x4*)(d + B_PLANE) = lo;
      }
    }
  };

  auto mma = [&](int stage) {
    const char* base = smem + stage * STAGE;
    mma_stage(base + wm * 64 * TLD, base + 2 * A_PLANE + wn * 64 * TLD, A_PLANE, B_PLANE, lane, acc);
  };
  if (nk > 0) {
    fetch(0, 0);
    stash(0, 0);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) fetch(0, kt + 1);
    mma(cur);
    if (kt + 1 < nk) stash(0, cur ^ 1);
    __syncthreads();
  }

#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int col = i0 + wn * 64 + ni * 32 + (lane & 31);
    if (col >= in) continue;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = o0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < g.out) accumulate(g.dW, g.fixW, (int64_t)row * g.ldw + col, acc[mi][ni][r]);
      }
  }
  if (want_db) {
    // the 8 sample-groups of a column group sit in 8 adjacent lanes
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int id = tid + j * NT;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float v = bsum[j][c];
        v += __shfl_xor(v, 1);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 4);
        const int col = o0 + (id >> 3) * 4 + c;
        if ((id & 7) == 0 && id < A_BLOCKS && col < g.out) accumulate(g.db, g.fixb, col, v);
      }
    }
  }
}

template <int WM, int WN>
constexpr int smem_bytes() { return 2 * (2 * 64 * WM * TLD + 2 * 64 * WN * TLD); }

template <int WM, int WN, int MODE>
static int launch_nt(const NtArgs& a, int ncols, hipStream_t st, const char* what) {
  constexpr int BM = 64 * WM, BN = 64 * WN;
  auto k = linear_nt_kernel<WM, WN, MODE>;
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes<WM, WN>());
    if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return NA_EHIP; }
    configured = true;
  }
  dim3 grid((unsigned)((a.a.rows + BM - 1) / BM), (unsigned)((ncols + BN - 1) / BN));
  const int lds = smem_bytes<WM, WN>();
  hipLaunchKernelGGL(k, grid, dim3(64 * WM * WN), lds, st, a);
  return check_launch(what);
}

template <int MODE>
static int dispatch_nt(const NtArgs& a, int ncols, hipStream_t st, const char* what) {
  // the narrowest column block that covers ncols with the least padding; ties go to the wider block (one activation
  // pass per sample row)
  const int pad256 = (ncols + 255) / 256 * 256, pad128 = (ncols + 127) / 128 * 128, pad64 = (ncols + 63) / 64 * 64;
  if (pad256 <= pad128 && pad256 <= pad64) return launch_nt<2, 4, MODE>(a, ncols, st, what);
  if (pad128 <= pad64) return launch_nt<4, 2, MODE>(a, ncols, st, what);
  return launch_nt<4, 1, MODE>(a, ncols, st, what);
}


}  // namespace na
