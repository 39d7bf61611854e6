// In-register encoders used by the fused kernels (same arithmetic as basic_ops.hip's standalone kernels).
#pragma once
#include "common.h"

namespace na {

// Four consecutive levels lvl0..lvl0+3 of the reference hash encoder (src/neural_blocks.py:143-190) for
// one point: f[4*k + comp].  lvl0 is 0 or 4 (the two lanes that share a sample split the 8 levels).
__device__ __forceinline__ void hash_levels4(float px, float py, float pz, const float4* __restrict__ tables,
                                             const HashRes& res, int lvl0, float (&f)[16]) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float Nl = lvl0 ? res.n[4 + k] : res.n[k];
    float vx = px * Nl, vy = py * Nl, vz = pz * Nl;
    float fx = floorf(vx), fy = floorf(vy), fz = floorf(vz);
    int lx = (int)fx, ly = (int)fy, lz = (int)fz;
    float wx = vx - fx, wy = vy - fy, wz = vz - fz;
    float iwx = 1.f - wx, iwy = 1.f - wy, iwz = 1.f - wz;
    const float4* tab = tables + (size_t)(lvl0 + k) * 65536;
    float4 e[8];
#pragma unroll
    for (int c = 0; c < 8; ++c)
      e[c] = tab[hash_index(lx + ((c >> 2) & 1), ly + ((c >> 1) & 1), lz + (c & 1))];
    float4 acc;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float w = (((c >> 2) & 1) ? wx : iwx) * (((c >> 1) & 1) ? wy : iwy) * ((c & 1) ? wz : iwz);
      if (c == 0) {
        acc.x = e[c].x * w; acc.y = e[c].y * w; acc.z = e[c].z * w; acc.w = e[c].w * w;
      } else {
        acc.x = acc.x + e[c].x * w; acc.y = acc.y + e[c].y * w; acc.z = acc.z + e[c].z * w; acc.w = acc.w + e[c].w * w;
      }
    }
    f[4 * k + 0] = acc.x; f[4 * k + 1] = acc.y; f[4 * k + 2] = acc.z; f[4 * k + 3] = acc.w;
  }
}

// One level of the same encoder (the arithmetic of one iteration of hash_levels4, bit for bit), split into the gather
// issue and the trilinear combine so that independent work can be scheduled between the two.
typedef __attribute__((ext_vector_type(4))) float hg_f32x4;  // (HIP's float4 as a struct member array lands in scratch)
struct HashGather {
  hg_f32x4 e[8];
  float wx, wy, wz;
};
__device__ __forceinline__ void hash_level_issue(float px, float py, float pz, const float4* __restrict__ tables,
                                                 float Nl, int level, HashGather& h) {
  float vx = px * Nl, vy = py * Nl, vz = pz * Nl;
  float fx = floorf(vx), fy = floorf(vy), fz = floorf(vz);
  int lx = (int)fx, ly = (int)fy, lz = (int)fz;
  h.wx = vx - fx; h.wy = vy - fy; h.wz = vz - fz;
  // one uniform base + a 32-bit element offset per gather (a per-lane 64-bit table pointer per level would be hoisted
  // out of the pass loop and spilled)
  const uint32_t lbase = (uint32_t)level << 16;
#pragma unroll
  for (int c = 0; c < 8; ++c)
    h.e[c] = *(const hg_f32x4*)(tables + (lbase | hash_index(lx + ((c >> 2) & 1), ly + ((c >> 1) & 1), lz + (c & 1))));
}
__device__ __forceinline__ void hash_level_finish(const HashGather& h, float (&f)[4]) {
  const float wx = h.wx, wy = h.wy, wz = h.wz;
  const float iwx = 1.f - wx, iwy = 1.f - wy, iwz = 1.f - wz;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    float w = (((c >> 2) & 1) ? wx : iwx) * (((c >> 1) & 1) ? wy : iwy) * ((c & 1) ? wz : iwz);
    if (c == 0) {
      a0 = h.e[c][0] * w; a1 = h.e[c][1] * w; a2 = h.e[c][2] * w; a3 = h.e[c][3] * w;
    } else {
      a0 = a0 + h.e[c][0] * w; a1 = a1 + h.e[c][1] * w; a2 = a2 + h.e[c][2] * w; a3 = a3 + h.e[c][3] * w;
    }
  }
  f[0] = a0; f[1] = a1; f[2] = a2; f[3] = a3;
}

}  // namespace na
