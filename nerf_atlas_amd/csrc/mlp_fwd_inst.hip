// Instantiations of the generic fused MLP forward kernel for ONE precision (NA_PREC_INST = 0 | 1); built as
// two translation units so the heavy kernels compile in parallel.
#include <atomic>
#include "mlp_forward_kernel.h"

#ifndef NA_PREC_INST
#error "compile with -DNA_PREC_INST=0 (bf16), 1 (bf16x3) or 2 (f16)"
#endif

namespace na {

template <int PREC, int ACT, int ENC, int NI, int NWAVES, int GEN = 0>
static int launch_forward(const MlpArgs& a, const TileTab& tab, hipStream_t stream) {
  auto kern = mlp_forward_kernel<PREC, ACT, ENC, NI, NWAVES, GEN>;
  // the attribute is per DEVICE (not per thread): one bit per device ordinal
  static std::atomic<uint64_t> attr_done{0};
  {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { set_error("hipGetDevice failed"); return NA_EHIP; }
    const uint64_t bit = 1ull << (dev & 63);
    if (!(attr_done.load(std::memory_order_acquire) & bit)) {
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return NA_EHIP; }
      attr_done.fetch_or(bit, std::memory_order_release);
    }
  }
  int grid = a.ngroups < 256 ? a.ngroups : 256;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NWAVES * 64), slots_for(PREC, NI) * a.buf_bytes + 8 * kMaxTiles, stream, a, tab);
  return check_launch("na_mlp_forward");
}

template <int PREC>
static int dispatch_forward(MlpArgs& a, const TileTab& tab, int NI, hipStream_t s) {
  const int act = a.d.activation;
  // 8 waves (2 per SIMD, <=256 VGPR) when the fragments fit, else 4 waves with the whole register file
  constexpr int NWS = PREC == NA_PREC_BF16X3 ? 4 : 8;
  // IPE latent generated in the prologue: the two MLPs of PlainNeRF(view) + mip (config 3)
#define NA_CASE_GEN(ACTV, ENCV, NIV, NW)                                                          \
  if (a.mip.rays != nullptr && act == ACTV && a.d.enc_kind == ENCV && NI == NIV &&                \
      (ENCV != NA_ENC_NONE || a.d.in_size == 5)) { /* (the encoder-less instantiation hard-wires the View head's 5 inputs) */ \
    a.ngroups = (int)((a.N + 32 * NW - 1) / (32 * NW));                                           \
    return launch_forward<PREC, ACTV, ENCV, NIV, NW, 1>(a, tab, s);                               \
  }
  NA_CASE_GEN(NA_ACT_LEAKY_RELU, NA_ENC_HASH, 9, NWS)
  NA_CASE_GEN(NA_ACT_SIN, NA_ENC_NONE, 11, NWS)
#undef NA_CASE_GEN
  if (a.mip.rays != nullptr) {
    set_error("na_mlp_forward_mip: no IPE-prologue kernel for activation %d, encoder %d, NI %d", act, a.d.enc_kind, NI);
    return NA_EUNSUPPORTED;
  }
#define NA_CASE(ACTV, ENCV, NIV, NW)                                                              \
  if (act == ACTV && a.d.enc_kind == ENCV && NI == NIV) {                                         \
    a.ngroups = (int)((a.N + 32 * NW - 1) / (32 * NW));                                           \
    return launch_forward<PREC, ACTV, ENCV, NIV, NW>(a, tab, s);                                  \
  }
  NA_CASE(NA_ACT_LEAKY_RELU, NA_ENC_NONE, 1, NWS)
  NA_CASE(NA_ACT_LEAKY_RELU, NA_ENC_NONE, 3, NWS)
  NA_CASE(NA_ACT_LEAKY_RELU, NA_ENC_HASH, 3, NWS)
  NA_CASE(NA_ACT_LEAKY_RELU, NA_ENC_HASH, 7, NWS)
  NA_CASE(NA_ACT_LEAKY_RELU, NA_ENC_HASH, 9, NWS)
  NA_CASE(NA_ACT_LEAKY_RELU, NA_ENC_FOURIER, 17, NWS)
  NA_CASE(NA_ACT_SIN, NA_ENC_NONE, 1, NWS)
  NA_CASE(NA_ACT_SIN, NA_ENC_NONE, 5, NWS)
  NA_CASE(NA_ACT_SIN, NA_ENC_NONE, 11, NWS)
#undef NA_CASE
  set_error("na_mlp_forward: no kernel for activation %d, encoder %d, NI %d", act, a.d.enc_kind, NI);
  return NA_EUNSUPPORTED;
}


#if NA_PREC_INST == 0
int dispatch_forward_bf16(MlpArgs& a, const TileTab& tab, int NI, hipStream_t s) {
  return dispatch_forward<NA_PREC_BF16>(a, tab, NI, s);
}
#elif NA_PREC_INST == 1
int dispatch_forward_bf16x3(MlpArgs& a, const TileTab& tab, int NI, hipStream_t s) {
  return dispatch_forward<NA_PREC_BF16X3>(a, tab, NI, s);
}
#else
int dispatch_forward_f16(MlpArgs& a, const TileTab& tab, int NI, hipStream_t s) {
  return dispatch_forward<NA_PREC_F16>(a, tab, NI, s);
}
#endif

}  // namespace na
