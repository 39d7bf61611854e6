// na_render_plain_view_ls: PlainNeRF.forward with the View head (src/nerf.py:326-361, src/refl.py:190-207) as ONE
// kernel on the LAYER-SYNCHRONOUS engine.  Same arithmetic and work items as render_fused.hip
// (a block = 32 consecutive steps of one ray, lane&31 = step, compositing = in-wave scan), different data flow:
//
//  * 8 waves = 4 row groups (64 output rows = two 32-row MFMA tiles) x 2 sample groups (NBLK blocks each).
//  * Activations live in LDS as ready-made MFMA B fragments, [group][block][chunk][lane] x 16 B: the lane that
//    produces 8 values of a chunk is the lane that consumes them, so ds_write_b128 / ds_read_b128 are lane-linear and
//    conflict-free; the implied k permutation is the engine's pi_perm, folded into the weight packing.
//  * A layer = an MFMA phase that is CHUNK-major: per 16-wide k chunk the wave streams its two A fragments straight
//    from global memory into registers (a 4-deep software-prefetched ring that runs across layer, MLP and pass
//    boundaries: no LDS-DMA, no weight ring in LDS) and feeds them to 2 x NBLK MFMAs whose B fragments come from LDS:
//    every LDS fragment read feeds TWO MFMAs (the one-read-per-MFMA structure of mlp_engine.h caps the matrix pipe
//    near 57 %).  All 2 x NBLK accumulator tiles stay in registers until the layer is complete; the activation
//    epilogue then overwrites the LDS fragments in place.
//  * The two sample groups run in ANTIPHASE, one workgroup barrier per phase: while one group streams MFMAs the
//    other runs its epilogue / encoder / compositing (VALU + LDS writes) on the same SIMDs, so VALU work never sits
//    between a wave's own MFMAs and each SIMD always has one MFMA-only wave.
//  * LDS: per group NBLK x (16 hidden + 4 init) chunks = 80 KiB, 160 KiB per workgroup (bf16: NBLK = 4; bf16x3 keeps
//    hi and lo planes: NBLK = 2).  The fifth init chunk of the View MLP (x, y, z, elev, azim) does not fit: every
//    consuming wave rebuilds its fragments in registers (scalar loads of the ray and of its per-ray elev/azim, which a
//    small pre-kernel writes once per ray; ~15 VALU per block) right before the two MFMAs that use them.
//  * `first.out` / `view.out` (65 and 3 rows) run block-per-wave (wave rg = block rg, all out tiles), which is also the
//    assignment of the hash-encoder prologue and of compositing, so density and colour never leave their wave.
//  * Sample group G = 2 * workgroup + g renders the rays G, G + nG, ... one after the other, block by block in step
//    order: the transmittance is carried across blocks (and passes) inside the group -- block partials meet in the
//    group's idle hidden region of LDS -- so there are no per-block partials in HBM and no finalize launch.
//
// Compiled once per precision (-DNA_PREC_INST=0|1|2).
#include <atomic>
#include <cstring>
#include <type_traits>
#include "mlp_layout.h"
#include "encoders.h"

#ifndef NA_PREC_INST
#error "compile with -DNA_PREC_INST=0 (bf16), 1 (bf16x3), 2 (f16) or 3 (f16x)"
#endif
// NA_LS_TRACE: waves 0 and 4 of workgroup 0 stamp s_memtime before and after every barrier of their second pass
// (tools/ls_trace.py).  Timing experiments only.
#ifndef NA_LS_MIP_ABLATE
#define NA_LS_MIP_ABLATE 0  // experiments (tools/ls_variant.py): 1 = MODEL 6 without its IPE generation (wrong output, timing only)
#endif
#ifndef NA_LS_TRACE
#define NA_LS_TRACE 0
#endif
#ifndef NA_LS_LAG_OVERRIDE
#define NA_LS_LAG_OVERRIDE 0  // experiments: -DNA_LS_LAG_OVERRIDE=n (odd)
#endif
namespace na {

namespace ls {

constexpr int kPF = 4;       // weight prefetch depth, in fragment pairs
constexpr int kNPhase = 16;  // bias blocks per row group (MFMA phases per pass: 12 PlainNeRF, 8 TinyNeRF, 6 View, 13 SIREN-VolSDF)
// fragment pairs a wave consumes per phase: first.init, first.L0 (3 skip + 16), L1..L3, first.out (ONE 32-row tile per
// row group: 16 fragments), view.init (4 latent + geometry), view.L0 (5 skip + 16), L1..L3, view.out (16 / 2)
__host__ __device__ constexpr int phase_pairs(int p) {
  return p == 0 ? 3 : p == 1 ? 19 : p == 5 ? 8 : p == 6 ? 5 : p == 7 ? 21 : p == 11 ? 8 : 16;
}
constexpr int kPairsPerPass = 160;
// TinyNeRF (src/nerf.py:278-305: one SkipConnMLP 3 -> 256 x 6 -> 4, skip 3, no encoder) on the same engine, MODEL = 1: phases
// init (x,y,z chunk + one zero chunk: 108 pairs per pass keep the 4-deep ring phase static), L0 (1 skip + 16), L1, L2,
// L3 (1 skip + 16), L4, L5, out (16 / 2: one 32-row tile, block per wave)
constexpr int kTinyPhases = 8;
__host__ __device__ constexpr int tiny_phase_pairs(int p) { return p == 0 ? 2 : (p == 1 || p == 4) ? 17 : p == 7 ? 8 : 16; }
constexpr int kTinyPairs = 108;
// The View head + compositing alone (MODEL 2; VolSDF's second half, src/nerf.py:981-1013): density and the 64-wide latent
// of every sample come from HBM (the SDF network's output rows), phases view.init (4 latent + geometry), L0 (5 skip + 16),
// L1..L3, out (16 / 2) + 2 zero pairs that keep a pass a multiple of the ring depth
constexpr int kViewPhases = 6;
__host__ __device__ constexpr int view_phase_pairs(int p) { return p == 0 ? 5 : p == 1 ? 21 : p == 5 ? 10 : 16; }
constexpr int kViewPairs = 84;
// VolSDF with the SIREN SDF network (src/sdf.py:278-287: 3 -> 5 x 256 sin, skip 3 -> 1 + 64) as ONE kernel (MODEL 3): the
// PlainNeRF schedule with `first` replaced by the SIREN -- sdf.init (x,y,z chunk + one zero chunk), L0 (1 skip + 16), L1, L2,
// L3 (1 + 16), L4, sdf.out (65 rows row-major: 16 / 2), then the View half of MODEL 2 (5, 21, 16, 16, 16, 8 + 2 zero pairs)
constexpr int kSirenPhases = 13;
__host__ __device__ constexpr int siren_phase_pairs(int p) {
  return p == 0 ? 2 : (p == 1 || p == 4) ? 17 : p == 6 ? 8 : p == 7 ? 5 : p == 8 ? 21 : p == 12 ? 10 : 16;
}
constexpr int kSirenPairs = 176;
constexpr int kHeaderBytes = 1024;
constexpr int kBiasBytes = 4 * kNPhase * 1024;  // [row group][phase] 1-KiB blocks: floats [slot][hi(2)][16]
constexpr uint32_t kMagic = 0x4C533032u;        // "LS02"
constexpr int kPartialFloats = 8;

// ---- NA_PREC_F16X (PlainNeRF schedule only): hidden activations and hidden-layer weights in the f16 + 2 x MX-fp6 format
// (tools/proto/ls_mlp_f16x.hip is the measured prototype of this data flow).
//   An fp6 OPERAND is 32 bytes per lane, two lane-linear 16-byte parts (1 KiB each): dwords 0..5 = the 32 fp6 values, dword 6 =
//   its E8M0 scale (byte 0), dword 7 unused -- two 16-byte loads give the scaled MFMA's 8-dword operand AND its scale register.
//   LDS, per (block, K64 group Q = the row group that produced those 64 features): 4 f16 fragments (4 KiB) | R = fp6 of the f16
//   rounding residual (2 KiB) | T = fp6 of the value (2 KiB).  A lane's 32 values of a group = its accumulator registers of
//   the producer's two tiles: the producing lane is the consuming lane (lane = (sample, k half)), as for the f16 fragments.
//   Weight stream per row group: 16 init / geometry chunk PAIRS in the bf16x3 layout with f16 elements (f16 hi + f16 lo planes,
//   three f16 products), then 40 uniform hidden RECORDS (one per (Linear, Q); the out Linears use tile 0 only):
//   2 tiles x 4 f16 fragments (8 KiB) | 48 bytes per lane {WL6 of tile 0 | WL6 of tile 1}, WL6 = fp6(W - f16 W), as three
//   lane-linear 16-byte parts (3 KiB: twelve consecutive registers hold both operands) | one dword per lane with the four E8M0
//   scale bytes (WL6 t0, WT6 t0, WL6 t1, WT6 t1).  The second correction operand WT6 = fp6(W) is NOT in the stream (round 4):
//   the consuming wave derives it from the tile's four f16 fragments with ONE v_cvt_scalef32_pk32_fp6_f16 (the lane's 32
//   halves of the K64 group sit in sixteen consecutive registers), so its slot order is the fragments' element order
//   (slot 8 c + e <-> chunk c, element e) and the activation side packs the residual plane R in that order.  The MFMA phase
//   is bound by the 64 B/clk the vector memory path delivers per CU (a record feeds 24 MFMAs = 768 cycles; four waves x
//   14.25 KiB were 912 cycles of that path, 11.25 KiB are 720), so every byte counts.
#ifndef NA_LSX_PRIO
#define NA_LSX_PRIO 0  // experiments: 0 the MFMA phases run at s_setprio 1 (like the other precisions), 1 no priorities, 2 the epilogues
#endif
#ifndef NA_LSX_EXP
#define NA_LSX_EXP 0  // timing experiments (tools/ls_variant.py): 2 no fp6 loads, 4 no f16 refills,
                     // 8 no LDS reads of the T plane, 16 no LDS writes of the T plane, 32 no WT6 derivation
#endif
namespace x {
constexpr int KQ = 4096 + 2 * 2048;          // LDS bytes per (block, K64 group)
constexpr int BLKH = 4 * KQ;                 // hidden activations of one block (32 KiB)
constexpr int REC = 8192 + 3072 + 256;       // stream bytes per record (11.25 KiB)
constexpr int PAIRB = 4096;                  // stream bytes per init / geometry chunk pair
// pairs / records per pass and row group of the four schedules (MODEL 0 PlainNeRF: first.init 3, first.L0 3, view.init 4 +
// geometry, view.L0 4 + geometry | first.L0..L3 16, first.out 4, view.L0..L3 16, view.out 4;  1 TinyNeRF: init, two skip
// chunks | six Linears + out;  2 View half: 4 + geometry twice | four Linears + out;  3 SIREN VolSDF: init, two skip chunks,
// the View half's ten | five Linears + sdf.out + the View half's twenty)
// (MODEL 0, round 4: the init / skip chunks of both MLPs are RECORDS too -- [hash | x] and the latent are one K64 group each,
// f16 + 2 x fp6 like the hidden groups -- so only the two geometry chunk pairs of the View MLP are left as pairs)
// MODEL 4 (round 4): a hash-encoded SkipConnMLP alone (D-NeRF's deformation network, src/nerf.py:1250-1257: 3 -> 5 x 256, skip 3,
// out 3 n + 1 <= 32 rows), rows to HBM: init group | skip group + 4 (L0) | L1 | L2 | skip group + 4 (L3) | L4 | out = 27 records
// MODEL 6 (round 4): PlainNeRF(view) + mip (src/nerf.py:256-261: the 96-wide integrated positional encoding as leading latent
// columns of BOTH MLPs): MODEL 0's schedule with two IPE K64 groups generated in the kernel wherever a Linear consumes them
// (first.init, first.L0, view.init, view.L0): 44 + 4 x 2 = 52 records, 2 geometry pairs
// MODEL 5 (round 4): a Fourier-encoded SkipConnMLP alone (VolSDF's MLP SDF network, src/sdf.py:250-258: 3 -> [p | sin, cos of 128
// frequencies] -> 6 x 256, skip 3, out 65), rows to HBM.  The 256 Fourier features are K64 groups in the HIDDEN format, generated by
// the row groups in a VALU phase wherever a Linear consumes them (init, L0, L3): init 4 | L0 4 + 4 | L1 | L2 | L3 4 + 4 | L4 | L5 |
// out 4 = 40 records, and 3 pairs for the 3-wide position chunk
__host__ __device__ constexpr int npair(int model) { return model == 1 ? 3 : model == 2 ? 2 : model == 3 ? 5 : model == 4 ? 0 : model == 5 ? 3 : 2; }  // (6: 2)
__host__ __device__ constexpr int nrec(int model) { return model == 1 ? 28 : model == 2 ? 22 : model == 3 ? 46 : model == 4 ? 27 : model == 5 ? 40 : model == 6 ? 52 : 44; }
__host__ __device__ constexpr int stream_rg(int model) { return npair(model) * PAIRB + nrec(model) * REC; }
__host__ __device__ constexpr int hdr_units(int model) { return npair(model) + nrec(model); }  // header word 2 of an F16X stream
}  // namespace x

template <int PREC>
struct Cfg {
  static constexpr int P = kTwoPlane<PREC> ? 2 : 1;
  static constexpr int NBLK = kTwoPlane<PREC> ? 2 : 4;         // 32-sample blocks per sample group
  static constexpr int FRAG = 1024 * P;                         // bytes of one fragment (hi plane [, lo plane])
  static constexpr int PAIR = 2 * FRAG;
  static constexpr int HREG = PREC == NA_PREC_F16X ? NBLK * x::BLKH : NBLK * 16 * FRAG;  // hidden activations of one group
  static constexpr int IREG = NBLK * 4 * FRAG;                  // init-input chunks of one group
  static constexpr int GROUP = HREG + IREG;                     // 80 KiB (f16x: 74 KiB)
  static constexpr int STREAM = kPairsPerPass * PAIR;           // weight stream of one row group
};

inline size_t packed_bytes_x(int model) { return (size_t)kHeaderBytes + kBiasBytes + 4 * (size_t)x::stream_rg(model); }
inline int model_of_pairs(int pairs) { return pairs == kTinyPairs ? 1 : pairs == kViewPairs ? 2 : pairs == kSirenPairs ? 3 : 0; }
inline size_t packed_bytes(int precision, int pairs = kPairsPerPass) {
  if (precision == NA_PREC_F16X) return (size_t)kHeaderBytes + kBiasBytes + 4 * (size_t)x::stream_rg(model_of_pairs(pairs));
  const int pair = 2048 * (precision == NA_PREC_BF16X3 ? 2 : 1);
  return (size_t)kHeaderBytes + kBiasBytes + 4 * (size_t)pairs * pair;
}

struct Args {
  const float* rays;     // [R,6]
  const float* ts;       // [T], or per-ray steps [R, ts_stride] (ts_stride = T: the fine pass of coarse -> fine rendering)
  int64_t ts_stride = 0; // 0: every ray marches the same steps
  const float* pts;      // nullable [T,R,3]
  const float4* tables;  // [8,65536]
  const char* packed;    // LS stream (na_render_ls_pack)
  float* alpha;          // nullable [T,R]
  float* weights;        // nullable [T,R]
  float* out;            // [R,3]
  const float* elaz;     // [R,2] elev/azim of every ray (ray_elaz_kernel)
  const float* feat;     // MODEL 2: [T*R, feat_ld] rows of the SDF network: column 0 = signed distance, 1..64 = latent
  const float* beta;     // MODEL 2: Laplace scale (one float)
  int feat_ld;
  int64_t R;
  int T, nb;
  int nG;                // sample groups of the launch (2 per workgroup): group G renders rays G, G + nG, G + 2 nG, ...
  int npg;               // passes per group: ceil(ceil(R / nG) * nb / NBLK)
  uint64_t nb_magic;     // 2^32 / nb + 1: block index / nb by multiplication
  int sigmoid_kind;
  int bg_kind;
  uint32_t packed_size;
  HashRes res;
  unsigned long long* trace;  // NA_LS_TRACE builds only: [2 groups][128] s_memtime stamps of workgroup 0, second pass
  uint32_t sat_gen;           // NA_PREC_F16X: this launch's id for the saturation flag (slot id % 256 of g_lsx_saturated)
  float* y;                   // MODEL 4: output rows [T * R, y_ld] (sample t * R + ray), n_out columns written
  int y_ld, n_out;
  // MODEL 6 (mip): the crop's geometry (rays = [B,H,W,6]: the pixel radius is a difference of neighbouring rows) and the IPE's shape
  int mip_H, mip_W, mip_kind, mip_min_deg, mip_nd;
  float mip_t_end;
};

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

// NA_PREC_F16X range guard.  IEEE half tops out at 65504; the LeakyReLU epilogue clamps there (act_apply) so that no inf / NaN
// is manufactured -- but a clamped activation is a WRONG finite value, and f16x is the mode whose claim is parity.  Every
// epilogue already has the block maximum in a register (it sets the fp6 scales): a maximum at the clamp, or a latent row beyond
// it, writes the launch's id here, and a tiny kernel behind the renderer turns the WHOLE frame into NaN when it finds its id
// (stream-ordered, no host synchronisation; ids instead of a reset: nothing to zero between launches).  Silence is never an
// option for the parity mode: switch to bf16x3 (fp32 range) for such weights.  tests/test_gpu_range.py.
// Round 5: the flag is PER LAUNCH, not per device -- a ring of NA_LSX_SAT_SLOTS words, launch id g owns slot g % SLOTS and a
// slot only ever matches the exact id, so two f16x launches in flight on different streams of one device cannot mask each
// other (one word, last writer wins, did: the earlier launch's poison pass found the later launch's id and left a clamped
// frame).  Two launches share a slot only if their ids are a multiple of 256 apart AND both are in flight at once; a renderer
// launch fills the chip (256 persistent workgroups), so 256 of them in flight is not a state the library can be driven into.
#ifndef NA_LSX_SAT_SLOTS  // (-DNA_LSX_SAT_SLOTS=1 rebuilds round 4's single word: tests/test_gpu_range.py's two-stream test then fails)
#define NA_LSX_SAT_SLOTS 256
#endif
static __device__ unsigned int g_lsx_saturated[NA_LSX_SAT_SLOTS] = {};
static __global__ void lsx_poison_kernel(uint32_t gen, float* __restrict__ out, int64_t n) {
  if (g_lsx_saturated[gen % NA_LSX_SAT_SLOTS] != gen) return;
  const float nan = __builtin_nanf("");
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = nan;
}

// NA_PREC_F16X stream schedules (pack side): the Linears of the model and which of them every pair / record / bias block packs
struct XLin { const float* W; const float* B; int in_dim, out_dim, desc; };  // nn.Linear layout [out,in]
struct XPairD { int8_t lin, q, skip; };      // init chunk q of Linear lin (skip: its columns sit behind the kHidden hidden ones)
// K64 group of Linear lin.  kind 0: hidden features 64 q .. 64 q + 63; kind 1 / 2: the init chunks 0..3 of the MLP (columns by
// init_slot_feature; 2: behind the kHidden hidden columns of a skip layer); kind 3 / 4: Fourier features 64 q .. 64 q + 63 in the
// generator's slot order (fourier_slot_col; 4: behind the hidden columns of a skip layer).  out_mode 0 hidden rows, 1 out row-major (tile
// min(rg, 2)), 2 out, one tile, 3 out split by block: row groups 0,1 hold tiles 0 and 1, row groups 2,3 tile 2
struct XRecD { int8_t lin, q, out_mode, kind; int16_t off; };  // off: added to the column (kinds 1, 2, 5: where the group's columns start)
struct XSched {
  int npair, nrec, nphase, nlin, ndesc;
  XLin lin[13];  // (<= 13 Linears: SIREN VolSDF 7 + 6)
  NaMlpDesc desc[2];
  XPairD pair[16];
  XRecD rec[56];
  int8_t bias_lin[16], bias_mode[16];
};
// model: 0 PlainNeRF(view) (w0 = first, w1 = View), 1 TinyNeRF (w0), 2 View half (w0), 3 SIREN VolSDF (w0 = SDF net, w1 = View).
// Defined in the NA_PREC_INST == 3 unit.
int render_lsx_pack(int model, const float* const* w0, const float* const* b0, const float* const* w1, const float* const* b1,
                    char* packed, hipStream_t stream, int n_out = 0);

template <int PREC, int AUX = 0>
__device__ __forceinline__ Frag<PREC> wload(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
  Frag<PREC> f;
  f.hi = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, AUX));
  if constexpr (kTwoPlane<PREC>)
    f.lo = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff + 1024, AUX));
  return f;
}


template <int PREC>
__device__ __forceinline__ Frag<PREC> fread(const char* p) {
  Frag<PREC> f;
  f.hi = *(const bf16x8*)p;
  if constexpr (kTwoPlane<PREC>) f.lo = *(const bf16x8*)(p + 1024);
  return f;
}

template <int PREC>
__device__ __forceinline__ void fwrite(char* p, const Frag<PREC>& f) {
  *(bf16x8*)p = f.hi;
  if constexpr (kTwoPlane<PREC>) *(bf16x8*)(p + 1024) = f.lo;
}

template <int PREC>
__device__ __forceinline__ void mma(f32x16& acc, const Frag<PREC>& A, const Frag<PREC>& B) {
  typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
  if constexpr (PREC == NA_PREC_BF16X3) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.lo, B.hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.hi, B.lo, acc, 0, 0, 0);
  }
  if constexpr (PREC == NA_PREC_F16X) {  // (init / geometry chunks: f16 hi + f16 lo, three products)
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A.lo), __builtin_bit_cast(f16x8, B.hi), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A.hi), __builtin_bit_cast(f16x8, B.lo), acc, 0, 0, 0);
  }
  if constexpr (kHalfElem<PREC>) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A.hi), __builtin_bit_cast(f16x8, B.hi), acc, 0, 0, 0);
  } else {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.hi, B.hi, acc, 0, 0, 0);
  }
}

// acc = cin + A x B (the first product of a phase reads the bias registers as its C operand: no copy into the accumulators)
template <int PREC>
__device__ __forceinline__ void mma_c(f32x16& acc, const f32x16& cin, const Frag<PREC>& A, const Frag<PREC>& B) {
  typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
  static_assert(PREC == NA_PREC_F16X, "mma_c: f16x only so far");
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A.lo), __builtin_bit_cast(f16x8, B.hi), cin, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A.hi), __builtin_bit_cast(f16x8, B.lo), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A.hi), __builtin_bit_cast(f16x8, B.hi), acc, 0, 0, 0);
}

// floats [slot][hi(2)][16] of one phase's bias block -> accumulator init of tile `slot`.  Buffer loads with the
// wave-uniform part in the scalar offset: no per-lane 64-bit pointers to hoist and spill.
__device__ __forceinline__ f32x16 bias_tile(__amdgpu_buffer_rsrc_t rs, int bias_soff, int slot, int lane) {
  const int voff = (lane >> 5) * 64;
  f32x16 a;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, bias_soff + slot * 128 + q * 16, 0));
    a[q * 4 + 0] = v[0]; a[q * 4 + 1] = v[1]; a[q * 4 + 2] = v[2]; a[q * 4 + 3] = v[3];
  }
  return a;
}

// ---- MFMA phase of a 256-row Linear: K = [NI init chunks from LDS | NH hidden chunks from LDS | geometry chunk]
// GEO: 0 none, 1 raw (view.init), 2 through the activation (skip layer).  geo(b) builds block b's fragment in registers.
template <int PREC, int RING0, int NI, int GEO, int NH, bool WRAP, int PPP = kPairsPerPass, class GeoLoad, class GeoMake>
__device__ __forceinline__ void m_hidden(f32x16 (&acc)[2][Cfg<PREC>::NBLK], Frag<PREC> (&ring)[kPF][2], int& cur,
                                         __amdgpu_buffer_rsrc_t rs, int wvoff, const char* hb, const char* ib, int lane,
                                         GeoLoad geo_load, GeoMake geo_make) {
  constexpr int NB = Cfg<PREC>::NBLK, FR = Cfg<PREC>::FRAG, PAIR = Cfg<PREC>::PAIR;
  constexpr int NL = NI + NH;                 // chunks whose B fragments come from LDS
  constexpr int NCH = NL + (GEO ? 1 : 0);
  // the partner wave on this SIMD is in a VALU-dense epilogue: MFMA issue must win the arbitration
  __builtin_amdgcn_s_setprio(1);
  auto bsrc = [&](int q, int b) -> Frag<PREC> {
    if (q < NI) return fread<PREC>(ib + (b * 4 + q) * FR + lane * 16);
    return fread<PREC>(hb + (b * 16 + (q - NI)) * FR + lane * 16);
  };
  auto refill = [&](int q) {
    int nx = cur + q + kPF;
    if (WRAP) nx = nx >= PPP ? nx - PPP : nx;
    ring[(RING0 + q) % kPF][0] = wload<PREC>(rs, wvoff, nx * PAIR);
    ring[(RING0 + q) % kPF][1] = wload<PREC>(rs, wvoff, nx * PAIR + FR);
  };
  // raw inputs of the geometry chunk (t or the explicit position of the lane's sample): requested two (bf16) or five (bf16x3)
  // chunks before the chunk that uses them; under the last LDS-fed chunk their latency was exposed (view.L0 7.3 k cycles)
  decltype(geo_load(0)) graw[GEO != 0 ? NB : 1];
  constexpr int LEAD = NB == 4 ? 2 : 5;  // (bf16 has no registers to hold them longer without spilling)
  constexpr int GQ = NL > LEAD ? NL - LEAD : 0;
  if constexpr (NB == 4) {
    // bf16 (4 blocks): ONE set of B fragments, refilled in place -- block b's fragment of chunk q+1 is requested right
    // after its two MFMAs of chunk q have issued and has the other three blocks' MFMAs (192 cycles) to arrive.  Halves
    // the fragment registers (16 instead of 32), which is what keeps this kernel inside 256 VGPRs without scratch.
    Frag<PREC> Bs[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) Bs[b] = bsrc(0, b);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < NL; ++q) {
      const Frag<PREC> A0 = ring[(RING0 + q) % kPF][0], A1 = ring[(RING0 + q) % kPF][1];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        mma<PREC>(acc[0][b], A0, Bs[b]);
        mma<PREC>(acc[1][b], A1, Bs[b]);
        if (q + 1 < NL) Bs[b] = bsrc(q + 1, b);
        if constexpr (GEO != 0) {
          if (q == GQ) graw[b] = geo_load(b);
        }
        if (b == 0) refill(q);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else {
    Frag<PREC> Bq[2][NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) Bq[0][b] = bsrc(0, b);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < NL; ++q) {
      const Frag<PREC> A0 = ring[(RING0 + q) % kPF][0], A1 = ring[(RING0 + q) % kPF][1];
      if (q + 1 < NL) {
#pragma unroll
        for (int b = 0; b < NB; ++b) Bq[(q + 1) & 1][b] = bsrc(q + 1, b);
      }
      if constexpr (GEO != 0) {
        if (q == GQ) {
#pragma unroll
          for (int b = 0; b < NB; ++b) graw[b] = geo_load(b);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        mma<PREC>(acc[0][b], A0, Bq[q & 1][b]);
        mma<PREC>(acc[1][b], A1, Bq[q & 1][b]);
        if (b == 0) refill(q);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if constexpr (GEO != 0) {
    constexpr int q = NL;
    const Frag<PREC> A0 = ring[(RING0 + q) % kPF][0], A1 = ring[(RING0 + q) % kPF][1];
    refill(q);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const Frag<PREC> B = geo_make(b, graw[b], GEO == 2);
      mma<PREC>(acc[0][b], A0, B);
      mma<PREC>(acc[1][b], A1, B);
    }
  }
  __builtin_amdgcn_s_setprio(0);
  cur += NCH;
}

// ---- MFMA phase of an out Linear, block-per-wave: NT 32-row tiles x 16 chunks for block `blk`; fragment f = c*NT + j
template <int PREC, int RING0, int NT, bool WRAP, int PPP = kPairsPerPass>
__device__ __forceinline__ void m_out(f32x16 (&o)[NT], Frag<PREC> (&ring)[kPF][2], int& cur, __amdgpu_buffer_rsrc_t rs,
                                      int wvoff, const char* hb, int lane, int blk) {
  constexpr int FR = Cfg<PREC>::FRAG, PAIR = Cfg<PREC>::PAIR;
  __builtin_amdgcn_s_setprio(1);
  Frag<PREC> Bq[2];
  const char* src = hb + blk * 16 * FR + lane * 16;
  Bq[0] = fread<PREC>(src);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    if (c + 1 < 16) Bq[(c + 1) & 1] = fread<PREC>(src + (c + 1) * FR);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int f = c * NT + j, p = f >> 1, t = f & 1;
      mma<PREC>(o[j], ring[(RING0 + p) % kPF][t], Bq[c & 1]);
      if (t == 1) {
        int nx = cur + p + kPF;
        if (WRAP) nx = nx >= PPP ? nx - PPP : nx;
        ring[(RING0 + p) % kPF][0] = wload<PREC>(rs, wvoff, nx * PAIR);
        ring[(RING0 + p) % kPF][1] = wload<PREC>(rs, wvoff, nx * PAIR + FR);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  __builtin_amdgcn_s_setprio(0);
  cur += NT * 8;
}

// N fragment pairs of the stream that carry no work (padding): keep the ring and `cur` in step
template <int PREC, int RING0, int N, bool WRAP, int PPP>
__device__ __forceinline__ void ring_skip(Frag<PREC> (&ring)[kPF][2], int& cur, __amdgpu_buffer_rsrc_t rs, int wvoff) {
  constexpr int FR = Cfg<PREC>::FRAG, PAIR = Cfg<PREC>::PAIR;
#pragma unroll
  for (int p = 0; p < N; ++p) {
    int nx = cur + p + kPF;
    if (WRAP) nx = nx >= PPP ? nx - PPP : nx;
    ring[(RING0 + p) % kPF][0] = wload<PREC>(rs, wvoff, nx * PAIR);
    ring[(RING0 + p) % kPF][1] = wload<PREC>(rs, wvoff, nx * PAIR + FR);
  }
  cur += N;
}

// ---- MFMA phase of `first.out` (65 rows = 3 tiles), ROW-major like the hidden layers: row group rg computes tile
// min(rg, 2) for all NBLK blocks of its sample group (16 chunks x NBLK MFMAs, one A fragment feeds NBLK MFMAs; row group 3
// repeats tile 2 to keep the four weight rings in step, its result is dropped).  Block-per-wave (every wave streaming all
// three tiles for its own block: 48 KiB of fragments) was bound by weight delivery: 3.9 k cycles for 1.5 k of MFMA work in
// bf16, 7.2 k for 4.6 k in bf16x3, where two of the four waves did redundant work on top.
template <int PREC, int RING0, bool WRAP>
__device__ __forceinline__ void m_out_rows(f32x16 (&o)[Cfg<PREC>::NBLK], Frag<PREC> (&ring)[kPF][2], int& cur,
                                           __amdgpu_buffer_rsrc_t rs, int wvoff, const char* hb, int lane) {
  constexpr int NB = Cfg<PREC>::NBLK, FR = Cfg<PREC>::FRAG, PAIR = Cfg<PREC>::PAIR;
  __builtin_amdgcn_s_setprio(1);
  Frag<PREC> Bq[2][NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) Bq[0][b] = fread<PREC>(hb + (b * 16) * FR + lane * 16);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const int p = c >> 1, t = c & 1;
    const Frag<PREC> A = ring[(RING0 + p) % kPF][t];
    if (c + 1 < 16) {
#pragma unroll
      for (int b = 0; b < NB; ++b) Bq[(c + 1) & 1][b] = fread<PREC>(hb + (b * 16 + c + 1) * FR + lane * 16);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int b = 0; b < NB; ++b) mma<PREC>(o[b], A, Bq[c & 1][b]);
    if (t == 1) {
      int nx = cur + p + kPF;
      if (WRAP) nx = nx >= kPairsPerPass ? nx - kPairsPerPass : nx;
      ring[(RING0 + p) % kPF][0] = wload<PREC>(rs, wvoff, nx * PAIR);
      ring[(RING0 + p) % kPF][1] = wload<PREC>(rs, wvoff, nx * PAIR + FR);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  __builtin_amdgcn_s_setprio(0);
  cur += 8;
}

// ---- epilogue of a 256-row Linear: act(acc) -> the group's hidden fragments in LDS (in place)
template <int PREC, int ACT, int T0 = 0, int T1 = 2>
__device__ __forceinline__ void store_acts(const f32x16 (&acc)[2][Cfg<PREC>::NBLK], char* hb, int rg, int lane) {
  constexpr int NB = Cfg<PREC>::NBLK, FR = Cfg<PREC>::FRAG;
#pragma unroll
  for (int t = T0; t < T1; ++t)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      Frag<PREC> f0, f1;
      acc_to_frags<PREC, ACT>(acc[t][b], f0, f1);
      char* dst = hb + (b * 16 + 2 * (2 * rg + t)) * FR + lane * 16;
      fwrite<PREC>(dst, f0);
      fwrite<PREC>(dst + FR, f1);
    }
}

// the skip connection re-enters through the activation (src/neural_blocks.py:291-293): act() on the init chunks of
// block `blk`, in place, once the init Linear has consumed the raw values
template <int PREC, int ACT, int NCH>
__device__ __forceinline__ void activate_init(char* ib, int blk, int lane) {
  constexpr int FR = Cfg<PREC>::FRAG;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    char* p = ib + (blk * 4 + c) * FR + lane * 16;
    Frag<PREC> f = fread<PREC>(p);
    frag_activate<PREC, ACT>(f);
    fwrite<PREC>(p, f);
  }
}

// ================================================================================================ NA_PREC_F16X phases
namespace x {
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(6))) int i32x6;
typedef __attribute__((ext_vector_type(12))) uint32_t u32x12;

// acc += A x B, both fp6 e2m3: A = 6 dwords with its E8M0 scale in byte SA of sa, B = 8 dwords from LDS: 0..5 the values, 6 its
// scale (byte 0)
template <int SA>
__device__ __forceinline__ void mma6(f32x16& acc, const i32x8& A, int sa, const i32x8& B) {
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc, 2, 2, SA, sa, 0, B[6]);
}
struct PairR {  // one init / geometry chunk pair: tiles 0 and 1, f16 hi and lo planes
  Frag<NA_PREC_F16X> t0, t1;
};
typedef __attribute__((ext_vector_type(16))) uint32_t u32x16;
struct Regs {   // weight registers that live across phases
  PairR pr[2];       // pair ring: slot i & 1 holds pair i
  // f16 fragments of the current record, one 16-dword vector per tile (chunk c = dwords 4 c .. 4 c + 3: the conversion that
  // derives WT6 takes the tile's 32 halves from sixteen consecutive registers), refilled in place with the next record's
  u32x16 a16[2];
  // WL6 of both tiles (dwords 0..5 tile 0, 6..11 tile 1) of the current record; the next record's are requested right behind
  // the group's scaled MFMAs and have the next group's sixteen f16 MFMAs to arrive (a second buffer costs registers the kernel
  // does not have).  A clang vector: as a {u32x4, u32x2} struct member it stayed in scratch memory
  u32x12 a6;
  int asc[2];        // scale dwords: record i's in slot i & 1, requested a whole K64 group ahead (the WT6 conversion needs it early)
};

__device__ __forceinline__ PairR wpair(__amdgpu_buffer_rsrc_t rs, int lane, int xbase, int i) {
  PairR p;
  p.t0 = wload<NA_PREC_F16X>(rs, lane * 16, xbase + i * PAIRB);
  p.t1 = wload<NA_PREC_F16X>(rs, lane * 16, xbase + i * PAIRB + 2048);
  return p;
}
// record loads: soff = a 4-KiB-aligned scalar base inside the record, the rest of the offset is an instruction immediate
__device__ __forceinline__ u32x4 wload16(__amdgpu_buffer_rsrc_t rs, int lane, int roff, int t, int c) {
  return __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16 + c * 1024, roff + t * 4096, 0);
}
__device__ __forceinline__ f16x8 a16frag(const u32x16& v, int c) {
  return __builtin_bit_cast(f16x8, u32x4{v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]});
}
__device__ __forceinline__ void a16set(u32x16& v, int c, const u32x4& q) {
  v[4 * c] = q[0]; v[4 * c + 1] = q[1]; v[4 * c + 2] = q[2]; v[4 * c + 3] = q[3];
}
// WL6 of both tiles: dwords 0..5 tile 0, 6..11 tile 1
__device__ __forceinline__ u32x12 wload6(__amdgpu_buffer_rsrc_t rs, int lane, int roff) {
  const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, roff + 8192, 0);
  const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16 + 1024, roff + 8192, 0);
  const u32x4 c = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16 + 2048, roff + 8192, 0);
  return u32x12{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3], c[0], c[1], c[2], c[3]};
}
__device__ __forceinline__ int wloadsc(__amdgpu_buffer_rsrc_t rs, int lane, int roff) {
  return (int)__builtin_amdgcn_raw_buffer_load_b32(rs, lane * 4, roff + 8192 + 3072, 0);
}
__device__ __forceinline__ i32x8 lo6(const u32x12& v) { return i32x8{(int)v[0], (int)v[1], (int)v[2], (int)v[3], (int)v[4], (int)v[5], 0, 0}; }
__device__ __forceinline__ i32x8 hi6(const u32x12& v) { return i32x8{(int)v[6], (int)v[7], (int)v[8], (int)v[9], (int)v[10], (int)v[11], 0, 0}; }
__device__ __forceinline__ i32x8 op6(const i32x6& v) { return i32x8{v[0], v[1], v[2], v[3], v[4], v[5], 0, 0}; }
// 32 halves (sixteen consecutive registers) -> 32 fp6 e2m3 in element order, divided by the scale's power of two.  Early-clobber
// like cvt_fp6_disjoint below: the multi-pass conversions write their destination while they still read their operands.
__device__ __forceinline__ i32x6 cvt_fp6_f16_disjoint(const u32x16& h, float scale) {
  i32x6 d;
  asm("v_cvt_scalef32_pk32_fp6_f16 %0, %1, %2" : "=&v"(d) : "v"(h), "v"(scale));
  return d;
}

// ---- N init chunk pairs (pairs I0 .. I0+N-1 of the pass) against the init chunks 0..N-1 of the NB blocks: three f16 products
// (the phase's first products take the bias registers `cb` as their C operand: the accumulators are written, never initialised)
template <int I0, int N, int NB, bool TAIL = false>
__device__ __forceinline__ void pairs(f32x16 (&acc)[2][NB], const f32x16 (&cb)[2], Regs& R, __amdgpu_buffer_rsrc_t rs, int xbase,
                                      const char* ib, int lane) {
  constexpr int PREC = NA_PREC_F16X, FR = 2048;
  if (NA_LSX_PRIO == 0) __builtin_amdgcn_s_setprio(1);
  Frag<PREC> Bq[2][NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) Bq[0][b] = fread<PREC>(ib + (b * 4) * FR + lane * 16);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int q = 0; q < N; ++q) {
    const int i = I0 + q;
    if (q + 1 < N) {
#pragma unroll
      for (int b = 0; b < NB; ++b) Bq[(q + 1) & 1][b] = fread<PREC>(ib + (b * 4 + q + 1) * FR + lane * 16);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (q == 0) {
        mma_c<PREC>(acc[0][b], cb[0], R.pr[i & 1].t0, Bq[q & 1][b]);
        mma_c<PREC>(acc[1][b], cb[1], R.pr[i & 1].t1, Bq[q & 1][b]);
      } else {
        mma<PREC>(acc[0][b], R.pr[i & 1].t0, Bq[q & 1][b]);
        mma<PREC>(acc[1][b], R.pr[i & 1].t1, Bq[q & 1][b]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (q + 2 < N + (TAIL ? 1 : 0)) R.pr[i & 1] = wpair(rs, lane, xbase, i + 2);  // (the phase's own pairs only, TAIL: + geometry)
    __builtin_amdgcn_sched_barrier(0);
  }
  if (NA_LSX_PRIO == 0) __builtin_amdgcn_s_setprio(0);
}
// the first two pairs of the NEXT pair phase: issued at the end of the epilogue in front of it (holding them across the
// epilogues of the hidden layers costs 32 registers the residual / fp6 conversion needs)
__device__ __forceinline__ void pairs_prefetch(Regs& R, __amdgpu_buffer_rsrc_t rs, int xbase, int lane, int i0, int n = 2) {
  R.pr[i0 & 1] = wpair(rs, lane, xbase, i0);
  if (n > 1) R.pr[(i0 + 1) & 1] = wpair(rs, lane, xbase, i0 + 1);
}

// ---- the geometry chunk pair (pair I of the pass): block b's fragment is built in registers by geo_make
template <int I, int NB, class GeoRawT, class GeoMake>
__device__ __forceinline__ void geo_pair(f32x16 (&acc)[2][NB], Regs& R, __amdgpu_buffer_rsrc_t rs, int xbase, int lane,
                                         const GeoRawT (&graw)[NB], GeoMake geo_make, bool act) {
  constexpr int PREC = NA_PREC_F16X;
  if (NA_LSX_PRIO == 0) __builtin_amdgcn_s_setprio(1);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const Frag<PREC> B = geo_make(b, graw[b], act);
    mma<PREC>(acc[0][b], R.pr[I & 1].t0, B);
    mma<PREC>(acc[1][b], R.pr[I & 1].t1, B);
  }
  if (NA_LSX_PRIO == 0) __builtin_amdgcn_s_setprio(0);
}

// ---- NG records starting at record rec0: per K64 group the f16 chunks, then the two fp6 correction products.  NT tiles (2:
// hidden Linear; 1: out Linear, tile 0 of the record) x NBk blocks.  The groups' B operands: with G0 = 1 the FIRST record is an
// init group, read from the init region (block b at ib0 + b * KQ; NCH0 of its four f16 chunks carry data: 3 for [hash | x],
// whose fourth chunk is padding that the fp6 operands hold as zeros and the f16 product skips); the others are the hidden
// groups Q = 0, 1, ... of the blocks (hb0 + b * BLKH + Q * KQ).  PAR0 = parity of rec0 (which scale slot its dword sits in).
// CB: the phase starts here -- the first MFMA of every accumulator reads the bias registers cb[t] as its C operand.
template <int NT, int NBk, bool CB, int NREC, int NG = 4, int G0 = 0, int PAR0 = 0, int NCH0 = 4, bool LAST0 = false, int BSTR = BLKH,
          int NCHL = 4, int NTAIL = 1>  // NCHL: live f16 chunks of the call's last NTAIL records (MODEL 6: an IPE group fills three)
__device__ __forceinline__ void recs(f32x16 (&acc)[NT][NBk], const f32x16 (&cb)[NT], Regs& R, __amdgpu_buffer_rsrc_t rs, int xrec,
                                     int rec0, const char* hb0, int lane, const char* ib0 = nullptr) {
  if (NA_LSX_PRIO == 0) __builtin_amdgcn_s_setprio(1);
  auto gbase = [&](int gi, int b) -> const char* {  // K64 group gi of this call, block b
    if (G0 != 0 && gi == 0) return ib0 + b * KQ;
    return hb0 + b * BSTR + (gi - G0) * KQ;  // (BSTR: MODEL 6 parks the two IPE groups of block b at groups 2 b, 2 b + 1 of block 0)
  };
  auto nch = [&](int gi) -> int { return (G0 != 0 && gi == 0) ? NCH0 : (gi >= NG - NTAIL ? NCHL : 4); };
  auto b16 = [&](int b, int gi, int c) -> f16x8 { return *(const f16x8*)(gbase(gi, b) + c * 1024 + lane * 16); };
  auto b6 = [&](int b, int gi, int k) -> i32x8 {  // k: 0 R, 1 T
    const char* p = gbase(gi, b) + 4096 + k * 2048 + lane * 16;
    const u32x4 a = *(const u32x4*)p, c = *(const u32x4*)(p + 1024);
    return i32x8{(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)c[0], (int)c[1], (int)c[2], (int)c[3]};
  };
  f16x8 Bq[2][NBk];
  i32x8 B6[NBk][2];
#pragma unroll
  for (int b = 0; b < NBk; ++b) Bq[0][b] = b16(b, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
  bool first = true;  // (no MFMA of this call has issued yet: the next one takes the bias as its C operand)
  int cur = 0;        // which half of Bq holds the chunk about to be consumed
#pragma unroll
  for (int gi = 0; gi < NG; ++gi) {
    int nrc = rec0 + gi + 1;
    nrc = nrc >= NREC ? 0 : nrc;
    const int noff = __builtin_amdgcn_readfirstlane(xrec + nrc * REC);
    // the NEXT record's scale bytes, a whole group ahead (its WT6 conversion runs behind the second chunk of its group)
    const int asc = R.asc[(PAR0 + gi) & 1];
    // (LAST0: a schedule with an ODD number of records per pass -- the record behind this call's last one is record 0 of the next
    // pass, whose scale belongs in slot 0 although the parity says 1; `asc` above was read first)
    if (!(NA_LSX_EXP & 2)) R.asc[(LAST0 && gi == NG - 1) ? 0 : (PAR0 + gi + 1) & 1] = wloadsc(rs, lane, noff);
    i32x6 wt[NT];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const bool live = c < nch(gi);  // (a padding chunk: no f16 product, its weight fragments are still streamed in step)
      // the B fragments of the next chunk that carries data (this group's, or chunk 0 of the next group)
      int ng = gi, nc = c + 1;
      if (nc >= nch(gi)) { ng = gi + 1; nc = 0; }
      if (live && ng < NG) {
#pragma unroll
        for (int b = 0; b < NBk; ++b) Bq[cur ^ 1][b] = b16(b, ng, nc);
      }
      if (c == 1) {  // this group's fp6 B operands: two chunks of lead
#pragma unroll
        for (int b = 0; b < NBk; ++b) { B6[b][0] = b6(b, gi, 0); B6[b][1] = (NA_LSX_EXP & 8) ? B6[b][0] : b6(b, gi, 1); }
      }
      __builtin_amdgcn_sched_barrier(0);
      const f16x8 A0 = a16frag(R.a16[0], c), A1 = a16frag(R.a16[1], c);
#pragma unroll
      for (int b = 0; b < NBk; ++b) {
        if (live) acc[0][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0, Bq[cur][b], (CB && first) ? cb[0] : acc[0][b], 0, 0, 0);
        // WT6 = fp6(f16 W / 2^scale) of this record, from the fragments while all four chunks are still in place (the newest,
        // chunk 3, was requested a group ago).  The conversion holds the wave's issue for ~46 cycles (measured: two of them
        // behind the MFMAs of a chunk cost 2.9 % of the frame), so each one sits directly behind ONE MFMA and runs in its shadow
        if (c == 1 && b == 0) {
          __builtin_amdgcn_sched_barrier(0);
          const float sc = __builtin_bit_cast(float, (((uint32_t)asc >> 8) & 0xFFu) << 23);
          wt[0] = (NA_LSX_EXP & 32) ? i32x6{(int)R.a16[0][0], (int)R.a16[0][1], (int)R.a16[0][2], (int)R.a16[0][3], (int)R.a16[0][4], (int)R.a16[0][5]}
                                    : cvt_fp6_f16_disjoint(R.a16[0], sc);
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (NT == 2) {
          if (live) acc[1][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1, Bq[cur][b], (CB && first) ? cb[NT - 1] : acc[1][b], 0, 0, 0);
          if (c == 1 && b == 0) {
            __builtin_amdgcn_sched_barrier(0);
            const float sc = __builtin_bit_cast(float, (((uint32_t)asc >> 24) & 0xFFu) << 23);
            wt[NT - 1] = (NA_LSX_EXP & 32) ? i32x6{(int)R.a16[1][0], (int)R.a16[1][1], (int)R.a16[1][2], (int)R.a16[1][3], (int)R.a16[1][4], (int)R.a16[1][5]}
                                           : cvt_fp6_f16_disjoint(R.a16[1], sc);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        if (b == NBk - 1 && c >= 1 && !(NA_LSX_EXP & 4)) {  // (behind the chunk's last MFMA: chunks 0 and 1 only once WT6 exists)
          if (c == 1) {
            a16set(R.a16[0], 0, wload16(rs, lane, noff, 0, 0));
            a16set(R.a16[1], 0, wload16(rs, lane, noff, 1, 0));
          }
          a16set(R.a16[0], c, wload16(rs, lane, noff, 0, c));
          a16set(R.a16[1], c, wload16(rs, lane, noff, 1, c));
        }
      }
      if (live) { first = false; cur ^= 1; }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int b = 0; b < NBk; ++b) {
      // W_lo x T(x) and W_top x R(x)
      mma6<0>(acc[0][b], lo6(R.a6), asc, B6[b][1]);
      mma6<1>(acc[0][b], op6(wt[0]), asc, B6[b][0]);
      if constexpr (NT == 2) {
        mma6<2>(acc[1][b], hi6(R.a6), asc, B6[b][1]);
        mma6<3>(acc[1][b], op6(wt[NT - 1]), asc, B6[b][0]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (!(NA_LSX_EXP & 2)) R.a6 = wload6(rs, lane, noff);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (NA_LSX_PRIO == 0) __builtin_amdgcn_s_setprio(0);
}

// v_cvt_scalef32_2xpk16_fp6_f32 writes its six destination registers while it still reads its scale and the tails of its
// sources (tools/hw/cvt_fp6_overlap.hip), and the compiler's builtin does not say so: this form marks the destination
// early-clobber, i.e. disjoint from every operand.  (The builtin form lets the allocator put the destination on the first six
// registers of a source, which the hardware handles and which saves six registers; nerf_atlas_amd/build.py checks every
// instance of the listing either way.)
#ifndef NA_LSX_CVT_ASM
#define NA_LSX_CVT_ASM 1  // the activation stores of the render kernel: 1 early-clobber asm (450 against 454 Msamples/s, same frame bit for bit), 0 builtin
#endif
__device__ __forceinline__ i32x6 cvt_fp6_disjoint(const f32x16& a, const f32x16& b, float scale) {
  i32x6 d;
  asm("v_cvt_scalef32_2xpk16_fp6_f32 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(scale));
  return d;
}

// ---- epilogue of a hidden Linear: the lane's 32 values of block b (accumulators of the row group's two tiles) -> the LDS
// operands of K64 group rg: f16 fragments, fp6 residual plane R, fp6 value plane T, scale bytes
// (NCHW: f16 chunks written -- the [hash | x] group leaves its padding chunk alone: the compositing partials live there)
// KEEP7: the last dword of the lane's T operand (never read by the MFMA) is left alone -- the View MLP's latent group keeps the
// block's density there from the epilogue of first.out to the compositing at the end of the pass
template <int ACT, int NCHW = 4, bool KEEP7 = false>
__device__ __forceinline__ void store_block(char* kq, const f32x16& a0, const f32x16& a1, int lane, uint32_t sat_gen) {
  constexpr int PREC = NA_PREC_F16X;
  f32x16 v0, v1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { v0[r] = act_apply<ACT, PREC>(a0[r]); v1[r] = act_apply<ACT, PREC>(a1[r]); }
  // The inline-asm consumers below (v_max3_f32, v_fma_mix_f32, the fp6 conversions) are INVISIBLE to the compiler's hazard
  // recogniser, and being non-volatile they may be scheduled across a barrier right behind the instruction that produces their
  // operand.  Two hardware rules then go unprotected (probes tools/hw/mfma_use_hazard.hip, trans_use_hazard.hip): an MFMA result
  // is only complete passes + 4 = 12 wait states after issue (ACT = NONE passes accumulators straight through), a transcendental
  // result (v_sin_f32) one wait state after.  A volatile fence that owns the 32 values and spends those wait states makes the
  // consumers safe by construction; build.check_mfma_use / check_trans_use verify every listing.  (Round 4: the latent group of
  // the mip renderer read first.out's accumulators 3 wait states behind the MFMA: run-to-run last-bit differences.)
  if constexpr (ACT == NA_ACT_NONE) asm volatile("s_nop 7\n\ts_nop 3" : "+v"(v0), "+v"(v1));
  else if constexpr (ACT == NA_ACT_SIN) asm volatile("s_nop 0" : "+v"(v0), "+v"(v1));
  else asm volatile("" : "+v"(v0), "+v"(v1));
  uint32_t pk[16];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    pk[u] = __builtin_bit_cast(uint32_t, f16x2{(_Float16)v0[2 * u], (_Float16)v0[2 * u + 1]});
    pk[8 + u] = __builtin_bit_cast(uint32_t, f16x2{(_Float16)v1[2 * u], (_Float16)v1[2 * u + 1]});
  }
#pragma unroll
  for (int c = 0; c < NCHW; ++c) *(u32x4*)(kq + c * 1024 + lane * 16) = u32x4{pk[4 * c], pk[4 * c + 1], pk[4 * c + 2], pk[4 * c + 3]};
  // E8M0 scales from the block maximum: T = v / 2^(e-2) lands in [4, 8) (fp6 e2m3 saturates at 7.5: 3 % at worst on a
  // correction operand), R = (v - f16 v) / 2^(e-13) in [-4, 4].  After a sine |v| <= 1: fixed scales, no maximum.
  int eT, eR;
  if constexpr (ACT == NA_ACT_SIN) {
    eT = 125; eR = 114;
  } else {
    float m = 0.f;
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(m) : "v"(v0[r]), "v"(v0[r + 1]));
      asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(m) : "v"(v1[r]), "v"(v1[r + 1]));
    }
    const int ev = (int)(__builtin_bit_cast(uint32_t, m) >> 23);
    eT = ev > 3 ? ev - 2 : 1;
    eR = ev > 14 ? ev - 13 : 1;
    if (__builtin_expect(!(m < 65504.0f), 0)) g_lsx_saturated[sat_gen % NA_LSX_SAT_SLOTS] = sat_gen;  // an activation sits at the half clamp (or is NaN)
  }
  const float sT = __builtin_bit_cast(float, (uint32_t)eT << 23);
  const float sR = __builtin_bit_cast(float, (uint32_t)eR << 23);
  // The residual plane R pairs with WT6, which the consumer derives from its f16 weight fragments in THEIR element order: slot
  // s = 8 c + e <-> chunk c, element e = value n[s] with n = (v0[0..15], v1[0..15]).  The conversion below puts a[i] into slot
  // 2 i and b[i] into slot 2 i + 1, so a = the even-indexed n, b = the odd-indexed n (just which register each residual is
  // written to).  The value plane T pairs with the streamed WL6 and keeps the interleaved order (v0[i], v1[i]).
  f32x16 r0, r1;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    // v - float(f16 half of the packed dword) in ONE instruction (the compiler's own sequence re-converts: 3.5 ops per value)
    float a, b, c, d;
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(a) : "v"(v0[2 * u]), "v"(pk[u]));
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(b) : "v"(v0[2 * u + 1]), "v"(pk[u]));
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(c) : "v"(v1[2 * u]), "v"(pk[8 + u]));
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(d) : "v"(v1[2 * u + 1]), "v"(pk[8 + u]));
    r0[u] = a; r1[u] = b; r0[8 + u] = c; r1[8 + u] = d;
  }
  // v_cvt_scalef32_2xpk16_fp6_f32 divides by the scale's power of two, rounds to nearest even, saturates, and puts a[i] into
  // slot 2 i, b[i] into slot 2 i + 1 (probed on the hardware: tools/proto/ls_f16x.py calibrate)
  const i32x6 Rr = NA_LSX_CVT_ASM ? cvt_fp6_disjoint(r0, r1, sR) : __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(r0, r1, sR);
  const i32x6 Tt = NA_LSX_CVT_ASM ? cvt_fp6_disjoint(v0, v1, sT) : __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(v0, v1, sT);
  char* p = kq + 4096 + lane * 16;
  *(u32x4*)p = u32x4{(uint32_t)Rr[0], (uint32_t)Rr[1], (uint32_t)Rr[2], (uint32_t)Rr[3]};
  *(u32x4*)(p + 1024) = u32x4{(uint32_t)Rr[4], (uint32_t)Rr[5], (uint32_t)eR, 0u};
  if (!(NA_LSX_EXP & 16)) {
    *(u32x4*)(p + 2048) = u32x4{(uint32_t)Tt[0], (uint32_t)Tt[1], (uint32_t)Tt[2], (uint32_t)Tt[3]};
    if constexpr (KEEP7) {
      typedef __attribute__((ext_vector_type(3))) uint32_t u32x3;
      *(u32x3*)(p + 3072) = u32x3{(uint32_t)Tt[4], (uint32_t)Tt[5], (uint32_t)eT};
    } else {
      *(u32x4*)(p + 3072) = u32x4{(uint32_t)Tt[4], (uint32_t)Tt[5], (uint32_t)eT, 0u};
    }
  }
}
// the latent rows (no activation in front of them: to_elem clamps them to the half range) are checked the same way
__device__ __forceinline__ void latent_range(const f32x16& v_in, uint32_t sat_gen) {
  f32x16 v = v_in;
  asm volatile("s_nop 7\n\ts_nop 3" : "+v"(v));  // (accumulators read by inline asm: see store_block)
  float m = 0.f;
#pragma unroll
  for (int r = 0; r < 16; r += 2) asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(m) : "v"(v[r]), "v"(v[r + 1]));
  if (__builtin_expect(!(m < 65504.0f), 0)) g_lsx_saturated[sat_gen % NA_LSX_SAT_SLOTS] = sat_gen;
}
// (blocks B0 .. B1 - 1: an epilogue that also re-enters an init group stores block 0, converts the group -- whose raw values
// wait in the wave's K64 region of block 1 -- with half of the accumulators already dead, then stores block 1)
template <int ACT, int NB, int B0 = 0, int B1 = NB>
__device__ __forceinline__ void store_acts(const f32x16 (&acc)[2][NB], char* hb, int rg, int lane, uint32_t sat_gen) {
  if (NA_LSX_PRIO == 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
  for (int b = B0; b < B1; ++b) store_block<ACT>(hb + b * BLKH + rg * KQ, acc[0][b], acc[1][b], lane, sat_gen);
  if (NA_LSX_PRIO == 2) __builtin_amdgcn_s_setprio(0);
}
}  // namespace x

// ---- compositing arithmetic: fast_exp / fast_sigmoid / fast_softplus live in common.h (hardware transcendentals).
__device__ __forceinline__ float fast_sigmoid_kind(float v, int kind) {
  switch (kind) {  // the sigmoid family on the fast path, everything else as in apply_sigmoid_kind
    case NA_SIG_NORMAL: return fast_sigmoid(v);
    case NA_SIG_THIN: return (fast_sigmoid(v) * (1.f + 2.f * -1e-2f) - -1e-2f) + 1e-2f;
    case NA_SIG_FAT: return fast_sigmoid(v) * (1.f + 2.f * 1e-2f) - 1e-2f;
    case NA_SIG_UPSHIFTED: return fast_sigmoid(v) + 1e-2f;
    default: return apply_sigmoid_kind(v, kind);
  }
}
// 32-lane scans on DPP (row shifts inside rows of 16 + row_bcast:15 into the odd rows): 5 VALU instructions instead of
// 5 dependent ds_bpermute round trips.  The two 32-lane halves of the wave scan independently.
#define NA_DPP(OLD, SRC, CTRL, ROWS) \
  __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, OLD), __builtin_bit_cast(int, SRC), CTRL, ROWS, 0xF, false))
__device__ __forceinline__ float scan32_mul(float x) {
  x *= NA_DPP(1.0f, x, 0x111, 0xF);
  x *= NA_DPP(1.0f, x, 0x112, 0xF);
  x *= NA_DPP(1.0f, x, 0x114, 0xF);
  x *= NA_DPP(1.0f, x, 0x118, 0xF);
  x *= NA_DPP(1.0f, x, 0x142, 0xA);
  return x;
}
__device__ __forceinline__ float scan32_add(float x) {
  x += NA_DPP(0.0f, x, 0x111, 0xF);
  x += NA_DPP(0.0f, x, 0x112, 0xF);
  x += NA_DPP(0.0f, x, 0x114, 0xF);
  x += NA_DPP(0.0f, x, 0x118, 0xF);
  x += NA_DPP(0.0f, x, 0x142, 0xA);
  return x;
}

template <int PREC, int MODEL = 0>
__global__ __launch_bounds__(512) void render_ls_kernel(Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using C = Cfg<PREC>;
  constexpr int NB = C::NBLK, FR = C::FRAG;
  constexpr bool M0 = MODEL == 0 || MODEL == 6;  // the PlainNeRF(view) schedule (6: + mip)
  constexpr bool MIP = MODEL == 6;
  constexpr int PPP = PREC == NA_PREC_F16X ? x::hdr_units(MODEL)
                      : MODEL == 1 ? kTinyPairs : MODEL == 2 ? kViewPairs : MODEL == 3 ? kSirenPairs : kPairsPerPass;  // pairs per pass and row group
  // rays / elaz are read with scalar (SMEM) loads below; both were written by kernels that ran just before this one, into
  // buffers the allocator recycles from call to call: drop whatever the scalar cache still holds of those addresses
  __builtin_amdgcn_s_dcache_inv();
  {
    // a stream packed for another precision or schedule (or not a stream at all) would be consumed without any fault:
    // refuse it -- NaN colour for every ray -- instead of rendering garbage (header: na_render_*_ls_pack)
    const uint32_t* hdr = (const uint32_t*)a.packed;
    if (hdr[0] != kMagic || hdr[1] != (uint32_t)PREC || hdr[2] != (uint32_t)PPP) {
      const float nan = __builtin_nanf("");
      for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.R * 3; i += (int64_t)gridDim.x * blockDim.x) a.out[i] = nan;
      return;
    }
  }
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int rg = wv & 3, g = wv >> 2;
  const int hi = lane >> 5, ln = lane & 31;
  char* hb = smem + g * C::GROUP;
  char* ib = hb + C::HREG;
  const bool owner = rg < NB;           // this wave owns block rg of its group (encoder, out layers, compositing)
  const int blk = owner ? rg : rg - NB;  // non-owners (bf16x3: rg 2,3) shadow a block to keep their weight ring in step
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.packed, 0, a.packed_size, 0x00020000);
  const int wvoff = kHeaderBytes + kBiasBytes + rg * (PPP * C::PAIR) + lane * 16;
  const int bias_rg = kHeaderBytes + rg * (kNPhase * 1024);  // scalar offset of this row group's bias blocks
  // Fifth init chunk of the View MLP for block b of this group (x, y, z, elev, azim in the hi = 0 lanes; src/refl.py:
  // 190-207).  The ray of a block is wave-uniform: its origin, direction and elev/azim (from the pre-kernel) are read once
  // per pass (in the short epilogue of first.out) and kept in SGPRs (geo_u, readfirstlane), together with the block's step
  // offset; the two MFMA phases that consume the chunk only load t (or the explicit position) of their lane at their start.
  float geo_u[NB][8];   // ox oy oz dx dy dz elev azim of block b's ray (uniform)
  int geo_t0[NB];       // first step of block b
  int geo_ray[NB];
  float own_u[6];       // origin | direction of the ray of this wave's OWN block (encoder + compositing), per pass
  float own_dn = 0.f;   // |direction| of that ray
  float prev_dn = 0.f;  // the same for the wave's block of the previous pass (its compositing runs one pass later)
  // Work distribution: sample group G = 2 * workgroup + g renders the rays G, G + nG, G + 2 nG, ... one after the other,
  // each as its nb 32-step blocks in step order, NB blocks per pass.  At any moment the launch works on ~nG consecutive
  // rays (hash-table locality in L2 as before), and the blocks of one ray pass through one group in order, so the
  // transmittance is carried from block to block inside the kernel (no per-block partials, no second launch).
  // XCD-aware order: hardware workgroup w runs on XCD w % 8 (round-robin dispatch, each XCD has its own L2), so logical
  // workgroup lw = (w % 8) * (n / 8) + w / 8 puts CONSECUTIVE sample groups -- neighbouring rays, the same hash-table
  // lines and weight fragments -- on one XCD instead of on all eight: HBM traffic per launch 173 -> 93 MB (bf16),
  // 210 -> 150 MB (bf16x3) at equal speed (profiles/r02).  Grids that are not a multiple of 8 keep the identity.
#ifndef NA_LS_NO_XCD_MAP
  const int nwg = (int)gridDim.x;
  const int lw = (nwg % 8 == 0) ? ((int)blockIdx.x % 8) * (nwg / 8) + (int)blockIdx.x / 8 : (int)blockIdx.x;
#else
  const int lw = (int)blockIdx.x;
#endif
  const int G = __builtin_amdgcn_readfirstlane(lw * 2 + g);
  struct Loc { int ray, tb; bool ok; };
  auto locate = [&](int pl, int b) {
    const int sidx = pl * NB + b;
    const int k = (int)(((uint64_t)(uint32_t)sidx * a.nb_magic) >> 32);  // sidx / nb (exact: sidx * nb < 2^32), scalar ALU
    Loc L;
    L.tb = sidx - k * a.nb;
    const int64_t r = G + (int64_t)k * a.nG;
    L.ok = r < a.R;
    L.ray = L.ok ? (int)r : (int)a.R - 1;
    if (!L.ok) L.tb = a.nb - 1;
    return L;
  };
  auto geo_setup = [&](int pl) {
    // scalar (SMEM) loads: the addresses are wave-uniform and rays / elaz are read-only for the whole launch.  All loads
    // and their wait sit in ONE asm statement, so the compiler can neither read nor spill a destination in flight.
    // (ox oy) (oz dx) (dy dz) (elev azim): 8-byte loads (a ray is 24 bytes), each into a 64-bit scalar -- vector-typed
    // SGPR asm outputs are mis-split by the compiler (element 1 read from element 0's register)
    uint64_t ra[NB], rb[NB], rc[NB], e2[NB];
    const float* ry[NB];
    const float* ea[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const Loc L = locate(pl, b);
      const int ray = __builtin_amdgcn_readfirstlane(L.ray);
      ry[b] = a.rays + (int64_t)ray * 6;
      ea[b] = a.elaz + (int64_t)ray * 2;
      geo_t0[b] = __builtin_amdgcn_readfirstlane(L.tb) * 32;
      geo_ray[b] = ray;
    }
#define NA_LS_GEO_LOAD(oa, ob, oc, oe, pr, pe)                                                              \
  "s_load_dwordx2 " oa ", " pr ", 0x0\n\ts_load_dwordx2 " ob ", " pr ", 0x8\n\ts_load_dwordx2 " oc ", " pr ", 0x10\n\t" \
  "s_load_dwordx2 " oe ", " pe ", 0x0\n\t"
    if constexpr (NB == 4)
      asm volatile(NA_LS_GEO_LOAD("%0", "%1", "%2", "%3", "%16", "%17") NA_LS_GEO_LOAD("%4", "%5", "%6", "%7", "%18", "%19")
                   NA_LS_GEO_LOAD("%8", "%9", "%10", "%11", "%20", "%21") NA_LS_GEO_LOAD("%12", "%13", "%14", "%15", "%22", "%23")
                   "s_waitcnt lgkmcnt(0)"
                   : "=&s"(ra[0]), "=&s"(rb[0]), "=&s"(rc[0]), "=&s"(e2[0]), "=&s"(ra[1]), "=&s"(rb[1]), "=&s"(rc[1]), "=&s"(e2[1]),
                     "=&s"(ra[2]), "=&s"(rb[2]), "=&s"(rc[2]), "=&s"(e2[2]), "=&s"(ra[3]), "=&s"(rb[3]), "=&s"(rc[3]), "=&s"(e2[3])
                   : "s"(ry[0]), "s"(ea[0]), "s"(ry[1]), "s"(ea[1]), "s"(ry[2]), "s"(ea[2]), "s"(ry[3]), "s"(ea[3]));
    else
      asm volatile(NA_LS_GEO_LOAD("%0", "%1", "%2", "%3", "%8", "%9") NA_LS_GEO_LOAD("%4", "%5", "%6", "%7", "%10", "%11")
                   "s_waitcnt lgkmcnt(0)"
                   : "=&s"(ra[0]), "=&s"(rb[0]), "=&s"(rc[0]), "=&s"(e2[0]), "=&s"(ra[1 % NB]), "=&s"(rb[1 % NB]), "=&s"(rc[1 % NB]),
                     "=&s"(e2[1 % NB])
                   : "s"(ry[0]), "s"(ea[0]), "s"(ry[1 % NB]), "s"(ea[1 % NB]));
#undef NA_LS_GEO_LOAD
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      auto lo = [](uint64_t v) { return __builtin_bit_cast(float, (uint32_t)v); };
      auto hi32 = [](uint64_t v) { return __builtin_bit_cast(float, (uint32_t)(v >> 32)); };
      geo_u[b][0] = lo(ra[b]); geo_u[b][1] = hi32(ra[b]);
      geo_u[b][2] = lo(rb[b]); geo_u[b][3] = hi32(rb[b]);
      geo_u[b][4] = lo(rc[b]); geo_u[b][5] = hi32(rc[b]);
      geo_u[b][6] = lo(e2[b]); geo_u[b][7] = hi32(e2[b]);
    }
  };
  // the ray of this wave's own block of pass `pl`: three scalar loads at the top of EP (short-lived SGPRs; the group-wide
  // table above is filled later, in the epilogue of first.out, for the two View phases -- holding it for the whole pass
  // made the compiler park it in scratch memory and re-store it every pass)
  auto own_setup = [&](int pl) {
    const Loc L = locate(pl, blk);
    const float* ry = a.rays + (int64_t)__builtin_amdgcn_readfirstlane(L.ray) * 6;
    uint64_t ra, rb, rc;
    asm volatile("s_load_dwordx2 %0, %3, 0x0\n\ts_load_dwordx2 %1, %3, 0x8\n\ts_load_dwordx2 %2, %3, 0x10\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(ra), "=&s"(rb), "=&s"(rc) : "s"(ry));
    auto lo = [](uint64_t v) { return __builtin_bit_cast(float, (uint32_t)v); };
    auto hi32 = [](uint64_t v) { return __builtin_bit_cast(float, (uint32_t)(v >> 32)); };
    own_u[0] = lo(ra); own_u[1] = hi32(ra); own_u[2] = lo(rb); own_u[3] = hi32(rb); own_u[4] = lo(rc); own_u[5] = hi32(rc);
    const float dx = own_u[3], dy = own_u[4], dz = own_u[5];
    own_dn = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sqrtf((dx * dx + dy * dy) + dz * dz))));
  };
  struct GeoRaw { float x, y, z; };  // t (x) or the explicit position of this lane's sample
  auto geo_load = [&](int b) -> GeoRaw {
    const int t = geo_t0[b] + ln;
    const int tc = t < a.T ? t : a.T - 1;
    GeoRaw r;
    if (a.pts != nullptr) {
      const float* p = a.pts + ((int64_t)tc * a.R + geo_ray[b]) * 3;
      r.x = p[0]; r.y = p[1]; r.z = p[2];
    } else {
      r.x = a.ts[(int64_t)geo_ray[b] * a.ts_stride + tc]; r.y = r.z = 0.f;
    }
    return r;
  };
  auto geo_make = [&](int b, const GeoRaw& r, bool act) -> Frag<PREC> {
    float px = r.x, py = r.y, pz = r.z;
    if (a.pts == nullptr) {
      const float tt = r.x;
      px = geo_u[b][0] + tt * geo_u[b][3]; py = geo_u[b][1] + tt * geo_u[b][4]; pz = geo_u[b][2] + tt * geo_u[b][5];
    }
    float v4[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v4[e] = 0.f;
    if (hi == 0) { v4[0] = px; v4[1] = py; v4[2] = pz; v4[3] = geo_u[b][6]; v4[4] = geo_u[b][7]; }
    Frag<PREC> f = make_frag<PREC>(v4);
    if (act) frag_activate<PREC, NA_ACT_SIN>(f);
    return f;
  };
  struct Geom {
    int64_t ray;
    bool item_ok, t_ok;
    int t;
    float px, py, pz, dist, dx, dy, dz;
  };
  // geometry of this lane's sample of the wave's own block of the CURRENT pass (own_setup(pass) has run)
  // ts[t] and ts[t + 1] of this lane's step of block `blk` of pass pl: requested first thing in EP, before the scalar loads
  // of the ray (whose wait would otherwise sit in front of them)
  struct TsPair { float t0, t1; };
  auto ts_load = [&](int pl) {
    const Loc L = locate(pl, blk);
    const int t = L.tb * 32 + ln;
    const int tc = t < a.T ? t : a.T - 1;
    TsPair r;
    const float* tsr = a.ts + (int64_t)L.ray * a.ts_stride;
    r.t0 = tsr[tc];
    r.t1 = tsr[tc < a.T - 1 ? tc + 1 : tc];
    return r;
  };
  auto geom = [&](int pass, int b, const TsPair& tp) {
    Geom q;
    const Loc L = locate(pass, b);
    q.item_ok = L.ok;
    q.ray = L.ray;
    const int tb = L.tb;
    q.t = tb * 32 + ln;
    q.t_ok = q.t < a.T;
    const int tc = q.t_ok ? q.t : a.T - 1;
    const float (&u)[6] = own_u;
    q.dx = u[3]; q.dy = u[4]; q.dz = u[5];
    const float tt = tp.t0;
    if (a.pts != nullptr) {
      const float* p = a.pts + ((int64_t)tc * a.R + q.ray) * 3;
      q.px = p[0]; q.py = p[1]; q.pz = p[2];
    } else {
      q.px = u[0] + tt * q.dx; q.py = u[1] + tt * q.dy; q.pz = u[2] + tt * q.dz;
    }
    const float d = tc < a.T - 1 ? fmaxf(tp.t1 - tt, 1e-5f) : 1e10f;
    q.dist = d * own_dn;
    return q;
  };
  // what the compositing of block `blk` of the PREVIOUS pass needs: issued at the top of EP next to the loads above
  struct Prev { int64_t ray; int t; bool ok, t_ok; float dist; };
  auto prev_geom = [&](int pl, const TsPair& tp) {
    Prev q;
    const Loc L = locate(pl, blk);
    q.ok = L.ok;
    q.ray = L.ray;
    q.t = L.tb * 32 + ln;
    q.t_ok = q.t < a.T;
    const int tc = q.t_ok ? q.t : a.T - 1;
    const float tt = tp.t0;
    const float d = tc < a.T - 1 ? fmaxf(tp.t1 - tt, 1e-5f) : 1e10f;
    q.dist = d * prev_dn;
    return q;
  };

  float w_local = 0.f;  // block-local weight of this lane's sample, until `combine` knows the transmittance in front
  // Where block b's compositing partials (P, S_rgb, W_head) wait for `combine`.  MODEL 0 composites a block at the END of
  // its own pass, in the view.out phase (round 4), while other waves may still read the hidden region: the slots are the head
  // of the block's fourth init chunk (the last latent chunk: dead once view.L0 has run, not written by EP, rewritten by the
  // epilogue of first.out).  The other schedules composite in the next pass's exposed phase, into the idle hidden region.
  auto part_of = [&](int b) -> float* {
    if constexpr (M0 && PREC == NA_PREC_F16X) return (float*)(ib + b * x::KQ + 3 * 1024);  // (f16 chunk 3 of the block's init group)
    return M0 ? (float*)(ib + (b * 4 + 3) * FR) : (float*)hb + b * kPartialFloats;
  };
  // compositing of block rg of pass `pass` (src/nerf.py:22-27,60-80); the hi=0 half holds the samples
  auto composite = [&](const Prev& q, const f32x16& oc, float density) {
    const float cr = fast_sigmoid_kind(oc[0], a.sigmoid_kind);
    const float cg = fast_sigmoid_kind(oc[1], a.sigmoid_kind);
    const float cb = fast_sigmoid_kind(oc[2], a.sigmoid_kind);
    // (MODEL 2: `density` is VolSDF's Laplace density, used as it is: src/nerf.py:1004-1006, softplus = False)
    const float sigma = (MODEL == 2 || MODEL == 3) ? fmaxf(density, 0.f) : fast_softplus(density - 1.0f);
    const float alpha = q.t_ok ? 1.0f - fast_exp(-sigma * q.dist) : 0.f;
    const float f = (1.0f - alpha) + 1e-10f;
    // exclusive product scan over the 32 steps of the block: shift by one lane (lane 0 of each half: 1), then scan
    float fs = NA_DPP(1.0f, f, 0x138, 0xF);  // wave_shr:1
    if (ln == 0) fs = 1.0f;
    const float excl = scan32_mul(fs);
    const float w = alpha * excl;
    const float P = excl * f;                 // lane 31: product of the whole block
    const float sr = scan32_add(w * cr), sg = scan32_add(w * cg), sb = scan32_add(w * cb);
    const float wh = scan32_add((q.t < a.T - 1) ? w : 0.f);
    if (owner && hi == 0) {
      if (ln == 31) {
        // block product and block-local sums -> the group's (idle) hidden region; combined by `combine` after the barrier
        float* o = part_of(blk);
        o[0] = P; o[1] = sr; o[2] = sg; o[3] = sb; o[4] = wh;
      }
      if (q.ok && q.t_ok && a.alpha != nullptr) a.alpha[(int64_t)q.t * a.R + q.ray] = alpha;
      // MODEL 0 composites at the end of a pass and combines in the next pass's exposed phase: the block-local weight waits in
      // the caller's weights array (scaled in place by `combine`) instead of in a register across the hash gathers
      if (M0 && q.ok && q.t_ok && a.weights != nullptr) a.weights[(int64_t)q.t * a.R + q.ray] = w;
    }
    if constexpr (!M0) w_local = w;
  };
  // Cross-block step of the compositing (the reference's cumprod runs over all T steps: src/nerf.py:22-27): every wave of
  // the group walks the NB blocks of pass `pl` in step order with the running transmittance / colour of the current ray
  // (uniform values, carried from pass to pass), scales its own block's weights by the transmittance in front of it and
  // wave 0 stores a ray's colour + background (src/nerf.py:96-98) after its last block.
  float cT = 1.f, cr0 = 0.f, cr1 = 0.f, cr2 = 0.f, cwh = 0.f;
  auto uni = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
  auto combine = [&](int pl) {
    float mine = 1.f;
    typedef __attribute__((ext_vector_type(4))) float f4;
    f4 pv[NB];
    float pw[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {  // all the group's partials in one batch of LDS reads (uniform addresses)
      pv[b] = *(const f4*)part_of(b);
      pw[b] = part_of(b)[4];
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const Loc L = locate(pl, b);
      if (!L.ok) continue;
      const float P = pv[b][0], s0 = pv[b][1], s1 = pv[b][2], s2 = pv[b][3], swh = pw[b];
      if (L.tb == 0) { cT = 1.f; cr0 = cr1 = cr2 = cwh = 0.f; }
      if (b == blk) mine = cT;
      cr0 = cr0 + cT * s0; cr1 = cr1 + cT * s1; cr2 = cr2 + cT * s2; cwh = cwh + cT * swh;
      cT = cT * P;
      if (L.tb == a.nb - 1 && rg == 0 && lane == 0) {
        const float sky = a.bg_kind == NA_BG_WHITE ? 1.0f - cwh : 0.f;
        float* o = a.out + (int64_t)L.ray * 3;
        o[0] = cr0 + sky; o[1] = cr1 + sky; o[2] = cr2 + sky;
      }
    }
    cT = uni(cT); cr0 = uni(cr0); cr1 = uni(cr1); cr2 = uni(cr2); cwh = uni(cwh);
    if (a.weights != nullptr && owner && hi == 0) {
      const Loc L = locate(pl, blk);
      const int t = L.tb * 32 + ln;
      if (L.ok && t < a.T) {
        float* wp = a.weights + (int64_t)t * a.R + L.ray;
        *wp = (M0 ? *wp : w_local) * mine;
      }
    }
  };

#if NA_LS_TRACE
  unsigned long long* tlog = (a.trace != nullptr && blockIdx.x == 0 && lane == 0 && rg == 0) ? a.trace + g * 128 : nullptr;
  int tpos = 0;
  bool ton = false;
#define SYNC()                                                                              \
  do {                                                                                      \
    if (tlog != nullptr && ton && tpos < 126) tlog[tpos++] = __builtin_amdgcn_s_memtime();  \
    __syncthreads();                                                                        \
    if (tlog != nullptr && ton && tpos < 126) tlog[tpos++] = __builtin_amdgcn_s_memtime();  \
  } while (0)
#define STAMP(i)                                                                              \
  do {                                                                                         \
    if (tlog != nullptr && ton) a.trace[256 + g * 16 + (i)] = __builtin_amdgcn_s_memtime();   \
  } while (0)
#else
#define SYNC() __syncthreads()
#define STAMP(i) \
  do {           \
  } while (0)
#endif
  f32x16 acc[2][NB];
  // One hash level (4*hi + k) of a sample -> bytes 8*(k&1)..+7 of this lane's 16 B of init chunk k>>1 (LDS).
  auto hash_finish = [&](int k, const HashGather& hg) {
    float f[4];
    hash_level_finish(hg, f);
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
    bf16x4 h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      h[e] = to_elem<PREC>(f[e]);
      l[e] = to_elem<PREC, false>(f[e] - from_elem<PREC>(h[e]));  // (two-plane precisions only)
    }
    char* dst = ib + (blk * 4 + (k >> 1)) * FR + lane * 16 + (k & 1) * 8;
    *(bf16x4*)dst = h;
    if constexpr (kTwoPlane<PREC>) *(bf16x4*)(dst + 1024) = l;
  };
  Frag<PREC> ring[kPF][2];
  x::Regs XR;  // (NA_PREC_F16X only)
  f32x16 bvx[2];
  // scalar bases of this row group's pair and record streams (F16X)
  const int xpair = kHeaderBytes + kBiasBytes + rg * x::stream_rg(MODEL);
  const int xrec = xpair + x::npair(MODEL) * x::PAIRB;
  constexpr int XNR = x::nrec(MODEL);
  if constexpr (PREC == NA_PREC_F16X) {
#pragma unroll
    for (int c = 0; c < 4; ++c) { x::a16set(XR.a16[0], c, x::wload16(wrs, lane, xrec, 0, c)); x::a16set(XR.a16[1], c, x::wload16(wrs, lane, xrec, 1, c)); }
    XR.a6 = x::wload6(wrs, lane, xrec);
    XR.asc[0] = x::wloadsc(wrs, lane, xrec);
    XR.asc[1] = 0;
  } else {
#pragma unroll
    for (int p = 0; p < kPF; ++p) {
      ring[p][0] = wload<PREC>(wrs, wvoff, p * C::PAIR);
      ring[p][1] = wload<PREC>(wrs, wvoff, p * C::PAIR + FR);
    }
  }

  f32x16 oc[1];
  float density = 0.f;
  int prev = -1;
  // Group 1 runs ONE phase behind group 0.  (A larger odd lag would put the ~10 k-cycle EP of either group opposite a full
  // hidden-layer MFMA phase of the other instead of its 1-k / 2.4-k-cycle view.out / first.init; measured with
  // -DNA_LS_LAG_OVERRIDE = 3 ... 11: the frame time is the same to 0.2 %.)
  constexpr int LAG = NA_LS_LAG_OVERRIDE > 0 ? NA_LS_LAG_OVERRIDE : 1;
  if (g == 1) {
#pragma unroll 1
    for (int i = 0; i < LAG; ++i) __syncthreads();
  }

  if constexpr (MODEL == 1 || MODEL == 3) {
    // the zero chunk behind (x, y, z): its weights are zero, its LDS words only have to be finite
    if (owner) {
      float z8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) z8[e] = 0.f;
      fwrite<PREC>(ib + (blk * 4 + 1) * FR + lane * 16, make_frag<PREC>(z8));
    }
  }

  // (NA_PREC_F16X) the bias of the NEXT phase waits in bvx and becomes the C operand of that phase's first MFMAs
  auto xbias = [&](int ph) {
#pragma unroll
    for (int t = 0; t < 2; ++t) bvx[t] = bias_tile(wrs, bias_rg + ph * 1024, t, lane);
  };
  // (MODEL 0) the end of a pass, inside its view.out phase: compositing of the wave's own block of THIS pass (its ts are
  // re-read: two loads that return under the MFMAs' tail), then everything the next pass's exposed phase would wait for --
  // its ts pair and the scalar loads of its ray
  TsPair tnext = {0.f, 0.f};
  auto pass_tail = [&](int pl) {
    if (NB == 4 || owner) {
      const TsPair tc = ts_load(pl);
      tnext = ts_load(pl + 1);
      prev_dn = own_dn;
      if constexpr (PREC == NA_PREC_F16X) density = *(const float*)(ib + blk * x::KQ + 6144 + 1024 + ln * 16 + 12);
      composite(prev_geom(pl, tc), oc[0], density);
      __builtin_amdgcn_sched_barrier(0);
      own_setup(pl + 1);
    } else {
      tnext = ts_load(pl + 1);
      own_setup(pl + 1);
    }
  };
  // ---- MODEL 6 (mip, NA_PREC_F16X): the 96 IPE features of a sample (src/utils.py:23-27, 83-140; hook src/nerf.py:256-261) as two
  // K64 groups in the hidden format.  Row group rg generates group g = rg & 1 of block rg >> 1: a lane (sample, k half h) computes
  // its 12 (degree, axis) pairs pidx = 24 g + 12 h + j -- sine and cosine feature of a pair from ONE reduced angle and one damping
  // factor, into slots 2 j and 2 j + 1 (three live chunks per group: the four generating waves of a sample group do equal work).
  // Block b's groups live at K64 groups 2 b, 2 b + 1 of block 0's hidden space (block 1's holds the raw [hash | x] / latent values that wait for the skip layer).
  float rad_u[NB];  // pixel radius of block b's ray (uniform)
  auto mip_setup = [&]() {
    if constexpr (MIP) {
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int r = geo_ray[b];
        const int W = a.mip_W, H = a.mip_H;
        const int row = r / W, wq = r - row * W, bi = row / H, hq = row - bi * H;
        rad_u[b] = uni(mip_radius(a.rays, H, W, bi, hq, wq));
      }
    }
  };
  auto gen_ipe = [&](auto act_tag) {
    if constexpr (MIP && PREC == NA_PREC_F16X && !(NA_LS_MIP_ABLATE & 1)) {
      constexpr int ACT = decltype(act_tag)::value;
      const int g = rg & 1, b = rg >> 1;  // this wave's unit (NB = 2)
      float ry[6], rad = 0.f;
      int t0i = 0;
#pragma unroll
      for (int bb = 0; bb < NB; ++bb)
        if (bb == b) {
#pragma unroll
          for (int e = 0; e < 6; ++e) ry[e] = geo_u[bb][e];
          rad = rad_u[bb];
          t0i = geo_t0[bb];
        }
      const int t = t0i + ln;
      const int tc = t < a.T ? t : a.T - 1;
      const float t0 = a.ts[tc];
      const float t1 = tc < a.T - 1 ? a.ts[tc + 1] : mip_last_edge(a.ts, a.T, a.mip_t_end);
      const MipGauss gs = mip_gaussian(ry, rad, t0, t1, a.mip_kind);
      // mip_feature's arithmetic (common.h) with the powers of two pulled out of the products -- bit-identical: scaling by
      // 2^deg commutes with every rounding here.  Per axis: the revolution count of the mean at degree 0 as a (rounded product,
      // recovered error) pair and the damping exponent; per (degree, axis) pair four v_ldexp, one reduction, two v_sin, one v_exp.
      float mm[3] = {gs.m0, gs.m1, gs.m2}, pr0[3], er0[3], ck[3];
      const float cc[3] = {gs.c0, gs.c1, gs.c2};
#pragma unroll
      for (int ax = 0; ax < 3; ++ax) {
        pr0[ax] = mm[ax] * 0.15915494309189535f;
        er0[ax] = fmaf(mm[ax], 0.15915494309189535f, -pr0[ax]) + mm[ax] * 6.4206383e-9f;
        ck[ax] = cc[ax] * -0.7213475204444817f;
      }
      int min_deg = a.mip_min_deg;
      asm volatile("" : "+s"(min_deg));  // (not loop-invariant for the optimiser: 90 hoisted per-pair constants were spilled)
      f32x16 n0, n1;
#pragma unroll
      for (int e = 0; e < 16; ++e) { n0[e] = 0.f; n1[e] = 0.f; }
      auto pairs = [&](auto g_tag) {
        constexpr int G = decltype(g_tag)::value;
#pragma unroll
        for (int j = 0; j < 12; ++j) {
          // pair index of lane half 0 | 1 (compile-time: no per-lane division, one select per operand)
          const int pl = 24 * G + j, ph = 24 * G + 12 + j;
          const int kl = pl / 3, al = pl - 3 * kl, kh = ph / 3, ah = ph - 3 * kh;
          const float m = hi ? mm[ah] : mm[al], p0 = hi ? pr0[ah] : pr0[al], e0 = hi ? er0[ah] : er0[al];
          const float cK = hi ? ck[ah] : ck[al];
          const int deg = min_deg + (hi ? kh : kl);
          const float pr = ldexpf(p0, deg);
          const float rev = (pr - rintf(pr)) + ldexpf(e0, deg);
          const float y = ldexpf(m, deg);
          const float yc = y + 1.5707963267948966f;  // the cosine half is sin(fl(y + pi/2)): the rounded sum differs from y by an
          const float delta = yc - y;                // exactly representable delta
          const float damp = __builtin_amdgcn_exp2f(ldexpf(cK, 2 * deg));
          const float sn = damp * __builtin_amdgcn_sinf(rev);
          const float cs = damp * __builtin_amdgcn_sinf(fmaf(delta, 0.15915494309189535f, rev));
          if (j < 8) { n0[2 * j] = sn; n0[2 * j + 1] = cs; } else { n1[2 * (j - 8)] = sn; n1[2 * (j - 8) + 1] = cs; }
        }
      };
      if (g == 0) pairs(std::integral_constant<int, 0>{}); else pairs(std::integral_constant<int, 1>{});
      x::store_block<ACT>(hb + (2 * b + g) * x::KQ, n0, n1, lane, a.sat_gen);
    }
  };
  // ---- NA_PREC_F16X, schedules whose first MLP takes the hash encoder (MODEL 0, 4): the [hash | x] group
  auto hash_group_ep = [&](int pass) {
    if constexpr (PREC == NA_PREC_F16X) {
      // f16x (round 4): [hash | x] is ONE K64 group of the init region in the hidden format (f16 fragments | R | T), so the
      // lane that converts must hold all 32 values of an MFMA lane (sample, k half h): h = 0 (levels 0..3 + x, y, z twice)
      // comes from the block's owner wave, h = 1 (levels 4..7) from the helper wave rg + 2.  All 64 lanes gather -- lane
      // (sample, j) the levels 4 h + 2 j, 4 h + 2 j + 1 -- then the j = 1 half hands its eight features to the j = 0 half
      // (ds_bpermute: no memory), which converts and stores for MFMA lane (sample, h).  The raw values wait in the idle hidden
      // region for the skip connection (E1 re-enters them through the activation).
      const int part = owner ? 0 : 1;
      const Geom q = geom(pass, blk, tnext);
      float f8[8];
      HashGather hg;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int lvl0 = 4 * part + k;                      // (j = 0) | + 2 (j = 1)
        hash_level_issue(q.px, q.py, q.pz, a.tables, hi ? a.res.n[lvl0 + 2] : a.res.n[lvl0], lvl0 + 2 * hi, hg);
        float f[4];
        hash_level_finish(hg, f);
#pragma unroll
        for (int e = 0; e < 4; ++e) f8[4 * k + e] = f[e];
        __builtin_amdgcn_sched_barrier(0);
        STAMP(3 + k);
      }
      f32x16 n0, n1;  // the MFMA lane's 32 values in slot order: chunk 0 = n0[0..7], 1 = n0[8..15], 2 = n1[0..7], 3 = n1[8..15]
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        n0[e] = f8[e];
        n0[8 + e] = __shfl_down(f8[e], 32, 64);
        n1[e] = 0.f; n1[8 + e] = 0.f;
      }
      if (part == 0) { n1[0] = q.px; n1[1] = q.py; n1[2] = q.pz; n1[3] = q.px; n1[4] = q.py; n1[5] = q.pz; }
      if (hi == 0) {
        const int ml = ln + 32 * part;  // the MFMA lane these values belong to
        char* st = hb + x::BLKH + rg * x::KQ + ml * 16;   // raw values: this wave's K64 region of block 1 (idle until E1 stores it LAST)
#pragma unroll
        for (int c = 0; c < 4; ++c) *(f32x4*)(st + c * 1024) = f32x4{n0[4 * c], n0[4 * c + 1], n0[4 * c + 2], n0[4 * c + 3]};
        *(f32x4*)(st + 4096) = f32x4{n1[0], n1[1], n1[2], n1[3]};
        *(f32x4*)(st + 5120) = f32x4{n1[4], n1[5], n1[6], n1[7]};
        x::store_block<NA_ACT_NONE, 3>(ib + blk * x::KQ, n0, n1, ml, a.sat_gen);
      }
      STAMP(7);
    }
  };
      // raw values of this wave's half of an init group (written by EP / E6 into the wave's own K64 region of the hidden
      // space): read back before store_acts overwrites the region, re-entered through the activation (src/neural_blocks.py:291-293)
  auto reenter_hash = [&]() {
    if constexpr (PREC == NA_PREC_F16X) {
        if (hi == 0) {
          const int part = owner ? 0 : 1, ml = ln + 32 * part;
          const char* st = hb + x::BLKH + rg * x::KQ + ml * 16;
          f32x16 n0, n1;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const f32x4 v = *(const f32x4*)(st + c * 1024);
            n0[4 * c] = v[0]; n0[4 * c + 1] = v[1]; n0[4 * c + 2] = v[2]; n0[4 * c + 3] = v[3];
          }
          const f32x4 u = *(const f32x4*)(st + 4096), w = *(const f32x4*)(st + 5120);
#pragma unroll
          for (int e = 0; e < 16; ++e) n1[e] = 0.f;
          n1[0] = u[0]; n1[1] = u[1]; n1[2] = u[2]; n1[3] = u[3]; n1[4] = w[0]; n1[5] = w[1]; n1[6] = w[2]; n1[7] = w[3];
          x::store_block<NA_ACT_LEAKY_RELU, 3>(ib + blk * x::KQ, n0, n1, ml, a.sat_gen);
        }
    }
  };
  if constexpr (M0 || MODEL == 4) {
    tnext = ts_load(0);
    own_setup(0);
  }
  for (int pass = 0; pass < a.npg; ++pass) {
    int cur = 0;
#if NA_LS_TRACE
    ton = pass == 1;
#endif
    if constexpr (MODEL == 5) {
      // ================= a Fourier-encoded SkipConnMLP alone, rows to HBM (NA_PREC_F16X; VolSDF's MLP SDF network).  The 256
      // Fourier features never exist outside LDS: row group rg GENERATES the K64 group rg of every block of its sample group
      // (16 frequencies per lane: three FMAs on the position, hardware sine / cosine on a two-constant reduction) straight into
      // the hidden format, in a VALU phase in front of each Linear that consumes them -- init: [features | p]; the skip layers
      // L0, L3: K = 256 hidden first, then (accumulators kept) the regenerated features through the activation + p.
      if constexpr (PREC == NA_PREC_F16X) {
        static_assert(NB == 2, "MODEL 5: two blocks per group");
        const float* basis = (const float*)a.tables;  // [3][128] (frequencies contiguous), extra_scale folded in by the host
        constexpr int F = 128;
        geo_setup(pass);
        auto gen = [&](auto act_tag) {
          constexpr int ACT = decltype(act_tag)::value;
          const int f0 = 32 * rg + 16 * hi;
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            const GeoRaw r = geo_load(b);
            float px = r.x, py = r.y, pz = r.z;
            if (a.pts == nullptr) { px = geo_u[b][0] + r.x * geo_u[b][3]; py = geo_u[b][1] + r.x * geo_u[b][4]; pz = geo_u[b][2] + r.x * geo_u[b][5]; }
            f32x16 n0, n1;
            // eight frequencies at a time (24 basis registers live, not 48: in the skip layers the 64 accumulator registers of the
            // K = 256 part stay live across this phase -- with all of a lane's basis rows fetched at once 118 registers spilled)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
              f32x4 bq[3][2];
#pragma unroll
              for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int k = 0; k < 2; ++k) bq[q][k] = *(const f32x4*)(basis + q * F + f0 + 8 * hf + 4 * k);
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                // mapped = x @ basis in the generic kernel's order (one product, two fmas); sin / cos of it
                float m = px * bq[0][j >> 2][j & 3];
                m = fmaf(py, bq[1][j >> 2][j & 3], m);
                m = fmaf(pz, bq[2][j >> 2][j & 3], m);
                const float qr = rintf(m * 0.15915493667125702f);
                float rv = fmaf(m, 0.15915493667125702f, -qr);
                rv = fmaf(m, 6.4206382432985265e-09f, rv);
                const float sn = __builtin_amdgcn_sinf(rv), cs = __builtin_amdgcn_cosf(rv);
                if (hf == 0) { n0[2 * j] = sn; n0[2 * j + 1] = cs; } else { n1[2 * j] = sn; n1[2 * j + 1] = cs; }
              }
              __builtin_amdgcn_sched_barrier(0);
            }
            x::store_block<ACT>(hb + b * x::BLKH + rg * x::KQ, n0, n1, lane, a.sat_gen);
            __builtin_amdgcn_sched_barrier(0);
          }
        };
        auto p_make = [&](int b, const GeoRaw& r, bool act) -> Frag<PREC> {
          float px = r.x, py = r.y, pz = r.z;
          if (a.pts == nullptr) { px = geo_u[b][0] + r.x * geo_u[b][3]; py = geo_u[b][1] + r.x * geo_u[b][4]; pz = geo_u[b][2] + r.x * geo_u[b][5]; }
          float v4[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v4[e] = 0.f;
          if (hi == 0) { v4[0] = px; v4[1] = py; v4[2] = pz; }
          Frag<PREC> f = make_frag<PREC>(v4);
          if (act) frag_activate<PREC, NA_ACT_LEAKY_RELU>(f);
          return f;
        };
        typedef std::integral_constant<int, NA_ACT_NONE> RawT;
        typedef std::integral_constant<int, NA_ACT_LEAKY_RELU> LeakyT;
        // ---- EP: the raw features; init = [features | p]
        gen(RawT{});
        xbias(0);
        x::pairs_prefetch(XR, wrs, xpair, lane, 0, 1);
        SYNC();
        {
          GeoRaw graw[NB];
#pragma unroll
          for (int b = 0; b < NB; ++b) graw[b] = geo_load(b);
          x::recs<2, NB, true, XNR>(acc, bvx, XR, wrs, xrec, 0, hb, lane);
          x::geo_pair<0, NB>(acc, XR, wrs, xpair, lane, graw, p_make, false);
        }
        SYNC();
        // ---- L0 (skip), L1, L2, L3 (skip), L4, L5
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
          const int r0 = half == 0 ? 4 : 20;  // first record of the skip layer
          {
            x::store_acts<NA_ACT_LEAKY_RELU, NB>(acc, hb, rg, lane, a.sat_gen);
            xbias(half == 0 ? 1 : 4);
          }
          SYNC();
          x::recs<2, NB, true, XNR>(acc, bvx, XR, wrs, xrec, r0, hb, lane);                       // K = 256 hidden
          SYNC();
          {
            gen(LeakyT{});                                                                          // the features again, activated
            XR.pr[0] = x::wpair(wrs, lane, xpair, 1 + half);  // (one ring slot for the three position pairs: one code path)
          }
          SYNC();
          {
            GeoRaw graw[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) graw[b] = geo_load(b);
            x::recs<2, NB, false, XNR>(acc, bvx, XR, wrs, xrec, r0 + 4, hb, lane);                  // + K = 256 features
            x::geo_pair<0, NB>(acc, XR, wrs, xpair, lane, graw, p_make, true);                      // + p
          }
          SYNC();
#pragma unroll 1
          for (int i = 0; i < 2; ++i) {
            x::store_acts<NA_ACT_LEAKY_RELU, NB>(acc, hb, rg, lane, a.sat_gen);
            xbias((half == 0 ? 2 : 5) + i);
            SYNC();
            x::recs<2, NB, true, XNR>(acc, bvx, XR, wrs, xrec, r0 + 8 + 4 * i, hb, lane);           // L1, L2 | L4, L5
            SYNC();
          }
        }
        // ---- out: 65 rows, row-major (row group rg: tile min(rg, 2) for the NB blocks); rows to HBM
        f32x16 oq[1][NB];
        f32x16 bo1[1];
        {
          bo1[0] = bias_tile(wrs, bias_rg + 7 * 1024, rg < 2 ? rg : 2, lane);
          x::store_acts<NA_ACT_LEAKY_RELU, NB>(acc, hb, rg, lane, a.sat_gen);
        }
        SYNC();
        x::recs<1, NB, true, XNR>(oq, bo1, XR, wrs, xrec, 36, hb, lane);
        if (rg < 3) {
          typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            const Loc L = locate(pass, b);
            const int t = L.tb * 32 + ln;
            if (L.ok && t < a.T) {
              float* yrow = a.y + ((int64_t)t * a.R + L.ray) * a.y_ld + 32 * rg;
              if (rg < 2) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                  *(f32x4u*)(yrow + 8 * q + 4 * hi) = f32x4u{oq[0][b][4 * q], oq[0][b][4 * q + 1], oq[0][b][4 * q + 2], oq[0][b][4 * q + 3]};
              } else if (hi == 0) {
                yrow[0] = oq[0][b][0];  // row 64
              }
            }
          }
        }
        SYNC();
      }
      prev = pass;
      continue;
    }
    if constexpr (MODEL == 4) {
      // ================= a hash-encoded SkipConnMLP alone, rows to HBM (NA_PREC_F16X; D-NeRF's deformation network): EP = the
      // [hash | x] group; init | L0 (skip group + K = 256) | L1 | L2 | L3 (skip group + K = 256) | L4 | out (one tile, block per
      // wave); 27 records per pass.  No compositing, nothing carried from pass to pass.
      if constexpr (PREC == NA_PREC_F16X) {
        static_assert(NB == 2, "MODEL 4: two blocks per group");
        hash_group_ep(pass);
        {
          f32x16 bv[2];
#pragma unroll
          for (int t = 0; t < 2; ++t) bv[t] = bias_tile(wrs, bias_rg + 0 * 1024, t, lane);
          SYNC();
          bvx[0] = bv[0]; bvx[1] = bv[1];
        }
        x::recs<2, NB, true, XNR, 1, 1, 0, 3>(acc, bvx, XR, wrs, xrec, 0, hb, lane, ib);          // init: record 0
        SYNC();
        {
          x::store_acts<NA_ACT_LEAKY_RELU, NB, 0, 1>(acc, hb, rg, lane, a.sat_gen);
          reenter_hash();  // act(init) stays in the init region for both skip layers (src/neural_blocks.py:291-293)
          x::store_acts<NA_ACT_LEAKY_RELU, NB, 1, 2>(acc, hb, rg, lane, a.sat_gen);
          xbias(1);
        }
        SYNC();
        x::recs<2, NB, true, XNR, 5, 1, 1, 3>(acc, bvx, XR, wrs, xrec, 1, hb, lane, ib);          // L0: records 1..5
        SYNC();
#pragma unroll 1
        for (int i = 0; i < 2; ++i) {
          x::store_acts<NA_ACT_LEAKY_RELU, NB>(acc, hb, rg, lane, a.sat_gen);
          xbias(2 + i);
          SYNC();
          x::recs<2, NB, true, XNR>(acc, bvx, XR, wrs, xrec, 6 + 4 * i, hb, lane);                  // L1, L2: records 6..13
          SYNC();
        }
        {
          x::store_acts<NA_ACT_LEAKY_RELU, NB>(acc, hb, rg, lane, a.sat_gen);
          xbias(4);
        }
        SYNC();
        x::recs<2, NB, true, XNR, 5, 1, 0, 3>(acc, bvx, XR, wrs, xrec, 14, hb, lane, ib);         // L3: records 14..18
        SYNC();
        {
          x::store_acts<NA_ACT_LEAKY_RELU, NB>(acc, hb, rg, lane, a.sat_gen);
          xbias(5);
        }
        SYNC();
        x::recs<2, NB, true, XNR, 4, 0, 1>(acc, bvx, XR, wrs, xrec, 19, hb, lane);                  // L4: records 19..22
        SYNC();
        f32x16 ocx[1][1];
        f32x16 bo1[1];
        {
          bo1[0] = bias_tile(wrs, bias_rg + 6 * 1024, 0, lane);
          x::store_acts<NA_ACT_LEAKY_RELU, NB>(acc, hb, rg, lane, a.sat_gen);
        }
        SYNC();
        x::recs<1, 1, true, XNR, 4, 0, 1, 4, true>(ocx, bo1, XR, wrs, xrec, 23, hb + blk * x::BLKH, lane);  // out: records 23..26
        if (owner) {
          // registers 4 q .. 4 q + 3 of a lane are the rows 8 q + 4 hi .. + 3 of its sample: 16-byte stores where the row allows
          const Loc L = locate(pass, blk);
          const int t = L.tb * 32 + ln;
          if (L.ok && t < a.T) {
            typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
            float* yrow = a.y + ((int64_t)t * a.R + L.ray) * a.y_ld;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int f0 = 8 * q + 4 * hi;
              if (f0 + 4 <= a.n_out) *(f32x4u*)(yrow + f0) = f32x4u{ocx[0][0][4 * q], ocx[0][0][4 * q + 1], ocx[0][0][4 * q + 2], ocx[0][0][4 * q + 3]};
              else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                  if (f0 + i < a.n_out) yrow[f0 + i] = ocx[0][0][4 * q + i];
              }
            }
          }
        }
        tnext = ts_load(pass + 1);
        own_setup(pass + 1);
        SYNC();
      }
      prev = pass;
      continue;
    }
    if constexpr (MODEL == 3) {
      // ================= VolSDF, SIREN SDF network + View head: EP = sample position + compositing of the previous pass
      auto none_l = [](int) { return 0; };
      auto none_m = [](int, int, bool) { return 0; };
      if (NB == 4 || owner) {
        prev_dn = own_dn;
        const TsPair tcur = ts_load(pass);
        TsPair tprev = tcur;
        if (prev >= 0) tprev = ts_load(prev);
        own_setup(pass);
        const Geom q = geom(pass, blk, tcur);
        if (prev >= 0) composite(prev_geom(prev, tprev), oc[0], density);
        float v2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v2[e] = 0.f;
        fwrite<PREC>(ib + (blk * 4 + 1) * FR + lane * 16, make_frag<PREC>(v2));  // (the latent of the last pass sat here)
        if (hi == 0) { v2[0] = q.px; v2[1] = q.py; v2[2] = q.pz; }
        fwrite<PREC>(ib + blk * 4 * FR + lane * 16, make_frag<PREC>(v2));
      }
      {
        f32x16 bv[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) bv[t] = bias_tile(wrs, bias_rg + 0 * 1024, t, lane);
        if constexpr (PREC == NA_PREC_F16X) x::pairs_prefetch(XR, wrs, xpair, lane, 0, 1);
        SYNC();
        if (prev >= 0) combine(prev);
        if constexpr (PREC == NA_PREC_F16X) {
          bvx[0] = bv[0]; bvx[1] = bv[1];
        } else {
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[t][b] = bv[t];
        }
      }
      if constexpr (PREC == NA_PREC_F16X) {
        // ---- NA_PREC_F16X: sdf.init (pair 0), L0 (skip pair 1 + records 0..3), L1, L2, L3 (skip pair 2 + records 12..15), L4,
        // sdf.out (records 20..23, row-major), then the View half (pairs 3..12, records 24..43)
        x::pairs<0, 1, NB>(acc, bvx, XR, wrs, xpair, ib, lane);
        SYNC();
        {
          x::store_acts<NA_ACT_SIN, NB>(acc, hb, rg, lane, a.sat_gen);
          xbias(1);
          if (owner) activate_init<PREC, NA_ACT_SIN, 1>(ib, blk, lane);
          x::pairs_prefetch(XR, wrs, xpair, lane, 1, 1);
        }
        SYNC();
        x::pairs<1, 1, NB>(acc, bvx, XR, wrs, xpair, ib, lane);
        x::recs<2, NB, false, XNR>(acc, bvx, XR, wrs, xrec, 0, hb, lane);                    // L0 (skip)
        SYNC();
#pragma unroll 1
        for (int i = 0; i < 2; ++i) {
          x::store_acts<NA_ACT_SIN, NB>(acc, hb, rg, lane, a.sat_gen);
          xbias(2 + i);
          SYNC();
          x::recs<2, NB, true, XNR>(acc, bvx, XR, wrs, xrec, 4 + 4 * i, hb, lane);             // L1, L2
          SYNC();
        }
        {
          x::store_acts<NA_ACT_SIN, NB>(acc, hb, rg, lane, a.sat_gen);
          xbias(4);
          x::pairs_prefetch(XR, wrs, xpair, lane, 2, 1);
        }
        SYNC();
        x::pairs<2, 1, NB>(acc, bvx, XR, wrs, xpair, ib, lane);
        x::recs<2, NB, false, XNR>(acc, bvx, XR, wrs, xrec, 12, hb, lane);                   // L3 (skip)
        SYNC();
        {
          x::store_acts<NA_ACT_SIN, NB>(acc, hb, rg, lane, a.sat_gen);
          xbias(5);
        }
        SYNC();
        x::recs<2, NB, true, XNR>(acc, bvx, XR, wrs, xrec, 16, hb, lane);                     // L4
        SYNC();
        // sdf.out split by BLOCK like PlainNeRF's first.out (NB = 2): every row group runs two tiles for ONE block, rg & 1 -- row
        // groups 0, 1 the two latent tiles (rows 0..63: the 32 values of an MFMA lane of the View MLP's latent group end up in one
        // lane), row groups 2, 3 the signed-distance tile (row 64) and an all-zero tile.  Then the View half on records: 24 view.init
        // (the latent group) | 25 skip group + 26..29 (view.L0) | 30.. L1..L3 | 42..45 view.out; pairs 3, 4 = the geometry chunk
        static_assert(NB == 2, "sdf.out by block");
        f32x16 ol[2][1];
        {
          x::store_acts<NA_ACT_SIN, NB>(acc, hb, rg, lane, a.sat_gen);
          bvx[0] = bias_tile(wrs, bias_rg + 6 * 1024, rg < 2 ? 0 : 2, lane);
          bvx[1] = bias_tile(wrs, bias_rg + 6 * 1024, rg < 2 ? 1 : 3, lane);  // (slot 3: zeros)
        }
        SYNC();
        x::recs<2, 1, true, XNR>(ol, bvx, XR, wrs, xrec, 20, hb + (rg & 1) * x::BLKH, lane);  // sdf.out
        SYNC();
        {
          xbias(7);
          geo_setup(pass);
          if (rg < 2) {
            // the latent group of block rg: raw rows into the wave's own (idle) K64 region of block 1 for the skip connection, the
            // group itself into the init region
            char* st = hb + x::BLKH + rg * x::KQ + lane * 16;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              *(f32x4*)(st + c * 1024) = f32x4{ol[0][0][4 * c], ol[0][0][4 * c + 1], ol[0][0][4 * c + 2], ol[0][0][4 * c + 3]};
              *(f32x4*)(st + 4096 + c * 1024) = f32x4{ol[1][0][4 * c], ol[1][0][4 * c + 1], ol[1][0][4 * c + 2], ol[1][0][4 * c + 3]};
            }
            x::store_block<NA_ACT_NONE, 4>(ib + rg * x::KQ, ol[0][0], ol[1][0], lane, a.sat_gen);
          } else if (hi == 0) {
            ((float*)hb)[(rg & 1) * 32 + ln] = ol[0][0][0];  // the signed distance of block rg & 1 (lane = step), for its owner
          }
          x::pairs_prefetch(XR, wrs, xpair, lane, 3, 1);
        }
        SYNC();
        if (owner) {  // signed distance -> Laplace density (src/utils.py:50-58, src/nerf.py:1000-1003), composited one pass later
          const float sdfv = ((const float*)hb)[blk * 32 + ln];
          const float sc = a.beta[0];
          const float scaled = (-sdfv) / sc;
          const float cdf = scaled <= 0.f ? fast_exp(fminf(scaled, 0.f)) * 0.5f : 1.f - fast_exp(-fmaxf(scaled, 0.f)) * 0.5f;
          density = (1.0f / sc) * cdf;
        }
        {
          GeoRaw graw[NB];
#pragma unroll
          for (int b = 0; b < NB; ++b) graw[b] = geo_load(b);
          x::recs<2, NB, true, XNR, 1, 1, 0, 4>(acc, bvx, XR, wrs, xrec, 24, hb, lane, ib);     // view.init: latent group + geometry
          x::geo_pair<1, NB>(acc, XR, wrs, xpair, lane, graw, geo_make, false);                  // (pair 3 sits in ring slot 1)
        }
        SYNC();
        {
          x::store_acts<NA_ACT_SIN, NB, 0, 1>(acc, hb, rg, lane, a.sat_gen);
          if (rg < 2) {  // sin(latent) for the skip connection, from the raw rows (before block 1's store overwrites their region)
            const char* st = hb + x::BLKH + rg * x::KQ + lane * 16;
            f32x16 l0, l1;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const f32x4 u = *(const f32x4*)(st + c * 1024), w = *(const f32x4*)(st + 4096 + c * 1024);
              l0[4 * c] = u[0]; l0[4 * c + 1] = u[1]; l0[4 * c + 2] = u[2]; l0[4 * c + 3] = u[3];
              l1[4 * c] = w[0]; l1[4 * c + 1] = w[1]; l1[4 * c + 2] = w[2]; l1[4 * c + 3] = w[3];
            }
            x::store_block<NA_ACT_SIN, 4>(ib + rg * x::KQ, l0, l1, lane, a.sat_gen);
          }
          x::store_acts<NA_ACT_SIN, NB, 1, 2>(acc, hb, rg, lane, a.sat_gen);
          xbias(8);
          XR.pr[0] = x::wpair(wrs, lane, xpair, 4);
        }
        SYNC();
        {
          GeoRaw graw[NB];
#pragma unroll
          for (int b = 0; b < NB; ++b) graw[b] = geo_load(b);
          x::recs<2, NB, true, XNR, 5, 1, 1, 4>(acc, bvx, XR, wrs, xrec, 25, hb, lane, ib);     // view.L0: skip group, K = 256, geometry
          x::geo_pair<0, NB>(acc, XR, wrs, xpair, lane, graw, geo_make, true);
        }
        SYNC();
#pragma unroll 1
        for (int i = 0; i < 3; ++i) {
          x::store_acts<NA_ACT_SIN, NB>(acc, hb, rg, lane, a.sat_gen);
          xbias(9 + i);
          SYNC();
          x::recs<2, NB, true, XNR>(acc, bvx, XR, wrs, xrec, 30 + 4 * i, hb, lane);        // view.L1..L3
          SYNC();
        }
        f32x16 ocx[1][1], bo1[1];
        {
          bo1[0] = bias_tile(wrs, bias_rg + 12 * 1024, 0, lane);
          x::store_acts<NA_ACT_SIN, NB>(acc, hb, rg, lane, a.sat_gen);
        }
        SYNC();
        x::recs<1, 1, true, XNR>(ocx, bo1, XR, wrs, xrec, 42, hb + blk * x::BLKH, lane);   // view.out (block per wave)
        oc[0] = ocx[0][0];
        SYNC();
        prev = pass;
        continue;
      }
      m_hidden<PREC, 0, 2, 0, 0, false, PPP>(acc, ring, cur, wrs, wvoff, hb, ib, lane, none_l, none_m);  // sdf.init
      SYNC();
#define NA_SIN_EPILOGUE(PH, NCHUNK)                                                              \
      {                                                                                            \
        f32x16 bv[2];                                                                              \
        store_acts<PREC, NA_ACT_SIN, 0, 1>(acc, hb, rg, lane);                                     \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        _Pragma("unroll") for (int t = 0; t < 2; ++t) bv[t] = bias_tile(wrs, bias_rg + (PH) * 1024, t, lane); \
        store_acts<PREC, NA_ACT_SIN, 1, 2>(acc, hb, rg, lane);                                     \
        if ((NCHUNK) > 0 && owner) activate_init<PREC, NA_ACT_SIN, ((NCHUNK) > 0 ? (NCHUNK) : 1)>(ib, blk, lane); \
        _Pragma("unroll") for (int t = 0; t < 2; ++t)                                              \
          _Pragma("unroll") for (int b = 0; b < NB; ++b) acc[t][b] = bv[t];                        \
      }                                                                                            \
      SYNC();
      NA_SIN_EPILOGUE(1, 1)
      m_hidden<PREC, 2, 1, 0, 16, false, PPP>(acc, ring, cur, wrs, wvoff, hb, ib, lane, none_l, none_m);  // L0 (skip)
      SYNC();
      NA_SIN_EPILOGUE(2, 0)
      m_hidden<PREC, 3, 0, 0, 16, false, PPP>(acc, ring, cur, wrs, wvoff, hb, ib, lane, none_l, none_m);  // L1
      SYNC();
      NA_SIN_EPILOGUE(3, 0)
      m_hidden<PREC, 3, 0, 0, 16, false, PPP>(acc, ring, cur, wrs, wvoff, hb, ib, lane, none_l, none_m);  // L2
      SYNC();
      NA_SIN_EPILOGUE(4, 0)
      m_hidden<PREC, 3, 1, 0, 16, false, PPP>(acc, ring, cur, wrs, wvoff, hb, ib, lane, none_l, none_m);  // L3 (skip)
      SYNC();
      NA_SIN_EPILOGUE(5, 0)
      m_hidden<PREC, 0, 0, 0, 16, false, PPP>(acc, ring, cur, wrs, wvoff, hb, ib, lane, none_l, none_m);  // L4
      SYNC();
      f32x16 oq[NB];  // sdf.out: this row group's tile (0, 1: latent rows 0..63; 2: the signed distance, row 64) for the NB blocks
      {
        const f32x16 bo = bias_tile(wrs, bias_rg + 6 * 1024, rg < 2 ? rg : 2, lane);
#pragma unroll
        for (int b = 0; b < NB; ++b) oq[b] = bo;
        store_acts<PREC, NA_ACT_SIN>(acc, hb, rg, lane);
      }
      SYNC();
      m_out_rows<PREC, 0, false>(oq, ring, cur, wrs, wvoff, hb, lane);
      SYNC();
      {
        f32x16 bv[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) bv[t] = bias_tile(wrs, bias_rg + 7 * 1024, t, lane);
        geo_setup(pass);
        if (rg < 2) {
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            Frag<PREC> f0, f1;
            acc_to_frags<PREC, NA_ACT_NONE>(oq[b], f0, f1);
            char* dst = ib + (b * 4 + 2 * rg) * FR + lane * 16;
            fwrite<PREC>(dst, f0);
            fwrite<PREC>(dst + FR, f1);
          }
        } else if (rg == 2 && hi == 0) {
#pragma unroll
          for (int b = 0; b < NB; ++b) ((float*)hb)[b * 32 + ln] = oq[b][0];
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int b = 0; b < NB; ++b) acc[t][b] = bv[t];
      }
      SYNC();
      if (owner) {  // signed distance -> Laplace density (src/utils.py:50-58, src/nerf.py:1000-1003), composited one pass later
        const float sdfv = ((const float*)hb)[blk * 32 + ln];
        const float sc = a.beta[0];
        const float scaled = (-sdfv) / sc;
        const float cdf = scaled <= 0.f ? fast_exp(fminf(scaled, 0.f)) * 0.5f : 1.f - fast_exp(-fmaxf(scaled, 0.f)) * 0.5f;
        density = (1.0f / sc) * cdf;
      }
      m_hidden<PREC, 0, 4, 1, 0, false, PPP>(acc, ring, cur, wrs, wvoff, hb, ib, lane, geo_load, geo_make);  // view.init
      SYNC();
      NA_SIN_EPILOGUE(8, 4)
      m_hidden<PREC, 1, 4, 2, 16, false, PPP>(acc, ring, cur, wrs, wvoff, hb, ib, lane, geo_load, geo_make);  // view.L0 (skip)
      SYNC();
      NA_SIN_EPILOGUE(9, 0)
      m_hidden<PREC, 2, 0, 0, 16, false, PPP>(acc, ring, cur, wrs, wvoff, hb, ib, lane, none_l, none_m);
      SYNC();
      NA_SIN_EPILOGUE(10, 0)
      m_hidden<PREC, 2, 0, 0, 16, false, PPP>(acc, ring, cur, wrs, wvoff, hb, ib, lane, none_l, none_m);
      SYNC();
      NA_SIN_EPILOGUE(11, 0)
      m_hidden<PREC, 2, 0, 0, 16, false, PPP>(acc, ring, cur, wrs, wvoff, hb, ib, lane, none_l, none_m);
      SYNC();
#undef NA_SIN_EPILOGUE
      {
        oc[0] = bias_tile(wrs, bias_rg + 12 * 1024, 0, lane);
        store_acts<PREC, NA_ACT_SIN>(acc, hb, rg, lane);
      }
      SYNC();
      m_out<PREC, 2, 1, true, PPP>(oc, ring, cur, wrs, wvoff, hb, lane, blk);
      ring_skip<PREC, 2, 2, true, PPP>(ring, cur, wrs, wvoff);
      SYNC();
      prev = pass;
      continue;
    }
    if constexpr (MODEL == 2) {
      // ================= View head + compositing: EP = compositing of the previous pass + this pass's density / latent rows
      auto none_l = [](int) { return 0; };
      auto none_m = [](int, int, bool) { return 0; };
      if (NB == 4 || owner) {
        prev_dn = own_dn;
        TsPair tprev = {0.f, 0.f};
        if (prev >= 0) tprev = ts_load(prev);
        const Loc L = locate(pass, blk);
        const int t = L.tb * 32 + ln;
        const float* row = a.feat + ((int64_t)(t < a.T ? t : a.T - 1) * a.R + L.ray) * a.feat_ld;
        typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
        const float sdfv = row[0];
        f32x4u lat[4][2];
#pragma unroll
        for (int c = 0; c < 4; ++c) {  // chunk c, slot 8 hi + e <-> latent 16 c + pi_perm(8 hi + e): two runs of four columns
          lat[c][0] = *(const f32x4u*)(row + 1 + 16 * c + 4 * hi);
          lat[c][1] = *(const f32x4u*)(row + 1 + 16 * c + 8 + 4 * hi);
        }
        own_setup(pass);
        if (prev >= 0) composite(prev_geom(prev, tprev), oc[0], density);
        {  // Laplace density of this pass's sample (src/utils.py:50-58, src/nerf.py:1000-1003), composited one pass later
          const float sc = a.beta[0];
          const float scaled = (-sdfv) / sc;
          const float cdf = scaled <= 0.f ? fast_exp(fminf(scaled, 0.f)) * 0.5f : 1.f - fast_exp(-fmaxf(scaled, 0.f)) * 0.5f;
          density = (1.0f / sc) * cdf;
        }
        if constexpr (PREC == NA_PREC_F16X) {
          // the latent is ONE K64 group of the init region in the hidden format (f16 fragments | R | T), like the [hash | x] and
          // latent groups of PlainNeRF: the lane (sample, k half hi) that loaded the 4 x 8 values of its slots converts them
          f32x16 n0, n1;
#pragma unroll
          for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              n0[8 * c + e] = lat[c][0][e]; n0[8 * c + 4 + e] = lat[c][1][e];
              n1[8 * c + e] = lat[2 + c][0][e]; n1[8 * c + 4 + e] = lat[2 + c][1][e];
            }
          x::store_block<NA_ACT_NONE, 4>(ib + blk * x::KQ, n0, n1, lane, a.sat_gen);
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            float v8[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v8[e] = lat[c][0][e]; v8[4 + e] = lat[c][1][e]; }
            fwrite<PREC>(ib + (blk * 4 + c) * FR + lane * 16, make_frag<PREC>(v8));
          }
        }
      }
      geo_setup(pass);
      {
        f32x16 bv[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) bv[t] = bias_tile(wrs, bias_rg + 0 * 1024, t, lane);
        if constexpr (PREC == NA_PREC_F16X) x::pairs_prefetch(XR, wrs, xpair, lane, 0, MODEL == 2 ? 2 : 1);
        SYNC();
        if (prev >= 0) combine(prev);
        if constexpr (PREC == NA_PREC_F16X) {
          bvx[0] = bv[0]; bvx[1] = bv[1];
        } else {
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[t][b] = bv[t];
        }
      }
      if constexpr (PREC == NA_PREC_F16X) {
        // ---- NA_PREC_F16X: the View half -- records 0 view.init (the latent group) | 1 skip group + 2..5 (view.L0) | 6.. L1..L3 |
        // 18..21 view.out; pairs 0, 1 = the 5-wide geometry chunk of init and skip layer
        {
          GeoRaw graw[NB];
#pragma unroll
          for (int b = 0; b < NB; ++b) graw[b] = geo_load(b);
          x::recs<2, NB, true, XNR, 1, 1, 0, 4>(acc, bvx, XR, wrs, xrec, 0, hb, lane, ib);      // view.init: latent group + geometry
          x::geo_pair<0, NB>(acc, XR, wrs, xpair, lane, graw, geo_make, false);
        }
        SYNC();
        {
          x::store_acts<NA_ACT_SIN, NB>(acc, hb, rg, lane, a.sat_gen);
          xbias(1);
          if (owner) {  // sin(latent) for the skip connection: the rows again (L2), through the activation, into the same group
            const Loc L = locate(pass, blk);
            const int t = L.tb * 32 + ln;
            const float* row = a.feat + ((int64_t)(t < a.T ? t : a.T - 1) * a.R + L.ray) * a.feat_ld;
            typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
            f32x16 n0, n1;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              const f32x4u p0 = *(const f32x4u*)(row + 1 + 16 * c + 4 * hi), p1 = *(const f32x4u*)(row + 1 + 16 * c + 8 + 4 * hi);
              const f32x4u q0 = *(const f32x4u*)(row + 33 + 16 * c + 4 * hi), q1 = *(const f32x4u*)(row + 33 + 16 * c + 8 + 4 * hi);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                n0[8 * c + e] = p0[e]; n0[8 * c + 4 + e] = p1[e];
                n1[8 * c + e] = q0[e]; n1[8 * c + 4 + e] = q1[e];
              }
            }
            x::store_block<NA_ACT_SIN, 4>(ib + blk * x::KQ, n0, n1, lane, a.sat_gen);
          }
          XR.pr[0] = x::wpair(wrs, lane, xpair, 1);
        }
        SYNC();
        {
          GeoRaw graw[NB];
#pragma unroll
          for (int b = 0; b < NB; ++b) graw[b] = geo_load(b);
          x::recs<2, NB, true, XNR, 5, 1, 1, 4>(acc, bvx, XR, wrs, xrec, 1, hb, lane, ib);      // view.L0: skip group, K = 256, geometry
          x::geo_pair<0, NB>(acc, XR, wrs, xpair, lane, graw, geo_make, true);
        }
        SYNC();
#pragma unroll 1
        for (int i = 0; i < 3; ++i) {
          x::store_acts<NA_ACT_SIN, NB>(acc, hb, rg, lane, a.sat_gen);
          xbias(2 + i);
          SYNC();
          x::recs<2, NB, true, XNR>(acc, bvx, XR, wrs, xrec, 6 + 4 * i, hb, lane);        // view.L1..L3
          SYNC();
        }
        f32x16 ocx[1][1], bo1[1];
        {
          bo1[0] = bias_tile(wrs, bias_rg + 5 * 1024, 0, lane);
          x::store_acts<NA_ACT_SIN, NB>(acc, hb, rg, lane, a.sat_gen);
        }
        SYNC();
        x::recs<1, 1, true, XNR>(ocx, bo1, XR, wrs, xrec, 18, hb + blk * x::BLKH, lane);   // view.out (block per wave)
        oc[0] = ocx[0][0];
        SYNC();
        prev = pass;
        continue;
      }
      m_hidden<PREC, 0, 4, 1, 0, false, PPP>(acc, ring, cur, wrs, wvoff, hb, ib, lane, geo_load, geo_make);  // view.init
      SYNC();
#define NA_VIEW_EPILOGUE(PH, FIRST)                                                              \
      {                                                                                            \
        f32x16 bv[2];                                                                              \
        store_acts<PREC, NA_ACT_SIN, 0, 1>(acc, hb, rg, lane);                                     \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        _Pragma("unroll") for (int t = 0; t < 2; ++t) bv[t] = bias_tile(wrs, bias_rg + (PH) * 1024, t, lane); \
        store_acts<PREC, NA_ACT_SIN, 1, 2>(acc, hb, rg, lane);                                     \
        if ((FIRST) && owner) activate_init<PREC, NA_ACT_SIN, 4>(ib, blk, lane);                   \
        _Pragma("unroll") for (int t = 0; t < 2; ++t)                                              \
          _Pragma("unroll") for (int b = 0; b < NB; ++b) acc[t][b] = bv[t];                        \
      }                                                                                            \
      SYNC();
      NA_VIEW_EPILOGUE(1, true)
      m_hidden<PREC, 1, 4, 2, 16, false, PPP>(acc, ring, cur, wrs, wvoff, hb, ib, lane, geo_load, geo_make);  // L0 (skip)
      SYNC();
      NA_VIEW_EPILOGUE(2, false)
      m_hidden<PREC, 2, 0, 0, 16, false, PPP>(acc, ring, cur, wrs, wvoff, hb, ib, lane, none_l, none_m);  // L1
      SYNC();
      NA_VIEW_EPILOGUE(3, false)
      m_hidden<PREC, 2, 0, 0, 16, false, PPP>(acc, ring, cur, wrs, wvoff, hb, ib, lane, none_l, none_m);  // L2
      SYNC();
      NA_VIEW_EPILOGUE(4, false)
      m_hidden<PREC, 2, 0, 0, 16, false, PPP>(acc, ring, cur, wrs, wvoff, hb, ib, lane, none_l, none_m);  // L3
      SYNC();
#undef NA_VIEW_EPILOGUE
      {
        oc[0] = bias_tile(wrs, bias_rg + 5 * 1024, 0, lane);
        store_acts<PREC, NA_ACT_SIN>(acc, hb, rg, lane);
      }
      SYNC();
      m_out<PREC, 2, 1, true, PPP>(oc, ring, cur, wrs, wvoff, hb, lane, blk);
      ring_skip<PREC, 2, 2, true, PPP>(ring, cur, wrs, wvoff);
      SYNC();
      prev = pass;
      continue;
    }
    if constexpr (MODEL == 1) {
      // ================= TinyNeRF: EP = sample position of this pass + compositing of the previous one
      auto none_l = [](int) { return 0; };
      auto none_m = [](int, int, bool) { return 0; };
      if (NB == 4 || owner) {
        prev_dn = own_dn;
        const TsPair tcur = ts_load(pass);
        TsPair tprev = tcur;
        if (prev >= 0) tprev = ts_load(prev);
        own_setup(pass);
        const Geom q = geom(pass, blk, tcur);
        if (prev >= 0) {
          // out tile of the previous pass: row 0 = density, rows 1..3 = colour (src/nerf.py:296-300, intended semantics)
          f32x16 rgbv = oc[0];
          rgbv[0] = oc[0][1]; rgbv[1] = oc[0][2]; rgbv[2] = oc[0][3];
          composite(prev_geom(prev, tprev), rgbv, oc[0][0]);
        }
        float v2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v2[e] = 0.f;
        if (hi == 0) { v2[0] = q.px; v2[1] = q.py; v2[2] = q.pz; }
        fwrite<PREC>(ib + blk * 4 * FR + lane * 16, make_frag<PREC>(v2));
      }
      {
        f32x16 bv[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) bv[t] = bias_tile(wrs, bias_rg + 0 * 1024, t, lane);
        if constexpr (PREC == NA_PREC_F16X) x::pairs_prefetch(XR, wrs, xpair, lane, 0, MODEL == 2 ? 2 : 1);
        SYNC();
        if (prev >= 0) combine(prev);
        if constexpr (PREC == NA_PREC_F16X) {
          bvx[0] = bv[0]; bvx[1] = bv[1];
        } else {
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[t][b] = bv[t];
        }
      }
      if constexpr (PREC == NA_PREC_F16X) {
        // ---- NA_PREC_F16X: init (pair 0), L0 (skip pair 1 + records 0..3), L1, L2, L3 (skip pair 2 + records 12..15), L4, L5,
        // out (records 24..27, one tile, block per wave)
        x::pairs<0, 1, NB>(acc, bvx, XR, wrs, xpair, ib, lane);
        SYNC();
        {
          x::store_acts<NA_ACT_LEAKY_RELU, NB>(acc, hb, rg, lane, a.sat_gen);
          xbias(1);
          if (owner) activate_init<PREC, NA_ACT_LEAKY_RELU, 1>(ib, blk, lane);
          x::pairs_prefetch(XR, wrs, xpair, lane, 1, 1);
        }
        SYNC();
        x::pairs<1, 1, NB>(acc, bvx, XR, wrs, xpair, ib, lane);
        x::recs<2, NB, false, XNR>(acc, bvx, XR, wrs, xrec, 0, hb, lane);                    // L0 (skip)
        SYNC();
#pragma unroll 1
        for (int i = 0; i < 2; ++i) {
          x::store_acts<NA_ACT_LEAKY_RELU, NB>(acc, hb, rg, lane, a.sat_gen);
          xbias(2 + i);
          SYNC();
          x::recs<2, NB, true, XNR>(acc, bvx, XR, wrs, xrec, 4 + 4 * i, hb, lane);             // L1, L2
          SYNC();
        }
        {
          x::store_acts<NA_ACT_LEAKY_RELU, NB>(acc, hb, rg, lane, a.sat_gen);
          xbias(4);
          x::pairs_prefetch(XR, wrs, xpair, lane, 2, 1);
        }
        SYNC();
        x::pairs<2, 1, NB>(acc, bvx, XR, wrs, xpair, ib, lane);
        x::recs<2, NB, false, XNR>(acc, bvx, XR, wrs, xrec, 12, hb, lane);                   // L3 (skip)
        SYNC();
#pragma unroll 1
        for (int i = 0; i < 2; ++i) {
          x::store_acts<NA_ACT_LEAKY_RELU, NB>(acc, hb, rg, lane, a.sat_gen);
          xbias(5 + i);
          SYNC();
          x::recs<2, NB, true, XNR>(acc, bvx, XR, wrs, xrec, 16 + 4 * i, hb, lane);            // L4, L5
          SYNC();
        }
        f32x16 ocx[1][1], bo1[1];
        {
          bo1[0] = bias_tile(wrs, bias_rg + 7 * 1024, 0, lane);
          x::store_acts<NA_ACT_LEAKY_RELU, NB>(acc, hb, rg, lane, a.sat_gen);
        }
        SYNC();
        x::recs<1, 1, true, XNR>(ocx, bo1, XR, wrs, xrec, 24, hb + blk * x::BLKH, lane);       // out
        oc[0] = ocx[0][0];
        SYNC();
        prev = pass;
        continue;
      }
      m_hidden<PREC, 0, 2, 0, 0, false, PPP>(acc, ring, cur, wrs, wvoff, hb, ib, lane, none_l, none_m);  // init
      SYNC();
      // six hidden layers; the input re-enters through the activation at layers 0 and 3 (src/neural_blocks.py:290-296)
#define NA_TINY_EPILOGUE(PH, FIRST)                                                              \
      {                                                                                            \
        f32x16 bv[2];                                                                              \
        store_acts<PREC, NA_ACT_LEAKY_RELU, 0, 1>(acc, hb, rg, lane);                              \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        _Pragma("unroll") for (int t = 0; t < 2; ++t) bv[t] = bias_tile(wrs, bias_rg + (PH) * 1024, t, lane); \
        store_acts<PREC, NA_ACT_LEAKY_RELU, 1, 2>(acc, hb, rg, lane);                              \
        if ((FIRST) && owner) activate_init<PREC, NA_ACT_LEAKY_RELU, 1>(ib, blk, lane);            \
        _Pragma("unroll") for (int t = 0; t < 2; ++t)                                              \
          _Pragma("unroll") for (int b = 0; b < NB; ++b) acc[t][b] = bv[t];                        \
      }                                                                                            \
      SYNC();
      NA_TINY_EPILOGUE(1, true)
      m_hidden<PREC, 2, 1, 0, 16, false, PPP>(acc, ring, cur, wrs, wvoff, hb, ib, lane, none_l, none_m);  // L0 (skip)
      SYNC();
      NA_TINY_EPILOGUE(2, false)
      m_hidden<PREC, 3, 0, 0, 16, false, PPP>(acc, ring, cur, wrs, wvoff, hb, ib, lane, none_l, none_m);  // L1
      SYNC();
      NA_TINY_EPILOGUE(3, false)
      m_hidden<PREC, 3, 0, 0, 16, false, PPP>(acc, ring, cur, wrs, wvoff, hb, ib, lane, none_l, none_m);  // L2
      SYNC();
      NA_TINY_EPILOGUE(4, false)
      m_hidden<PREC, 3, 1, 0, 16, false, PPP>(acc, ring, cur, wrs, wvoff, hb, ib, lane, none_l, none_m);  // L3 (skip)
      SYNC();
      NA_TINY_EPILOGUE(5, false)
      m_hidden<PREC, 0, 0, 0, 16, false, PPP>(acc, ring, cur, wrs, wvoff, hb, ib, lane, none_l, none_m);  // L4
      SYNC();
      NA_TINY_EPILOGUE(6, false)
      m_hidden<PREC, 0, 0, 0, 16, false, PPP>(acc, ring, cur, wrs, wvoff, hb, ib, lane, none_l, none_m);  // L5
      SYNC();
#undef NA_TINY_EPILOGUE
      {
        oc[0] = bias_tile(wrs, bias_rg + 7 * 1024, 0, lane);
        store_acts<PREC, NA_ACT_LEAKY_RELU>(acc, hb, rg, lane);
      }
      SYNC();
      m_out<PREC, 0, 1, true, PPP>(oc, ring, cur, wrs, wvoff, hb, lane, blk);
      SYNC();
      prev = pass;
      continue;
    }
    // ================= EP: hash encoder of this pass (round 4: the compositing of a pass runs at the end of that pass, in
    // the view.out phase, which is 2.5 k cycles of MFMAs opposite the partner's 8 k-cycle EP; ts / ray of this pass were
    // requested there too, so EP is the cross-block carry + positions + gathers: the group-wide critical path of the three
    // slots around the pass boundary is 17 k instead of 20 k cycles)
    // The 4 x 8 table gathers of this lane half go out one level at a time (35 live registers).  They are TA-bound (64
    // distinct 128-B lines per instruction, 8 MiB of tables): ~10k cycles per group and pass whether issued as four rounds,
    // two or one (measured); spreading the levels over the epilogues of the other layers was slower still (every gathering
    // wave stalls ~3k cycles per round and those epilogues have ~1.8k cycles of slack).
    STAMP(0);
    if (prev >= 0) combine(prev);  // (partials: written before the barrier that closed view.out; EP does not touch their chunk)
    if constexpr (PREC == NA_PREC_F16X) {
      if constexpr (MIP) { geo_setup(pass); mip_setup(); }
      hash_group_ep(pass);
      if constexpr (MIP) gen_ipe(std::integral_constant<int, NA_ACT_NONE>{});
    } else
    if (NB == 4 || owner) {  // (bf16x3: row groups 2,3 own no block -- nothing to encode or composite)
      STAMP(8);
      STAMP(9);
      const Geom q = geom(pass, blk, tnext);
      STAMP(10);
      __builtin_amdgcn_sched_barrier(0);
      STAMP(1);
      // bf16x3: levels 2,3 of this lane half are gathered by the partner row group rg + 2 (below), which owns no block
      constexpr int KEND = NB == 4 ? 4 : 2;
      HashGather hg;
#pragma unroll
      for (int k = 0; k < KEND; ++k) {
        hash_level_issue(q.px, q.py, q.pz, a.tables, (hi ? a.res.n[4 + k] : a.res.n[k]), 4 * hi + k, hg);
        hash_finish(k, hg);
        __builtin_amdgcn_sched_barrier(0);
        STAMP(3 + k);
      }
      float v2[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v2[e] = 0.f;
      if (hi == 0) { v2[0] = q.px; v2[1] = q.py; v2[2] = q.pz; v2[3] = q.px; v2[4] = q.py; v2[5] = q.pz; }
      fwrite<PREC>(ib + blk * 4 * FR + lane * 16 + 2 * FR, make_frag<PREC>(v2));
      STAMP(7);
    } else {
      // bf16x3, row groups 2 and 3: half of the hash encoder of block rg - 2 (levels 2,3 of each lane half), so that the
      // exposed phase is two gather rounds long instead of four
      const Geom q = geom(pass, blk, tnext);
      HashGather hg;
#pragma unroll
      for (int k = 2; k < 4; ++k) {
        hash_level_issue(q.px, q.py, q.pz, a.tables, (hi ? a.res.n[4 + k] : a.res.n[k]), 4 * hi + k, hg);
        hash_finish(k, hg);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    {
      f32x16 bv[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) bv[t] = bias_tile(wrs, bias_rg + 0 * 1024, t, lane);
      SYNC();
      if constexpr (PREC == NA_PREC_F16X) {
        bvx[0] = bv[0]; bvx[1] = bv[1];  // (C operand of first.init's first products)
      } else {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int b = 0; b < NB; ++b) acc[t][b] = bv[t];
      }
    }
    if constexpr (PREC == NA_PREC_F16X) {
      // ================= NA_PREC_F16X: the same twelve phases on pairs (init / geometry chunks) and records (hidden K)
      // the bias of the NEXT phase waits in bvx and becomes the C operand of that phase's first MFMAs (no accumulator init)
      auto load_bias2 = [&](int ph) {
#pragma unroll
        for (int t = 0; t < 2; ++t) bvx[t] = bias_tile(wrs, bias_rg + ph * 1024, t, lane);
      };
      // records per pass (44, parity of the index = scale slot): 0 first.init | 1 skip group + 2..5 first.L0 | 6.. L1..L3 |
      // 18..21 first.out | 22 view.init | 23 skip group + 24..27 view.L0 | 28.. L1..L3 | 40..43 view.out
      // MODEL 6 (mip): 52 records -- 0..2 first.init ([hash | x], IPE 0, IPE 1) | 3 skip group + 4..7 hidden + 8, 9 IPE (first.L0) |
      // 10.. L1..L3 | 22..25 first.out | 26..28 view.init (latent, IPE 0, IPE 1) | 29 skip group + 30..33 + 34, 35 IPE (view.L0) |
      // 36.. L1..L3 | 48..51 view.out; the IPE groups of a skip layer are regenerated (through the activation) between the
      // layer's two MFMA phases, whose accumulators stay in registers
      constexpr int IPS = 2 * x::KQ;  // block stride of the parked IPE groups
      constexpr int RL1 = MIP ? 10 : 6, ROUT = MIP ? 22 : 18, RVI = MIP ? 26 : 22, RVL0 = MIP ? 29 : 23, RVL1 = MIP ? 36 : 28,
                    RVO = MIP ? 48 : 40;
      if constexpr (MIP) x::recs<2, NB, true, XNR, 3, 1, 0, 3, false, IPS, 3, 2>(acc, bvx, XR, wrs, xrec, 0, hb, lane, ib);
      else x::recs<2, NB, true, XNR, 1, 1, 0, 3>(acc, bvx, XR, wrs, xrec, 0, hb, lane, ib);  // first.init: the [hash | x] group
      SYNC();
      {
        x::store_acts<NA_ACT_LEAKY_RELU, NB, 0, 1>(acc, hb, rg, lane, a.sat_gen);
        reenter_hash();
        x::store_acts<NA_ACT_LEAKY_RELU, NB, 1, 2>(acc, hb, rg, lane, a.sat_gen);
        load_bias2(1);
      }
      SYNC();
      if constexpr (MIP) {
        x::recs<2, NB, true, XNR, 5, 1, 1, 3>(acc, bvx, XR, wrs, xrec, 3, hb, lane, ib);     // first.L0: skip group + K = 256 ...
        SYNC();
        gen_ipe(std::integral_constant<int, NA_ACT_LEAKY_RELU>{});
        SYNC();
        x::recs<2, NB, false, XNR, 2, 0, 0, 4, false, IPS, 3, 2>(acc, bvx, XR, wrs, xrec, 8, hb, lane);  // ... + the IPE groups
      } else {
        x::recs<2, NB, true, XNR, 5, 1, 1, 3>(acc, bvx, XR, wrs, xrec, 1, hb, lane, ib);     // first.L0: skip group, then K = 256
      }
      SYNC();
#pragma unroll 1
      for (int i = 0; i < 3; ++i) {
        x::store_acts<NA_ACT_LEAKY_RELU, NB>(acc, hb, rg, lane, a.sat_gen);
        load_bias2(2 + i);
        SYNC();
        x::recs<2, NB, true, XNR>(acc, bvx, XR, wrs, xrec, RL1 + 4 * i, hb, lane);                // first.L1..L3
        SYNC();
      }
      // first.out, split by BLOCK (NB = 2): every row group runs two tiles for ONE block, rg & 1 -- row groups 0, 1 the two
      // latent tiles (rows 0..63: the 32 values of an MFMA lane of the View MLP's latent group end up in one lane, which converts
      // them like a hidden epilogue), row groups 2, 3 the density tile (row 64) and an all-zero tile.  One code path for the
      // four waves: a wave-dependent choice between two instantiations of the record loop made the compiler reconcile the
      // weight registers at the join through scratch memory (165 spilled registers, every phase 30 % slower).
      static_assert(NB == 2, "first.out by block");
      f32x16 ol[2][1];
      {
        x::store_acts<NA_ACT_LEAKY_RELU, NB>(acc, hb, rg, lane, a.sat_gen);
        bvx[0] = bias_tile(wrs, bias_rg + 5 * 1024, rg < 2 ? 0 : 2, lane);
        bvx[1] = bias_tile(wrs, bias_rg + 5 * 1024, rg < 2 ? 1 : 3, lane);  // (slot 3: zeros)
      }
      SYNC();
      x::recs<2, 1, true, XNR>(ol, bvx, XR, wrs, xrec, ROUT, hb + (rg & 1) * x::BLKH, lane);
      SYNC();
      {
        load_bias2(6);
        geo_setup(pass);
        if (rg < 2) {
          // the latent group of block rg: raw rows into the wave's own (idle) K64 region for the skip connection, the group
          // itself into the init region
          char* st = hb + x::BLKH + rg * x::KQ + lane * 16;  // (the wave's K64 region of block 1: E7 stores it last)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            *(f32x4*)(st + c * 1024) = f32x4{ol[0][0][4 * c], ol[0][0][4 * c + 1], ol[0][0][4 * c + 2], ol[0][0][4 * c + 3]};
            *(f32x4*)(st + 4096 + c * 1024) = f32x4{ol[1][0][4 * c], ol[1][0][4 * c + 1], ol[1][0][4 * c + 2], ol[1][0][4 * c + 3]};
          }
          x::store_block<NA_ACT_NONE, 4, true>(ib + rg * x::KQ, ol[0][0], ol[1][0], lane, a.sat_gen);
        } else if (hi == 0) {
          // density of block rg & 1 -> the spare dword of the latent group's T operand (lane = step), where it waits for the
          // compositing at the end of the pass: carried in a register it was the one value spilled AND re-stored every pass
          *(float*)(ib + (rg & 1) * x::KQ + 6144 + 1024 + ln * 16 + 12) = ol[0][0][0];
        }
        if constexpr (MIP) { mip_setup(); gen_ipe(std::integral_constant<int, NA_ACT_NONE>{}); }
        x::pairs_prefetch(XR, wrs, xpair, lane, 0, 1);
      }
      SYNC();
      {
        GeoRaw graw[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) graw[b] = geo_load(b);
        if constexpr (MIP) x::recs<2, NB, true, XNR, 3, 1, 0, 4, false, IPS, 3, 2>(acc, bvx, XR, wrs, xrec, RVI, hb, lane, ib);
        else x::recs<2, NB, true, XNR, 1, 1, 0, 4>(acc, bvx, XR, wrs, xrec, RVI, hb, lane, ib);  // view.init: latent group + geometry
        x::geo_pair<0, NB>(acc, XR, wrs, xpair, lane, graw, geo_make, false);
      }
      SYNC();
      {
        x::store_acts<NA_ACT_SIN, NB, 0, 1>(acc, hb, rg, lane, a.sat_gen);
        if (rg < 2) {  // sin(latent) for the skip connection, from the raw rows (before block 1's store overwrites their region)
          const char* st = hb + x::BLKH + rg * x::KQ + lane * 16;
          f32x16 l0, l1;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const f32x4 u = *(const f32x4*)(st + c * 1024), w = *(const f32x4*)(st + 4096 + c * 1024);
            l0[4 * c] = u[0]; l0[4 * c + 1] = u[1]; l0[4 * c + 2] = u[2]; l0[4 * c + 3] = u[3];
            l1[4 * c] = w[0]; l1[4 * c + 1] = w[1]; l1[4 * c + 2] = w[2]; l1[4 * c + 3] = w[3];
          }
          x::store_block<NA_ACT_SIN, 4, true>(ib + rg * x::KQ, l0, l1, lane, a.sat_gen);
        }
        x::store_acts<NA_ACT_SIN, NB, 1, 2>(acc, hb, rg, lane, a.sat_gen);
        load_bias2(7);
        if constexpr (!MIP) XR.pr[0] = x::wpair(wrs, lane, xpair, 1);  // (the second geometry pair, into the same ring slot: the first one is spent)
      }
      SYNC();
      if constexpr (MIP) {
        x::recs<2, NB, true, XNR, 5, 1, 1, 4>(acc, bvx, XR, wrs, xrec, RVL0, hb, lane, ib);   // view.L0: skip group + K = 256 ...
        SYNC();
        gen_ipe(std::integral_constant<int, NA_ACT_SIN>{});
        XR.pr[0] = x::wpair(wrs, lane, xpair, 1);
        SYNC();
        GeoRaw graw[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) graw[b] = geo_load(b);
        x::recs<2, NB, false, XNR, 2, 0, 0, 4, false, IPS, 3, 2>(acc, bvx, XR, wrs, xrec, RVL0 + 5, hb, lane);  // ... + IPE + geometry
        x::geo_pair<0, NB>(acc, XR, wrs, xpair, lane, graw, geo_make, true);
      } else {
        GeoRaw graw[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) graw[b] = geo_load(b);
        x::recs<2, NB, true, XNR, 5, 1, 1, 4>(acc, bvx, XR, wrs, xrec, RVL0, hb, lane, ib);  // view.L0: skip group, K = 256, geometry
        x::geo_pair<0, NB>(acc, XR, wrs, xpair, lane, graw, geo_make, true);
      }
      SYNC();
#pragma unroll 1
      for (int i = 0; i < 3; ++i) {
        x::store_acts<NA_ACT_SIN, NB>(acc, hb, rg, lane, a.sat_gen);
        load_bias2(8 + i);
        SYNC();
        x::recs<2, NB, true, XNR>(acc, bvx, XR, wrs, xrec, RVL1 + 4 * i, hb, lane);              // view.L1..L3
        SYNC();
      }
      f32x16 ocx[1][1];
      f32x16 bo[1];
      {
        bo[0] = bias_tile(wrs, bias_rg + 11 * 1024, 0, lane);
        x::store_acts<NA_ACT_SIN, NB>(acc, hb, rg, lane, a.sat_gen);
      }
      SYNC();
      x::recs<1, 1, true, XNR>(ocx, bo, XR, wrs, xrec, RVO, hb + blk * x::BLKH, lane);            // view.out (block per wave)
      oc[0] = ocx[0][0];
      pass_tail(pass);
      SYNC();
    } else {
    // ================= `first` MLP (LeakyReLU)
    m_hidden<PREC, 0, 3, 0, 0, false>(acc, ring, cur, wrs, wvoff, hb, ib, lane, [](int) { return 0; }, [](int, int, bool) { return 0; });
    SYNC();
    {
      f32x16 bv[2];
      // tile 0 first: its 16 x NBLK accumulator registers are dead before the bias of the next layer is loaded
      store_acts<PREC, NA_ACT_LEAKY_RELU, 0, 1>(acc, hb, rg, lane);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 2; ++t) bv[t] = bias_tile(wrs, bias_rg + 1 * 1024, t, lane);
      store_acts<PREC, NA_ACT_LEAKY_RELU, 1, 2>(acc, hb, rg, lane);
      if (owner) activate_init<PREC, NA_ACT_LEAKY_RELU, 3>(ib, blk, lane);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[t][b] = bv[t];
    }
    SYNC();
    m_hidden<PREC, 3, 3, 0, 16, false>(acc, ring, cur, wrs, wvoff, hb, ib, lane, [](int) { return 0; }, [](int, int, bool) { return 0; });
    SYNC();
    for (int i = 0; i < 3; ++i) {
      f32x16 bv[2];
      // tile 0 first: its 16 x NBLK accumulator registers are dead before the bias of the next layer is loaded
      store_acts<PREC, NA_ACT_LEAKY_RELU, 0, 1>(acc, hb, rg, lane);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 2; ++t) bv[t] = bias_tile(wrs, bias_rg + (2 + i) * 1024, t, lane);
      store_acts<PREC, NA_ACT_LEAKY_RELU, 1, 2>(acc, hb, rg, lane);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[t][b] = bv[t];
      SYNC();
      m_hidden<PREC, 2, 0, 0, 16, false>(acc, ring, cur, wrs, wvoff, hb, ib, lane, [](int) { return 0; }, [](int, int, bool) { return 0; });
      SYNC();
    }
    f32x16 oq[NB];  // first.out: this row group's tile (0, 1: latent rows 0..63; 2: density row 64) for the NB blocks
    {
      const f32x16 bo = bias_tile(wrs, bias_rg + 5 * 1024, rg < 2 ? rg : 2, lane);
#pragma unroll
      for (int b = 0; b < NB; ++b) oq[b] = bo;
      store_acts<PREC, NA_ACT_LEAKY_RELU>(acc, hb, rg, lane);
    }
    SYNC();
    // first.out: rows 0..63 = intermediate (-> View latent), row 64 = density
    m_out_rows<PREC, 2, false>(oq, ring, cur, wrs, wvoff, hb, lane);
    SYNC();
    {
      f32x16 bv[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) bv[t] = bias_tile(wrs, bias_rg + 6 * 1024, t, lane);
      geo_setup(pass);  // ray / elev / azim of the group's blocks -> SGPRs, for the two View phases that follow
      if (rg < 2) {
        // latent rows 32 rg .. 32 rg + 31 -> init chunks 2 rg, 2 rg + 1 of every block (the View MLP's B fragments)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          Frag<PREC> f0, f1;
          acc_to_frags<PREC, NA_ACT_NONE>(oq[b], f0, f1);
          char* dst = ib + (b * 4 + 2 * rg) * FR + lane * 16;
          fwrite<PREC>(dst, f0);
          fwrite<PREC>(dst + FR, f1);
        }
      } else if (rg == 2 && hi == 0) {
        // density (row 64 = register 0 of the hi = 0 lanes) of block b's 32 samples -> the idle hidden region; block b's
        // owner picks it up after the barrier (it composites the block one pass later)
#pragma unroll
        for (int b = 0; b < NB; ++b) ((float*)hb)[b * 32 + ln] = oq[b][0];
      }
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[t][b] = bv[t];
    }
    SYNC();
    if (owner) density = ((const float*)hb)[blk * 32 + ln];
    // ================= View MLP (sin)
    m_hidden<PREC, 2, 4, 1, 0, false>(acc, ring, cur, wrs, wvoff, hb, ib, lane, geo_load, geo_make);
    SYNC();
    {
      f32x16 bv[2];
      // tile 0 first: its 16 x NBLK accumulator registers are dead before the bias of the next layer is loaded
      store_acts<PREC, NA_ACT_SIN, 0, 1>(acc, hb, rg, lane);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 2; ++t) bv[t] = bias_tile(wrs, bias_rg + 7 * 1024, t, lane);
      store_acts<PREC, NA_ACT_SIN, 1, 2>(acc, hb, rg, lane);
      if (owner) activate_init<PREC, NA_ACT_SIN, 4>(ib, blk, lane);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[t][b] = bv[t];
    }
    SYNC();
    m_hidden<PREC, 3, 4, 2, 16, false>(acc, ring, cur, wrs, wvoff, hb, ib, lane, geo_load, geo_make);
    SYNC();
    for (int i = 0; i < 3; ++i) {
      f32x16 bv[2];
      // tile 0 first: its 16 x NBLK accumulator registers are dead before the bias of the next layer is loaded
      store_acts<PREC, NA_ACT_SIN, 0, 1>(acc, hb, rg, lane);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 2; ++t) bv[t] = bias_tile(wrs, bias_rg + (8 + i) * 1024, t, lane);
      store_acts<PREC, NA_ACT_SIN, 1, 2>(acc, hb, rg, lane);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[t][b] = bv[t];
      SYNC();
      m_hidden<PREC, 0, 0, 0, 16, false>(acc, ring, cur, wrs, wvoff, hb, ib, lane, [](int) { return 0; }, [](int, int, bool) { return 0; });
      SYNC();
    }
    {
      oc[0] = bias_tile(wrs, bias_rg + 11 * 1024, 0, lane);
      store_acts<PREC, NA_ACT_SIN>(acc, hb, rg, lane);
    }
    SYNC();
    m_out<PREC, 0, 1, true>(oc, ring, cur, wrs, wvoff, hb, lane, blk);
    pass_tail(pass);
    SYNC();
    }  // (PREC != NA_PREC_F16X)
    prev = pass;
  }
  if (!M0 && MODEL < 4 && prev >= 0 && (NB == 4 || owner)) {  // (MODEL 0 composited its last pass in that pass's view.out phase)
    prev_dn = own_dn;
    if constexpr (MODEL == 1) {
      f32x16 rgbv = oc[0];
      rgbv[0] = oc[0][1]; rgbv[1] = oc[0][2]; rgbv[2] = oc[0][3];
      composite(prev_geom(prev, ts_load(prev)), rgbv, oc[0][0]);
    } else {
      composite(prev_geom(prev, ts_load(prev)), oc[0], density);
    }
  }
  __syncthreads();
  if ((MODEL < 4 || M0) && prev >= 0) combine(prev);
  if (g == 0) {  // group 0 takes its extra barriers at the end
#pragma unroll 1
    for (int i = 0; i < LAG; ++i) __syncthreads();
  }
}

// ================================================================================================ pack
#if NA_PREC_INST == 0
// dir_to_elev_azim of every ray, once (src/utils.py:247-254): the View MLP's geometry chunk reads it per block
__global__ void ray_elaz_kernel(const float* __restrict__ rays, int64_t R, float* __restrict__ elaz) {
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < R; r += (int64_t)gridDim.x * blockDim.x) {
    float el, az;
    elev_azim(rays[r * 6 + 3], rays[r * 6 + 4], rays[r * 6 + 5], el, az);
    elaz[r * 2] = el;
    elaz[r * 2 + 1] = az;
  }
}

struct PackArgs {
  const float* w_first[6];  // init, layers.0..3, out   (nn.Linear layout [out,in])
  const float* b_first[6];
  const float* w_view[6];
  const float* b_view[6];
};

__host__ __device__ inline int phase_first_frag(int p) {
  int s = 0;
  for (int i = 0; i < p; ++i) s += 2 * phase_pairs(i);
  return s;
}

// One thread per bf16 element of the hi plane of every fragment, plus the bias blocks.
__global__ void pack_ls_kernel(PackArgs w, int planes, int f16, char* __restrict__ dst) {
  const NaMlpDesc d1 = {3, NA_ENC_HASH, 35, 0, 4, 256, 65, 3, NA_ACT_LEAKY_RELU, NA_LAYOUT_PLAIN_FIRST};
  const NaMlpDesc d2 = {5, NA_ENC_NONE, 0, 64, 4, 256, 3, 3, NA_ACT_SIN, NA_LAYOUT_PLAIN_VIEW};
  const int frag_bytes = 1024 * planes;
  const int64_t nfrag_rg = 2 * kPairsPerPass;
  const int64_t nelem = 4 * nfrag_rg * 512;
  const int64_t nbias = 4 * kNPhase * 256;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nelem + nbias; i += (int64_t)gridDim.x * blockDim.x) {
    if (i < nelem) {
      const int e = (int)(i & 7), l = (int)((i >> 3) & 63);
      const int64_t fg = i >> 9;
      const int rg = (int)(fg / nfrag_rg);
      int f = (int)(fg % nfrag_rg);
      int p = 0;
      while (f >= 2 * phase_pairs(p)) { f -= 2 * phase_pairs(p); ++p; }
      const int kappa = 8 * (l >> 5) + e;
      const bool view = p >= 6;
      const NaMlpDesc& d = view ? d2 : d1;
      const int lp = view ? p - 6 : p;  // 0 init, 1 skip layer, 2..4 hidden, 5 out
      const float* W = view ? w.w_view[lp] : w.w_first[lp];
      const int dim_p = d.in_size + d.enc_dims + d.latent_size;
      int row, col, in_dim, out_dim;
      if (lp == 5) {  // out layers.  view.out: fragment f = chunk c (one tile); first.out: row group rg holds tile min(rg, 2)
        const int c = f, j = view ? 0 : (rg < 2 ? rg : 2);
        row = out_row_map(d, 32 * j + (l & 31));
        col = 16 * c + pi_perm(kappa);
        in_dim = kHidden; out_dim = d.out_size;
      } else {
        const int q = f >> 1, t = f & 1;
        row = 32 * (2 * rg + t) + (l & 31);
        out_dim = kHidden;
        if (lp == 0) { col = init_slot_feature(d, q, kappa); in_dim = dim_p; }
        else if (lp == 1) {
          // chunk order of the skip layer: init chunks from LDS, the 16 hidden chunks, then (View) the geometry chunk
          const int nlds = view ? 4 : 3;
          if (q < nlds) { col = init_slot_feature(d, q, kappa); if (col >= 0) col += kHidden; }
          else if (q < nlds + kHC) col = 16 * (q - nlds) + pi_perm(kappa);
          else { col = init_slot_feature(d, 4, kappa); if (col >= 0) col += kHidden; }
          in_dim = kHidden + dim_p;
        } else { col = 16 * q + pi_perm(kappa); in_dim = kHidden; }
      }
      float v = 0.f;
      if (row >= 0 && row < out_dim && col >= 0 && col < in_dim) v = W[(int64_t)row * in_dim + col];
      char* o = dst + kHeaderBytes + kBiasBytes + ((int64_t)rg * nfrag_rg + (fg % nfrag_rg)) * frag_bytes + l * 16 + e * 2;
      const __bf16 h = f16 ? to_elem<NA_PREC_F16>(v) : (__bf16)v;
      *(uint16_t*)o = __builtin_bit_cast(uint16_t, h);
      if (planes == 2) {
        const __bf16 lo = (__bf16)(v - (float)h);
        *(uint16_t*)(o + 1024) = __builtin_bit_cast(uint16_t, lo);
      }
    } else {
      const int64_t q = i - nelem;
      const int k = (int)(q & 255), p = (int)((q >> 8) % kNPhase), rg = (int)((q >> 8) / kNPhase);
      const bool view = p >= 6;
      const NaMlpDesc& d = view ? d2 : d1;
      const int lp = view ? p - 6 : p;
      const float* B = p >= 12 ? nullptr : view ? w.b_view[lp] : w.b_first[lp];  // (bias blocks 12..kNPhase-1: other schedules)
      const int slot = k >> 5, hi = (k >> 4) & 1, r = k & 15;
      const int rin = (r & 3) + 8 * (r >> 2) + 4 * hi;
      float v = 0.f;
      if (lp == 5) {
        const int row = slot < (view ? 1 : 3) ? out_row_map(d, 32 * slot + rin) : -1;
        if (row >= 0 && row < d.out_size && B != nullptr) v = B[row];
      } else if (slot < 2 && B != nullptr) {
        v = B[32 * (2 * rg + slot) + rin];
      }
      *(float*)(dst + kHeaderBytes + ((int64_t)rg * kNPhase + p) * 1024 + k * 4) = v;
    }
  }
}

__global__ void pack_ls_header_kernel(uint32_t magic, uint32_t precision, uint32_t* __restrict__ dst, uint32_t pairs) {
  if (threadIdx.x == 0) { dst[0] = magic; dst[1] = precision; dst[2] = pairs; dst[3] = kNPhase; }
}

// TinyNeRF stream (MODEL 1): same element order as pack_ls_kernel, phases per tiny_phase_pairs
struct TinyPackArgs {
  const float* w[8];  // init, layers.0..5, out   (nn.Linear layout [out,in])
  const float* b[8];
};
__global__ void pack_ls_tiny_kernel(TinyPackArgs w, int planes, int f16, char* __restrict__ dst) {
  const NaMlpDesc d = {3, NA_ENC_NONE, 0, 0, 6, 256, 4, 3, NA_ACT_LEAKY_RELU, NA_LAYOUT_GENERIC};
  const int frag_bytes = 1024 * planes;
  const int64_t nfrag_rg = 2 * kTinyPairs;
  const int64_t nelem = 4 * nfrag_rg * 512;
  const int64_t nbias = 4 * kNPhase * 256;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nelem + nbias; i += (int64_t)gridDim.x * blockDim.x) {
    if (i < nelem) {
      const int e = (int)(i & 7), l = (int)((i >> 3) & 63);
      const int64_t fg = i >> 9;
      const int rg = (int)(fg / nfrag_rg);
      int f = (int)(fg % nfrag_rg);
      int p = 0;
      while (f >= 2 * tiny_phase_pairs(p)) { f -= 2 * tiny_phase_pairs(p); ++p; }
      const int kappa = 8 * (l >> 5) + e;
      const float* W = w.w[p];  // p: 0 init, 1..6 layers.0..5, 7 out
      int row, col, in_dim, out_dim;
      if (p == 7) {  // fragment f = chunk c of the single out tile
        row = out_row_map(d, l & 31);
        col = 16 * f + pi_perm(kappa);
        in_dim = kHidden; out_dim = d.out_size;
      } else {
        const int q = f >> 1, t = f & 1;
        row = 32 * (2 * rg + t) + (l & 31);
        out_dim = kHidden;
        if (p == 0) { col = q == 0 ? init_slot_feature(d, 0, kappa) : -1; in_dim = d.in_size; }
        else if (p == 1 || p == 4) {  // [hidden | init] in the reference's column order, init chunk first in the stream
          if (q == 0) { col = init_slot_feature(d, 0, kappa); if (col >= 0) col += kHidden; }
          else col = 16 * (q - 1) + pi_perm(kappa);
          in_dim = kHidden + d.in_size;
        } else { col = 16 * q + pi_perm(kappa); in_dim = kHidden; }
      }
      float v = 0.f;
      if (row >= 0 && row < out_dim && col >= 0 && col < in_dim) v = W[(int64_t)row * in_dim + col];
      char* o = dst + kHeaderBytes + kBiasBytes + ((int64_t)rg * nfrag_rg + (fg % nfrag_rg)) * frag_bytes + l * 16 + e * 2;
      const __bf16 h = f16 ? to_elem<NA_PREC_F16>(v) : (__bf16)v;
      *(uint16_t*)o = __builtin_bit_cast(uint16_t, h);
      if (planes == 2) {
        const __bf16 lo = (__bf16)(v - (float)h);
        *(uint16_t*)(o + 1024) = __builtin_bit_cast(uint16_t, lo);
      }
    } else {
      const int64_t q = i - nelem;
      const int k = (int)(q & 255), p = (int)((q >> 8) % kNPhase), rg = (int)((q >> 8) / kNPhase);
      const int slot = k >> 5, hi = (k >> 4) & 1, r = k & 15;
      const int rin = (r & 3) + 8 * (r >> 2) + 4 * hi;
      float v = 0.f;
      if (p < kTinyPhases && w.b[p] != nullptr) {
        if (p == 7) {
          const int row = slot < 1 ? out_row_map(d, rin) : -1;
          if (row >= 0 && row < d.out_size) v = w.b[p][row];
        } else if (slot < 2) {
          v = w.b[p][32 * (2 * rg + slot) + rin];
        }
      }
      *(float*)(dst + kHeaderBytes + ((int64_t)rg * kNPhase + p) * 1024 + k * 4) = v;
    }
  }
}

// View head stream (MODEL 2): same element order, phases per view_phase_pairs; the last two pairs of the out phase are zero
struct ViewPackArgs {
  const float* w[6];  // init, layers.0..3, out
  const float* b[6];
};
__global__ void pack_ls_view_kernel(ViewPackArgs w, int planes, int f16, char* __restrict__ dst) {
  const NaMlpDesc d = {5, NA_ENC_NONE, 0, 64, 4, 256, 3, 3, NA_ACT_SIN, NA_LAYOUT_PLAIN_VIEW};
  const int frag_bytes = 1024 * planes;
  const int64_t nfrag_rg = 2 * kViewPairs;
  const int64_t nelem = 4 * nfrag_rg * 512;
  const int64_t nbias = 4 * kNPhase * 256;
  const int dim_p = d.in_size + d.latent_size;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nelem + nbias; i += (int64_t)gridDim.x * blockDim.x) {
    if (i < nelem) {
      const int e = (int)(i & 7), l = (int)((i >> 3) & 63);
      const int64_t fg = i >> 9;
      const int rg = (int)(fg / nfrag_rg);
      int f = (int)(fg % nfrag_rg);
      int p = 0;
      while (f >= 2 * view_phase_pairs(p)) { f -= 2 * view_phase_pairs(p); ++p; }
      const int kappa = 8 * (l >> 5) + e;
      const float* W = w.w[p];  // p: 0 init, 1..4 layers.0..3, 5 out
      int row = -1, col = -1, in_dim = 1, out_dim = 0;
      if (p == 5) {
        if (f < 16) {  // fragment f = chunk f of the single out tile; fragments 16..19 are padding
          row = out_row_map(d, l & 31);
          col = 16 * f + pi_perm(kappa);
          in_dim = kHidden; out_dim = d.out_size;
        }
      } else {
        const int q = f >> 1, t = f & 1;
        row = 32 * (2 * rg + t) + (l & 31);
        out_dim = kHidden;
        if (p == 0) { col = init_slot_feature(d, q, kappa); in_dim = dim_p; }  // q = 0..3 latent chunks, 4 = geometry chunk
        else if (p == 1) {  // init chunks from LDS, the 16 hidden chunks, then the geometry chunk
          if (q < 4) { col = init_slot_feature(d, q, kappa); if (col >= 0) col += kHidden; }
          else if (q < 4 + kHC) col = 16 * (q - 4) + pi_perm(kappa);
          else { col = init_slot_feature(d, 4, kappa); if (col >= 0) col += kHidden; }
          in_dim = kHidden + dim_p;
        } else { col = 16 * q + pi_perm(kappa); in_dim = kHidden; }
      }
      float v = 0.f;
      if (row >= 0 && row < out_dim && col >= 0 && col < in_dim) v = W[(int64_t)row * in_dim + col];
      char* o = dst + kHeaderBytes + kBiasBytes + ((int64_t)rg * nfrag_rg + (fg % nfrag_rg)) * frag_bytes + l * 16 + e * 2;
      const __bf16 h = f16 ? to_elem<NA_PREC_F16>(v) : (__bf16)v;
      *(uint16_t*)o = __builtin_bit_cast(uint16_t, h);
      if (planes == 2) {
        const __bf16 lo = (__bf16)(v - (float)h);
        *(uint16_t*)(o + 1024) = __builtin_bit_cast(uint16_t, lo);
      }
    } else {
      const int64_t q = i - nelem;
      const int k = (int)(q & 255), p = (int)((q >> 8) % kNPhase), rg = (int)((q >> 8) / kNPhase);
      const int slot = k >> 5, hi = (k >> 4) & 1, r = k & 15;
      const int rin = (r & 3) + 8 * (r >> 2) + 4 * hi;
      float v = 0.f;
      if (p < kViewPhases && w.b[p] != nullptr) {
        if (p == 5) {
          const int row = slot < 1 ? out_row_map(d, rin) : -1;
          if (row >= 0 && row < d.out_size) v = w.b[p][row];
        } else if (slot < 2) {
          v = w.b[p][32 * (2 * rg + slot) + rin];
        }
      }
      *(float*)(dst + kHeaderBytes + ((int64_t)rg * kNPhase + p) * 1024 + k * 4) = v;
    }
  }
}

// SIREN-VolSDF stream (MODEL 3): the SIREN SDF network (7 Linears) followed by the View head (6 Linears)
struct SirenPackArgs {
  const float* ws[7];  // sdf: init, layers.0..4, out
  const float* bs[7];
  const float* wv[6];  // view: init, layers.0..3, out
  const float* bv[6];
};
__global__ void pack_ls_siren_kernel(SirenPackArgs w, int planes, int f16, char* __restrict__ dst) {
  const NaMlpDesc d1 = {3, NA_ENC_NONE, 0, 0, 5, 256, 65, 3, NA_ACT_SIN, NA_LAYOUT_PLAIN_FIRST};
  const NaMlpDesc d2 = {5, NA_ENC_NONE, 0, 64, 4, 256, 3, 3, NA_ACT_SIN, NA_LAYOUT_PLAIN_VIEW};
  const int frag_bytes = 1024 * planes;
  const int64_t nfrag_rg = 2 * kSirenPairs;
  const int64_t nelem = 4 * nfrag_rg * 512;
  const int64_t nbias = 4 * kNPhase * 256;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nelem + nbias; i += (int64_t)gridDim.x * blockDim.x) {
    if (i < nelem) {
      const int e = (int)(i & 7), l = (int)((i >> 3) & 63);
      const int64_t fg = i >> 9;
      const int rg = (int)(fg / nfrag_rg);
      int f = (int)(fg % nfrag_rg);
      int p = 0;
      while (f >= 2 * siren_phase_pairs(p)) { f -= 2 * siren_phase_pairs(p); ++p; }
      const int kappa = 8 * (l >> 5) + e;
      const bool view = p >= 7;
      const int lp = view ? p - 7 : p;  // sdf: 0 init, 1..5 layers.0..4, 6 out;  view: 0 init, 1..4 layers.0..3, 5 out
      const float* W = view ? w.wv[lp] : w.ws[lp];
      int row = -1, col = -1, in_dim = 1, out_dim = 0;
      if (!view) {
        if (lp == 6) {  // row group rg holds tile min(rg, 2) of the 65 rows; fragment f = chunk f
          row = out_row_map(d1, 32 * (rg < 2 ? rg : 2) + (l & 31));
          col = 16 * f + pi_perm(kappa);
          in_dim = kHidden; out_dim = d1.out_size;
        } else {
          const int q = f >> 1, t = f & 1;
          row = 32 * (2 * rg + t) + (l & 31);
          out_dim = kHidden;
          if (lp == 0) { col = q == 0 ? init_slot_feature(d1, 0, kappa) : -1; in_dim = d1.in_size; }
          else if (lp == 1 || lp == 4) {
            if (q == 0) { col = init_slot_feature(d1, 0, kappa); if (col >= 0) col += kHidden; }
            else col = 16 * (q - 1) + pi_perm(kappa);
            in_dim = kHidden + d1.in_size;
          } else { col = 16 * q + pi_perm(kappa); in_dim = kHidden; }
        }
      } else {
        const int dim_p = d2.in_size + d2.latent_size;
        if (lp == 5) {
          if (f < 16) {
            row = out_row_map(d2, l & 31);
            col = 16 * f + pi_perm(kappa);
            in_dim = kHidden; out_dim = d2.out_size;
          }
        } else {
          const int q = f >> 1, t = f & 1;
          row = 32 * (2 * rg + t) + (l & 31);
          out_dim = kHidden;
          if (lp == 0) { col = init_slot_feature(d2, q, kappa); in_dim = dim_p; }
          else if (lp == 1) {
            if (q < 4) { col = init_slot_feature(d2, q, kappa); if (col >= 0) col += kHidden; }
            else if (q < 4 + kHC) col = 16 * (q - 4) + pi_perm(kappa);
            else { col = init_slot_feature(d2, 4, kappa); if (col >= 0) col += kHidden; }
            in_dim = kHidden + dim_p;
          } else { col = 16 * q + pi_perm(kappa); in_dim = kHidden; }
        }
      }
      float v = 0.f;
      if (row >= 0 && row < out_dim && col >= 0 && col < in_dim) v = W[(int64_t)row * in_dim + col];
      char* o = dst + kHeaderBytes + kBiasBytes + ((int64_t)rg * nfrag_rg + (fg % nfrag_rg)) * frag_bytes + l * 16 + e * 2;
      const __bf16 h = f16 ? to_elem<NA_PREC_F16>(v) : (__bf16)v;
      *(uint16_t*)o = __builtin_bit_cast(uint16_t, h);
      if (planes == 2) {
        const __bf16 lo = (__bf16)(v - (float)h);
        *(uint16_t*)(o + 1024) = __builtin_bit_cast(uint16_t, lo);
      }
    } else {
      const int64_t q = i - nelem;
      const int k = (int)(q & 255), p = (int)((q >> 8) % kNPhase), rg = (int)((q >> 8) / kNPhase);
      const int slot = k >> 5, hi = (k >> 4) & 1, r = k & 15;
      const int rin = (r & 3) + 8 * (r >> 2) + 4 * hi;
      float v = 0.f;
      if (p < kSirenPhases) {
        const bool view = p >= 7;
        const int lp = view ? p - 7 : p;
        const float* B = view ? w.bv[lp] : w.bs[lp];
        if (B != nullptr) {
          if (!view && lp == 6) {
            const int row = slot < 3 ? out_row_map(d1, 32 * slot + rin) : -1;
            if (row >= 0 && row < d1.out_size) v = B[row];
          } else if (view && lp == 5) {
            const int row = slot < 1 ? out_row_map(d2, rin) : -1;
            if (row >= 0 && row < d2.out_size) v = B[row];
          } else if (slot < 2) {
            v = B[32 * (2 * rg + slot) + rin];
          }
        }
      }
      *(float*)(dst + kHeaderBytes + ((int64_t)rg * kNPhase + p) * 1024 + k * 4) = v;
    }
  }
}

#endif  // NA_PREC_INST == 0

static std::atomic<uint32_t> g_lsx_launch_id{0};  // ids of the NA_PREC_F16X launches (range guard), shared by every schedule
// per-device hipFuncSetAttribute bookkeeping (the attribute is per device, not per thread)
template <int PREC, int MODEL = 0>
static int launch(Args& a, hipStream_t stream) {
  using C = Cfg<PREC>;
  auto kern = render_ls_kernel<PREC, MODEL>;
  static std::atomic<uint64_t> attr_done{0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { set_error("hipGetDevice failed"); return NA_EHIP; }
  const uint64_t bit = 1ull << (dev & 63);
  if (!(attr_done.load(std::memory_order_acquire) & bit)) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return NA_EHIP; }
    attr_done.fetch_or(bit, std::memory_order_release);
  }
  const int64_t wgs = (a.R + 1) / 2;  // at least one ray per sample group
  const int grid = wgs < 256 ? (int)wgs : 256;
  a.nG = 2 * grid;
  const int64_t rays_per_group = (a.R + a.nG - 1) / a.nG;
  a.npg = (int)((rays_per_group * a.nb + C::NBLK - 1) / C::NBLK);
  a.nb_magic = (1ull << 32) / (uint64_t)a.nb + 1;
  if ((int64_t)(a.npg + 1) * C::NBLK * a.nb >= (1ll << 32)) { set_error("na_render_plain_view_ls: batch too large"); return NA_EINVAL; }
  if constexpr (PREC == NA_PREC_F16X) {
    // ONE counter for all schedules: the flag ring is shared, and a stale id left by a saturated launch of one schedule
    // must never equal the id of a later launch of another (a per-instantiation counter did exactly that: the frame after
    // tests/test_gpu_range.py's saturated PlainNeRF launch came out poisoned in whichever VolSDF kernel reached the same count)
    uint32_t g = g_lsx_launch_id.fetch_add(1, std::memory_order_relaxed) + 1;
    if (g == 0) g = g_lsx_launch_id.fetch_add(1, std::memory_order_relaxed) + 1;  // (0 is the flag's initial value: never an id)
    a.sat_gen = g;
  } else {
    a.sat_gen = 0;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 2 * C::GROUP, stream, a);
  if constexpr (PREC == NA_PREC_F16X) {  // range guard: NaN output if any activation of this launch sat at the half clamp
    if constexpr (MODEL == 4 || MODEL == 5)
      hipLaunchKernelGGL(lsx_poison_kernel, dim3(grid_for((int64_t)a.T * a.R * a.y_ld, 256, 1024)), dim3(256), 0, stream, a.sat_gen, a.y,
                         (int64_t)a.T * a.R * a.y_ld);
    else
      hipLaunchKernelGGL(lsx_poison_kernel, dim3(grid_for(a.R * 3, 256, 256)), dim3(256), 0, stream, a.sat_gen, a.out, a.R * 3);
  }
  return check_launch("na_render_plain_view_ls");
}

}  // namespace ls

#if NA_PREC_INST == 3
namespace ls {
// ---- NA_PREC_F16X weight streams (layout: namespace x above), built from a schedule table: which Linear every pair / record /
// bias block belongs to.  One set of kernels for the four schedules.
// column of Linear rd.lin's weight matrix that sits in k-slot kappa of chunk c of the record's K64 group; -1 = zero
__device__ __forceinline__ int xrec_col(const XSched& sc, const XRecD& rd, int c, int kappa) {
  if (rd.kind == 0) return 64 * rd.q + 16 * c + pi_perm(kappa);  // hidden feature (the skip layers store [hidden | init])
  if (rd.kind == 3 || rd.kind == 4) {
    // Fourier group q as the MODEL 5 generator lays it out: slot s = 8 c + e of lane half h holds frequency f = 32 q + 16 h + s / 2,
    // its sine (s even) or cosine (s odd).  Reference columns: [p | sin(128) | cos(128)] (src/neural_blocks.py:36-55, 283-287)
    const NaMlpDesc& d = sc.desc[sc.lin[rd.lin].desc];
    const int F = d.enc_dims / 2, s = 8 * c + (kappa & 7), f = 32 * rd.q + 16 * (kappa >> 3) + (s >> 1);
    return (rd.kind == 4 ? kHidden : 0) + d.in_size + ((s & 1) ? F + f : f);
  }
  if (rd.kind == 5) {
    // IPE group q (0, 1) as the MODEL 6 generator lays it out: slot s = 8 c + e < 24 of lane half h holds the (degree, axis)
    // pair pidx = 24 q + 12 h + s / 2 (the fourth chunk of both groups is padding: the two generating waves do the same work),
    // its sine feature (s even: latent column pidx) or cosine feature (s odd: column 48 + pidx); src/utils.py:23-27 layout
    // [sin | cos], degree-major
    const int s = 8 * c + (kappa & 7), h = kappa >> 3;
    const int pidx = s < 24 ? 24 * rd.q + 12 * h + (s >> 1) : -1;
    return pidx < 0 ? -1 : rd.off + ((s & 1) ? 48 : 0) + pidx;
  }
  int col = init_slot_feature(sc.desc[sc.lin[rd.lin].desc], c, kappa);
  if (col >= 0 && rd.kind == 2) col += kHidden;
  return col < 0 ? col : col + rd.off;
}
// weight row held by lane l of tile t of row group rg; -1 = zero
__device__ __forceinline__ int xrec_row(const XSched& sc, const XRecD& rd, int rg, int t, int l) {
  const NaMlpDesc& d = sc.desc[sc.lin[rd.lin].desc];
  if (rd.out_mode == 0) return 32 * (2 * rg + t) + (l & 31);
  if (rd.out_mode == 3) return rg < 2 ? out_row_map(d, 32 * t + (l & 31)) : (t == 0 ? out_row_map(d, 64 + (l & 31)) : -1);
  return t == 0 ? out_row_map(d, (rd.out_mode == 1 ? 32 * (rg < 2 ? rg : 2) : 0) + (l & 31)) : -1;
}
__global__ void pack_lsx_f16_kernel(XSched sc, char* __restrict__ dst) {
  // one thread per 16-bit element of the f16 planes of the pairs and of the records' f16 fragments
  const int srg = sc.npair * x::PAIRB + sc.nrec * x::REC;
  const int64_t npair_e = 4ll * sc.npair * 2 * 512;  // [rg][pair][tile][lane][8]
  const int64_t nrec_e = 4ll * sc.nrec * 8 * 512;    // [rg][rec][tile*4+chunk][lane][8]
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < npair_e + nrec_e; i += (int64_t)gridDim.x * blockDim.x) {
    if (i < npair_e) {
      const int e = (int)(i & 7), l = (int)((i >> 3) & 63), t = (int)((i >> 9) & 1);
      const int pi = (int)((i >> 10) % sc.npair), rg = (int)((i >> 10) / sc.npair);
      const XPairD pd = sc.pair[pi];
      const XLin L = sc.lin[pd.lin];
      int col = init_slot_feature(sc.desc[L.desc], pd.q, 8 * (l >> 5) + e);
      if (col >= 0 && pd.skip) col += kHidden;
      const int row = 32 * (2 * rg + t) + (l & 31);
      float v = 0.f;
      if (col >= 0 && col < L.in_dim) v = L.W[(int64_t)row * L.in_dim + col];
      const __bf16 h = to_elem<NA_PREC_F16X>(v);
      const __bf16 lo = to_elem<NA_PREC_F16X, false>(v - from_elem<NA_PREC_F16X>(h));
      char* o = dst + kHeaderBytes + kBiasBytes + (int64_t)rg * srg + pi * x::PAIRB + t * 2048 + l * 16 + e * 2;
      *(uint16_t*)o = __builtin_bit_cast(uint16_t, h);
      *(uint16_t*)(o + 1024) = __builtin_bit_cast(uint16_t, lo);
    } else {
      const int64_t k = i - npair_e;
      const int e = (int)(k & 7), l = (int)((k >> 3) & 63), f = (int)((k >> 9) & 7);
      const int ri = (int)((k >> 12) % sc.nrec), rg = (int)((k >> 12) / sc.nrec);
      const int t = f >> 2, c = f & 3;
      const XRecD rd = sc.rec[ri];
      const XLin L = sc.lin[rd.lin];
      const int col = xrec_col(sc, rd, c, 8 * (l >> 5) + e);
      const int row = xrec_row(sc, rd, rg, t, l);
      float v = 0.f;
      if (row >= 0 && row < L.out_dim && col >= 0 && col < L.in_dim) v = L.W[(int64_t)row * L.in_dim + col];
      char* o = dst + kHeaderBytes + kBiasBytes + (int64_t)rg * srg + sc.npair * x::PAIRB + (int64_t)ri * x::REC + f * 1024 + l * 16 + e * 2;
      *(uint16_t*)o = __builtin_bit_cast(uint16_t, to_elem<NA_PREC_F16X>(v));
    }
  }
}
// one thread per (row group, record, tile, lane): the lane's 32 weights of the K64 group -> WL6 and the scale bytes of WL6 and
// of WT6 (which the render kernel derives from the f16 fragments with that scale).  WL6 pairs with the activations' T plane:
// slot order = what v_cvt_scalef32_2xpk16_fp6_f32 gives it: slot 2 r <-> (producer tile 0, register r), slot 2 r + 1 <->
// (producer tile 1, register r), i.e. hidden feature 64 Q + 32 tt + (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
__global__ void pack_lsx_fp6_kernel(XSched sc, char* __restrict__ dst) {
  const int srg = sc.npair * x::PAIRB + sc.nrec * x::REC;
  const int64_t n = 4ll * sc.nrec * 2 * 64;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int l = (int)(i & 63), t = (int)((i >> 6) & 1);
    const int ri = (int)((i >> 7) % sc.nrec), rg = (int)((i >> 7) / sc.nrec);
    const int h = l >> 5;
    const XRecD rd = sc.rec[ri];
    const XLin L = sc.lin[rd.lin];
    const int row = xrec_row(sc, rd, rg, t, l);
    f32x16 wl0, wl1;
    float mt = 0.f, ml = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      // (hidden groups: columns 64 q + (r & 3) + 8 (r >> 2) + 4 h and + 32, the producer tiles' registers r)
      const int ca = xrec_col(sc, rd, r >> 3, 8 * h + (r & 7)), cb = xrec_col(sc, rd, 2 + (r >> 3), 8 * h + (r & 7));
      float a = 0.f, b = 0.f;
      if (row >= 0 && row < L.out_dim) {
        if (ca >= 0 && ca < L.in_dim) a = L.W[(int64_t)row * L.in_dim + ca];
        if (cb >= 0 && cb < L.in_dim) b = L.W[(int64_t)row * L.in_dim + cb];
      }
      const float ah = from_elem<NA_PREC_F16X>(to_elem<NA_PREC_F16X>(a)), bh = from_elem<NA_PREC_F16X>(to_elem<NA_PREC_F16X>(b));
      wl0[r] = a - ah;
      wl1[r] = b - bh;
      mt = fmaxf(mt, fmaxf(fabsf(ah), fabsf(bh)));  // (of the f16 values: that is what the kernel converts)
      ml = fmaxf(ml, fmaxf(fabsf(wl0[r]), fabsf(wl1[r])));
    }
    // block scale 2^(floor(log2 max) - 2): the largest element lands in [4, 8) (saturating at 7.5)
    auto scale_byte = [](float m) { const int ev = (int)(__builtin_bit_cast(uint32_t, m) >> 23); return ev > 3 ? ev - 2 : 1; };
    const int et = scale_byte(mt), el = scale_byte(ml);
    // (by construction here, where a few registers cost nothing: destination disjoint from every operand)
    const x::i32x6 L6 = x::cvt_fp6_disjoint(wl0, wl1, __builtin_bit_cast(float, (uint32_t)el << 23));
    char* rec = dst + kHeaderBytes + kBiasBytes + (int64_t)rg * srg + sc.npair * x::PAIRB + (int64_t)ri * x::REC;
    // {WL6 t0 | WL6 t1}: dword d of the lane's twelve sits in 16-byte part d >> 2 (three lane-linear parts)
#pragma unroll
    for (int d = 0; d < 6; ++d) {
      const int dd = 6 * t + d;
      *(uint32_t*)(rec + 8192 + (dd >> 2) * 1024 + l * 16 + (dd & 3) * 4) = (uint32_t)L6[d];
    }
    uint8_t* scb = (uint8_t*)(rec + 8192 + 3072 + l * 4);
    scb[2 * t] = (uint8_t)el;
    scb[2 * t + 1] = (uint8_t)et;
  }
}
// bias blocks: the layout of pack_ls_kernel ([row group][phase] 1-KiB blocks, floats [slot][hi(2)][16])
__global__ void pack_lsx_bias_kernel(XSched sc, char* __restrict__ dst) {
  const int64_t nbias = 4 * kNPhase * 256;
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < nbias; q += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(q & 255), p = (int)((q >> 8) % kNPhase), rg = (int)((q >> 8) / kNPhase);
    const int slot = k >> 5, hi = (k >> 4) & 1, r = k & 15;
    const int rin = (r & 3) + 8 * (r >> 2) + 4 * hi;
    float v = 0.f;
    if (p < sc.nphase) {
      const XLin L = sc.lin[sc.bias_lin[p]];
      const int mode = sc.bias_mode[p];  // 0 hidden rows, 1 / 3 out Linear with 3 tiles, 2 out Linear, one tile
      if (L.B != nullptr) {
        if (mode == 0) { if (slot < 2) v = L.B[32 * (2 * rg + slot) + rin]; }
        else {
          const int row = slot < ((mode == 1 || mode == 3) ? 3 : 1) ? out_row_map(sc.desc[L.desc], 32 * slot + rin) : -1;
          if (row >= 0 && row < L.out_dim) v = L.B[row];
        }
      }
    }
    *(float*)(dst + kHeaderBytes + ((int64_t)rg * kNPhase + p) * 1024 + k * 4) = v;
  }
}
__global__ void pack_lsx_header_kernel(uint32_t* __restrict__ dst, uint32_t units) {
  if (threadIdx.x == 0) { dst[0] = kMagic; dst[1] = (uint32_t)NA_PREC_F16X; dst[2] = units; dst[3] = kNPhase; }
}

// `nl` Linears of one SkipConnMLP appended to the schedule: init (NI chunk pairs), hidden Linears (skip layers take the NI init
// chunks again, + kHidden), out.  geo: the View MLP's fifth init chunk is its own pair behind the init / skip chunks.
static void xs_add_mlp(XSched& sc, const NaMlpDesc& d, const float* const* w, const float* const* b, int nl, int ni, bool geo,
                       int out_mode, bool init_rec = false) {
  const int di = sc.ndesc++;
  sc.desc[di] = d;
  const int dim_p = d.in_size + d.enc_dims + d.latent_size;
  const int l0 = sc.nlin;
  for (int i = 0; i < nl; ++i) {
    const bool first = i == 0, last = i == nl - 1;
    const bool skip = !first && !last && ((i - 1) % d.skip) == 0 && (i - 1) != d.num_layers - 1;
    XLin L;
    L.W = w[i]; L.B = b[i]; L.desc = di;
    L.in_dim = first ? dim_p : skip ? kHidden + dim_p : kHidden;
    L.out_dim = last ? d.out_size : kHidden;
    sc.lin[sc.nlin++] = L;
    sc.bias_lin[sc.nphase] = (int8_t)(l0 + i);
    sc.bias_mode[sc.nphase++] = (int8_t)(last ? out_mode : 0);
    if (first || skip) {
      // init_rec: the (<= 4) init chunks as ONE record in front of the Linear's hidden records, consumed from the init region
      if (init_rec) sc.rec[sc.nrec++] = XRecD{(int8_t)(l0 + i), 0, 0, (int8_t)(skip ? 2 : 1), 0};
      else for (int q = 0; q < ni; ++q) sc.pair[sc.npair++] = XPairD{(int8_t)(l0 + i), (int8_t)q, (int8_t)(skip ? 1 : 0)};
    }
    if (!first) {
      for (int q = 0; q < 4; ++q) sc.rec[sc.nrec++] = XRecD{(int8_t)(l0 + i), (int8_t)q, (int8_t)(last ? out_mode : 0), 0, 0};
    }
    if ((first || skip) && geo) sc.pair[sc.npair++] = XPairD{(int8_t)(l0 + i), 4, (int8_t)(skip ? 1 : 0)};
  }
}

int render_lsx_pack(int model, const float* const* w0, const float* const* b0, const float* const* w1, const float* const* b1,
                    char* packed, hipStream_t stream, int n_out) {
  XSched sc;
  memset(&sc, 0, sizeof(sc));
  const NaMlpDesc view = {5, NA_ENC_NONE, 0, 64, 4, 256, 3, 3, NA_ACT_SIN, NA_LAYOUT_PLAIN_VIEW};
  if (model == 0) {
    const NaMlpDesc first = {3, NA_ENC_HASH, 35, 0, 4, 256, 65, 3, NA_ACT_LEAKY_RELU, NA_LAYOUT_PLAIN_FIRST};
    xs_add_mlp(sc, first, w0, b0, 6, 3, false, 3, true);
    xs_add_mlp(sc, view, w1, b1, 6, 4, true, 2, true);
  } else if (model == 1) {
    const NaMlpDesc tiny = {3, NA_ENC_NONE, 0, 0, 6, 256, 4, 3, NA_ACT_LEAKY_RELU, NA_LAYOUT_GENERIC};
    xs_add_mlp(sc, tiny, w0, b0, 8, 1, false, 2);
  } else if (model == 2) {
    xs_add_mlp(sc, view, w0, b0, 6, 4, true, 2, true);
  } else if (model == 6) {
    // PlainNeRF(view) + mip.  Column layouts (src/neural_blocks.py:283-287: [p | enc(p) | latent]): first [p 3 | x 3 + hash 32 |
    // IPE 96] (134), skip layer [hidden 256 | the same]; View [x y z elev azim | IPE 96 | intermediate 64] (165)
    // (src/nerf.py:352-358: latent = cat(mip, cat(intermediate, refl_latent))), skip layer [hidden 256 | the same].
    // The slot maps of the [hash | x] and latent groups come from descs WITHOUT the IPE columns; `off` puts them in place.
    const NaMlpDesc first = {3, NA_ENC_HASH, 35, 0, 4, 256, 65, 3, NA_ACT_LEAKY_RELU, NA_LAYOUT_PLAIN_FIRST};
    sc.desc[0] = first; sc.desc[1] = view; sc.ndesc = 2;
    for (int m = 0; m < 2; ++m) {
      const float* const* w = m == 0 ? w0 : w1;
      const float* const* b = m == 0 ? b0 : b1;
      const int dim_p = m == 0 ? 134 : 165, ipe0 = m == 0 ? 38 : 5, grp_off = m == 0 ? 0 : 96;
      for (int i = 0; i < 6; ++i) {  // init, layers.0..3, out
        const bool fst = i == 0, last = i == 5, skip = i == 1;
        XLin L;
        L.W = w[i]; L.B = b[i]; L.desc = m;
        L.in_dim = fst ? dim_p : skip ? kHidden + dim_p : kHidden;
        L.out_dim = last ? (m == 0 ? 65 : 3) : kHidden;
        const int li = sc.nlin;
        sc.lin[sc.nlin++] = L;
        sc.bias_lin[sc.nphase] = (int8_t)li;
        sc.bias_mode[sc.nphase++] = (int8_t)(last ? (m == 0 ? 3 : 2) : 0);
        if (fst || skip) {
          const int so = skip ? kHidden : 0;
          // consumption order: the [hash | x] / latent group (init region), then (skip layers) the four hidden groups, then the IPE groups
          sc.rec[sc.nrec++] = XRecD{(int8_t)li, 0, 0, (int8_t)(skip ? 2 : 1), (int16_t)grp_off};
          if (skip) for (int q = 0; q < 4; ++q) sc.rec[sc.nrec++] = XRecD{(int8_t)li, (int8_t)q, 0, 0, 0};
          for (int q = 0; q < 2; ++q) sc.rec[sc.nrec++] = XRecD{(int8_t)li, (int8_t)q, 0, 5, (int16_t)(so + ipe0)};
          if (m == 1) sc.pair[sc.npair++] = XPairD{(int8_t)li, 4, (int8_t)(skip ? 1 : 0)};
        } else {
          for (int q = 0; q < 4; ++q) sc.rec[sc.nrec++] = XRecD{(int8_t)li, (int8_t)q, (int8_t)(last ? (m == 0 ? 3 : 2) : 0), 0, 0};
        }
      }
    }
  } else if (model == 5) {
    const NaMlpDesc fmlp = {3, NA_ENC_FOURIER, 256, 0, 6, 256, 65, 3, NA_ACT_LEAKY_RELU, NA_LAYOUT_GENERIC};
    sc.desc[sc.ndesc++] = fmlp;
    const int dim_p = 3 + 256;
    for (int i = 0; i < 8; ++i) {  // init, layers.0..5, out
      const bool first = i == 0, last = i == 7, skip = i == 1 || i == 4;
      XLin L;
      L.W = w0[i]; L.B = b0[i]; L.desc = 0;
      L.in_dim = first ? dim_p : skip ? kHidden + dim_p : kHidden;
      L.out_dim = last ? 65 : kHidden;
      sc.lin[sc.nlin++] = L;
      sc.bias_lin[sc.nphase] = (int8_t)i;
      sc.bias_mode[sc.nphase++] = (int8_t)(last ? 1 : 0);
      if (!first) for (int q = 0; q < 4; ++q) sc.rec[sc.nrec++] = XRecD{(int8_t)i, (int8_t)q, (int8_t)(last ? 1 : 0), 0, 0};
      if (first || skip) {
        for (int q = 0; q < 4; ++q) sc.rec[sc.nrec++] = XRecD{(int8_t)i, (int8_t)q, 0, (int8_t)(skip ? 4 : 3), 0};
        sc.pair[sc.npair++] = XPairD{(int8_t)i, 16, (int8_t)(skip ? 1 : 0)};  // the position chunk: init chunk F / 8 of the Fourier layout
      }
    }
  } else if (model == 4) {
    const NaMlpDesc hashmlp = {3, NA_ENC_HASH, 35, 0, 5, 256, n_out, 3, NA_ACT_LEAKY_RELU, NA_LAYOUT_GENERIC};
    xs_add_mlp(sc, hashmlp, w0, b0, 7, 3, false, 2, true);
  } else {
    const NaMlpDesc siren = {3, NA_ENC_NONE, 0, 0, 5, 256, 65, 3, NA_ACT_SIN, NA_LAYOUT_PLAIN_FIRST};
    xs_add_mlp(sc, siren, w0, b0, 7, 1, false, 3);
    xs_add_mlp(sc, view, w1, b1, 6, 4, true, 2, true);
  }
  if (sc.npair != x::npair(model) || sc.nrec != x::nrec(model)) {
    set_error("render_lsx_pack: schedule of model %d has %d pairs / %d records, the kernel expects %d / %d", model, sc.npair,
              sc.nrec, x::npair(model), x::nrec(model));
    return NA_EINVAL;
  }
  hipLaunchKernelGGL(pack_lsx_header_kernel, dim3(1), dim3(64), 0, stream, (uint32_t*)packed, (uint32_t)x::hdr_units(model));
  const int64_t ne = 4ll * sc.npair * 2 * 512 + 4ll * sc.nrec * 8 * 512;
  hipLaunchKernelGGL(pack_lsx_f16_kernel, dim3(grid_for(ne, 256, 4096)), dim3(256), 0, stream, sc, packed);
  hipLaunchKernelGGL(pack_lsx_fp6_kernel, dim3(grid_for(4ll * sc.nrec * 2 * 64, 64, 4096)), dim3(64), 0, stream, sc, packed);
  hipLaunchKernelGGL(pack_lsx_bias_kernel, dim3(grid_for(4 * kNPhase * 256, 256, 4096)), dim3(256), 0, stream, sc, packed);
  return check_launch("na_render_*_ls_pack (f16x)");
}
}  // namespace ls
int render_ls_dispatch_f16x(ls::Args& a, hipStream_t s, int model) {
  return model == 1 ? ls::launch<NA_PREC_F16X, 1>(a, s) : model == 2 ? ls::launch<NA_PREC_F16X, 2>(a, s)
         : model == 3 ? ls::launch<NA_PREC_F16X, 3>(a, s) : model == 4 ? ls::launch<NA_PREC_F16X, 4>(a, s)
         : model == 5 ? ls::launch<NA_PREC_F16X, 5>(a, s) : model == 6 ? ls::launch<NA_PREC_F16X, 6>(a, s)
         : ls::launch<NA_PREC_F16X>(a, s);
}
int render_lsx_pack_hashmlp(const float* const* w, const float* const* b, int n_out, char* packed, hipStream_t stream) {
  return ls::render_lsx_pack(4, w, b, nullptr, nullptr, packed, stream, n_out);
}
int render_lsx_pack_fouriermlp(const float* const* w, const float* const* b, char* packed, hipStream_t stream) {
  return ls::render_lsx_pack(5, w, b, nullptr, nullptr, packed, stream);
}
int render_lsx_pack_mip(const float* const* w0, const float* const* b0, const float* const* w1, const float* const* b1, char* packed,
                        hipStream_t stream) {
  return ls::render_lsx_pack(6, w0, b0, w1, b1, packed, stream);
}
#elif NA_PREC_INST == 0
int render_ls_dispatch_bf16(ls::Args& a, hipStream_t s, int model) {
  return model == 1 ? ls::launch<NA_PREC_BF16, 1>(a, s) : model == 2 ? ls::launch<NA_PREC_BF16, 2>(a, s)
         : model == 3 ? ls::launch<NA_PREC_BF16, 3>(a, s) : ls::launch<NA_PREC_BF16>(a, s);
}
#elif NA_PREC_INST == 1
int render_ls_dispatch_bf16x3(ls::Args& a, hipStream_t s, int model) {
  return model == 1 ? ls::launch<NA_PREC_BF16X3, 1>(a, s) : model == 2 ? ls::launch<NA_PREC_BF16X3, 2>(a, s)
         : model == 3 ? ls::launch<NA_PREC_BF16X3, 3>(a, s) : ls::launch<NA_PREC_BF16X3>(a, s);
}
#elif NA_PREC_INST == 2
int render_ls_dispatch_f16(ls::Args& a, hipStream_t s, int model) {
  return model == 1 ? ls::launch<NA_PREC_F16, 1>(a, s) : model == 2 ? ls::launch<NA_PREC_F16, 2>(a, s)
         : model == 3 ? ls::launch<NA_PREC_F16, 3>(a, s) : ls::launch<NA_PREC_F16>(a, s);
}
#endif
int render_ls_dispatch_bf16(ls::Args& a, hipStream_t s, int model);
int render_ls_dispatch_bf16x3(ls::Args& a, hipStream_t s, int model);
int render_ls_dispatch_f16(ls::Args& a, hipStream_t s, int model);
int render_ls_dispatch_f16x(ls::Args& a, hipStream_t s, int model);
int render_lsx_pack_hashmlp(const float* const* w, const float* const* b, int n_out, char* packed, hipStream_t stream);
int render_lsx_pack_fouriermlp(const float* const* w, const float* const* b, char* packed, hipStream_t stream);
int render_lsx_pack_mip(const float* const* w0, const float* const* b0, const float* const* w1, const float* const* b1, char* packed,
                        hipStream_t stream);

}  // namespace na

#if NA_PREC_INST == 0
using namespace na;

extern "C" size_t na_render_ls_packed_bytes(int precision) {
  if (precision != NA_PREC_BF16 && precision != NA_PREC_BF16X3 && precision != NA_PREC_F16 && precision != NA_PREC_F16X) return 0;
  return ls::packed_bytes(precision);
}

extern "C" int na_render_ls_pack(int precision, const float* const* w_first, const float* const* b_first,
                                 const float* const* w_view, const float* const* b_view, void* packed, void* stream) {
  NA_REQUIRE(w_first && b_first && w_view && b_view && packed, NA_ENULL, "na_render_ls_pack: null pointer");
  NA_REQUIRE(precision == NA_PREC_BF16 || precision == NA_PREC_BF16X3 || precision == NA_PREC_F16 || precision == NA_PREC_F16X,
             NA_EUNSUPPORTED, "na_render_ls_pack: precision %d", precision);
  if (precision == NA_PREC_F16X) {
    for (int i = 0; i < 6; ++i) NA_REQUIRE(w_first[i] && w_view[i], NA_ENULL, "na_render_ls_pack: weights[%d] is null", i);
    return ls::render_lsx_pack(0, w_first, b_first, w_view, b_view, (char*)packed, (hipStream_t)stream);
  }
  ls::PackArgs w;
  for (int i = 0; i < 6; ++i) {
    NA_REQUIRE(w_first[i] && w_view[i], NA_ENULL, "na_render_ls_pack: weights[%d] is null", i);
    w.w_first[i] = w_first[i]; w.b_first[i] = b_first[i];
    w.w_view[i] = w_view[i]; w.b_view[i] = b_view[i];
  }
  const int planes = precision == NA_PREC_BF16X3 ? 2 : 1;
  hipLaunchKernelGGL(ls::pack_ls_header_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ls::kMagic, (uint32_t)precision,
                     (uint32_t*)packed, (uint32_t)ls::kPairsPerPass);
  const int64_t total = 4 * 2 * (int64_t)ls::kPairsPerPass * 512 + 4 * ls::kNPhase * 256;
  hipLaunchKernelGGL(ls::pack_ls_kernel, dim3(grid_for(total, 256, 4096)), dim3(256), 0, (hipStream_t)stream, w, planes,
                     precision == NA_PREC_F16 ? 1 : 0, (char*)packed);
  return check_launch("na_render_ls_pack");
}

extern "C" size_t na_render_ls_workspace_bytes(int T, int64_t R) {
  if (T < 1 || R < 0) return 0;
  const int64_t nb = (T + 31) / 32;
  (void)nb;
  return (size_t)R * 2 * sizeof(float) + 256 + (NA_LS_TRACE ? 4096 + 256 : 0);
}

static int render_plain_view_ls_impl(const float* rays, const float* pts, int64_t R, const float* ts, int64_t ts_stride, int T,
                                     const float* hash_tables, const void* packed, int precision, int sigmoid_kind,
                                     int bg_kind, float* alpha, float* weights, float* out, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  NA_REQUIRE(T >= 1 && R >= 0, NA_EINVAL, "na_render_plain_view_ls: bad shape T=%d R=%lld", T, (long long)R);
  if (R == 0) return NA_OK;  // empty batch: a no-op before any pointer check (zero-size tensors carry null pointers)
  NA_REQUIRE(rays && ts && hash_tables && packed && out && workspace, NA_ENULL, "na_render_plain_view_ls: null pointer");
  NA_REQUIRE(precision == NA_PREC_BF16 || precision == NA_PREC_BF16X3 || precision == NA_PREC_F16 || precision == NA_PREC_F16X,
             NA_EUNSUPPORTED, "na_render_plain_view_ls: precision %d", precision);
  NA_REQUIRE(sigmoid_kind >= 0 && sigmoid_kind <= NA_SIG_IDENTITY, NA_EUNSUPPORTED, "na_render_plain_view_ls: sigmoid %d",
             sigmoid_kind);
  NA_REQUIRE(bg_kind == NA_BG_BLACK || bg_kind == NA_BG_WHITE, NA_EUNSUPPORTED, "na_render_plain_view_ls: bg %d", bg_kind);
  NA_REQUIRE(workspace_bytes >= na_render_ls_workspace_bytes(T, R), NA_EWORKSPACE,
             "na_render_plain_view_ls: workspace %zu < %zu bytes", workspace_bytes, na_render_ls_workspace_bytes(T, R));
  if (R == 0) return NA_OK;
  ls::Args a;
  a.rays = rays; a.ts = ts; a.ts_stride = ts_stride; a.pts = pts; a.tables = (const float4*)hash_tables;
  a.feat = nullptr; a.beta = nullptr; a.feat_ld = 0;
  a.packed = (const char*)packed; a.packed_size = (uint32_t)ls::packed_bytes(precision);
  a.alpha = alpha; a.weights = weights; a.out = out; a.bg_kind = bg_kind;
  a.R = R; a.T = T; a.nb = (T + 31) / 32;
  float* elaz = (float*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  a.elaz = elaz;
  hipLaunchKernelGGL(ls::ray_elaz_kernel, dim3(grid_for(R, 256, 4096)), dim3(256), 0, (hipStream_t)stream, rays, R, elaz);
  a.sigmoid_kind = sigmoid_kind;
  a.res = hash_resolutions();
  a.trace = NA_LS_TRACE ? (unsigned long long*)(((uintptr_t)(elaz + R * 2) + 255) & ~(uintptr_t)255) : nullptr;
  if (precision == NA_PREC_BF16) return render_ls_dispatch_bf16(a, (hipStream_t)stream, 0);
  if (precision == NA_PREC_F16) return render_ls_dispatch_f16(a, (hipStream_t)stream, 0);
  if (precision == NA_PREC_F16X) return render_ls_dispatch_f16x(a, (hipStream_t)stream, 0);
  return render_ls_dispatch_bf16x3(a, (hipStream_t)stream, 0);
}

extern "C" int na_render_plain_view_ls(const float* rays, const float* pts, int64_t R, const float* ts, int T,
                                       const float* hash_tables, const void* packed, int precision, int sigmoid_kind,
                                       int bg_kind, float* alpha, float* weights, float* out, void* workspace,
                                       size_t workspace_bytes, void* stream) {
  return render_plain_view_ls_impl(rays, pts, R, ts, 0, T, hash_tables, packed, precision, sigmoid_kind, bg_kind, alpha, weights, out,
                                   workspace, workspace_bytes, stream);
}

// The same launch with PER-RAY steps ts_ray[R,T] (row r = the increasing sample distances of ray r): the fine pass of coarse ->
// fine rendering, whose steps come from na_resample_ts.  Positions are o + t d, interval lengths t[i + 1] - t[i] of the ray's own row.
extern "C" int na_render_plain_view_ls_rayts(const float* rays, int64_t R, const float* ts_ray, int T, const float* hash_tables,
                                             const void* packed, int precision, int sigmoid_kind, int bg_kind, float* alpha,
                                             float* weights, float* out, void* workspace, size_t workspace_bytes, void* stream) {
  return render_plain_view_ls_impl(rays, nullptr, R, ts_ray, T, T, hash_tables, packed, precision, sigmoid_kind, bg_kind, alpha, weights,
                                   out, workspace, workspace_bytes, stream);
}

// ---- PlainNeRF(view) + mip (config 3; src/nerf.py:256-261, 326-361, src/utils.py:23-140) as ONE launch, NA_PREC_F16X only:
// the 96 IPE features of every sample are generated in the kernel (MODEL 6) for the init and skip Linears of both MLPs
extern "C" size_t na_render_plain_mip_ls_packed_bytes(int precision) {
  return precision == NA_PREC_F16X ? ls::packed_bytes_x(6) : 0;
}

extern "C" int na_render_plain_mip_ls_pack(int precision, const float* const* w_first, const float* const* b_first,
                                           const float* const* w_view, const float* const* b_view, void* packed, void* stream) {
  NA_REQUIRE(w_first && b_first && w_view && b_view && packed, NA_ENULL, "na_render_plain_mip_ls_pack: null pointer");
  NA_REQUIRE(precision == NA_PREC_F16X, NA_EUNSUPPORTED, "na_render_plain_mip_ls_pack: precision %d (f16x only)", precision);
  for (int i = 0; i < 6; ++i)
    NA_REQUIRE(w_first[i] && w_view[i] && b_first[i] && b_view[i], NA_ENULL, "na_render_plain_mip_ls_pack: Linear %d is null", i);
  return render_lsx_pack_mip(w_first, b_first, w_view, b_view, (char*)packed, (hipStream_t)stream);
}

extern "C" int na_render_plain_mip_ls(const float* rays, int B, int H, int W, const float* ts, int T, const float* hash_tables,
                                      const void* packed, int precision, int mip_kind, int min_deg, int max_deg, float t_end,
                                      int sigmoid_kind, int bg_kind, float* alpha, float* weights, float* out, void* workspace,
                                      size_t workspace_bytes, void* stream) {
  NA_REQUIRE(T >= 1 && B >= 0 && H >= 0 && W >= 0, NA_EINVAL, "na_render_plain_mip_ls: bad shape T=%d B=%d H=%d W=%d", T, B, H, W);
  const int64_t R = (int64_t)B * H * W;
  if (R == 0) return NA_OK;
  NA_REQUIRE(rays && ts && hash_tables && packed && out && workspace, NA_ENULL, "na_render_plain_mip_ls: null pointer");
  NA_REQUIRE(precision == NA_PREC_F16X, NA_EUNSUPPORTED, "na_render_plain_mip_ls: precision %d (f16x only)", precision);
  NA_REQUIRE(mip_kind == 0 || mip_kind == 1, NA_EUNSUPPORTED, "na_render_plain_mip_ls: mip kind %d", mip_kind);
  NA_REQUIRE(max_deg - min_deg == 16, NA_EUNSUPPORTED, "na_render_plain_mip_ls: %d IPE degrees (the schedule is built for 16)",
             max_deg - min_deg);
  NA_REQUIRE(H >= 2, NA_EINVAL, "na_render_plain_mip_ls: pixel radii need H >= 2 rows per crop (H=%d)", H);
  NA_REQUIRE(sigmoid_kind >= 0 && sigmoid_kind <= NA_SIG_IDENTITY, NA_EUNSUPPORTED, "na_render_plain_mip_ls: sigmoid %d", sigmoid_kind);
  NA_REQUIRE(bg_kind == NA_BG_BLACK || bg_kind == NA_BG_WHITE, NA_EUNSUPPORTED, "na_render_plain_mip_ls: bg %d", bg_kind);
  NA_REQUIRE(workspace_bytes >= na_render_ls_workspace_bytes(T, R), NA_EWORKSPACE,
             "na_render_plain_mip_ls: workspace %zu < %zu bytes", workspace_bytes, na_render_ls_workspace_bytes(T, R));
  ls::Args a;
  a.rays = rays; a.ts = ts; a.pts = nullptr; a.tables = (const float4*)hash_tables;
  a.feat = nullptr; a.beta = nullptr; a.feat_ld = 0;
  a.packed = (const char*)packed; a.packed_size = (uint32_t)ls::packed_bytes_x(6);
  a.alpha = alpha; a.weights = weights; a.out = out; a.bg_kind = bg_kind;
  a.R = R; a.T = T; a.nb = (T + 31) / 32;
  a.mip_H = H; a.mip_W = W; a.mip_kind = mip_kind; a.mip_min_deg = min_deg; a.mip_nd = max_deg - min_deg; a.mip_t_end = t_end;
  float* elaz = (float*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  a.elaz = elaz;
  hipLaunchKernelGGL(ls::ray_elaz_kernel, dim3(grid_for(R, 256, 4096)), dim3(256), 0, (hipStream_t)stream, rays, R, elaz);
  a.sigmoid_kind = sigmoid_kind;
  a.res = hash_resolutions();
  a.trace = NA_LS_TRACE ? (unsigned long long*)(((uintptr_t)(elaz + R * 2) + 255) & ~(uintptr_t)255) : nullptr;
  return render_ls_dispatch_f16x(a, (hipStream_t)stream, 6);
}

// ---- TinyNeRF on the same engine (src/nerf.py:278-305)
extern "C" size_t na_render_tiny_ls_packed_bytes(int precision) {
  if (precision != NA_PREC_BF16 && precision != NA_PREC_BF16X3 && precision != NA_PREC_F16 && precision != NA_PREC_F16X) return 0;
  return ls::packed_bytes(precision, ls::kTinyPairs);
}

extern "C" int na_render_tiny_ls_pack(int precision, const float* const* w, const float* const* b, void* packed, void* stream) {
  NA_REQUIRE(w && b && packed, NA_ENULL, "na_render_tiny_ls_pack: null pointer");
  NA_REQUIRE(precision == NA_PREC_BF16 || precision == NA_PREC_BF16X3 || precision == NA_PREC_F16 || precision == NA_PREC_F16X, NA_EUNSUPPORTED,
             "na_render_tiny_ls_pack: precision %d", precision);
  ls::TinyPackArgs pa;
  for (int i = 0; i < 8; ++i) {
    NA_REQUIRE(w[i], NA_ENULL, "na_render_tiny_ls_pack: weights[%d] is null", i);
    pa.w[i] = w[i]; pa.b[i] = b[i];
  }
  if (precision == NA_PREC_F16X) return ls::render_lsx_pack(1, w, b, nullptr, nullptr, (char*)packed, (hipStream_t)stream);
  const int planes = precision == NA_PREC_BF16X3 ? 2 : 1;
  hipLaunchKernelGGL(ls::pack_ls_header_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ls::kMagic, (uint32_t)precision,
                     (uint32_t*)packed, (uint32_t)ls::kTinyPairs);
  const int64_t total = 4 * 2 * (int64_t)ls::kTinyPairs * 512 + 4 * ls::kNPhase * 256;
  hipLaunchKernelGGL(ls::pack_ls_tiny_kernel, dim3(grid_for(total, 256, 4096)), dim3(256), 0, (hipStream_t)stream, pa, planes,
                     precision == NA_PREC_F16 ? 1 : 0, (char*)packed);
  return check_launch("na_render_tiny_ls_pack");
}

extern "C" int na_render_tiny_ls(const float* rays, const float* pts, int64_t R, const float* ts, int T, const void* packed,
                                 int precision, int sigmoid_kind, int bg_kind, float* alpha, float* weights, float* out,
                                 void* stream) {
  NA_REQUIRE(T >= 1 && R >= 0, NA_EINVAL, "na_render_tiny_ls: bad shape T=%d R=%lld", T, (long long)R);
  if (R == 0) return NA_OK;
  NA_REQUIRE(rays && ts && packed && out, NA_ENULL, "na_render_tiny_ls: null pointer");
  NA_REQUIRE(precision == NA_PREC_BF16 || precision == NA_PREC_BF16X3 || precision == NA_PREC_F16 || precision == NA_PREC_F16X, NA_EUNSUPPORTED,
             "na_render_tiny_ls: precision %d", precision);
  NA_REQUIRE(sigmoid_kind >= 0 && sigmoid_kind <= NA_SIG_IDENTITY, NA_EUNSUPPORTED, "na_render_tiny_ls: sigmoid %d", sigmoid_kind);
  NA_REQUIRE(bg_kind == NA_BG_BLACK || bg_kind == NA_BG_WHITE, NA_EUNSUPPORTED, "na_render_tiny_ls: bg %d", bg_kind);
  ls::Args a;
  a.rays = rays; a.ts = ts; a.pts = pts; a.tables = nullptr;
  a.feat = nullptr; a.beta = nullptr; a.feat_ld = 0;
  a.packed = (const char*)packed; a.packed_size = (uint32_t)ls::packed_bytes(precision, ls::kTinyPairs);
  a.alpha = alpha; a.weights = weights; a.out = out; a.bg_kind = bg_kind;
  a.R = R; a.T = T; a.nb = (T + 31) / 32;
  a.elaz = nullptr;
  a.sigmoid_kind = sigmoid_kind;
  a.res = hash_resolutions();
  a.trace = nullptr;
  if (precision == NA_PREC_BF16) return render_ls_dispatch_bf16(a, (hipStream_t)stream, 1);
  if (precision == NA_PREC_F16) return render_ls_dispatch_f16(a, (hipStream_t)stream, 1);
  if (precision == NA_PREC_F16X) return render_ls_dispatch_f16x(a, (hipStream_t)stream, 1);
  return render_ls_dispatch_bf16x3(a, (hipStream_t)stream, 1);
}

// ---- View head + compositing on per-sample (density source, latent) rows: VolSDF's second half (src/nerf.py:981-1013)
extern "C" size_t na_render_view_ls_packed_bytes(int precision) {
  if (precision != NA_PREC_BF16 && precision != NA_PREC_BF16X3 && precision != NA_PREC_F16 && precision != NA_PREC_F16X) return 0;
  return ls::packed_bytes(precision, ls::kViewPairs);
}

extern "C" int na_render_view_ls_pack(int precision, const float* const* w, const float* const* b, void* packed, void* stream) {
  NA_REQUIRE(w && b && packed, NA_ENULL, "na_render_view_ls_pack: null pointer");
  NA_REQUIRE(precision == NA_PREC_BF16 || precision == NA_PREC_BF16X3 || precision == NA_PREC_F16 || precision == NA_PREC_F16X, NA_EUNSUPPORTED,
             "na_render_view_ls_pack: precision %d", precision);
  ls::ViewPackArgs pa;
  for (int i = 0; i < 6; ++i) {
    NA_REQUIRE(w[i], NA_ENULL, "na_render_view_ls_pack: weights[%d] is null", i);
    pa.w[i] = w[i]; pa.b[i] = b[i];
  }
  if (precision == NA_PREC_F16X) return ls::render_lsx_pack(2, w, b, nullptr, nullptr, (char*)packed, (hipStream_t)stream);
  const int planes = precision == NA_PREC_BF16X3 ? 2 : 1;
  hipLaunchKernelGGL(ls::pack_ls_header_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ls::kMagic, (uint32_t)precision,
                     (uint32_t*)packed, (uint32_t)ls::kViewPairs);
  const int64_t total = 4 * 2 * (int64_t)ls::kViewPairs * 512 + 4 * ls::kNPhase * 256;
  hipLaunchKernelGGL(ls::pack_ls_view_kernel, dim3(grid_for(total, 256, 4096)), dim3(256), 0, (hipStream_t)stream, pa, planes,
                     precision == NA_PREC_F16 ? 1 : 0, (char*)packed);
  return check_launch("na_render_view_ls_pack");
}

extern "C" int na_render_view_ls(const float* rays, const float* pts, int64_t R, const float* ts, int T, const float* feat,
                                 int feat_ld, const float* beta, const void* packed, int precision, int sigmoid_kind,
                                 int bg_kind, float* alpha, float* weights, float* out, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  NA_REQUIRE(T >= 1 && R >= 0, NA_EINVAL, "na_render_view_ls: bad shape T=%d R=%lld", T, (long long)R);
  if (R == 0) return NA_OK;
  NA_REQUIRE(rays && ts && feat && beta && packed && out && workspace, NA_ENULL, "na_render_view_ls: null pointer");
  NA_REQUIRE(feat_ld >= 65, NA_EINVAL, "na_render_view_ls: feat_ld %d < 65 (signed distance + 64 latent columns)", feat_ld);
  NA_REQUIRE(precision == NA_PREC_BF16 || precision == NA_PREC_BF16X3 || precision == NA_PREC_F16 || precision == NA_PREC_F16X, NA_EUNSUPPORTED,
             "na_render_view_ls: precision %d", precision);
  NA_REQUIRE(sigmoid_kind >= 0 && sigmoid_kind <= NA_SIG_IDENTITY, NA_EUNSUPPORTED, "na_render_view_ls: sigmoid %d", sigmoid_kind);
  NA_REQUIRE(bg_kind == NA_BG_BLACK || bg_kind == NA_BG_WHITE, NA_EUNSUPPORTED, "na_render_view_ls: bg %d", bg_kind);
  NA_REQUIRE(workspace_bytes >= na_render_ls_workspace_bytes(T, R), NA_EWORKSPACE, "na_render_view_ls: workspace %zu < %zu bytes",
             workspace_bytes, na_render_ls_workspace_bytes(T, R));
  ls::Args a;
  a.rays = rays; a.ts = ts; a.pts = pts; a.tables = nullptr;
  a.packed = (const char*)packed; a.packed_size = (uint32_t)ls::packed_bytes(precision, ls::kViewPairs);
  a.alpha = alpha; a.weights = weights; a.out = out; a.bg_kind = bg_kind;
  a.R = R; a.T = T; a.nb = (T + 31) / 32;
  a.feat = feat; a.feat_ld = feat_ld; a.beta = beta;
  float* elaz = (float*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  a.elaz = elaz;
  hipLaunchKernelGGL(ls::ray_elaz_kernel, dim3(grid_for(R, 256, 4096)), dim3(256), 0, (hipStream_t)stream, rays, R, elaz);
  a.sigmoid_kind = sigmoid_kind;
  a.res = hash_resolutions();
  a.trace = nullptr;
  if (precision == NA_PREC_BF16) return render_ls_dispatch_bf16(a, (hipStream_t)stream, 2);
  if (precision == NA_PREC_F16) return render_ls_dispatch_f16(a, (hipStream_t)stream, 2);
  if (precision == NA_PREC_F16X) return render_ls_dispatch_f16x(a, (hipStream_t)stream, 2);
  return render_ls_dispatch_bf16x3(a, (hipStream_t)stream, 2);
}

// ---- VolSDF with the SIREN SDF network as one kernel (src/sdf.py:278-287 + src/nerf.py:981-1013)
extern "C" size_t na_render_volsdf_siren_ls_packed_bytes(int precision) {
  if (precision != NA_PREC_BF16 && precision != NA_PREC_BF16X3 && precision != NA_PREC_F16 && precision != NA_PREC_F16X) return 0;
  return ls::packed_bytes(precision, ls::kSirenPairs);
}

extern "C" int na_render_volsdf_siren_ls_pack(int precision, const float* const* w_sdf, const float* const* b_sdf,
                                              const float* const* w_view, const float* const* b_view, void* packed, void* stream) {
  NA_REQUIRE(w_sdf && b_sdf && w_view && b_view && packed, NA_ENULL, "na_render_volsdf_siren_ls_pack: null pointer");
  NA_REQUIRE(precision == NA_PREC_BF16 || precision == NA_PREC_BF16X3 || precision == NA_PREC_F16 || precision == NA_PREC_F16X, NA_EUNSUPPORTED,
             "na_render_volsdf_siren_ls_pack: precision %d", precision);
  ls::SirenPackArgs pa;
  for (int i = 0; i < 7; ++i) {
    NA_REQUIRE(w_sdf[i], NA_ENULL, "na_render_volsdf_siren_ls_pack: sdf weights[%d] is null", i);
    pa.ws[i] = w_sdf[i]; pa.bs[i] = b_sdf[i];
  }
  for (int i = 0; i < 6; ++i) {
    NA_REQUIRE(w_view[i], NA_ENULL, "na_render_volsdf_siren_ls_pack: view weights[%d] is null", i);
    pa.wv[i] = w_view[i]; pa.bv[i] = b_view[i];
  }
  if (precision == NA_PREC_F16X) return ls::render_lsx_pack(3, w_sdf, b_sdf, w_view, b_view, (char*)packed, (hipStream_t)stream);
  const int planes = precision == NA_PREC_BF16X3 ? 2 : 1;
  hipLaunchKernelGGL(ls::pack_ls_header_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ls::kMagic, (uint32_t)precision,
                     (uint32_t*)packed, (uint32_t)ls::kSirenPairs);
  const int64_t total = 4 * 2 * (int64_t)ls::kSirenPairs * 512 + 4 * ls::kNPhase * 256;
  hipLaunchKernelGGL(ls::pack_ls_siren_kernel, dim3(grid_for(total, 256, 4096)), dim3(256), 0, (hipStream_t)stream, pa, planes,
                     precision == NA_PREC_F16 ? 1 : 0, (char*)packed);
  return check_launch("na_render_volsdf_siren_ls_pack");
}

extern "C" int na_render_volsdf_siren_ls(const float* rays, const float* pts, int64_t R, const float* ts, int T, const float* beta,
                                         const void* packed, int precision, int sigmoid_kind, int bg_kind, float* alpha,
                                         float* weights, float* out, void* workspace, size_t workspace_bytes, void* stream) {
  NA_REQUIRE(T >= 1 && R >= 0, NA_EINVAL, "na_render_volsdf_siren_ls: bad shape T=%d R=%lld", T, (long long)R);
  if (R == 0) return NA_OK;
  NA_REQUIRE(rays && ts && beta && packed && out && workspace, NA_ENULL, "na_render_volsdf_siren_ls: null pointer");
  NA_REQUIRE(precision == NA_PREC_BF16 || precision == NA_PREC_BF16X3 || precision == NA_PREC_F16 || precision == NA_PREC_F16X, NA_EUNSUPPORTED,
             "na_render_volsdf_siren_ls: precision %d", precision);
  NA_REQUIRE(sigmoid_kind >= 0 && sigmoid_kind <= NA_SIG_IDENTITY, NA_EUNSUPPORTED, "na_render_volsdf_siren_ls: sigmoid %d", sigmoid_kind);
  NA_REQUIRE(bg_kind == NA_BG_BLACK || bg_kind == NA_BG_WHITE, NA_EUNSUPPORTED, "na_render_volsdf_siren_ls: bg %d", bg_kind);
  NA_REQUIRE(workspace_bytes >= na_render_ls_workspace_bytes(T, R), NA_EWORKSPACE,
             "na_render_volsdf_siren_ls: workspace %zu < %zu bytes", workspace_bytes, na_render_ls_workspace_bytes(T, R));
  ls::Args a;
  a.rays = rays; a.ts = ts; a.pts = pts; a.tables = nullptr;
  a.packed = (const char*)packed; a.packed_size = (uint32_t)ls::packed_bytes(precision, ls::kSirenPairs);
  a.alpha = alpha; a.weights = weights; a.out = out; a.bg_kind = bg_kind;
  a.R = R; a.T = T; a.nb = (T + 31) / 32;
  a.feat = nullptr; a.feat_ld = 0; a.beta = beta;
  float* elaz = (float*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  a.elaz = elaz;
  hipLaunchKernelGGL(ls::ray_elaz_kernel, dim3(grid_for(R, 256, 4096)), dim3(256), 0, (hipStream_t)stream, rays, R, elaz);
  a.sigmoid_kind = sigmoid_kind;
  a.res = hash_resolutions();
  a.trace = nullptr;
  if (precision == NA_PREC_BF16) return render_ls_dispatch_bf16(a, (hipStream_t)stream, 3);
  if (precision == NA_PREC_F16) return render_ls_dispatch_f16(a, (hipStream_t)stream, 3);
  if (precision == NA_PREC_F16X) return render_ls_dispatch_f16x(a, (hipStream_t)stream, 3);
  return render_ls_dispatch_bf16x3(a, (hipStream_t)stream, 3);
}
// ---- a hash-encoded SkipConnMLP on the layer-synchronous engine, rows to HBM (D-NeRF's deformation network, src/nerf.py:1250-1257,
// 1267-1270): NA_PREC_F16X only -- the other precisions run it through na_mlp_forward
extern "C" size_t na_mlp_hash_ls_packed_bytes(int precision) {
  return precision == NA_PREC_F16X ? ls::packed_bytes_x(4) : 0;
}

extern "C" int na_mlp_hash_ls_pack(int precision, const float* const* w, const float* const* b, int n_out, void* packed, void* stream) {
  NA_REQUIRE(w && b && packed, NA_ENULL, "na_mlp_hash_ls_pack: null pointer");
  NA_REQUIRE(precision == NA_PREC_F16X, NA_EUNSUPPORTED, "na_mlp_hash_ls_pack: precision %d (f16x only)", precision);
  NA_REQUIRE(n_out >= 1 && n_out <= 32, NA_EUNSUPPORTED, "na_mlp_hash_ls_pack: n_out %d (1..32: one output tile)", n_out);
  for (int i = 0; i < 7; ++i) NA_REQUIRE(w[i], NA_ENULL, "na_mlp_hash_ls_pack: weights[%d] is null", i);
  return render_lsx_pack_hashmlp(w, b, n_out, (char*)packed, (hipStream_t)stream);
}

extern "C" int na_mlp_hash_ls(const float* rays, const float* pts, int64_t R, const float* ts, int T, const float* hash_tables,
                              const void* packed, int precision, int n_out, float* y, int64_t y_ld, void* stream) {
  NA_REQUIRE(T >= 1 && R >= 0, NA_EINVAL, "na_mlp_hash_ls: bad shape T=%d R=%lld", T, (long long)R);
  if (R == 0) return NA_OK;
  NA_REQUIRE(rays && ts && hash_tables && packed && y, NA_ENULL, "na_mlp_hash_ls: null pointer");
  NA_REQUIRE(precision == NA_PREC_F16X, NA_EUNSUPPORTED, "na_mlp_hash_ls: precision %d (f16x only)", precision);
  NA_REQUIRE(n_out >= 1 && n_out <= 32 && y_ld >= n_out && y_ld < (1 << 20), NA_EINVAL, "na_mlp_hash_ls: n_out %d, y_ld %lld", n_out,
             (long long)y_ld);
  ls::Args a;
  a.rays = rays; a.ts = ts; a.pts = pts; a.tables = (const float4*)hash_tables;
  a.feat = nullptr; a.beta = nullptr; a.feat_ld = 0;
  a.packed = (const char*)packed; a.packed_size = (uint32_t)ls::packed_bytes_x(4);
  a.alpha = nullptr; a.weights = nullptr; a.out = nullptr; a.bg_kind = NA_BG_BLACK;
  a.R = R; a.T = T; a.nb = (T + 31) / 32;
  a.elaz = nullptr; a.sigmoid_kind = 0;
  a.res = hash_resolutions();
  a.trace = nullptr;
  a.y = y; a.y_ld = (int)y_ld; a.n_out = n_out;
  return render_ls_dispatch_f16x(a, (hipStream_t)stream, 4);
}
// ---- a Fourier-encoded SkipConnMLP on the layer-synchronous engine, rows to HBM (VolSDF's MLP SDF network, src/sdf.py:250-258):
// NA_PREC_F16X only
extern "C" size_t na_mlp_fourier_ls_packed_bytes(int precision) {
  return precision == NA_PREC_F16X ? ls::packed_bytes_x(5) : 0;
}

extern "C" int na_mlp_fourier_ls_pack(int precision, const float* const* w, const float* const* b, void* packed, void* stream) {
  NA_REQUIRE(w && b && packed, NA_ENULL, "na_mlp_fourier_ls_pack: null pointer");
  NA_REQUIRE(precision == NA_PREC_F16X, NA_EUNSUPPORTED, "na_mlp_fourier_ls_pack: precision %d (f16x only)", precision);
  for (int i = 0; i < 8; ++i) NA_REQUIRE(w[i], NA_ENULL, "na_mlp_fourier_ls_pack: weights[%d] is null", i);
  return render_lsx_pack_fouriermlp(w, b, (char*)packed, (hipStream_t)stream);
}

extern "C" int na_mlp_fourier_ls(const float* rays, const float* pts, int64_t R, const float* ts, int T, const float* basis,
                                 const void* packed, int precision, float* y, int64_t y_ld, void* stream) {
  NA_REQUIRE(T >= 1 && R >= 0, NA_EINVAL, "na_mlp_fourier_ls: bad shape T=%d R=%lld", T, (long long)R);
  if (R == 0) return NA_OK;
  NA_REQUIRE(rays && ts && basis && packed && y, NA_ENULL, "na_mlp_fourier_ls: null pointer");
  NA_REQUIRE(precision == NA_PREC_F16X, NA_EUNSUPPORTED, "na_mlp_fourier_ls: precision %d (f16x only)", precision);
  NA_REQUIRE(y_ld >= 65 && y_ld < (1 << 20), NA_EINVAL, "na_mlp_fourier_ls: y_ld %lld < 65", (long long)y_ld);
  NA_REQUIRE(((uintptr_t)basis & 15) == 0, NA_EINVAL, "na_mlp_fourier_ls: basis must be 16-byte aligned");
  ls::Args a;
  a.rays = rays; a.ts = ts; a.pts = pts; a.tables = (const float4*)basis;
  a.feat = nullptr; a.beta = nullptr; a.feat_ld = 0;
  a.packed = (const char*)packed; a.packed_size = (uint32_t)ls::packed_bytes_x(5);
  a.alpha = nullptr; a.weights = nullptr; a.out = nullptr; a.bg_kind = NA_BG_BLACK;
  a.R = R; a.T = T; a.nb = (T + 31) / 32;
  a.elaz = rays;  // (the group-wide ray table also fetches two floats per ray from here: any readable 8 R bytes)
  a.sigmoid_kind = 0;
  a.res = hash_resolutions();
  a.trace = nullptr;
  a.y = y; a.y_ld = (int)y_ld; a.n_out = 65;
  return render_ls_dispatch_f16x(a, (hipStream_t)stream, 5);
}
#endif
