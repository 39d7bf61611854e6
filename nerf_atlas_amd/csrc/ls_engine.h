// Building blocks of the layer-synchronous engine (csrc/render_ls.hip): configuration per precision, the kernel's argument
// block, fragment reads / writes, the register weight ring, the MFMA phases, the activation epilogues, the NA_PREC_F16X
// arithmetic (namespace x), the in-wave scans.  The kernel itself: ls_kernel.h; its schedules: ls_sched_*.inc.
#pragma once
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
// A hash-encoded SkipConnMLP alone, rows to HBM (MODEL 4: D-NeRF's deformation network) in the bf16 / bf16x3 / f16 formats (round 6;
// the f16x stream is ls_xsched.h's): init (3 chunks [hash | x]), L0 (3 skip + 16), L1, L2, L3 (3 + 16), L4, out (ONE 32-row tile per
// wave: row groups 0, 1 hold rows 0..31 for their block; row groups 2, 3 -- which own no block in the two-plane formats and would
// only shadow their partner -- rows 32..63 for block rg - 2: up to 64 output rows, 3 n + 1 + (n refl_latent + 1) = 38 for
// `make dnerf`) + 3 zero pairs that keep a pass a multiple of the ring depth
constexpr int kHashMlpPhases = 7;
__host__ __device__ constexpr int hashmlp_phase_pairs(int p) { return p == 0 ? 3 : (p == 1 || p == 4) ? 19 : p == 6 ? 8 + 3 : 16; }
constexpr int kHashMlpPairs = 100;
constexpr int kHeaderBytes = 1024;
constexpr int kBiasBytes = 4 * kNPhase * 1024;  // [row group][phase] 1-KiB blocks: floats [slot][hi(2)][16]
constexpr uint32_t kMagic = 0x4C533032u;        // "LS02"
constexpr int kPartialFloats = 8;

// ---- NA_PREC_F16X (PlainNeRF schedule only): hidden activations and hidden-layer weights in the f16 + 2 x MX-fp6 format
// (the measured prototype of this data flow, tools/proto/ls_mlp_f16x.hip, left the tree in round 6: git history up to 5f9153f; its logs are profiles/r03/f16x_proto_*.log).
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
// MODEL 7 (round 6): PlainNeRF + the Positional head (`make original`, src/refl.py:230-245: a second hash-encoded SkipConnMLP 3 -> 5 x 256,
// skip 3, latent 64 -> 3): MODEL 0's 22 records of `first`, then pos.init 2 ([hash' | x] group, latent group) | L0 5 + 1 | L1 4 | L2 4 |
// L3 5 + 1 | L4 4 | out 4 = 52 records, no pairs.  The latent group of a skip layer is consumed in a second MFMA phase of that layer: the
// init region holds ONE K64 group per block (the [hash' | x] group), so act(latent) is rebuilt in a dead hidden slot from the raw rows,
// which wait in a per-workgroup scratch in global memory (Args::park)
// MODEL 8 (round 6): PlainNeRF + PosLinearView (`make dnerf`, src/refl.py:248-290): `first` 22 | pos (2 x 256 -> 3 + 64): init 2 | L0 5 + 1 |
// L1 4 | out 4 (three tiles, split by block like first.out) | view (2 x 128, sin; [x | dir | latent | refl_latent | intermediate] -> 1;
// split by BLOCK: a row group computes rows 64 (rg & 1) .. + 63 for block rg >> 1, so all four waves work and convert):
// init 2 | L0 4 | L1 2 | out 2 = 48 records + 2 geometry pairs ([x, y, z, dir, refl_latent]); up to 3 refl_latent columns
// (--dyn-refl-latent) ride in the spare slots of the [hash' | x] group and of the geometry chunk
__host__ __device__ constexpr int npair(int model) { return model == 1 ? 3 : model == 2 ? 2 : model == 3 ? 5 : model == 4 ? 0 : model == 5 ? 3 : model == 7 ? 0 : 2; }  // (6, 8: 2)
__host__ __device__ constexpr int nrec(int model) { return model == 1 ? 28 : model == 2 ? 22 : model == 3 ? 46 : model == 4 ? 27 : model == 5 ? 40 : model == 6 ? 52 : model == 7 ? 52 : model == 8 ? 48 : 44; }
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
  // MODEL 7 / 8 (appended: the kernel-argument offsets of the older schedules do not move)
  const float4* tables2 = nullptr;  // hash tables of the reflectance head's own encoder [8,65536]
  float* park = nullptr;            // per-workgroup scratch: [workgroup][group][block] x 8 KiB of raw latent rows
  const float* rl = nullptr;        // MODEL 8: refl_latent rows [T * R, rl_ld] (--dyn-refl-latent), nullable
  int rl_ld = 0, n_rl = 0;
  // MODEL 9 (the training forward, round 6) takes its four output buffers through fields the PlainNeRF schedule does not use -- y
  // (planes [10][N, 256], N = T * R, dense), park ([N, 69] the View MLP's init rows), rl ([N] density), feat ([N, 3] view.out) -- so that
  // the argument block keeps its size: the implicit kernel arguments behind it, and with them three instructions of the pinned
  // headline kernel (csrc/ls_headline_isa.sha256), do not move
};
constexpr int kParkSlots = 1;       // raw K64 groups a block parks in global memory: the latent rows (8 KiB)
constexpr size_t kParkBytes = (size_t)256 * 2 * 2 * kParkSlots * 8192;  // 256 workgroups x 2 groups x 2 blocks

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
// init chunk q of Linear lin (skip: its columns sit behind the kHidden hidden ones).  q = 32: MODEL 8's geometry chunk [x y z | dir | refl_latent]
// (columns 0..5, then 6 + 64 + j; skip: behind the 128 hidden columns of PosLinearView's view MLP)
struct XPairD { int8_t lin, q, skip; };
// K64 group of Linear lin.  kind 0: hidden features 64 q .. 64 q + 63; kind 1 / 2: the init chunks 0..3 of the MLP (columns by
// init_slot_feature; 2: behind the kHidden hidden columns of a skip layer); kind 3 / 4: Fourier features 64 q .. 64 q + 63 in the
// generator's slot order (fourier_slot_col; 4: behind the hidden columns of a skip layer).  out_mode 0 hidden rows, 1 out row-major (tile
// min(rg, 2)), 2 out, one tile, 3 out split by block: row groups 0,1 hold tiles 0 and 1, row groups 2,3 tile 2; 4: ls_xsched.h; 5 hidden rows
// split by block (a 128-wide Linear: row group rg holds rows 64 (rg & 1) .. + 63)
// kind 6: 64 columns off .. off + 63 in the hidden slot order (a group that was produced by an out Linear's accumulators: the latent /
// intermediate rows); kind 7: the [hash' | x | refl_latent] group of MODEL 8 -- kind 1's slots + refl_latent column j in slot 6 + j of chunk 2
// (weight column off2 + j, off2 = q * 1 ... see xrec_col), everything shifted by `off`
struct XRecD { int8_t lin, q, out_mode, kind; int16_t off; };  // off: added to the column (kinds 1, 2, 5: where the group's columns start)
struct XSched {
  int npair, nrec, nphase, nlin, ndesc, n_rl;
  XLin lin[16];  // (<= 14 Linears: PlainNeRF + PosLinearView 6 + 4 + 4)
  NaMlpDesc desc[3];
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
#if defined(NA_LS_TRAIN_EXP) && (NA_LS_TRAIN_EXP & 16)  // timing experiment (wrong values): no bias loads at all
  for (int q = 0; q < 16; ++q) a[q] = 0.f;
  return a;
#endif
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
  // slot 2 i, b[i] into slot 2 i + 1 (probed on the hardware by the round-3 prototype: profiles/r03/f16x_proto_v1.log)
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

}  // namespace ls
}  // namespace na
