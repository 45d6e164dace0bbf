// Implicit-GEMM convolution (forward and data-gradient) on fp32 MFMA, NHWC, gfx950.
//
//   Y[m][n] = epilogue( sum_{tap} sum_{k<C0+C1} A[m][tap,k] * Wp[tap][k][n] )
//
// m = flattened output pixel (n, oy, ox); A is never materialised: each K-step (one tap x 16 channels) is
// gathered straight from the NHWC source(s) -- zero / reflection padding, the nearest-x2 upsample, the skip
// concat and (for dgrad) the fold-back of the reflection halo are all index arithmetic in the tile loader
// (fp_gather_tap).  Replaces aten::convolution + reflection_pad2d + upsample_nearest2d + cat + elu_ and the
// dgrad half of convolution_backward (reference: footprints/network.py:109-136,151-158,167-170 and the
// torchvision BasicBlock convs behind network.py:38-44).
//
// Tiling: workgroup = 4 waves (256 threads), tile BM pixels x BN channels, K-step 16.  Both operands are staged
// in LDS as [row][16 k + 4 pad] (LD = 20 floats): a lane's MFMA operand is "row idx = lane&31, k-slot = lane>>5",
// and one ds_read_b128 at [row][4*(lane>>5) (+8)] feeds four v_mfma_f32_32x32x2_f32 (k = j for the low half-wave,
// 4 + j for the high one; A and B use the same k permutation so the dot product is unchanged).  LD = 20 makes
// that b128 read conflict-free (5*idx mod 16 is a bijection over each 16-lane service group).
// v_mfma_f32_32x32x2_f32 is an exact k-ordered fmaf chain (64 cycles/SIMD, 157 TF peak), so the kernel is
// MFMA-issue bound by design; global->LDS staging is software-pipelined through registers with two LDS
// buffers and a single barrier per K-step.
#include <stdlib.h>

#include "fp_common.h"

int fp_stem_tile_dispatch(const fp_conv_desc* d, const float* img, const float* wpacked, const float* bias, float* y, hipStream_t stream,
                          const FpBnSink& sink);
int fp_splitk_reduce_stats_launch(const float* part, int SK, int64_t M, int Nout, float* y, hipStream_t stream, unsigned* amax_out,
                                  float* stats, int64_t cap_floats, int* rc_out);
int fp_conv3x3_tile_dispatch(const fp_conv_desc* d, const float* src0, const float* src1, const float* wpacked, const float* bias,
                             const float* addend, const float* addend_mask, const float* actsrc, float* y, hipStream_t stream);

namespace {

struct IgemmArgs {
  const float* src0;
  const float* src1;
  const float* w;
  const float* bias;
  const float* addend;
  const float* addend_mask;
  const float* actsrc;
  float* y;
  FpGeom g;
  int Nout, act;
  unsigned epi;
  int M, KC16, T, tilesN, nwg;
  int SK, stepsPerSplit;  // split-K over the flattened (tap, 16-channel) step range; SK > 1 => raw partials to `part`
  float* part;            // [SK][M][Nout]
  // Parity-major rows for the 3x3 stride-2 data gradient (pm = 1): an input-gradient pixel (y, x) receives only the taps with
  // ky = y + pad and kx = x + pad (mod 2) -- 1, 2, 2 or 4 of the 9.  The M dimension is enumerated class by class ((y & 1, x & 1),
  // heaviest first, each class padded to whole tiles: Mc real rows, McP padded), so a workgroup's rows share one class and it
  // walks only that class's taps: 2.25 taps per pixel on average instead of 9 with three quarters of them all-zero.
  int pm, Mc, McP;
  unsigned* amax_out;     // split-K reduce only: amax slot receiving max |y| (fp16-pair consumers), or null
  // igemm_hp_kernel (fp16-pair operands): packed weights [tap][chunk][2 planes][Nout][16] fp16 and the two operand amax slots
  const unsigned short* w_hp;
  const unsigned* amax_a;
  const unsigned* amax_w;
};

// row of the parity-major enumeration -> class (cpy, cpx), validity, (n, oy, ox)
__device__ __forceinline__ bool pm_decode(const IgemmArgs& a, int m, int& n, int& oy, int& ox) {
  const int cls = m / a.McP, r = m - cls * a.McP;
  const bool valid = r < a.Mc;
  const int rr = valid ? r : 0;
  const int hw = a.g.OW >> 1, hh = a.g.OH >> 1;
  const int j = rr % hw, q = rr / hw;
  ox = 2 * j + (1 - (cls & 1));
  oy = 2 * (q % hh) + (1 - (cls >> 1));
  n = q / hh;
  return valid;
}

// v = acc (+bias) -> addend -> activation gradient -> activation -> accumulate (see FP_EPI_* in the header)
__device__ __forceinline__ float igemm_epilogue(const IgemmArgs& a, size_t o, int n, float v) {
  if (a.epi & FP_EPI_BIAS) v += a.bias[n];
  if (a.epi & FP_EPI_ADDEND) {
    float ad = a.addend[o];
    if (a.epi & FP_EPI_ADDEND_MASK) ad = a.addend_mask[o] > 0.f ? ad : 0.f;
    v += ad;
  }
  if (a.epi & FP_EPI_ACTGRAD_ELU) {
    const float sv = a.actsrc[o];
    v *= (sv > 0.f ? 1.f : sv + 1.f);
  }
  if (a.epi & FP_EPI_ACTGRAD_RELU) v = a.actsrc[o] > 0.f ? v : 0.f;
  if (a.act == FP_ACT_ELU) v = fp_elu(v);
  if (a.act == FP_ACT_RELU) v = fmaxf(v, 0.f);
  if (a.epi & FP_EPI_ACCUM) v += a.y[o];
  return v;
}

// ---- shared epilogue of the flattened kernels.  C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
// unscale: power of two the accumulators are multiplied by before the bias (1 for fp32 operands: fmaf(acc, 1, bias) == acc + bias)
template <int TM, int TN>
__device__ __forceinline__ void igemm_store_tile(const IgemmArgs& a, f32x16 (&acc)[TM][TN], int m0, int n0, int wm, int wn, int idx, int h, int split,
                                                 float unscale) {
  const FpGeom& g = a.g;
  // flag tests hoisted, optional operands of eight rows loaded before any arithmetic (element-at-a-time code serialises the
  // load latencies; see conv3x3_tile_bf3.hip)
  const unsigned epi = a.SK > 1 ? 0u : a.epi;
  const int act = a.SK > 1 ? 0 : a.act;
  float* const dst = a.SK > 1 ? a.part + (size_t)split * a.M * a.Nout : a.y;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + (wn * TN + j) * 32 + idx;
      if (n >= a.Nout) continue;
      const float bias = (epi & FP_EPI_BIAS) ? a.bias[n] : 0.f;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        size_t off[8];
        bool ok[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int r = half * 8 + k;
          const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (a.pm) {
            int en, ey, ex;
            ok[k] = pm_decode(a, m, en, ey, ex);
            off[k] = ((size_t)(en * g.OH + ey) * g.OW + ex) * a.Nout + n;
            continue;
          }
          ok[k] = m < a.M;
          off[k] = (size_t)min(m, a.M - 1) * a.Nout + n;
        }
        float ad[8], mk[8], sv[8], yo[8];
        if (epi & FP_EPI_ADDEND) {
#pragma unroll
          for (int k = 0; k < 8; ++k) ad[k] = a.addend[off[k]];
        }
        if (epi & FP_EPI_ADDEND_MASK) {
#pragma unroll
          for (int k = 0; k < 8; ++k) mk[k] = a.addend_mask[off[k]];
        }
        if (epi & (FP_EPI_ACTGRAD_ELU | FP_EPI_ACTGRAD_RELU)) {
#pragma unroll
          for (int k = 0; k < 8; ++k) sv[k] = a.actsrc[off[k]];
        }
        if (epi & FP_EPI_ACCUM) {
#pragma unroll
          for (int k = 0; k < 8; ++k) yo[k] = a.y[off[k]];
        }
        float v[8];                                   // one wave-uniform branch per flag around an 8-element body
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = fmaf(acc[i][j][half * 8 + k], unscale, bias);
        if (epi & FP_EPI_ADDEND) {
          if (epi & FP_EPI_ADDEND_MASK) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] += mk[k] > 0.f ? ad[k] : 0.f;
          } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] += ad[k];
          }
        }
        if (epi & FP_EPI_ACTGRAD_ELU) {
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] *= (sv[k] > 0.f ? 1.f : sv[k] + 1.f);
        }
        if (epi & FP_EPI_ACTGRAD_RELU) {
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = sv[k] > 0.f ? v[k] : 0.f;
        }
        if (act == FP_ACT_ELU) {
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = fp_elu(v[k]);
        } else if (act == FP_ACT_RELU) {
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
        }
        if (epi & FP_EPI_ACCUM) {
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] += yo[k];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (ok[k]) dst[off[k]] = v[k];
      }
    }
}

constexpr int LD = 20;

template <int BM, int BN, int WM, int WN, bool STEM>
__global__ void __launch_bounds__(256) igemm_kernel(const IgemmArgs a) {
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int AV = BM / 64;         // float4 A slots per thread per K-step
  constexpr int BV = (BN + 63) / 64;  // float4 B slots per thread per K-step
  static_assert(WM * WN == 4, "4 waves");
  __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * LD];
  float* const As = lds;                // [2][BM*LD]
  float* const Bs = lds + 2 * BM * LD;  // [2][BN*LD]

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int idx = lane & 31, h = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  const int wg = fp_xcd_remap(blockIdx.x, a.nwg);
  const int split = wg % a.SK, tile = wg / a.SK;
  const int tile_n = tile % a.tilesN, tile_m = tile / a.tilesN;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const FpGeom& g = a.g;

  // ---- per-thread staging coordinates -----------------------------------------------------------------
  const int q = t & 3;  // which float4 of the 16-channel K-step
  int pn[AV], py[AV], px[AV];
  bool pvalid[AV];
#pragma unroll
  for (int i = 0; i < AV; ++i) {
    const int m = m0 + (t >> 2) + 64 * i;
    if (a.pm) {
      pvalid[i] = pm_decode(a, m, pn[i], py[i], px[i]);
      continue;
    }
    pvalid[i] = m < a.M;
    const int mm = pvalid[i] ? m : 0;
    const int ox = mm % g.OW, r = mm / g.OW;
    px[i] = ox;
    py[i] = r % g.OH;
    pn[i] = r / g.OH;
  }
  // pm: this workgroup's tap list (class = m0 / McP, uniform): ky in {1} or {0, 2}, kx likewise
  const int pm_cls = a.pm ? m0 / a.McP : 0;
  const int pm_odd_y = (1 - (pm_cls >> 1) + g.pad) & 1, pm_odd_x = (1 - (pm_cls & 1) + g.pad) & 1;   // 1: only k = 1 matches
  const int pm_ny = pm_odd_y ? 1 : 2, pm_nx = pm_odd_x ? 1 : 2;
  auto pm_tap = [&](int lt) {
    const int ly = lt / pm_nx, lx = lt - ly * pm_nx;
    return (pm_odd_y ? 1 : 2 * ly) * 3 + (pm_odd_x ? 1 : 2 * lx);
  };
  int pix[AV][4], pix1[AV];
  float4 areg[AV], breg[BV];

  auto set_tap = [&](int tap) {
    if (STEM) return;
    const int ky = tap / g.KW, kx = tap - ky * g.KW;
#pragma unroll
    for (int i = 0; i < AV; ++i) {
      fp_gather_tap(g, pn[i], py[i], px[i], ky, kx, pix[i], pix1[i]);
      if (!pvalid[i]) { pix[i][0] = pix[i][1] = pix[i][2] = pix[i][3] = -1; pix1[i] = -1; }
    }
  };
  auto load_step = [&](int tap, int cc) {
#pragma unroll
    for (int i = 0; i < AV; ++i) {
      if (STEM) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (pvalid[i]) {
          const int kk = cc * 16 + q * 4;
          v.x = fp_stem_load(g, a.src0, pn[i], py[i], px[i], kk + 0);
          v.y = fp_stem_load(g, a.src0, pn[i], py[i], px[i], kk + 1);
          v.z = fp_stem_load(g, a.src0, pn[i], py[i], px[i], kk + 2);
          v.w = fp_stem_load(g, a.src0, pn[i], py[i], px[i], kk + 3);
        }
        areg[i] = v;
      } else {
        areg[i] = fp_gather_load4(g, a.src0, a.src1, pix[i], pix1[i], cc * 16 + q * 4);
      }
    }
    const float* wstep = a.w + (size_t)(tap * a.KC16 + cc) * a.Nout * 16;
#pragma unroll
    for (int j = 0; j < BV; ++j) {
      const int nb = (t >> 2) + 64 * j;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (nb < BN && n0 + nb < a.Nout) v = *reinterpret_cast<const float4*>(wstep + (size_t)(n0 + nb) * 16 + q * 4);
      breg[j] = v;
    }
  };
  auto store_step = [&](int buf) {
#pragma unroll
    for (int i = 0; i < AV; ++i)
      *reinterpret_cast<float4*>(As + buf * BM * LD + ((t >> 2) + 64 * i) * LD + q * 4) = areg[i];
#pragma unroll
    for (int j = 0; j < BV; ++j) {
      const int nb = (t >> 2) + 64 * j;
      if (nb < BN) *reinterpret_cast<float4*>(Bs + buf * BN * LD + nb * LD + q * 4) = breg[j];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int total_steps = (a.pm ? pm_ny * pm_nx : a.T) * a.KC16;      // pm: never split (SK = 1)
  const int s_begin = split * a.stepsPerSplit;
  const int steps = a.pm ? total_steps : min(a.stepsPerSplit, total_steps - s_begin);
  int ltap = s_begin / a.KC16, lcc = s_begin - ltap * a.KC16;
  int tap = a.pm ? pm_tap(ltap) : ltap;
  set_tap(tap);
  load_step(tap, lcc);
  store_step(0);
  __syncthreads();

  for (int s = 0; s < steps; ++s) {
    const bool more = s + 1 < steps;
    if (more) {
      if (++lcc == a.KC16) { lcc = 0; ++ltap; tap = a.pm ? pm_tap(ltap) : ltap; set_tap(tap); }
      load_step(tap, lcc);  // global loads for step s+1 stay in flight under this step's MFMAs
    }
    const float* Ab = As + (s & 1) * BM * LD + (wm * TM * 32 + idx) * LD + h * 4;
    const float* Bb = Bs + (s & 1) * BN * LD + (wn * TN * 32 + idx) * LD + h * 4;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      float4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const float4*>(Ab + i * 32 * LD + kh * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const float4*>(Bb + j * 32 * LD + kh * 8);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
        }
    }
    if (more) store_step((s + 1) & 1);
    __syncthreads();
  }

  igemm_store_tile<TM, TN>(a, acc, m0, n0, wm, wn, idx, h, split, 1.f);
}

// ---- fp16-pair variant (round 3): the same flattened implicit GEMM for the convolutions the halo-tile kernel does not take -- 3x3 stride 2,
// 1x1 (any stride), and their data gradients (zero-padding gathers, parity-major rows included) -- with the operand format of
// conv3x3_tile_bf3.hip: x * 2^ka = h + m in fp16 (per-tensor exponent from the source's amax slot), weights from the FP_PACK_{FWD,DGRAD}_HP
// planes, three v_mfma_f32_32x32x16_f16 products per 16-channel K-step instead of sixteen v_mfma_f32_32x32x2_f32: 192 MFMA cycles per step
// and 32 x 32 tile instead of 1024.
// No LDS and no barriers: the A operand of v_mfma_f32_32x32x16_f16 is "lane = row, 8 consecutive k", i.e. 8 consecutive CHANNELS of the
// lane's own source pixel -- 32 contiguous bytes of the NHWC tensor -- so every lane gathers its own fragment (two float4), splits it in
// registers, and nothing is shared between waves but the cache; the B fragments come straight from the packed planes (1 KB contiguous per
// wave and plane, L1 / L2 resident).  A wave owns 32 rows x 32 TN columns and runs its K loop on its own, loads two K-steps ahead in three
// rotating register sets: waves drift apart and hide each other's latencies.  (The first version of this kernel staged A through LDS
// like igemm_kernel, one barrier per two K-steps with the next pair's loads behind it: 50 us per launch against 65 for the fp32 kernel
// -- every pair exposed a full memory latency; profiles/round3_notes.md.)
typedef _Float16 ig_f16x8 __attribute__((ext_vector_type(8)));
typedef float ig_f32x8 __attribute__((ext_vector_type(8)));

// NP = 2: scaled fp16 pairs (three products).  NP = 3 (round 5): the EXACT bf16 split -- x = h + m + l, six products, no amax slots, no
// scaling -- for the default operand format: the stride-2 3x3 / 1x1 convolutions of the encoder and their data gradients leave the fp32
// MFMA (1/16 of the bf16 rate) without giving up an operand bit.  Weights [tap][chunk][NP planes][Nout][16] from FP_PACK_{FWD,DGRAD}_BF3 jobs.
template <int TN, int NP = 2>
__global__ void __launch_bounds__(256) igemm_hp_kernel(const IgemmArgs a) {
  constexpr int BM = 128, BN = 32 * TN;
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int idx = lane & 31, h = lane >> 5;
  const int wg = fp_xcd_remap(blockIdx.x, a.nwg);
  const int split = wg % a.SK, tile = wg / a.SK;
  const int tile_n = tile % a.tilesN, tile_m = tile / a.tilesN;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const FpGeom& g = a.g;
  float sa = 1.f;
  int kunscale = 0;
  if constexpr (NP == 2) {
    unsigned ma, mw, unused;
    fp_amax3_reduce(fp_amax3_issue(a.amax_a, a.amax_w, nullptr), ma, mw, unused);           // one round trip for both slots (fp_common.h)
    const int ka = fp_hp_exponent(ma, FP_HP_TARGET_ACT);
    sa = ldexpf(1.f, ka);
    kunscale = -(ka + fp_hp_exponent(mw, FP_HP_TARGET_W));
  }

  // this lane's row of the GEMM = one output pixel (forward) / one input-gradient pixel (data gradient)
  int pn, py, px;
  bool pvalid;
  {
    const int m = m0 + wave * 32 + idx;
    if (a.pm) {
      pvalid = pm_decode(a, m, pn, py, px);
    } else {
      pvalid = m < a.M;
      const int mm = pvalid ? m : 0;
      px = mm % g.OW;
      const int r = mm / g.OW;
      py = r % g.OH;
      pn = r / g.OH;
    }
  }
  const int pm_cls = a.pm ? m0 / a.McP : 0;
  const int pm_odd_y = (1 - (pm_cls >> 1) + g.pad) & 1, pm_odd_x = (1 - (pm_cls & 1) + g.pad) & 1;
  const int pm_ny = pm_odd_y ? 1 : 2, pm_nx = pm_odd_x ? 1 : 2;
  auto pm_tap = [&](int lt) {
    const int ly = lt / pm_nx, lx = lt - ly * pm_nx;
    return (pm_odd_y ? 1 : 2 * ly) * 3 + (pm_odd_x ? 1 : 2 * lx);
  };
  const int total_steps = (a.pm ? pm_ny * pm_nx : a.T) * a.KC16;
  const int s_begin = split * a.stepsPerSplit;
  const int steps = a.pm ? total_steps : min(a.stepsPerSplit, total_steps - s_begin);

  // load pointer: walks the K-steps (tap, 16-channel chunk) two ahead of the MFMAs
  int ltap = s_begin / a.KC16, lcc = s_begin - ltap * a.KC16, tap = a.pm ? pm_tap(ltap) : ltap;
  // this lane's source pixel for `tap`: loads are unconditional (padding / out-of-range rows read pixel 0 of the tensor, channel chunks
  // past C0 its last 8 channels) and the fragment is zeroed in registers when it is consumed -- a conditional load compiles into eight
  // scalar flat loads through a select of addresses
  const float* prow = a.src0;
  bool prow_ok = false;
  auto set_tap = [&]() {                             // FP_GATHER_FWD_ZERO / FP_GATHER_DGRAD_ZERO of fp_gather_tap, strides 1 and 2 without divisions
    const int ky = tap / g.KW, kx = tap - ky * g.KW;
    int sy, sx;
    bool ok = pvalid;
    if (g.gather == FP_GATHER_FWD_ZERO) {
      sy = py * g.stride + ky - g.pad;
      sx = px * g.stride + kx - g.pad;
    } else {
      sy = py + g.pad - ky;
      sx = px + g.pad - kx;
      if (g.stride == 2) {
        ok = ok && !((sy | sx) & 1);
        sy >>= 1;
        sx >>= 1;
      } else if (g.stride != 1) {
        ok = ok && sy >= 0 && sx >= 0 && sy % g.stride == 0 && sx % g.stride == 0;
        sy /= g.stride;
        sx /= g.stride;
      }
    }
    ok = ok && sy >= 0 && sy < g.IH && sx >= 0 && sx < g.IW;
    sy = min(max(sy, 0), g.IH - 1);
    sx = min(max(sx, 0), g.IW - 1);
    prow_ok = ok;
    prow = a.src0 + ((size_t)(pn * g.IH + sy) * g.IW + sx) * g.C0;
  };
  set_tap();
  float4 ar[3][2];
  bool aok[3][2];
  uint4 br[3][TN][NP];
  const int cmax = max(g.C0 - 4, 0);
  auto load_next = [&](int set) {
    const int c8 = lcc * 16 + h * 8;
    ar[set][0] = *reinterpret_cast<const float4*>(prow + min(c8, cmax));
    ar[set][1] = *reinterpret_cast<const float4*>(prow + min(c8 + 4, cmax));
    aok[set][0] = prow_ok && c8 < g.C0;
    aok[set][1] = prow_ok && c8 + 4 < g.C0;
    const unsigned short* ws = a.w_hp + (size_t)(tap * a.KC16 + lcc) * NP * a.Nout * 16 + h * 8;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = min(n0 + j * 32 + idx, a.Nout - 1);
#pragma unroll
      for (int p = 0; p < NP; ++p) br[set][j][p] = *reinterpret_cast<const uint4*>(ws + ((size_t)p * a.Nout + n) * 16);
    }
    if (++lcc == a.KC16) {                            // position the pointer for the following step
      lcc = 0;
      ++ltap;
      tap = a.pm ? pm_tap(min(ltap, pm_ny * pm_nx - 1)) : min(ltap, a.T - 1);
      set_tap();
    }
  };

  f32x16 acc[1][TN];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;

  auto consume = [&](int set) {
    const float4 a0 = aok[set][0] ? ar[set][0] : make_float4(0.f, 0.f, 0.f, 0.f), a1 = aok[set][1] ? ar[set][1] : make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (NP == 3) {
      // exact split of the lane's eight channels: h = bf16(x), m = bf16(x - h), l = bf16(x - h - m); each difference is exact in fp32
      typedef float ig_f32x2 __attribute__((ext_vector_type(2)));
      typedef __bf16 ig_bf16x2 __attribute__((ext_vector_type(2)));
      typedef __bf16 ig_bf16x8 __attribute__((ext_vector_type(8)));
      // two elements per conversion (v_cvt_pk_bf16_f32, round to nearest even); float(term) back by a shift / a mask of the packed pair
      float e[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
      unsigned pk[3][4];
#pragma unroll
      for (int lv = 0; lv < 3; ++lv)
#pragma unroll
        for (int q2 = 0; q2 < 4; ++q2) {
          const unsigned u = __builtin_bit_cast(unsigned, __builtin_convertvector(ig_f32x2{e[2 * q2], e[2 * q2 + 1]}, ig_bf16x2));
          pk[lv][q2] = u;
          if (lv < 2) {
            e[2 * q2] -= __builtin_bit_cast(float, u << 16);
            e[2 * q2 + 1] -= __builtin_bit_cast(float, u & 0xffff0000u);
          }
        }
      uint2 pl[3][2];
#pragma unroll
      for (int lv = 0; lv < 3; ++lv) { pl[lv][0] = make_uint2(pk[lv][0], pk[lv][1]); pl[lv][1] = make_uint2(pk[lv][2], pk[lv][3]); }
      ig_bf16x8 va[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) va[p] = __builtin_bit_cast(ig_bf16x8, make_uint4(pl[p][0].x, pl[p][0].y, pl[p][1].x, pl[p][1].y));
      // six of the nine products, smallest first (conv3x3_tile_bf3.hip): l*h, h*l, m*m, m*h, h*m, h*h
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int qq = 0; qq < 6; ++qq)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[PA[qq]], __builtin_bit_cast(ig_bf16x8, br[set][j][PB[qq]]), acc[0][j], 0, 0, 0);
      return;
    }
    uint2 h0, m0, h1, m1;
    fp_hp_split4(a0.x, a0.y, a0.z, a0.w, sa, h0, m0);
    fp_hp_split4(a1.x, a1.y, a1.z, a1.w, sa, h1, m1);
    const ig_f16x8 vh = __builtin_bit_cast(ig_f16x8, make_uint4(h0.x, h0.y, h1.x, h1.y));
    const ig_f16x8 vm = __builtin_bit_cast(ig_f16x8, make_uint4(m0.x, m0.y, m1.x, m1.y));
    // products mh, hm, hh (smallest first), as in conv3x3_tile_bf3.hip
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vm, __builtin_bit_cast(ig_f16x8, br[set][j][0]), acc[0][j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, __builtin_bit_cast(ig_f16x8, br[set][j][1]), acc[0][j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, __builtin_bit_cast(ig_f16x8, br[set][j][0]), acc[0][j], 0, 0, 0);
  };

  if (steps > 0) load_next(0);
  if (steps > 1) load_next(1);
  for (int s = 0; s < steps; s += 3) {
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      if (s + u < steps) {                            // wave-uniform
        if (s + u + 2 < steps) load_next((u + 2) % 3);
        consume(u);
      }
    }
  }
  igemm_store_tile<1, TN>(a, acc, m0, n0, wave, 0, idx, h, split, ldexpf(1.f, kunscale));
}

// y = epilogue(sum_s part[s]) -- fixed summation order
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const IgemmArgs a) {
  const size_t total = (size_t)a.M * a.Nout;
  float ymax = 0.f;
  for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (size_t)gridDim.x * 256) {
    float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;              // four loads in flight, fixed combination order
    int s = 0;
    for (; s + 4 <= a.SK; s += 4) {
      const float* q = a.part + (size_t)s * total + o;
      p0 += q[0]; p1 += q[total]; p2 += q[2 * total]; p3 += q[3 * total];
    }
    for (; s < a.SK; ++s) p0 += a.part[(size_t)s * total + o];
    const float v = igemm_epilogue(a, o, (int)(o % a.Nout), (p0 + p1) + (p2 + p3));
    a.y[o] = v;
    ymax = fmaxf(ymax, fabsf(v));
  }
  if (a.amax_out) fp_amax_publish_block(a.amax_out, ymax);
}

// the same sum for a forward convolution in front of a train-mode BatchNorm (no epilogue options): y = sum_s part[s], plus the (count, mean,
// M2) of the block's rows per channel -> stats[block][Nout][3] (round 4: the statistics of the split-K levels, layer 4 and the 6 x 20 ... 12 x 40
// grids, were a pass of their own behind every such convolution).  Rows and columns as in bn_stats_kernel: a thread owns four channels and
// walks rows blockIdx.x * R + rr, + gridDim.x * R, ...; the R row groups of a block merge through LDS in a fixed order.
__global__ void __launch_bounds__(256) splitk_reduce_stats_kernel(const float* __restrict__ part, int SK, int M, int Nout, float* __restrict__ y,
                                                                  float* __restrict__ stats, unsigned* amax_out) {
  __shared__ float sm[3 * 256 * 4];
  const int C4 = Nout >> 2, R = 256 / C4;
  const int cq = threadIdx.x % C4, rr = threadIdx.x / C4;
  const size_t total = (size_t)M * Nout;
  FpWf w[4] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  float cnt = 0.f, ymax = 0.f;
  for (int m = blockIdx.x * R + rr; m < M; m += gridDim.x * R) {
    const size_t o = (size_t)m * Nout + cq * 4;
    float4 p0 = make_float4(0.f, 0.f, 0.f, 0.f), p1 = p0, p2 = p0, p3 = p0;      // four loads in flight, fixed combination order (as above)
    int s = 0;
    for (; s + 4 <= SK; s += 4) {
      const float* q = part + (size_t)s * total + o;
      const float4 a0 = *reinterpret_cast<const float4*>(q), a1 = *reinterpret_cast<const float4*>(q + total),
                   a2 = *reinterpret_cast<const float4*>(q + 2 * total), a3 = *reinterpret_cast<const float4*>(q + 3 * total);
      p0.x += a0.x; p0.y += a0.y; p0.z += a0.z; p0.w += a0.w;
      p1.x += a1.x; p1.y += a1.y; p1.z += a1.z; p1.w += a1.w;
      p2.x += a2.x; p2.y += a2.y; p2.z += a2.z; p2.w += a2.w;
      p3.x += a3.x; p3.y += a3.y; p3.z += a3.z; p3.w += a3.w;
    }
    for (; s < SK; ++s) {
      const float4 a0 = *reinterpret_cast<const float4*>(part + (size_t)s * total + o);
      p0.x += a0.x; p0.y += a0.y; p0.z += a0.z; p0.w += a0.w;
    }
    const float4 v = make_float4((p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y), (p0.z + p1.z) + (p2.z + p3.z),
                                 (p0.w + p1.w) + (p2.w + p3.w));
    *reinterpret_cast<float4*>(y + o) = v;
    ymax = fp_amax4(ymax, v);
    cnt += 1.f;
    const float rn = 1.f / cnt;
    fp_wf_add(w[0], v.x, cnt, rn); fp_wf_add(w[1], v.y, cnt, rn); fp_wf_add(w[2], v.z, cnt, rn); fp_wf_add(w[3], v.w, cnt, rn);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    sm[(0 * 256 + threadIdx.x) * 4 + j] = w[j].n;
    sm[(1 * 256 + threadIdx.x) * 4 + j] = w[j].mean;
    sm[(2 * 256 + threadIdx.x) * 4 + j] = w[j].m2;
  }
  __syncthreads();
  if (rr == 0) {
    for (int r = 1; r < R; ++r) {
      const int tt = r * C4 + cq;
#pragma unroll
      for (int j = 0; j < 4; ++j) fp_wf_merge(w[j], FpWf{sm[(0 * 256 + tt) * 4 + j], sm[(1 * 256 + tt) * 4 + j], sm[(2 * 256 + tt) * 4 + j]});
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float* p = stats + ((size_t)blockIdx.x * Nout + cq * 4 + j) * 3;
      p[0] = w[j].n; p[1] = w[j].mean; p[2] = w[j].m2;
    }
  }
  if (amax_out) fp_amax_publish_block(amax_out, ymax);
}

// ... and for a data gradient whose output is the masked gradient g entering a train-mode BatchNorm's backward (fp_aux.bnb_*): y =
// epilogue(sum_s part[s]) as in splitk_reduce_kernel, plus (sum g, sum g * xhat) of the block's rows per channel -> bpart[block][Nout][2],
// xhat = (z - mean) * invstd of that BatchNorm in the arithmetic form of bn_bwd_reduce_kernel
__global__ void __launch_bounds__(256) splitk_reduce_bnb_kernel(const IgemmArgs a, const float* __restrict__ z, const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, float* __restrict__ bpart) {
  __shared__ float sm[2 * 256 * 4];
  const int C4 = a.Nout >> 2, R = 256 / C4;
  const int cq = threadIdx.x % C4, rr = threadIdx.x / C4;
  const size_t total = (size_t)a.M * a.Nout;
  const float4 mu = reinterpret_cast<const float4*>(mean)[cq], is = reinterpret_cast<const float4*>(invstd)[cq];
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  float ymax = 0.f;
  for (int m = blockIdx.x * R + rr; m < a.M; m += gridDim.x * R) {
    const size_t o = (size_t)m * a.Nout + cq * 4;
    float4 p0 = make_float4(0.f, 0.f, 0.f, 0.f), p1 = p0, p2 = p0, p3 = p0;
    int s = 0;
    for (; s + 4 <= a.SK; s += 4) {
      const float* q = a.part + (size_t)s * total + o;
      const float4 a0 = *reinterpret_cast<const float4*>(q), a1 = *reinterpret_cast<const float4*>(q + total),
                   a2 = *reinterpret_cast<const float4*>(q + 2 * total), a3 = *reinterpret_cast<const float4*>(q + 3 * total);
      p0.x += a0.x; p0.y += a0.y; p0.z += a0.z; p0.w += a0.w;
      p1.x += a1.x; p1.y += a1.y; p1.z += a1.z; p1.w += a1.w;
      p2.x += a2.x; p2.y += a2.y; p2.z += a2.z; p2.w += a2.w;
      p3.x += a3.x; p3.y += a3.y; p3.z += a3.z; p3.w += a3.w;
    }
    for (; s < a.SK; ++s) {
      const float4 a0 = *reinterpret_cast<const float4*>(a.part + (size_t)s * total + o);
      p0.x += a0.x; p0.y += a0.y; p0.z += a0.z; p0.w += a0.w;
    }
    const float4 zv = *reinterpret_cast<const float4*>(z + o);
    float4 g;
    g.x = igemm_epilogue(a, o + 0, cq * 4 + 0, (p0.x + p1.x) + (p2.x + p3.x));
    g.y = igemm_epilogue(a, o + 1, cq * 4 + 1, (p0.y + p1.y) + (p2.y + p3.y));
    g.z = igemm_epilogue(a, o + 2, cq * 4 + 2, (p0.z + p1.z) + (p2.z + p3.z));
    g.w = igemm_epilogue(a, o + 3, cq * 4 + 3, (p0.w + p1.w) + (p2.w + p3.w));
    *reinterpret_cast<float4*>(a.y + o) = g;
    ymax = fp_amax4(ymax, g);
    s1[0] += g.x; s2[0] += g.x * ((zv.x - mu.x) * is.x);
    s1[1] += g.y; s2[1] += g.y * ((zv.y - mu.y) * is.y);
    s1[2] += g.z; s2[2] += g.z * ((zv.z - mu.z) * is.z);
    s1[3] += g.w; s2[3] += g.w * ((zv.w - mu.w) * is.w);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    sm[(0 * 256 + threadIdx.x) * 4 + j] = s1[j];
    sm[(1 * 256 + threadIdx.x) * 4 + j] = s2[j];
  }
  __syncthreads();
  if (rr == 0) {
    for (int r = 1; r < R; ++r) {
      const int tt = r * C4 + cq;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s1[j] += sm[(0 * 256 + tt) * 4 + j];
        s2[j] += sm[(1 * 256 + tt) * 4 + j];
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float* p = bpart + ((size_t)blockIdx.x * a.Nout + cq * 4 + j) * 2;
      p[0] = s1[j]; p[1] = s2[j];
    }
  }
  if (a.amax_out) fp_amax_publish_block(a.amax_out, ymax);
}

// split-K factor for a grid of `tiles` workgroups over `steps` K-steps: fill ~3 workgroups per CU, >= 8 steps each
int pick_splitk(int64_t tiles, int steps, int64_t MN, int64_t ws_floats) {
  static const int sk1_from = getenv("FP_IGEMM_SK1_FROM") ? atoi(getenv("FP_IGEMM_SK1_FROM")) : 160;     // as in conv3x3_tile_bf3.hip (plan3)
  if (tiles >= sk1_from || ws_floats <= 0) return 1;
  int64_t sk = fp_ceil_div(768, tiles);
  if (sk > steps / 8) sk = steps / 8;
  if (sk * MN > ws_floats) sk = ws_floats / MN;
  return sk < 2 ? 1 : (int)sk;
}

template <int BM, int BN, int WM, int WN, bool STEM>
int launch(IgemmArgs& a, hipStream_t stream, int64_t ws_floats) {
  int tilesM = (int)fp_ceil_div(a.M, BM);
  if (a.pm) {                       // four parity classes, each padded to whole BM-row tiles; never split
    a.McP = (int)fp_ceil_div(a.Mc, BM) * BM;
    tilesM = 4 * (a.McP / BM);
    ws_floats = 0;
  }
  a.tilesN = (int)fp_ceil_div(a.Nout, BN);
  const int steps = a.T * a.KC16;
  const int sk = pick_splitk((int64_t)tilesM * a.tilesN, steps, (int64_t)a.M * a.Nout, ws_floats);
  a.stepsPerSplit = (int)fp_ceil_div(steps, sk);
  a.SK = (int)fp_ceil_div(steps, a.stepsPerSplit);
  a.nwg = tilesM * a.tilesN * a.SK;
  fp_launch((igemm_kernel<BM, BN, WM, WN, STEM>), dim3(a.nwg), dim3(256), 0, stream, a);
  if (a.SK > 1) {
    int64_t g = fp_ceil_div((int64_t)a.M * a.Nout, 256);
    if (g > 4096) g = 4096;
    fp_launch(splitk_reduce_kernel, dim3((int)g), dim3(256), 0, stream, a);
  }
  return fp_check_launch("fp_conv_igemm");
}

template <int TN, int NP = 2>
int launch_hp(IgemmArgs& a, hipStream_t stream, int64_t ws_floats, const FpBnSink& sink) {
  constexpr int BM = 128, BN = 32 * TN;
  int tilesM = (int)fp_ceil_div(a.M, BM);
  if (a.pm) {
    a.McP = (int)fp_ceil_div(a.Mc, BM) * BM;
    tilesM = 4 * (a.McP / BM);
    ws_floats = 0;
  }
  a.tilesN = (int)fp_ceil_div(a.Nout, BN);
  const int steps = a.T * a.KC16;
  const int sk = pick_splitk((int64_t)tilesM * a.tilesN, steps, (int64_t)a.M * a.Nout, ws_floats);
  a.stepsPerSplit = (int)fp_ceil_div(steps, sk);
  a.SK = (int)fp_ceil_div(steps, a.stepsPerSplit);
  a.nwg = tilesM * a.tilesN * a.SK;
  fp_launch((igemm_hp_kernel<TN, NP>), dim3(a.nwg), dim3(256), 0, stream, a);
  if (a.SK > 1 && sink.part && !sink.z && a.epi == 0 && a.act == FP_ACT_NONE && !a.pm) {
    // a strided / 1 x 1 forward convolution in front of a train-mode BatchNorm: the statistics out of the reduce launch (fp_aux.bn_part)
    int rc2 = 0;
    const int nb = fp_splitk_reduce_stats_launch(a.part, a.SK, a.M, a.Nout, a.y, stream, a.amax_out, sink.part, sink.cap_floats, &rc2);
    if (nb > 0) {
      if (sink.nblk_out) *sink.nblk_out = nb;
      return rc2 ? rc2 : fp_check_launch("fp_conv_igemm_hp");
    }
  }
  if (a.SK > 1) {
    int64_t g = fp_ceil_div((int64_t)a.M * a.Nout, 256);
    if (g > 4096) g = 4096;
    fp_launch(splitk_reduce_kernel, dim3((int)g), dim3(256), 0, stream, a);
  }
  return fp_check_launch("fp_conv_igemm_hp");
}

// worst-case split-K workspace: the launcher never uses more than 24 partial copies of the output
constexpr int64_t MAX_SK = 24;

}  // namespace

// y = epilogue(sum over `SK` raw partial copies [SK][M][Nout]) in a fixed order -- shared with conv3x3_tile_bf3.hip
int fp_splitk_reduce_launch(const float* part, int SK, int64_t M, int Nout, const float* bias, const float* addend, const float* addend_mask,
                            const float* actsrc, float* y, int act, unsigned epi, hipStream_t stream, unsigned* amax_out) {
  IgemmArgs a = {};
  a.amax_out = amax_out;
  a.part = const_cast<float*>(part); a.SK = SK; a.M = (int)M; a.Nout = Nout;
  a.bias = bias; a.addend = addend; a.addend_mask = addend_mask; a.actsrc = actsrc; a.y = y; a.act = act; a.epi = epi;
  int64_t g = fp_ceil_div(M * Nout, 256);
  if (g > 4096) g = 4096;
  fp_launch(splitk_reduce_kernel, dim3((int)g), dim3(256), 0, stream, a);
  return fp_check_launch("splitk_reduce");
}

// ... with BatchNorm statistics of y (splitk_reduce_stats_kernel); returns the number of partial blocks written to `stats` (capacity
// `cap_floats`), 0 = shape not handled / capacity too small: the caller then uses fp_splitk_reduce_launch
int fp_splitk_reduce_stats_launch(const float* part, int SK, int64_t M, int Nout, float* y, hipStream_t stream, unsigned* amax_out,
                                  float* stats, int64_t cap_floats, int* rc_out) {
  *rc_out = 0;
  const int C4 = Nout / 4;
  if (Nout % 4 || C4 < 1 || C4 > 256 || 256 % C4 || M >= ((int64_t)1 << 31)) return 0;
  const int R = 256 / C4;
  int64_t blocks = fp_ceil_div(M, (int64_t)R * 4);           // four rows per thread, as fp_bn_train_stats on these small tensors (one row per thread: four
  if (blocks > 512) blocks = 512;                            // times the partial triples for the final stage, measured +0.05 ms per step)
  if (blocks * Nout * 3 > cap_floats) return 0;
  fp_launch(splitk_reduce_stats_kernel, dim3((int)blocks), dim3(256), 0, stream, part, SK, (int)M, Nout, y, stats, amax_out);
  *rc_out = fp_check_launch("splitk_reduce(stats)");
  return (int)blocks;
}

// ... with the BatchNorm-backward sums of y (splitk_reduce_bnb_kernel); returns the number of partial blocks written, 0 = not handled
int fp_splitk_reduce_bnb_launch(const float* part, int SK, int64_t M, int Nout, const float* bias, const float* addend, const float* addend_mask,
                                const float* actsrc, float* y, int act, unsigned epi, hipStream_t stream, unsigned* amax_out, const float* z,
                                const float* mean, const float* invstd, float* bpart, int64_t cap_floats, int* rc_out) {
  *rc_out = 0;
  const int C4 = Nout / 4;
  if (Nout % 4 || C4 < 1 || C4 > 256 || 256 % C4 || M >= ((int64_t)1 << 31)) return 0;
  const int R = 256 / C4;
  int64_t blocks = fp_ceil_div(M, (int64_t)R * 4);
  if (blocks > 512) blocks = 512;
  if (blocks * Nout * 2 > cap_floats) return 0;
  IgemmArgs a = {};
  a.amax_out = amax_out;
  a.part = const_cast<float*>(part); a.SK = SK; a.M = (int)M; a.Nout = Nout;
  a.bias = bias; a.addend = addend; a.addend_mask = addend_mask; a.actsrc = actsrc; a.y = y; a.act = act; a.epi = epi;
  fp_launch(splitk_reduce_bnb_kernel, dim3((int)blocks), dim3(256), 0, stream, a, z, mean, invstd, bpart);
  *rc_out = fp_check_launch("splitk_reduce(bn backward)");
  return (int)blocks;
}

extern "C" int64_t fp_conv_igemm_workspace(const fp_conv_desc* d) {
  if (!d) return 0;
  const int64_t M = (int64_t)d->N * d->OH * d->OW;
  if (fp_ceil_div(M, 128) * fp_ceil_div(d->Nout, 64) >= 384) return 0;   // big grids never split
  return MAX_SK * M * d->Nout * (int64_t)sizeof(float);
}

extern "C" int fp_conv_igemm(const fp_conv_desc* d, const float* src0, const float* src1, const float* wpacked,
                             const float* bias, const float* addend, const float* addend_mask, const float* actsrc,
                             float* y, void* workspace, int64_t workspace_bytes, const fp_aux* aux, fp_stream_t stream_) {
  const FpBnSink bn_sink = fp_bn_sink_of(aux);    // only the stem's tile kernel can emit from this entry point
  hipStream_t stream = (hipStream_t)stream_;
  FP_REQUIRE(d && src0 && wpacked && y, "fp_conv_igemm: null pointer");
  FP_REQUIRE(d->N > 0 && d->OH > 0 && d->OW > 0 && d->Nout > 0, "fp_conv_igemm: empty problem");
  const bool stem = d->gather == FP_GATHER_STEM;
  if (stem) {
    FP_REQUIRE(d->KH == 7 && d->KW == 7 && d->stride == 2 && d->pad == 3 && d->C0 == 3 && d->C1 == 0,
               "fp_conv_igemm: STEM expects 7x7/2 pad 3 on 3 channels");
  } else {
    FP_REQUIRE(d->C0 > 0 && d->C0 % 4 == 0 && d->C1 >= 0 && d->C1 % 4 == 0, "fp_conv_igemm: C0=%d C1=%d must be multiples of 4",
               d->C0, d->C1);
    if (d->gather == FP_GATHER_FWD_REFLECT || d->gather == FP_GATHER_FWD_REFLECT_UP2 || d->gather == FP_GATHER_DGRAD_REFLECT)
      FP_REQUIRE(d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && d->IH == d->OH && d->IW == d->OW && d->IH >= 2 &&
                     d->IW >= 2,
                 "fp_conv_igemm: reflect modes need 3x3 stride 1 pad 1, dims >= 2");
    if (d->gather == FP_GATHER_FWD_REFLECT_UP2) FP_REQUIRE(d->IH % 2 == 0 && d->IW % 2 == 0, "fp_conv_igemm: UP2 needs even dims");
    FP_REQUIRE(d->C1 == 0 || (d->gather == FP_GATHER_FWD_REFLECT_UP2 && src1), "fp_conv_igemm: C1 only with UP2 concat");
  }
  FP_REQUIRE(!(d->epi & FP_EPI_BIAS) || bias, "fp_conv_igemm: bias flag without pointer");
  FP_REQUIRE(!(d->epi & FP_EPI_ADDEND) || addend, "fp_conv_igemm: addend flag without pointer");
  FP_REQUIRE(!(d->epi & FP_EPI_ADDEND_MASK) || addend_mask, "fp_conv_igemm: addend_mask flag without pointer");
  FP_REQUIRE(!(d->epi & (FP_EPI_ACTGRAD_ELU | FP_EPI_ACTGRAD_RELU)) || actsrc, "fp_conv_igemm: actgrad flag without pointer");
  const int64_t M64 = (int64_t)d->N * d->OH * d->OW;
  FP_REQUIRE(M64 * (int64_t)(d->Nout > d->C0 + d->C1 ? d->Nout : d->C0 + d->C1) < (int64_t)1 << 40 && M64 < (int64_t)1 << 31,
             "fp_conv_igemm: problem too large");

  if (stem) {    // patch-in-LDS kernel (stem_tile.hip); shapes / epilogues it does not take stay on the flattened path
    const int rc = fp_stem_tile_dispatch(d, src0, wpacked, bias, y, stream, bn_sink);
    if (rc != -1000) return rc;
  }
  if (!stem) {   // 3x3 stride-1 convs on large grids: halo-tile kernel (conv3x3_tile.hip)
    static const bool no_tile = getenv("FP_NO_TILE") && atoi(getenv("FP_NO_TILE"));
    if (!no_tile) {
      const int rc = fp_conv3x3_tile_dispatch(d, src0, src1, wpacked, bias, addend, addend_mask, actsrc, y, stream);
      if (rc != -1000) return rc;
    }
  }
  IgemmArgs a;
  a.amax_out = nullptr;
  a.src0 = src0; a.src1 = src1; a.w = wpacked; a.bias = bias; a.addend = addend; a.addend_mask = addend_mask;
  a.actsrc = actsrc; a.y = y;
  a.g = FpGeom{d->N, d->OH, d->OW, d->IH, d->IW, d->C0, d->C1, d->KH, d->KW, d->stride, d->pad, d->gather};
  a.Nout = d->Nout; a.act = d->act; a.epi = d->epi;
  a.M = (int)M64;
  a.T = stem ? 1 : d->KH * d->KW;
  a.KC16 = stem ? 10 : (d->C0 + d->C1 + 15) / 16;
  a.part = (float*)workspace;
  a.SK = 1;
  // 3x3 stride-2 data gradient on even dims: parity-major rows, only the taps a pixel class receives (see IgemmArgs)
  static const bool no_pm = getenv("FP_NO_PM") && atoi(getenv("FP_NO_PM"));
  a.pm = !no_pm && d->gather == FP_GATHER_DGRAD_ZERO && d->stride == 2 && d->KH == 3 && d->KW == 3 && d->OH % 2 == 0 && d->OW % 2 == 0 &&
         d->IH * 2 == d->OH && d->IW * 2 == d->OW;
  a.Mc = a.pm ? d->N * (d->OH / 2) * (d->OW / 2) : 0;
  a.McP = a.Mc;
  int64_t ws = workspace ? workspace_bytes / (int64_t)sizeof(float) : 0;
  if (ws > MAX_SK * M64 * d->Nout) ws = MAX_SK * M64 * d->Nout;

  if (stem) return launch<128, 64, 2, 2, true>(a, stream, 0);
  const int64_t M = a.M;
  if (d->Nout <= 32) {
    if (fp_ceil_div(M, 256) >= 512) return launch<256, 32, 4, 1, false>(a, stream, ws);
    return launch<128, 32, 4, 1, false>(a, stream, ws);
  }
  const int64_t t128 = fp_ceil_div(M, 128);
  if (d->Nout % 128 == 0 && t128 * (d->Nout / 128) >= 256) return launch<128, 128, 2, 2, false>(a, stream, ws);
  if (M >= 256) return launch<128, 64, 2, 2, false>(a, stream, ws);   // small grids are filled by split-K
  return launch<64, 64, 2, 2, false>(a, stream, ws);
}

// ---- fp16-pair operands for the flattened kernel (round 3): 3x3 stride-2 / 1x1 convolutions and their data gradients ------------------
// Same operation and epilogue flags as fp_conv_igemm for zero-padding gathers of ONE source tensor (FP_GATHER_FWD_ZERO / FP_GATHER_DGRAD_ZERO,
// C1 = 0, C0 a multiple of 4); weights from FP_PACK_FWD_HP / FP_PACK_DGRAD_HP jobs (any kernel size) with the slot `amax_w` they were scaled
// by; `amax_src` holds max |src|.  Replaces aten::convolution / convolution_backward(data) of torchvision's stride-2 BasicBlock convs and
// 1x1 downsample convs (footprints/network.py:38-44) on the fp16 matrix path instead of the fp32 one.
extern "C" int fp_conv_igemm_hp_supported(const fp_conv_desc* d) {
  if (!d || d->C1 != 0 || d->C0 <= 0 || d->C0 % 4 || d->Nout <= 0) return 0;
  if (d->gather != FP_GATHER_FWD_ZERO && d->gather != FP_GATHER_DGRAD_ZERO) return 0;
  return 1;
}

static int igemm_split_operands(const char* who, const fp_conv_desc* d, const float* src, const void* wpacked, const float* bias, const float* addend,
                                const float* addend_mask, const float* actsrc, float* y, void* workspace, int64_t workspace_bytes,
                                const uint32_t* amax_src, const uint32_t* amax_w, bool exact, const fp_aux* aux, fp_stream_t stream_) {
  const FpBnSink bn_sink = fp_bn_sink_of(aux);    // only a split grid's reduce launch can emit (launch_hp)
  hipStream_t stream = (hipStream_t)stream_;
  (void)who;
  FP_REQUIRE(d && src && wpacked && y && (exact || (amax_src && amax_w)), "fp_conv_igemm_hp / _bf3: null pointer");
  FP_REQUIRE(fp_conv_igemm_hp_supported(d), "fp_conv_igemm_hp / _bf3: shape / gather not supported (see fp_conv_igemm_hp_supported)");
  FP_REQUIRE(!(d->epi & FP_EPI_BIAS) || bias, "fp_conv_igemm_hp / _bf3: bias flag without pointer");
  FP_REQUIRE(!(d->epi & FP_EPI_ADDEND) || addend, "fp_conv_igemm_hp / _bf3: addend flag without pointer");
  FP_REQUIRE(!(d->epi & FP_EPI_ADDEND_MASK) || addend_mask, "fp_conv_igemm_hp / _bf3: addend_mask flag without pointer");
  FP_REQUIRE(!(d->epi & (FP_EPI_ACTGRAD_ELU | FP_EPI_ACTGRAD_RELU)) || actsrc, "fp_conv_igemm_hp / _bf3: actgrad flag without pointer");
  const int64_t M64 = (int64_t)d->N * d->OH * d->OW;
  FP_REQUIRE(M64 > 0 && M64 < (int64_t)1 << 31 && M64 * d->Nout < (int64_t)1 << 40, "fp_conv_igemm_hp / _bf3: problem too large / empty");
  IgemmArgs a = {};
  a.src0 = src; a.w_hp = (const unsigned short*)wpacked; a.bias = bias; a.addend = addend; a.addend_mask = addend_mask; a.actsrc = actsrc; a.y = y;
  a.amax_a = amax_src; a.amax_w = amax_w; a.amax_out = nullptr;
  a.g = FpGeom{d->N, d->OH, d->OW, d->IH, d->IW, d->C0, 0, d->KH, d->KW, d->stride, d->pad, d->gather};
  a.Nout = d->Nout; a.act = d->act; a.epi = d->epi;
  a.M = (int)M64;
  a.T = d->KH * d->KW;
  a.KC16 = (d->C0 + 15) / 16;
  a.part = (float*)workspace;
  a.SK = 1;
  static const bool no_pm = getenv("FP_NO_PM") && atoi(getenv("FP_NO_PM"));
  a.pm = !no_pm && d->gather == FP_GATHER_DGRAD_ZERO && d->stride == 2 && d->KH == 3 && d->KW == 3 && d->OH % 2 == 0 && d->OW % 2 == 0 &&
         d->IH * 2 == d->OH && d->IW * 2 == d->OW;
  a.Mc = a.pm ? d->N * (d->OH / 2) * (d->OW / 2) : 0;
  a.McP = a.Mc;
  int64_t ws = workspace ? workspace_bytes / (int64_t)sizeof(float) : 0;
  if (ws > MAX_SK * M64 * d->Nout) ws = MAX_SK * M64 * d->Nout;
  // wave tile 32 rows x 32 TN columns: wider tiles re-read the A rows less often, narrower ones fill the chip on small grids
  if (exact) {                                    // three planes of weights in flight per K-step: two column blocks per wave at most
    if (d->Nout <= 32) return launch_hp<1, 3>(a, stream, ws, bn_sink);
    return launch_hp<2, 3>(a, stream, ws, bn_sink);
  }
  if (d->Nout <= 32) return launch_hp<1>(a, stream, ws, bn_sink);
  const int64_t t128 = fp_ceil_div(M64, 128);
  if (d->Nout % 128 == 0 && t128 * (d->Nout / 128) >= 512) return launch_hp<4>(a, stream, ws, bn_sink);
  return launch_hp<2>(a, stream, ws, bn_sink);
}

extern "C" int fp_conv_igemm_hp(const fp_conv_desc* d, const float* src, const void* wpacked_hp, const float* bias, const float* addend,
                                const float* addend_mask, const float* actsrc, float* y, void* workspace, int64_t workspace_bytes,
                                const uint32_t* amax_src, const uint32_t* amax_w, const fp_aux* aux, fp_stream_t stream_) {
  return igemm_split_operands("fp_conv_igemm_hp", d, src, wpacked_hp, bias, addend, addend_mask, actsrc, y, workspace, workspace_bytes, amax_src,
                              amax_w, false, aux, stream_);
}

// The same operation with EXACTLY split bf16x3 operands (round 5; the default operand format's path for the encoder's stride-2 3x3 and 1x1
// convolutions and their data gradients, footprints/network.py:38-44): weights from FP_PACK_FWD_BF3 / FP_PACK_DGRAD_BF3 jobs (any kernel
// size; fp_packed_weight_elems_bf3 floats), no amax slots.  Shapes: fp_conv_igemm_hp_supported.
extern "C" int fp_conv_igemm_bf3(const fp_conv_desc* d, const float* src, const void* wpacked_bf3, const float* bias, const float* addend,
                                 const float* addend_mask, const float* actsrc, float* y, void* workspace, int64_t workspace_bytes,
                                 const fp_aux* aux, fp_stream_t stream_) {
  return igemm_split_operands("fp_conv_igemm_bf3", d, src, wpacked_bf3, bias, addend, addend_mask, actsrc, y, workspace, workspace_bytes, nullptr,
                              nullptr, true, aux, stream_);
}
