// 3x3 stride-1 weight gradient with split operands (gfx950): exact bf16x3 (six products) or scaled fp16 pairs (three).
//
//   dW[tap][ci][co] = sum over pixels p of  X[p + tap][ci] * dZ[p][co]
//
// Kernels in this file (the entry points pick: fp16 pairs -> wgrad3x3_hp_pf_kernel, exact split -> wgrad3x3_bf3_v3_kernel):
//   wgrad3x3_bf3_v3_kernel    [pixel][channel] planes in LDS + hardware transpose reads (ds_read_b64_tr_b16); exact split and fp16 pairs
//   wgrad3x3_hp_pf_kernel     the same LDS layout / MFMA order / sums, operands two chunks ahead in a register ring of raw buffer loads
//   wgrad_reduce_bias[_t]_kernel   the fixed-order sum over the pixel splits into OIHW (+ bias gradient); _t: through an LDS transpose
//
// The contraction runs over PIXELS, so both MFMA operands want "8 consecutive pixels of one channel" per lane while the tensors are NHWC.
// A workgroup owns a (32 ci x 32 co) block of all nine taps and walks 4 x 16 pixel chunks; wave w takes chunk row w: ONE 16-pixel k-step per
// tap.  The first generation (transposition at staging time, funnel shifts for the kx taps; rounds 1-3 behind FP_WGRAD_BF3_V=1) and the second
// (fp32 transposed in LDS) have left the tree: commits ea6fa4d / df15caf hold them, profiles/round2_notes.md their counters.
#include <stdio.h>

#include <type_traits>

#include "fp_common.h"

int fp_wgrad_reduce_launch(const float* part, float* dw, int S, int T, int Kc, int Nout, int stem, int accumulate, int kc_total,
                           int k_begin, hipStream_t stream);

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct W3Args {
  const float* x;     // [N][H][W][C]
  const float* dz;    // [N][H][W][Nout]
  float* part;        // [S][9][C][Nout]
  float* bpart;       // [S][Nout] column sums of dZ (bias gradient), or null: written by the ci-tile-0 workgroups
  int N, H, W, C, Nout;
  int mode;           // 0 zero padding, 1 reflection, 2 reflection over the nearest-x2 upsampling of x ([N][H/2][W/2][C])
  int chunksY, chunksX, nchunks, chunksPerSplit, S, citiles, cotiles;
  unsigned long long* stamps;   // debugging (FP_W3_STAMPS=file): per workgroup {start, loop start, loop end, end} shader clocks + HW id
  int nofast;         // A/B switch: every chunk through the general (reflect / clamp / mask) staging path
  const unsigned* amax_x;    // fp16-pair variant: amax slots of x and dz (null = exact bf16 split)
  const unsigned* amax_dz;
  int xcd;            // logical workgroup ids run contiguously inside an XCD: the citiles x cotiles workgroups of one pixel split read
                      // the same X / dZ chunks and then share that XCD's L2 (consecutive hardware ids go to different XCDs)
};

constexpr int CH = 4, CW = 16, HR = CH + 2;
#ifndef FP_WGRAD_PF_DEFAULT
#define FP_WGRAD_PF_DEFAULT 2
#endif
#ifndef FP_W3_INTERLEAVE
#define FP_W3_INTERLEAVE 1          // 1 (default since round 5): the ring kernels stage the next chunk BETWEEN their MFMAs (see the kernel); 0: behind them
#endif


// One level of the exact split of a float4, two elements per conversion: q = bf16 pair (the stored form: v_cvt_pk_bf16_f32, round to nearest even),
// r = v - float(q) (exact).  The generic vector conversion made the compiler convert every element twice (once alone with a zero partner to
// rebuild float(h) by a shift, once in pairs for the store): 30 VALU instructions per float4 and plane pair instead of 22.
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint2 split_level(f32x4& v) {
  const unsigned lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v.x, v.y}, bf16x2));
  const unsigned hi = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v.z, v.w}, bf16x2));
  v.x -= __builtin_bit_cast(float, lo << 16);
  v.y -= __builtin_bit_cast(float, lo & 0xffff0000u);
  v.z -= __builtin_bit_cast(float, hi << 16);
  v.w -= __builtin_bit_cast(float, hi & 0xffff0000u);
  return make_uint2(lo, hi);
}
__device__ __forceinline__ uint2 split_last(const f32x4 v) {
  return make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v.x, v.y}, bf16x2)),
                    __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v.z, v.w}, bf16x2)));
}

__device__ __forceinline__ void split_store(unsigned char* p, int plane_stride, f32x4 v) {
  *reinterpret_cast<uint2*>(p) = split_level(v);
  *reinterpret_cast<uint2*>(p + plane_stride) = split_level(v);
  *reinterpret_cast<uint2*>(p + 2 * plane_stride) = split_last(v);
}

// (The first generation -- bf16 planes transposed at staging time, funnel shifts for the kx taps -- was kept behind FP_WGRAD_BF3_V=1 through round 3
// and left the tree in round 4: commit ea6fa4d still has it.)

// (A second generation -- fp32 transposed in LDS, exact split at read time, double-buffered, one barrier per chunk -- measured the same
// as the first in the training step and is gone from the tree: commit df15caf still has it, profiles/round2_notes.md its counters.)

// ---- third generation: [pixel][channel] bf16 planes in LDS, hardware transpose reads (ds_read_b64_tr_b16) ----------------------
// PMC of the second kernel (64 -> 64 @ 96 x 320): per chunk and wave 54 MFMAs against ~470 VALU instructions; the waves spend 41 %
// of their cycles issuing, the VALU pipe is 58 % busy, the MFMA pipe 40 %: the kernel is bound by the instructions that TRANSPOSE
// (NHWC has channels contiguous, the contraction runs over pixels), shift (v_alignbit for the kx taps) and re-convert the operands.
// gfx950's LDS can transpose on the way out: ds_read_b64_tr_b16 hands lane l the 4 ROWS x 1 column of a [4][16] bf16 block whose
// 16 lanes each name one 8-byte quarter row (verified lane by lane on the box: scripts/ubench/bin/tr_probe).  So the planes stay in
// the tensors' own order
//     Xs[buf][plane][halo pixel 6 x 18][32 ci]   Zs[buf][plane][pixel 4 x 16][32 co]     (bf16, 64 bytes per pixel)
// and   * staging is a straight copy: a thread converts one float4 (4 channels of one pixel) into its three exact bf16 terms and
//         writes three 8-byte pieces; a wave's writes are 512 contiguous bytes (conflict-free), each element is converted ONCE;
//       * an MFMA fragment (lane = channel, 8 consecutive pixels) is two transpose reads; the kx = 1, 2 taps are the same reads one /
//         two pixels (64 / 128 bytes) further: no funnel shifts, no register copies; a half-wave's 32 lanes cover 256 contiguous
//         bytes per read (all 64 banks once);
//       * 33 KB per buffer: two buffers, two workgroups per CU, one barrier per chunk (as in the second kernel).
// Per chunk and wave: 54 MFMAs + 60 LDS reads + ~110 VALU + 18 LDS writes instead of 54 + 11 + ~470 + 8.
constexpr int HWD = CW + 2;                                     // halo width in pixels
constexpr int PXB = 64;                                         // bytes per pixel per plane (32 channels bf16)
constexpr int XP3 = HR * HWD * PXB, ZP3 = CH * CW * PXB;        // 6912, 4096 bytes per plane
// a buffer holds NP planes of each: 3 x (6912 + 4096) = 33024 bytes (bf16 split), 2 x = 22016 bytes (fp16 pairs)
constexpr int XITEMS = HR * HWD * 8;                            // (pixel, channel quad) staging items of X: 864 (dZ: 4 x 16 x 8 = 512 = two per thread)

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
template <int NP>
__device__ __forceinline__ void split_store_np(unsigned char* p, int plane_stride, f32x4 v, int kscale) {
  if (NP == 3) {
    split_store(p, plane_stride, v);
    return;
  }
  uint2 hq, mq;
  fp_hp_split4(v.x, v.y, v.z, v.w, ldexpf(1.f, kscale), hq, mq);      // (kscale is wave-uniform: the power of two is formed once by the scalar unit)
  *reinterpret_cast<uint2*>(p) = hq;
  *reinterpret_cast<uint2*>(p + plane_stride) = mq;
}
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 lds_tr16(const unsigned char* p) {
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  return __builtin_bit_cast(uint2, v);
}

// NP = 3: exact bf16 split (six products); NP = 2: scaled fp16 pairs (fp_common.h: FP_HP_PRODUCTS products on v_mfma_f32_32x32x16_f16; X and dZ
// each scaled by their own amax slot, the partial sums are unscaled -- exactly -- before they leave the workgroup)
template <int MODE, int NP = 3>
__global__ void __launch_bounds__(256, 2) wgrad3x3_bf3_v3_kernel(const W3Args a) {
  constexpr int XBN = NP * XP3, BUFN = NP * (XP3 + ZP3);
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUFN];
  int kx_ = 0, kz_ = 0;
  if (NP == 2) {
    unsigned mx, mz, unused;
    fp_amax3_reduce(fp_amax3_issue(a.amax_x, a.amax_dz, nullptr), mx, mz, unused);
    kx_ = fp_hp_exponent(mx, FP_HP_TARGET_ACT);
    kz_ = fp_hp_exponent(mz, FP_HP_TARGET_ACT);
  }
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  int b = a.xcd ? fp_xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
  const int cot = b % a.cotiles; b /= a.cotiles;
  const int cit = b % a.citiles; b /= a.citiles;
  const int s = b;
  const int ci0 = cit * 32, co0 = cot * 32;
  // split s walks chunks s, s + S, s + 2 S, ...: image-border chunks (general staging path, +30 % per chunk) spread evenly over the
  // workgroups -- with contiguous ranges the workgroups that own an image's top / bottom chunk rows ran 26 % longer than the median
  // and set the kernel's duration -- and neighbouring workgroups of an XCD stage neighbouring chunks at the same time (L2 reuse)
  const int c_begin = s, c_end = a.nchunks, c_step = a.S;
  unsigned long long st0 = 0, st1 = 0, st2 = 0;
  if (a.stamps) st0 = __builtin_readcyclecounter();

  // staging items e = t + 256 k: pixel e / 8 (row-major over the halo / the chunk), channel quad e % 8
  const int q = t & 7;
  int xhy[4], xhx[4], xgo[4];
  bool xit[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int e = t + 256 * k, p = e >> 3;
    xit[k] = e < XITEMS;
    xhy[k] = p / HWD;
    xhx[k] = p - xhy[k] * HWD;
    xgo[k] = ((xhy[k] - 1) * a.W + xhx[k] - 1) * a.C + ci0 + q * 4;      // offset from the chunk origin (interior chunks)
  }
  int zy[2], zx[2], zgo[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int p = (t + 256 * k) >> 3;
    zy[k] = p / CW;
    zx[k] = p - zy[k] * CW;
    zgo[k] = (zy[k] * a.W + zx[k]) * a.Nout + co0 + q * 4;
  }
  float4 xr[4], zv[2];
  const bool want_bias = a.bpart != nullptr && cit == 0;
  float bs[4] = {0.f, 0.f, 0.f, 0.f};

  auto issue = [&](int c) {
    const int cx = c % a.chunksX;
    const int r = c / a.chunksX;
    const int cy = r % a.chunksY, n = r / a.chunksY;
    const int y0 = cy * CH, x0 = cx * CW;
    const bool fast = !a.nofast && MODE != 2 && y0 >= 1 && y0 + CH + 1 <= a.H && x0 >= 1 && x0 + CW + 1 <= a.W;
    if (fast) {                                        // wave-uniform: no reflection, clamping or masks inside the image
      const size_t org = (size_t)(n * a.H + y0) * a.W + x0;
      const float* px = a.x + org * a.C;
      const float* pz = a.dz + org * a.Nout;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (k < 3 || xit[k]) xr[k] = *reinterpret_cast<const float4*>(px + xgo[k]);
#pragma unroll
      for (int k = 0; k < 2; ++k) zv[k] = *reinterpret_cast<const float4*>(pz + zgo[k]);
      return;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int sy = y0 + xhy[k] - 1, sx = x0 + xhx[k] - 1;
      bool ok = xit[k];
      if (MODE == 0) ok = ok && sy >= 0 && sy < a.H && sx >= 0 && sx < a.W;
      else { ok = ok && sy >= -1 && sy <= a.H && sx >= -1 && sx <= a.W; sy = fp_reflect(sy, a.H); sx = fp_reflect(sx, a.W); }
      sy = min(max(sy, 0), a.H - 1);
      sx = min(max(sx, 0), a.W - 1);
      const size_t xpix = MODE == 2 ? (size_t)(n * (a.H >> 1) + (sy >> 1)) * (a.W >> 1) + (sx >> 1) : (size_t)(n * a.H + sy) * a.W + sx;
      const float4 v = *reinterpret_cast<const float4*>(a.x + xpix * a.C + ci0 + q * 4);
      xr[k] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int oy = y0 + zy[k], ox = x0 + zx[k];
      const bool ok = oy < a.H && ox < a.W;
      const float4 v = *reinterpret_cast<const float4*>(a.dz + ((size_t)(n * a.H + min(oy, a.H - 1)) * a.W + min(ox, a.W - 1)) * a.Nout + co0 + q * 4);
      zv[k] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto stage = [&](int buf) {
    unsigned char* const base = lds + buf * BUFN;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < 3 || xit[k]) split_store_np<NP>(base + (t + 256 * k) * 8, XP3, f32x4{xr[k].x, xr[k].y, xr[k].z, xr[k].w}, kx_);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (want_bias) { bs[0] += zv[k].x; bs[1] += zv[k].y; bs[2] += zv[k].z; bs[3] += zv[k].w; }
      split_store_np<NP>(base + XBN + (t + 256 * k) * 8, ZP3, f32x4{zv[k].x, zv[k].y, zv[k].z, zv[k].w}, kz_);
    }
  };

  f32x16 acc[9];
#pragma unroll
  for (int tp = 0; tp < 9; ++tp)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[tp][r] = 0.f;

  if (c_begin < c_end) {
    issue(c_begin);
    stage(0);
  }
  __syncthreads();
  if (a.stamps) st1 = __builtin_readcyclecounter();
  // transpose-read addressing: lane -> (channel = lane % 32, pixel group = lane / 32); its 16-lane group reads a [4 px][16 ch] block,
  // lane i of the group naming pixel i / 4, channels 4 (i % 4) .. + 3 of it
  const int li = lane & 15;
  const int lrow = 8 * (lane >> 5) + (li >> 2), lcol = 32 * ((lane >> 4) & 1) + 8 * (li & 3);
  const int xrd = (wave * HWD + lrow) * PXB + lcol;
  const int zrd = XBN + (wave * CW + lrow) * PXB + lcol;
  int par = 0;
  for (int c = c_begin; c < c_end; c += c_step, par ^= 1) {
    const unsigned char* const Bb = lds + par * BUFN;
#if !defined(FP_W3_ABL) || FP_W3_ABL != 1               // ablation 1: operands loaded once
    if (c + c_step < c_end) issue(c + c_step);       // next chunk's global loads fly under this chunk's MFMAs
#endif
    uint4 bz[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const uint2 lo = lds_tr16(Bb + zrd + p * ZP3), hi = lds_tr16(Bb + zrd + p * ZP3 + 4 * PXB);
      bz[p] = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      uint4 af[3][NP];                               // [kx][plane]
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const unsigned char* src = Bb + xrd + (ky * HWD + kx) * PXB + p * XP3;
          const uint2 lo = lds_tr16(src), hi = lds_tr16(src + 4 * PXB);
          af[kx][p] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
#if defined(FP_W3_ABL) && FP_W3_ABL == 2                 // ablation 2: operands consumed, no MFMA
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int p = 0; p < NP; ++p) asm volatile("" ::"v"(af[kx][p].x), "v"(af[kx][p].y), "v"(af[kx][p].z), "v"(af[kx][p].w), "v"(bz[p].x), "v"(bz[p].y));
      continue;
#endif
      constexpr int NPROD = NP == 3 ? 6 : 4;         // smallest products first
      constexpr int PA[6] = {NP == 3 ? 2 : 1, NP == 3 ? 0 : 1, NP == 3 ? 1 : 0, NP == 3 ? 1 : 0, 0, 0};
      constexpr int PB[6] = {NP == 3 ? 0 : 1, NP == 3 ? 2 : 0, 1, 0, 1, 0};
#pragma unroll
      for (int qq = (NP == 2 ? 4 - FP_HP_PRODUCTS : 0); qq < NPROD; ++qq) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
          if (NP == 2)
            acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[kx][PA[qq]]), __builtin_bit_cast(f16x8, bz[PB[qq]]),
                                                                      acc[ky * 3 + kx], 0, 0, 0);
          else
            acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[kx][PA[qq]]), __builtin_bit_cast(bf16x8, bz[PB[qq]]),
                                                                       acc[ky * 3 + kx], 0, 0, 0);
      }
    }
    if (c + c_step < c_end) stage(par ^ 1);           // the other buffer was last read in the previous chunk, before the barrier that ended it
    __syncthreads();
  }

  if (a.stamps) st2 = __builtin_readcyclecounter();
  // ---- sum the four waves' tiles through LDS (fixed order), one tap at a time -----------------------------------------------------
  float* red = reinterpret_cast<float*>(lds);        // [4 waves][16 regs][64 lanes] = 16 KB
  float* out = a.part + (size_t)s * 9 * a.C * a.Nout;
  if (want_bias) {                                   // 32 staging threads per channel quad -> one partial per output channel
#pragma unroll
    for (int k = 0; k < 4; ++k) red[(t >> 3) * 32 + q * 4 + k] = bs[k];
    __syncthreads();
    if (t < 32) {
      float v = 0.f;
#pragma unroll
      for (int g = 0; g < 32; ++g) v += red[g * 32 + t];
      a.bpart[(size_t)s * a.Nout + co0 + t] = v;
    }
    __syncthreads();
  }
  const int idx = lane & 31;
#pragma unroll
  for (int tp = 0; tp < 9; ++tp) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[tp][r];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int e = t + 256 * k;
      float v = ((red[e] + red[1024 + e]) + red[2048 + e]) + red[3072 + e];
      if (NP == 2) v = ldexpf(v, -(kx_ + kz_));
      const int r = e >> 6, ln = e & 63;
      const int ci = ci0 + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5);
      out[((size_t)tp * a.C + ci) * a.Nout + co0 + (ln & 31)] = v;
    }
    __syncthreads();
  }
  (void)idx;
  if (a.stamps && t == 0) {
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long* o = a.stamps + (size_t)blockIdx.x * 6;
    o[0] = st0; o[1] = st1; o[2] = st2; o[3] = __builtin_readcyclecounter(); o[4] = hwid; o[5] = xcc;
  }
}

// ---- fourth generation (fp16 pairs): the third's LDS layout and MFMA schedule, operands D chunks ahead in a register ring ----------
// The third kernel runs one workgroup per CU (the planner's choice: fewer partial tensors, the other half of every SIMD's registers
// stays with the data-gradient chain on the main stream) -- ONE wave per SIMD -- and its chunk period is a memory latency plus the
// staging work: the next chunk's loads are issued at the top of a chunk and needed at its bottom, 27 MFMAs (~0.4 us) later.  Here a
// chunk's loads are issued D chunk periods before they are staged: ring slot j holds chunk k + 1 + j while chunk k is multiplied;
// the bottom of chunk k stages slot (k + 1) % D into the other LDS buffer and refills it with chunk k + 1 + D.  Every path issues the
// same six vector loads (past the last chunk: from one cache line), so the wait counters the compiler derives are exact on all of
// them; the staging has no branches either (the X plane is padded to 1024 items, the bias sum is masked by a multiplier).
// Round 5: NP = 3 runs the same ring with the EXACT bf16 split (three planes, six products, no amax slots, no un-scaling) -- the default
// operand format's weight gradient had stayed on the third generation; same LDS layout per plane, same MFMA order per accumulator, so its
// results are bit-identical to wgrad3x3_bf3_v3_kernel<MODE, 3> (tests/test_gpu_switches.py: FP_WGRAD_PF=0 against the default).
constexpr int XP4 = 1024 * 8;                                   // X plane: 864 staging items padded to 4 per thread

template <int MODE, int D, int NP = 2>
__global__ void __launch_bounds__(256, D > 2 ? 1 : 2) wgrad3x3_hp_pf_kernel(const W3Args a) {
  constexpr int XBN = NP * XP4;
  constexpr int BUF4 = NP * (XP4 + ZP3);                        // 24576 bytes per buffer (fp16 pairs), 36864 (bf16 split)
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUF4];
  // (fp16 pairs) both amax slots: one vector load, reduced behind the first chunk's loads
  const unsigned amax_raw = NP == 2 ? fp_amax3_issue(a.amax_x, a.amax_dz, nullptr) : 0u;
  int kx_ = 0, kz_ = 0;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  int b = a.xcd ? fp_xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
  const int cot = b % a.cotiles; b /= a.cotiles;
  const int cit = b % a.citiles; b /= a.citiles;
  const int s = b;
  const int ci0 = cit * 32, co0 = cot * 32;
  const int cnt = s < a.nchunks ? (a.nchunks - 1 - s) / a.S + 1 : 0;     // this split's chunks: s, s + S, s + 2 S, ...
  // coordinates of the next chunk to issue, advanced by S chunks with carries (no divisions in the loop)
  int cx = s % a.chunksX, cy = (s / a.chunksX) % a.chunksY, cn = s / (a.chunksX * a.chunksY);
  const int dcx = a.S % a.chunksX, dcy = (a.S / a.chunksX) % a.chunksY, dcn = a.S / (a.chunksX * a.chunksY);
  int issued = 0;

  // staging items e = t + 256 k: pixel e / 8 (row-major over the halo / the chunk), channel quad e % 8; X items past the halo's 108
  // pixels re-load its last pixel into the plane's padding.
  // Loads are raw buffer loads (`buffer_load_dwordx4 v, v_offset, s[rsrc], s_offset offen`): a 32-bit per-lane byte offset plus a
  // wave-uniform one, no 64-bit address registers, and a lane whose offset has bit 31 set is out of the buffer's range and gets
  // zeros without a memory access -- padding, masked-out corners and the loads past the last chunk need no branch around the load
  // (tensors of 2 GB and more take the third-generation kernel).
  const int q = t & 7;
  unsigned xgb[4];                                                      // interior chunks: offsets from the halo's first pixel
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int p = min((t + 256 * k) >> 3, HR * HWD - 1);
    const int hy = p / HWD, hx = p - hy * HWD;
    xgb[k] = (unsigned)(((hy * a.W + hx) * a.C + ci0 + q * 4) * 4);
  }
  const unsigned zgb = (unsigned)((((t >> 7) * a.W + ((t >> 3) & 15)) * a.Nout + co0 + q * 4) * 4);
  const unsigned zstep = (unsigned)(2 * a.W * a.Nout * 4);
  const unsigned xbytes = (unsigned)((size_t)a.N * (MODE == 2 ? (a.H >> 1) * (a.W >> 1) : a.H * a.W) * a.C * 4);
  const unsigned zbytes = (unsigned)((size_t)a.N * a.H * a.W * a.Nout * 4);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dz), 0, zbytes, 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  float4 xr[D][4], zv[D][2];
#pragma unroll
  for (int j = 0; j < D; ++j) {
#pragma unroll
    for (int k = 0; k < 4; ++k) xr[j][k] = make_float4(0.f, 0.f, 0.f, 0.f);
    zv[j][0] = zv[j][1] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const bool want_bias = a.bpart != nullptr && cit == 0;
  float bs[4] = {0.f, 0.f, 0.f, 0.f};

  auto issue = [&](auto slot_) {
    constexpr int sl = decltype(slot_)::value;
    unsigned vo[6], sox = 0, soz = 0;
    if (issued >= cnt) {                               // past the last chunk: six loads that touch no memory
#pragma unroll
      for (int k = 0; k < 6; ++k) vo[k] = OOB;
    } else {
      ++issued;
      const int y0 = cy * CH, x0 = cx * CW, n = cn;
      cx += dcx;
      int carry = cx >= a.chunksX ? 1 : 0;
      cx -= carry * a.chunksX;
      cy += dcy + carry;
      carry = cy >= a.chunksY ? 1 : 0;
      cy -= carry * a.chunksY;
      cn += dcn + carry;
      const bool fast = !a.nofast && MODE != 2 && y0 >= 1 && y0 + CH + 1 <= a.H && x0 >= 1 && x0 + CW + 1 <= a.W;
      if (fast) {                                      // wave-uniform: no reflection, clamping or masks inside the image
        const unsigned org = (unsigned)((n * a.H + y0) * a.W + x0);
        sox = (org - a.W - 1) * a.C * 4;
        soz = org * a.Nout * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) vo[k] = xgb[k];
        vo[4] = zgb;
        vo[5] = zgb + zstep;
      } else {
        int tt = t;                                    // opaque copy: the halo coordinates are recomputed here instead of living in
        asm volatile("" : "+v"(tt));                   // eight registers (two of them spilled) across the whole kernel for the border chunks
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int p = min((tt + 256 * k) >> 3, HR * HWD - 1);
          const int hy = p / HWD, hx = p - hy * HWD;
          int sy = y0 + hy - 1, sx = x0 + hx - 1;
          bool ok;
          if (MODE == 0) ok = sy >= 0 && sy < a.H && sx >= 0 && sx < a.W;
          else { ok = sy >= -1 && sy <= a.H && sx >= -1 && sx <= a.W; sy = fp_reflect(sy, a.H); sx = fp_reflect(sx, a.W); }
          sy = min(max(sy, 0), a.H - 1);
          sx = min(max(sx, 0), a.W - 1);
          const unsigned xpix = MODE == 2 ? (unsigned)((n * (a.H >> 1) + (sy >> 1)) * (a.W >> 1) + (sx >> 1)) : (unsigned)((n * a.H + sy) * a.W + sx);
          vo[k] = ((xpix * a.C + ci0 + q * 4) * 4) | (ok ? 0u : OOB);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int oy = y0 + (tt >> 7) + 2 * k, ox = x0 + ((tt >> 3) & 15);
          const bool ok = oy < a.H && ox < a.W;
          vo[4 + k] = ((unsigned)(((n * a.H + min(oy, a.H - 1)) * a.W + min(ox, a.W - 1)) * a.Nout + co0 + q * 4) * 4) | (ok ? 0u : OOB);
        }
      }
    }
    sox = __builtin_amdgcn_readfirstlane(sox);
    soz = __builtin_amdgcn_readfirstlane(soz);
#pragma unroll
    for (int k = 0; k < 4; ++k) xr[sl][k] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rx, vo[k], sox, 0));
#pragma unroll
    for (int k = 0; k < 2; ++k) zv[sl][k] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rz, vo[4 + k], soz, 0));
  };
  // `live` = 1.f when the slot holds a chunk of this split and the workgroup owns the bias partial, else 0.f
  auto stage = [&](auto slot_, int buf, float live) {
    constexpr int sl = decltype(slot_)::value;
    unsigned char* const base = lds + buf * BUF4;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      split_store_np<NP>(base + (t + 256 * k) * 8, XP4, f32x4{xr[sl][k].x, xr[sl][k].y, xr[sl][k].z, xr[sl][k].w}, kx_);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      bs[0] = fmaf(zv[sl][k].x, live, bs[0]); bs[1] = fmaf(zv[sl][k].y, live, bs[1]);
      bs[2] = fmaf(zv[sl][k].z, live, bs[2]); bs[3] = fmaf(zv[sl][k].w, live, bs[3]);
      split_store_np<NP>(base + XBN + (t + 256 * k) * 8, ZP3, f32x4{zv[sl][k].x, zv[sl][k].y, zv[sl][k].z, zv[sl][k].w}, kz_);
    }
  };

  f32x16 acc[9];
#pragma unroll
  for (int tp = 0; tp < 9; ++tp)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[tp][r] = 0.f;

  const float bias_on = want_bias ? 1.f : 0.f;
  // prologue: chunk 0 through slot 0 into buffer 0, then chunks 1 .. D into slots 1 .. D - 1, 0
  issue(std::integral_constant<int, 0>{});
  if (NP == 2) {
    unsigned mx, mz, unused;
    fp_amax3_reduce(amax_raw, mx, mz, unused);
    kx_ = fp_hp_exponent(mx, FP_HP_TARGET_ACT);
    kz_ = fp_hp_exponent(mz, FP_HP_TARGET_ACT);
  }
  stage(std::integral_constant<int, 0>{}, 0, cnt > 0 ? bias_on : 0.f);
  if (D > 1) issue(std::integral_constant<int, 1 % D>{});
  if (D > 2) issue(std::integral_constant<int, 2 % D>{});
  issue(std::integral_constant<int, 0>{});
  __syncthreads();

  const int li = lane & 15;
  const int lrow = 8 * (lane >> 5) + (li >> 2), lcol = 32 * ((lane >> 4) & 1) + 8 * (li & 3);
  const int xrd = (wave * HWD + lrow) * PXB + lcol;
  const int zrd = XBN + (wave * CW + lrow) * PXB + lcol;
  // chunk k: multiply buffer k & 1; stage chunk k + 1 (slot (k + 1) % D, loaded D chunk periods ago) into the other buffer -- last
  // read in chunk k - 1, before the barrier that ended it -- and refill the slot with chunk k + 1 + D
  auto body = [&](auto slot_, int k) {
    const unsigned char* const Bb = lds + (k & 1) * BUF4;
    uint4 bz[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const uint2 lo = lds_tr16(Bb + zrd + p * ZP3), hi = lds_tr16(Bb + zrd + p * ZP3 + 4 * PXB);
      bz[p] = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
    if constexpr (NP == 3) {
      // exact split: one tap at a time -- three A planes (12 registers) and the six products of that tap's accumulator in the third
      // generation's order (smallest first: the same sums, bit for bit).  A dependent chain on one accumulator issues back to back on
      // gfx950 (scripts/ubench/mfma_bf16_chain.hip), and keeping one tap's planes live instead of a row's 36 registers is what lets the
      // ring's two register slots fit beside 144 accumulator registers
      constexpr int PA3[6] = {2, 0, 1, 1, 0, 0}, PB3[6] = {0, 2, 1, 0, 1, 0};
#if FP_W3_INTERLEAVE
      // The staging of the NEXT chunk (slot `slot_`, into the other LDS buffer) in the shadow of this chunk's MFMAs: a wave issues in order,
      // so VALU work placed BEHIND the 54 MFMAs runs after them (one wave per SIMD: nothing else fills the matrix pipe's shadow).  The split
      // of the six staged float4 is cut into 18 parts (per item: h + store + residual; m + store + residual; l + store) and one part follows
      // every three MFMAs, pinned by scheduling fences -- the compiler's own pipeline solver gives up on this region (round 3, and again
      // with (MFMA 1, VALU 3) x 54 in round 5: every MFMA first, every VALU behind).  Same values, same stores, same sums.
      constexpr int sl_ = decltype(slot_)::value;
      unsigned char* const sbase = lds + ((k + 1) & 1) * BUF4;
      const float live_ = k + 1 < cnt ? bias_on : 0.f;
      f32x4 sres = {0.f, 0.f, 0.f, 0.f};
      auto stage_part = [&](int i) {                  // i = 3 * item + phase; items 0..3 = X, 4..5 = dZ
        const int j = i / 3, ph = i % 3;
        unsigned char* dst = j < 4 ? sbase + (t + 256 * j) * 8 : sbase + XBN + (t + 256 * (j - 4)) * 8;
        const int pstride = j < 4 ? XP4 : ZP3;
        if (ph == 0) {
          const float4 q4 = j < 4 ? xr[sl_][j < 4 ? j : 0] : zv[sl_][j >= 4 ? j - 4 : 0];
          if (j >= 4) { bs[0] = fmaf(q4.x, live_, bs[0]); bs[1] = fmaf(q4.y, live_, bs[1]); bs[2] = fmaf(q4.z, live_, bs[2]); bs[3] = fmaf(q4.w, live_, bs[3]); }
          sres = f32x4{q4.x, q4.y, q4.z, q4.w};
          *reinterpret_cast<uint2*>(dst) = split_level(sres);
        } else if (ph == 1) {
          *reinterpret_cast<uint2*>(dst + pstride) = split_level(sres);
        } else {
          *reinterpret_cast<uint2*>(dst + 2 * pstride) = split_last(sres);
        }
      };
#endif
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          uint4 a3[3];
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            const unsigned char* src = Bb + xrd + (ky * HWD + kx) * PXB + p * XP4;
            const uint2 lo = lds_tr16(src), hi = lds_tr16(src + 4 * PXB);
            a3[p] = make_uint4(lo.x, lo.y, hi.x, hi.y);
          }
#pragma unroll
          for (int qq = 0; qq < 6; ++qq) {
            acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a3[PA3[qq]]), __builtin_bit_cast(bf16x8, bz[PB3[qq]]),
                                                                       acc[ky * 3 + kx], 0, 0, 0);
#if FP_W3_INTERLEAVE
            {                                        // a part behind every FP_W3_INTERLEAVE-th MFMA (3: evenly over the 54; 2: over the first 36)
              constexpr int every = FP_W3_INTERLEAVE == 1 ? 3 : (FP_W3_INTERLEAVE == 4 ? 2 : FP_W3_INTERLEAVE);
              const int done = (ky * 3 + kx) * 6 + qq + 1;
              if (done % every == 0 && done / every <= 18) {
                stage_part(done / every - 1);
                __builtin_amdgcn_sched_barrier(0);
              }
#if FP_W3_INTERLEAVE == 4
              // variant: parts behind every second MFMA (all staged after 36), then the NEXT-next chunk's address math and loads (issue())
              // behind MFMA 38 -- in the shadow of the last 16 MFMAs instead of behind them; border chunks' ~300 VALU of reflection / clamp math too
              if (done == 38) {
                issue(slot_);
                __builtin_amdgcn_sched_barrier(0);
              }
#endif
            }
#endif
          }
        }
#if FP_W3_INTERLEAVE
      __builtin_amdgcn_sched_barrier(0);
#if FP_W3_INTERLEAVE != 4
      issue(slot_);
#endif
      __syncthreads();
      return;
#endif
    } else {
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      uint4 af[3][2];                                // [kx][plane]
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const unsigned char* src = Bb + xrd + (ky * HWD + kx) * PXB + p * XP4;
          const uint2 lo = lds_tr16(src), hi = lds_tr16(src + 4 * PXB);
          af[kx][p] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
      constexpr int PA[4] = {1, 1, 0, 0}, PB[4] = {1, 0, 1, 0};          // smallest products first
#pragma unroll
      for (int qq = 4 - FP_HP_PRODUCTS; qq < 4; ++qq)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[kx][PA[qq]]), __builtin_bit_cast(f16x8, bz[PB[qq]]),
                                                                    acc[ky * 3 + kx], 0, 0, 0);
#if FP_W3_INTERLEAVE
          {                                          // fp16 pairs: one staged float4 (split + two stores) behind every fourth of the 9 x FP_HP_PRODUCTS MFMAs
            constexpr int sl2 = decltype(slot_)::value;
            const int done = (ky * FP_HP_PRODUCTS + (qq - (4 - FP_HP_PRODUCTS))) * 3 + kx + 1;
            if (done % 4 == 0 && done / 4 <= 6) {
              const int j = done / 4 - 1;
              unsigned char* const sb2 = lds + ((k + 1) & 1) * BUF4;
              const float live2 = k + 1 < cnt ? bias_on : 0.f;
              if (j < 4) {
                const float4 q4 = xr[sl2][j < 4 ? j : 0];
                split_store_np<2>(sb2 + (t + 256 * j) * 8, XP4, f32x4{q4.x, q4.y, q4.z, q4.w}, kx_);
              } else {
                const float4 q4 = zv[sl2][j >= 4 ? j - 4 : 0];
                bs[0] = fmaf(q4.x, live2, bs[0]); bs[1] = fmaf(q4.y, live2, bs[1]); bs[2] = fmaf(q4.z, live2, bs[2]); bs[3] = fmaf(q4.w, live2, bs[3]);
                split_store_np<2>(sb2 + XBN + (t + 256 * (j - 4)) * 8, ZP3, f32x4{q4.x, q4.y, q4.z, q4.w}, kz_);
              }
              __builtin_amdgcn_sched_barrier(0);
            }
          }
#endif
        }
    }
#if FP_W3_INTERLEAVE
    if (FP_HP_PRODUCTS * 9 >= 24) {                  // all six items were staged between the MFMAs
      __builtin_amdgcn_sched_barrier(0);
      issue(slot_);
      __syncthreads();
      return;
    }
#endif
    }
    // nothing of the staging moves up among the MFMAs: its first instruction waits for the slot's loads, and every MFMA issued
    // before that wait is time the loads have to land (D chunk periods instead of D - 1)
    __builtin_amdgcn_sched_barrier(0);
    stage(slot_, (k + 1) & 1, k + 1 < cnt ? bias_on : 0.f);
    issue(slot_);
    __syncthreads();
  };
  int k = 0;
  if (D == 2) {
    for (; k + 2 <= cnt; k += 2) {
      body(std::integral_constant<int, 1 % D>{}, k);
      body(std::integral_constant<int, 0>{}, k + 1);
    }
    if (k < cnt) body(std::integral_constant<int, 1 % D>{}, k);
  } else {
    for (; k + 3 <= cnt; k += 3) {
      body(std::integral_constant<int, 1 % D>{}, k);
      body(std::integral_constant<int, 2 % D>{}, k + 1);
      body(std::integral_constant<int, 0>{}, k + 2);
    }
    if (k < cnt) body(std::integral_constant<int, 1 % D>{}, k);
    if (k + 1 < cnt) body(std::integral_constant<int, 2 % D>{}, k + 1);
  }

  // ---- sum the four waves' tiles through LDS (fixed order), one tap at a time -----------------------------------------------------
  float* red = reinterpret_cast<float*>(lds);        // [4 waves][16 regs][64 lanes] = 16 KB
  float* out = a.part + (size_t)s * 9 * a.C * a.Nout;
  if (want_bias) {                                   // 32 staging threads per channel quad -> one partial per output channel
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) red[(t >> 3) * 32 + q * 4 + kk] = bs[kk];
    __syncthreads();
    if (t < 32) {
      float v = 0.f;
#pragma unroll
      for (int g = 0; g < 32; ++g) v += red[g * 32 + t];
      a.bpart[(size_t)s * a.Nout + co0 + t] = v;
    }
    __syncthreads();
  }
  const float unscale = NP == 2 ? ldexpf(1.f, -(kx_ + kz_)) : 1.f;
#pragma unroll
  for (int tp = 0; tp < 9; ++tp) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[tp][r];
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int e = t + 256 * kk;
      float v = ((red[e] + red[1024 + e]) + red[2048 + e]) + red[3072 + e];
      if (NP == 2) v *= unscale;
      const int r = e >> 6, ln = e & 63;
      const int ci = ci0 + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5);
      out[((size_t)tp * a.C + ci) * a.Nout + co0 + (ln & 31)] = v;
    }
    __syncthreads();
  }
}

// (A fifth generation -- wave-specialised: three MFMA waves owning the taps of one ky for all four rows of a chunk with 48 accumulator
// registers each, the fourth wave loading and converting the X halo -- was built in round 4, passed every weight-gradient and network parity
// test, and measured SLOWER: 164.7 vs 133.3 us on 64 -> 64 @ 96 x 320, 72.5 vs 58.8 on 256 -> 256 @ 12 x 40, step 12.35 vs 11.98 ms.  One
// wave converting 14 float4 per chunk is a longer serial chain than the 36 MFMAs it feeds: the conversion parallelises over four waves
// better than it hides behind three.  Commit 8701f82 has the kernel, profiles/round4_notes.md the table.)

// dW_oihw[n][k_begin + k][tap] (+)= sum_s part[s][tap][k][n] and, in the SAME launch (tail blocks), db[n] (+)= sum_s bpart[s][n]: one
// dependent launch instead of two behind every weight-gradient kernel (56 per training step).  Fixed combination order.
__global__ void __launch_bounds__(256) wgrad_reduce_bias_kernel(const float* __restrict__ part, float* __restrict__ dw, int S, int Kc, int Nout,
                                                                int accumulate, int kc_total, int k_begin, int main_blocks,
                                                                const float* __restrict__ bpart, float* __restrict__ db) {
  if ((int)blockIdx.x >= main_blocks) {
    const int n = ((int)blockIdx.x - main_blocks) * 256 + threadIdx.x;
    if (n >= Nout) return;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    int s = 0;
    for (; s + 3 < S; s += 4) {
      v0 += bpart[(size_t)s * Nout + n]; v1 += bpart[(size_t)(s + 1) * Nout + n];
      v2 += bpart[(size_t)(s + 2) * Nout + n]; v3 += bpart[(size_t)(s + 3) * Nout + n];
    }
    for (; s < S; ++s) v0 += bpart[(size_t)s * Nout + n];
    const float v = (v0 + v1) + (v2 + v3);
    db[n] = accumulate ? db[n] + v : v;
    return;
  }
  const size_t total = (size_t)9 * Kc * Nout;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)main_blocks * 256) {
    float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f, p4 = 0.f, p5 = 0.f, p6 = 0.f, p7 = 0.f;
    int s = 0;
    for (; s + 8 <= S; s += 8) {
      const float* q = part + (size_t)s * total + e;
      p0 += q[0]; p1 += q[total]; p2 += q[2 * total]; p3 += q[3 * total];
      p4 += q[4 * total]; p5 += q[5 * total]; p6 += q[6 * total]; p7 += q[7 * total];
    }
    for (; s < S; ++s) p0 += part[(size_t)s * total + e];
    const float sum = ((p0 + p1) + (p2 + p3)) + ((p4 + p5) + (p6 + p7));
    const int n = (int)(e % Nout);
    const size_t r = e / Nout;
    const int kk = (int)(r % Kc), tap = (int)(r / Kc);
    const size_t o = ((size_t)n * kc_total + k_begin + kk) * 9 + tap;
    dw[o] = accumulate ? dw[o] + sum : sum;
  }
}

// The same sums (same combination order, bit for bit) for large weight tensors, written through an LDS transpose: the partial tiles
// are [tap][k][n] (n fastest) while dW is OIHW ([n][k][tap]), so the kernel above writes every element into a cache line of its own
// (stride 36 kc_total bytes) -- 590 k scattered 4-byte writes for 256 -> 256.  Here a workgroup owns 32 output x 8 input channels x
// 9 taps: 128-byte reads along n, 288-byte runs along (k, tap) on the way out.
__global__ void __launch_bounds__(256) wgrad_reduce_bias_t_kernel(const float* __restrict__ part, float* __restrict__ dw, int S, int Kc, int Nout,
                                                                  int accumulate, int kc_total, int k_begin, int main_blocks,
                                                                  const float* __restrict__ bpart, float* __restrict__ db) {
  if ((int)blockIdx.x >= main_blocks) {
    const int n = ((int)blockIdx.x - main_blocks) * 256 + threadIdx.x;
    if (n >= Nout) return;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    int s = 0;
    for (; s + 3 < S; s += 4) {
      v0 += bpart[(size_t)s * Nout + n]; v1 += bpart[(size_t)(s + 1) * Nout + n];
      v2 += bpart[(size_t)(s + 2) * Nout + n]; v3 += bpart[(size_t)(s + 3) * Nout + n];
    }
    for (; s < S; ++s) v0 += bpart[(size_t)s * Nout + n];
    const float v = (v0 + v1) + (v2 + v3);
    db[n] = accumulate ? db[n] + v : v;
    return;
  }
  __shared__ float tile[32][73];                       // [n][k * 9 + tap], odd row length: conflict-free both ways
  const int nblocks = Nout >> 5;
  const int n0 = ((int)blockIdx.x % nblocks) * 32, kk0 = ((int)blockIdx.x / nblocks) * 8;
  const int ln = threadIdx.x & 31, g = threadIdx.x >> 5;
  const size_t total = (size_t)9 * Kc * Nout;
#pragma unroll 3                                       // few splits (S = 1 .. 4 on the deep levels): three pairs' loads in flight together
  for (int i = 0; i < 9; ++i) {
    const int pair = g + 8 * i, tap = pair >> 3, kk = pair & 7;
    const float* q0 = part + ((size_t)tap * Kc + kk0 + kk) * Nout + n0 + ln;
    float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f, p4 = 0.f, p5 = 0.f, p6 = 0.f, p7 = 0.f;
    int s = 0;
    for (; s + 8 <= S; s += 8) {
      const float* q = q0 + (size_t)s * total;
      p0 += q[0]; p1 += q[total]; p2 += q[2 * total]; p3 += q[3 * total];
      p4 += q[4 * total]; p5 += q[5 * total]; p6 += q[6 * total]; p7 += q[7 * total];
    }
    for (; s < S; ++s) p0 += q0[(size_t)s * total];
    tile[ln][kk * 9 + tap] = ((p0 + p1) + (p2 + p3)) + ((p4 + p5) + (p6 + p7));
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int idx = threadIdx.x + 256 * i, n = idx / 72, c = idx - n * 72;
    const size_t o = ((size_t)(n0 + n) * kc_total + k_begin + kk0) * 9 + c;
    dw[o] = accumulate ? dw[o] + tile[n][c] : tile[n][c];
  }
}

// db[n] (+)= sum_s bpart[s][n], fixed order
__global__ void __launch_bounds__(256) wgrad_bias_reduce_kernel(const float* __restrict__ bpart, int S, int Nout, float* __restrict__ db,
                                                                int accumulate) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= Nout) return;
  float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
  int s = 0;
  for (; s + 3 < S; s += 4) {
    v0 += bpart[(size_t)s * Nout + n]; v1 += bpart[(size_t)(s + 1) * Nout + n];
    v2 += bpart[(size_t)(s + 2) * Nout + n]; v3 += bpart[(size_t)(s + 3) * Nout + n];
  }
  for (; s < S; ++s) v0 += bpart[(size_t)s * Nout + n];
  const float v = (v0 + v1) + (v2 + v3);
  db[n] = accumulate ? db[n] + v : v;
}

bool eligible(const fp_conv_desc* d) {
  if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != 1 || d->C1 != 0) return false;
  if (d->gather != FP_GATHER_FWD_ZERO && d->gather != FP_GATHER_FWD_REFLECT && d->gather != FP_GATHER_FWD_REFLECT_UP2) return false;
  if (d->gather == FP_GATHER_FWD_REFLECT_UP2 && (d->OH % 2 || d->OW % 2)) return false;
  if (d->OH != d->IH || d->OW != d->IW || d->IH < 2 || d->IW < 2) return false;
  if (d->C0 % 32 || d->Nout % 32) return false;
  const int64_t cy = fp_ceil_div(d->OH, CH), cx = fp_ceil_div(d->OW, CW);
  // padded work up to 2.2x is still ahead of the flattened fp32 kernel (6 x 20 levels pad to 8 x 32 = 2.13x: 119 vs 145 us on 512 -> 512,
  // 63 vs 72 on 512 -> 256, 39 vs 46 on 256 -> 256 with the third-generation kernel); FP_WGRAD_BF3_TIGHT=1 restores the 30 % limit
  static const bool tight = fp_env_flag("FP_WGRAD_BF3_TIGHT");
  if (cy * CH * cx * CW * 10 > (int64_t)d->OH * d->OW * (tight ? 13 : 22)) return false;
  return (int64_t)d->N * cy * cx >= 16;
}

struct WPlan { int S, chunksPerSplit, nchunks, citiles, cotiles, cy, cx; };
WPlan plan(const fp_conv_desc* d) {
  WPlan p;
  p.cy = (int)fp_ceil_div(d->OH, CH); p.cx = (int)fp_ceil_div(d->OW, CW);
  p.nchunks = d->N * p.cy * p.cx;
  p.citiles = d->C0 / 32; p.cotiles = d->Nout / 32;
  const int64_t base = (int64_t)p.citiles * p.cotiles;
  static const int target = getenv("FP_WGRAD_TARGET_WGS") ? atoi(getenv("FP_WGRAD_TARGET_WGS")) : 256;
  // pixel splits for ~one workgroup per CU: measured in the training step 14.73 (512 workgroups: one full round of the two resident per CU),
  // 14.39 (384), 14.27 / 13.93 (256), 14.02 (192), 14.30 (128) ms -- fewer splits write and re-read fewer partial tensors, and the
  // weight gradients run on side streams beside the data-gradient chain, which gets the other half of the CUs
  int64_t S = fp_ceil_div(target, base);
  if (S > p.nchunks / 4) S = p.nchunks / 4;
  if (S < 1) S = 1;
  if (S > 512) S = 512;
  p.chunksPerSplit = (int)fp_ceil_div(p.nchunks, S);
  p.S = (int)fp_ceil_div(p.nchunks, p.chunksPerSplit);
  return p;
}

}  // namespace

// bytes of workspace, or -1 when the shape is not handled (caller uses fp_conv_wgrad)
extern "C" int64_t fp_conv_wgrad_bf3_workspace(const fp_conv_desc* d) {
  if (!d || !eligible(d)) return -1;
  const WPlan p = plan(d);
  return ((int64_t)p.S * 9 * d->C0 * d->Nout + (int64_t)p.S * d->Nout) * (int64_t)sizeof(float);
}

static int wgrad_split(const fp_conv_desc* d, const float* x, const float* dz, float* dw_oihw, float* db, int32_t kc_total, int32_t k_begin,
                       int accumulate, void* workspace, int64_t workspace_bytes, const uint32_t* amax_x, const uint32_t* amax_dz,
                       fp_stream_t stream_);
extern "C" int fp_conv_wgrad_bf3(const fp_conv_desc* d, const float* x, const float* dz, float* dw_oihw, float* db, int32_t kc_total,
                                 int32_t k_begin, int accumulate, void* workspace, int64_t workspace_bytes, fp_stream_t stream_) {
  return wgrad_split(d, x, dz, dw_oihw, db, kc_total, k_begin, accumulate, workspace, workspace_bytes, nullptr, nullptr, stream_);
}
// fp16-pair operands (fp_conv3x3_hp): `amax_x` / `amax_dz` = amax slots of the two tensors; same shapes, workspace and results layout
extern "C" int fp_conv_wgrad_hp(const fp_conv_desc* d, const float* x, const float* dz, float* dw_oihw, float* db, int32_t kc_total,
                                int32_t k_begin, int accumulate, void* workspace, int64_t workspace_bytes, const uint32_t* amax_x,
                                const uint32_t* amax_dz, fp_stream_t stream_) {
  FP_REQUIRE(amax_x && amax_dz, "fp_conv_wgrad_hp: amax slots missing");
  return wgrad_split(d, x, dz, dw_oihw, db, kc_total, k_begin, accumulate, workspace, workspace_bytes, amax_x, amax_dz, stream_);
}
static int wgrad_split(const fp_conv_desc* d, const float* x, const float* dz, float* dw_oihw, float* db, int32_t kc_total, int32_t k_begin,
                       int accumulate, void* workspace, int64_t workspace_bytes, const uint32_t* amax_x, const uint32_t* amax_dz,
                       fp_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  FP_REQUIRE(d && x && dz && dw_oihw && workspace, "fp_conv_wgrad_bf3: null pointer");
  FP_REQUIRE(eligible(d), "fp_conv_wgrad_bf3: shape not supported (see fp_conv_wgrad_bf3_workspace)");
  FP_REQUIRE(k_begin >= 0 && k_begin + d->C0 <= kc_total, "fp_conv_wgrad_bf3: input-channel slice out of range");
  const WPlan p = plan(d);
  FP_REQUIRE(workspace_bytes >= fp_conv_wgrad_bf3_workspace(d), "fp_conv_wgrad_bf3: workspace too small");
  W3Args a;
  a.amax_x = amax_x; a.amax_dz = amax_dz;
  const bool hp = amax_x != nullptr;
  a.x = x; a.dz = dz; a.part = (float*)workspace;
  a.bpart = db ? (float*)workspace + (size_t)p.S * 9 * d->C0 * d->Nout : nullptr;
  a.N = d->N; a.H = d->OH; a.W = d->OW; a.C = d->C0; a.Nout = d->Nout;
  a.mode = d->gather == FP_GATHER_FWD_ZERO ? 0 : (d->gather == FP_GATHER_FWD_REFLECT_UP2 ? 2 : 1);
  a.chunksY = p.cy; a.chunksX = p.cx; a.nchunks = p.nchunks; a.chunksPerSplit = p.chunksPerSplit; a.S = p.S;
  a.citiles = p.citiles; a.cotiles = p.cotiles;
  static const bool no_xcd = fp_env_flag("FP_WGRAD_NO_XCD");
  a.xcd = no_xcd ? 0 : 1;
  static const bool no_fast = fp_env_flag("FP_WGRAD_NO_FAST");
  a.nofast = no_fast ? 1 : 0;
  a.stamps = nullptr;
  static const char* stamp_file = getenv("FP_W3_STAMPS");
  static unsigned long long* stamp_buf = nullptr;
  const int nwg = p.S * p.citiles * p.cotiles;
  if (stamp_file) {
    if (!stamp_buf) (void)hipMalloc(&stamp_buf, (size_t)8192 * 6 * 8);
    if (nwg <= 8192) a.stamps = stamp_buf;
  }
  // FP_WGRAD_PF: 0 = third generation; 1, 2 = the ring kernel with two slots (the default); 3 = three slots (308 VGPRs: no faster alone,
  // slower in the step -- it no longer shares a SIMD with the small-grid tile kernels; profiles/round3_notes.md)
  static const int pf = getenv("FP_WGRAD_PF") ? atoi(getenv("FP_WGRAD_PF")) : FP_WGRAD_PF_DEFAULT;
  const int64_t xbytes = (int64_t)d->N * (a.mode == 2 ? (d->OH / 2) * (int64_t)(d->OW / 2) : (int64_t)d->OH * d->OW) * d->C0 * 4;
  const int64_t zbytes = (int64_t)d->N * d->OH * d->OW * d->Nout * 4;
  const bool pf_fits = xbytes < (int64_t(1) << 31) && zbytes < (int64_t(1) << 31);       // 32-bit buffer offsets, bit 31 = "out of range"
  if (hp && pf >= 1 && pf_fits && !a.stamps) {
#define FP_W3_PF_LAUNCH(MODE_)                                                                                              \
    do {                                                                                                                     \
      if (pf <= 2) fp_launch((wgrad3x3_hp_pf_kernel<MODE_, 2>), dim3(nwg), dim3(256), 0, stream, a);                        \
      else fp_launch((wgrad3x3_hp_pf_kernel<MODE_, 3>), dim3(nwg), dim3(256), 0, stream, a);                                \
    } while (0)
    if (a.mode == 0) FP_W3_PF_LAUNCH(0);
    else if (a.mode == 1) FP_W3_PF_LAUNCH(1);
    else FP_W3_PF_LAUNCH(2);
#undef FP_W3_PF_LAUNCH
  } else if (!hp && pf >= 1 && pf_fits && !a.stamps && a.mode != 2) {
    // exact split on the ring (round 5); FP_WGRAD_PF=0: third generation.  The nearest-x2 gather form (two launches per step) stays on the
    // third generation: its border-chunk address math does not fit beside three operand planes without 12 bytes of scratch
    if (a.mode == 0) fp_launch((wgrad3x3_hp_pf_kernel<0, 2, 3>), dim3(nwg), dim3(256), 0, stream, a);
    else fp_launch((wgrad3x3_hp_pf_kernel<1, 2, 3>), dim3(nwg), dim3(256), 0, stream, a);
  } else if (hp) {
    if (a.mode == 0) fp_launch((wgrad3x3_bf3_v3_kernel<0, 2>), dim3(nwg), dim3(256), 0, stream, a);
    else if (a.mode == 1) fp_launch((wgrad3x3_bf3_v3_kernel<1, 2>), dim3(nwg), dim3(256), 0, stream, a);
    else fp_launch((wgrad3x3_bf3_v3_kernel<2, 2>), dim3(nwg), dim3(256), 0, stream, a);
  } else if (a.mode == 0) fp_launch((wgrad3x3_bf3_v3_kernel<0, 3>), dim3(nwg), dim3(256), 0, stream, a);
  else if (a.mode == 1) fp_launch((wgrad3x3_bf3_v3_kernel<1, 3>), dim3(nwg), dim3(256), 0, stream, a);
  else fp_launch((wgrad3x3_bf3_v3_kernel<2, 3>), dim3(nwg), dim3(256), 0, stream, a);
  int rc = fp_check_launch("fp_conv_wgrad_bf3");
  if (rc) return rc;
  if (a.stamps) {                                   // debugging only: synchronous dump of the last launch's stamps
    (void)hipStreamSynchronize(stream);
    unsigned long long* h = (unsigned long long*)malloc((size_t)nwg * 6 * 8);
    (void)hipMemcpy(h, stamp_buf, (size_t)nwg * 6 * 8, hipMemcpyDeviceToHost);
    FILE* f = fopen(stamp_file, "w");
    if (f) {
      for (int i = 0; i < nwg; ++i) fprintf(f, "%d %llu %llu %llu %llu %llu %llu\n", i, h[i * 6], h[i * 6 + 1], h[i * 6 + 2], h[i * 6 + 3], h[i * 6 + 4], h[i * 6 + 5]);
      fclose(f);
    }
    free(h);
  }
  static const bool split_reduce = fp_env_flag("FP_WGRAD_SPLIT_REDUCE");     // A/B switch: the two former reduce launches
  if (split_reduce) {
    if (db) {
      fp_launch(wgrad_bias_reduce_kernel, dim3((d->Nout + 255) / 256), dim3(256), 0, stream, (const float*)a.bpart, p.S, d->Nout, db,
                         accumulate);
      rc = fp_check_launch("fp_conv_wgrad_bf3(bias)");
      if (rc) return rc;
    }
    return fp_wgrad_reduce_launch((const float*)workspace, dw_oihw, p.S, 9, d->C0, d->Nout, 0, accumulate, kc_total, k_begin, stream);
  }
  const int64_t total = (int64_t)9 * d->C0 * d->Nout;
  const int bias_blocks = db ? (d->Nout + 255) / 256 : 0;
  // transposing reduce from FP_WGRAD_REDUCE_T_MIN workgroups (0 = never).  Measured: 512 -> 512 @ 6 x 20: 111 -> 90 us per weight gradient,
  // but 256 -> 256 (256 workgroups of it, S = 4): 60 -> 63 us
  static const int t_min = getenv("FP_WGRAD_REDUCE_T_MIN") ? atoi(getenv("FP_WGRAD_REDUCE_T_MIN")) : 512;
  const int t_blocks = (d->Nout / 32) * (d->C0 / 8);
  if (t_min > 0 && t_blocks >= t_min) {
    fp_launch(wgrad_reduce_bias_t_kernel, dim3(t_blocks + bias_blocks), dim3(256), 0, stream, (const float*)workspace, dw_oihw, p.S,
                       d->C0, d->Nout, accumulate, kc_total, k_begin, t_blocks, (const float*)a.bpart, db);
    return fp_check_launch("fp_conv_wgrad_bf3(reduce)");
  }
  int main_blocks = (int)fp_ceil_div(total, 256);
  if (main_blocks > 4096) main_blocks = 4096;
  fp_launch(wgrad_reduce_bias_kernel, dim3(main_blocks + bias_blocks), dim3(256), 0, stream, (const float*)workspace, dw_oihw, p.S,
                     d->C0, d->Nout, accumulate, kc_total, k_begin, main_blocks, (const float*)a.bpart, db);
  return fp_check_launch("fp_conv_wgrad_bf3(reduce)");
}
