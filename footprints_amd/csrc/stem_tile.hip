// Stem convolution 7x7 / stride 2 / pad 3 over the NCHW image, (x - 0.45) / 0.225 applied before the zero padding
// (reference footprints/network.py:48-52 + torchvision resnet conv1), fp32 MFMA (v_mfma_f32_32x32x2_f32).
//
// The flattened kernel (conv_igemm.hip, STEM) gathers every A element from global memory, four scalar loads per float4 of its
// K-step: 37 TFLOP/s.  Here a workgroup owns 8 x 16 output pixels x 64 channels and stages the 21 x 37 x 3 input patch it needs
// ONCE, normalised and zero-padded, in LDS (10 KB); the A operand of every K-step is then one ds_read_b32 per lane
// (pixel = lane & 31, k = 2 * step + (lane >> 5)), the B operand comes from the FP_PACK_STEM weights [10][64][16] (k = (ky*7 + kx)*3 + ci,
// zero beyond 147).  Same arithmetic per element as the flattened kernel (fp32 products, fp32 MFMA accumulation, k ascending).
#include <stdlib.h>

#include "fp_common.h"

namespace {

struct StemArgs {
  const float* img;    // [N][3][IH][IW]
  const float* w;      // FP_PACK_STEM
  const float* bias;   // optional (folded BatchNorm shift)
  float* y;            // [N][OH][OW][64]
  int N, IH, IW, OH, OW, tilesX, tilesY, act;
  float* bn_part;      // train-mode BatchNorm behind the stem (no bias, no activation): Welford partials [pixel tile][64][3], or null
};

constexpr int TH = 8, TW = 16, PH = 2 * TH + 5, PW = 2 * TW + 5, PWS = 40;   // 21 x 37 patch, row stride 40

__global__ void __launch_bounds__(256) stem_tile_kernel(const StemArgs a) {
  __shared__ float P[3 * PH * PWS];
  __shared__ __attribute__((aligned(16))) int KT[2 * 80];      // patch offset of k = 2 * step + h, [h][step]; -1 beyond the 147 taps
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, idx = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  if (t < 160) {                                    // (ky, kx, ci) of every K index once per workgroup: two divisions per K-step and lane otherwise
    const int hh = t / 80, st = t - hh * 80, kk = 2 * st + hh;
    const int ky = kk / 21, rem = kk - ky * 21, kx = rem / 3, ci = rem - kx * 3;
    KT[t] = kk < 147 ? (ci * PH + ky) * PWS + (kx & 1) * (PWS / 2) + (kx >> 1) : -1;
  }
  int b = blockIdx.x;
  const int tx = b % a.tilesX; b /= a.tilesX;
  const int ty = b % a.tilesY;
  const int n = b / a.tilesY;
  const int y0 = ty * TH, x0 = tx * TW;
  for (int e = t; e < 3 * PH * PW; e += 256) {
    const int ci = e / (PH * PW), r = e - ci * (PH * PW);
    const int py = r / PW, px = r - py * PW;
    const int iy = 2 * y0 + py - 3, ix = 2 * x0 + px - 3;
    float v = 0.f;
    if (iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW) v = (a.img[((size_t)(n * 3 + ci) * a.IH + iy) * a.IW + ix] - 0.45f) / 0.225f;
    P[(ci * PH + py) * PWS + (px & 1) * (PWS / 2) + (px >> 1)] = v;      // columns de-interleaved by parity (see below)
  }
  __syncthreads();

  int pbase[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int pt = (wm * 2 + i) * 32 + idx;
    pbase[i] = (2 * (pt / TW)) * PWS + (pt % TW);        // patch column 2 * px + kx lives at (kx & 1) * 20 + px + (kx >> 1)
  }
  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const float* wl = a.w + (size_t)(wn * 32 + idx) * 16;
#pragma unroll 1
  for (int cc = 0; cc < 10; ++cc) {
    const float4 w0 = *reinterpret_cast<const float4*>(wl + (size_t)cc * 64 * 16);
    const float4 w1 = *reinterpret_cast<const float4*>(wl + (size_t)cc * 64 * 16 + 4);
    const float4 w2 = *reinterpret_cast<const float4*>(wl + (size_t)cc * 64 * 16 + 8);
    const float4 w3 = *reinterpret_cast<const float4*>(wl + (size_t)cc * 64 * 16 + 12);
    // lane-parity select of the weight pair of every K-step by bit-select (a ?: over an array element was lowered to scratch)
    const unsigned hm = 0u - (unsigned)h;
    const int4 ko0 = *reinterpret_cast<const int4*>(KT + h * 80 + cc * 8), ko1 = *reinterpret_cast<const int4*>(KT + h * 80 + cc * 8 + 4);
    const int kos[8] = {ko0.x, ko0.y, ko0.z, ko0.w, ko1.x, ko1.y, ko1.z, ko1.w};
    auto step = [&](int s, float we, float wo) {
      const int koff = kos[s];                         // adjacent lanes -> adjacent banks (stride-2 reads conflicted)
      const float bv = __uint_as_float((__float_as_uint(wo) & hm) | (__float_as_uint(we) & ~hm));   // zero beyond the 147 taps (packed zeros)
      const int kq = max(koff, 0);
      const float a0 = koff >= 0 ? P[pbase[0] + kq] : 0.f, a1 = koff >= 0 ? P[pbase[1] + kq] : 0.f;
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv, acc[1], 0, 0, 0);
    };
    step(0, w0.x, w0.y); step(1, w0.z, w0.w); step(2, w1.x, w1.y); step(3, w1.z, w1.w);
    step(4, w2.x, w2.y); step(5, w2.z, w2.w); step(6, w3.x, w3.y); step(7, w3.z, w3.w);
  }
  const int co = wn * 32 + idx;
  const float bias = a.bias ? a.bias[co] : 0.f;
  if (a.bn_part) {
    // BatchNorm statistics out of the epilogue, as in conv3x3_tile_bf3.hip (round 4: the statistics pass over the 94 MB activation was 35-50 us
    // at the very start of the step, alone on the GPU): a lane's 32 pixels of channel `co` -> two-pass (count, mean, M2) -> Chan merge with the
    // other half-wave -> across the two waves that share the channel through LDS, fixed order -> part[pixel tile][co]
    auto ok_at = [&](int i, int r) {
      const int pt = (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      return y0 + pt / TW < a.OH && x0 + pt % TW < a.OW;
    };
    float cnt = 0.f, sum = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bool ok = ok_at(i, r);
        cnt += ok ? 1.f : 0.f;
        sum += ok ? acc[i][r] : 0.f;
      }
    FpWf w{cnt, cnt > 0.f ? sum / cnt : 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float dv = acc[i][r] - w.mean;
        w.m2 += ok_at(i, r) ? dv * dv : 0.f;
      }
    const FpWf o{__shfl_xor(w.n, 32, 64), __shfl_xor(w.mean, 32, 64), __shfl_xor(w.m2, 32, 64)};
    FpWf lo = h == 0 ? w : o;                            // both half-waves form merge(h = 0, h = 1)
    fp_wf_merge(lo, h == 0 ? o : w);
    __syncthreads();                                     // every wave is done with the patch
    float* st = P;                                       // [wave][32][3]
    if (h == 0) {
      float* q = st + (wave * 32 + idx) * 3;
      q[0] = lo.n; q[1] = lo.mean; q[2] = lo.m2;
    }
    __syncthreads();
    if (t < 64) {                                        // channel t: waves (wm = 0, wn) then (wm = 1, wn)
      const float* q0 = st + ((t >> 5) * 32 + (t & 31)) * 3;
      const float* q1 = st + ((2 + (t >> 5)) * 32 + (t & 31)) * 3;
      FpWf m{q0[0], q0[1], q0[2]};
      fp_wf_merge(m, FpWf{q1[0], q1[1], q1[2]});
      float* out = a.bn_part + ((size_t)blockIdx.x * 64 + t) * 3;
      out[0] = m.n; out[1] = m.mean; out[2] = m.m2;
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int pt = (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      const int oy = y0 + pt / TW, ox = x0 + pt % TW;
      if (oy >= a.OH || ox >= a.OW) continue;
      float v = acc[i][r] + bias;
      if (a.act == FP_ACT_RELU) v = fmaxf(v, 0.f);
      else if (a.act == FP_ACT_ELU) v = fp_elu(v);
      a.y[((size_t)(n * a.OH + oy) * a.OW + ox) * 64 + co] = v;
    }
}

// ---- the stem with fp16-pair operands (round 4) -------------------------------------------------------------------------------------
// v_mfma_f32_32x32x2_f32 spends 64 cycles on two K indices; the fp16-pair form spends 3 x 32 cycles on sixteen: 66 MFMAs per wave and tile
// instead of 160 at half the cycles each.  K is re-indexed so that a lane's eight consecutive K indices are eight consecutive halves of the
// patch: k' = ky * 24 + (kx * 3 + ci) -- one ROW of the 7 x 7 x 3 window is the 21 consecutive elements (2 px + kx) * 3 + ci of an image row
// stored [column][channel]; three zero weights pad the row to 24 = 3 x 8, so every group of eight stays inside one row -- 168 of 176 K indices
// (11 K-steps) are real.  The patch planes (h, m of (x - 0.45) / 0.225 * 2^12; |x| <= 2.45, so the scale is a constant) are 22 rows x 120
// halves; fragments are four ds_read_b32 (the start, 12 px + 2 j0 bytes, is 4-byte aligned).  B fragments: 16 bytes per lane, K-step and plane
// straight from the FP_PACK_STEM_HP planes.
constexpr int HRS = 120, HROWS = 22, STEM_KA = 12, STEM_STEPS = 11;
typedef _Float16 st_f16x8 __attribute__((ext_vector_type(8)));
struct StemHpArgs {
  const float* img;
  const unsigned short* w;    // FP_PACK_STEM_HP
  const float* bias;
  float* y;
  const unsigned* amax_w;
  unsigned* amax_out;
  float* bn_part;
  int N, IH, IW, OH, OW, tilesX, tilesY, act;
};

__global__ void __launch_bounds__(256, 2) stem_tile_hp_kernel(const StemHpArgs a) {
  __shared__ __attribute__((aligned(16))) _Float16 Ph[HROWS * HRS];
  __shared__ __attribute__((aligned(16))) _Float16 Pm[HROWS * HRS];
  const unsigned amax_raw = fp_amax3_issue(a.amax_w, nullptr, nullptr);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, idx = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  int pbase[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int pt = (wm * 2 + i) * 32 + idx;
    pbase[i] = (2 * (pt / TW)) * HRS + 6 * (pt % TW);
  }
  // persistent workgroups (tiles blockIdx.x, + gridDim.x, ...): the lane's weights of all eleven K-steps stay in 88 registers, the amax slot and
  // the tile-independent geometry are read once -- a tile is then staging, 66 MFMAs per wave and the stores
  const unsigned short* wl = a.w + (size_t)(wn * 32 + idx) * 16 + h * 8;
  uint4 bh[STEM_STEPS], bm[STEM_STEPS];
#pragma unroll
  for (int st = 0; st < STEM_STEPS; ++st) {
    bh[st] = *reinterpret_cast<const uint4*>(wl + (size_t)(st * 2 + 0) * 64 * 16);
    bm[st] = *reinterpret_cast<const uint4*>(wl + (size_t)(st * 2 + 1) * 64 * 16);
  }
  unsigned mw, u1, u2;
  fp_amax3_reduce(amax_raw, mw, u1, u2);
  const float unscale = ldexpf(1.f, -(STEM_KA + fp_hp_exponent(mw, FP_HP_TARGET_W)));
  const int co = wn * 32 + idx;
  const float bias = a.bias ? a.bias[co] : 0.f;
  float ymax = 0.f;
  const int ntiles = a.N * a.tilesX * a.tilesY;
  // the patch of a tile: (row, column) entries t, t + 256, ... of the 22 x 40 grid, three channels each, raw image values in registers (NaN marks
  // "no source pixel": zero AFTER the normalisation) -- loaded for the NEXT tile while the current one is multiplied and stored
  constexpr int NE = (HROWS * 40 + 255) / 256;         // 4 entries per thread
  float raw[NE][3];
  auto load_patch = [&](int tile) __attribute__((always_inline)) {
    int b = tile;
    const int tx = b % a.tilesX; b /= a.tilesX;
    const int ty = b % a.tilesY;
    const int n = b / a.tilesY;
#pragma unroll
    for (int k = 0; k < NE; ++k) {
      const int e = t + 256 * k, r = e / 40, c = e - r * 40;
      const int iy = 2 * ty * TH + r - 3, ix = 2 * tx * TW + c - 3;
      const bool ok = tile < ntiles && r < PH && c < PW && iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW;
#pragma unroll
      for (int ci = 0; ci < 3; ++ci) raw[k][ci] = ok ? a.img[((size_t)(n * 3 + ci) * a.IH + iy) * a.IW + ix] : __builtin_nanf("");
    }
  };
  load_patch(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int b = tile;
    const int tx = b % a.tilesX; b /= a.tilesX;
    const int ty = b % a.tilesY;
    const int n = b / a.tilesY;
    const int y0 = ty * TH, x0 = tx * TW;
    __syncthreads();                                   // the previous tile's patch (and statistics scratch) has been read by every wave
#pragma unroll
    for (int k = 0; k < NE; ++k) {
      const int e = t + 256 * k, r = e / 40, c = e - r * 40;
      if (e >= HROWS * 40) break;
#pragma unroll
      for (int ci = 0; ci < 3; ++ci) {
        const float v = raw[k][ci] == raw[k][ci] ? (raw[k][ci] - 0.45f) / 0.225f : 0.f;
        const float sc = v * (float)(1 << STEM_KA);
        const _Float16 hh = (_Float16)sc;
        Ph[r * HRS + c * 3 + ci] = hh;
        Pm[r * HRS + c * 3 + ci] = (_Float16)(sc - (float)hh);
      }
    }
    load_patch(tile + gridDim.x);                      // in flight under the MFMAs and the stores below
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    __syncthreads();
#pragma unroll
    for (int st = 0; st < STEM_STEPS; ++st) {
      const int k0 = st * 16, k1 = st * 16 + 8;        // first K index of the lane's group: h = 0 / h = 1
      const int koff = h ? (k1 / 24) * HRS + k1 % 24 : (k0 / 24) * HRS + k0 % 24;
      uint4 ah[2], am[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const unsigned* ph = reinterpret_cast<const unsigned*>(Ph + pbase[i] + koff);
        const unsigned* pm = reinterpret_cast<const unsigned*>(Pm + pbase[i] + koff);
        ah[i] = make_uint4(ph[0], ph[1], ph[2], ph[3]);
        am[i] = make_uint4(pm[0], pm[1], pm[2], pm[3]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {                    // smallest products first, as in the tile kernel
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(st_f16x8, am[i]), __builtin_bit_cast(st_f16x8, bh[st]), acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(st_f16x8, ah[i]), __builtin_bit_cast(st_f16x8, bm[st]), acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(st_f16x8, ah[i]), __builtin_bit_cast(st_f16x8, bh[st]), acc[i], 0, 0, 0);
      }
    }
    if (a.bn_part) {                                   // statistics of what is stored (acc * unscale): as in stem_tile_kernel
      auto ok_at = [&](int i, int r) {
        const int pt = (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        return y0 + pt / TW < a.OH && x0 + pt % TW < a.OW;
      };
      float cnt = 0.f, sum = 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const bool ok = ok_at(i, r);
          cnt += ok ? 1.f : 0.f;
          sum += ok ? acc[i][r] * unscale : 0.f;
        }
      FpWf w{cnt, cnt > 0.f ? sum / cnt : 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float dv = acc[i][r] * unscale - w.mean;
          w.m2 += ok_at(i, r) ? dv * dv : 0.f;
        }
      const FpWf o{__shfl_xor(w.n, 32, 64), __shfl_xor(w.mean, 32, 64), __shfl_xor(w.m2, 32, 64)};
      FpWf lo = h == 0 ? w : o;
      fp_wf_merge(lo, h == 0 ? o : w);
      __syncthreads();                                 // every wave is done with the patch
      float* st = reinterpret_cast<float*>(Ph);        // [wave][32][3]
      if (h == 0) {
        float* q = st + (wave * 32 + idx) * 3;
        q[0] = lo.n; q[1] = lo.mean; q[2] = lo.m2;
      }
      __syncthreads();
      if (t < 64) {
        const float* q0 = st + ((t >> 5) * 32 + (t & 31)) * 3;
        const float* q1 = st + ((2 + (t >> 5)) * 32 + (t & 31)) * 3;
        FpWf m{q0[0], q0[1], q0[2]};
        fp_wf_merge(m, FpWf{q1[0], q1[1], q1[2]});
        float* out = a.bn_part + ((size_t)tile * 64 + t) * 3;
        out[0] = m.n; out[1] = m.mean; out[2] = m.m2;
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int pt = (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int oy = y0 + pt / TW, ox = x0 + pt % TW;
        if (oy >= a.OH || ox >= a.OW) continue;
        float v = fmaf(acc[i][r], unscale, bias);
        if (a.act == FP_ACT_RELU) v = fmaxf(v, 0.f);
        else if (a.act == FP_ACT_ELU) v = fp_elu(v);
        a.y[((size_t)(n * a.OH + oy) * a.OW + ox) * 64 + co] = v;
        ymax = fmaxf(ymax, fabsf(v));
      }
  }
  if (a.amax_out) fp_amax_publish_block(a.amax_out, ymax);
}

// ---- weight gradient of the stem, same patch-in-LDS scheme -------------------------------------------------------------------------
// dW[k][co] = sum over output pixels of patch(k; pixel) * dZ[pixel][co], k = (ky*7 + kx)*3 + ci: the contraction runs over pixels, so an MFMA
// K-step is two output pixels; A = one patch word per lane (row = k, the pixel from the step), B = one dZ word per lane (column = co) from the
// staged [128 pixels][64] tile.  The 5 x 2 accumulator tiles (147 -> 160 rows x 64 columns) are spread over the four waves (3, 3, 2, 2); a
// workgroup walks pixel tiles tile = id, id + grid, ... and writes ONE partial [147][64] at the end (wgrad_reduce_kernel sums them in a fixed
// order).  The flattened kernel it replaces gathered every A element with per-element index math: 264 us for the KITTI stem, alone on the GPU
// at the very end of the backward pass.
struct StemWArgs {
  const float* img;    // [N][3][IH][IW]
  const float* dz;     // [N][OH][OW][64]
  float* part;         // [gridDim.x][147][64]
  int N, IH, IW, OH, OW, tilesX, tilesY, ntiles;
};

__global__ void __launch_bounds__(256) stem_wgrad_tile_kernel(const StemWArgs a) {
  __shared__ float P[3 * PH * PWS];
  __shared__ __attribute__((aligned(16))) float Z[TH * TW * 64];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, idx = lane & 31, h = lane >> 5;
  int koff[3], ctile[3], krow0[3];
  bool kval[3], tvalid[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int id = wave + 4 * j;                       // accumulator tile id: k tile = id % 5, co tile = id / 5 (ids 10, 11 do not exist)
    tvalid[j] = id < 10;
    const int kt = id % 5;
    ctile[j] = tvalid[j] ? id / 5 : 0;
    krow0[j] = kt * 32;
    const int k = kt * 32 + idx, kc = min(k, 146);
    const int ky = kc / 21, rem = kc - ky * 21, kx = rem / 3, ci = rem - kx * 3;
    koff[j] = (ci * PH + ky) * PWS + (kx & 1) * (PWS / 2) + (kx >> 1);
    kval[j] = tvalid[j] && k < 147;
  }
  f32x16 acc[3];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    int b = tile;
    const int tx = b % a.tilesX; b /= a.tilesX;
    const int ty = b % a.tilesY;
    const int n = b / a.tilesY;
    const int y0 = ty * TH, x0 = tx * TW;
    __syncthreads();                                   // the previous tile has been read by every wave
    for (int e = t; e < 3 * PH * PW; e += 256) {
      const int ci = e / (PH * PW), r = e - ci * (PH * PW);
      const int py = r / PW, px = r - py * PW;
      const int iy = 2 * y0 + py - 3, ix = 2 * x0 + px - 3;
      float v = 0.f;
      if (iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW) v = (a.img[((size_t)(n * 3 + ci) * a.IH + iy) * a.IW + ix] - 0.45f) / 0.225f;
      P[(ci * PH + py) * PWS + (px & 1) * (PWS / 2) + (px >> 1)] = v;
    }
    for (int e = t; e < TH * TW * 16; e += 256) {
      const int pt = e >> 4, q = e & 15;
      const int oy = y0 + pt / TW, ox = x0 + pt % TW;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (oy < a.OH && ox < a.OW) v = *reinterpret_cast<const float4*>(a.dz + ((size_t)(n * a.OH + oy) * a.OW + ox) * 64 + q * 4);
      *reinterpret_cast<float4*>(Z + pt * 64 + q * 4) = v;
    }
    __syncthreads();
#pragma unroll 8
    for (int s = 0; s < TH * TW / 2; ++s) {
      const int p = 2 * s + h;
      const int pb = (2 * (p / TW)) * PWS + (p % TW);
      const float b0 = Z[p * 64 + idx], b1 = Z[p * 64 + 32 + idx];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        if (!tvalid[j]) continue;                       // wave-uniform
        const float av = kval[j] ? P[pb + koff[j]] : 0.f;
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, ctile[j] ? b1 : b0, acc[j], 0, 0, 0);
      }
    }
  }
  float* out = a.part + (size_t)blockIdx.x * 147 * 64;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    if (!tvalid[j]) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = krow0[j] + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (k < 147) out[(size_t)k * 64 + ctile[j] * 32 + idx] = acc[j][r];
    }
  }
}

// ---- ... and its weight gradient with fp16-pair operands (round 4) ---------------------------------------------------------------------
// dW[co][k] = sum over pixels of dZ[pixel][co] * patch(k; pixel): the contraction runs over pixels, so both MFMA operands have to be
// pixel-major while both tensors are channel-major in memory.  As in wgrad3x3_bf3.hip the staging keeps them channel-major in LDS and the
// fragments come out of gfx950's transposing LDS read (ds_read_b64_tr_b16: a group of sixteen lanes reads 4 pixel rows x 16 columns and every
// lane receives one column): rows of the A operand = 64 output channels (two blocks), columns of the B operand = K indices, re-indexed
// k'' = ky * 32 + kx * 4 + ci so that FOUR consecutive K indices are the 8-byte-aligned group (three channels + a zero) of ONE patch pixel --
// seven column blocks of 32 (one per kernel row; kx = 7 and ci = 3 are padding: 147 of 224 columns are real, still 3.8x fewer MFMA cycles
// than 32x32x2 fp32).  A K-step is one row of the 8 x 16 output tile.  One partial tensor [147][64] per workgroup, as the fp32 kernel.
constexpr int WRS = 152, WROWS = 21, ZPX = 72;        // halves per patch row (38 columns x 4), patch rows, halves per dZ pixel (64 + 8 pad)
struct StemWHpArgs {
  const float* img;
  const float* dz;
  float* part;
  const unsigned* amax_dz;
  int N, IH, IW, OH, OW, tilesX, tilesY, ntiles;
};
typedef short st_s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 st_tr16(const _Float16* p) {
  const st_s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) st_s16x4*)p);
  return __builtin_bit_cast(uint2, v);
}

__global__ void __launch_bounds__(256) stem_wgrad_hp_kernel(const StemWHpArgs a) {
  __shared__ __attribute__((aligned(16))) _Float16 Ph[WROWS * WRS];
  __shared__ __attribute__((aligned(16))) _Float16 Pm[WROWS * WRS];
  __shared__ __attribute__((aligned(16))) _Float16 Zh[TH * TW * ZPX];
  __shared__ __attribute__((aligned(16))) _Float16 Zm[TH * TW * ZPX];
  unsigned mz, u1, u2;
  fp_amax3_reduce(fp_amax3_issue(a.amax_dz, nullptr, nullptr), mz, u1, u2);
  const int kz = fp_hp_exponent(mz, FP_HP_TARGET_ACT);
  const float sz = ldexpf(1.f, kz);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int li = lane & 15, r4 = li >> 2, q = li & 3, b4 = (lane >> 4) & 1, kg = lane >> 5;
  const int nky = wave + 4 < 7 ? 2 : 1;               // this wave's kernel rows: wave, wave + 4
  f32x16 acc[2][2];                                    // [kernel row][channel block]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][m][r] = 0.f;
  // lane offsets of the transposing reads (halves): dZ pixel 8 kg + r4 of a tile row, columns 16 b4 + 4 q of a channel block; patch pixel
  // likewise, kernel column kx = 4 b4 + q
  const int zoff = (8 * kg + r4) * ZPX + 16 * b4 + 4 * q;
  const int poff = (2 * (8 * kg + r4) + 4 * b4 + q) * 4;

  // a tile's operands in registers (raw image values, NaN = no source pixel; dZ quads), loaded for the NEXT tile while this one is multiplied
  constexpr int NE = (WROWS * 38 + 255) / 256;         // 4 patch entries per thread
  float raw[NE][3];
  float4 zr[8];
  auto load_tile = [&](int tile) __attribute__((always_inline)) {
    int b = tile;
    const int tx = b % a.tilesX; b /= a.tilesX;
    const int ty = b % a.tilesY;
    const int n = b / a.tilesY;
    const bool live = tile < a.ntiles;
#pragma unroll
    for (int k = 0; k < NE; ++k) {
      const int e = t + 256 * k, r = e / 38, c = e - r * 38;
      const int iy = 2 * ty * TH + r - 3, ix = 2 * tx * TW + c - 3;
      const bool ok = live && r < WROWS && c < PW && iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW;
#pragma unroll
      for (int ci = 0; ci < 3; ++ci) raw[k][ci] = ok ? a.img[((size_t)(n * 3 + ci) * a.IH + iy) * a.IW + ix] : __builtin_nanf("");
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int e = t + 256 * k, pt = e >> 4, q4 = e & 15;
      const int oy = ty * TH + pt / TW, ox = tx * TW + pt % TW;
      zr[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (live && oy < a.OH && ox < a.OW) zr[k] = *reinterpret_cast<const float4*>(a.dz + ((size_t)(n * a.OH + oy) * a.OW + ox) * 64 + q4 * 4);
    }
  };
  load_tile(blockIdx.x);
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    __syncthreads();                                   // the previous tile has been read by every wave
#pragma unroll
    for (int k = 0; k < NE; ++k) {
      const int e = t + 256 * k, r = e / 38, c = e - r * 38;
      if (e >= WROWS * 38) break;
      float v[3];
#pragma unroll
      for (int ci = 0; ci < 3; ++ci) v[ci] = raw[k][ci] == raw[k][ci] ? (raw[k][ci] - 0.45f) / 0.225f : 0.f;
      uint2 hq, mq;
      fp_hp_split4(v[0], v[1], v[2], 0.f, (float)(1 << STEM_KA), hq, mq);
      *reinterpret_cast<uint2*>(Ph + r * WRS + c * 4) = hq;
      *reinterpret_cast<uint2*>(Pm + r * WRS + c * 4) = mq;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int e = t + 256 * k, pt = e >> 4, q4 = e & 15;
      uint2 hq, mq;
      fp_hp_split4(zr[k].x, zr[k].y, zr[k].z, zr[k].w, sz, hq, mq);
      *reinterpret_cast<uint2*>(Zh + pt * ZPX + q4 * 4) = hq;
      *reinterpret_cast<uint2*>(Zm + pt * ZPX + q4 * 4) = mq;
    }
    __syncthreads();
    load_tile(tile + gridDim.x);                       // in flight under the MFMAs below
#pragma unroll
    for (int s = 0; s < TH; ++s) {                     // K-step = tile row s (sixteen pixels)
      uint4 zh[2], zm[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int o = s * TW * ZPX + zoff + m * 32;
        const uint2 hl = st_tr16(Zh + o), hh = st_tr16(Zh + o + 4 * ZPX), ml = st_tr16(Zm + o), mh = st_tr16(Zm + o + 4 * ZPX);
        zh[m] = make_uint4(hl.x, hl.y, hh.x, hh.y);
        zm[m] = make_uint4(ml.x, ml.y, mh.x, mh.y);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (i >= nky) break;                           // wave-uniform
        const int o = (2 * s + wave + 4 * i) * WRS + poff;
        const uint2 hl = st_tr16(Ph + o), hh = st_tr16(Ph + o + 32), ml = st_tr16(Pm + o), mh = st_tr16(Pm + o + 32);
        const st_f16x8 ph = __builtin_bit_cast(st_f16x8, make_uint4(hl.x, hl.y, hh.x, hh.y));
        const st_f16x8 pm = __builtin_bit_cast(st_f16x8, make_uint4(ml.x, ml.y, mh.x, mh.y));
#pragma unroll
        for (int m = 0; m < 2; ++m) {                  // smallest products first
          acc[i][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(st_f16x8, zm[m]), ph, acc[i][m], 0, 0, 0);
          acc[i][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(st_f16x8, zh[m]), pm, acc[i][m], 0, 0, 0);
          acc[i][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(st_f16x8, zh[m]), ph, acc[i][m], 0, 0, 0);
        }
      }
    }
  }
  const float unscale = ldexpf(1.f, -(STEM_KA + kz));
  float* out = a.part + (size_t)blockIdx.x * 147 * 64;
  const int kx = (lane & 31) >> 2, ci = lane & 3;      // the lane's column of the K block
  if (kx < 7 && ci < 3) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (i >= nky) break;
      const int k = ((wave + 4 * i) * 7 + kx) * 3 + ci;
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[(size_t)k * 64 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg] = acc[i][m][r] * unscale;
    }
  }
}

}  // namespace

// the same partial tensors from the fp16-pair kernel; `amax_dz` = the amax slot of dz
int fp_stem_wgrad_hp_dispatch(const fp_conv_desc* d, const float* img, const float* dz, float* part, int splits, const uint32_t* amax_dz,
                              hipStream_t stream) {
  if (d->Nout != 64 || d->IH != 2 * d->OH || d->IW != 2 * d->OW || splits < 1 || !amax_dz) return -1000;
  StemWHpArgs a;
  a.img = img; a.dz = dz; a.part = part; a.amax_dz = amax_dz;
  a.N = d->N; a.IH = d->IH; a.IW = d->IW; a.OH = d->OH; a.OW = d->OW;
  a.tilesX = (int)fp_ceil_div(d->OW, TW); a.tilesY = (int)fp_ceil_div(d->OH, TH);
  a.ntiles = d->N * a.tilesX * a.tilesY;
  fp_launch(stem_wgrad_hp_kernel, dim3(splits), dim3(256), 0, stream, a);
  return fp_check_launch("fp_conv_stem_wgrad_hp");
}

// stem weight gradient into `splits` partial tensors [splits][147][64] (then fp_wgrad_reduce_launch); -1000 = not handled
int fp_stem_wgrad_tile_dispatch(const fp_conv_desc* d, const float* img, const float* dz, float* part, int splits, hipStream_t stream) {
  static const bool off = getenv("FP_NO_STEM_WTILE") && atoi(getenv("FP_NO_STEM_WTILE"));
  if (off || d->Nout != 64 || d->IH != 2 * d->OH || d->IW != 2 * d->OW || splits < 1) return -1000;
  StemWArgs a;
  a.img = img; a.dz = dz; a.part = part;
  a.N = d->N; a.IH = d->IH; a.IW = d->IW; a.OH = d->OH; a.OW = d->OW;
  a.tilesX = (int)fp_ceil_div(d->OW, TW); a.tilesY = (int)fp_ceil_div(d->OH, TH);
  a.ntiles = d->N * a.tilesX * a.tilesY;
  fp_launch(stem_wgrad_tile_kernel, dim3(splits), dim3(256), 0, stream, a);
  return fp_check_launch("fp_conv_wgrad(stem)");
}

// -1000 = not handled (caller falls back to the flattened kernel)
int fp_stem_tile_dispatch(const fp_conv_desc* d, const float* img, const float* wpacked, const float* bias, float* y, hipStream_t stream,
                          const FpBnSink& sink) {
  static const bool off = getenv("FP_NO_STEM_TILE") && atoi(getenv("FP_NO_STEM_TILE"));
  if (off || d->Nout != 64 || (d->epi & ~(unsigned)FP_EPI_BIAS) || d->IH != 2 * d->OH || d->IW != 2 * d->OW) return -1000;
  StemArgs a;
  a.img = img; a.w = wpacked; a.bias = (d->epi & FP_EPI_BIAS) ? bias : nullptr; a.y = y;
  a.N = d->N; a.IH = d->IH; a.IW = d->IW; a.OH = d->OH; a.OW = d->OW; a.act = d->act;
  a.tilesX = (int)fp_ceil_div(d->OW, TW); a.tilesY = (int)fp_ceil_div(d->OH, TH);
  a.bn_part = nullptr;
  // statistics sink (fp_aux.bn_part, forward form): the stored value is the accumulator itself only without bias / activation
  const int64_t ntiles = (int64_t)d->N * a.tilesX * a.tilesY;
  if (sink.part && !sink.z && !a.bias && d->act == FP_ACT_NONE && ntiles * 64 * 3 <= sink.cap_floats) {
    a.bn_part = sink.part;
    if (sink.nblk_out) *sink.nblk_out = (int32_t)ntiles;
  }
  fp_launch(stem_tile_kernel, dim3(d->N * a.tilesX * a.tilesY), dim3(256), 0, stream, a);
  return fp_check_launch("fp_conv_igemm(stem)");
}

extern "C" int fp_conv_stem_hp_supported(const fp_conv_desc* d) {
  return d && d->gather == FP_GATHER_STEM && d->KH == 7 && d->KW == 7 && d->stride == 2 && d->pad == 3 && d->C0 == 3 && d->C1 == 0 && d->Nout == 64 &&
         !(d->epi & ~(unsigned)FP_EPI_BIAS) && d->IH == 2 * d->OH && d->IW == 2 * d->OW && d->N > 0 && d->OH > 0 && d->OW > 0;
}
extern "C" int fp_conv_stem_hp(const fp_conv_desc* d, const float* img, const void* wpacked_hp, const float* bias, float* y, const uint32_t* amax_w,
                               const fp_aux* aux, fp_stream_t stream) {
  const FpBnSink sink = fp_bn_sink_of(aux);
  unsigned* amax_out = fp_amax_out_of(aux);
  FP_REQUIRE(d && img && wpacked_hp && y && amax_w, "fp_conv_stem_hp: null pointer");
  FP_REQUIRE(fp_conv_stem_hp_supported(d), "fp_conv_stem_hp: shape not supported (see fp_conv_stem_hp_supported)");
  FP_REQUIRE(!(d->epi & FP_EPI_BIAS) || bias, "fp_conv_stem_hp: bias flag without pointer");
  StemHpArgs a;
  a.img = img; a.w = (const unsigned short*)wpacked_hp; a.bias = (d->epi & FP_EPI_BIAS) ? bias : nullptr; a.y = y; a.amax_w = amax_w; a.amax_out = amax_out;
  a.N = d->N; a.IH = d->IH; a.IW = d->IW; a.OH = d->OH; a.OW = d->OW; a.act = d->act;
  a.tilesX = (int)fp_ceil_div(d->OW, TW); a.tilesY = (int)fp_ceil_div(d->OH, TH);
  a.bn_part = nullptr;
  const int64_t ntiles = (int64_t)d->N * a.tilesX * a.tilesY;
  if (sink.nblk_out) *sink.nblk_out = 0;
  if (sink.part && !sink.z && !a.bias && d->act == FP_ACT_NONE && ntiles * 64 * 3 <= sink.cap_floats) {
    a.bn_part = sink.part;
    if (sink.nblk_out) *sink.nblk_out = (int32_t)ntiles;
  }
  static const int wgs = getenv("FP_STEM_HP_WGS") ? atoi(getenv("FP_STEM_HP_WGS")) : 512;          // persistent: two workgroups per CU (194 VGPRs)
  const int64_t budget = wgs > 0 ? wgs : 512;        // (an unusable value of the experiment knob falls back to the default)
  fp_launch(stem_tile_hp_kernel, dim3((unsigned)(ntiles < budget ? ntiles : budget)), dim3(256), 0, (hipStream_t)stream, a);
  return fp_check_launch("fp_conv_stem_hp");
}

