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

}  // namespace

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
  // statistics sink (fp_bn_stats_out_next, forward form): the stored value is the accumulator itself only without bias / activation
  const int64_t ntiles = (int64_t)d->N * a.tilesX * a.tilesY;
  if (sink.part && !sink.z && !a.bias && d->act == FP_ACT_NONE && ntiles * 64 * 3 <= sink.cap_floats) {
    a.bn_part = sink.part;
    if (sink.nblk_out) *sink.nblk_out = (int32_t)ntiles;
  }
  fp_launch(stem_tile_kernel, dim3(d->N * a.tilesX * a.tilesY), dim3(256), 0, stream, a);
  return fp_check_launch("fp_conv_igemm(stem)");
}
