// 3x3 stride-1 weight gradient with all nine taps per workgroup and LDS-DMA staging (gfx950).
//
//   dW[tap][ci][co] = sum over pixels p of  X[p + tap][ci] * dZ[p][co]
//
// The flattened wgrad kernel (conv_wgrad.hip) spends a workgroup per tap, so every X / dZ element is fetched nine times
// and the per-chunk gather index math (two integer divisions per staged pixel) costs more issue cycles than the
// MFMAs it feeds.  Here a workgroup owns one (32 input channels x 32 output channels) block of ALL nine taps and
// walks 4 x 16 pixel chunks of the image: per chunk it brings the 6 x 18 halo of X (32 channels) and the 64 dZ pixels
// (32 channels) into LDS with global_load_lds (direct global->LDS DMA: no staging registers, 128-byte lines, the
// padding / nearest-x2 / concat index math once per 16-byte piece), then every wave takes one chunk row: 8 k-steps
// (2 pixels each) x 9 taps = 72 MFMAs per wave per chunk between barriers.  NHWC is already the MFMA layout for this
// reduction (lanes = channels, k-slots = pixels) so all LDS reads are conflict-free ds_read_b32; the three kx taps
// of one row share operands across consecutive k-steps (column 2k+2+h is tap kx=2 of step k and tap kx=0 of step
// k+1), so a k-step costs 6 A reads + 1 B read for 9 MFMAs.
// The four waves' tiles are summed through LDS in a fixed order; wgrad_reduce_kernel (conv_wgrad.hip) then sums the S
// per-split partials in a fixed order => deterministic.
#include "fp_common.h"

int fp_wgrad_reduce_launch(const float* part, float* dw, int S, int T, int Kc, int Nout, int stem, int accumulate, int kc_total,
                           int k_begin, hipStream_t stream);

namespace {

__device__ __attribute__((aligned(128))) float g_zero_line[32];   // source of zero-padding for the LDS DMA

struct WTileArgs {
  const float* src0;
  const float* src1;
  const float* dz;
  float* part;
  int N, OH, OW, IH, IW, C0, C1, Nout, Kc;
  int mode;             // 0 zero padding, 1 reflection, 2 reflection of cat[nearest_x2(src0), src1]
  int chunksY, chunksX, nchunks, chunksPerSplit, S, citiles, cotiles;
};

constexpr int CH = 4, CW = 16;                    // chunk: 4 rows x 16 columns, one row per wave
constexpr int HH = CH + 2, HWD = CW + 2, HP = HH * HWD;   // 6 x 18 = 108 halo pixels
constexpr int HPP = 112;                          // padded to a multiple of 8 pixels (one DMA instruction = 8 pixels x 128 B)
constexpr int XF = HPP * 32, ZF = CH * CW * 32;   // floats per buffer
constexpr int NX = HPP / 8, NZ = CH * CW / 8;     // DMA instructions per chunk: 14 for X, 8 for dZ

__global__ void __launch_bounds__(256) wgrad3x3_tile_kernel(const WTileArgs a) {
  __shared__ __attribute__((aligned(128))) float lds[2 * (XF + ZF)];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, idx = lane & 31, h = lane >> 5;
  int b = blockIdx.x;
  const int cot = b % a.cotiles; b /= a.cotiles;
  const int cit = b % a.citiles; b /= a.citiles;
  const int s = b;
  const int ci0 = cit * 32, co0 = cot * 32;
  // which source tensor holds this 32-channel block (C0 is a multiple of 32 when C1 > 0)
  const bool from1 = ci0 >= a.C0;
  const float* xsrc = from1 ? a.src1 + (ci0 - a.C0) : a.src0 + ci0;
  const int xC = from1 ? a.C1 : a.C0;
  const bool low = a.mode == 2 && !from1;          // nearest-x2 source lives at half resolution

  const int c_begin = s * a.chunksPerSplit;
  const int c_end = min(a.nchunks, c_begin + a.chunksPerSplit);
  const int piece_px = lane >> 3, piece_q = (lane & 7) * 4;   // this lane's pixel within an 8-pixel DMA piece, channel quad

  auto issue_chunk = [&](int c, int buf) {
    const int cx = c % a.chunksX;
    const int r = c / a.chunksX;
    const int cy = r % a.chunksY, n = r / a.chunksY;
    const int y0 = cy * CH, x0 = cx * CW;
    float* Xb = lds + buf * (XF + ZF);
    float* Zb = Xb + XF;
    for (int i = wave; i < NX + NZ; i += 4) {                    // wave-uniform trip
      const float* src = g_zero_line + piece_q;
      if (i < NX) {
        const int hp = i * 8 + piece_px;
        if (hp < HP) {
          const int hy = hp / HWD, hx = hp - hy * HWD;
          int sy = y0 + hy - 1, sx = x0 + hx - 1;
          bool ok;
          if (a.mode == 0) {
            ok = sy >= 0 && sy < a.IH && sx >= 0 && sx < a.IW;
          } else {
            ok = sy >= -1 && sy <= a.IH && sx >= -1 && sx <= a.IW;
            sy = fp_reflect(sy, a.IH);
            sx = fp_reflect(sx, a.IW);
          }
          if (ok) {
            const size_t pix = low ? ((size_t)(n * (a.IH >> 1) + (sy >> 1)) * (a.IW >> 1) + (sx >> 1))
                                   : ((size_t)(n * a.IH + sy) * a.IW + sx);
            src = xsrc + pix * xC + piece_q;
          }
        }
        __builtin_amdgcn_global_load_lds(src, Xb + i * 256, 16, 0, 0);
      } else {
        const int j = i - NX;
        const int p = j * 8 + piece_px;                         // chunk pixel 0..63
        const int oy = y0 + p / CW, ox = x0 + p % CW;
        if (oy < a.OH && ox < a.OW) src = a.dz + ((size_t)(n * a.OH + oy) * a.OW + ox) * a.Nout + co0 + piece_q;
        __builtin_amdgcn_global_load_lds(src, Zb + j * 256, 16, 0, 0);
      }
    }
  };

  f32x16 acc[9];
#pragma unroll
  for (int tp = 0; tp < 9; ++tp)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[tp][r] = 0.f;

  if (c_begin < c_end) issue_chunk(c_begin, 0);
  __syncthreads();                                              // (compiler drains vmcnt before the barrier)
  for (int c = c_begin; c < c_end; ++c) {
    const int buf = (c - c_begin) & 1;
    if (c + 1 < c_end) issue_chunk(c + 1, buf ^ 1);             // DMA of the next chunk flies under this chunk's MFMAs
    const float* Xb = lds + buf * (XF + ZF);
    const float* Zb = Xb + XF;
    // wave w owns chunk row w: halo rows w .. w+2
    const float* xr0 = Xb + ((wave + 0) * HWD + h) * 32 + idx;
    const float* xr1 = Xb + ((wave + 1) * HWD + h) * 32 + idx;
    const float* xr2 = Xb + ((wave + 2) * HWD + h) * 32 + idx;
    const float* zr = Zb + (wave * CW + h) * 32 + idx;
    float a00 = xr0[0], a10 = xr1[0], a20 = xr2[0];             // column 2k+h   (tap kx = 0)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float bz = zr[(2 * k) * 32];
      const float a01 = xr0[(2 * k + 1) * 32], a11 = xr1[(2 * k + 1) * 32], a21 = xr2[(2 * k + 1) * 32];   // kx = 1
      const float a02 = xr0[(2 * k + 2) * 32], a12 = xr1[(2 * k + 2) * 32], a22 = xr2[(2 * k + 2) * 32];   // kx = 2
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a00, bz, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a01, bz, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a02, bz, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a10, bz, acc[3], 0, 0, 0);
      acc[4] = __builtin_amdgcn_mfma_f32_32x32x2f32(a11, bz, acc[4], 0, 0, 0);
      acc[5] = __builtin_amdgcn_mfma_f32_32x32x2f32(a12, bz, acc[5], 0, 0, 0);
      acc[6] = __builtin_amdgcn_mfma_f32_32x32x2f32(a20, bz, acc[6], 0, 0, 0);
      acc[7] = __builtin_amdgcn_mfma_f32_32x32x2f32(a21, bz, acc[7], 0, 0, 0);
      acc[8] = __builtin_amdgcn_mfma_f32_32x32x2f32(a22, bz, acc[8], 0, 0, 0);
      a00 = a02; a10 = a12; a20 = a22;                           // kx = 2 of this step is kx = 0 of the next
    }
    __syncthreads();                                            // next chunk landed; this buffer is free
  }

  // ---- sum the four waves' tiles through LDS (fixed order), one tap at a time, then the split's partial ------------
  float* red = lds;                                            // [4 waves][16 regs][64 lanes]
  float* out = a.part + (size_t)s * 9 * a.Kc * a.Nout;
#pragma unroll
  for (int tp = 0; tp < 9; ++tp) {   // unrolled: acc[] must stay statically indexed (registers, not scratch)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[tp][r];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = t + 256 * q;                               // element (r, lane) of the 32x32 tile
      const float v = ((red[e] + red[1024 + e]) + red[2048 + e]) + red[3072 + e];
      const int r = e >> 6, ln = e & 63;
      const int ci = ci0 + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5);
      out[((size_t)tp * a.Kc + ci) * a.Nout + co0 + (ln & 31)] = v;
    }
    __syncthreads();
  }
}

bool eligible(const fp_conv_desc* d) {
  if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != 1) return false;
  if (d->gather != FP_GATHER_FWD_ZERO && d->gather != FP_GATHER_FWD_REFLECT && d->gather != FP_GATHER_FWD_REFLECT_UP2) return false;
  if (d->OH != d->IH || d->OW != d->IW) return false;
  if (d->C0 % 32 || d->C1 % 32 || d->Nout % 32) return false;
  const int64_t cy = fp_ceil_div(d->OH, CH), cx = fp_ceil_div(d->OW, CW);
  if (cy * CH * cx * CW * 10 > (int64_t)d->OH * d->OW * 13) return false;    // > 30 % padded work
  if ((int64_t)d->N * cy * cx < 16) return false;
  return true;
}

struct WPlan { int S, chunksPerSplit, nchunks, citiles, cotiles, cy, cx; };
WPlan plan(const fp_conv_desc* d) {
  WPlan p;
  p.cy = (int)fp_ceil_div(d->OH, CH); p.cx = (int)fp_ceil_div(d->OW, CW);
  p.nchunks = d->N * p.cy * p.cx;
  p.citiles = (d->C0 + d->C1) / 32; p.cotiles = d->Nout / 32;
  const int64_t base = (int64_t)p.citiles * p.cotiles;
  int64_t S = fp_ceil_div(512, base);
  if (S > p.nchunks / 4) S = p.nchunks / 4;      // at least 4 chunks per workgroup
  if (S < 1) S = 1;
  if (S > 512) S = 512;
  p.chunksPerSplit = (int)fp_ceil_div(p.nchunks, S);
  p.S = (int)fp_ceil_div(p.nchunks, p.chunksPerSplit);
  return p;
}

}  // namespace

// -1 when the shape is not handled by the tile kernel
int64_t fp_wgrad3x3_tile_workspace(const fp_conv_desc* d) {
  if (!eligible(d)) return -1;
  const WPlan p = plan(d);
  return (int64_t)p.S * 9 * (d->C0 + d->C1) * d->Nout * (int64_t)sizeof(float);
}

int fp_wgrad3x3_tile_dispatch(const fp_conv_desc* d, const float* src0, const float* src1, const float* dz, float* dw_oihw,
                              int accumulate, int kc_total, int k_begin, void* workspace, int64_t workspace_bytes, hipStream_t stream) {
  if (!eligible(d)) return -1000;
  const WPlan p = plan(d);
  if (workspace_bytes < fp_wgrad3x3_tile_workspace(d)) return fp_set_error(FP_EWORKSPACE, "fp_conv_wgrad(tile): workspace too small");
  WTileArgs a;
  a.src0 = src0; a.src1 = src1; a.dz = dz; a.part = (float*)workspace;
  a.N = d->N; a.OH = d->OH; a.OW = d->OW; a.IH = d->IH; a.IW = d->IW; a.C0 = d->C0; a.C1 = d->C1; a.Nout = d->Nout;
  a.Kc = d->C0 + d->C1;
  a.mode = d->gather == FP_GATHER_FWD_ZERO ? 0 : (d->gather == FP_GATHER_FWD_REFLECT ? 1 : 2);
  a.chunksY = p.cy; a.chunksX = p.cx; a.nchunks = p.nchunks; a.chunksPerSplit = p.chunksPerSplit; a.S = p.S;
  a.citiles = p.citiles; a.cotiles = p.cotiles;
  fp_launch(wgrad3x3_tile_kernel, dim3(p.S * p.citiles * p.cotiles), dim3(256), 0, stream, a);
  int rc = fp_check_launch("fp_conv_wgrad(tile)");
  if (rc) return rc;
  return fp_wgrad_reduce_launch((const float*)workspace, dw_oihw, p.S, 9, a.Kc, d->Nout, 0, accumulate, kc_total, k_begin, stream);
}
