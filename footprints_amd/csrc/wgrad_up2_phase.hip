// Weight gradient of a reflection-padded 3x3 conv over a nearest-x2 upsampled input, by output phase (gfx950).
//
// With the phase decomposition of conv_up2_phase.hip the gradient of the collapsed weights is
//     dWc[dy,dx][a][b][ci][co] = sum_{y,x}  low[clamp(y + dy - 1 + a)][clamp(x + dx - 1 + b)][ci] * dZ[2y + dy][2x + dx][co]
// (16 products per low-res pixel instead of 9 per hi-res pixel: 2.25x fewer MACs) and the 3x3 gradient is its un-collapse
//     dW[ky] = sum of dWc over the (dy, a) pairs whose collapsed row contains ky:  ky=0: (0,0),(1,0)  ky=1: (0,1),(1,0)  ky=2: (0,1),(1,1)
// (same along x).  Same scheme as wgrad3x3_tile.hip -- global_load_lds staging of 128-byte channel lines, lanes = channels,
// k-slots = pixels -- but a wave owns one PHASE instead of one chunk row: the workgroup walks chunks of 2 x 16 low-res positions,
// stages the 4 x 18 replicate-padded halo of `low` (32 input channels) and the 4 x 32 hi-res dZ pixels under it (32 output
// channels), de-interleaved by phase on the way into LDS (the DMA picks the source pixel per 8-lane group), and wave p runs
// the four taps of phase p over both rows: 64 MFMAs per wave per chunk.  No cross-wave reduction: every wave writes its own
// four 32 x 32 tiles; up2_wgrad_reduce_kernel sums the S splits and un-collapses in a fixed order => deterministic.
#include "fp_common.h"

namespace {

__device__ __attribute__((aligned(128))) float g_zero_line_p[32];

struct PWArgs {
  const float* low;   // [N][h][w][C0]
  const float* dz;    // [N][2h][2w][Nout]
  float* part;        // [S][16][C0][Nout]
  int N, h, w, C0, Nout;
  int chunksY, chunksX, nchunks, chunksPerSplit, S, citiles, cotiles;
};

constexpr int CHL = 2, CW = 16;
constexpr int XR = CHL + 2, XW = CW + 2, XP = XR * XW;   // 4 x 18 = 72 halo pixels = 9 DMA instructions
constexpr int NXI = XP / 8;                              // 9
constexpr int ZP = 4 * CHL * CW;                         // 128 dZ pixels (4 phases x 2 x 16) = 16 DMA instructions
constexpr int NZI = ZP / 8;                              // 16
constexpr int XF = XP * 32, ZF = ZP * 32;

__global__ void __launch_bounds__(256) wgrad_up2_phase_kernel(const PWArgs a) {
  __shared__ __attribute__((aligned(128))) float lds[2 * (XF + ZF)];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, idx = lane & 31, hh = lane >> 5;
  int b = blockIdx.x;
  const int cot = b % a.cotiles; b /= a.cotiles;
  const int cit = b % a.citiles; b /= a.citiles;
  const int s = b;
  const int ci0 = cit * 32, co0 = cot * 32;
  const int c_begin = s * a.chunksPerSplit;
  const int c_end = min(a.nchunks, c_begin + a.chunksPerSplit);
  const int piece_px = lane >> 3, piece_q = (lane & 7) * 4;
  const int H2 = 2 * a.h, W2 = 2 * a.w;

  auto issue_chunk = [&](int c, int buf) {
    const int cx = c % a.chunksX;
    const int r = c / a.chunksX;
    const int cy = r % a.chunksY, n = r / a.chunksY;
    const int y0 = cy * CHL, x0 = cx * CW;
    float* Xb = lds + buf * (XF + ZF);
    float* Zb = Xb + XF;
    for (int i = wave; i < NXI + NZI; i += 4) {
      const float* src = g_zero_line_p + piece_q;
      if (i < NXI) {
        const int hp = i * 8 + piece_px;
        const int hy = hp / XW, hx = hp - hy * XW;
        const int sy = min(max(y0 + hy - 1, 0), a.h - 1), sx = min(max(x0 + hx - 1, 0), a.w - 1);   // replicate padding
        src = a.low + ((size_t)(n * a.h + sy) * a.w + sx) * a.C0 + ci0 + piece_q;
        __builtin_amdgcn_global_load_lds(src, Xb + i * 256, 16, 0, 0);
      } else {
        const int j = i - NXI;                   // 4 instructions per phase
        const int ph = j >> 2;
        const int p = (j & 3) * 8 + piece_px;    // position within the 2 x 16 chunk
        const int ly = y0 + p / CW, lx = x0 + p % CW;
        if (ly < a.h && lx < a.w)
          src = a.dz + ((size_t)(n * H2 + 2 * ly + (ph >> 1)) * W2 + 2 * lx + (ph & 1)) * a.Nout + co0 + piece_q;
        __builtin_amdgcn_global_load_lds(src, Zb + j * 256, 16, 0, 0);
      }
    }
  };

  f32x16 acc[4];
#pragma unroll
  for (int tp = 0; tp < 4; ++tp)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[tp][r] = 0.f;
  const int dy = wave >> 1, dx = wave & 1;

  if (c_begin < c_end) issue_chunk(c_begin, 0);
  __syncthreads();
  for (int c = c_begin; c < c_end; ++c) {
    const int buf = (c - c_begin) & 1;
    if (c + 1 < c_end) issue_chunk(c + 1, buf ^ 1);
    const float* Xb = lds + buf * (XF + ZF);
    const float* Zb = Xb + XF;
#pragma unroll
    for (int ry = 0; ry < CHL; ++ry) {
      const float* x0r = Xb + ((ry + dy) * XW + dx + hh) * 32 + idx;     // tap row a = 0
      const float* x1r = x0r + XW * 32;                                  // tap row a = 1
      const float* zr = Zb + (wave * (CHL * CW) + ry * CW + hh) * 32 + idx;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float bz = zr[(2 * k) * 32];
        const float a00 = x0r[(2 * k) * 32], a01 = x0r[(2 * k + 1) * 32];
        const float a10 = x1r[(2 * k) * 32], a11 = x1r[(2 * k + 1) * 32];
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a00, bz, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a01, bz, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a10, bz, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a11, bz, acc[3], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  float* out = a.part + ((size_t)s * 16 + wave * 4) * a.C0 * a.Nout;
#pragma unroll
  for (int tp = 0; tp < 4; ++tp)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ci = ci0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
      out[((size_t)tp * a.C0 + ci) * a.Nout + co0 + idx] = acc[tp][r];
    }
}

// stage 1: part[0][e] = sum_s part[s][e] (in place; a thread only ever touches its own e).  64 elements x 4 s-groups per block,
// four loads in flight per thread, fixed combination order.
__global__ void __launch_bounds__(256) up2_wgrad_sum_kernel(float* __restrict__ part, int S, size_t total) {
  __shared__ float red[256];
  const int t = threadIdx.x, g = t >> 6;
  const size_t e = (size_t)blockIdx.x * 64 + (t & 63);
  float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
  if (e < total) {
    int s = g;
    for (; s + 12 < S; s += 16) {
      p0 += part[(size_t)s * total + e]; p1 += part[(size_t)(s + 4) * total + e];
      p2 += part[(size_t)(s + 8) * total + e]; p3 += part[(size_t)(s + 12) * total + e];
    }
    for (; s < S; s += 4) p0 += part[(size_t)s * total + e];
  }
  red[t] = (p0 + p1) + (p2 + p3);
  __syncthreads();
  if (g == 0 && e < total) part[e] = (red[t] + red[64 + t]) + (red[128 + t] + red[192 + t]);
}

// stage 2: dW[n][k_begin + c][ky][kx] (+)= sum_{(dy,a) in rows(ky)} sum_{(dx,b) in cols(kx)} dWc[(dy*2+dx)*4 + a*2 + b][c][n]
__global__ void __launch_bounds__(256) up2_wgrad_uncollapse_kernel(const float* __restrict__ wc, float* __restrict__ dw, int C0, int Nout,
                                                                   int kc_total, int k_begin, int accumulate) {
  const size_t cn = (size_t)C0 * Nout, total = 9 * cn;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const size_t q = e % cn;
    const int kk = (int)(e / cn), ky = kk / 3, kx = kk - ky * 3;
    const int a0 = ky == 0 ? 0 : 1, a1 = ky == 2 ? 1 : 0;   // (dy=0, a0) and (dy=1, a1) hold ky
    const int b0 = kx == 0 ? 0 : 1, b1 = kx == 2 ? 1 : 0;   // (dx=0, b0) and (dx=1, b1) hold kx
    const float v00 = wc[(size_t)(0 * 4 + a0 * 2 + b0) * cn + q], v01 = wc[(size_t)(1 * 4 + a0 * 2 + b1) * cn + q];
    const float v10 = wc[(size_t)(2 * 4 + a1 * 2 + b0) * cn + q], v11 = wc[(size_t)(3 * 4 + a1 * 2 + b1) * cn + q];
    const float sum = (v00 + v01) + (v10 + v11);
    const int n = (int)(q % Nout), c = (int)(q / Nout);
    const size_t o = ((size_t)n * kc_total + k_begin + c) * 9 + kk;
    dw[o] = accumulate ? dw[o] + sum : sum;
  }
}

struct PPlan { int S, chunksPerSplit, nchunks, citiles, cotiles, cy, cx; };
bool eligible(int N, int h, int w, int C0, int Nout) {
  if (C0 % 32 || Nout % 32 || h < 1 || w < 1) return false;
  const int64_t cy = fp_ceil_div(h, CHL), cx = fp_ceil_div(w, CW);
  if (cy * CHL * cx * CW * 10 > (int64_t)h * w * 13) return false;   // > 30 % padded work
  if ((int64_t)N * cy * cx < 16) return false;
  return true;
}
PPlan plan(int N, int h, int w, int C0, int Nout) {
  PPlan p;
  p.cy = (int)fp_ceil_div(h, CHL); p.cx = (int)fp_ceil_div(w, CW);
  p.nchunks = N * p.cy * p.cx;
  p.citiles = C0 / 32; p.cotiles = Nout / 32;
  const int64_t base = (int64_t)p.citiles * p.cotiles;
  int64_t S = fp_ceil_div(512, base);
  if (S > p.nchunks / 4) S = p.nchunks / 4;
  if (S < 1) S = 1;
  if (S > 512) S = 512;
  p.chunksPerSplit = (int)fp_ceil_div(p.nchunks, S);
  p.S = (int)fp_ceil_div(p.nchunks, p.chunksPerSplit);
  return p;
}

}  // namespace

// bytes of workspace, or -1 when the shape is not handled (caller keeps the fused-gather fp_conv_wgrad path)
extern "C" int64_t fp_conv_up2_phase_wgrad_workspace(int32_t N, int32_t h, int32_t w, int32_t C0, int32_t Nout) {
  if (!eligible(N, h, w, C0, Nout)) return -1;
  const PPlan p = plan(N, h, w, C0, Nout);
  return (int64_t)p.S * 16 * C0 * Nout * (int64_t)sizeof(float);
}

extern "C" int fp_conv_up2_phase_wgrad(const float* low, const float* dz, float* dw_oihw, int32_t N, int32_t h, int32_t w, int32_t C0,
                                       int32_t Nout, int32_t kc_total, int32_t k_begin, int accumulate, void* workspace,
                                       int64_t workspace_bytes, fp_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  FP_REQUIRE(low && dz && dw_oihw && workspace, "fp_conv_up2_phase_wgrad: null pointer");
  FP_REQUIRE(eligible(N, h, w, C0, Nout), "fp_conv_up2_phase_wgrad: shape not supported (see fp_conv_up2_phase_wgrad_workspace)");
  FP_REQUIRE(k_begin >= 0 && k_begin + C0 <= kc_total, "fp_conv_up2_phase_wgrad: input-channel slice out of range");
  const PPlan p = plan(N, h, w, C0, Nout);
  FP_REQUIRE(workspace_bytes >= (int64_t)p.S * 16 * C0 * Nout * (int64_t)sizeof(float), "fp_conv_up2_phase_wgrad: workspace too small");
  PWArgs a;
  a.low = low; a.dz = dz; a.part = (float*)workspace;
  a.N = N; a.h = h; a.w = w; a.C0 = C0; a.Nout = Nout;
  a.chunksY = p.cy; a.chunksX = p.cx; a.nchunks = p.nchunks; a.chunksPerSplit = p.chunksPerSplit; a.S = p.S;
  a.citiles = p.citiles; a.cotiles = p.cotiles;
  hipLaunchKernelGGL(wgrad_up2_phase_kernel, dim3(p.S * p.citiles * p.cotiles), dim3(256), 0, stream, a);
  int rc = fp_check_launch("fp_conv_up2_phase_wgrad");
  if (rc) return rc;
  const size_t tot16 = (size_t)16 * C0 * Nout;
  if (p.S > 1) {
    hipLaunchKernelGGL(up2_wgrad_sum_kernel, dim3((unsigned)fp_ceil_div((int64_t)tot16, 64)), dim3(256), 0, stream, (float*)workspace, p.S, tot16);
    rc = fp_check_launch("fp_conv_up2_phase_wgrad(sum)");
    if (rc) return rc;
  }
  int rgrid = (int)fp_ceil_div((int64_t)9 * C0 * Nout, 256);
  if (rgrid > 4096) rgrid = 4096;
  hipLaunchKernelGGL(up2_wgrad_uncollapse_kernel, dim3(rgrid), dim3(256), 0, stream, (const float*)workspace, dw_oihw, C0, Nout, kc_total,
                     k_begin, accumulate);
  return fp_check_launch("fp_conv_up2_phase_wgrad(uncollapse)");
}
