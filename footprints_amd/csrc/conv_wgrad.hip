// Convolution weight gradient on fp32 MFMA (NHWC, gfx950), deterministic two-stage reduction.
//
//   dW[tap][k][n] = sum_m A[m][tap,k] * dZ[m][n]           (m = output pixel, k = input channel, n = Cout)
//
// The reduction dimension is the pixel index, so NHWC is already the MFMA-friendly layout: for
// v_mfma_f32_32x32x2_f32 a lane supplies "row idx = lane&31 (a channel), k-slot = lane>>5 (a pixel)", i.e.
// lanes read 32 consecutive channels of one staged pixel -- conflict-free ds_read_b32, no transposition.
// A is gathered exactly as in the forward conv (fp_gather_tap: padding / nearest-x2 / concat are index math).
//
// Work split: workgroup = (pixel split s, tap, 64/128-channel K block, 32/64-channel N block); every workgroup
// walks its pixel range in 32-pixel chunks (register-prefetched, double-buffered LDS, one barrier per chunk)
// and writes one partial tile; fp_wgrad_reduce then sums the S partials in a fixed order and writes the
// torch OIHW gradient.  Replaces the weight half of aten::convolution_backward (reference call site:
// footprints/training/train.py:155 `batch_loss.backward()`).
#include <stdlib.h>

#include "fp_common.h"

int fp_stem_wgrad_tile_dispatch(const fp_conv_desc* d, const float* img, const float* dz, float* part, int splits, hipStream_t stream);

int64_t fp_wgrad3x3_tile_workspace(const fp_conv_desc* d);
int fp_wgrad3x3_tile_dispatch(const fp_conv_desc* d, const float* src0, const float* src1, const float* dz, float* dw_oihw,
                              int accumulate, int kc_total, int k_begin, void* workspace, int64_t workspace_bytes, hipStream_t stream);

namespace {

struct WgradArgs {
  const float* src0;
  const float* src1;
  const float* dz;
  float* part;
  FpGeom g;
  int Nout, M, Kc, T, S, chunksPerSplit, kblocks, nblocks;
};

constexpr int PK = 32;  // pixels per chunk

template <int BI, int BJ, int WI, int WJ, int WK, bool STEM>
__global__ void __launch_bounds__(256) wgrad_kernel(const WgradArgs a) {
  constexpr int TI = BI / WI / 32, TJ = BJ / WJ / 32;
  constexpr int AQ = BI / 4;             // float4 per staged pixel (A)
  constexpr int AP = 256 / AQ;           // pixels per pass (A)
  constexpr int AV = PK / AP;            // passes (A)
  constexpr int ZQ = BJ / 4;
  constexpr int ZP = 256 / ZQ;
  constexpr int ZV = (PK + ZP - 1) / ZP;
  static_assert(WI * WJ * WK == 4, "4 waves");
  __shared__ __attribute__((aligned(16))) float lds[2 * PK * (BI + BJ)];
  float* const As = lds;                 // [2][PK][BI]
  float* const Zs = lds + 2 * PK * BI;   // [2][PK][BJ]

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, idx = lane & 31, h = lane >> 5;
  const int wk = wave % WK, wj = (wave / WK) % WJ, wi = wave / (WK * WJ);
  const FpGeom& g = a.g;

  int b = blockIdx.x;
  const int nb = b % a.nblocks; b /= a.nblocks;
  const int kb = b % a.kblocks; b /= a.kblocks;
  const int tap = b % a.T;      b /= a.T;
  const int s = b;
  const int ky = tap / g.KW, kx = tap - ky * g.KW;
  const int k0 = kb * BI, n0 = nb * BJ;

  const int chunk0 = s * a.chunksPerSplit;
  int nchunks = a.chunksPerSplit;
  {
    const int total = (a.M + PK - 1) / PK;
    if (chunk0 + nchunks > total) nchunks = total - chunk0;
    if (nchunks < 0) nchunks = 0;
  }

  const int aq = t % AQ, ap = t / AQ;
  const int zq = t % ZQ, zp = t / ZQ;
  float4 areg[AV], zreg[ZV];

  auto load_chunk = [&](int c) {
    const int mbase = (chunk0 + c) * PK;
#pragma unroll
    for (int i = 0; i < AV; ++i) {
      const int m = mbase + ap + i * AP;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < a.M) {
        const int ox = m % g.OW, r = m / g.OW, oy = r % g.OH, n = r / g.OH;
        const int c4 = k0 + aq * 4;
        if (STEM) {
          v.x = fp_stem_load(g, a.src0, n, oy, ox, c4 + 0);
          v.y = fp_stem_load(g, a.src0, n, oy, ox, c4 + 1);
          v.z = fp_stem_load(g, a.src0, n, oy, ox, c4 + 2);
          v.w = fp_stem_load(g, a.src0, n, oy, ox, c4 + 3);
        } else {
          int pix[4], pix1;
          fp_gather_tap(g, n, oy, ox, ky, kx, pix, pix1);
          v = fp_gather_load4(g, a.src0, a.src1, pix, pix1, c4);
        }
      }
      areg[i] = v;
    }
#pragma unroll
    for (int i = 0; i < ZV; ++i) {
      const int p = zp + i * ZP;
      const int m = mbase + p;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      const int n4 = n0 + zq * 4;
      if (p < PK && m < a.M && n4 < a.Nout) {
        if (n4 + 3 < a.Nout && (a.Nout & 3) == 0) {
          v = *reinterpret_cast<const float4*>(a.dz + (size_t)m * a.Nout + n4);
        } else {
          const float* q = a.dz + (size_t)m * a.Nout + n4;
          v.x = q[0];
          if (n4 + 1 < a.Nout) v.y = q[1];
          if (n4 + 2 < a.Nout) v.z = q[2];
          if (n4 + 3 < a.Nout) v.w = q[3];
        }
      }
      zreg[i] = v;
    }
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int i = 0; i < AV; ++i)
      *reinterpret_cast<float4*>(As + buf * PK * BI + (ap + i * AP) * BI + aq * 4) = areg[i];
#pragma unroll
    for (int i = 0; i < ZV; ++i) {
      const int p = zp + i * ZP;
      if (p < PK) *reinterpret_cast<float4*>(Zs + buf * PK * BJ + p * BJ + zq * 4) = zreg[i];
    }
  };

  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (nchunks > 0) {
    load_chunk(0);
    store_chunk(0);
  }
  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {
    const bool more = c + 1 < nchunks;
    if (more) load_chunk(c + 1);
    const float* Ab = As + (c & 1) * PK * BI + wi * TI * 32 + idx;
    const float* Zb = Zs + (c & 1) * PK * BJ + wj * TJ * 32 + idx;
    constexpr int KPW = (PK / 2) / WK;  // k-pairs per wave
#pragma unroll
    for (int kk = 0; kk < KPW; ++kk) {
      const int p = 2 * (wk * KPW + kk) + h;
      float af[TI], zf[TJ];
#pragma unroll
      for (int i = 0; i < TI; ++i) af[i] = Ab[p * BI + i * 32];
#pragma unroll
      for (int j = 0; j < TJ; ++j) zf[j] = Zb[p * BJ + j * 32];
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], zf[j], acc[i][j], 0, 0, 0);
    }
    if (more) store_chunk((c + 1) & 1);
    __syncthreads();
  }

  // ---- cross-wave reduction over the WK pixel sub-ranges (fixed order), then the partial tile ------------
  if (WK > 1) {
    float* red = lds;  // reuse staging LDS: [TI*TJ][16][64] floats per (wi,wj) group
    static_assert(WK == 1 || (TI * TJ * 16 * 64 * (4 / WK)) <= 2 * PK * (BI + BJ), "LDS too small for the wk reduction");
    const int grp = wi * WJ + wj;
    for (int w = 1; w < WK; ++w) {
      __syncthreads();
      if (wk == w) {
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((grp * TI * TJ + i * TJ + j) * 16 + r) * 64 + lane] = acc[i][j][r];
      }
      __syncthreads();
      if (wk == 0) {
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] += red[((grp * TI * TJ + i * TJ + j) * 16 + r) * 64 + lane];
      }
    }
  }
  if (wk == 0) {
    float* out = a.part + (size_t)(s * a.T + tap) * a.Kc * a.Nout;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const int n = n0 + (wj * TJ + j) * 32 + idx;
        if (n >= a.Nout) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int k = k0 + (wi * TI + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (k < a.Kc) out[(size_t)k * a.Nout + n] = acc[i][j][r];
        }
      }
  }
}

// dW_oihw[n][k][tap] (+)= sum_s part[s][tap][k][n]; STEM: part k index = (ky*7+kx)*3+ci -> OIHW [n][ci][ky][kx]
// The destination may be an input-channel slice [k_begin, k_begin + Kc) of a wider [Nout][kc_total][KH][KW] gradient.
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int S, int T,
                                                           int Kc, int Nout, int stem, int accumulate, int kc_total, int k_begin) {
  const size_t total = (size_t)T * Kc * Nout;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    // eight independent partial sums keep eight loads in flight (fixed combination order => deterministic)
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
    const int k = (int)(r % Kc), tap = (int)(r / Kc);
    size_t o;
    if (stem) {
      const int ci = k % 3, kpos = k / 3;
      o = ((size_t)n * 3 + ci) * 49 + kpos;
    } else {
      o = ((size_t)n * kc_total + k_begin + k) * T + tap;
    }
    dw[o] = accumulate ? dw[o] + sum : sum;
  }
}

// few outputs, many partial tensors (the stem: 9 408 outputs x 768 partials = 29 MB, which 37 workgroups of the kernel above read in 35 us at
// the very end of the step, alone on the GPU): 32 outputs x 8 slices of the partials per workgroup, four loads in flight per thread, the
// slices combined through LDS in a fixed order
__global__ void __launch_bounds__(256) wgrad_reduce_wide_kernel(const float* __restrict__ part, float* __restrict__ dw, int S, int T,
                                                                int Kc, int Nout, int stem, int accumulate, int kc_total, int k_begin) {
  __shared__ float red[8][32];
  const int total = T * Kc * Nout;
  const int l = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int e = blockIdx.x * 32 + l;
  float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
  if (e < total) {
    int s = sl;
    for (; s + 24 < S; s += 32) {
      const float* q = part + (size_t)s * total + e;
      p0 += q[0]; p1 += q[(size_t)8 * total]; p2 += q[(size_t)16 * total]; p3 += q[(size_t)24 * total];
    }
    for (; s < S; s += 8) p0 += part[(size_t)s * total + e];
  }
  red[sl][l] = (p0 + p1) + (p2 + p3);
  __syncthreads();
  if (sl != 0 || e >= total) return;
  const float sum = ((red[0][l] + red[1][l]) + (red[2][l] + red[3][l])) + ((red[4][l] + red[5][l]) + (red[6][l] + red[7][l]));
  const int n = e % Nout;
  const int r = e / Nout;
  const int k = r % Kc, tap = r / Kc;
  size_t o;
  if (stem) {
    const int ci = k % 3, kpos = k / 3;
    o = ((size_t)n * 3 + ci) * 49 + kpos;
  } else {
    o = ((size_t)n * kc_total + k_begin + k) * T + tap;
  }
  dw[o] = accumulate ? dw[o] + sum : sum;
}

struct Plan {
  int BI, BJ, kblocks, nblocks, T, Kc, S, chunksPerSplit;
};

Plan make_plan(const fp_conv_desc* d) {
  Plan p;
  const bool stem = d->gather == FP_GATHER_STEM;
  p.T = stem ? 1 : d->KH * d->KW;
  p.Kc = stem ? 147 : d->C0 + d->C1;
  p.BJ = d->Nout <= 32 ? 32 : 64;
  p.BI = (!stem && p.BJ == 64 && p.Kc >= 128) ? 128 : 64;   // the stem instantiation is <64,64>
  p.kblocks = (int)fp_ceil_div(p.Kc, p.BI);
  p.nblocks = (int)fp_ceil_div(d->Nout, p.BJ);
  const int64_t M = (int64_t)d->N * d->OH * d->OW;
  const int64_t chunks = fp_ceil_div(M, PK);
  const int64_t base = (int64_t)p.T * p.kblocks * p.nblocks;
  static const int target = getenv("FP_WGRAD32_TARGET_WGS") ? atoi(getenv("FP_WGRAD32_TARGET_WGS")) : 1024;
  int64_t S = fp_ceil_div(target, base);
  if (S > chunks / 4) S = chunks / 4;  // at least 4 chunks per split
  if (S < 1) S = 1;
  if (S > 256) S = 256;
  if (stem) {      // stem_tile.hip's weight-gradient kernel: one partial tensor per workgroup, three workgroups per CU (42 KB of LDS each)
    static const int stem_wgs = getenv("FP_STEM_WGRAD_WGS") ? atoi(getenv("FP_STEM_WGRAD_WGS")) : 768;
    S = stem_wgs;
    if (S > chunks / 4) S = chunks / 4;
    if (S < 1) S = 1;
  }
  p.chunksPerSplit = (int)fp_ceil_div(chunks, S);
  p.S = (int)fp_ceil_div(chunks, p.chunksPerSplit);
  return p;
}

bool use_tile() {
  static const bool off = getenv("FP_NO_WTILE") && atoi(getenv("FP_NO_WTILE"));
  return !off;
}

}  // namespace

int fp_wgrad_reduce_launch(const float* part, float* dw, int S, int T, int Kc, int Nout, int stem, int accumulate, int kc_total,
                           int k_begin, hipStream_t stream) {
  const int64_t total = (int64_t)T * Kc * Nout;
  if (total <= 32768 && S >= 64) {
    fp_launch(wgrad_reduce_wide_kernel, dim3((int)fp_ceil_div(total, 32)), dim3(256), 0, stream, part, dw, S, T, Kc, Nout, stem, accumulate, kc_total,
              k_begin);
    return fp_check_launch("fp_conv_wgrad(reduce)");
  }
  int rgrid = (int)fp_ceil_div(total, 256);
  if (rgrid > 4096) rgrid = 4096;
  fp_launch(wgrad_reduce_kernel, dim3(rgrid), dim3(256), 0, stream, part, dw, S, T, Kc, Nout, stem, accumulate,
                     kc_total, k_begin);
  return fp_check_launch("fp_conv_wgrad(reduce)");
}

extern "C" int64_t fp_conv_wgrad_workspace(const fp_conv_desc* d) {
  if (!d) return 0;
  const Plan p = make_plan(d);
  int64_t need = (int64_t)p.S * p.T * p.Kc * d->Nout * (int64_t)sizeof(float);
  if (d->gather != FP_GATHER_STEM && use_tile()) {
    const int64_t t = fp_wgrad3x3_tile_workspace(d);
    if (t > need) need = t;
  }
  return need;
}

extern "C" int fp_conv_wgrad(const fp_conv_desc* d, const float* src0, const float* src1, const float* dz, float* dw_oihw,
                             int accumulate, void* workspace, int64_t workspace_bytes, fp_stream_t stream_) {
  FP_REQUIRE(d, "fp_conv_wgrad: null pointer");
  const int kc = d->gather == FP_GATHER_STEM ? 3 : d->C0 + d->C1;
  return fp_conv_wgrad_slice(d, src0, src1, dz, dw_oihw, kc, 0, accumulate, workspace, workspace_bytes, stream_);
}

// the stem's weight gradient with fp16-pair operands (stem_tile.hip, stem_wgrad_hp_kernel): same desc, workspace (fp_conv_wgrad_workspace) and
// fixed-order reduce as fp_conv_wgrad on the STEM gather; `amax_dz` = the amax slot of dz
int fp_stem_wgrad_hp_dispatch(const fp_conv_desc* d, const float* img, const float* dz, float* part, int splits, const uint32_t* amax_dz,
                              hipStream_t stream);
extern "C" int fp_conv_stem_wgrad_hp(const fp_conv_desc* d, const float* img_nchw, const float* dz, float* dw_oihw, int accumulate, void* workspace,
                                     int64_t workspace_bytes, const uint32_t* amax_dz, fp_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  FP_REQUIRE(d && img_nchw && dz && dw_oihw && workspace && amax_dz, "fp_conv_stem_wgrad_hp: null pointer");
  FP_REQUIRE(d->gather == FP_GATHER_STEM && d->KH == 7 && d->KW == 7 && d->stride == 2 && d->pad == 3 && d->C0 == 3 && d->C1 == 0 && d->Nout == 64 &&
                 d->IH == 2 * d->OH && d->IW == 2 * d->OW,
             "fp_conv_stem_wgrad_hp: the 7x7 / 2 stem on even image dims only");
  const Plan p = make_plan(d);
  FP_REQUIRE(workspace_bytes >= fp_conv_wgrad_workspace(d), "fp_conv_stem_wgrad_hp: workspace too small");
  static const int hp_wgs = getenv("FP_STEM_WGRAD_HP_WGS") ? atoi(getenv("FP_STEM_WGRAD_HP_WGS")) : 512;      // two resident workgroups per CU (184 registers)
  const int S = (hp_wgs < 1 || p.S < hp_wgs) ? p.S : hp_wgs;   // persistent workgroups = partial tensors (the workspace holds p.S of them)
  const int rc = fp_stem_wgrad_hp_dispatch(d, img_nchw, dz, (float*)workspace, S, amax_dz, stream);
  FP_REQUIRE(rc != -1000, "fp_conv_stem_wgrad_hp: shape not handled");
  if (rc) return rc;
  return fp_wgrad_reduce_launch((const float*)workspace, dw_oihw, S, p.T, p.Kc, d->Nout, 1, accumulate, 3, 0, stream);
}

extern "C" int fp_conv_wgrad_slice(const fp_conv_desc* d, const float* src0, const float* src1, const float* dz, float* dw_oihw,
                                   int32_t kc_total, int32_t k_begin, int accumulate, void* workspace, int64_t workspace_bytes,
                                   fp_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  FP_REQUIRE(d && src0 && dz && dw_oihw && workspace, "fp_conv_wgrad: null pointer");
  FP_REQUIRE(d->gather == FP_GATHER_STEM ? (kc_total == 3 && k_begin == 0) : (k_begin >= 0 && k_begin + d->C0 + d->C1 <= kc_total),
             "fp_conv_wgrad: input-channel slice out of range");
  const bool stem = d->gather == FP_GATHER_STEM;
  FP_REQUIRE(stem || d->gather == FP_GATHER_FWD_ZERO || d->gather == FP_GATHER_FWD_REFLECT || d->gather == FP_GATHER_FWD_REFLECT_UP2,
             "fp_conv_wgrad: gather must be a forward mode");
  if (!stem) FP_REQUIRE(d->C0 > 0 && d->C0 % 4 == 0 && d->C1 % 4 == 0, "fp_conv_wgrad: C0/C1 must be multiples of 4");
  FP_REQUIRE(d->C1 == 0 || src1, "fp_conv_wgrad: src1 missing");
  const Plan p = make_plan(d);
  FP_REQUIRE(workspace_bytes >= fp_conv_wgrad_workspace(d), "fp_conv_wgrad: workspace too small");
  if (!stem && use_tile()) {    // 3x3 stride-1 convs with 32-aligned channels: all-taps LDS-DMA kernel (wgrad3x3_tile.hip)
    const int rc = fp_wgrad3x3_tile_dispatch(d, src0, src1, dz, dw_oihw, accumulate, kc_total, k_begin, workspace, workspace_bytes, stream);
    if (rc != -1000) return rc;
  }
  WgradArgs a;
  a.src0 = src0; a.src1 = src1; a.dz = dz; a.part = (float*)workspace;
  a.g = FpGeom{d->N, d->OH, d->OW, d->IH, d->IW, d->C0, d->C1, d->KH, d->KW, d->stride, d->pad, d->gather};
  a.Nout = d->Nout; a.M = d->N * d->OH * d->OW; a.Kc = p.Kc; a.T = p.T; a.S = p.S; a.chunksPerSplit = p.chunksPerSplit;
  a.kblocks = p.kblocks; a.nblocks = p.nblocks;
  const int grid = p.S * p.T * p.kblocks * p.nblocks;
  if (stem) {      // patch-in-LDS kernel (stem_tile.hip): p.S partial tensors, one per workgroup
    const int rc = fp_stem_wgrad_tile_dispatch(d, src0, dz, (float*)workspace, p.S, stream);
    if (rc != -1000) {
      if (rc) return rc;
      return fp_wgrad_reduce_launch((const float*)workspace, dw_oihw, p.S, p.T, p.Kc, d->Nout, 1, accumulate, kc_total, k_begin, stream);
    }
  }
  if (stem) {
    fp_launch((wgrad_kernel<64, 64, 2, 2, 1, true>), dim3(grid), dim3(256), 0, stream, a);
  } else if (p.BJ == 32) {
    fp_launch((wgrad_kernel<64, 32, 2, 1, 2, false>), dim3(grid), dim3(256), 0, stream, a);
  } else if (p.BI == 128) {
    fp_launch((wgrad_kernel<128, 64, 2, 2, 1, false>), dim3(grid), dim3(256), 0, stream, a);
  } else {
    fp_launch((wgrad_kernel<64, 64, 2, 2, 1, false>), dim3(grid), dim3(256), 0, stream, a);
  }
  int rc = fp_check_launch("fp_conv_wgrad");
  if (rc) return rc;
  return fp_wgrad_reduce_launch((const float*)workspace, dw_oihw, p.S, p.T, p.Kc, d->Nout, stem ? 1 : 0, accumulate, kc_total, k_begin,
                                stream);
}
