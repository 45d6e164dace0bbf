// Weight packing for the implicit-GEMM kernels (gfx950).  One element function per layout, one generic kernel for a single
// tensor, and a batched kernel that repacks EVERY convolution of the network in one launch after the optimizer step
// (the per-tensor launches were ~200 x 4.5 us of GPU time and, worse, ~2 ms of host launch time per training step).
//
//   FWD        wp[tap][ceil(Cs/16)][Cout][16]      = W[n][c_begin + k][tap]                       (k < Cs input-channel slice)
//   DGRAD      wp[tap][ceil(Cout/16)][Cs][16]      = W[co][c_begin + ci][tap]
//   STEM       wp[10][64][16]: k = (ky*7 + kx)*3 + ci over the 7x7x3 stem
//   UP2_FWD    wp[phase 4][tap 4][ceil(Cs/16)][Cout][16]: row/column-collapsed weights of the nearest-x2 phase decomposition
//   UP2_DGRAD  wp[16 taps r*4+s][ceil(Cout/16)][Cs][16]: the 4x4 stride-2 kernel of its data gradient   (conv_up2_phase.hip)
#include <stdlib.h>

#include "fp_common.h"

namespace {

__device__ __forceinline__ float pack_elem(const fp_pack_job& j, size_t e) {
  const float* __restrict__ w = j.w;
  const int kr = (int)(e & 15);
  size_t r = e >> 4;
  switch (j.kind) {
    case FP_PACK_FWD_HP:
    case FP_PACK_FWD_BF3:
    case FP_PACK_FWD: {
      const int T = j.KH * j.KW, KC16 = (j.c_count + 15) / 16;
      const int n = (int)(r % j.Cout); r /= j.Cout;
      const int kc = (int)(r % KC16), tap = (int)(r / KC16);
      const int k = kc * 16 + kr;
      return k < j.c_count ? w[((size_t)n * j.Cin + j.c_begin + k) * T + tap] : 0.f;
    }
    case FP_PACK_DGRAD_HP:
    case FP_PACK_DGRAD_BF3:
    case FP_PACK_DGRAD: {
      const int T = j.KH * j.KW, KC16 = (j.Cout + 15) / 16;
      const int ci = (int)(r % j.c_count); r /= j.c_count;
      const int kc = (int)(r % KC16), tap = (int)(r / KC16);
      const int co = kc * 16 + kr;
      return co < j.Cout ? w[((size_t)co * j.Cin + j.c_begin + ci) * T + tap] : 0.f;
    }
    case FP_PACK_STEM: {
      const int n = (int)(r % 64), kc = (int)(r / 64);
      const int kk = kc * 16 + kr;
      if (kk >= 147) return 0.f;
      const int ky = kk / 21, rem = kk - ky * 21, kx = rem / 3, ci = rem - kx * 3;
      return w[((n * 3 + ci) * 7 + ky) * 7 + kx];
    }
    case FP_PACK_STEM_HP: {     // K index = ky * 24 + (kx * 3 + ci): a row of the 7 x 7 x 3 patch is 21 consecutive elements, padded to 24
      const int n = (int)(r % 64), st = (int)(r / 64);
      const int kq = st * 16 + kr, ky = kq / 24, jj = kq - ky * 24;
      if (ky >= 7 || jj >= 21) return 0.f;
      const int kx = jj / 3, ci = jj - kx * 3;
      return w[((n * 3 + ci) * 7 + ky) * 7 + kx];
    }
    case FP_PACK_UP2_FWD_HP:
    case FP_PACK_UP2_FWD_BF3:
    case FP_PACK_UP2_FWD: {
      const int KC16 = (j.c_count + 15) / 16;
      const int n = (int)(r % j.Cout); r /= j.Cout;
      const int kc = (int)(r % KC16); r /= KC16;
      const int tap = (int)(r & 3), phase = (int)(r >> 2);
      const int dy = phase >> 1, dx = phase & 1, ta = tap >> 1, tb = tap & 1;
      const int k = kc * 16 + kr;
      if (k >= j.c_count) return 0.f;
      const float* wk = w + ((size_t)n * j.Cin + j.c_begin + k) * 9;
      // rows collapsed into (dy, a): dy=0: a=0 <- {0}, a=1 <- {1,2};  dy=1: a=0 <- {0,1}, a=1 <- {2}   (same for columns)
      const int ky_lo = dy == 0 ? (ta == 0 ? 0 : 1) : (ta == 0 ? 0 : 2), ky_hi = dy == 0 ? (ta == 0 ? 0 : 2) : (ta == 0 ? 1 : 2);
      const int kx_lo = dx == 0 ? (tb == 0 ? 0 : 1) : (tb == 0 ? 0 : 2), kx_hi = dx == 0 ? (tb == 0 ? 0 : 2) : (tb == 0 ? 1 : 2);
      float v = 0.f;
      for (int ky = ky_lo; ky <= ky_hi; ++ky)
        for (int kx = kx_lo; kx <= kx_hi; ++kx) v += wk[ky * 3 + kx];
      return v;
    }
    case FP_PACK_UP2_DGRAD: {
      const int KC16 = (j.Cout + 15) / 16;
      const int c = (int)(r % j.c_count); r /= j.c_count;
      const int kc = (int)(r % KC16), tap = (int)(r / KC16);
      const int tr = tap >> 2, ts = tap & 3;
      const int n = kc * 16 + kr;
      if (n >= j.Cout) return 0.f;
      const float* wk = w + ((size_t)n * j.Cin + j.c_begin + c) * 9;
      // K4[0] = W[2], K4[1] = W[1] + W[2], K4[2] = W[0] + W[1], K4[3] = W[0]
      const int ky_lo = tr == 0 ? 2 : (tr == 1 ? 1 : 0), ky_hi = tr == 0 ? 2 : (tr == 1 ? 2 : (tr == 2 ? 1 : 0));
      const int kx_lo = ts == 0 ? 2 : (ts == 1 ? 1 : 0), kx_hi = ts == 0 ? 2 : (ts == 1 ? 2 : (ts == 2 ? 1 : 0));
      float v = 0.f;
      for (int ky = ky_lo; ky <= ky_hi; ++ky)
        for (int kx = kx_lo; kx <= kx_hi; ++kx) v += wk[ky * 3 + kx];
      return v;
    }
    case FP_PACK_UP2_DGRAD_HP:
    case FP_PACK_UP2_DGRAD_BF3: {   // [phase 4][tap 4][KC16 over Cout][c][16 n]: K4[(py+1)%2 + 2a][(px+1)%2 + 2b][n][c]
      const int KC16 = (j.Cout + 15) / 16;
      const int c = (int)(r % j.c_count); r /= j.c_count;
      const int kc = (int)(r % KC16); r /= KC16;
      const int tap = (int)(r & 3), ph = (int)(r >> 2);
      const int tr = ((ph >> 1) + 1) % 2 + 2 * (tap >> 1), ts = ((ph & 1) + 1) % 2 + 2 * (tap & 1);
      const int n = kc * 16 + kr;
      if (n >= j.Cout) return 0.f;
      const float* wk = w + ((size_t)n * j.Cin + j.c_begin + c) * 9;
      const int ky_lo = tr == 0 ? 2 : (tr == 1 ? 1 : 0), ky_hi = tr == 0 ? 2 : (tr == 1 ? 2 : (tr == 2 ? 1 : 0));
      const int kx_lo = ts == 0 ? 2 : (ts == 1 ? 1 : 0), kx_hi = ts == 0 ? 2 : (ts == 1 ? 2 : (ts == 2 ? 1 : 0));
      float v = 0.f;
      for (int ky = ky_lo; ky <= ky_hi; ++ky)
        for (int kx = kx_lo; kx <= kx_hi; ++kx) v += wk[ky * 3 + kx];
      return v;
    }
    default: return 0.f;
  }
}

__host__ __device__ inline int64_t pack_elems(int kind, int Cout, int KH, int KW, int c_count) {
  const int64_t T = (int64_t)KH * KW;
  switch (kind) {
    case FP_PACK_FWD_HP:
    case FP_PACK_FWD_BF3:
    case FP_PACK_FWD: return T * ((c_count + 15) / 16) * Cout * 16;
    case FP_PACK_DGRAD_HP:
    case FP_PACK_DGRAD_BF3:
    case FP_PACK_DGRAD: return T * ((Cout + 15) / 16) * c_count * 16;
    case FP_PACK_STEM: return 10 * 64 * 16;
    case FP_PACK_STEM_HP: return 11 * 64 * 16;
    case FP_PACK_UP2_FWD_HP:
    case FP_PACK_UP2_FWD_BF3:
    case FP_PACK_UP2_FWD: return (int64_t)16 * ((c_count + 15) / 16) * Cout * 16;
    case FP_PACK_UP2_DGRAD_HP:
    case FP_PACK_UP2_DGRAD_BF3:
    case FP_PACK_UP2_DGRAD: return (int64_t)16 * ((Cout + 15) / 16) * c_count * 16;
    default: return 0;
  }
}

// fp32 layouts: wp[e] = v.  *_BF3 layouts: v is split exactly into three bf16 terms (h = bf16(v), m = bf16(v - h), l = v - h - m)
// stored as planes [tap][chunk][plane][ncols][16] (conv3x3_tile_bf3.hip); element e = ((tap*KC16 + kc)*ncols + n)*16 + k.
__device__ __forceinline__ unsigned short bf16_rne(float x) {
  unsigned u = __float_as_uint(x);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ bool pack_is_hp(int kind) {
  return kind == FP_PACK_FWD_HP || kind == FP_PACK_DGRAD_HP || kind == FP_PACK_UP2_FWD_HP || kind == FP_PACK_UP2_DGRAD_HP || kind == FP_PACK_STEM_HP;
}
// *_HP layouts: v * 2^kw (kw from the weight tensor's amax slot, FP_HP_TARGET_W) as an fp16 pair, planes [tap][chunk][2][ncols][16]
__device__ __forceinline__ void pack_store(const fp_pack_job& j, size_t e, float v, int kw) {
  if (pack_is_hp(j.kind)) {
    const size_t ncols = (j.kind == FP_PACK_DGRAD_HP || j.kind == FP_PACK_UP2_DGRAD_HP) ? j.c_count : j.Cout;
    const size_t k = e & 15, n = (e >> 4) % ncols, blk = (e >> 4) / ncols;
    _Float16* o = reinterpret_cast<_Float16*>(j.wp) + (blk * 2 * ncols + n) * 16 + k;
    const float vs = ldexpf(v, kw);
    const _Float16 h = (_Float16)vs;
    o[0] = h;
    o[ncols * 16] = (_Float16)(vs - (float)h);
    return;
  }
  if (j.kind != FP_PACK_FWD_BF3 && j.kind != FP_PACK_DGRAD_BF3 && j.kind != FP_PACK_UP2_FWD_BF3 && j.kind != FP_PACK_UP2_DGRAD_BF3) {
    j.wp[e] = v;
    return;
  }
  const size_t ncols = (j.kind == FP_PACK_DGRAD_BF3 || j.kind == FP_PACK_UP2_DGRAD_BF3) ? j.c_count : j.Cout;
  const size_t k = e & 15, n = (e >> 4) % ncols, blk = (e >> 4) / ncols;
  unsigned short* o = reinterpret_cast<unsigned short*>(j.wp) + (blk * 3 * ncols + n) * 16 + k;
  const unsigned short h = bf16_rne(v);
  const float r1 = v - __uint_as_float((unsigned)h << 16);
  const unsigned short m = bf16_rne(r1);
  const float r2 = r1 - __uint_as_float((unsigned)m << 16);
  o[0] = h;
  o[ncols * 16] = m;
  o[2 * ncols * 16] = bf16_rne(r2);
}

__global__ void __launch_bounds__(256) pack_one_kernel(const fp_pack_job j, size_t total) {
  const int kw = pack_is_hp(j.kind) ? fp_hp_exponent(fp_amax_bits(j.amax), FP_HP_TARGET_W) : 0;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) pack_store(j, e, pack_elem(j, e), kw);
}

// ---- the batched repack, tile form (round 4) ---------------------------------------------------------------------------------------
// The element-wise form above reads OIHW with a stride of KH*KW floats between neighbouring lanes and stores two bytes per lane: 443 us per
// step for 31 M weights on an idle GPU, and -- launched under the first encoder layers of the next step with ~25 k workgroups -- it took the
// wave slots of the latency-bound chain it was meant to hide under (first layer-1 convolutions 142 / 107 us instead of 30; the ordered trace
// in profiles/round4_notes.md).  Here a workgroup moves a tile of 16 output channels x 32 input channels x all taps: rows of 32*T contiguous
// floats in (coalesced), through LDS, and out as runs of 512 contiguous bytes per (tap, 16-channel chunk, plane); the values and their
// arithmetic (scaling, fp16 / bf16 splits, the tap sums of the phase layouts in their old order) are those of pack_elem / pack_store, bit for
// bit.  The launch is persistent: `gridDim.x` workgroups walk the virtual blocks b = blockIdx.x, + gridDim.x, ... so the caller bounds how many
// CUs a repack on a side stream may occupy.
constexpr int PT_N = 16, PT_K = 32;

__host__ __device__ inline bool pack_is_dgrad_layout(int kind) {
  return kind == FP_PACK_DGRAD || kind == FP_PACK_DGRAD_BF3 || kind == FP_PACK_DGRAD_HP || kind == FP_PACK_UP2_DGRAD || kind == FP_PACK_UP2_DGRAD_BF3 ||
         kind == FP_PACK_UP2_DGRAD_HP;
}
__host__ __device__ inline bool pack_is_up2(int kind) {
  return kind == FP_PACK_UP2_FWD || kind == FP_PACK_UP2_FWD_BF3 || kind == FP_PACK_UP2_FWD_HP || kind == FP_PACK_UP2_DGRAD || kind == FP_PACK_UP2_DGRAD_BF3 ||
         kind == FP_PACK_UP2_DGRAD_HP;
}
__host__ __device__ inline int pack_tiles(int kind, int Cout, int c_count) {
  return (kind == FP_PACK_STEM || kind == FP_PACK_STEM_HP) ? 0 : ((Cout + PT_N - 1) / PT_N) * ((c_count + PT_K - 1) / PT_K);
}

// value of virtual tap `vt` from the nine (T = 9) or one (T = 1) taps of one (n, k) at `wk` (stride 1 between taps)
template <int T>
__device__ __forceinline__ float pack_tap_value(int kind, const float* wk, int vt) {
  if (T == 1) return wk[0];
  int ky_lo, ky_hi, kx_lo, kx_hi;
  switch (kind) {
    case FP_PACK_UP2_FWD_HP:
    case FP_PACK_UP2_FWD_BF3:
    case FP_PACK_UP2_FWD: {
      const int tap = vt & 3, phase = vt >> 2;
      const int dy = phase >> 1, dx = phase & 1, ta = tap >> 1, tb = tap & 1;
      ky_lo = dy == 0 ? (ta == 0 ? 0 : 1) : (ta == 0 ? 0 : 2); ky_hi = dy == 0 ? (ta == 0 ? 0 : 2) : (ta == 0 ? 1 : 2);
      kx_lo = dx == 0 ? (tb == 0 ? 0 : 1) : (tb == 0 ? 0 : 2); kx_hi = dx == 0 ? (tb == 0 ? 0 : 2) : (tb == 0 ? 1 : 2);
      break;
    }
    case FP_PACK_UP2_DGRAD:
    case FP_PACK_UP2_DGRAD_HP:
    case FP_PACK_UP2_DGRAD_BF3: {
      int tr, ts;
      if (kind == FP_PACK_UP2_DGRAD) { tr = vt >> 2; ts = vt & 3; }
      else { const int tap = vt & 3, ph = vt >> 2; tr = ((ph >> 1) + 1) % 2 + 2 * (tap >> 1); ts = ((ph & 1) + 1) % 2 + 2 * (tap & 1); }
      ky_lo = tr == 0 ? 2 : (tr == 1 ? 1 : 0); ky_hi = tr == 0 ? 2 : (tr == 1 ? 2 : (tr == 2 ? 1 : 0));
      kx_lo = ts == 0 ? 2 : (ts == 1 ? 1 : 0); kx_hi = ts == 0 ? 2 : (ts == 1 ? 2 : (ts == 2 ? 1 : 0));
      break;
    }
    default: return wk[vt];
  }
  float v = 0.f;
  for (int ky = ky_lo; ky <= ky_hi; ++ky)
    for (int kx = kx_lo; kx <= kx_hi; ++kx) v += wk[ky * 3 + kx];
  return v;
}

template <int T>
__device__ __forceinline__ void pack_tile(const fp_pack_job& j, int ti, float* __restrict__ tile, int kw) {
  constexpr int KT = PT_K * T, RS = KT + 1;          // row stride odd: conflict-free for lanes that walk n as for lanes that walk k
  const int t = threadIdx.x;
  const int ntk = (j.c_count + PT_K - 1) / PT_K;
  const int n0 = (ti / ntk) * PT_N, k0 = (ti % ntk) * PT_K;
  const int klen = min(PT_K, j.c_count - k0) * T;
  constexpr int NLD = PT_N * KT / 256;               // 18 (3 x 3) or 2 (1 x 1) loads per thread, all in flight before the first LDS write
  float ld[NLD];
  const float* __restrict__ src = j.w + ((size_t)n0 * j.Cin + j.c_begin + k0) * T;      // wave-uniform base, 32-bit per-lane offsets
  const unsigned rowstride = (unsigned)j.Cin * T;
  const int nrows = min(PT_N, j.Cout - n0);
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int f = t + i * 256, r = f / KT, c = f - r * KT;
    const bool ok = r < nrows && c < klen;
    const unsigned off = ok ? (unsigned)r * rowstride + (unsigned)c : 0u;
    const float v = src[off];
    ld[i] = ok ? v : 0.f;
  }
  __syncthreads();                                   // the previous tile's readers are done
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int f = t + i * 256, r = f / KT;
    tile[r * RS + (f - r * KT)] = ld[i];
  }
  __syncthreads();
  const int kind = j.kind;
  const bool dg = pack_is_dgrad_layout(kind);
  const int VT = pack_is_up2(kind) ? 16 : T;
  const int ncols = dg ? j.c_count : j.Cout;
  const int KC16 = ((dg ? j.Cout : j.c_count) + 15) / 16;
  const bool hp = pack_is_hp(kind);
  const bool bf3 = kind == FP_PACK_FWD_BF3 || kind == FP_PACK_DGRAD_BF3 || kind == FP_PACK_UP2_FWD_BF3 || kind == FP_PACK_UP2_DGRAD_BF3;
  for (int kc2 = 0; kc2 < PT_K / 16; ++kc2) {
    // FWD layouts: chunk = 16 input channels, column = output channel; DGRAD layouts: chunk = the tile's 16 output channels, column = input channel
    const int chunk = dg ? n0 / 16 : k0 / 16 + kc2;
    if (!dg && chunk >= KC16) break;
    if (!hp && !bf3) {                               // fp32: one element per thread, 1 KB contiguous per (tap, chunk)
      const int col_l = t >> 4, kr = t & 15;
      const int n_l = dg ? kr : col_l, k_l = kc2 * 16 + (dg ? col_l : kr);
      const int col = dg ? k0 + k_l : n0 + n_l;
      if (col < ncols)
        for (int vt = 0; vt < VT; ++vt)
          j.wp[((size_t)(vt * KC16 + chunk) * ncols + col) * 16 + kr] = pack_tap_value<T>(kind, tile + n_l * RS + k_l * T, vt);
      continue;
    }
    // split layouts: two neighbouring kr per thread, one 32-bit store per plane
    const int idx = t & 127, col_l = idx >> 3, kr0 = (idx & 7) * 2;
    const int col = dg ? k0 + kc2 * 16 + col_l : n0 + col_l;
    if (col >= ncols) continue;
    const float* a0 = dg ? tile + kr0 * RS + (kc2 * 16 + col_l) * T : tile + col_l * RS + (kc2 * 16 + kr0) * T;
    const float* a1 = a0 + (dg ? RS : T);
    if (hp) {
      const int plane = t >> 7;
      for (int vt = 0; vt < VT; ++vt) {
        const float s0 = ldexpf(pack_tap_value<T>(kind, a0, vt), kw), s1 = ldexpf(pack_tap_value<T>(kind, a1, vt), kw);
        const _Float16 h0 = (_Float16)s0, h1 = (_Float16)s1;
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        h2 o;
        if (plane == 0) { o[0] = h0; o[1] = h1; }
        else { o[0] = (_Float16)(s0 - (float)h0); o[1] = (_Float16)(s1 - (float)h1); }
        _Float16* dst = reinterpret_cast<_Float16*>(j.wp) + ((size_t)(vt * KC16 + chunk) * 2 * ncols + col) * 16 + kr0 + (size_t)plane * ncols * 16;
        *reinterpret_cast<h2*>(dst) = o;
      }
    } else if (t < 128) {
      for (int vt = 0; vt < VT; ++vt) {
        unsigned short* dst = reinterpret_cast<unsigned short*>(j.wp) + ((size_t)(vt * KC16 + chunk) * 3 * ncols + col) * 16 + kr0;
        unsigned pl[3] = {0u, 0u, 0u};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const float v = pack_tap_value<T>(kind, q ? a1 : a0, vt);
          const unsigned short h = bf16_rne(v);
          const float r1 = v - __uint_as_float((unsigned)h << 16);
          const unsigned short m = bf16_rne(r1);
          const float r2 = r1 - __uint_as_float((unsigned)m << 16);
          pl[0] |= (unsigned)h << (16 * q); pl[1] |= (unsigned)m << (16 * q); pl[2] |= (unsigned)bf16_rne(r2) << (16 * q);
        }
        *reinterpret_cast<unsigned*>(dst) = pl[0];
        *reinterpret_cast<unsigned*>(dst + (size_t)ncols * 16) = pl[1];
        *reinterpret_cast<unsigned*>(dst + (size_t)2 * ncols * 16) = pl[2];
      }
    }
  }
}

// virtual block b serves job blk2job[b]; a job's virtual blocks are contiguous starting at jobs[job].block_begin and share its tiles
__global__ void __launch_bounds__(256) pack_batched_kernel(const fp_pack_job* __restrict__ jobs, const int32_t* __restrict__ blk2job, int nblocks) {
  __shared__ float tile[PT_N * (PT_K * 9 + 1)];
  for (int vb = blockIdx.x; vb < nblocks; vb += gridDim.x) {
    const fp_pack_job j = jobs[blk2job[vb]];
    const int kw = pack_is_hp(j.kind) ? fp_hp_exponent(fp_amax_bits(j.amax), FP_HP_TARGET_W) : 0;
    const int T = j.KH * j.KW;
    if (j.kind == FP_PACK_STEM || j.kind == FP_PACK_STEM_HP || (T != 9 && T != 1)) {          // the stem's 7 x 7 x 3 table (10 K elements): element-wise as before
      const size_t total = (size_t)pack_elems(j.kind, j.Cout, j.KH, j.KW, j.c_count);
      for (size_t e = (size_t)(vb - j.block_begin) * 256 + threadIdx.x; e < total; e += (size_t)j.block_count * 256) pack_store(j, e, pack_elem(j, e), kw);
      continue;
    }
    const int ntiles = pack_tiles(j.kind, j.Cout, j.c_count);
    for (int ti = vb - j.block_begin; ti < ntiles; ti += j.block_count) {
      if (T == 9) pack_tile<9>(j, ti, tile, kw);
      else pack_tile<1>(j, ti, tile, kw);
    }
  }
}

// the element-wise form of rounds 2-3, kept for A/B runs (FP_PACK_TILED=0): one workgroup per virtual block, eight elements per thread
__global__ void __launch_bounds__(256) pack_batched_elem_kernel(const fp_pack_job* __restrict__ jobs, const int32_t* __restrict__ blk2job) {
  const fp_pack_job j = jobs[blk2job[blockIdx.x]];
  const size_t total = (size_t)pack_elems(j.kind, j.Cout, j.KH, j.KW, j.c_count);
  const int kw = pack_is_hp(j.kind) ? fp_hp_exponent(fp_amax_bits(j.amax), FP_HP_TARGET_W) : 0;
  for (size_t e = (size_t)(blockIdx.x - j.block_begin) * 256 + threadIdx.x; e < total; e += (size_t)j.block_count * 256) pack_store(j, e, pack_elem(j, e), kw);
}

// amax of every fp16-pair job's raw weight tensor into the job's slot (jobs of one tensor share a slot; slots zeroed by the caller)
__global__ void __launch_bounds__(256) pack_amax_kernel(const fp_pack_job* __restrict__ jobs, const int32_t* __restrict__ blk2job) {
  const int ji = blk2job[blockIdx.x];
  const fp_pack_job j = jobs[ji];
  if (!pack_is_hp(j.kind)) return;                    // uniform per block
  const size_t total = (size_t)j.Cout * j.Cin * j.KH * j.KW, stride = (size_t)j.block_count * 256;
  float m = 0.f;
  if ((reinterpret_cast<uintptr_t>(j.w) & 15) == 0) {
    const size_t t4 = total >> 2;
    for (size_t e = (size_t)(blockIdx.x - j.block_begin) * 256 + threadIdx.x; e < t4; e += stride) m = fp_amax4(m, reinterpret_cast<const float4*>(j.w)[e]);
    if (blockIdx.x == j.block_begin && threadIdx.x < (total & 3)) m = fmaxf(m, fabsf(j.w[(t4 << 2) + threadIdx.x]));
  } else {
    for (size_t e = (size_t)(blockIdx.x - j.block_begin) * 256 + threadIdx.x; e < total; e += stride) m = fmaxf(m, fabsf(j.w[e]));
  }
  fp_amax_publish_block(j.amax, m);
}
__global__ void __launch_bounds__(256) pack_amax_one_kernel(const float* __restrict__ w, size_t total, unsigned* __restrict__ slot) {
  float m = 0.f;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) m = fmaxf(m, fabsf(w[e]));
  m = fp_wave_max(m);
  if ((threadIdx.x & 63) == 0) fp_amax_publish(slot, blockIdx.x * 4 + (threadIdx.x >> 6), m);
}
__global__ void __launch_bounds__(64) pack_zero_slot_kernel(unsigned* __restrict__ slot) {
  if (threadIdx.x < FP_AMAX_SLOTS) slot[threadIdx.x * FP_AMAX_STRIDE] = 0u;
}

int launch_one(int kind, const float* w, float* wp, int Cout, int Cin, int KH, int KW, int c_begin, int c_count, hipStream_t stream,
               const char* what, uint32_t* amax = nullptr) {
  fp_pack_job j;
  j.amax = amax;
  j.w = w; j.wp = wp; j.Cout = Cout; j.Cin = Cin; j.KH = KH; j.KW = KW; j.kind = kind; j.c_begin = c_begin; j.c_count = c_count;
  j.block_begin = 0; j.block_count = 0;
  const size_t total = (size_t)pack_elems(kind, Cout, KH, KW, c_count);
  size_t g = (total + 255) / 256;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  fp_launch(pack_one_kernel, dim3((unsigned)g), dim3(256), 0, stream, j, total);
  return fp_check_launch(what);
}

}  // namespace

extern "C" int64_t fp_packed_weight_elems(int32_t Cout, int32_t Cin, int32_t KH, int32_t KW, int32_t for_dgrad, int32_t stem) {
  if (stem) return pack_elems(FP_PACK_STEM, 64, 7, 7, 3);
  return pack_elems(for_dgrad ? FP_PACK_DGRAD : FP_PACK_FWD, Cout, KH, KW, Cin);
}
// storage of the bf16x3 layouts, in floats (3 bf16 per weight = 1.5 floats)
extern "C" int64_t fp_packed_weight_elems_bf3(int32_t Cout, int32_t Cin, int32_t KH, int32_t KW, int32_t for_dgrad) {
  return pack_elems(for_dgrad ? FP_PACK_DGRAD_BF3 : FP_PACK_FWD_BF3, Cout, KH, KW, Cin) * 3 / 2;
}
extern "C" int fp_pack_conv_weight_bf3(const float* w_oihw, void* wp, int32_t Cout, int32_t Cin, int32_t KH, int32_t KW, int32_t for_dgrad,
                                       fp_stream_t stream) {
  FP_REQUIRE(w_oihw && wp, "fp_pack_conv_weight_bf3: null pointer");
  return launch_one(for_dgrad ? FP_PACK_DGRAD_BF3 : FP_PACK_FWD_BF3, w_oihw, (float*)wp, Cout, Cin, KH, KW, 0, Cin, (hipStream_t)stream,
                    "fp_pack_conv_weight_bf3");
}
// ---- fp16-pair layouts (two fp16 per weight = one float of storage) --------------------------------------------------------
extern "C" int64_t fp_packed_weight_elems_hp(int32_t Cout, int32_t Cin, int32_t KH, int32_t KW, int32_t for_dgrad) {
  return pack_elems(for_dgrad ? FP_PACK_DGRAD_HP : FP_PACK_FWD_HP, Cout, KH, KW, Cin);
}
// fills the weight tensor's amax slot (zero + reduce) unless `amax_ready`, then packs
extern "C" int fp_weight_amax(const float* w, int64_t n, uint32_t* amax_slot, fp_stream_t stream) {
  FP_REQUIRE(w && amax_slot && n > 0, "fp_weight_amax: bad arguments");
  fp_launch(pack_zero_slot_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, amax_slot);
  size_t g = ((size_t)n + 2047) / 2048;
  if (g > 1024) g = 1024;
  fp_launch(pack_amax_one_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, w, (size_t)n, amax_slot);
  return fp_check_launch("fp_weight_amax");
}
extern "C" int fp_pack_conv_weight_hp(const float* w_oihw, void* wp, int32_t Cout, int32_t Cin, int32_t KH, int32_t KW, int32_t for_dgrad,
                                      uint32_t* amax_slot, int32_t amax_ready, fp_stream_t stream) {
  FP_REQUIRE(w_oihw && wp && amax_slot, "fp_pack_conv_weight_hp: null pointer");
  if (!amax_ready) {
    const int rc = fp_weight_amax(w_oihw, (int64_t)Cout * Cin * KH * KW, amax_slot, stream);
    if (rc) return rc;
  }
  return launch_one(for_dgrad ? FP_PACK_DGRAD_HP : FP_PACK_FWD_HP, w_oihw, (float*)wp, Cout, Cin, KH, KW, 0, Cin, (hipStream_t)stream,
                    "fp_pack_conv_weight_hp", amax_slot);
}
extern "C" int fp_pack_weights_amax(const fp_pack_job* jobs_dev, const int32_t* blk2job_dev, int32_t nblocks, fp_stream_t stream) {
  FP_REQUIRE(jobs_dev && blk2job_dev && nblocks > 0, "fp_pack_weights_amax: bad arguments");
  fp_launch(pack_amax_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, jobs_dev, blk2job_dev);
  return fp_check_launch("fp_pack_weights_amax");
}

extern "C" int64_t fp_up2_packed_weight_elems(int32_t Ncols, int32_t K) { return pack_elems(FP_PACK_UP2_FWD, Ncols, 3, 3, K); }
extern "C" int fp_pack_up2_weight_dgrad_bf3(const float* w_oihw, void* wp, int32_t Cout, int32_t Cin, int32_t c_begin, int32_t c_count,
                                            fp_stream_t stream) {
  FP_REQUIRE(w_oihw && wp && c_begin >= 0 && c_count > 0 && c_begin + c_count <= Cin, "fp_pack_up2_weight_dgrad_bf3: bad arguments");
  return launch_one(FP_PACK_UP2_DGRAD_BF3, w_oihw, (float*)wp, Cout, Cin, 3, 3, c_begin, c_count, (hipStream_t)stream,
                    "fp_pack_up2_weight_dgrad_bf3");
}
extern "C" int fp_pack_up2_weight_bf3(const float* w_oihw, void* wp, int32_t Cout, int32_t Cin, int32_t c_begin, int32_t c_count,
                                      fp_stream_t stream) {
  FP_REQUIRE(w_oihw && wp && c_begin >= 0 && c_count > 0 && c_begin + c_count <= Cin, "fp_pack_up2_weight_bf3: bad arguments");
  return launch_one(FP_PACK_UP2_FWD_BF3, w_oihw, (float*)wp, Cout, Cin, 3, 3, c_begin, c_count, (hipStream_t)stream, "fp_pack_up2_weight_bf3");
}

extern "C" int fp_pack_conv_weight(const float* w_oihw, float* wp, int32_t Cout, int32_t Cin, int32_t KH, int32_t KW, int32_t stem,
                                   fp_stream_t stream) {
  FP_REQUIRE(w_oihw && wp, "fp_pack_conv_weight: null pointer");
  if (stem) FP_REQUIRE(Cout == 64 && Cin == 3 && KH == 7 && KW == 7, "fp_pack_conv_weight: stem must be [64,3,7,7]");
  return launch_one(stem ? FP_PACK_STEM : FP_PACK_FWD, w_oihw, wp, Cout, Cin, KH, KW, 0, Cin, (hipStream_t)stream, "fp_pack_conv_weight");
}
extern "C" int fp_pack_conv_weight_dgrad(const float* w_oihw, float* wp, int32_t Cout, int32_t Cin, int32_t KH, int32_t KW,
                                         fp_stream_t stream) {
  FP_REQUIRE(w_oihw && wp, "fp_pack_conv_weight_dgrad: null pointer");
  return launch_one(FP_PACK_DGRAD, w_oihw, wp, Cout, Cin, KH, KW, 0, Cin, (hipStream_t)stream, "fp_pack_conv_weight_dgrad");
}

#define FP_SLICE_OK(name) FP_REQUIRE(w_oihw && wp && c_begin >= 0 && c_count > 0 && c_begin + c_count <= Cin, name ": bad arguments")
extern "C" int fp_pack_conv_weight_slice(const float* w_oihw, float* wp, int32_t Cout, int32_t Cin, int32_t c_begin, int32_t c_count,
                                         fp_stream_t stream) {
  FP_SLICE_OK("fp_pack_conv_weight_slice");
  return launch_one(FP_PACK_FWD, w_oihw, wp, Cout, Cin, 3, 3, c_begin, c_count, (hipStream_t)stream, "fp_pack_conv_weight_slice");
}
extern "C" int fp_pack_conv_weight_dgrad_slice(const float* w_oihw, float* wp, int32_t Cout, int32_t Cin, int32_t c_begin,
                                               int32_t c_count, fp_stream_t stream) {
  FP_SLICE_OK("fp_pack_conv_weight_dgrad_slice");
  return launch_one(FP_PACK_DGRAD, w_oihw, wp, Cout, Cin, 3, 3, c_begin, c_count, (hipStream_t)stream, "fp_pack_conv_weight_dgrad_slice");
}
extern "C" int fp_pack_up2_weight(const float* w_oihw, float* wp, int32_t Cout, int32_t Cin, int32_t c_begin, int32_t c_count,
                                  fp_stream_t stream) {
  FP_SLICE_OK("fp_pack_up2_weight");
  return launch_one(FP_PACK_UP2_FWD, w_oihw, wp, Cout, Cin, 3, 3, c_begin, c_count, (hipStream_t)stream, "fp_pack_up2_weight");
}
extern "C" int fp_pack_up2_weight_dgrad(const float* w_oihw, float* wp, int32_t Cout, int32_t Cin, int32_t c_begin, int32_t c_count,
                                        fp_stream_t stream) {
  FP_SLICE_OK("fp_pack_up2_weight_dgrad");
  return launch_one(FP_PACK_UP2_DGRAD, w_oihw, wp, Cout, Cin, 3, 3, c_begin, c_count, (hipStream_t)stream, "fp_pack_up2_weight_dgrad");
}

extern "C" int32_t fp_pack_job_blocks(int32_t kind, int32_t Cout, int32_t KH, int32_t KW, int32_t c_count) {
  int64_t b;
  if (kind == FP_PACK_STEM || kind == FP_PACK_STEM_HP || (KH * KW != 9 && KH * KW != 1)) b = (pack_elems(kind, Cout, KH, KW, c_count) + 2047) / 2048;     // element-wise: ~8 elements per thread
  else b = pack_tiles(kind, Cout, c_count);                                                                                   // one virtual block per tile ...
  if (b < 1) b = 1;
  if (b > 512) b = 512;                                                                                                       // ... up to 512 (a 512 x 512 x 3 x 3 tensor)
  return (int32_t)b;
}

// max_wgs > 0: at most that many workgroups walk the `nblocks` virtual blocks (a repack beside a latency-bound chain: Engine.refresh_packed)
extern "C" int fp_pack_weights_batched_capped(const fp_pack_job* jobs_dev, const int32_t* blk2job_dev, int32_t nblocks, int32_t max_wgs,
                                              fp_stream_t stream) {
  FP_REQUIRE(jobs_dev && blk2job_dev && nblocks > 0, "fp_pack_weights_batched: bad arguments");
  static const bool tiled = !getenv("FP_PACK_TILED") || atoi(getenv("FP_PACK_TILED")) != 0;
  if (!tiled) {
    fp_launch(pack_batched_elem_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, jobs_dev, blk2job_dev);
    return fp_check_launch("fp_pack_weights_batched");
  }
  const int grid = max_wgs > 0 && max_wgs < nblocks ? max_wgs : nblocks;
  fp_launch(pack_batched_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, jobs_dev, blk2job_dev, nblocks);
  return fp_check_launch("fp_pack_weights_batched");
}
extern "C" int fp_pack_weights_batched(const fp_pack_job* jobs_dev, const int32_t* blk2job_dev, int32_t nblocks, fp_stream_t stream) {
  return fp_pack_weights_batched_capped(jobs_dev, blk2job_dev, nblocks, 0, stream);
}
