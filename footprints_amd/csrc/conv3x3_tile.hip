// 3x3 stride-1 convolution (forward and data-gradient) with the input tile + halo staged ONCE per 16-channel chunk.
//
// Why a second conv kernel: the flattened implicit-GEMM kernel (conv_igemm.hip) re-gathers the A operand from
// global memory for every tap, i.e. 9x per input element, with one barrier per 16 MFMAs.  Ablations on MI355X
// (profiles/round1_notes.md) showed that kernel bound by that data-movement critical path (global load-to-use
// ~4-5k cycles under load, ds_write bandwidth, a barrier per tap), not by the fp32 MFMA pipe.  Here the workgroup owns a
// TH x TW pixel tile of one image; per 16-channel chunk it stages the (TH+2) x (TW+2) halo tile once into LDS and runs
// all nine taps out of it by shifting the LDS read address -- 6.4x fewer global A loads and ds_writes, the gather
// index math (zero / reflect padding, nearest-x2, concat) runs once per workgroup instead of once per tap, one
// barrier per 9 taps (144 MFMAs per wave), and the global prefetch distance is a whole chunk (>= 9 tap-steps).
// The B operand (a 16 x BN weight slice per tap, shared by every workgroup) is read straight from L1/L2 into
// registers one tap ahead: no LDS round trip and no barrier for it.
//
// LDS image of the halo tile: [halo pixel][16 ch + 4 pad] (LD = 20 floats); an MFMA row block is 32 tile pixels =
// 2 rows x 16 columns; lane (idx, h) reads [pixel(idx) + tap offset][4h (+8)] with one ds_read_b128 that feeds four
// v_mfma_f32_32x32x2_f32 (same k permutation as the B fragment, so the dot product is unchanged).  Accumulation order
// per output element: 16-channel chunks outer, taps inner (conv_igemm.hip: taps outer) -- both are plain fp32 fmaf
// chains, equal to round-off but not bitwise.
#include "fp_common.h"

namespace {

struct TileArgs {
  const float* src0;
  const float* src1;
  const float* w;
  const float* bias;
  const float* addend;
  const float* addend_mask;
  const float* actsrc;
  float* y;
  int N, OH, OW, IH, IW, C0, C1, Nout, KC16;
  int mode;      // 0 zero padding, 1 reflection padding, 2 reflection padding of cat[nearest_x2(src0), src1]
  int off;       // source coordinate = output coordinate + halo offset - off (always 1 here)
  int fold;      // data-gradient of a REFLECTION-padded conv: the gradient of the virtual halo rows/cols -1 and H/W is
                 // folded back onto rows/cols 1 and H-2/W-2 by extra, lane-masked tap steps on the border tiles
  int act;
  unsigned epi;
  int tilesX, tilesY, tilesN, nwg;
};

constexpr int LD = 20;

// The 16 extra tap steps of the reflection fold: weight tap, halo offsets, and which lane mask applies
// (rsel: 0 none, 1 output row == 1, 2 output row == H-2; csel likewise for columns).
struct FoldTap { int wtap, ao, bo, rsel, csel; };
__constant__ FoldTap kFoldTaps[16] = {
    // ky = 0 (top extra: source row offset 0, rows == 1)          normal row offset for ky: 2 - ky
    {0, 0, 2, 1, 0}, {0, 2, 0, 0, 1}, {0, 0, 0, 1, 1},   // (ky,kx)=(0,0): row-extra | col-extra | both
    {1, 0, 1, 1, 0},                                       // (0,1): row-extra
    {2, 0, 0, 1, 0}, {2, 2, 2, 0, 2}, {2, 0, 2, 1, 2},   // (0,2): row-extra | col-extra(right) | both
    {3, 1, 0, 0, 1},                                       // (1,0): col-extra(left)
    {5, 1, 2, 0, 2},                                       // (1,2): col-extra(right)
    {6, 2, 2, 2, 0}, {6, 0, 0, 0, 1}, {6, 2, 0, 2, 1},   // (2,0): row-extra(bottom) | col-extra(left) | both
    {7, 2, 1, 2, 0},                                       // (2,1): row-extra(bottom)
    {8, 2, 0, 2, 0}, {8, 0, 2, 0, 2}, {8, 2, 2, 2, 2},   // (2,2): row-extra | col-extra | both
};

__device__ __forceinline__ float tile_epilogue(const TileArgs& a, size_t o, int n, float v) {
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

// TH x TW output pixels, BN output channels, WM x WN waves, FLIP = data-gradient (taps mirrored)
template <int TH, int TW, int BN, int WM, int WN, bool FLIP, bool FOLD>
__global__ void __launch_bounds__(256) conv3x3_tile_kernel(const TileArgs a) {
  constexpr int BM = TH * TW;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int HW2 = TW + 2, HP = (TH + 2) * HW2;   // halo tile
  constexpr int NS = (HP * 4 + 255) / 256;           // float4 staging slots per thread
  static_assert(WM * WN == 4 && TW == 16 && BM % (WM * 32) == 0, "tile shape");
  __shared__ __attribute__((aligned(16))) float lds[2 * HP * LD];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, idx = lane & 31, h = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  int wg = fp_xcd_remap(blockIdx.x, a.nwg);
  const int tile_n = wg % a.tilesN; wg /= a.tilesN;
  const int tile_x = wg % a.tilesX; wg /= a.tilesX;
  const int tile_y = wg % a.tilesY;
  const int n_img = wg / a.tilesY;
  const int y0 = tile_y * TH, x0 = tile_x * TW, n0 = tile_n * BN;

  // ---- halo staging slots: source pixel of every halo pixel, computed once per workgroup ----------------------
  // Loads are UNCONDITIONAL (invalid slots read pixel 0 and are zeroed on the way into LDS): straight-line loads let hipcc
  // count vmcnt exactly, so waiting for a weight slice does not also wait for the younger halo prefetch.
  int pix0[NS], pix1[NS], lds_off[NS];
  bool hvalid[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const int lin = t + 256 * k, hp = lin >> 2;
    pix0[k] = pix1[k] = 0;
    hvalid[k] = false;
    lds_off[k] = hp < HP ? hp * LD + (lin & 3) * 4 : -1;
    if (hp < HP) {
      const int hy = hp / HW2, hx = hp - hy * HW2;
      int sy = y0 + hy - a.off, sx = x0 + hx - a.off;
      if (a.mode == 0) {
        hvalid[k] = sy >= 0 && sy < a.IH && sx >= 0 && sx < a.IW;
      } else {
        hvalid[k] = sy >= -1 && sy <= a.IH && sx >= -1 && sx <= a.IW;
        sy = fp_reflect(sy, a.IH);
        sx = fp_reflect(sx, a.IW);
      }
      // invalid slots read the nearest in-image pixel (distinct, cache-friendly addresses) and are zeroed on the way into LDS
      sy = min(max(sy, 0), a.IH - 1);
      sx = min(max(sx, 0), a.IW - 1);
      if (a.mode != 2) {
        pix0[k] = (n_img * a.IH + sy) * a.IW + sx;
      } else {
        pix0[k] = (n_img * (a.IH >> 1) + (sy >> 1)) * (a.IW >> 1) + (sx >> 1);
        pix1[k] = (n_img * a.IH + sy) * a.IW + sx;
      }
    }
  }
  float4 hreg[NS];
  bool hzero = false;     // this thread's channel quad lies beyond C0 + C1 (last, partial chunk)
  auto load_halo = [&](int cc) {
    const int c4 = cc * 16 + (t & 3) * 4;
    const bool from1 = c4 >= a.C0;
    hzero = c4 >= a.C0 + a.C1;
    const float* base = from1 ? a.src1 : a.src0;
    const int cstride = from1 ? a.C1 : a.C0;
    const int coff = hzero ? 0 : (from1 ? c4 - a.C0 : c4);
    if (from1 && a.C1 == 0) { base = a.src0; }            // (hzero) any valid address
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      const int px = (from1 && a.C1 > 0) ? pix1[k] : pix0[k];
      hreg[k] = *reinterpret_cast<const float4*>(base + (size_t)px * ((from1 && a.C1 == 0) ? a.C0 : cstride) + coff);
    }
  };
  auto store_halo = [&](int buf) {
#pragma unroll
    for (int k = 0; k < NS; ++k)
      if (lds_off[k] >= 0)
        *reinterpret_cast<float4*>(lds + buf * HP * LD + lds_off[k]) = (hvalid[k] && !hzero) ? hreg[k] : make_float4(0.f, 0.f, 0.f, 0.f);
  };

  // ---- B fragments straight from global (L1/L2-resident weight slices), TWO taps ahead in three rotating register sets
  // (9 taps per chunk = 3 full rotations, so the set index is static).  Columns beyond Nout re-read column Nout-1 (never stored).
  float4 bq[3][TN][2];
  auto load_b = [&](int tap, int cc, float4 (&bf)[TN][2]) {
    const float* ws = a.w + (size_t)(tap * a.KC16 + cc) * a.Nout * 16 + h * 4;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = min(n0 + (wn * TN + j) * 32 + idx, a.Nout - 1);
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) bf[j][kh] = *reinterpret_cast<const float4*>(ws + (size_t)n * 16 + kh * 8);
    }
  };

  // per-lane LDS base of each row block: tile pixel pt = block*32 + idx -> (pt / TW, pt % TW)
  int abase[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int pt = (wm * TM + i) * 32 + idx;
    abase[i] = ((pt / TW) * HW2 + (pt % TW)) * LD + h * 4;
  }

  // reflection-fold lane masks (data-gradient of a reflection-padded conv only)
  const bool has_r1 = y0 <= 1 && 1 < y0 + TH, has_rH = y0 <= a.OH - 2 && a.OH - 2 < y0 + TH;
  const bool has_c1 = x0 <= 1 && 1 < x0 + TW, has_cW = x0 <= a.OW - 2 && a.OW - 2 < x0 + TW;
  const bool border_tile = has_r1 || has_rH || has_c1 || has_cW;
  float m_r1[TM], m_rH[TM], m_c1[TM], m_cW[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int pt = (wm * TM + i) * 32 + idx;
    const int yy = y0 + pt / TW, xx = x0 + pt % TW;
    m_r1[i] = yy == 1 ? 1.f : 0.f;
    m_rH[i] = yy == a.OH - 2 ? 1.f : 0.f;
    m_c1[i] = xx == 1 ? 1.f : 0.f;
    m_cW[i] = xx == a.OW - 2 ? 1.f : 0.f;
  }

  // Two accumulator sets (k 0..7 and k 8..15 of every 16-channel chunk) so that consecutive MFMAs never target the
  // same accumulator: a non-MFMA instruction issued between two MFMAs on the SAME accumulator costs ~43 extra cycles
  // on gfx950 (MI355X_MICROARCH.md, per-instruction constants); with >= 2 independent chains the B loads, LDS reads
  // and address math interleave for free.  The two sets are summed in the epilogue.
  f32x16 acc[TM][TN][2];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][0][r] = acc[i][j][1][r] = 0.f;

  load_halo(0);
  store_halo(0);
  load_b(0, 0, bq[0]);
  load_b(1, 0, bq[1]);
  load_halo(min(1, a.KC16 - 1));     // halo tiles are fetched two chunks ahead (unconditionally: the tail re-reads the last chunk)
  __syncthreads();

  for (int cc = 0; cc < a.KC16; ++cc) {
    const float* Hb = lds + (cc & 1) * HP * LD;
    const int ccn = min(cc + 1, a.KC16 - 1);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap % 3;
      const int toff = ((FLIP ? 2 - ky : ky) * HW2 + (FLIP ? 2 - kx : kx)) * LD;
      float4 af[TM][2];
#pragma unroll
      for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i][kh] = *reinterpret_cast<const float4*>(Hb + abase[i] + toff + kh * 8);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (c == 1) {
          // Prefetch the weight slice of tap + 2 (wrapping into the next chunk).  All loads of this loop are unconditional and
          // in straight-line code, so hipcc emits exact vmcnt(N) waits: a tap waits for ITS slice only, never for the younger
          // slice or the halo prefetch behind it (measured: +4-5 % on the 96x320 / 192x640 layers over vmcnt(0) waits).
          __builtin_amdgcn_sched_barrier(0);
          // (unconditional: the last chunk re-reads itself, so the wait counts are the same on every path)
          if (tap < 7) load_b(tap + 2, cc, bq[(tap + 2) % 3]);
          else load_b(tap - 7, ccn, bq[(tap + 2) % 3]);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              const float4 bb = bq[tap % 3][j][kh];
              const float av = c == 0 ? af[i][kh].x : c == 1 ? af[i][kh].y : c == 2 ? af[i][kh].z : af[i][kh].w;
              const float bv = c == 0 ? bb.x : c == 1 ? bb.y : c == 2 ? bb.z : bb.w;
              acc[i][j][kh] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j][kh], 0, 0, 0);
            }
      }
    }
    if (FOLD && border_tile) {
      // reflection fold: dX[y][x] += sum over the extra (row, col) source choices of tap (ky, kx):
      //   ky == 0 & y == 1   -> source row 0   (halo row offset 0)      ky == 2 & y == H-2 -> source row H-1 (offset 2)
      //   kx == 0 & x == 1   -> source col 0                            kx == 2 & x == W-2 -> source col W-1
      // every combination except (normal row, normal col), which the nine regular taps above already did.
      // Rolled loop on purpose (rare path: border tiles only; keeps the register budget of the main loop).
#pragma unroll 1
      for (int e = 0; e < 16; ++e) {
        const FoldTap ft = kFoldTaps[e];
        const bool need_r = ft.rsel == 0 || (ft.rsel == 1 ? has_r1 : has_rH);
        const bool need_c = ft.csel == 0 || (ft.csel == 1 ? has_c1 : has_cW);
        if (!(need_r && need_c)) continue;                             // uniform per workgroup
        const int toff = (ft.ao * HW2 + ft.bo) * LD;
        float4 bx[TN][2];
        load_b(ft.wtap, cc, bx);
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            float4 av = *reinterpret_cast<const float4*>(Hb + abase[i] + toff + kh * 8);
            const float mr = ft.rsel == 0 ? 1.f : (ft.rsel == 1 ? m_r1[i] : m_rH[i]);
            const float mc = ft.csel == 0 ? 1.f : (ft.csel == 1 ? m_c1[i] : m_cW[i]);
            const float mk = mr * mc;
            av.x *= mk; av.y *= mk; av.z *= mk; av.w *= mk;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              acc[i][j][kh] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bx[j][kh].x, acc[i][j][kh], 0, 0, 0);
              acc[i][j][kh] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bx[j][kh].y, acc[i][j][kh], 0, 0, 0);
              acc[i][j][kh] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bx[j][kh].z, acc[i][j][kh], 0, 0, 0);
              acc[i][j][kh] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bx[j][kh].w, acc[i][j][kh], 0, 0, 0);
            }
          }
      }
    }
    if (cc + 1 < a.KC16) {
      store_halo((cc + 1) & 1);           // chunk cc+1 was fetched during chunk cc-1 (or in the prologue)
      load_halo(min(cc + 2, a.KC16 - 1));
      __syncthreads();                    // one barrier per 9 taps
    }
  }

  // ---- epilogue (C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)) --------------------------------
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + (wn * TN + j) * 32 + idx;
      if (n >= a.Nout) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int pt = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int oy = y0 + pt / TW, ox = x0 + pt % TW;
        if (oy >= a.OH || ox >= a.OW) continue;
        const size_t o = ((size_t)(n_img * a.OH + oy) * a.OW + ox) * a.Nout + n;
        a.y[o] = tile_epilogue(a, o, n, acc[i][j][0][r] + acc[i][j][1][r]);
      }
    }
}

template <int TH, int TW, int BN, int WM, int WN, bool FLIP, bool FOLD>
int launch_tile(TileArgs& a, hipStream_t stream) {
  a.tilesX = (int)fp_ceil_div(a.OW, TW);
  a.tilesY = (int)fp_ceil_div(a.OH, TH);
  a.tilesN = (int)fp_ceil_div(a.Nout, BN);
  a.nwg = a.N * a.tilesY * a.tilesX * a.tilesN;
  fp_launch((conv3x3_tile_kernel<TH, TW, BN, WM, WN, FLIP, FOLD>), dim3(a.nwg), dim3(256), 0, stream, a);
  return fp_check_launch("fp_conv_igemm(tile)");
}

}  // namespace

// Eligibility + dispatch, called by fp_conv_igemm.  Returns -1000 when the shape is not handled here.
int fp_conv3x3_tile_dispatch(const fp_conv_desc* d, const float* src0, const float* src1, const float* wpacked, const float* bias,
                             const float* addend, const float* addend_mask, const float* actsrc, float* y, hipStream_t stream) {
  if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != 1) return -1000;
  int mode, flip, fold = 0;
  switch (d->gather) {
    case FP_GATHER_FWD_ZERO: mode = 0; flip = 0; break;
    case FP_GATHER_FWD_REFLECT: mode = 1; flip = 0; break;
    case FP_GATHER_FWD_REFLECT_UP2: mode = 2; flip = 0; break;
    case FP_GATHER_DGRAD_ZERO: mode = 0; flip = 1; break;
    case FP_GATHER_DGRAD_REFLECT: mode = 0; flip = 1; fold = 1; break;
    default: return -1000;
  }
  if (d->OH != d->IH || d->OW != d->IW) return -1000;
  // tile waste and grid size: 8x16 tiles must cover the image without much padding and fill the chip
  const int64_t ty = fp_ceil_div(d->OH, 8), tx = fp_ceil_div(d->OW, 16);
  if (ty * 8 * tx * 16 * 4 > (int64_t)d->OH * d->OW * 5) return -1000;          // > 25 % padded work
  const int64_t wgs = (int64_t)d->N * ty * tx * fp_ceil_div(d->Nout, d->Nout <= 32 ? 32 : 64);
  if (wgs < 256) return -1000;                                                   // small grids: split-K kernel
  TileArgs a;
  a.src0 = src0; a.src1 = src1; a.w = wpacked; a.bias = bias; a.addend = addend; a.addend_mask = addend_mask;
  a.actsrc = actsrc; a.y = y;
  a.N = d->N; a.OH = d->OH; a.OW = d->OW; a.IH = d->IH; a.IW = d->IW; a.C0 = d->C0; a.C1 = d->C1; a.Nout = d->Nout;
  a.KC16 = (d->C0 + d->C1 + 15) / 16;
  a.mode = mode; a.off = 1; a.fold = fold; a.act = d->act; a.epi = d->epi;
  if (d->Nout <= 32) {
    if (fold) return launch_tile<8, 16, 32, 4, 1, true, true>(a, stream);
    return flip ? launch_tile<8, 16, 32, 4, 1, true, false>(a, stream) : launch_tile<8, 16, 32, 4, 1, false, false>(a, stream);
  }
  if (fold) return launch_tile<8, 16, 64, 2, 2, true, true>(a, stream);
  return flip ? launch_tile<8, 16, 64, 2, 2, true, false>(a, stream) : launch_tile<8, 16, 64, 2, 2, false, false>(a, stream);
}
