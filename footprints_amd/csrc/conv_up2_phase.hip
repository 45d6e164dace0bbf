// 3x3 reflection-padded convolution on a nearest-x2 upsampled input, WITHOUT the upsample and with 2.25x fewer MACs.
//
// nearest-x2 repeats every low-resolution pixel 2x2 times, so the nine taps of a hi-res output pixel only ever see
// 2x2 distinct low-res pixels.  For output phase (dy, dx) = (row parity, column parity):
//
//     out[2y+dy][2x+dx] = sum_{a,b in {0,1}}  Wc[dy,dx][a][b] . low[clamp(y + dy - 1 + a)][clamp(x + dx - 1 + b)]
//
// with collapsed weights  Wc[0][0] = W[ky=0], Wc[0][1] = W[1] + W[2],  Wc[1][0] = W[0] + W[1], Wc[1][1] = W[2]
// (same along x), and ReflectionPad2d(1) at hi-res == replicate (clamp) padding at low-res (hi-res row -1 mirrors to
// row 1 -> low-res row 0; row H mirrors to H-2 -> low-res row h-1).  Exact same mathematics as
// F.interpolate(nearest, x2) -> ReflectionPad2d(1) -> conv3x3 (reference footprints/network.py:98,154,126-134); only
// the floating-point association of the weight sums differs (1 ulp-level).
//
// Kernel = the halo-tile scheme of conv3x3_tile.hip on the LOW-RES grid: a workgroup owns one phase of an 8 x 16 tile of
// low-res positions (128 strided hi-res output pixels), stages the 10 x 18 low-res halo once per 16-channel chunk and
// runs the phase's four taps from LDS; weights [phase][tap][chunk][n][16] come straight from L1/L2.
// The skip half of a concat conv is done by conv3x3_tile.hip on the skip tensor alone; this kernel then adds its
// partial sums through the `addend` epilogue input (addend == y, in place) before bias / ELU.
#include "fp_common.h"

namespace {

constexpr int LD = 20, TH = 8, TW = 16, HW2 = TW + 2, HP = (TH + 2) * HW2;

struct PhaseArgs {
  const float* low;     // [N][h][w][C0]
  const float* w;       // [4 phases][4 taps][KC16][Nout][16]
  const float* bias;
  const float* addend;  // [N][2h][2w][Nout] or null
  float* y;             // [N][2h][2w][Nout]
  int N, h, w_, C0, Nout, KC16, act, tilesX, tilesY, tilesN, nwg;
  unsigned epi;
  const unsigned* amax_a;   // fp16-pair variant: amax slots of `low` and of the weights; optional slot receiving max |y|
  const unsigned* amax_w;
  unsigned* amax_out;
};

template <int BN, int WM, int WN>
__global__ void __launch_bounds__(256) up2_phase_fwd_kernel(const PhaseArgs a) {
  constexpr int BM = TH * TW;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int NS = (HP * 4 + 255) / 256;
  __shared__ __attribute__((aligned(16))) float lds[2 * HP * LD];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, idx = lane & 31, h = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  int wg = fp_xcd_remap(blockIdx.x, a.nwg);
  const int phase = wg & 3; wg >>= 2;                      // the four phases of a tile are adjacent (share the halo in L2)
  const int tile_n = wg % a.tilesN; wg /= a.tilesN;
  const int tile_x = wg % a.tilesX; wg /= a.tilesX;
  const int tile_y = wg % a.tilesY;
  const int n_img = wg / a.tilesY;
  const int dy = phase >> 1, dx = phase & 1;
  const int y0 = tile_y * TH, x0 = tile_x * TW, n0 = tile_n * BN;

  // all loads unconditional + straight-line (exact vmcnt waits, see conv3x3_tile.hip); replicate padding has no invalid pixels,
  // only the slots beyond the halo tile (not stored) and channel quads beyond C0 (stored as zero).
  int pix[NS], lds_off[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const int lin = t + 256 * k, hp = min(lin >> 2, HP - 1);
    lds_off[k] = (lin >> 2) < HP ? hp * LD + (lin & 3) * 4 : -1;
    const int hy = hp / HW2, hx = hp - hy * HW2;
    const int sy = min(max(y0 + hy - 1, 0), a.h - 1), sx = min(max(x0 + hx - 1, 0), a.w_ - 1);   // replicate padding
    pix[k] = (n_img * a.h + sy) * a.w_ + sx;
  }
  float4 hreg[NS];
  bool hzero = false;
  auto load_halo = [&](int cc) {
    const int c4 = cc * 16 + (t & 3) * 4;
    hzero = c4 >= a.C0;
    const int coff = hzero ? 0 : c4;
#pragma unroll
    for (int k = 0; k < NS; ++k) hreg[k] = *reinterpret_cast<const float4*>(a.low + (size_t)pix[k] * a.C0 + coff);
  };
  auto store_halo = [&](int buf) {
#pragma unroll
    for (int k = 0; k < NS; ++k)
      if (lds_off[k] >= 0) *reinterpret_cast<float4*>(lds + buf * HP * LD + lds_off[k]) = hzero ? make_float4(0.f, 0.f, 0.f, 0.f) : hreg[k];
  };
  float4 bq[4][TN][2];     // weight slices two taps ahead; 4 taps per chunk => the register set of a tap is static
  auto load_b = [&](int tap, int cc, float4 (&bf)[TN][2]) {
    const float* ws = a.w + (size_t)((phase * 4 + tap) * a.KC16 + cc) * a.Nout * 16 + h * 4;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = min(n0 + (wn * TN + j) * 32 + idx, a.Nout - 1);
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) bf[j][kh] = *reinterpret_cast<const float4*>(ws + (size_t)n * 16 + kh * 8);
    }
  };
  int abase[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int pt = (wm * TM + i) * 32 + idx;
    abase[i] = ((pt / TW + dy) * HW2 + (pt % TW) + dx) * LD + h * 4;   // phase offset folded into the base
  }
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
  load_halo(min(1, a.KC16 - 1));
  __syncthreads();
  for (int cc = 0; cc < a.KC16; ++cc) {
    const float* Hb = lds + (cc & 1) * HP * LD;
    const int ccn = min(cc + 1, a.KC16 - 1);
#pragma unroll
    for (int tap = 0; tap < 4; ++tap) {
      const int toff = ((tap >> 1) * HW2 + (tap & 1)) * LD;      // tap (a, b): halo offset (dy + a, dx + b)
      float4 af[TM][2];
#pragma unroll
      for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i][kh] = *reinterpret_cast<const float4*>(Hb + abase[i] + toff + kh * 8);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (c == 1) {
          __builtin_amdgcn_sched_barrier(0);
          if (tap < 2) load_b(tap + 2, cc, bq[tap + 2]);
          else load_b(tap - 2, ccn, bq[tap - 2]);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              const float4 bb = bq[tap][j][kh];
              const float av = c == 0 ? af[i][kh].x : c == 1 ? af[i][kh].y : c == 2 ? af[i][kh].z : af[i][kh].w;
              const float bv = c == 0 ? bb.x : c == 1 ? bb.y : c == 2 ? bb.z : bb.w;
              acc[i][j][kh] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j][kh], 0, 0, 0);
            }
      }
    }
    if (cc + 1 < a.KC16) {
      store_halo((cc + 1) & 1);
      load_halo(min(cc + 2, a.KC16 - 1));
      __syncthreads();
    }
  }
  const int OH = 2 * a.h, OW = 2 * a.w_;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + (wn * TN + j) * 32 + idx;
      if (n >= a.Nout) continue;
      const float bias = (a.epi & FP_EPI_BIAS) ? a.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int pt = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int ly = y0 + pt / TW, lx = x0 + pt % TW;
        if (ly >= a.h || lx >= a.w_) continue;
        const size_t o = ((size_t)(n_img * OH + 2 * ly + dy) * OW + 2 * lx + dx) * a.Nout + n;
        float v = acc[i][j][0][r] + acc[i][j][1][r] + bias;
        if (a.epi & FP_EPI_ADDEND) v += a.addend[o];
        if (a.act == FP_ACT_ELU) v = fp_elu(v);
        a.y[o] = v;
      }
    }
}

// ---- bf16x3-split variant (operands split exactly into three bf16 terms, six v_mfma_f32_32x32x16_bf16 products; see
// conv3x3_tile_bf3.hip).  Halo in LDS as three bf16 planes [plane][pixel][16 ch + 8 pad]; weights [phase][tap][chunk][plane][n][16].
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
constexpr int PIXB = 48, PLANE3 = HP * PIXB;      // bytes per halo pixel and per operand plane (a buffer holds NP planes)
// NP = 3: the exact bf16 split (six products); NP = 2: scaled fp16 pairs (fp_common.h; FP_HP_PRODUCTS products on v_mfma_f32_32x32x16_f16)
template <int NP>
__device__ __forceinline__ void phase_store_split(unsigned char* p, f32x4_t v, int ka) {
  if (NP == 2) {
    uint2 hq, mq;
    fp_hp_split4(v.x, v.y, v.z, v.w, ldexpf(1.f, ka), hq, mq);
    *reinterpret_cast<uint2*>(p) = hq;
    *reinterpret_cast<uint2*>(p + PLANE3) = mq;
  } else {
    const bf16x4_t vh = __builtin_convertvector(v, bf16x4_t);
    const f32x4_t r1 = v - __builtin_convertvector(vh, f32x4_t);
    const bf16x4_t vm = __builtin_convertvector(r1, bf16x4_t);
    const f32x4_t r2 = r1 - __builtin_convertvector(vm, f32x4_t);
    const bf16x4_t vl = __builtin_convertvector(r2, bf16x4_t);
    *reinterpret_cast<uint2*>(p) = __builtin_bit_cast(uint2, vh);
    *reinterpret_cast<uint2*>(p + PLANE3) = __builtin_bit_cast(uint2, vm);
    *reinterpret_cast<uint2*>(p + 2 * PLANE3) = __builtin_bit_cast(uint2, vl);
  }
}
template <int NP, int TM, int TN>
__device__ __forceinline__ void phase_mma(f32x16 (&acc)[TM][TN], const uint4 (&af)[TM][NP], const uint4 (&bf)[TN][NP]) {
  constexpr int NPROD = NP == 3 ? 6 : 4, Q0 = NP == 3 ? 0 : 4 - FP_HP_PRODUCTS;     // fp16 pairs: optionally without the mm product
  constexpr int PA[6] = {NP == 3 ? 2 : 1, NP == 3 ? 0 : 1, NP == 3 ? 1 : 0, NP == 3 ? 1 : 0, 0, 0};
  constexpr int PB[6] = {NP == 3 ? 0 : 1, NP == 3 ? 2 : 0, 1, 0, 1, 0};            // smallest products first
#pragma unroll
  for (int q = Q0; q < NPROD; ++q)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        if (NP == 2)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, af[i][PA[q]]), __builtin_bit_cast(f16x8_t, bf[j][PB[q]]),
                                                             acc[i][j], 0, 0, 0);
        else
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af[i][PA[q]]), __builtin_bit_cast(bf16x8_t, bf[j][PB[q]]),
                                                              acc[i][j], 0, 0, 0);
}

template <int BN, int WM, int WN, int NP = 3>
__global__ void __launch_bounds__(256) up2_phase_fwd_bf3_kernel(const PhaseArgs a) {
  constexpr int BUFN = NP * PLANE3;
  constexpr int BM = TH * TW;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int NS = (HP * 4 + 255) / 256;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUFN];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, idx = lane & 31, h = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  int wg = fp_xcd_remap(blockIdx.x, a.nwg);
  const int phase = wg & 3; wg >>= 2;
  const int tile_n = wg % a.tilesN; wg /= a.tilesN;
  const int tile_x = wg % a.tilesX; wg /= a.tilesX;
  const int tile_y = wg % a.tilesY;
  const int n_img = wg / a.tilesY;
  const int dy = phase >> 1, dx = phase & 1;
  const int y0 = tile_y * TH, x0 = tile_x * TW, n0 = tile_n * BN;
  const unsigned short* wq = reinterpret_cast<const unsigned short*>(a.w);
  int ka = 0, kunscale = 0;
  if (NP == 2) {
    unsigned ma, mw, unused;
    fp_amax3_reduce(fp_amax3_issue(a.amax_a, a.amax_w, nullptr), ma, mw, unused);           // one round trip for both slots (fp_common.h)
    ka = fp_hp_exponent(ma, FP_HP_TARGET_ACT);
    kunscale = -(ka + fp_hp_exponent(mw, FP_HP_TARGET_W));
  }

  // operands through raw buffer loads: a 32-bit per-lane byte offset + a wave-uniform one (conv3x3_tile_bf3.hip, round 4)
  const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.low), 0, a.N * a.h * a.w_ * a.C0 * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(wq), 0, 16 * a.KC16 * NP * a.Nout * 32, 0x00020000);
  const int wchunk = NP * a.Nout * 32, wplane = a.Nout * 32;
  int wtapoff[4];
#pragma unroll
  for (int tp = 0; tp < 4; ++tp) wtapoff[tp] = (phase * 4 + tp) * a.KC16 * wchunk;
  unsigned voff[NS];
  int lds_off[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const int lin = t + 256 * k, hp = min(lin >> 2, HP - 1);
    lds_off[k] = (lin >> 2) < HP ? hp * PIXB + (lin & 3) * 8 : -1;
    const int hy = hp / HW2, hx = hp - hy * HW2;
    const int sy = min(max(y0 + hy - 1, 0), a.h - 1), sx = min(max(x0 + hx - 1, 0), a.w_ - 1);   // replicate padding
    voff[k] = (unsigned)((n_img * a.h + sy) * a.w_ + sx) * (unsigned)(a.C0 * 4) + (t & 3) * 16;
  }
  float4 hreg[NS];
  bool hzero = false;
  auto load_halo = [&](int cc) {
    hzero = (a.C0 & 15) != 0 && cc * 16 + (t & 3) * 4 >= a.C0;
#pragma unroll
    for (int k = 0; k < NS; ++k) hreg[k] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rl, voff[k], cc * 64, 0));
  };
  auto store_halo = [&](int buf) {
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      if (lds_off[k] < 0) continue;
      f32x4_t v = {hreg[k].x, hreg[k].y, hreg[k].z, hreg[k].w};
      if (hzero) v = f32x4_t{0.f, 0.f, 0.f, 0.f};
      phase_store_split<NP>(lds + buf * BUFN + lds_off[k], v, ka);
    }
  };
  uint4 bq[4][TN][NP];
  unsigned wvoff[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) wvoff[j] = (unsigned)(min(n0 + (wn * TN + j) * 32 + idx, a.Nout - 1) * 32 + h * 16);
  auto load_b = [&](int tap, int cc, uint4 (&bf)[TN][NP]) {
    const int so = wtapoff[tap] + cc * wchunk;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int p = 0; p < NP; ++p) bf[j][p] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rw, wvoff[j], so + p * wplane, 0));
  };
  int abase[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int pt = (wm * TM + i) * 32 + idx;
    abase[i] = ((pt / TW + dy) * HW2 + (pt % TW) + dx) * PIXB + h * 16;
  }
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  load_halo(0);
  load_b(0, 0, bq[0]);
  load_b(1, 0, bq[1]);
  store_halo(0);
  load_halo(min(1, a.KC16 - 1));
  __syncthreads();
  for (int cc = 0; cc < a.KC16; ++cc) {
    const unsigned char* Hb = lds + (cc & 1) * BUFN;
    const int ccn = min(cc + 1, a.KC16 - 1);
    uint4 af[2][TM][NP];                                     // A fragments one tap ahead (see conv3x3_tile_bf3.hip)
    auto load_a = [&](int tap, uint4 (&dst)[TM][NP]) {
      const int toff = ((tap >> 1) * HW2 + (tap & 1)) * PIXB;
#pragma unroll
      for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int i = 0; i < TM; ++i) dst[i][p] = *reinterpret_cast<const uint4*>(Hb + p * PLANE3 + abase[i] + toff);
    };
    load_a(0, af[0]);
#pragma unroll
    for (int tap = 0; tap < 4; ++tap) {
      __builtin_amdgcn_sched_barrier(0);
      if (tap < 3) load_a(tap + 1, af[(tap + 1) & 1]);
      if (tap < 2) load_b(tap + 2, cc, bq[tap + 2]);
      else load_b(tap - 2, ccn, bq[tap - 2]);
      phase_mma<NP, TM, TN>(acc, af[tap & 1], bq[tap]);
      if (tap < 3) fp_sched_interleave<TM * NP, TN * NP, (NP == 3 ? 6 : FP_HP_PRODUCTS) * TM * TN>();      // one read between consecutive MFMAs (fp_common.h)
      else fp_sched_interleave<0, TN * NP, (NP == 3 ? 6 : FP_HP_PRODUCTS) * TM * TN>();
    }
    if (cc + 1 < a.KC16) {
      store_halo((cc + 1) & 1);
      load_halo(min(cc + 2, a.KC16 - 1));
      __syncthreads();
    }
  }
  const int OH = 2 * a.h, OW = 2 * a.w_;
  float ymax = 0.f;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + (wn * TN + j) * 32 + idx;
      if (n >= a.Nout) continue;
      const float bias = (a.epi & FP_EPI_BIAS) ? a.bias[n] : 0.f;
      // addend loads of all 16 rows first, then the arithmetic and the stores (see conv3x3_tile_bf3.hip); 32-bit unsigned byte offsets
      // (the host checks the output size): `global_* v, v_offset, s[base]`, no 64-bit address arithmetic per element
      unsigned off[16];
      bool ok[16];
      float ad[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int pt = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int ly = y0 + pt / TW, lx = x0 + pt % TW;
        ok[r] = ly < a.h && lx < a.w_;
        off[r] = (unsigned)(((n_img * OH + 2 * min(ly, a.h - 1) + dy) * OW + 2 * min(lx, a.w_ - 1) + dx) * a.Nout + n) * 4u;
      }
      if (a.epi & FP_EPI_ADDEND) {
#pragma unroll
        for (int r = 0; r < 16; ++r) ad[r] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.addend) + off[r]);
      }
      const float unscale = NP == 2 ? ldexpf(1.f, kunscale) : 1.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = NP == 2 ? fmaf(acc[i][j][r], unscale, bias) : acc[i][j][r] + bias;       // * 2^kunscale is exact
        if (a.epi & FP_EPI_ADDEND) v += ad[r];
        if (a.act == FP_ACT_ELU) v = fp_elu(v);
        if (ok[r]) {
          *reinterpret_cast<float*>(reinterpret_cast<char*>(a.y) + off[r]) = v;
          ymax = fmaxf(ymax, fabsf(v));
        }
      }
    }
  if (a.amax_out) fp_amax_publish_block(a.amax_out, ymax);
}

// ---- data gradient of the upsampled half, bf16x3: the 4x4 stride-2 convolution over dZ (see the comment further down), phase by
// phase.  For dZ phase (py, px) (hi-res pixel (2yp + py, 2xp + px)) only the taps r = (py+1)%2 + 2a, s = (px+1)%2 + 2b (a, b in
// {0,1}) hit it, at phase-plane position yp = Y' - 2 + a + (1 - py): per phase a 2x2 convolution over the zero-padded phase plane.
// A workgroup owns 8 x 16 positions of the extended (h+2) x (w+2) output grid x BN input channels and runs 4 phases x Cout/16
// chunks, each like a chunk of up2_phase_fwd_bf3_kernel (halo = 10 x 18 phase-plane pixels, four taps from LDS).
struct PhaseDgradArgs {
  const float* dz;      // [N][2h][2w][Cout]
  const unsigned short* w;   // bf16 [phase 4][tap 4][KC16 over Cout][3][C0][16]
  float* ext;           // [N][h+2][w+2][C0]
  int N, h, w_, Cout, C0, KC16, tilesX, tilesY, tilesN, nwg;
  const unsigned* amax_a;   // fp16-pair variant: amax slots of dz and of the weights
  const unsigned* amax_w;
};

template <int BN, int WM, int WN, int NP = 3>
__global__ void __launch_bounds__(256) up2_phase_dgrad_bf3_kernel(const PhaseDgradArgs a) {
  constexpr int BUFN = NP * PLANE3;
  constexpr int BM = TH * TW;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int NS = (HP * 4 + 255) / 256;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUFN];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, idx = lane & 31, h = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  int wg = fp_xcd_remap(blockIdx.x, a.nwg);
  const int tile_n = wg % a.tilesN; wg /= a.tilesN;
  const int tile_x = wg % a.tilesX; wg /= a.tilesX;
  const int tile_y = wg % a.tilesY;
  const int n_img = wg / a.tilesY;
  const int y0 = tile_y * TH, x0 = tile_x * TW, n0 = tile_n * BN;
  const int H2 = 2 * a.h, W2 = 2 * a.w_, EH = a.h + 2, EW = a.w_ + 2;
  const int nsteps = 4 * a.KC16;
  int ka = 0, kunscale = 0;
  if (NP == 2) {
    unsigned ma, mw, unused;
    fp_amax3_reduce(fp_amax3_issue(a.amax_a, a.amax_w, nullptr), ma, mw, unused);           // one round trip for both slots (fp_common.h)
    ka = fp_hp_exponent(ma, FP_HP_TARGET_ACT);
    kunscale = -(ka + fp_hp_exponent(mw, FP_HP_TARGET_W));
  }

  // operands through raw buffer loads (32-bit lane offset + wave-uniform offset; an offset with bit 31 set reads zeros: the zero padding
  // of the phase planes needs no select at staging time); (phase, chunk) of a step advance by carries instead of a division per use
  constexpr unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dz), 0, a.N * H2 * W2 * a.Cout * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.w), 0, 16 * a.KC16 * NP * a.C0 * 32, 0x00020000);
  const int wchunk = NP * a.C0 * 32, wplane = a.C0 * 32, wtap = a.KC16 * wchunk;
  unsigned voff[NS];
  int lds_off[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const int lin = t + 256 * k, hp = min(lin >> 2, HP - 1);
    lds_off[k] = (lin >> 2) < HP ? hp * PIXB + (lin & 3) * 8 : -1;
    const int hy = hp / HW2, hx = hp - hy * HW2;
    const int yp = y0 + hy - 2, xp = x0 + hx - 2;                      // phase-plane coordinates
    const bool valid = yp >= 0 && yp < a.h && xp >= 0 && xp < a.w_;
    const unsigned pix = (unsigned)((n_img * H2 + 2 * min(max(yp, 0), a.h - 1)) * W2 + 2 * min(max(xp, 0), a.w_ - 1));   // + py * W2 + px per phase
    voff[k] = valid ? pix * (unsigned)(a.Cout * 4) + (t & 3) * 16 : OOB;
  }
  float4 hreg[NS];
  bool hzero = false;
  auto load_halo = [&](int ph, int cc) {
    const int so = (((ph >> 1) * W2 + (ph & 1)) * a.Cout + cc * 16) * 4;
    hzero = (a.Cout & 15) != 0 && cc * 16 + (t & 3) * 4 >= a.Cout;
#pragma unroll
    for (int k = 0; k < NS; ++k) hreg[k] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rz, voff[k], so, 0));
  };
  auto store_halo = [&](int buf) {
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      if (lds_off[k] < 0) continue;
      f32x4_t v = {hreg[k].x, hreg[k].y, hreg[k].z, hreg[k].w};
      if (hzero) v = f32x4_t{0.f, 0.f, 0.f, 0.f};
      phase_store_split<NP>(lds + buf * BUFN + lds_off[k], v, ka);
    }
  };
  uint4 bq[4][TN][NP];
  unsigned wvoff[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) wvoff[j] = (unsigned)(min(n0 + (wn * TN + j) * 32 + idx, a.C0 - 1) * 32 + h * 16);
  auto load_b = [&](int tap, int ph, int cc, uint4 (&bf)[TN][NP]) {
    const int so = (ph * 4 + tap) * wtap + cc * wchunk;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int p = 0; p < NP; ++p) bf[j][p] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rw, wvoff[j], so + p * wplane, 0));
  };
  int abase[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int pt = (wm * TM + i) * 32 + idx;
    abase[i] = ((pt / TW) * HW2 + (pt % TW)) * PIXB + h * 16;
  }
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // (ph, cc) of step, step + 1 and step + 2 (clamped to the last step), advanced by carries
  auto advance = [&](int& ph_, int& cc_) {
    if (ph_ * a.KC16 + cc_ + 1 >= nsteps) return;      // stay on the last step (its operands are re-loaded, harmlessly)
    if (++cc_ == a.KC16) { cc_ = 0; ++ph_; }
  };
  int ph = 0, cc = 0, ph1 = 0, cc1 = 0, ph2 = 0, cc2 = 0;
  advance(ph1, cc1);
  ph2 = ph1; cc2 = cc1;
  advance(ph2, cc2);
  load_halo(0, 0);
  load_b(0, 0, 0, bq[0]);
  load_b(1, 0, 0, bq[1]);
  store_halo(0);
  load_halo(ph1, cc1);
  __syncthreads();
  for (int step = 0; step < nsteps; ++step) {
    const unsigned char* Hb = lds + (step & 1) * BUFN;
    const int poff = ((1 - (ph >> 1)) * HW2 + (1 - (ph & 1))) * PIXB;     // tap (a, b) reads halo offset (a + 1 - py, b + 1 - px)
    uint4 af[2][TM][NP];
    auto load_a = [&](int tap, uint4 (&dst)[TM][NP]) {
      const int toff = poff + ((tap >> 1) * HW2 + (tap & 1)) * PIXB;
#pragma unroll
      for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int i = 0; i < TM; ++i) dst[i][p] = *reinterpret_cast<const uint4*>(Hb + p * PLANE3 + abase[i] + toff);
    };
    load_a(0, af[0]);
#pragma unroll
    for (int tap = 0; tap < 4; ++tap) {
      __builtin_amdgcn_sched_barrier(0);
      if (tap < 3) load_a(tap + 1, af[(tap + 1) & 1]);
      if (tap < 2) load_b(tap + 2, ph, cc, bq[tap + 2]);
      else load_b(tap - 2, ph1, cc1, bq[tap - 2]);
      phase_mma<NP, TM, TN>(acc, af[tap & 1], bq[tap]);
      if (tap < 3) fp_sched_interleave<TM * NP, TN * NP, (NP == 3 ? 6 : FP_HP_PRODUCTS) * TM * TN>();      // one read between consecutive MFMAs (fp_common.h)
      else fp_sched_interleave<0, TN * NP, (NP == 3 ? 6 : FP_HP_PRODUCTS) * TM * TN>();
    }
    if (step + 1 < nsteps) {
      store_halo((step + 1) & 1);
      load_halo(ph2, cc2);
      __syncthreads();
    }
    ph = ph1; cc = cc1; ph1 = ph2; cc1 = cc2;
    advance(ph2, cc2);
  }
  const float unscale_d = NP == 2 ? ldexpf(1.f, kunscale) : 1.f;      // * 2^kunscale is exact
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + (wn * TN + j) * 32 + idx;
      if (n >= a.C0) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int pt = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int ey = y0 + pt / TW, ex = x0 + pt % TW;
        const unsigned boff = (unsigned)(((n_img * EH + ey) * EW + ex) * a.C0 + n) * 4u;     // (output smaller than 2^29 elements: host check)
        if (ey < EH && ex < EW)
          *reinterpret_cast<float*>(reinterpret_cast<char*>(a.ext) + boff) = NP == 2 ? acc[i][j][r] * unscale_d : acc[i][j][r];
      }
    }
}

// ---- backward ----------------------------------------------------------------------------------------------------
// d(low) of the phase decomposition is a 4x4 stride-2 convolution over dZ (taps r = hi-res row - (2Y - 1)):
//     K4[0] = W[2], K4[1] = W[1] + W[2], K4[2] = W[0] + W[1], K4[3] = W[0]         (rows; same for columns)
// plus the replicate-padding fold: the virtual low-res rows -1 / h (columns -1 / w) receive gradient too and belong to
// rows 0 / h-1.  fp_conv_igemm (FWD_ZERO gather, K=4, stride 2, pad 3) evaluates the conv on the (h+2) x (w+2) extended grid
// with these packed weights; up2_fold_bwd_kernel folds the border back, adds the second consumer's gradient and applies ELU'.
// (packing: pack.hip, FP_PACK_UP2_DGRAD)
__global__ void __launch_bounds__(256) up2_fold_bwd_kernel(const float* __restrict__ ext, int N, int h, int w, int C,
                                                           const float* __restrict__ addend, const float* __restrict__ ylow,
                                                           float* __restrict__ dlow, unsigned* amax_out) {
  const int Q = C >> 2, we = w + 2;
  const size_t total = (size_t)N * h * w * Q;
  float ymax = 0.f;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int q = (int)(e % Q);
    size_t r = e / Q;
    const int x = (int)(r % w); r /= w;
    const int y = (int)(r % h);
    const int n = (int)(r / h);
    // extended rows / columns folded onto this pixel (fixed order: centre, then the border copies)
    int ys[3], xs[3], ny = 1, nx = 1;
    ys[0] = y + 1; xs[0] = x + 1;
    if (y == 0) ys[ny++] = 0;
    if (y == h - 1) ys[ny++] = h + 1;
    if (x == 0) xs[nx++] = 0;
    if (x == w - 1) xs[nx++] = w + 1;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = 0; i < ny; ++i)
      for (int j = 0; j < nx; ++j) {
        const float4 v = *reinterpret_cast<const float4*>(ext + (((size_t)n * (h + 2) + ys[i]) * we + xs[j]) * C + q * 4);
        g.x += v.x; g.y += v.y; g.z += v.z; g.w += v.w;
      }
    const size_t o = (e / Q) * C + q * 4;
    if (addend) {
      const float4 ad = *reinterpret_cast<const float4*>(addend + o);
      g.x += ad.x; g.y += ad.y; g.z += ad.z; g.w += ad.w;
    }
    if (ylow) {
      const float4 s = *reinterpret_cast<const float4*>(ylow + o);
      g.x *= (s.x > 0.f ? 1.f : s.x + 1.f); g.y *= (s.y > 0.f ? 1.f : s.y + 1.f);
      g.z *= (s.z > 0.f ? 1.f : s.z + 1.f); g.w *= (s.w > 0.f ? 1.f : s.w + 1.f);
    }
    *reinterpret_cast<float4*>(dlow + o) = g;
    ymax = fp_amax4(ymax, g);
  }
  if (amax_out) fp_amax_publish_block(amax_out, ymax);
}

int grid_for(size_t total) {
  size_t g = (total + 255) / 256;
  return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int fp_conv_up2_phase_fwd(const float* low, const float* wphase, const float* bias, const float* addend, float* y,
                                     int32_t N, int32_t h, int32_t w, int32_t C0, int32_t Nout, int32_t act, fp_stream_t stream) {
  FP_REQUIRE(low && wphase && y && N > 0 && h >= 1 && w >= 1 && C0 > 0 && C0 % 4 == 0 && Nout > 0, "fp_conv_up2_phase_fwd: bad arguments");
  PhaseArgs a;
  a.amax_a = a.amax_w = nullptr; a.amax_out = nullptr;
  a.low = low; a.w = wphase; a.bias = bias; a.addend = addend; a.y = y;
  a.N = N; a.h = h; a.w_ = w; a.C0 = C0; a.Nout = Nout; a.KC16 = (C0 + 15) / 16; a.act = act;
  a.epi = (bias ? FP_EPI_BIAS : 0u) | (addend ? FP_EPI_ADDEND : 0u);
  a.tilesX = (int)fp_ceil_div(w, TW); a.tilesY = (int)fp_ceil_div(h, TH);
  if (Nout <= 32) {
    a.tilesN = (int)fp_ceil_div(Nout, 32);
    a.nwg = N * a.tilesY * a.tilesX * a.tilesN * 4;
    fp_launch((up2_phase_fwd_kernel<32, 4, 1>), dim3(a.nwg), dim3(256), 0, (hipStream_t)stream, a);
  } else {
    a.tilesN = (int)fp_ceil_div(Nout, 64);
    a.nwg = N * a.tilesY * a.tilesX * a.tilesN * 4;
    fp_launch((up2_phase_fwd_kernel<64, 2, 2>), dim3(a.nwg), dim3(256), 0, (hipStream_t)stream, a);
  }
  return fp_check_launch("fp_conv_up2_phase_fwd");
}

static int phase_fwd_split(const float* low, const void* wphase_bf3, const float* bias, const float* addend, float* y, int32_t N, int32_t h,
                           int32_t w, int32_t C0, int32_t Nout, int32_t act, const uint32_t* amax_low, const uint32_t* amax_w,
                           uint32_t* amax_out, fp_stream_t stream) {
  FP_REQUIRE(low && wphase_bf3 && y && N > 0 && h >= 1 && w >= 1 && C0 > 0 && C0 % 4 == 0 && Nout > 0,
             "fp_conv_up2_phase_fwd_bf3 / _hp: bad arguments");
  FP_REQUIRE((int64_t)N * 4 * h * w * Nout < ((int64_t)1 << 29) && (int64_t)N * h * w * C0 < ((int64_t)1 << 29),
             "fp_conv_up2_phase_fwd_bf3 / _hp: tensor larger than 2^29 elements (operands are addressed with 32-bit byte offsets)");
  PhaseArgs a;
  a.amax_a = amax_low; a.amax_w = amax_w; a.amax_out = amax_out;
  const bool hp = amax_low != nullptr;
  a.low = low; a.w = (const float*)wphase_bf3; a.bias = bias; a.addend = addend; a.y = y;
  a.N = N; a.h = h; a.w_ = w; a.C0 = C0; a.Nout = Nout; a.KC16 = (C0 + 15) / 16; a.act = act;
  a.epi = (bias ? FP_EPI_BIAS : 0u) | (addend ? FP_EPI_ADDEND : 0u);
  a.tilesX = (int)fp_ceil_div(w, TW); a.tilesY = (int)fp_ceil_div(h, TH);
  if (Nout <= 32) {
    a.tilesN = (int)fp_ceil_div(Nout, 32);
    a.nwg = N * a.tilesY * a.tilesX * a.tilesN * 4;
    if (hp) fp_launch((up2_phase_fwd_bf3_kernel<32, 4, 1, 2>), dim3(a.nwg), dim3(256), 0, (hipStream_t)stream, a);
    else fp_launch((up2_phase_fwd_bf3_kernel<32, 4, 1, 3>), dim3(a.nwg), dim3(256), 0, (hipStream_t)stream, a);
  } else {
    a.tilesN = (int)fp_ceil_div(Nout, 64);
    a.nwg = N * a.tilesY * a.tilesX * a.tilesN * 4;
    if (hp) fp_launch((up2_phase_fwd_bf3_kernel<64, 2, 2, 2>), dim3(a.nwg), dim3(256), 0, (hipStream_t)stream, a);
    else fp_launch((up2_phase_fwd_bf3_kernel<64, 2, 2, 3>), dim3(a.nwg), dim3(256), 0, (hipStream_t)stream, a);
  }
  return fp_check_launch("fp_conv_up2_phase_fwd_bf3 / _hp");
}
extern "C" int fp_conv_up2_phase_fwd_bf3(const float* low, const void* wphase_bf3, const float* bias, const float* addend, float* y,
                                         int32_t N, int32_t h, int32_t w, int32_t C0, int32_t Nout, int32_t act, const fp_aux* aux,
                                         fp_stream_t stream) {
  return phase_fwd_split(low, wphase_bf3, bias, addend, y, N, h, w, C0, Nout, act, nullptr, nullptr, fp_amax_out_of(aux), stream);
}
// fp16-pair operands (fp_conv3x3_hp): weights from FP_PACK_UP2_FWD_HP jobs / fp_pack_up2_weight_hp with the slot they were scaled by
extern "C" int fp_conv_up2_phase_fwd_hp(const float* low, const void* wphase_hp, const float* bias, const float* addend, float* y, int32_t N,
                                        int32_t h, int32_t w, int32_t C0, int32_t Nout, int32_t act, const uint32_t* amax_low,
                                        const uint32_t* amax_w, uint32_t* amax_out, fp_stream_t stream) {
  FP_REQUIRE(amax_low && amax_w, "fp_conv_up2_phase_fwd_hp: amax slots missing");
  return phase_fwd_split(low, wphase_hp, bias, addend, y, N, h, w, C0, Nout, act, amax_low, amax_w, amax_out, stream);
}

static int phase_dgrad_split(const float* dz, const void* wpacked_bf3, float* ext, int32_t N, int32_t h, int32_t w, int32_t Cout, int32_t C0,
                             const uint32_t* amax_dz, const uint32_t* amax_w, fp_stream_t stream) {
  FP_REQUIRE(dz && wpacked_bf3 && ext && N > 0 && h >= 1 && w >= 1 && Cout > 0 && Cout % 4 == 0 && C0 > 0,
             "fp_conv_up2_phase_dgrad_bf3 / _hp: bad arguments");
  FP_REQUIRE((int64_t)N * 4 * h * w * Cout < ((int64_t)1 << 29) && (int64_t)N * (h + 2) * (w + 2) * C0 < ((int64_t)1 << 29),
             "fp_conv_up2_phase_dgrad_bf3 / _hp: tensor larger than 2^29 elements (operands are addressed with 32-bit byte offsets)");
  PhaseDgradArgs a;
  a.amax_a = amax_dz; a.amax_w = amax_w;
  const bool hp = amax_dz != nullptr;
  a.dz = dz; a.w = (const unsigned short*)wpacked_bf3; a.ext = ext;
  a.N = N; a.h = h; a.w_ = w; a.Cout = Cout; a.C0 = C0; a.KC16 = (Cout + 15) / 16;
  a.tilesX = (int)fp_ceil_div(w + 2, TW); a.tilesY = (int)fp_ceil_div(h + 2, TH);
  if (C0 <= 32) {
    a.tilesN = (int)fp_ceil_div(C0, 32);
    a.nwg = N * a.tilesY * a.tilesX * a.tilesN;
    if (hp) fp_launch((up2_phase_dgrad_bf3_kernel<32, 4, 1, 2>), dim3(a.nwg), dim3(256), 0, (hipStream_t)stream, a);
    else fp_launch((up2_phase_dgrad_bf3_kernel<32, 4, 1, 3>), dim3(a.nwg), dim3(256), 0, (hipStream_t)stream, a);
  } else {
    a.tilesN = (int)fp_ceil_div(C0, 64);
    a.nwg = N * a.tilesY * a.tilesX * a.tilesN;
    if (hp) fp_launch((up2_phase_dgrad_bf3_kernel<64, 2, 2, 2>), dim3(a.nwg), dim3(256), 0, (hipStream_t)stream, a);
    else fp_launch((up2_phase_dgrad_bf3_kernel<64, 2, 2, 3>), dim3(a.nwg), dim3(256), 0, (hipStream_t)stream, a);
  }
  return fp_check_launch("fp_conv_up2_phase_dgrad_bf3 / _hp");
}
extern "C" int fp_conv_up2_phase_dgrad_bf3(const float* dz, const void* wpacked_bf3, float* ext, int32_t N, int32_t h, int32_t w,
                                           int32_t Cout, int32_t C0, fp_stream_t stream) {
  return phase_dgrad_split(dz, wpacked_bf3, ext, N, h, w, Cout, C0, nullptr, nullptr, stream);
}
extern "C" int fp_conv_up2_phase_dgrad_hp(const float* dz, const void* wpacked_hp, float* ext, int32_t N, int32_t h, int32_t w, int32_t Cout,
                                          int32_t C0, const uint32_t* amax_dz, const uint32_t* amax_w, fp_stream_t stream) {
  FP_REQUIRE(amax_dz && amax_w, "fp_conv_up2_phase_dgrad_hp: amax slots missing");
  return phase_dgrad_split(dz, wpacked_hp, ext, N, h, w, Cout, C0, amax_dz, amax_w, stream);
}

extern "C" int fp_up2_fold_bwd(const float* ext, int32_t N, int32_t h, int32_t w, int32_t C, const float* addend, const float* ylow_elu,
                               float* dlow, const fp_aux* aux, fp_stream_t stream) {
  unsigned* amax_out = fp_amax_out_of(aux);
  FP_REQUIRE(ext && dlow && N > 0 && h > 0 && w > 0 && C > 0 && C % 4 == 0, "fp_up2_fold_bwd: bad arguments");
  int g = grid_for((size_t)N * h * w * (C / 4));
  fp_launch(up2_fold_bwd_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, ext, N, h, w, C, addend, ylow_elu, dlow, amax_out);
  return fp_check_launch("fp_up2_fold_bwd");
}
