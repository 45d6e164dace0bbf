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
  float* bpart;       // [S][Nout] column sums of dZ (bias gradient) or null -- bf16x3 kernel only, written by the ci-tile-0 workgroups
  int N, h, w, C0, Nout;
  int chunksY, chunksX, nchunks, chunksPerSplit, S, citiles, cotiles;
  const unsigned* amax_low;   // fp16-pair variant: amax slots of low and dz
  const unsigned* amax_dz;
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

// ---- the same gradient with exactly split bf16x3 operands (see wgrad3x3_bf3.hip for the scheme) -------------------------------
// Contraction over pixels => both operands are transposed to "8 consecutive low-res columns of one channel" and split into three
// bf16 planes on their way into LDS (thread = 4 columns x 4 channels).  Chunk = 2 x 16 low-res positions:
//     Xs[plane][halo row 4][ci 32][20 cols, 48-byte lines]      Zs[plane][phase 4][row 2][co 32][16 cols, 32-byte lines]   (43 KB)
// Wave p owns phase p = (dy, dx): per chunk row one 16-column k-step per tap (a, b), the tap's operand being halo row
// r + dy + a shifted by dx + b columns (funnel shift of the aligned read), six v_mfma_f32_32x32x16_bf16 each: 48 MFMAs x 32 cycles
// per wave per chunk instead of 64 x 64.  No cross-wave sum (a wave writes its own four tap tiles); same partial layout as above.
typedef __bf16 pbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 pbf16x4 __attribute__((ext_vector_type(4)));
typedef float pf32x4 __attribute__((ext_vector_type(4)));

constexpr int PXROW = 48, PZROW = 32;
constexpr int PXPLANE = XR * 32 * PXROW, PZPLANE = 4 * CHL * 32 * PZROW;      // 6144, 8192 bytes

typedef _Float16 pf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 pf16x4 __attribute__((ext_vector_type(4)));
// NP = 3: exact bf16 split; NP = 2: fp16 pair of v * 2^kscale (fp_common.h)
template <int NP>
__device__ __forceinline__ void psplit_store(unsigned char* p, int plane_stride, pf32x4 v, int kscale) {
  if (NP == 2) {
    uint2 hq, mq;
    fp_hp_split4(v.x, v.y, v.z, v.w, ldexpf(1.f, kscale), hq, mq);
    *reinterpret_cast<uint2*>(p) = hq;
    *reinterpret_cast<uint2*>(p + plane_stride) = mq;
    return;
  }
  const pbf16x4 vh = __builtin_convertvector(v, pbf16x4);
  const pf32x4 r1 = v - __builtin_convertvector(vh, pf32x4);
  const pbf16x4 vm = __builtin_convertvector(r1, pbf16x4);
  const pf32x4 r2 = r1 - __builtin_convertvector(vm, pf32x4);
  const pbf16x4 vl = __builtin_convertvector(r2, pbf16x4);
  *reinterpret_cast<uint2*>(p) = __builtin_bit_cast(uint2, vh);
  *reinterpret_cast<uint2*>(p + plane_stride) = __builtin_bit_cast(uint2, vm);
  *reinterpret_cast<uint2*>(p + 2 * plane_stride) = __builtin_bit_cast(uint2, vl);
}

template <int NP>
__global__ void __launch_bounds__(256, 3) wgrad_up2_phase_bf3_kernel(const PWArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[NP * (PXPLANE + PZPLANE)];
  unsigned char* const Xs = lds;
  unsigned char* const Zs = lds + NP * PXPLANE;
  int kx_ = 0, kz_ = 0;
  if (NP == 2) {
    unsigned mx, mz, unused;
    fp_amax3_reduce(fp_amax3_issue(a.amax_low, a.amax_dz, nullptr), mx, mz, unused);      // one round trip for both slots (fp_common.h)
    kx_ = fp_hp_exponent(mx, FP_HP_TARGET_ACT);
    kz_ = fp_hp_exponent(mz, FP_HP_TARGET_ACT);
  }
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, idx = lane & 31, h = lane >> 5;
  // XCD-contiguous logical ids: the ci x co tile workgroups of one pixel split stage the same X / dZ chunks and then share that XCD's
  // L2 (consecutive hardware ids go to different XCDs; PMC showed 3x the fused-minimum HBM bytes per launch); split s walks chunks
  // s, s + S, ... so that neighbouring workgroups of an XCD work on neighbouring chunks at the same time
  int b = fp_xcd_remap(blockIdx.x, gridDim.x);
  const int cot = b % a.cotiles; b /= a.cotiles;
  const int cit = b % a.citiles; b /= a.citiles;
  const int s = b;
  const int ci0 = cit * 32, co0 = cot * 32;
  const int c_begin = s, c_end = a.nchunks, c_step = a.S;
  const int H2 = 2 * a.h, W2 = 2 * a.w;
  const int dy = wave >> 1, dx = wave & 1;
  const bool dxb = dx != 0;

  // staging items: X: (halo row, column group of 4, channel quad) for t < 160;  dZ: (phase = wave, row, column group, channel quad)
  const int q = t & 7;
  const int xcg = (t >> 3) % 5, xhr = t / 40;
  const int zcg = (t >> 3) & 3, zr = (t >> 5) & 1;
  const bool xitem = t < 160;
  float4 xr[4], zv[4];
  unsigned zmask = 0;                 // bit j: position j of the group lies inside the image (else its dZ is stored as zero)
  const bool want_bias = a.bpart != nullptr && cit == 0;     // the four phases together stage every dZ pixel exactly once
  float bs[4] = {0.f, 0.f, 0.f, 0.f};

  auto issue = [&](int c) {
    const int cx = c % a.chunksX;
    const int r = c / a.chunksX;
    const int cy = r % a.chunksY, n = r / a.chunksY;
    const int y0 = cy * CHL, x0 = cx * CW;
    {
      const int sy = min(max(y0 + xhr - 1, 0), a.h - 1);                          // replicate padding (threads >= 160 load too, unused)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int sx = min(max(x0 + xcg * 4 + j - 1, 0), a.w - 1);
        xr[j] = *reinterpret_cast<const float4*>(a.low + ((size_t)(n * a.h + sy) * a.w + sx) * a.C0 + ci0 + q * 4);
      }
    }
    zmask = 0;
    const int ly = y0 + zr;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int lx = x0 + zcg * 4 + j;
      const bool ok = ly < a.h && lx < a.w;
      const int hy = 2 * min(ly, a.h - 1) + dy, hx = 2 * min(lx, a.w - 1) + dx;
      zv[j] = *reinterpret_cast<const float4*>(a.dz + ((size_t)(n * H2 + hy) * W2 + hx) * a.Nout + co0 + q * 4);
      zmask |= ok ? (1u << j) : 0u;
    }
  };
  auto stage = [&]() {
    if (xitem) {
      unsigned char* p = Xs + (xhr * 32 + q * 4) * PXROW + xcg * 8;
      psplit_store<NP>(p, PXPLANE, pf32x4{xr[0].x, xr[1].x, xr[2].x, xr[3].x}, kx_);
      psplit_store<NP>(p + PXROW, PXPLANE, pf32x4{xr[0].y, xr[1].y, xr[2].y, xr[3].y}, kx_);
      psplit_store<NP>(p + 2 * PXROW, PXPLANE, pf32x4{xr[0].z, xr[1].z, xr[2].z, xr[3].z}, kx_);
      psplit_store<NP>(p + 3 * PXROW, PXPLANE, pf32x4{xr[0].w, xr[1].w, xr[2].w, xr[3].w}, kx_);
    }
    {
      unsigned char* p = Zs + ((wave * CHL + zr) * 32 + q * 4) * PZROW + zcg * 8;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (!(zmask & (1u << j))) zv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (want_bias) {
        bs[0] += (zv[0].x + zv[1].x) + (zv[2].x + zv[3].x); bs[1] += (zv[0].y + zv[1].y) + (zv[2].y + zv[3].y);
        bs[2] += (zv[0].z + zv[1].z) + (zv[2].z + zv[3].z); bs[3] += (zv[0].w + zv[1].w) + (zv[2].w + zv[3].w);
      }
      psplit_store<NP>(p, PZPLANE, pf32x4{zv[0].x, zv[1].x, zv[2].x, zv[3].x}, kz_);
      psplit_store<NP>(p + PZROW, PZPLANE, pf32x4{zv[0].y, zv[1].y, zv[2].y, zv[3].y}, kz_);
      psplit_store<NP>(p + 2 * PZROW, PZPLANE, pf32x4{zv[0].z, zv[1].z, zv[2].z, zv[3].z}, kz_);
      psplit_store<NP>(p + 3 * PZROW, PZPLANE, pf32x4{zv[0].w, zv[1].w, zv[2].w, zv[3].w}, kz_);
    }
  };

  f32x16 acc[4];
#pragma unroll
  for (int tp = 0; tp < 4; ++tp)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[tp][r] = 0.f;

  if (c_begin < c_end) {
    issue(c_begin);
    stage();
  }
  __syncthreads();
  for (int c = c_begin; c < c_end; c += c_step) {
    if (c + c_step < c_end) issue(c + c_step);       // next chunk's global loads fly under this chunk's MFMAs
#pragma unroll
    for (int r = 0; r < CHL; ++r) {
      uint4 bz[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) bz[p] = *reinterpret_cast<const uint4*>(Zs + p * PZPLANE + ((wave * CHL + r) * 32 + idx) * PZROW + h * 16);
#pragma unroll
      for (int ta = 0; ta < 2; ++ta) {
        uint4 a0[NP], a1[NP];           // taps b = 0, 1: halo columns shifted by dx, dx + 1
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const unsigned char* row = Xs + p * PXPLANE + ((r + dy + ta) * 32 + idx) * PXROW + h * 16;
          const uint4 d = *reinterpret_cast<const uint4*>(row);                  // columns 8h .. 8h+7
          const unsigned e = *reinterpret_cast<const unsigned*>(row + 16);       // columns 8h+8, 8h+9
          const uint4 s1 = make_uint4(__builtin_amdgcn_alignbit(d.y, d.x, 16), __builtin_amdgcn_alignbit(d.z, d.y, 16),
                                      __builtin_amdgcn_alignbit(d.w, d.z, 16), __builtin_amdgcn_alignbit(e, d.w, 16));
          const uint4 s2 = make_uint4(d.y, d.z, d.w, e);
          // wave-uniform choice, one v_cndmask per 32-bit word (a ?: on whole vectors was compiled to a scratch-memory indexed select; the
          // and / or form of round 3 cost two instructions per word; two straight-line bodies behind a uniform branch spilled 64 registers)
          a0[p] = make_uint4(dxb ? s1.x : d.x, dxb ? s1.y : d.y, dxb ? s1.z : d.z, dxb ? s1.w : d.w);
          a1[p] = make_uint4(dxb ? s2.x : s1.x, dxb ? s2.y : s1.y, dxb ? s2.z : s1.z, dxb ? s2.w : s1.w);
        }
        constexpr int NPROD = NP == 3 ? 6 : 4;       // smallest products first
        constexpr int PA[6] = {NP == 3 ? 2 : 1, NP == 3 ? 0 : 1, NP == 3 ? 1 : 0, NP == 3 ? 1 : 0, 0, 0};
        constexpr int PB[6] = {NP == 3 ? 0 : 1, NP == 3 ? 2 : 0, 1, 0, 1, 0};
#pragma unroll
        for (int qq = (NP == 2 ? 4 - FP_HP_PRODUCTS : 0); qq < NPROD; ++qq) {
          if (NP == 2) {
            const pf16x8 bb = __builtin_bit_cast(pf16x8, bz[PB[qq]]);
            acc[ta * 2 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(pf16x8, a0[PA[qq]]), bb, acc[ta * 2 + 0], 0, 0, 0);
            acc[ta * 2 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(pf16x8, a1[PA[qq]]), bb, acc[ta * 2 + 1], 0, 0, 0);
          } else {
            const pbf16x8 bb = __builtin_bit_cast(pbf16x8, bz[PB[qq]]);
            acc[ta * 2 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(pbf16x8, a0[PA[qq]]), bb, acc[ta * 2 + 0], 0, 0, 0);
            acc[ta * 2 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(pbf16x8, a1[PA[qq]]), bb, acc[ta * 2 + 1], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();                                  // every wave has read this chunk
    if (c + c_step < c_end) stage();
    __syncthreads();                                  // next chunk visible
  }
  float* out = a.part + ((size_t)s * 16 + wave * 4) * a.C0 * a.Nout;
#pragma unroll
  for (int tp = 0; tp < 4; ++tp)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ci = ci0 + (r & 3) + 8 * (r >> 2) + 4 * h;
      out[((size_t)tp * a.C0 + ci) * a.Nout + co0 + idx] = NP == 2 ? ldexpf(acc[tp][r], -(kx_ + kz_)) : acc[tp][r];
    }
  if (want_bias) {                                   // 32 staging threads per channel quad -> one partial per output channel
    float* red = reinterpret_cast<float*>(lds);      // the main loop ended with a barrier: the planes are dead
#pragma unroll
    for (int k = 0; k < 4; ++k) red[(t >> 3) * 32 + q * 4 + k] = bs[k];
    __syncthreads();
    if (t < 32) {
      float v = 0.f;
#pragma unroll
      for (int g = 0; g < 32; ++g) v += red[g * 32 + t];
      a.bpart[(size_t)s * a.Nout + co0 + t] = v;
    }
  }
}

// db[n] (+)= sum_s bpart[s][n], fixed order
__global__ void __launch_bounds__(256) up2_wgrad_bias_reduce_kernel(const float* __restrict__ bpart, int S, int Nout, float* __restrict__ db,
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
  static const int target = getenv("FP_PWGRAD_TARGET_WGS") ? atoi(getenv("FP_PWGRAD_TARGET_WGS")) : 768;
  int64_t S = fp_ceil_div(target, base);   // three workgroups per CU are resident (43-51 KB of LDS): one full round
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
  return ((int64_t)p.S * 16 * C0 * Nout + (int64_t)p.S * Nout) * (int64_t)sizeof(float);
}

static int phase_wgrad_launch(bool bf3, const float* low, const float* dz, float* dw_oihw, float* db, int32_t N, int32_t h, int32_t w, int32_t C0,
                              int32_t Nout, int32_t kc_total, int32_t k_begin, int accumulate, void* workspace,
                              int64_t workspace_bytes, fp_stream_t stream_, const uint32_t* amax_low = nullptr,
                              const uint32_t* amax_dz = nullptr) {
  hipStream_t stream = (hipStream_t)stream_;
  FP_REQUIRE(low && dz && dw_oihw && workspace, "fp_conv_up2_phase_wgrad: null pointer");
  FP_REQUIRE(eligible(N, h, w, C0, Nout), "fp_conv_up2_phase_wgrad: shape not supported (see fp_conv_up2_phase_wgrad_workspace)");
  FP_REQUIRE(k_begin >= 0 && k_begin + C0 <= kc_total, "fp_conv_up2_phase_wgrad: input-channel slice out of range");
  const PPlan p = plan(N, h, w, C0, Nout);
  FP_REQUIRE(workspace_bytes >= fp_conv_up2_phase_wgrad_workspace(N, h, w, C0, Nout), "fp_conv_up2_phase_wgrad: workspace too small");
  PWArgs a;
  a.amax_low = amax_low; a.amax_dz = amax_dz;
  a.low = low; a.dz = dz; a.part = (float*)workspace;
  a.bpart = db ? (float*)workspace + (size_t)p.S * 16 * C0 * Nout : nullptr;
  a.N = N; a.h = h; a.w = w; a.C0 = C0; a.Nout = Nout;
  a.chunksY = p.cy; a.chunksX = p.cx; a.nchunks = p.nchunks; a.chunksPerSplit = p.chunksPerSplit; a.S = p.S;
  a.citiles = p.citiles; a.cotiles = p.cotiles;
  if (bf3 && amax_low) fp_launch(wgrad_up2_phase_bf3_kernel<2>, dim3(p.S * p.citiles * p.cotiles), dim3(256), 0, stream, a);
  else if (bf3) fp_launch(wgrad_up2_phase_bf3_kernel<3>, dim3(p.S * p.citiles * p.cotiles), dim3(256), 0, stream, a);
  else fp_launch(wgrad_up2_phase_kernel, dim3(p.S * p.citiles * p.cotiles), dim3(256), 0, stream, a);
  int rc = fp_check_launch("fp_conv_up2_phase_wgrad");
  if (rc) return rc;
  if (db) {
    fp_launch(up2_wgrad_bias_reduce_kernel, dim3((Nout + 255) / 256), dim3(256), 0, stream, (const float*)a.bpart, p.S, Nout, db,
                       accumulate);
    rc = fp_check_launch("fp_conv_up2_phase_wgrad(bias)");
    if (rc) return rc;
  }
  const size_t tot16 = (size_t)16 * C0 * Nout;
  if (p.S > 1) {
    fp_launch(up2_wgrad_sum_kernel, dim3((unsigned)fp_ceil_div((int64_t)tot16, 64)), dim3(256), 0, stream, (float*)workspace, p.S, tot16);
    rc = fp_check_launch("fp_conv_up2_phase_wgrad(sum)");
    if (rc) return rc;
  }
  int rgrid = (int)fp_ceil_div((int64_t)9 * C0 * Nout, 256);
  if (rgrid > 4096) rgrid = 4096;
  fp_launch(up2_wgrad_uncollapse_kernel, dim3(rgrid), dim3(256), 0, stream, (const float*)workspace, dw_oihw, C0, Nout, kc_total,
                     k_begin, accumulate);
  return fp_check_launch("fp_conv_up2_phase_wgrad(uncollapse)");
}

extern "C" int fp_conv_up2_phase_wgrad(const float* low, const float* dz, float* dw_oihw, int32_t N, int32_t h, int32_t w, int32_t C0,
                                       int32_t Nout, int32_t kc_total, int32_t k_begin, int accumulate, void* workspace,
                                       int64_t workspace_bytes, fp_stream_t stream) {
  return phase_wgrad_launch(false, low, dz, dw_oihw, nullptr, N, h, w, C0, Nout, kc_total, k_begin, accumulate, workspace, workspace_bytes, stream);
}

// same contract, operands split exactly into three bf16 terms (six bf16 MFMA products, fp32 accumulate); db (optional, [Nout]) receives
// the bias gradient = column sums of dz from the same pass
extern "C" int fp_conv_up2_phase_wgrad_bf3(const float* low, const float* dz, float* dw_oihw, float* db, int32_t N, int32_t h, int32_t w,
                                           int32_t C0, int32_t Nout, int32_t kc_total, int32_t k_begin, int accumulate, void* workspace,
                                           int64_t workspace_bytes, fp_stream_t stream) {
  return phase_wgrad_launch(true, low, dz, dw_oihw, db, N, h, w, C0, Nout, kc_total, k_begin, accumulate, workspace, workspace_bytes, stream);
}

// fp16-pair operands (fp_conv3x3_hp): `amax_low` / `amax_dz` = amax slots of the two tensors
extern "C" int fp_conv_up2_phase_wgrad_hp(const float* low, const float* dz, float* dw_oihw, float* db, int32_t N, int32_t h, int32_t w,
                                          int32_t C0, int32_t Nout, int32_t kc_total, int32_t k_begin, int accumulate, void* workspace,
                                          int64_t workspace_bytes, const uint32_t* amax_low, const uint32_t* amax_dz, fp_stream_t stream) {
  FP_REQUIRE(amax_low && amax_dz, "fp_conv_up2_phase_wgrad_hp: amax slots missing");
  return phase_wgrad_launch(true, low, dz, dw_oihw, db, N, h, w, C0, Nout, kc_total, k_begin, accumulate, workspace, workspace_bytes, stream,
                            amax_low, amax_dz);
}
