// 3x3 stride-1 weight gradient with exactly split bf16x3 operands (gfx950).
//
//   dW[tap][ci][co] = sum over pixels p of  X[p + tap][ci] * dZ[p][co]
//
// The contraction runs over PIXELS, so both MFMA operands want "8 consecutive pixels of one channel" per lane while the
// tensors are NHWC.  The transposition happens once, on the way into LDS, together with the exact three-way bf16 split
// (x = h + m + l, see conv3x3_tile_bf3.hip): a thread takes 4 consecutive pixels x 4 channels (four float4 loads), regroups
// the registers per channel (free), converts and writes one 8-byte [4 pixels] group per channel and plane:
//     Xs[plane][halo row 6][ci 32][20 cols]   Zs[plane][row 4][co 32][16 cols]   (48-byte rows: conflict-free ds_read_b128)
// A workgroup owns a (32 ci x 32 co) block of all nine taps and walks 4 x 16 pixel chunks; wave w takes chunk row w: ONE
// 16-pixel k-step per tap, six v_mfma_f32_32x32x16_bf16 each (54 MFMAs x 32 cycles per wave per chunk instead of 72 x 64 of the
// fp32 kernel, wgrad3x3_tile.hip).  The kx = 1, 2 taps are the kx = 0 operand shifted by one / two bf16: a funnel shift
// (v_alignbit) of the aligned 16-byte read plus the next dword.  Global loads of the next chunk fly under the MFMAs of the
// current one (register prefetch, single LDS buffer, two barriers per chunk).  Sums across waves / splits: fixed order.
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
};

constexpr int CH = 4, CW = 16, HR = CH + 2;
constexpr int XROW = 48, ZROW = 48;                          // bytes per (row, channel) line: 20 / 16 bf16 + pad
constexpr int XPLANE = HR * 32 * XROW, ZPLANE = CH * 32 * ZROW;
constexpr int XBYTES = 3 * XPLANE, ZBYTES = 3 * ZPLANE;      // 27648 + 18432 = 46080

__device__ __forceinline__ void split_store(unsigned char* p, int plane_stride, const f32x4 v) {
  const bf16x4 vh = __builtin_convertvector(v, bf16x4);
  const f32x4 r1 = v - __builtin_convertvector(vh, f32x4);
  const bf16x4 vm = __builtin_convertvector(r1, bf16x4);
  const f32x4 r2 = r1 - __builtin_convertvector(vm, f32x4);
  const bf16x4 vl = __builtin_convertvector(r2, bf16x4);
  *reinterpret_cast<uint2*>(p) = __builtin_bit_cast(uint2, vh);
  *reinterpret_cast<uint2*>(p + plane_stride) = __builtin_bit_cast(uint2, vm);
  *reinterpret_cast<uint2*>(p + 2 * plane_stride) = __builtin_bit_cast(uint2, vl);
}

__global__ void __launch_bounds__(256) wgrad3x3_bf3_kernel(const W3Args a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[XBYTES + ZBYTES];
  unsigned char* const Xs = lds;
  unsigned char* const Zs = lds + XBYTES;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, idx = lane & 31, h = lane >> 5;
  int b = blockIdx.x;
  const int cot = b % a.cotiles; b /= a.cotiles;
  const int cit = b % a.citiles; b /= a.citiles;
  const int s = b;
  const int ci0 = cit * 32, co0 = cot * 32;
  const int c_begin = s * a.chunksPerSplit;
  const int c_end = min(a.nchunks, c_begin + a.chunksPerSplit);

  // staging items: X: (halo row hr, column group cg of 4, channel quad q) for t < 240;  dZ: (row, cg, q) for t < 128
  const int q = t & 7;
  const int xcg = (t >> 3) % 5, xhr = t / 40;
  const int zcg = (t >> 3) & 3, zr = t >> 5;
  const bool xitem = t < 240, zitem = t < 128;
  float4 xr[4], zv[4];
  unsigned xmask = 0, zmask = 0;      // bit j: pixel j of the group is real data (else zero)
  const bool want_bias = a.bpart != nullptr && cit == 0;     // dZ passes through this workgroup's registers exactly once
  float bs[4] = {0.f, 0.f, 0.f, 0.f};

  auto issue = [&](int c) {
    const int cx = c % a.chunksX;
    const int r = c / a.chunksX;
    const int cy = r % a.chunksY, n = r / a.chunksY;
    const int y0 = cy * CH, x0 = cx * CW;
    xmask = zmask = 0;
    {
      int sy = y0 + xhr - 1;
      bool rowok = xitem;
      if (a.mode == 0) rowok = rowok && sy >= 0 && sy < a.H;
      else { rowok = rowok && sy >= -1 && sy <= a.H; sy = fp_reflect(sy, a.H); }
      sy = min(max(sy, 0), a.H - 1);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int hx = xcg * 4 + j;                    // halo column 0..19 (18, 19 are padding)
        int sx = x0 + hx - 1;
        bool ok = rowok && hx < CW + 2;
        if (a.mode == 0) ok = ok && sx >= 0 && sx < a.W;
        else { ok = ok && sx >= -1 && sx <= a.W; sx = fp_reflect(sx, a.W); }
        sx = min(max(sx, 0), a.W - 1);
        const size_t xpix = a.mode == 2 ? (size_t)(n * (a.H >> 1) + (sy >> 1)) * (a.W >> 1) + (sx >> 1) : (size_t)(n * a.H + sy) * a.W + sx;
        xr[j] = *reinterpret_cast<const float4*>(a.x + xpix * a.C + ci0 + q * 4);
        xmask |= ok ? (1u << j) : 0u;
      }
    }
    {
      const int oy = min(y0 + zr, a.H - 1);
      const bool rowok = zitem && y0 + zr < a.H;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ox = x0 + zcg * 4 + j;
        const bool ok = rowok && ox < a.W;
        zv[j] = *reinterpret_cast<const float4*>(a.dz + ((size_t)(n * a.H + oy) * a.W + min(ox, a.W - 1)) * a.Nout + co0 + q * 4);
        zmask |= ok ? (1u << j) : 0u;
      }
    }
  };
  auto stage = [&]() {
    if (xitem) {
      unsigned char* p = Xs + (xhr * 32 + q * 4) * XROW + xcg * 8;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (!(xmask & (1u << j))) xr[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      split_store(p, XPLANE, f32x4{xr[0].x, xr[1].x, xr[2].x, xr[3].x});
      split_store(p + XROW, XPLANE, f32x4{xr[0].y, xr[1].y, xr[2].y, xr[3].y});
      split_store(p + 2 * XROW, XPLANE, f32x4{xr[0].z, xr[1].z, xr[2].z, xr[3].z});
      split_store(p + 3 * XROW, XPLANE, f32x4{xr[0].w, xr[1].w, xr[2].w, xr[3].w});
    }
    if (zitem) {
      unsigned char* p = Zs + (zr * 32 + q * 4) * ZROW + zcg * 8;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (!(zmask & (1u << j))) zv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (want_bias) {
        bs[0] += (zv[0].x + zv[1].x) + (zv[2].x + zv[3].x); bs[1] += (zv[0].y + zv[1].y) + (zv[2].y + zv[3].y);
        bs[2] += (zv[0].z + zv[1].z) + (zv[2].z + zv[3].z); bs[3] += (zv[0].w + zv[1].w) + (zv[2].w + zv[3].w);
      }
      split_store(p, ZPLANE, f32x4{zv[0].x, zv[1].x, zv[2].x, zv[3].x});
      split_store(p + ZROW, ZPLANE, f32x4{zv[0].y, zv[1].y, zv[2].y, zv[3].y});
      split_store(p + 2 * ZROW, ZPLANE, f32x4{zv[0].z, zv[1].z, zv[2].z, zv[3].z});
      split_store(p + 3 * ZROW, ZPLANE, f32x4{zv[0].w, zv[1].w, zv[2].w, zv[3].w});
    }
  };

  f32x16 acc[9];
#pragma unroll
  for (int tp = 0; tp < 9; ++tp)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[tp][r] = 0.f;

  if (c_begin < c_end) {
    issue(c_begin);
    stage();
  }
  __syncthreads();
  for (int c = c_begin; c < c_end; ++c) {
    if (c + 1 < c_end) issue(c + 1);                 // next chunk's global loads fly under this chunk's MFMAs
    // B fragments: dZ row `wave`, lane (co = idx, pixel group h)
    uint4 bz[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) bz[p] = *reinterpret_cast<const uint4*>(Zs + p * ZPLANE + (wave * 32 + idx) * ZROW + h * 16);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      uint4 a0[3], a1[3], a2[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const unsigned char* row = Xs + p * XPLANE + ((wave + ky) * 32 + idx) * XROW + h * 16;
        const uint4 d = *reinterpret_cast<const uint4*>(row);                  // columns 8h .. 8h+7
        const unsigned e = *reinterpret_cast<const unsigned*>(row + 16);       // columns 8h+8, 8h+9
        a0[p] = d;
        a1[p] = make_uint4(__builtin_amdgcn_alignbit(d.y, d.x, 16), __builtin_amdgcn_alignbit(d.z, d.y, 16),
                           __builtin_amdgcn_alignbit(d.w, d.z, 16), __builtin_amdgcn_alignbit(e, d.w, 16));   // shifted by one column
        a2[p] = make_uint4(d.y, d.z, d.w, e);                                                                  // by two
      }
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int qq = 0; qq < 6; ++qq) {
        const bf16x8 bb = __builtin_bit_cast(bf16x8, bz[PB[qq]]);
        acc[ky * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0[PA[qq]]), bb, acc[ky * 3 + 0], 0, 0, 0);
        acc[ky * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1[PA[qq]]), bb, acc[ky * 3 + 1], 0, 0, 0);
        acc[ky * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a2[PA[qq]]), bb, acc[ky * 3 + 2], 0, 0, 0);
      }
    }
    __syncthreads();                                  // every wave has read this chunk
    if (c + 1 < c_end) stage();
    __syncthreads();                                  // next chunk visible
  }

  // ---- sum the four waves' tiles through LDS (fixed order), one tap at a time ----------------------------------
  float* red = reinterpret_cast<float*>(lds);        // [4 waves][16 regs][64 lanes] = 16 KB
  float* out = a.part + (size_t)s * 9 * a.C * a.Nout;
  if (want_bias) {                                   // 16 staging threads per channel quad -> one partial per output channel
    if (zitem) {
#pragma unroll
      for (int k = 0; k < 4; ++k) red[(t >> 3) * 32 + q * 4 + k] = bs[k];
    }
    __syncthreads();
    if (t < 32) {
      float v = 0.f;
#pragma unroll
      for (int g = 0; g < 16; ++g) v += red[g * 32 + t];
      a.bpart[(size_t)s * a.Nout + co0 + t] = v;
    }
    __syncthreads();
  }
#pragma unroll
  for (int tp = 0; tp < 9; ++tp) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[tp][r];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int e = t + 256 * k;
      const float v = ((red[e] + red[1024 + e]) + red[2048 + e]) + red[3072 + e];
      const int r = e >> 6, ln = e & 63;
      const int ci = ci0 + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5);
      out[((size_t)tp * a.C + ci) * a.Nout + co0 + (ln & 31)] = v;
    }
    __syncthreads();
  }
}

// ---- second generation: fp32 transposed in LDS, exact split at READ time, double-buffered, ONE barrier per chunk -----------
// The first kernel splits while staging (three bf16 planes, 46 KB per workgroup: no room for a second buffer beside a second
// resident workgroup) and needs two barriers per 64-pixel chunk; its staging writes (ds_write_b64 at a 192-byte lane stride) are
// 4-way bank-conflicted and every operand row is re-read as three planes.  Here the staging is a pure transpose
//     Xs[buf][halo row 6][ci 32][20 px fp32]   Zs[buf][row 4][co 32][16 px fp32 (+4 pad)]       (80-byte lines = 5 x 16-byte slots:
// lane (channel c, pixel half h) -> slot 5c + 2h, a bijection mod 16 for any 16 distinct c: conflict-free ds_read_b128 AND, with
// the staging map (channel quad q, pixel group cg) -> slot 5k + 4q + cg, conflict-free ds_write_b128), 25 KB per buffer, so two
// buffers fit twice per CU; a chunk is: issue the next chunk's global loads -> read this chunk's rows (2 x b128 + b64 per row),
// split each fp32 into its three bf16 terms in registers (same round-to-nearest chain as split_store: bit-identical planes, so
// bit-identical sums) -> 54 MFMAs -> write the next chunk into the other buffer -> one barrier.
constexpr int LP2 = 80;
constexpr int XB2 = HR * 32 * LP2, ZB2 = CH * 32 * LP2, BUF2 = XB2 + ZB2;      // 15360 + 10240 = 25600 bytes per buffer
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// two fp32 -> packed bf16 pairs of the three exact terms
__device__ __forceinline__ void split_pair(const float x, const float y, unsigned& hi, unsigned& mid, unsigned& lo) {
  const f32x2 v = {x, y};
  const bf16x2 vh = __builtin_convertvector(v, bf16x2);
  const f32x2 r1 = v - __builtin_convertvector(vh, f32x2);
  const bf16x2 vm = __builtin_convertvector(r1, bf16x2);
  const f32x2 r2 = r1 - __builtin_convertvector(vm, f32x2);
  const bf16x2 vl = __builtin_convertvector(r2, bf16x2);
  hi = __builtin_bit_cast(unsigned, vh);
  mid = __builtin_bit_cast(unsigned, vm);
  lo = __builtin_bit_cast(unsigned, vl);
}

__global__ void __launch_bounds__(256, 2) wgrad3x3_bf3_v2_kernel(const W3Args a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BUF2];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, idx = lane & 31, h = lane >> 5;
  int b = blockIdx.x;
  const int cot = b % a.cotiles; b /= a.cotiles;
  const int cit = b % a.citiles; b /= a.citiles;
  const int s = b;
  const int ci0 = cit * 32, co0 = cot * 32;
  const int c_begin = s * a.chunksPerSplit;
  const int c_end = min(a.nchunks, c_begin + a.chunksPerSplit);

  const int q = t & 7;
  const int xcg = (t >> 3) % 5, xhr = t / 40;
  const int zcg = (t >> 3) & 3, zr = t >> 5;
  const bool xitem = t < 240, zitem = t < 128;
  float4 xr[4], zv[4];
  unsigned xmask = 0, zmask = 0;
  const bool want_bias = a.bpart != nullptr && cit == 0;
  float bs[4] = {0.f, 0.f, 0.f, 0.f};

  auto issue = [&](int c) {
    const int cx = c % a.chunksX;
    const int r = c / a.chunksX;
    const int cy = r % a.chunksY, n = r / a.chunksY;
    const int y0 = cy * CH, x0 = cx * CW;
    xmask = zmask = 0;
    {
      int sy = y0 + xhr - 1;
      bool rowok = xitem;
      if (a.mode == 0) rowok = rowok && sy >= 0 && sy < a.H;
      else { rowok = rowok && sy >= -1 && sy <= a.H; sy = fp_reflect(sy, a.H); }
      sy = min(max(sy, 0), a.H - 1);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int hx = xcg * 4 + j;
        int sx = x0 + hx - 1;
        bool ok = rowok && hx < CW + 2;
        if (a.mode == 0) ok = ok && sx >= 0 && sx < a.W;
        else { ok = ok && sx >= -1 && sx <= a.W; sx = fp_reflect(sx, a.W); }
        sx = min(max(sx, 0), a.W - 1);
        const size_t xpix = a.mode == 2 ? (size_t)(n * (a.H >> 1) + (sy >> 1)) * (a.W >> 1) + (sx >> 1) : (size_t)(n * a.H + sy) * a.W + sx;
        xr[j] = *reinterpret_cast<const float4*>(a.x + xpix * a.C + ci0 + q * 4);
        xmask |= ok ? (1u << j) : 0u;
      }
    }
    {
      const int oy = min(y0 + zr, a.H - 1);
      const bool rowok = zitem && y0 + zr < a.H;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ox = x0 + zcg * 4 + j;
        const bool ok = rowok && ox < a.W;
        zv[j] = *reinterpret_cast<const float4*>(a.dz + ((size_t)(n * a.H + oy) * a.W + min(ox, a.W - 1)) * a.Nout + co0 + q * 4);
        zmask |= ok ? (1u << j) : 0u;
      }
    }
  };
  auto stage = [&](int buf) {
    unsigned char* const base = lds + buf * BUF2;
    if (xitem) {
      unsigned char* p = base + (xhr * 32 + q * 4) * LP2 + xcg * 16;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (!(xmask & (1u << j))) xr[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(p) = make_float4(xr[0].x, xr[1].x, xr[2].x, xr[3].x);
      *reinterpret_cast<float4*>(p + LP2) = make_float4(xr[0].y, xr[1].y, xr[2].y, xr[3].y);
      *reinterpret_cast<float4*>(p + 2 * LP2) = make_float4(xr[0].z, xr[1].z, xr[2].z, xr[3].z);
      *reinterpret_cast<float4*>(p + 3 * LP2) = make_float4(xr[0].w, xr[1].w, xr[2].w, xr[3].w);
    }
    if (zitem) {
      unsigned char* p = base + XB2 + (zr * 32 + q * 4) * LP2 + zcg * 16;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (!(zmask & (1u << j))) zv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (want_bias) {
        bs[0] += (zv[0].x + zv[1].x) + (zv[2].x + zv[3].x); bs[1] += (zv[0].y + zv[1].y) + (zv[2].y + zv[3].y);
        bs[2] += (zv[0].z + zv[1].z) + (zv[2].z + zv[3].z); bs[3] += (zv[0].w + zv[1].w) + (zv[2].w + zv[3].w);
      }
      *reinterpret_cast<float4*>(p) = make_float4(zv[0].x, zv[1].x, zv[2].x, zv[3].x);
      *reinterpret_cast<float4*>(p + LP2) = make_float4(zv[0].y, zv[1].y, zv[2].y, zv[3].y);
      *reinterpret_cast<float4*>(p + 2 * LP2) = make_float4(zv[0].z, zv[1].z, zv[2].z, zv[3].z);
      *reinterpret_cast<float4*>(p + 3 * LP2) = make_float4(zv[0].w, zv[1].w, zv[2].w, zv[3].w);
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
  const int zoff = XB2 + (wave * 32 + idx) * LP2 + h * 32;
  const int xoff = (wave * 32 + idx) * LP2 + h * 32;
  for (int c = c_begin; c < c_end; ++c) {
    const unsigned char* const Bb = lds + ((c - c_begin) & 1) * BUF2;
    if (c + 1 < c_end) issue(c + 1);                 // next chunk's global loads fly under this chunk's conversions and MFMAs
    // B fragments: dZ row `wave`, lane (co = idx, pixel group h): 8 fp32 -> three planes of 8 bf16
    uint4 bz[3];
    {
      const float4 z0 = *reinterpret_cast<const float4*>(Bb + zoff), z1 = *reinterpret_cast<const float4*>(Bb + zoff + 16);
      split_pair(z0.x, z0.y, bz[0].x, bz[1].x, bz[2].x);
      split_pair(z0.z, z0.w, bz[0].y, bz[1].y, bz[2].y);
      split_pair(z1.x, z1.y, bz[0].z, bz[1].z, bz[2].z);
      split_pair(z1.z, z1.w, bz[0].w, bz[1].w, bz[2].w);
    }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const unsigned char* row = Bb + xoff + ky * 32 * LP2;
      const float4 x0 = *reinterpret_cast<const float4*>(row), x1 = *reinterpret_cast<const float4*>(row + 16);
      const float2 x2 = *reinterpret_cast<const float2*>(row + 32);                   // columns 8h+8, 8h+9
      unsigned P[3][5];
      split_pair(x0.x, x0.y, P[0][0], P[1][0], P[2][0]);
      split_pair(x0.z, x0.w, P[0][1], P[1][1], P[2][1]);
      split_pair(x1.x, x1.y, P[0][2], P[1][2], P[2][2]);
      split_pair(x1.z, x1.w, P[0][3], P[1][3], P[2][3]);
      split_pair(x2.x, x2.y, P[0][4], P[1][4], P[2][4]);
      uint4 a0[3], a1[3], a2[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        a0[p] = make_uint4(P[p][0], P[p][1], P[p][2], P[p][3]);
        a1[p] = make_uint4(__builtin_amdgcn_alignbit(P[p][1], P[p][0], 16), __builtin_amdgcn_alignbit(P[p][2], P[p][1], 16),
                           __builtin_amdgcn_alignbit(P[p][3], P[p][2], 16), __builtin_amdgcn_alignbit(P[p][4], P[p][3], 16));
        a2[p] = make_uint4(P[p][1], P[p][2], P[p][3], P[p][4]);
      }
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int qq = 0; qq < 6; ++qq) {
        const bf16x8 bb = __builtin_bit_cast(bf16x8, bz[PB[qq]]);
        acc[ky * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0[PA[qq]]), bb, acc[ky * 3 + 0], 0, 0, 0);
        acc[ky * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1[PA[qq]]), bb, acc[ky * 3 + 1], 0, 0, 0);
        acc[ky * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a2[PA[qq]]), bb, acc[ky * 3 + 2], 0, 0, 0);
      }
    }
    // the other buffer was last read during chunk c - 1, and every wave has passed the barrier that ended that chunk
    if (c + 1 < c_end) stage((c + 1 - c_begin) & 1);
    __syncthreads();                                  // next chunk visible; this chunk's buffer free for chunk c + 2
  }

  // ---- sum the four waves' tiles through LDS (fixed order), one tap at a time: same order as the first kernel ---------------------
  float* red = reinterpret_cast<float*>(lds);        // [4 waves][16 regs][64 lanes] = 16 KB
  float* out = a.part + (size_t)s * 9 * a.C * a.Nout;
  if (want_bias) {
    if (zitem) {
#pragma unroll
      for (int k = 0; k < 4; ++k) red[(t >> 3) * 32 + q * 4 + k] = bs[k];
    }
    __syncthreads();
    if (t < 32) {
      float v = 0.f;
#pragma unroll
      for (int g = 0; g < 16; ++g) v += red[g * 32 + t];
      a.bpart[(size_t)s * a.Nout + co0 + t] = v;
    }
    __syncthreads();
  }
#pragma unroll
  for (int tp = 0; tp < 9; ++tp) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[tp][r];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int e = t + 256 * k;
      const float v = ((red[e] + red[1024 + e]) + red[2048 + e]) + red[3072 + e];
      const int r = e >> 6, ln = e & 63;
      const int ci = ci0 + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5);
      out[((size_t)tp * a.C + ci) * a.Nout + co0 + (ln & 31)] = v;
    }
    __syncthreads();
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
  if (cy * CH * cx * CW * 10 > (int64_t)d->OH * d->OW * 13) return false;    // > 30 % padded work
  return (int64_t)d->N * cy * cx >= 16;
}

struct WPlan { int S, chunksPerSplit, nchunks, citiles, cotiles, cy, cx; };
WPlan plan(const fp_conv_desc* d) {
  WPlan p;
  p.cy = (int)fp_ceil_div(d->OH, CH); p.cx = (int)fp_ceil_div(d->OW, CW);
  p.nchunks = d->N * p.cy * p.cx;
  p.citiles = d->C0 / 32; p.cotiles = d->Nout / 32;
  const int64_t base = (int64_t)p.citiles * p.cotiles;
  int64_t S = fp_ceil_div(512, base);      // two workgroups per CU are resident (240 registers): one full round
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

extern "C" int fp_conv_wgrad_bf3(const fp_conv_desc* d, const float* x, const float* dz, float* dw_oihw, float* db, int32_t kc_total,
                                 int32_t k_begin, int accumulate, void* workspace, int64_t workspace_bytes, fp_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  FP_REQUIRE(d && x && dz && dw_oihw && workspace, "fp_conv_wgrad_bf3: null pointer");
  FP_REQUIRE(eligible(d), "fp_conv_wgrad_bf3: shape not supported (see fp_conv_wgrad_bf3_workspace)");
  FP_REQUIRE(k_begin >= 0 && k_begin + d->C0 <= kc_total, "fp_conv_wgrad_bf3: input-channel slice out of range");
  const WPlan p = plan(d);
  FP_REQUIRE(workspace_bytes >= fp_conv_wgrad_bf3_workspace(d), "fp_conv_wgrad_bf3: workspace too small");
  W3Args a;
  a.x = x; a.dz = dz; a.part = (float*)workspace;
  a.bpart = db ? (float*)workspace + (size_t)p.S * 9 * d->C0 * d->Nout : nullptr;
  a.N = d->N; a.H = d->OH; a.W = d->OW; a.C = d->C0; a.Nout = d->Nout;
  a.mode = d->gather == FP_GATHER_FWD_ZERO ? 0 : (d->gather == FP_GATHER_FWD_REFLECT_UP2 ? 2 : 1);
  a.chunksY = p.cy; a.chunksX = p.cx; a.nchunks = p.nchunks; a.chunksPerSplit = p.chunksPerSplit; a.S = p.S;
  a.citiles = p.citiles; a.cotiles = p.cotiles;
  static const bool v1 = fp_env_flag("FP_WGRAD_BF3_V1");       // A/B switch: the first-generation kernel (split while staging)
  if (v1) hipLaunchKernelGGL(wgrad3x3_bf3_kernel, dim3(p.S * p.citiles * p.cotiles), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(wgrad3x3_bf3_v2_kernel, dim3(p.S * p.citiles * p.cotiles), dim3(256), 0, stream, a);
  int rc = fp_check_launch("fp_conv_wgrad_bf3");
  if (rc) return rc;
  if (db) {
    hipLaunchKernelGGL(wgrad_bias_reduce_kernel, dim3((d->Nout + 255) / 256), dim3(256), 0, stream, (const float*)a.bpart, p.S, d->Nout, db,
                       accumulate);
    rc = fp_check_launch("fp_conv_wgrad_bf3(bias)");
    if (rc) return rc;
  }
  return fp_wgrad_reduce_launch((const float*)workspace, dw_oihw, p.S, 9, d->C0, d->Nout, 0, accumulate, kc_total, k_begin, stream);
}
