// Halo-tile 3x3 convolution (forward / data-gradient) with fp32 operands split EXACTLY into three bf16 terms (gfx950).
//
//   x = h + m + l   (h = bf16(x), m = bf16(x - h), l = bf16(x - h - m): 8 + 8 + 8 significant bits, the sum is exact)
//   a*b ~= ah*bh + (ah*bm + am*bh) + (ah*bl + am*bm + al*bh)      -- the dropped terms (am*bl, al*bm, al*bl) are <= 2^-24 |a*b|
//
// Every bf16 x bf16 product is exact in the MFMA's fp32 accumulator, so the result carries LESS error than the fp32 MFMA of
// conv3x3_tile.hip (measured 2.0e-7 vs 5.0e-7 of max|y| on K = 144..4608 dot products) while six v_mfma_f32_32x32x16_bf16
// (32 cycles, K = 16) replace eight v_mfma_f32_32x32x2_f32 (64 cycles, K = 2): 192 instead of 512 MFMA cycles per 32x32x16 block.
// Same tiling and pipeline as conv3x3_tile.hip (8 x 16 output pixels x 64 / 32 channels per workgroup, halo staged once per
// 16-channel chunk, nine taps from LDS, weight slices from L1/L2 two taps ahead, unconditional straight-line loads => exact
// vmcnt waits, one barrier per chunk); differences:
//   * the halo is converted on its way into LDS: three bf16 planes [plane][pixel][16 ch + 8 pad] (48-byte pixel rows: the
//     per-lane 16-byte A fragment "pixel = lane & 31, k-group = lane >> 5" is a conflict-free ds_read_b128);
//   * weights are packed [tap][chunk][plane][n][16] bf16 (pack.hip, FP_PACK_*_BF3): a wave's B fragment of one plane is 1 KB contiguous.
#include <type_traits>

#include "fp_common.h"

int fp_splitk_reduce_launch(const float* part, int SK, int64_t M, int Nout, const float* bias, const float* addend, const float* addend_mask,
                            const float* actsrc, float* y, int act, unsigned epi, hipStream_t stream, unsigned* amax_out = nullptr);
int fp_splitk_reduce_bnb_launch(const float* part, int SK, int64_t M, int Nout, const float* bias, const float* addend, const float* addend_mask,
                                const float* actsrc, float* y, int act, unsigned epi, hipStream_t stream, unsigned* amax_out, const float* z,
                                const float* mean, const float* invstd, float* bpart, int64_t cap_floats, int* rc_out);
int fp_splitk_reduce_stats_launch(const float* part, int SK, int64_t M, int Nout, float* y, hipStream_t stream, unsigned* amax_out, float* stats,
                                  int64_t cap_floats, int* rc_out);

// waves per SIMD requested from the register allocator for the fp16-pair variants: 4 = 128 VGPRs (34 KB of LDS per workgroup allow four
// workgroups per CU; the reflection-fold variants spill 3-4 registers).  Training step 15.16 / 15.24 vs 15.30 / 15.40 ms with 1 (= 160
// VGPRs, three waves), the kernel alone measures the same: a few more waves to cover prologues and epilogues now that a third of the MFMA
// work per wave is left.
#ifndef FP_TILE_HP_WAVES
#define FP_TILE_HP_WAVES 4
#endif
#ifndef FP_BF2_PRODUCTS
#define FP_BF2_PRODUCTS 3
#endif

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

struct Tile3Args {
  const float* src_lo;       // FWD_REFLECT_UP2: the half-resolution tensor whose nearest x2 upsampling forms input channels [0, Clo)
  int Clo;                   // (0 otherwise); `src` then holds the remaining C - Clo channels at full resolution (the skip tensor)
  const float* src;
  const unsigned short* w;   // bf16 [tap][KC16][3][ncols][16]
  const float* bias;
  const float* addend;
  const float* addend_mask;
  const float* actsrc;
  float* y;
  int N, OH, OW, IH, IW, C, Nout, KC16;
  int mode;      // 0 zero padding, 1 reflection padding
  int act;
  unsigned epi;
  int tilesX, tilesY, tilesN, nwg;
  int SK, chunksPerSplit;   // split-K over 16-channel chunks for small grids: raw partials [SK][N*OH*OW][Nout] -> fp_splitk_reduce_launch
  int wmajor;               // workgroup ids enumerate pixel tiles fastest, (channel tile, split) slowest
  float* part;
  // fp16-pair mode (HP): amax slots of the source tensor(s) and of the weights; optional slot receiving max |y| of this launch
  const unsigned* amax_a;
  const unsigned* amax_a1;
  const unsigned* amax_w;
  unsigned* amax_out;
  float* bn_part;     // forward conv in front of a train-mode BatchNorm (unsplit, no epilogue options): Welford partials [pixel tile][Nout][3], or null
  // data gradient whose output is the masked gradient g entering a train-mode BatchNorm's backward (unsplit grids): partial sums
  // [pixel tile][Nout][2] = (sum g, sum g * xhat), xhat = (z - mean) * invstd of THAT BatchNorm -- or null (fp_aux.bnb_*)
  float* bnb_part;
  const float* bnb_z;
  const float* bnb_mean;
  const float* bnb_invstd;
};

struct FoldTap { int wtap, ao, bo, rsel, csel; };
// reflection-fold taps of the data gradient (derivation in conv3x3_tile.hip): weight tap, halo offset (ao, bo), row / column
// selector (0: every pixel, 1: pixels on row / column 1, 2: pixels on row / column n-2)
// (a function, so unrolled loops fold it at compile time)
__device__ __forceinline__ constexpr FoldTap fold_tap3(int e) {
  switch (e) {
    case 0: return {0, 0, 2, 1, 0}; case 1: return {0, 2, 0, 0, 1}; case 2: return {0, 0, 0, 1, 1}; case 3: return {1, 0, 1, 1, 0};
    case 4: return {2, 0, 0, 1, 0}; case 5: return {2, 2, 2, 0, 2}; case 6: return {2, 0, 2, 1, 2}; case 7: return {3, 1, 0, 0, 1};
    case 8: return {5, 1, 2, 0, 2}; case 9: return {6, 2, 2, 2, 0}; case 10: return {6, 0, 0, 0, 1}; case 11: return {6, 2, 0, 2, 1};
    case 12: return {7, 2, 1, 2, 0}; case 13: return {8, 2, 0, 2, 0}; case 14: return {8, 0, 2, 0, 2}; default: return {8, 2, 2, 2, 2};
  }
}

// Round 6 EXPERIMENT, off by default (-DFP_TILE_PIXB32=1 builds it): halo pixels of the 16-wide tiles of 32 bytes (no padding) with the two
// 16-byte k-group halves of a pixel swapped on odd halo rows.  Measured (profiles/round6_notes.md section 5): the fourth workgroup per CU it
// buys the 32-channel forms moves the training step by 0.03 ms (14.66 -> 14.63, inside the noise), and a 128-register budget for the
// 64-channel forms (-DFP_TILE_EXACT64_WAVES=4) costs 0.4 ms -- occupancy is not what these kernels wait for.  Kernel and golden tests pass
// with it (363); it stays a build option because it changes the order in which a tile's BatchNorm partials are merged (last-bit
// differences in the statistics) for no gain.  The idea:  With 48-byte pixels two halo buffers of three planes are 51.8 KB: three workgroups per CU, and the
// counters (profiles/round5_pmc_sq_exact.txt) show the matrix pipe idle 38 % of the time with the three waves of a SIMD all outside their
// taps at once (prologue: three dependent memory latencies; epilogue: statistics + stores).  32-byte pixels are 34.6 KB: FOUR workgroups
// per CU wherever the registers allow it (the 32-channel forms: 99-113 of 128; see fp_tile_min_waves).  Bank conflicts: a 16-lane service
// group of ds_read_b128 ({0-3,12-15,20-27}, {4-11,16-19,28-31}, same k-group) covers 8 pixels of tile row r and 8 of row r + 1; a lane's
// 16-byte slot mod 16 is 4 (py + ky) + 2 (px + kx) + half: each row's 8 pixels take the 8 even residues, so the two rows must differ in
// the half bit -- half = k-group ^ (halo row & 1).  (18-pixel halo pitch = 36 slots = 4 mod 16: an even shift, the parity argument
// holds for every tap.)  The 20-wide tiles keep 48-byte pixels (one workgroup per CU there; their pitch needs another scheme).
#ifndef FP_TILE_PIXB32
#define FP_TILE_PIXB32 0
#endif
template <int TW>
constexpr int fp_tile_pixb() { return (FP_TILE_PIXB32 && TW == 16) ? 32 : 48; }

// MFMA row -> tile pixel.  ds_read_b128 is serviced in four fixed 16-lane groups (lanes {0-3,12-15,20-27}, ...), conflict-free
// when the 16 lanes hit 16 distinct 16-byte slots mod 256 B.  With 48-byte pixels a lane's slot is 3 * halo_pixel mod 16; a
// 32-row block spans two tile rows and the 18-pixel halo pitch shifts the second row by 6 slots, which collides inside the
// groups (SQ_LDS_BANK_CONFLICT was half of all LDS cycles).  Rotating the columns of odd tile rows by two pixels cancels the
// shift: slot = 3 * (row & 15) for every lane.  (16-wide tiles only; the mapping is private to this kernel.)
template <int TW>
__device__ __forceinline__ void fp_tile_pixel(int pt, int& py, int& px) {
  py = pt / TW;
  px = pt - py * TW;
  if (TW == 16 && fp_tile_pixb<TW>() == 48) px = (px - 2 * (py & 1)) & 15;     // (the 32-byte layout swaps halves instead of rotating columns)
}

// NP = number of bf16 terms per operand: 3 = the exact split (six products); 2 = h + m only (three products: ah*bh + ah*bm + am*bh,
// operands rounded to 16 significant bits, ~2^-17 relative) -- an opt-in INFERENCE mode (FP_EPI_BF16X2), never used for training
// HP = the fp16-pair format of fp_common.h (NP = 2 planes, FP_HP_PRODUCTS = three products hh + hm + mh on v_mfma_f32_32x32x16_f16): operands carry
// 22 significant bits after a per-tensor power-of-two scaling; two thirds of the MFMA work and of the LDS / weight traffic of the exact split.
// WPF ("weights per chunk in flight", round 3): the grids of the 6 x 20 ... 24 x 80 pyramid levels put less than two workgroups on a CU, so a
// wave has its SIMD almost to itself and nothing hides the weight loads it issues two taps (2 x 192 MFMA cycles) ahead of their use:
// SQ counters of 256 -> 256 @ 12 x 40 show the MFMA pipe 18 % busy and the waves parked on counters half of the time
// (profiles/round3_pmc_sq_hp.txt).  With WPF the kernel gives up occupancy it does not have anyway (256 VGPRs) and keeps the NEXT chunk's
// nine weight slices in flight in a second register set while the current chunk's are consumed: every load has a whole chunk to land.
#ifndef FP_TILE_WPF_HALO_SETS
#define FP_TILE_WPF_HALO_SETS 2
#endif
// the reflection-fold WPF instantiations keep ONE halo set: with two they need 260 registers, i.e. one workgroup per CU instead of two
#ifndef FP_TILE_WPF_FOLD_HALO_SETS
#define FP_TILE_WPF_FOLD_HALO_SETS 1
#endif
// waves per SIMD requested for the fp16-pair reflection-fold variants on the large grids (4 = 128 VGPRs)
#ifndef FP_TILE_HP_FOLD_WAVES
#define FP_TILE_HP_FOLD_WAVES FP_TILE_HP_WAVES
#endif
#ifndef FP_TILE_T16
#define FP_TILE_T16 0                // 1: build the 16 x 16-pixel 32-channel exact instantiations (round 6 experiment, needs FP_TILE_PIXB32=1 for two workgroups per CU)
#endif
#ifndef FP_TILE_WPF_EXACT
#define FP_TILE_WPF_EXACT 0          // 1: build the exact-format (NP = 3) weights-per-chunk-in-flight instantiations (round 6 experiment: 256 VGPRs + 60 AGPRs, no scratch; measured no gain, profiles/round6_notes.md 5e)
#endif
#ifndef FP_TILE_WPF_EXACT_DEFAULT_MAX_WG
#define FP_TILE_WPF_EXACT_DEFAULT_MAX_WG 0     // grids up to this many workgroups take them (0: off; FP_TILE_WPF_EXACT_MAX_WG overrides at run time)
#endif
#ifndef FP_TILE_PERSIST_BUILD
#define FP_TILE_PERSIST_BUILD 0     // 0: the tile loop compiled out (one tile per workgroup, as in rounds 1-3)
#endif

// Round 4.  (1) PERSISTENT workgroups: a launch of more tiles than the chip holds at once runs `gridDim.x` < nwg workgroups, each walking
// tiles blockIdx.x, blockIdx.x + gridDim.x, ... (gridDim.x is a multiple of 8, so a workgroup stays inside the logical-id range of its XCD).
// Per-wave clock stamps (profiles/round2_notes.md) put a tile's wave lifetime at prologue 5.4 k + epilogue 10.7 k cycles around 4 x ~2-3 k
// cycles of its own MFMA work: the MFMA pipe is saturated while the co-resident waves are in their taps and idle while they sit in
// prologues (kernarg -> amax slots -> first halo: three dependent memory latencies) and in epilogues that end with a store drain.  A
// persistent workgroup pays the prologue once; the NEXT tile's first halo and weight slices are issued before the current tile's epilogue and
// land under it; stores are fire-and-forget; the amax publication happens once per workgroup.
// (2) Operands arrive through raw buffer loads: a 32-bit per-lane byte offset computed once per tile + a wave-uniform (SGPR) offset per
// chunk / tap / plane.  The flat-address form cost five VALU (v_mad_u64_u32, v_lshl_add_u64 x2, v_mov, v_add) and six SALU instructions per
// tap in the steady loop (20 non-MFMA instructions per 6 MFMAs); an offset with bit 31 set is out of range and returns zeros without touching
// memory, which is the zero padding and the invalid halo of ragged tiles (no per-slot select at staging time).
// (3) Tried and removed (profiles/round4_notes.md): 16 x 16 pixel tiles with 64 x 64 per-wave register tiles (half the LDS reads and weight
// loads per MFMA, two workgroups per CU): 64 -> 64 @ 96 x 320 124.6 us against 95.9 with 8 x 16 tiles and four workgroups per CU.
struct TileGeo { int split, tile_n, tile_x, tile_y, n_img, y0, x0, n0; };
constexpr bool fp_tile_persistent(int tw, bool wpf, bool hp) { return FP_TILE_PERSIST_BUILD != 0 && !wpf && hp && tw == 16; }

// waves per SIMD asked of the register allocator.  Exact operands (round 6): with 32-byte halo pixels four workgroups fit a CU's LDS; the
// 32-channel forms use 99-113 registers (accumulators included) and take the fourth wave for free, the 64-channel forms need 140 and keep
// three unless FP_TILE_EXACT64_WAVES says otherwise (A/B knob: 4 = a 128-register budget).
#ifndef FP_TILE_EXACT32_WAVES
#define FP_TILE_EXACT32_WAVES 4
#endif
#ifndef FP_TILE_EXACT64_WAVES
#define FP_TILE_EXACT64_WAVES 1
#endif
template <int TH, int TW, int BN, bool FOLD, bool HP, bool WPF>
constexpr int fp_tile_min_waves() {
  if (WPF) return 1;
  if (TH * TW > 128) return 2;
  if (HP) return FOLD ? FP_TILE_HP_FOLD_WAVES : FP_TILE_HP_WAVES;
  if (fp_tile_pixb<TW>() == 32) return BN == 32 ? FP_TILE_EXACT32_WAVES : FP_TILE_EXACT64_WAVES;
  return 1;
}

template <int TH, int TW, int BN, int WM, int WN, bool FLIP, bool FOLD, int NP = 3, bool HP = false, bool WPF = false>
__global__ void __launch_bounds__(256, (fp_tile_min_waves<TH, TW, BN, FOLD, HP, WPF>()))
conv3x3_tile_bf3_kernel(const Tile3Args a) {
  static_assert(!HP || NP == 2, "the fp16-pair format has two planes");
  constexpr int WPL = HP ? 2 : 3;                    // planes per weight slice in the packed buffer
  constexpr int BM = (TH * TW + 127) / 128 * 128;    // rows of the M tile: 8x16 = 128; 6x20 = 120 (rows 120..127 idle); 16x16 = 256
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int HW2 = TW + 2, HPX = (TH + 2) * HW2;
  constexpr int NS = (HPX * 4 + 255) / 256;
  constexpr int PIXB = fp_tile_pixb<TW>();           // bytes per halo pixel per plane: 16 bf16 (+ 8 pad in the 48-byte layout)
  constexpr bool SWZ = PIXB == 32;                   // halves of a pixel swapped on odd halo rows (see fp_tile_pixb)
  constexpr int PLANE = HPX * PIXB;                   // bytes per plane
  constexpr int BUF = NP * PLANE;                    // bytes per halo buffer
  constexpr int NPIX = TH * TW;                      // valid rows of the M tile
  constexpr bool PERSIST = fp_tile_persistent(TW, WPF, HP);   // (the WPF grids are at most 400 workgroups, the 6 x 20 levels at most 768: one tile per workgroup)
  constexpr unsigned OOB = 0x80000000u;              // buffer offset of a load that must return zeros
  static_assert(WM * WN == 4 && NPIX <= BM && TM >= 1 && TN >= 1, "tile shape");
  __shared__ __attribute__((aligned(16))) unsigned char lds[(2 * BUF + 1023) / 1024 * 1024];   // whole LDS allocation granules

  // lane geometry: re-derived from an OPAQUE copy of the thread id at the top of every tile, so that nothing computed from it is invariant
  // of the tile loop -- left alone, loop-invariant code motion hoisted every tap's LDS address, every epilogue constant and the halo slot
  // table out of the loop and kept them live across it (128-VGPR instantiations spilled 40-120 registers)
  int t, lane, wave, idx, h, wm, wn;

  int ka = 0, kunscale = 0;                          // HP: source scale exponent, and -(ka + kw) for the epilogue (wave-uniform)
  float sa = 1.f;                                    // 2^ka, the source's scale
  float unscale = 1.f;                               // 2^kunscale: one fused multiply-add per element un-scales and adds the bias (wave-uniform)
  // the three amax slots: one vector load now, reduced behind the first tile's operand loads (fp_amax3_issue, fp_common.h)
  const unsigned amax_raw = HP ? fp_amax3_issue(a.amax_a, a.amax_a1, a.amax_w) : 0u;
  auto scales_ready = [&]() __attribute__((always_inline)) {
    if (!HP) return;
    unsigned ma, ma1, mw;
    fp_amax3_reduce(amax_raw, ma, ma1, mw);
    ka = fp_hp_exponent(max(ma, ma1), FP_HP_TARGET_ACT);
    kunscale = -(ka + fp_hp_exponent(mw, FP_HP_TARGET_W));
    sa = ldexpf(1.f, ka);
    unscale = ldexpf(1.f, kunscale);
  };
  // buffer resources (sizes checked on the host: every operand is smaller than 2^31 bytes, so bit 31 of an offset means "out of range")
  const int cs = a.C - a.Clo;                        // channels of `src` (the skip tensor of the concat gather, else the whole input)
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.src), 0, a.N * a.IH * a.IW * cs * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.src_lo ? a.src_lo : a.src), 0, a.src_lo ? a.N * (a.IH >> 1) * (a.IW >> 1) * a.Clo * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.w), 0, 9 * a.KC16 * WPL * a.Nout * 32, 0x00020000);
  const int wtap = a.KC16 * WPL * a.Nout * 32, wchunk = WPL * a.Nout * 32, wplane = a.Nout * 32;     // byte strides of the packed weights

  // ---- tile-independent lane geometry ----------------------------------------------------------------------------------------
  int lds_off[NS];
  int abase[SWZ ? 2 : 1][TM];                        // [parity of the tap's row offset]: the swapped halves follow the HALO row's parity
  auto lane_setup = [&]() {
    int tq = threadIdx.x;
    if (PERSIST) asm volatile("" : "+v"(tq));
    t = tq; lane = t & 63; wave = __builtin_amdgcn_readfirstlane(t >> 6); idx = lane & 31; h = lane >> 5;     // the wave index lives in an SGPR
    wm = wave / WN; wn = wave % WN;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      const int lin = t + 256 * k;
      if (SWZ) {
        const int hp = lin >> 2, hy = hp / HW2;
        lds_off[k] = hp < HPX ? hp * PIXB + ((((lin & 3) >> 1) ^ (hy & 1)) << 4) + (lin & 1) * 8 : -1;
      } else {
        lds_off[k] = (lin >> 2) < HPX ? (lin >> 2) * PIXB + (lin & 3) * 8 : -1;
      }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int pt = min((wm * TM + i) * 32 + idx, NPIX - 1);
      int py, px;
      fp_tile_pixel<TW>(pt, py, px);
      if (SWZ) {
        abase[0][i] = (py * HW2 + px) * PIXB + ((h ^ (py & 1)) << 4);
        abase[1][i] = (py * HW2 + px) * PIXB + ((h ^ ((py + 1) & 1)) << 4);
      } else {
        abase[0][i] = (py * HW2 + px) * PIXB + h * 16;
      }
    }
  };
  lane_setup();

  // ---- per-tile state ------------------------------------------------------------------------------------------------------
  TileGeo g;
  unsigned voff[NS], vofflo[NS];                     // byte offsets of the halo slots into `src` / `src_lo` (OOB: stored as zero)
  unsigned wvoff[TN];                                // byte offset of this lane's weight row inside a [n][16] plane
  bool has_r1 = false, has_rH = false, has_c1 = false, has_cW = false;
  unsigned m_r1[TM], m_rH[TM], m_c1[TM], m_cW[TM];   // all-ones / zero lane masks (reflection fold)
  int c_begin = 0, c_end = 0;
  auto decode = [&](int vb) {
    int wg = fp_xcd_remap(vb, a.nwg);
    TileGeo q;
    if (a.wmajor) {
      // layers whose bf16x3 weights do not fit an XCD's 4 MB L2 (512-channel 6x20 layers: 14 MB against 3 MB of activations):
      // pixel tiles vary fastest and (output-channel tile, split) slowest, so the contiguous id range of an XCD covers a few weight
      // slices for ALL pixel tiles and its private L2 keeps them -- with pixel-major ids every XCD streams every weight
      // (512->512 @6x20: 56.2 -> 49.5 us; 256->256 @12x40, 3.5 MB of weights, is better off pixel-major: 53.8 vs 57.3 us)
      q.tile_x = wg % a.tilesX; wg /= a.tilesX;
      q.tile_y = wg % a.tilesY; wg /= a.tilesY;
      q.n_img = wg % a.N; wg /= a.N;
      q.split = wg % a.SK;
      q.tile_n = wg / a.SK;
    } else {
      q.split = wg % a.SK; wg /= a.SK;
      q.tile_n = wg % a.tilesN; wg /= a.tilesN;
      q.tile_x = wg % a.tilesX; wg /= a.tilesX;
      q.tile_y = wg % a.tilesY;
      q.n_img = wg / a.tilesY;
    }
    q.y0 = q.tile_y * TH; q.x0 = q.tile_x * TW; q.n0 = q.tile_n * BN;
    return q;
  };
  // halo staging slots (unconditional loads; invalid slots carry the out-of-range offset and arrive as zeros), weight rows, fold masks
  auto setup = [&]() {
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      const int lin = t + 256 * k, hp = min(lin >> 2, HPX - 1);
      const int hy = hp / HW2, hx = hp - hy * HW2;
      int sy = g.y0 + hy - 1, sx = g.x0 + hx - 1;
      bool valid;
      if (a.mode == 0) {
        valid = sy >= 0 && sy < a.IH && sx >= 0 && sx < a.IW;
      } else {
        valid = sy >= -1 && sy <= a.IH && sx >= -1 && sx <= a.IW;
        sy = fp_reflect(sy, a.IH);
        sx = fp_reflect(sx, a.IW);
      }
      sy = min(max(sy, 0), a.IH - 1);
      sx = min(max(sx, 0), a.IW - 1);
      const unsigned pix = (g.n_img * a.IH + sy) * a.IW + sx;
      const unsigned pixlo = (g.n_img * (a.IH >> 1) + (sy >> 1)) * (a.IW >> 1) + (sx >> 1);     // nearest x2: src = dst // 2 (after the reflection)
      voff[k] = valid ? pix * (unsigned)(cs * 4) + (t & 3) * 16 : OOB;
      vofflo[k] = valid ? pixlo * (unsigned)(a.Clo * 4) + (t & 3) * 16 : OOB;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) wvoff[j] = (unsigned)(min(g.n0 + (wn * TN + j) * 32 + idx, a.Nout - 1) * 32 + h * 16);
    if (FOLD) {
      has_r1 = g.y0 <= 1 && 1 < g.y0 + TH; has_rH = g.y0 <= a.OH - 2 && a.OH - 2 < g.y0 + TH;
      has_c1 = g.x0 <= 1 && 1 < g.x0 + TW; has_cW = g.x0 <= a.OW - 2 && a.OW - 2 < g.x0 + TW;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int pt = min((wm * TM + i) * 32 + idx, NPIX - 1);
        int py, px;
        fp_tile_pixel<TW>(pt, py, px);
        const int yy = g.y0 + py, xx = g.x0 + px;
        m_r1[i] = yy == 1 ? ~0u : 0u;
        m_rH[i] = yy == a.OH - 2 ? ~0u : 0u;
        m_c1[i] = xx == 1 ? ~0u : 0u;
        m_cW[i] = xx == a.OW - 2 ? ~0u : 0u;
      }
    }
    c_begin = g.split * a.chunksPerSplit;
    c_end = min(a.KC16, c_begin + a.chunksPerSplit);
  };

  // HD register sets: the small grids of the WPF variant (less than one wave per SIMD, nothing else to hide a load behind) keep the
  // halos of the next TWO chunks in flight -- chunk j's halo lives in set (j - c_begin) & 1
  constexpr int HD = WPF ? (FOLD ? FP_TILE_WPF_FOLD_HALO_SETS : FP_TILE_WPF_HALO_SETS) : 1;
  float4 hreg[HD][NS];
  bool hzero[HD] = {};
  auto load_halo = [&](int cc, auto set_tag) {
    constexpr int hs = decltype(set_tag)::value % HD;
    hzero[hs] = cc * 16 + (t & 3) * 4 >= a.C;        // only ever true in the last chunk of a channel count that is not a multiple of 16
    if (cc * 16 < a.Clo) {                 // uniform: chunks of the upsampled half (Clo is a multiple of 16)
      const int so = cc * 64;
#pragma unroll
      for (int k = 0; k < NS; ++k) hreg[hs][k] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rl, vofflo[k], so, 0));
    } else {
      const int so = (cc * 16 - a.Clo) * 4;
#pragma unroll
      for (int k = 0; k < NS; ++k) hreg[hs][k] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff[k], so, 0));
    }
  };
  // fp32 -> three bf16 planes, 4 channels (8 bytes) per plane per slot
  auto store_halo = [&](int buf, auto set_tag) {
    constexpr int hs = decltype(set_tag)::value % HD;
    const bool zero_tail = (a.C & 15) != 0 && hzero[hs];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      if (lds_off[k] < 0) continue;
      f32x4 v = {hreg[hs][k].x, hreg[hs][k].y, hreg[hs][k].z, hreg[hs][k].w};
      if (zero_tail) v = f32x4{0.f, 0.f, 0.f, 0.f};
      unsigned char* p = lds + buf * BUF + lds_off[k];
      if (HP) {
        uint2 hq, mq;
        fp_hp_split4(v.x, v.y, v.z, v.w, sa, hq, mq);
        *reinterpret_cast<uint2*>(p) = hq;
        *reinterpret_cast<uint2*>(p + PLANE) = mq;
        continue;
      }
      const bf16x4 vh = __builtin_convertvector(v, bf16x4);
      const f32x4 r1 = v - __builtin_convertvector(vh, f32x4);
      const bf16x4 vm = __builtin_convertvector(r1, bf16x4);
      const f32x4 r2 = r1 - __builtin_convertvector(vm, f32x4);
      const bf16x4 vl = __builtin_convertvector(r2, bf16x4);
      *reinterpret_cast<uint2*>(p) = __builtin_bit_cast(uint2, vh);
      *reinterpret_cast<uint2*>(p + PLANE) = __builtin_bit_cast(uint2, vm);
      if (NP == 3) *reinterpret_cast<uint2*>(p + 2 * PLANE) = __builtin_bit_cast(uint2, vl);
    }
  };

  // ---- weight slices: [tap][chunk][plane][n][16] bf16; lane (n = idx, k-group = h) reads 16 bytes per plane -----------------
  uint4 bq[WPF ? 1 : 3][TN][NP];
  uint4 bw[WPF ? 2 : 1][WPF ? 9 : 1][TN][NP];          // WPF: [chunk parity][tap]
  int wtapoff[9];                                     // tap * wtap: nine SGPRs instead of a scalar multiply per load
#pragma unroll
  for (int tp = 0; tp < 9; ++tp) wtapoff[tp] = tp * wtap;
  auto load_b = [&](int tap, int cbase, uint4 (&bf)[TN][NP]) {      // cbase = chunk * wchunk
    const int so = wtapoff[tap] + cbase;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int p = 0; p < NP; ++p) bf[j][p] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rw, wvoff[j], so + p * wplane, 0));
  };

  f32x16 acc[TM][TN];

  // six products, smallest first; consecutive MFMAs alternate accumulators (i, j)
  constexpr int NPROD = NP == 3 ? 6 : (HP ? FP_HP_PRODUCTS : (FP_BF2_PRODUCTS));
  auto mma6 = [&](const uint4 (&af)[TM][NP], const uint4 (&bf)[TN][NP]) {
    // NP == 2: four products in the order mm, mh, hm, hh; three: mh, hm, hh
    constexpr int PA[6] = {NP == 3 ? 2 : 1, NP == 3 ? 0 : (NPROD == 4 ? 1 : 0), NP == 3 ? 1 : 0, NP == 3 ? 1 : 0, 0, 0};
    constexpr int PB[6] = {NP == 3 ? 0 : (NPROD == 4 ? 1 : 0), NP == 3 ? 2 : (NPROD == 4 ? 0 : 1), NP == 3 ? 1 : (NPROD == 4 ? 1 : 0), 0, 1, 0};
#pragma unroll
    for (int q = 0; q < NPROD; ++q)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          if (HP)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[i][PA[q]]), __builtin_bit_cast(f16x8, bf[j][PB[q]]),
                                                               acc[i][j], 0, 0, 0);
          else
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[i][PA[q]]), __builtin_bit_cast(bf16x8, bf[j][PB[q]]),
                                                                acc[i][j], 0, 0, 0);
  };

  // the first halo chunk and the first weight slices of the current tile (g): issued before the previous tile's epilogue
  auto prefetch_tile = [&]() {
    load_halo(c_begin, std::integral_constant<int, 0>{});
    if constexpr (WPF) {
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) load_b(tp, c_begin * wchunk, bw[0][tp]);
    } else {
      load_b(0, c_begin * wchunk, bq[0]);  // issued before the halo is consumed: one exposed load latency in the prologue, not two
      load_b(1, c_begin * wchunk, bq[1]);
    }
  };

  auto chunk = [&](auto par_tag, int cc) {
    constexpr int PAR = decltype(par_tag)::value;          // (cc - c_begin) & 1, as a constant: register-set index of the WPF weights
    const unsigned char* Hb = lds + ((cc - c_begin) & 1) * BUF;
    const int ccn = min(cc + 1, c_end - 1);
    const int wb = cc * wchunk, wbn = ccn * wchunk;
    // A fragments are read one tap ahead (two register sets): with 32-cycle MFMAs an LDS read issued right before its
    // consumer costs ~150 cycles per 384-cycle tap
    uint4 af[2][TM][NP];
    auto load_a = [&](int tap, uint4 (&dst)[TM][NP]) {
      const int ky = tap / 3, kx = tap % 3;
      const int toff = ((FLIP ? 2 - ky : ky) * HW2 + (FLIP ? 2 - kx : kx)) * PIXB;
      const int ap = SWZ ? ((FLIP ? 2 - ky : ky) & 1) : 0;
#pragma unroll
      for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int i = 0; i < TM; ++i) dst[i][p] = *reinterpret_cast<const uint4*>(Hb + p * PLANE + abase[ap][i] + toff);
    };
    // Border tiles of the reflection data-gradient run their fold taps (masked halo rows / columns, see kFoldTaps3) right behind
    // the regular tap that uses the SAME weight slice, from the registers it is already in: no extra weight loads (the former
    // separate fold loop paid one exposed global-load latency per fold tap, up to 16 per chunk on corner tiles).
    {
      constexpr bool WF = FOLD;      // interior tiles skip every fold tap through the uniform `need` tests
      load_a(0, af[0]);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        __builtin_amdgcn_sched_barrier(0);
        if (tap < 8) load_a(tap + 1, af[(tap + 1) & 1]);
        if constexpr (WPF) {
          load_b(tap, wbn, bw[PAR ^ 1][tap]);              // the next chunk's slice of this tap: a whole chunk ahead of its use
          mma6(af[tap & 1], bw[PAR][tap]);
        } else {
          if (tap < 7) load_b(tap + 2, wb, bq[(tap + 2) % 3]);
          else load_b(tap - 7, wbn, bq[(tap + 2) % 3]);
          mma6(af[tap & 1], bq[tap % 3]);
        }
        // issue order: one LDS / global read between consecutive MFMAs (this tap's MFMAs only depend on older reads)
        if (tap < 8) fp_sched_interleave<TM * NP, TN * NP, NPROD * TM * TN>();
        else fp_sched_interleave<0, TN * NP, NPROD * TM * TN>();
        if (WF) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const FoldTap ft = fold_tap3(e);
            if (ft.wtap != tap) continue;                                  // compile time
            const bool need_r = ft.rsel == 0 || (ft.rsel == 1 ? has_r1 : has_rH);
            const bool need_c = ft.csel == 0 || (ft.csel == 1 ? has_c1 : has_cW);
            if (!(need_r && need_c)) continue;                             // uniform per workgroup
            const int toff = (ft.ao * HW2 + ft.bo) * PIXB;
            const int ap = SWZ ? (ft.ao & 1) : 0;
            uint4 ax[TM][NP];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
              const unsigned mr = ft.rsel == 0 ? ~0u : (ft.rsel == 1 ? m_r1[i] : m_rH[i]);
              const unsigned mc = ft.csel == 0 ? ~0u : (ft.csel == 1 ? m_c1[i] : m_cW[i]);
              const unsigned mk = mr & mc;
#pragma unroll
              for (int p = 0; p < NP; ++p) {
                uint4 v = *reinterpret_cast<const uint4*>(Hb + p * PLANE + abase[ap][i] + toff);
                v.x &= mk; v.y &= mk; v.z &= mk; v.w &= mk;
                ax[i][p] = v;
              }
            }
            if constexpr (WPF) mma6(ax, bw[PAR][tap]);
            else mma6(ax, bq[tap % 3]);
          }
        }
      }
    }
    if (cc + 1 < c_end) {
      store_halo((cc + 1 - c_begin) & 1, std::integral_constant<int, PAR ^ 1>{});
      load_halo(min(cc + 1 + HD, c_end - 1), std::integral_constant<int, PAR ^ 1>{});
      __syncthreads();
    }
  };

  float ymax = 0.f;                                  // HP: largest stored magnitude of this lane over all its tiles (the consumer's scale)

  // ---- epilogue of one tile.  The flag tests are hoisted and every optional operand (addend, its mask, the activation source, the old
  // output) is loaded for eight rows BEFORE any arithmetic: element-at-a-time code serialised 16 dependent load latencies per
  // 32x32 block (and reloaded the bias 16 times).  Tiles that lie completely inside the image (all but the last row / column of
  // tiles) take a path without per-element bounds predication; for 16-wide tiles the row -> pixel map folds to constants.
  auto epilogue = [&](const TileGeo& e) {
    const unsigned epi = a.SK > 1 ? 0u : a.epi;
    const int act = a.SK > 1 ? 0 : a.act;
    float* const dst = a.SK > 1 ? a.part + (size_t)e.split * a.N * a.OH * a.OW * a.Nout : a.y;
    const bool interior = NPIX == BM && e.y0 + TH <= a.OH && e.x0 + TW <= a.OW;
    float bias_j[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = e.n0 + (wn * TN + j) * 32 + idx;
      bias_j[j] = ((epi & FP_EPI_BIAS) && n < a.Nout) ? a.bias[n] : 0.f;
    }
    // BatchNorm-backward reduction out of the data gradient's epilogue (wave-uniform; compile time for the forward variants): what this
    // workgroup stores IS g = dy * (relu_out > 0) of the BatchNorm below it, so its per-channel sums need no pass of their own
    const bool bnb = FLIP && !FOLD && a.bnb_part != nullptr;     // (encoder data gradients: zero padding; the reflection-fold variants belong to the decoders)
    float bs1[TN], bs2[TN], bmu[TN], bis[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = min(e.n0 + (wn * TN + j) * 32 + idx, a.Nout - 1);
      bs1[j] = bs2[j] = 0.f;
      bmu[j] = bnb ? a.bnb_mean[n] : 0.f;
      bis[j] = bnb ? a.bnb_invstd[n] : 0.f;
    }
    // accumulator register r of M block i -> tile pixel
    auto acc_pixel = [&](int i, int r, int& py, int& px) {
      if (TW == 16) {              // pt = blk*32 + (r&3) + 8*(r>>2) + 4h  =>  row = 2*blk + (r>>3), column from constants and h
        py = 2 * (wm * TM + i) + (r >> 3);
        px = ((r & 3) + 8 * ((r >> 2) & 1) + 4 * h - (SWZ ? 0 : 2 * ((r >> 3) & 1))) & 15;
      } else {
        fp_tile_pixel<TW>((wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, py, px);
      }
    };
    // byte offsets are 32-bit and unsigned (the host checks that the output is smaller than 2^29 elements): every access is
    // `global_* v, v_offset, s[base]` -- no 64-bit address arithmetic per element
    auto ldf = [](const float* base, unsigned boff) { return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + boff); };
    auto rows8 = [&](auto full_tag, int i, int j, int n, float bias, int half) {
      constexpr bool FULL = decltype(full_tag)::value;
      unsigned off[8];
      bool ok[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int r = half * 8 + k;
        int py, px;
        acc_pixel(i, r, py, px);
        const int oy = e.y0 + py, ox = e.x0 + px;
        ok[k] = FULL || ((wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h < NPIX && oy < a.OH && ox < a.OW);
        off[k] = (unsigned)(((e.n_img * a.OH + (FULL ? oy : min(oy, a.OH - 1))) * a.OW + (FULL ? ox : min(ox, a.OW - 1))) * a.Nout + n) * 4u;
      }
      float ad[8], mk[8], sv[8], yo[8];
      if (epi & FP_EPI_ADDEND) {
#pragma unroll
        for (int k = 0; k < 8; ++k) ad[k] = ldf(a.addend, off[k]);
      }
      if (epi & FP_EPI_ADDEND_MASK) {
#pragma unroll
        for (int k = 0; k < 8; ++k) mk[k] = ldf(a.addend_mask, off[k]);
      }
      if (epi & (FP_EPI_ACTGRAD_ELU | FP_EPI_ACTGRAD_RELU)) {
#pragma unroll
        for (int k = 0; k < 8; ++k) sv[k] = ldf(a.actsrc, off[k]);
      }
      if (epi & FP_EPI_ACCUM) {
#pragma unroll
        for (int k = 0; k < 8; ++k) yo[k] = ldf(a.y, off[k]);
      }
      float zb[8];
      if (bnb) {
#pragma unroll
        for (int k = 0; k < 8; ++k) zb[k] = ldf(a.bnb_z, off[k]);
      }
      // one wave-uniform branch per flag around an 8-element body (per-element tests get if-converted into selects that execute
      // every option for every element)
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = HP ? fmaf(acc[i][j][half * 8 + k], unscale, bias) : acc[i][j][half * 8 + k] + bias;   // * 2^kunscale is exact
      if (epi & FP_EPI_ADDEND) {
        if (epi & FP_EPI_ADDEND_MASK) {
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] += mk[k] > 0.f ? ad[k] : 0.f;
        } else {
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] += ad[k];
        }
      }
      if (epi & FP_EPI_ACTGRAD_ELU) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] *= (sv[k] > 0.f ? 1.f : sv[k] + 1.f);
      }
      if (epi & FP_EPI_ACTGRAD_RELU) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = sv[k] > 0.f ? v[k] : 0.f;
      }
      if (act == FP_ACT_ELU) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = fp_elu(v[k]);
      } else if (act == FP_ACT_RELU) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
      }
      if (epi & FP_EPI_ACCUM) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += yo[k];
      }
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (FULL || ok[k]) {
          *reinterpret_cast<float*>(reinterpret_cast<char*>(dst) + off[k]) = v[k];
          if (HP) ymax = fmaxf(ymax, fabsf(v[k]));
        }
      if (bnb) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float gk = (FULL || ok[k]) ? v[k] : 0.f;
          bs1[j] += gk;
          bs2[j] += gk * ((zb[k] - bmu[j]) * bis[j]);       // the arithmetic form of bn_bwd_reduce_kernel
        }
      }
    };
    // (before the stores: behind them the accumulators would have to outlive the whole store loop, and its address arithmetic spilled)
    if (!FLIP && a.bn_part) {                          // (compile time for the data-gradient variants: their register budgets are unchanged)
      // BatchNorm statistics out of the epilogue (wave-uniform; the launcher sets bn_part only for unsplit grids without bias / addend /
      // activation, so the stored value is acc * unscale): a lane's TM x 16 values of output channel n -> two-pass (count, mean, M2) in
      // registers -> Chan merge with the other half-wave's pixels -> across the WM waves that share the channel through LDS, fixed order
      // -> part[pixel tile][n].  bn_stats_final_kernel (bn_pool.hip) merges the tiles; the activation is never read for its statistics.
      float* const st = reinterpret_cast<float*>(lds) + 16;          // [wave][TN * 32][3] floats
      __syncthreads();                                               // every wave is done with the halo buffers
      auto lane_stats = [&](auto full_tag, int j) {    // two passes over the lane's registers; validity recomputed, not kept
        constexpr bool FULL = decltype(full_tag)::value;
        auto ok_at = [&](int i, int r) {
          if (FULL) return NPIX == BM || (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h < NPIX;     // tile inside the image
          int py, px;
          acc_pixel(i, r, py, px);
          return (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h < NPIX && e.y0 + py < a.OH && e.x0 + px < a.OW;
        };
        float cnt = 0.f, sum = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const bool ok = ok_at(i, r);
            cnt += ok ? 1.f : 0.f;
            sum += ok ? acc[i][j][r] * unscale : 0.f;
          }
        FpWf w{cnt, cnt > 0.f ? sum / cnt : 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float dv = acc[i][j][r] * unscale - w.mean;
            w.m2 += ok_at(i, r) ? dv * dv : 0.f;
          }
        return w;
      };
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const FpWf w = (e.y0 + TH <= a.OH && e.x0 + TW <= a.OW) ? lane_stats(std::true_type{}, j) : lane_stats(std::false_type{}, j);
        const FpWf o{__shfl_xor(w.n, 32, 64), __shfl_xor(w.mean, 32, 64), __shfl_xor(w.m2, 32, 64)};
        FpWf lo = h == 0 ? w : o;                                      // both half-waves form merge(h = 0, h = 1)
        fp_wf_merge(lo, h == 0 ? o : w);
        if (h == 0) {
          float* q = st + ((wave * TN + j) * 32 + idx) * 3;
          q[0] = lo.n; q[1] = lo.mean; q[2] = lo.m2;
        }
      }
      __syncthreads();
      if (t < BN) {                                                  // one thread per output channel of the tile: merge the WM waves in order
        const int cw = t / (TN * 32), cj = (t / 32) % TN, ci = t & 31;   // wn, j, lane of the channel
        const float* q = st + (((0 * WN + cw) * TN + cj) * 32 + ci) * 3;
        FpWf m{q[0], q[1], q[2]};
#pragma unroll
        for (int k = 1; k < WM; ++k) {
          const float* qk = st + (((k * WN + cw) * TN + cj) * 32 + ci) * 3;
          fp_wf_merge(m, FpWf{qk[0], qk[1], qk[2]});
        }
        const int n = e.n0 + t;
        if (n < a.Nout) {
          float* o = a.bn_part + ((size_t)((e.n_img * a.tilesY + e.tile_y) * a.tilesX + e.tile_x) * a.Nout + n) * 3;
          o[0] = m.n; o[1] = m.mean; o[2] = m.m2;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = e.n0 + (wn * TN + j) * 32 + idx;
        if (n >= a.Nout) continue;
        if (interior) {
          rows8(std::true_type{}, i, j, n, bias_j[j], 0);
          rows8(std::true_type{}, i, j, n, bias_j[j], 1);
        } else {
          rows8(std::false_type{}, i, j, n, bias_j[j], 0);
          rows8(std::false_type{}, i, j, n, bias_j[j], 1);
        }
      }
    if (bnb) {       // lane -> half-wave pair -> the WM waves that share the channel (through LDS), fixed order -> part[pixel tile][n]
      float* const st = reinterpret_cast<float*>(lds) + 16;          // [wave][TN * 32][2] floats
      __syncthreads();                                               // every wave is done with the halo buffers
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const float t1 = bs1[j] + __shfl_xor(bs1[j], 32, 64), t2 = bs2[j] + __shfl_xor(bs2[j], 32, 64);
        if (h == 0) {
          float* q = st + ((wave * TN + j) * 32 + idx) * 2;
          q[0] = t1; q[1] = t2;
        }
      }
      __syncthreads();
      if (t < BN) {
        const int cw = t / (TN * 32), cj = (t / 32) % TN, ci = t & 31;   // wn, j, lane of the channel
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int k = 0; k < WM; ++k) {
          const float* qk = st + (((k * WN + cw) * TN + cj) * 32 + ci) * 2;
          t1 += qk[0]; t2 += qk[1];
        }
        const int n = e.n0 + t;
        if (n < a.Nout) {
          float* o = a.bnb_part + ((size_t)((e.n_img * a.tilesY + e.tile_y) * a.tilesX + e.tile_x) * a.Nout + n) * 2;
          o[0] = t1; o[1] = t2;
        }
      }
    }
  };

  // ---- the tile loop ---------------------------------------------------------------------------------------------------------
  int vb = blockIdx.x;
  g = decode(vb);
  setup();
  prefetch_tile();
  scales_ready();
  for (;;) {
    if (PERSIST) lane_setup();
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    store_halo(0, std::integral_constant<int, 0>{});
    load_halo(min(c_begin + 1, c_end - 1), std::integral_constant<int, 1>{});
    if (HD == 2) load_halo(min(c_begin + 2, c_end - 1), std::integral_constant<int, 0>{});
    __syncthreads();
    for (int cc = c_begin; cc < c_end; cc += 2) {
      chunk(std::integral_constant<int, 0>{}, cc);
      if (cc + 1 < c_end) chunk(std::integral_constant<int, 1>{}, cc + 1);
    }
    const TileGeo e = g;
    vb += gridDim.x;
    const bool more = PERSIST && vb < a.nwg;
#ifndef FP_TILE_PREFETCH_LATE
    if (more) {                            // the next tile's first operands travel while this tile's epilogue runs
      g = decode(vb);
      setup();
      prefetch_tile();
    }
    epilogue(e);
    if (!more) break;
#else
    epilogue(e);
    if (!more) break;
    g = decode(vb);
    setup();
    prefetch_tile();
#endif
    __syncthreads();                       // every wave has left the halo buffers (and the statistics scratch) of tile e
  }
  if (HP && a.amax_out && a.SK <= 1) {               // one publication per workgroup (the halo buffers are free by now)
    ymax = fp_wave_max(ymax);
    float* wmax = reinterpret_cast<float*>(lds);
    __syncthreads();
    if (lane == 0) wmax[wave] = ymax;
    __syncthreads();
    if (t == 0) fp_amax_publish(a.amax_out, blockIdx.x, fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3])));
  }
}

// workgroups of a launch: every tile its own workgroup, or -- when the chip cannot hold them at once -- an even share of whole rounds
// (a multiple of 8: a persistent workgroup keeps to the logical-id range of its XCD, see fp_xcd_remap)
template <int TH, int TW, int BN, int WM, int WN, bool FLIP, bool FOLD, int NP = 3, bool HP = false, bool WPF = false>
int launch3(Tile3Args& a, hipStream_t stream) {
  static const int persist = getenv("FP_TILE_PERSIST") ? atoi(getenv("FP_TILE_PERSIST")) : 1;   // 0: one workgroup per tile (rounds 1-3)
  static int resident = 0;                           // workgroups of THIS instantiation the chip holds at once
  if (!resident) {
    int per_cu = 0, dev = 0, cus = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)conv3x3_tile_bf3_kernel<TH, TW, BN, WM, WN, FLIP, FOLD, NP, HP, WPF>, 256, 0) != hipSuccess ||
        per_cu < 1)
      per_cu = 1;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    resident = per_cu * (cus > 0 ? cus : 256);
    if (persist > 1) resident = persist;             // experiments: an explicit workgroup budget
  }
  int grid = a.nwg;
  if (persist && fp_tile_persistent(TW, WPF, HP) && a.nwg > resident) {
    const int rounds = (a.nwg + resident - 1) / resident;
    grid = ((a.nwg + rounds - 1) / rounds + 7) & ~7;
    if (grid > a.nwg) grid = a.nwg;
  }
  fp_launch((conv3x3_tile_bf3_kernel<TH, TW, BN, WM, WN, FLIP, FOLD, NP, HP, WPF>), dim3(grid), dim3(256), 0u, stream, a);
  return fp_check_launch("fp_conv3x3_bf3");
}


// tile geometry + split factor for a problem, or ok = false
struct Plan3 { bool ok; int th, tw, bn, tilesX, tilesY, tilesN, SK, chunksPerSplit; };
Plan3 plan3(const fp_conv_desc* d) {
  Plan3 p = {};
  if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != 1 || d->C0 % 4) return p;
  if (d->gather == FP_GATHER_FWD_REFLECT_UP2) {       // cat[nearest_x2(low), skip]: chunks switch source at C0
    if (d->C0 % 16 || d->C1 % 4 || d->C1 < 0 || d->IH % 2 || d->IW % 2) return p;
  } else {
    if (d->C1 != 0) return p;
    if (d->gather != FP_GATHER_FWD_ZERO && d->gather != FP_GATHER_FWD_REFLECT && d->gather != FP_GATHER_DGRAD_ZERO &&
        d->gather != FP_GATHER_DGRAD_REFLECT)
      return p;
  }
  if (d->OH != d->IH || d->OW != d->IW || d->IH < 2 || d->IW < 2) return p;
  // 8x16 tiles, or 6x20 tiles for the 6x20 / 12x40 levels of a 192x640 pyramid; <= 25 % padded work
  auto waste_ok = [&](int th, int tw, int rows) {
    const int64_t ty = fp_ceil_div(d->OH, th), tx = fp_ceil_div(d->OW, tw);
    return ty * tx * rows * 4 <= (int64_t)d->OH * d->OW * 5;
  };
  if (waste_ok(8, 16, 128)) { p.th = 8; p.tw = 16; }
  else if (waste_ok(6, 20, 128)) { p.th = 6; p.tw = 20; }
  else return p;
  p.bn = d->Nout <= 32 ? 32 : 64;
  static const int exp_bn32 = getenv("FP_TILE_BN32_BELOW") ? atoi(getenv("FP_TILE_BN32_BELOW")) : 0;     // experiment knobs
  static const int exp_maxsk = getenv("FP_TILE_MAX_SK") ? atoi(getenv("FP_TILE_MAX_SK")) : 16;
  {
    const int64_t t64 = (int64_t)d->N * fp_ceil_div(d->OW, p.tw) * fp_ceil_div(d->OH, p.th) * fp_ceil_div(d->Nout, 64);
    if (exp_bn32 && t64 < exp_bn32) p.bn = 32;      // more, narrower workgroups instead of split-K partials
  }
#if FP_TILE_T16
  // round 6 experiment: 16 x 16 pixel tiles for the 32-channel forms on large grids (two accumulators and twice the MFMAs per wave; with
  // 32-byte halo pixels 62 KB of LDS: two workgroups per CU).  FP_TILE_T16_BN32=1 selects it at run time (exact operands only, see run_tile3).
  static const int t16 = getenv("FP_TILE_T16_BN32") ? atoi(getenv("FP_TILE_T16_BN32")) : 0;
  if (t16 && p.bn == 32 && p.th == 8 && d->OH % 16 == 0 && d->OW % 16 == 0 && (int64_t)d->N * (d->OH / 16) * (d->OW / 16) >= 2048) p.th = 16;
#endif
  p.tilesX = (int)fp_ceil_div(d->OW, p.tw); p.tilesY = (int)fp_ceil_div(d->OH, p.th); p.tilesN = (int)fp_ceil_div(d->Nout, p.bn);
  const int64_t tiles = (int64_t)d->N * p.tilesY * p.tilesX * p.tilesN;
  const int KC16 = (d->C0 + d->C1 + 15) / 16;
  // small grids: split the channel chunks up to one full round of resident workgroups, >= 2 chunks per split, <= 16 partial copies
  // grids of >= 160 tiles run unsplit even though they fill < 1 workgroup per CU: with split-K every workgroup of the single round ends in
  // the same burst of partial stores (23.6 MB for 256 -> 256 @ 12 x 40: ~8 us, not overlapped with anything) and a reduce launch follows;
  // 256 -> 256 @ 12 x 40: 45.6 -> 36.7 us, 128 -> 128 @ 24 x 80: 44.8 -> 31.7 us, training step 15.07 -> 14.71 ms (FP_TILE_SK1_FROM=384: old rule)
  static const int sk1_from = getenv("FP_TILE_SK1_FROM") ? atoi(getenv("FP_TILE_SK1_FROM")) : 160;
  int64_t sk = 1;
  if (tiles < sk1_from) {
    static const int sk_target = getenv("FP_TILE_SK_TARGET") ? atoi(getenv("FP_TILE_SK_TARGET")) : 768;
    sk = sk_target / tiles;                         // three workgroups per CU are resident: at most one full round of 768
    if (sk > KC16 / 2) sk = KC16 / 2;
    if (sk > 16) sk = 16;
    if (sk > exp_maxsk) sk = exp_maxsk;
    if (sk < 1) sk = 1;
  }
  p.chunksPerSplit = (int)fp_ceil_div(KC16, sk);
  p.SK = (int)fp_ceil_div(KC16, p.chunksPerSplit);
  if (tiles * p.SK < 128) return p;                  // still far too small to fill the chip: leave it to the flattened kernel
  p.ok = true;
  return p;
}

}  // namespace

extern "C" int fp_conv3x3_bf3_supported(const fp_conv_desc* d) { return d && plan3(d).ok ? 1 : 0; }

// bytes of split-K scratch fp_conv3x3_bf3 needs for this problem (0 = none)
extern "C" int64_t fp_conv3x3_bf3_workspace(const fp_conv_desc* d) {
  if (!d) return 0;
  const Plan3 p = plan3(d);
  if (!p.ok || p.SK <= 1) return 0;
  return (int64_t)p.SK * d->N * d->OH * d->OW * d->Nout * (int64_t)sizeof(float);
}

namespace {
struct HpSlots { const unsigned* a; const unsigned* a1; const unsigned* w; unsigned* out; };

int run_tile3(const char* who, const fp_conv_desc* d, const float* src, const float* src1, const void* wpacked, const float* bias,
              const float* addend, const float* addend_mask, const float* actsrc, float* y, void* workspace, int64_t workspace_bytes,
              const HpSlots* hp, const fp_aux* aux, hipStream_t stream) {
  const FpBnSink bn_sink = fp_bn_sink_of(aux);       // (zeroes *bn_nblk_out: an argument error below reports "nothing emitted")
  FP_REQUIRE(d && src && wpacked && y, "fp_conv3x3_bf3 / fp_conv3x3_hp: null pointer");
  const Plan3 p = plan3(d);
  FP_REQUIRE(p.ok, "fp_conv3x3_bf3 / fp_conv3x3_hp: shape not supported (see fp_conv3x3_bf3_supported)");
  FP_REQUIRE(!(d->epi & FP_EPI_BIAS) || bias, "fp_conv3x3_bf3 / fp_conv3x3_hp: bias missing");
  FP_REQUIRE(!(d->epi & FP_EPI_ADDEND) || addend, "fp_conv3x3_bf3 / fp_conv3x3_hp: addend missing");
  FP_REQUIRE(!(d->epi & FP_EPI_ADDEND_MASK) || addend_mask, "fp_conv3x3_bf3 / fp_conv3x3_hp: addend_mask missing");
  FP_REQUIRE(!(d->epi & (FP_EPI_ACTGRAD_ELU | FP_EPI_ACTGRAD_RELU)) || actsrc, "fp_conv3x3_bf3 / fp_conv3x3_hp: actsrc missing");
  FP_REQUIRE(p.SK <= 1 || (workspace && workspace_bytes >= fp_conv3x3_bf3_workspace(d)), "fp_conv3x3_bf3 / fp_conv3x3_hp: workspace too small");
  FP_REQUIRE((int64_t)d->N * d->OH * d->OW * d->Nout < ((int64_t)1 << 29), "fp_conv3x3_bf3 / fp_conv3x3_hp: output larger than 2^29 elements");
  // operands are addressed with 32-bit byte offsets (raw buffer loads; bit 31 = "out of range, reads zero")
  FP_REQUIRE((int64_t)d->N * d->IH * d->IW * (d->C0 + d->C1) * 4 < ((int64_t)1 << 31), "fp_conv3x3_bf3 / fp_conv3x3_hp: input larger than 2^31 bytes");
  Tile3Args a;
  a.bn_part = nullptr;
  a.bnb_part = nullptr; a.bnb_z = a.bnb_mean = a.bnb_invstd = nullptr;
  const bool up2 = d->gather == FP_GATHER_FWD_REFLECT_UP2;
  FP_REQUIRE(!up2 || d->C1 == 0 || src1, "fp_conv3x3_bf3 / fp_conv3x3_hp: the concat gather needs the skip tensor (src1)");
  a.src_lo = up2 ? src : nullptr; a.Clo = up2 ? d->C0 : 0;
  a.src = up2 ? (d->C1 ? src1 : src) : src; a.w = (const unsigned short*)wpacked; a.bias = bias; a.addend = addend; a.addend_mask = addend_mask; a.actsrc = actsrc;
  a.y = y;
  a.N = d->N; a.OH = d->OH; a.OW = d->OW; a.IH = d->IH; a.IW = d->IW; a.C = d->C0 + d->C1; a.Nout = d->Nout;
  a.KC16 = (d->C0 + d->C1 + 15) / 16;
  const bool flip = d->gather == FP_GATHER_DGRAD_ZERO || d->gather == FP_GATHER_DGRAD_REFLECT;
  const bool fold = d->gather == FP_GATHER_DGRAD_REFLECT;
  a.mode = (d->gather == FP_GATHER_FWD_REFLECT || up2) ? 1 : 0;
  a.act = d->act; a.epi = d->epi & ~FP_EPI_BF16X2;
  a.tilesX = p.tilesX; a.tilesY = p.tilesY; a.tilesN = p.tilesN; a.SK = p.SK; a.chunksPerSplit = p.chunksPerSplit;
  a.part = (float*)workspace;
  a.amax_a = hp ? hp->a : nullptr; a.amax_a1 = hp && up2 && d->C1 ? hp->a1 : nullptr; a.amax_w = hp ? hp->w : nullptr;
  a.amax_out = hp ? hp->out : nullptr;
  {
    // BatchNorm-statistics sink (fp_aux.bn_part): only the plain forward form on an unsplit grid emits -- its stored value is the
    // accumulator itself -- everything else reports 0 blocks and the caller runs fp_bn_train_stats as before
    const int64_t blocks = (int64_t)d->N * p.tilesY * p.tilesX;
    bool emit;
    if (bn_sink.z) {       // backward form (fp_aux.bnb_*): a data gradient on an unsplit grid that overwrites its output
      emit = bn_sink.part && p.SK <= 1 && flip && !fold && !(d->epi & FP_EPI_ACCUM) && d->act == 0 && blocks * d->Nout * 2 <= bn_sink.cap_floats;
      if (emit) { a.bnb_part = bn_sink.part; a.bnb_z = bn_sink.z; a.bnb_mean = bn_sink.mean; a.bnb_invstd = bn_sink.invstd; }
    } else {
      emit = bn_sink.part && p.SK <= 1 && !flip && (d->epi & ~(unsigned)FP_EPI_BF16X2) == 0 && d->act == 0 &&
             blocks * d->Nout * 3 <= bn_sink.cap_floats;
      if (emit) a.bn_part = bn_sink.part;
    }
    if (bn_sink.nblk_out) *bn_sink.nblk_out = emit ? (int32_t)blocks : 0;
  }
  // a split forward grid with the statistics sink armed: its reduce launch writes the statistics (fp_splitk_reduce_stats_launch below)
  const bool bnb_in_reduce = bn_sink.part && bn_sink.z && p.SK > 1 && flip && !fold && !(d->epi & FP_EPI_ACCUM) && d->act == 0;
  const bool stats_in_reduce = bn_sink.part && !bn_sink.z && p.SK > 1 && !flip && (d->epi & ~(unsigned)FP_EPI_BF16X2) == 0 && d->act == 0;
  const int planes = hp ? 4 : 6;                    // bytes of packed weight per element
  a.wmajor = (int64_t)9 * (d->C0 + d->C1) * d->Nout * planes > ((int64_t)4 << 20);
#if FP_TILE_T16
  FP_REQUIRE(!(p.th == 16 && (hp || (d->epi & FP_EPI_BF16X2))), "fp_conv3x3: FP_TILE_T16_BN32 is an exact-operand experiment");
#endif
  a.nwg = d->N * p.tilesY * p.tilesX * p.tilesN * p.SK;
  int rc;
#define FP_L3X(TH_, TW_, NP_, HP_)                                                                                            \
  (p.bn == 32 ? (fold ? launch3<TH_, TW_, 32, 4, 1, true, true, NP_, HP_>(a, stream)                                          \
                      : (flip ? launch3<TH_, TW_, 32, 4, 1, true, false, NP_, HP_>(a, stream)                                 \
                              : launch3<TH_, TW_, 32, 4, 1, false, false, NP_, HP_>(a, stream)))                              \
              : (fold ? launch3<TH_, TW_, 64, 2, 2, true, true, NP_, HP_>(a, stream)                                          \
                      : (flip ? launch3<TH_, TW_, 64, 2, 2, true, false, NP_, HP_>(a, stream)                                 \
                              : launch3<TH_, TW_, 64, 2, 2, false, false, NP_, HP_>(a, stream))))
  // small grids (64-channel tiles of the 6 x 20 ... 24 x 80 levels: under ~1.5 workgroups per CU): the weights-per-chunk-in-flight variant
  static const int wpf_max = getenv("FP_TILE_WPF_MAX_WG") ? atoi(getenv("FP_TILE_WPF_MAX_WG")) : 400;
  static const int wpf_exact_max = getenv("FP_TILE_WPF_EXACT_MAX_WG") ? atoi(getenv("FP_TILE_WPF_EXACT_MAX_WG")) : FP_TILE_WPF_EXACT_DEFAULT_MAX_WG;
  (void)wpf_exact_max;
  if (hp && p.bn == 64 && a.nwg <= wpf_max) {
#define FP_L3W(TH_, TW_)                                                                                                     \
  (fold ? launch3<TH_, TW_, 64, 2, 2, true, true, 2, true, true>(a, stream)                                                 \
        : (flip ? launch3<TH_, TW_, 64, 2, 2, true, false, 2, true, true>(a, stream)                                        \
                : launch3<TH_, TW_, 64, 2, 2, false, false, 2, true, true>(a, stream)))
    rc = p.th == 8 ? FP_L3W(8, 16) : FP_L3W(6, 20);
#undef FP_L3W
  } else if (hp) {
    rc = p.th == 8 ? FP_L3X(8, 16, 2, true) : FP_L3X(6, 20, 2, true);
#if FP_TILE_WPF_EXACT
  } else if (!hp && !(d->epi & FP_EPI_BF16X2) && p.bn == 64 && a.nwg <= wpf_exact_max) {
    // round 6: the exact format's small grids get the weights-per-chunk-in-flight form as well (one workgroup per CU there: the second
    // register set -- 216 registers of weight slices -- costs occupancy the grid does not provide)
#define FP_L3WE(TH_, TW_)                                                                                                    \
  (fold ? launch3<TH_, TW_, 64, 2, 2, true, true, 3, false, true>(a, stream)                                                \
        : (flip ? launch3<TH_, TW_, 64, 2, 2, true, false, 3, false, true>(a, stream)                                       \
                : launch3<TH_, TW_, 64, 2, 2, false, false, 3, false, true>(a, stream)))
    rc = p.th == 8 ? FP_L3WE(8, 16) : FP_L3WE(6, 20);
#undef FP_L3WE
#endif
  } else if ((d->epi & FP_EPI_BF16X2) && !flip) {   // opt-in inference mode: two bf16 terms per operand, three products (forward only)
    if (p.th == 8) rc = p.bn == 32 ? launch3<8, 16, 32, 4, 1, false, false, 2>(a, stream) : launch3<8, 16, 64, 2, 2, false, false, 2>(a, stream);
    else rc = p.bn == 32 ? launch3<6, 20, 32, 4, 1, false, false, 2>(a, stream) : launch3<6, 20, 64, 2, 2, false, false, 2>(a, stream);
#if FP_TILE_T16
  } else if (p.th == 16) {
    rc = fold ? launch3<16, 16, 32, 4, 1, true, true, 3, false>(a, stream)
              : (flip ? launch3<16, 16, 32, 4, 1, true, false, 3, false>(a, stream) : launch3<16, 16, 32, 4, 1, false, false, 3, false>(a, stream));
#endif
  } else {
    rc = p.th == 8 ? FP_L3X(8, 16, 3, false) : FP_L3X(6, 20, 3, false);
  }
#undef FP_L3X
  (void)who;
  if (rc || p.SK <= 1) return rc;
  if (bnb_in_reduce) {
    int rc2 = 0;
    const int nb = fp_splitk_reduce_bnb_launch(a.part, p.SK, (int64_t)d->N * d->OH * d->OW, d->Nout, bias, addend, addend_mask, actsrc, y, d->act,
                                               d->epi & ~FP_EPI_BF16X2, stream, hp ? hp->out : nullptr, bn_sink.z, bn_sink.mean, bn_sink.invstd,
                                               bn_sink.part, bn_sink.cap_floats, &rc2);
    if (nb > 0) {
      if (bn_sink.nblk_out) *bn_sink.nblk_out = nb;
      return rc2;
    }
  }
  if (stats_in_reduce) {
    int rc2 = 0;
    const int nb = fp_splitk_reduce_stats_launch(a.part, p.SK, (int64_t)d->N * d->OH * d->OW, d->Nout, y, stream, hp ? hp->out : nullptr, bn_sink.part,
                                                 bn_sink.cap_floats, &rc2);
    if (nb > 0) {
      if (bn_sink.nblk_out) *bn_sink.nblk_out = nb;
      return rc2;
    }
  }
  return fp_splitk_reduce_launch(a.part, p.SK, (int64_t)d->N * d->OH * d->OW, d->Nout, bias, addend, addend_mask, actsrc, y, d->act,
                                 d->epi & ~FP_EPI_BF16X2, stream, hp ? hp->out : nullptr);      // split-K launches publish max |y| here
}
}  // namespace

extern "C" int fp_conv3x3_bf3(const fp_conv_desc* d, const float* src, const float* src1, const void* wpacked_bf3, const float* bias,
                              const float* addend, const float* addend_mask, const float* actsrc, float* y, void* workspace,
                              int64_t workspace_bytes, const fp_aux* aux, fp_stream_t stream_) {
  return run_tile3("fp_conv3x3_bf3", d, src, src1, wpacked_bf3, bias, addend, addend_mask, actsrc, y, workspace, workspace_bytes, nullptr,
                   aux, (hipStream_t)stream_);
}

// Same operation with fp16-pair operands (fp_common.h): weights from fp_pack_conv_weight_hp / FP_PACK_{FWD,DGRAD}_HP jobs with the
// slot `amax_w` they were scaled by; `amax_src` (and `amax_src1` for the skip tensor of the concat gather) hold max |x| of the
// source tensor(s) -- fp_amax_f32 or a producer's `amax_out`; `amax_out` (optional, zeroed by the caller) receives max |y|.
extern "C" int fp_conv3x3_hp(const fp_conv_desc* d, const float* src, const float* src1, const void* wpacked_hp, const float* bias,
                             const float* addend, const float* addend_mask, const float* actsrc, float* y, void* workspace,
                             int64_t workspace_bytes, const uint32_t* amax_src, const uint32_t* amax_src1, const uint32_t* amax_w,
                             uint32_t* amax_out, const fp_aux* aux, fp_stream_t stream_) {
  FP_REQUIRE(amax_src && amax_w, "fp_conv3x3_hp: amax slots missing");
  FP_REQUIRE(!(d && d->gather == FP_GATHER_FWD_REFLECT_UP2 && d->C1) || amax_src1, "fp_conv3x3_hp: the skip tensor's amax slot is missing");
  const HpSlots hp = {amax_src, amax_src1, amax_w, amax_out};
  return run_tile3("fp_conv3x3_hp", d, src, src1, wpacked_hp, bias, addend, addend_mask, actsrc, y, workspace, workspace_bytes, &hp,
                   aux, (hipStream_t)stream_);
}
