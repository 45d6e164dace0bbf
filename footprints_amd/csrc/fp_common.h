// Shared device/host helpers for libfootprints_hip (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "footprints_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

int fp_set_error(int code, const char* fmt, ...);
int fp_check_launch(const char* what);

#define FP_REQUIRE(cond, ...)                          \
  do {                                                 \
    if (!(cond)) return fp_set_error(FP_EINVAL, __VA_ARGS__); \
  } while (0)

static inline int64_t fp_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- every kernel launch of the library goes through fp_launch: it launches, and while a launch plan is recording (plan.cpp) it also
// appends the launch -- kernel, geometry, stream and a private copy of the argument bytes -- to the plan
#include <tuple>
#include <utility>
bool fp_plan_recording();
void fp_plan_push_kernel(const void* func, dim3 grid, dim3 block, unsigned shmem, hipStream_t stream, void** args, const size_t* sizes, int nargs);
void fp_plan_mark_failed();
hipError_t fp_launch_timed(const void* func, dim3 grid, dim3 block, void** args, unsigned shmem, hipStream_t stream);   // plan.cpp: + fp_ktime_* events when on

template <typename... KArgs, size_t... I>
inline void fp_launch_impl(void (*kernel)(KArgs...), dim3 grid, dim3 block, unsigned shmem, hipStream_t stream, std::tuple<KArgs...>& st,
                           std::index_sequence<I...>) {
  void* ptrs[sizeof...(KArgs) ? sizeof...(KArgs) : 1] = {(void*)&std::get<I>(st)...};
  // launch first: only a launch the runtime accepted becomes a plan node (the entry point's fp_check_launch reports the error itself)
  const hipError_t e = fp_launch_timed((const void*)kernel, grid, block, ptrs, shmem, stream);
  if (fp_plan_recording()) {
    if (e != hipSuccess) {
      fp_plan_mark_failed();
      return;
    }
    const size_t sizes[sizeof...(KArgs) ? sizeof...(KArgs) : 1] = {sizeof(KArgs)...};
    fp_plan_push_kernel((const void*)kernel, grid, block, shmem, stream, ptrs, sizes, (int)sizeof...(KArgs));
  }
}

template <typename... KArgs, typename... Args>
inline void fp_launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, unsigned shmem, hipStream_t stream, Args&&... args) {
  static_assert(sizeof...(KArgs) == sizeof...(Args), "fp_launch: argument count differs from the kernel's parameter count");
  std::tuple<KArgs...> st{static_cast<KArgs>(std::forward<Args>(args))...};       // the kernel's exact parameter types
  fp_launch_impl(kernel, grid, block, shmem, stream, st, std::index_sequence_for<KArgs...>{});
}

// A/B switches of the library (read once per call site through a function-local static)
#include <stdlib.h>
static inline bool fp_env_flag(const char* name) {
  const char* v = getenv(name);
  return v && atoi(v) != 0;
}

// ---- XCD-aware workgroup id remap (8 XCDs, private L2 each): consecutive logical tiles share halo rows
// and weight slices, so give each XCD a contiguous range of logical ids. Bijective for any grid size.
__device__ __forceinline__ int fp_xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, local = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + local;
}

// ---- gather geometry shared by the implicit-GEMM forward/dgrad kernel and the wgrad kernel ------------
struct FpGeom {
  int N, OH, OW, IH, IW, C0, C1, KH, KW, stride, pad, gather;
};

__device__ __forceinline__ int fp_reflect(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

// Source pixel indices for output pixel (n, oy, ox) and tap (ky, kx).
// pix[0..3]: pixels of src0 whose values are SUMMED (only DGRAD_REFLECT uses more than one: the reflection
//            halo folds gradient rows -1/H and cols -1/W back onto rows 1/H-2, cols 1/W-2); -1 = absent.
// pix1:      pixel of src1 (the full-resolution skip tensor of the UP2 concat); -1 = absent.
__device__ __forceinline__ void fp_gather_tap(const FpGeom& g, int n, int oy, int ox, int ky, int kx, int pix[4],
                                              int& pix1) {
  pix[0] = pix[1] = pix[2] = pix[3] = -1;
  pix1 = -1;
  switch (g.gather) {
    case FP_GATHER_FWD_ZERO: {
      const int iy = oy * g.stride + ky - g.pad, ix = ox * g.stride + kx - g.pad;
      if (iy >= 0 && iy < g.IH && ix >= 0 && ix < g.IW) pix[0] = (n * g.IH + iy) * g.IW + ix;
      break;
    }
    case FP_GATHER_FWD_REFLECT: {
      const int iy = fp_reflect(oy + ky - 1, g.IH), ix = fp_reflect(ox + kx - 1, g.IW);
      pix[0] = (n * g.IH + iy) * g.IW + ix;
      break;
    }
    case FP_GATHER_FWD_REFLECT_UP2: {
      const int iy = fp_reflect(oy + ky - 1, g.IH), ix = fp_reflect(ox + kx - 1, g.IW);
      pix[0] = (n * (g.IH >> 1) + (iy >> 1)) * (g.IW >> 1) + (ix >> 1);  // nearest x2: src = dst // 2
      pix1 = (n * g.IH + iy) * g.IW + ix;
      break;
    }
    case FP_GATHER_DGRAD_ZERO: {
      const int ry = oy + g.pad - ky, rx = ox + g.pad - kx;
      if (ry >= 0 && rx >= 0) {
        const int sy = ry / g.stride, sx = rx / g.stride;
        if (sy * g.stride == ry && sx * g.stride == rx && sy < g.IH && sx < g.IW) pix[0] = (n * g.IH + sy) * g.IW + sx;
      }
      break;
    }
    case FP_GATHER_DGRAD_REFLECT: {
      const int r0 = oy + 1 - ky, c0 = ox + 1 - kx;
      const bool r0v = r0 >= 0 && r0 < g.IH, c0v = c0 >= 0 && c0 < g.IW;
      // virtual padded row -1 mirrors onto row 1 (tap ky=0 reads dZ row 0); row H mirrors onto H-2 (ky=2 reads row H-1)
      const int er = (ky == 0 && oy == 1) ? 0 : ((ky == 2 && oy == g.OH - 2) ? g.IH - 1 : -1);
      const int ec = (kx == 0 && ox == 1) ? 0 : ((kx == 2 && ox == g.OW - 2) ? g.IW - 1 : -1);
      const int nb = n * g.IH;
      if (r0v && c0v) pix[0] = (nb + r0) * g.IW + c0;
      if (er >= 0 && c0v) pix[1] = (nb + er) * g.IW + c0;
      if (r0v && ec >= 0) pix[2] = (nb + r0) * g.IW + ec;
      if (er >= 0 && ec >= 0) pix[3] = (nb + er) * g.IW + ec;
      break;
    }
    default: break;
  }
}

// 4 consecutive K-channels [c4, c4+4) of one tap for one pixel (C0, C1 multiples of 4 => never straddles).
__device__ __forceinline__ float4 fp_gather_load4(const FpGeom& g, const float* __restrict__ src0,
                                                  const float* __restrict__ src1, const int pix[4], int pix1, int c4) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c4 < g.C0) {
    if (pix[0] >= 0) v = *reinterpret_cast<const float4*>(src0 + (size_t)pix[0] * g.C0 + c4);
    if ((pix[1] & pix[2] & pix[3]) >= 0) {  // any extra present (entries are -1 or >= 0)
#pragma unroll
      for (int j = 1; j < 4; ++j)
        if (pix[j] >= 0) {
          const float4 u = *reinterpret_cast<const float4*>(src0 + (size_t)pix[j] * g.C0 + c4);
          v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
    }
  } else if (c4 < g.C0 + g.C1) {
    if (pix1 >= 0) v = *reinterpret_cast<const float4*>(src1 + (size_t)pix1 * g.C1 + (c4 - g.C0));
  }
  return v;
}

// Stem: K index kk = (ky*7 + kx)*3 + ci over the NCHW image, (x - 0.45)/0.225 applied before zero padding.
__device__ __forceinline__ float fp_stem_load(const FpGeom& g, const float* __restrict__ img, int n, int oy, int ox,
                                              int kk) {
  if (kk >= 147) return 0.f;
  const int ky = kk / 21, rem = kk - ky * 21, kx = rem / 3, ci = rem - kx * 3;
  const int iy = oy * 2 + ky - 3, ix = ox * 2 + kx - 3;
  if (iy < 0 || iy >= g.IH || ix < 0 || ix >= g.IW) return 0.f;
  const float x = img[((size_t)(n * 3 + ci) * g.IH + iy) * g.IW + ix];
  return (x - 0.45f) / 0.225f;
}

// ---- ELU (alpha = 1) for the conv epilogues ---------------------------------------------------------------
// expm1f (ocml) is ~40 instructions and made the epilogue 10-20 % of the bf16x3 conv kernels.  For -0.25 < v <= 0 a degree-7
// Taylor polynomial (relative error < 2e-9); below that e^v - 1 has no cancellation and the hardware exponential is accurate to
// ~1e-7 absolute (argument rounding + 1 ulp), i.e. < 5e-7 of the result.  Same value is stored and later used for ELU'.
// Round 3 A/B: the five-instruction form e = exp2(v * log2 e) - 1 (absolute error ~1e-7 near zero; -DFP_ELU_EXP) leaves the training
// step where it was (13.73 / 13.78 vs 13.77 / 13.77 ms): the epilogue's ELU is not what the tile kernel waits for in the step, so the
// accurate form stays.
__device__ __forceinline__ float fp_elu(float v) {
#ifdef FP_ELU_EXP
  const float e = __builtin_amdgcn_exp2f(v * 1.44269504088896341f) - 1.f;
  return v > 0.f ? v : e;
#else
  // branch-free: with the early return `if (v > 0) return v` every element of an epilogue got its own exec-masked region (two
  // s_and_saveexec / s_cbranch pairs, ~25 instructions per element of which 9 scalar); both candidates and two selects are 15 VALU
  // and the same values (round 4)
  const float p = v * (1.f + v * (0.5f + v * (1.f / 6.f + v * (1.f / 24.f + v * (1.f / 120.f + v * (1.f / 720.f + v * (1.f / 5040.f)))))));
  const float e = __expf(v) - 1.f;
  const float neg = v > -0.25f ? p : e;
  return v > 0.f ? v : neg;
#endif
}

// ---- fp16-pair ("hp") operands ----------------------------------------------------------------------------
// x * 2^k = h + m with h = fp16(x * 2^k), m = fp16(x * 2^k - h): 11 + 11 significant bits (relative error <= 2^-22, ~2^-23.5 rms);
// every fp16 x fp16 product is exact in the fp32 accumulator of v_mfma_f32_32x32x16_f16 (FP_HP_PRODUCTS below says which are formed).  k is a per-tensor power of two
// taken from the tensor's largest magnitude (an "amax slot": FP_AMAX_SLOTS uint32 holding float bit patterns of |x|, combined
// with max; a producer on XCD k publishes into the sub-slots of XCD k, see fp_amax_publish), which maps that
// magnitude to [2^target, 2^(target+1)): no overflow (fp16 max 65504), and anything above 2^-24 / 2^k is still represented, i.e.
// the absolute error floor is 2^-37 of the tensor's largest element.  Scaling by powers of two is exact and undone in the epilogue.
static_assert(FP_AMAX_SLOTS == 16, "amax slot width (include/footprints_hip.h)");
// products per multiply-add: hh + hm + mh; the fourth, mm, is <= 2^-22 of a product (~2^-24 rms, random sign): measured relative L2
// error against float64 2.733e-7 with three products, 2.722e-7 with four, 2.6e-7 for an fp32 convolution -- and a quarter of the MFMA
// work.  -DFP_HP_PRODUCTS=4 builds the four-product form for A/B runs.
#ifndef FP_HP_PRODUCTS
#define FP_HP_PRODUCTS 3
#endif
#ifndef FP_HP_TARGET_ACT_V
#define FP_HP_TARGET_ACT_V 12
#endif
constexpr int FP_HP_TARGET_ACT = FP_HP_TARGET_ACT_V;    // activations / gradients: amax -> [2^12, 2^13)
constexpr int FP_HP_TARGET_W = 11;      // weights: amax -> [2^11, 2^12) (the nearest-x2 phase kernels add up to four of them)
__device__ __forceinline__ unsigned fp_amax_bits(const unsigned* __restrict__ slot) {
  unsigned m = 0;
#pragma unroll
  for (int i = 0; i < FP_AMAX_SLOTS; ++i) m = max(m, slot[i * FP_AMAX_STRIDE]);
  return m;
}
// Up to three slots in ONE memory round trip, issued early and reduced late (round 4).  fp_amax_bits reads a slot's sixteen sub-slots
// through the scalar cache, eight at a time: a kernel that needs the source's, the skip tensor's and the weights' amax waited for four
// DEPENDENT scalar round trips (2-3 us) before it issued its first operand load -- in front of every one of the ~250 split-operand launches
// of a step, most of them 25-45 us long and alone on the GPU.  Here lanes 0-15 / 16-31 / 32-47 of the calling wave load one sub-slot each
// with ONE vector load (`fp_amax3_issue`, result not waited for); the kernel sets up its addresses and issues its first operand loads; then
// `fp_amax3_reduce` folds the sixteen lanes of each group by DPP row shifts and broadcasts the three maxima.  A null slot reads as 0.
__device__ __forceinline__ unsigned fp_amax3_issue(const unsigned* __restrict__ s0, const unsigned* __restrict__ s1,
                                                   const unsigned* __restrict__ s2) {
  const int ln = (int)(threadIdx.x & 63), grp = ln >> 4;
  const unsigned* sp = grp == 0 ? s0 : (grp == 1 ? s1 : (grp == 2 ? s2 : nullptr));
  return sp ? sp[(ln & 15) * FP_AMAX_STRIDE] : 0u;
}
__device__ __forceinline__ void fp_amax3_reduce(unsigned raw, unsigned& m0, unsigned& m1, unsigned& m2) {
  int v = (int)raw;                                  // bit patterns of non-negative floats: signed and unsigned order agree
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true));      // row_shr:1 (zero fill at the row start)
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true));      // row_shr:2
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true));      // row_shr:4
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true));      // row_shr:8 -> lane 15 of each row holds the row's maximum
  m0 = (unsigned)__builtin_amdgcn_readlane(v, 15);
  m1 = (unsigned)__builtin_amdgcn_readlane(v, 31);
  m2 = (unsigned)__builtin_amdgcn_readlane(v, 47);
}
// exponent k with amax * 2^k in [2^target, 2^(target+1)); 0 for an all-zero tensor; inf / nan propagate through the data itself
__device__ __forceinline__ int fp_hp_exponent(unsigned amax_bits, int target) {
  // (clamped so that 2^k is a finite float: the staging code multiplies by it; tensors whose largest element is below 2^-114 lose
  // low-order bits instead -- they are beyond any gradient this network produces)
  return amax_bits ? min(target - ((int)(amax_bits >> 23) - 127), 126) : 0;
}
// x * s = h + m (s = 2^k, wave-uniform): the two fp16 planes of four floats -- scale, convert (packed), convert back, subtract, convert
// (packed): 4 VALU per element.  Round 4 measured the two-instruction form (v_fma_mixlo/hi_f16: fp16(x * s), then fp16(x * s - h) with the
// fp16 term read in place; bit-identical, -DFP_HP_SPLIT_MIX) SLOWER in the training step: 12.39 vs 12.17 ms, alternating runs, and
// 97.2 vs 95.9 us on the 64 -> 64 @ 96 x 320 tile convolution -- the mix instructions are VOP3P-encoded like the MFMAs they sit between,
// and half as many of them cost more than the plain VALU sequence (profiles/round4_notes.md; MI355X_MICROARCH.md prices packed-fp32 VALU
// beside MFMAs the same way).
typedef _Float16 fp_f16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void fp_hp_split4(float x0, float x1, float x2, float x3, float s, uint2& hq, uint2& mq) {
  fp_f16x4 h, m;
#ifndef FP_HP_SPLIT_MIX
  typedef float fp_f32x4 __attribute__((ext_vector_type(4)));
  const fp_f32x4 v = {x0 * s, x1 * s, x2 * s, x3 * s};
  h = __builtin_convertvector(v, fp_f16x4);
  const fp_f32x4 r1 = v - __builtin_convertvector(h, fp_f32x4);
  m = __builtin_convertvector(r1, fp_f16x4);
#elif FP_HP_SPLIT_MIX == 2      // A/B: only the residual through v_fma_mix_f32 (fp32 result, the fp16 term read in place): 3 VALU per element
  typedef float fp_f32x4 __attribute__((ext_vector_type(4)));
  const fp_f32x4 v = {x0 * s, x1 * s, x2 * s, x3 * s};
  h = __builtin_convertvector(v, fp_f16x4);
  const fp_f32x4 r1 = {__builtin_fmaf((float)h.x, -1.f, v.x), __builtin_fmaf((float)h.y, -1.f, v.y), __builtin_fmaf((float)h.z, -1.f, v.z),
                       __builtin_fmaf((float)h.w, -1.f, v.w)};
  m = __builtin_convertvector(r1, fp_f16x4);
#else
  h.x = (_Float16)__builtin_fmaf(x0, s, 0.f); h.y = (_Float16)__builtin_fmaf(x1, s, 0.f);
  h.z = (_Float16)__builtin_fmaf(x2, s, 0.f); h.w = (_Float16)__builtin_fmaf(x3, s, 0.f);
  m.x = (_Float16)__builtin_fmaf(x0, s, -(float)h.x); m.y = (_Float16)__builtin_fmaf(x1, s, -(float)h.y);
  m.z = (_Float16)__builtin_fmaf(x2, s, -(float)h.z); m.w = (_Float16)__builtin_fmaf(x3, s, -(float)h.w);
#endif
  hq = __builtin_bit_cast(uint2, h);
  mq = __builtin_bit_cast(uint2, m);
}
// publish a magnitude into an amax slot.  Agent-scope atomics execute at the memory side and same-address ones serialise at ~90 ns
// each (measured: 11 520 publishing waves = +64 us; 1 024 workgroups of an element-wise kernel finishing together = +20 us), so a
// publication is a WORKGROUP-scope atomic -- executed in the issuing XCD's L2 -- on a sub-slot that only this XCD touches (index from
// the XCC id, one 128-byte line per sub-slot: no line is ever dirty in two L2s).  The L2 writes the line back at the end of the kernel
// like any other store, and the consumer -- a later kernel -- combines the sub-slots with max.
__device__ __forceinline__ unsigned fp_xcc_id() {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  return x & 7u;
}
__device__ __forceinline__ void fp_amax_publish(unsigned* slot, unsigned id, float m) {
  unsigned* s = slot + ((fp_xcc_id() * 2 + (id & 1u)) % FP_AMAX_SLOTS) * FP_AMAX_STRIDE;
  const unsigned bits = __float_as_uint(m);
  if (bits) __hip_atomic_fetch_max(s, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// ---- Welford / Chan triples (BatchNorm statistics: bn_pool.hip, and the tile kernel's epilogue partials) -------------------------------
struct FpWf {  // count, mean, sum of squared deviations
  float n, mean, m2;
};
__device__ __forceinline__ void fp_wf_add(FpWf& a, float x, float n, float rn) {  // n = new count, rn = 1/n
  a.n = n;
  const float d = x - a.mean;
  a.mean += d * rn;
  a.m2 += d * (x - a.mean);
}
__device__ __forceinline__ void fp_wf_merge(FpWf& a, const FpWf& b) {
  const float n = a.n + b.n;
  if (n == 0.f) return;
  const float d = b.mean - a.mean;
  const float f = b.n / n;
  a.mean += d * f;
  a.m2 += b.m2 + d * d * a.n * f;
  a.n = n;
}
// BatchNorm side output of a convolution launch (include/footprints_hip.h, fp_aux.bn_*)
struct FpBnSink {
  float* part;
  int64_t cap_floats;
  int32_t* nblk_out;
  const float* z;        // backward form (fp_aux.bnb_*): the BatchNorm's input and its saved statistics; null = forward statistics
  const float* mean;
  const float* invstd;
};
// round 6: side outputs arrive as an explicit `const fp_aux*` argument (include/footprints_hip.h); these two read it (null = none)
static inline FpBnSink fp_bn_sink_of(const fp_aux* aux) {
  if (!aux || !aux->bn_part) {
    if (aux && aux->bn_nblk_out) *aux->bn_nblk_out = 0;
    return FpBnSink{nullptr, 0, nullptr, nullptr, nullptr, nullptr};
  }
  if (aux->bn_nblk_out) *aux->bn_nblk_out = 0;
  return FpBnSink{aux->bn_part, aux->bn_capacity_floats, aux->bn_nblk_out, aux->bnb_z, aux->bnb_mean, aux->bnb_invstd};
}
static inline unsigned* fp_amax_out_of(const fp_aux* aux) { return aux ? aux->amax_out : nullptr; }
__device__ __forceinline__ float fp_wave_max(float v);
// one candidate per workgroup: every thread of the block calls this once (wave maxima through 64 bytes of shared scratch)
__device__ __forceinline__ void fp_amax_publish_block(unsigned* slot, float m) {
  __shared__ float fp_amax_scratch[16];
  m = fp_wave_max(m);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) fp_amax_scratch[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (unsigned i = 1; i < (blockDim.x + 63) / 64; ++i) m = fmaxf(m, fp_amax_scratch[i]);
    fp_amax_publish(slot, blockIdx.x, m);
  }
}
__device__ __forceinline__ float fp_amax4(float m, const float4& v) {
  return fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
}
__device__ __forceinline__ float fp_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- wave / block reductions (wave = 64) ---------------------------------------------------------------
__device__ __forceinline__ float fp_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// Issue-order hint for a block of NM MFMAs whose operands were read earlier plus NDS LDS reads and NVM global reads for later
// blocks: one read between consecutive MFMAs, so the wave's memory instructions issue while the matrix pipe is busy instead of in
// a burst ahead of it (measured 4-9 % on the bf16x3 tile kernel).  Call right after the MFMA block; the region must start with
// __builtin_amdgcn_sched_barrier(0).
template <int NDS, int NVM, int NM>
__device__ __forceinline__ void fp_sched_interleave() {
#pragma unroll
  for (int i = 0; i < (NDS < NM ? NDS : NM); ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
  }
#pragma unroll
  for (int i = 0; i < (NVM < NM - NDS ? NVM : (NM - NDS > 0 ? NM - NDS : 0)); ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
  }
  if (NM - NDS - NVM > 0) __builtin_amdgcn_sched_group_barrier(0x008, NM - NDS - NVM > 0 ? NM - NDS - NVM : 1, 0);
}
