// Test-set metrics (reference footprints/evaluation/evaluate_model.py:50-99, 160-177) as per-image device reductions:
// integer confusion counts for the footprint / free-space masks and float64 sums for the hidden-depth errors.  One 1024-thread
// workgroup per image, fixed-order shared-memory tree => deterministic; the host turns counts / sums into IoU, F1, a1, rmse ...
// exactly like the reference does with numpy scalars.  Predictions may be the float16 arrays the inference pass saves
// (datasets/inference_dataset.py:35-38): numpy then evaluates `1 - pred` and sigmoid_to_depth in float16, and so does this file
// (every float16 operation = the float32 operation on the two halves rounded once to float16).
#include <hip/hip_fp16.h>

#include "fp_common.h"

namespace {

constexpr int EVT = 1024;

__device__ __forceinline__ float rh(float x) { return __half2float(__float2half_rn(x)); }   // round to float16 and back

template <bool HALF>
__device__ __forceinline__ float load_pred(const void* p, size_t i) {
  return HALF ? __half2float(reinterpret_cast<const __half*>(p)[i]) : reinterpret_cast<const float*>(p)[i];
}

// counts[b] = {n_true, tp, fp, fn} over the pixels of image b (inside `region` when given)
template <bool HALF>
__global__ void __launch_bounds__(EVT) eval_mask_kernel(const void* __restrict__ pred, const float* __restrict__ gt,
                                                        const unsigned char* __restrict__ region, int invert, long long pixels,
                                                        long long pred_stride, long long* __restrict__ counts) {
  __shared__ long long sm[4][EVT];
  const int b = blockIdx.x;
  const float* g = gt + (size_t)b * pixels;
  const unsigned char* rg = region ? region + (size_t)b * pixels : nullptr;
  long long c[4] = {0, 0, 0, 0};
  for (long long i = threadIdx.x; i < pixels; i += EVT) {
    if (rg && !rg[i]) continue;
    const float gv = g[i], pv = load_pred<HALF>(pred, (size_t)b * pred_stride + i);
    bool t, p;
    if (invert) {                       // evaluate_mask(1 - ground_truth[free_space], 1 - pred[free_space])
      t = __fsub_rn(1.f, gv) > 0.1f;
      const float q = __fsub_rn(1.f, pv);
      p = (HALF ? rh(q) : q) > 0.5f;
    } else {
      t = gv > 0.1f;
      p = pv > 0.5f;
    }
    c[0] += t; c[1] += t && p; c[2] += !t && p; c[3] += t && !p;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) sm[k][threadIdx.x] = c[k];
  __syncthreads();
  for (int s = EVT / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
#pragma unroll
      for (int k = 0; k < 4; ++k) sm[k][threadIdx.x] += sm[k][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x < 4) counts[(size_t)b * 4 + threadIdx.x] = sm[threadIdx.x][0];
}

// sums[b] = {n, n(thresh < 1.25), sum (gt-pred)^2, sum |gt-pred|/gt, sum (gt-pred)^2/gt} over the pixels with gt > 0
template <bool HALF>
__global__ void __launch_bounds__(EVT) eval_depth_kernel(const void* __restrict__ disp, const float* __restrict__ gt, long long pixels,
                                                         long long pred_stride, float min_disp, float disp_range, float clip_min,
                                                         float clip_max, double* __restrict__ sums) {
  __shared__ double sm[5][EVT];
  const int b = blockIdx.x;
  const float* g = gt + (size_t)b * pixels;
  double c[5] = {0, 0, 0, 0, 0};
  const float mdh = HALF ? rh(min_disp) : min_disp, drh = HALF ? rh(disp_range) : disp_range;
  for (long long i = threadIdx.x; i < pixels; i += EVT) {
    const float graw = g[i];
    if (!(graw > 0.f)) continue;
    const float d = load_pred<HALF>(disp, (size_t)b * pred_stride + i);
    // sigmoid_to_depth: 1 / (min_disp + (max_disp - min_disp) * disp), each operation rounded in the array's own dtype
    float scaled = __fmul_rn(drh, d);
    if (HALF) scaled = rh(scaled);
    scaled = __fadd_rn(mdh, scaled);
    if (HALF) scaled = rh(scaled);
    float p = __fdiv_rn(1.f, scaled);
    if (HALF) p = rh(p);
    p = fminf(fmaxf(p, clip_min), clip_max);
    const float gv = fminf(fmaxf(graw, clip_min), clip_max);
    const float thresh = fmaxf(__fdiv_rn(gv, p), __fdiv_rn(p, gv));
    const float diff = __fsub_rn(gv, p);
    const float sq = __fmul_rn(diff, diff);
    c[0] += 1.0;
    c[1] += thresh < 1.25f ? 1.0 : 0.0;
    c[2] += (double)sq;
    c[3] += (double)__fdiv_rn(fabsf(diff), gv);
    c[4] += (double)__fdiv_rn(sq, gv);
  }
#pragma unroll
  for (int k = 0; k < 5; ++k) sm[k][threadIdx.x] = c[k];
  __syncthreads();
  for (int s = EVT / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
#pragma unroll
      for (int k = 0; k < 5; ++k) sm[k][threadIdx.x] += sm[k][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x < 5) sums[(size_t)b * 5 + threadIdx.x] = sm[threadIdx.x][0];
}

}  // namespace

extern "C" int fp_eval_mask_counts(const void* pred, int32_t pred_is_half, const float* gt, const uint8_t* region, int32_t invert,
                                   int32_t B, int64_t pixels, int64_t pred_stride, int64_t* counts, fp_stream_t stream) {
  FP_REQUIRE(pred && gt && counts && B > 0 && pixels > 0 && pred_stride >= pixels, "fp_eval_mask_counts: bad arguments");
  if (pred_is_half)
    fp_launch(eval_mask_kernel<true>, dim3(B), dim3(EVT), 0, (hipStream_t)stream, pred, gt, region, invert, (long long)pixels,
                       (long long)pred_stride, (long long*)counts);
  else
    fp_launch(eval_mask_kernel<false>, dim3(B), dim3(EVT), 0, (hipStream_t)stream, pred, gt, region, invert, (long long)pixels,
                       (long long)pred_stride, (long long*)counts);
  return fp_check_launch("fp_eval_mask_counts");
}

extern "C" int fp_eval_depth_sums(const void* pred_disp, int32_t pred_is_half, const float* gt, int32_t B, int64_t pixels,
                                  int64_t pred_stride, double min_depth, double max_depth, double clip_min, double clip_max, double* sums,
                                  fp_stream_t stream) {
  FP_REQUIRE(pred_disp && gt && sums && B > 0 && pixels > 0 && pred_stride >= pixels && min_depth > 0 && max_depth > min_depth,
             "fp_eval_depth_sums: bad arguments");
  const double min_disp = 1.0 / max_depth, max_disp = 1.0 / min_depth;       // python floats in the reference
  const float md = (float)min_disp, dr = (float)(max_disp - min_disp);
  if (pred_is_half)
    fp_launch(eval_depth_kernel<true>, dim3(B), dim3(EVT), 0, (hipStream_t)stream, pred_disp, gt, (long long)pixels,
                       (long long)pred_stride, md, dr, (float)clip_min, (float)clip_max, sums);
  else
    fp_launch(eval_depth_kernel<false>, dim3(B), dim3(EVT), 0, (hipStream_t)stream, pred_disp, gt, (long long)pixels,
                       (long long)pred_stride, md, dr, (float)clip_min, (float)clip_max, sums);
  return fp_check_launch("fp_eval_depth_sums");
}
