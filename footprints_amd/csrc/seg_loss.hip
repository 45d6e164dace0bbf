// Loss of the ground-segmentation trainer on the device (round 3; VERDICT r2 "missing" 6): the four logit maps of
// `Segmentor.forward` ([B,1,H/8,W/8] .. [B,1,H,W]) are up-sized bilinearly (align_corners = False) to the input resolution, each gets a
// per-image masked BCE-with-logits mean  sum(bce(p, ground) * labelled) / (sum(labelled) + 1e-7),  the four are averaged, then the batch
// (footprints/preprocessing/segmentation/train.py:184-193, evaluation.py:39-58).  Forward AND the gradient with respect to the four
// low-resolution maps in three launches, no up-sized tensor, no autograd graph:
//   seg_valid_kernel   labelled pixels per image (two stages, fixed order)
//   seg_loss_kernel    every full-resolution pixel samples its four logits, accumulates the four masked BCE sums per image and writes
//                      the four gradient planes  labelled * (sigmoid(p) - ground) / ((valid_b + 1e-7) * 4 B)
//   seg_down_kernel    d loss / d low-resolution logit = the transpose of the bilinear up-sizing, in gather form (deterministic)
// + seg_final_kernel for the per-image / per-scale bookkeeping the reference's Evaluator tracks.  HBM-bound, latency-sized problems.
#include "fp_common.h"

namespace {

constexpr int SEG_BLK = 64;     // partial-sum blocks per image

// torch's bilinear source index for align_corners = False: src = (dst + 0.5) * (in / out) - 0.5, clamped below at 0
__device__ __forceinline__ void seg_src(int dst, float ratio, int n_in, int& i0, int& i1, float& w1) {
  float s = ((float)dst + 0.5f) * ratio - 0.5f;
  s = s < 0.f ? 0.f : s;
  i0 = (int)s;
  i0 = min(i0, n_in - 1);
  i1 = min(i0 + 1, n_in - 1);
  w1 = s - (float)i0;
}

struct SegMaps {
  const float* p[4];      // logit maps, element (b, y, x) at p[s][b * bstride[s] + y * w[s] + x]
  float* g[4];            // gradient maps, same addressing
  int h[4], w[4];
  long long bstride[4];
};

__global__ void __launch_bounds__(256) seg_valid_kernel(const float* __restrict__ lmask, int HW, float* __restrict__ vpart) {
  const int b = blockIdx.y, blk = blockIdx.x;
  float s = 0.f;
  for (int i = blk * 256 + threadIdx.x; i < HW; i += SEG_BLK * 256) s += lmask[(size_t)b * HW + i];
  s = fp_wave_sum(s);
  __shared__ float sm[4];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) vpart[b * SEG_BLK + blk] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

__device__ __forceinline__ float seg_valid_of(const float* __restrict__ vpart, int b) {
  float v = 0.f;
  for (int k = 0; k < SEG_BLK; ++k) v += vpart[b * SEG_BLK + k];      // fixed order, every reader gets the same bits
  return v;
}

__global__ void __launch_bounds__(256) seg_loss_kernel(const SegMaps m, const float* __restrict__ gmask, const float* __restrict__ lmask, int B, int H,
                                                       int W, const float* __restrict__ vpart, float* __restrict__ lpart, float* __restrict__ gfull) {
  const int b = blockIdx.y, blk = blockIdx.x, HW = H * W;
  const float valid = seg_valid_of(vpart, b);
  const float gs = 1.f / ((valid + 1e-7f) * 4.f * (float)B);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int i = blk * 256 + threadIdx.x; i < HW; i += SEG_BLK * 256) {
    const int y = i / W, x = i - y * W;
    const float gt = gmask[(size_t)b * HW + i], lm = lmask[(size_t)b * HW + i];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      int y0, y1, x0, x1;
      float wy, wx;
      seg_src(y, (float)m.h[s] / (float)H, m.h[s], y0, y1, wy);
      seg_src(x, (float)m.w[s] / (float)W, m.w[s], x0, x1, wx);
      const float* p = m.p[s] + (size_t)b * m.bstride[s];
      const float v00 = p[y0 * m.w[s] + x0], v01 = p[y0 * m.w[s] + x1], v10 = p[y1 * m.w[s] + x0], v11 = p[y1 * m.w[s] + x1];
      // ATen's upsample_bilinear2d: w0 * (w0x * v00 + w1x * v01) + w1 * (w0x * v10 + w1x * v11)
      const float v = (1.f - wy) * ((1.f - wx) * v00 + wx * v01) + wy * ((1.f - wx) * v10 + wx * v11);
      const float bce = fmaxf(v, 0.f) - v * gt + log1pf(expf(-fabsf(v)));
      acc[s] += bce * lm;
      const float sg = 1.f / (1.f + expf(-v));
      gfull[((size_t)s * B + b) * HW + i] = lm * (sg - gt) * gs;
    }
  }
  __shared__ float sm[4][4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const float r = fp_wave_sum(acc[s]);
    if ((threadIdx.x & 63) == 0) sm[s][threadIdx.x >> 6] = r;
  }
  __syncthreads();
  if (threadIdx.x < 4) lpart[(b * SEG_BLK + blk) * 4 + threadIdx.x] = (sm[threadIdx.x][0] + sm[threadIdx.x][1]) + (sm[threadIdx.x][2] + sm[threadIdx.x][3]);
}

// losses[0 .. 4B): per image the four per-scale means and their average (the Evaluator's tracked values, scale-major then "loss");
// losses[5B] = batch mean of the average (the value the trainer back-propagates)
__global__ void __launch_bounds__(64) seg_final_kernel(const float* __restrict__ vpart, const float* __restrict__ lpart, int B, float* __restrict__ losses) {
  float tot = 0.f;
  for (int b = 0; b < B; ++b) {
    const float valid = seg_valid_of(vpart, b);
    float sum4 = 0.f;
    for (int s = 0; s < 4; ++s) {
      float a = 0.f;
      for (int k = 0; k < SEG_BLK; ++k) a += lpart[(b * SEG_BLK + k) * 4 + s];
      const float l = a / (valid + 1e-7f);
      if (threadIdx.x == 0) losses[s * B + b] = l;
      sum4 += l;
    }
    if (threadIdx.x == 0) losses[4 * B + b] = sum4 / 4.f;
    tot += sum4 / 4.f;
  }
  if (threadIdx.x == 0) losses[5 * B] = tot / (float)B;
}

// gradient of the up-sizing, gather form: low-resolution pixel (ly, lx) collects w * g from every full-resolution pixel whose bilinear
// sample touches it -- rows [S (ly - 1), S (ly + 2)) for an integer scale S, found by re-deriving each candidate's own source indices
__global__ void __launch_bounds__(256) seg_down_kernel(const SegMaps m, int s, int B, int H, int W, const float* __restrict__ gfull) {
  const int h = m.h[s], w = m.w[s];
  const int total = B * h * w;
  const float ry = (float)h / (float)H, rx = (float)w / (float)W;
  const int Sy = (H + h - 1) / h, Sx = (W + w - 1) / w;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    const int lx = e % w, r = e / w, ly = r % h, b = r / h;
    const float* g = gfull + ((size_t)s * B + b) * H * W;
    float acc = 0.f;
    const int ya = max(0, Sy * (ly - 1)), yb = min(H, Sy * (ly + 2));
    const int xa = max(0, Sx * (lx - 1)), xb = min(W, Sx * (lx + 2));
    for (int y = ya; y < yb; ++y) {
      int y0, y1;
      float wy;
      seg_src(y, ry, h, y0, y1, wy);
      const float cy = (y0 == ly ? 1.f - wy : 0.f) + (y1 == ly ? wy : 0.f);
      if (cy == 0.f) continue;
      float row = 0.f;
      for (int x = xa; x < xb; ++x) {
        int x0, x1;
        float wx;
        seg_src(x, rx, w, x0, x1, wx);
        const float cx = (x0 == lx ? 1.f - wx : 0.f) + (x1 == lx ? wx : 0.f);
        if (cx != 0.f) row += cx * g[y * W + x];
      }
      acc += cy * row;
    }
    m.g[s][(size_t)b * m.bstride[s] + ly * w + lx] = acc;
  }
}

}  // namespace

extern "C" int64_t fp_seg_loss_workspace(int32_t B, int32_t H, int32_t W) {
  return ((int64_t)4 * B * H * W + (int64_t)B * SEG_BLK * 5) * (int64_t)sizeof(float);
}

// preds[s]: logit map of scale s ([B][h_s][w_s] with batch stride bstride[s] elements: a channel slice of a wider tensor is fine);
// dpreds[s] (nullable as a whole): same addressing, receives d (batch-mean loss) / d preds[s]; losses: 5 B + 1 floats (see seg_final_kernel)
extern "C" int fp_seg_loss_fwd_bwd(const float* const* preds, float* const* dpreds, const int32_t* hs, const int32_t* ws, const int64_t* bstrides,
                                   const float* ground_mask, const float* loss_mask, int32_t B, int32_t H, int32_t W, float* losses,
                                   void* workspace, int64_t workspace_bytes, fp_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  FP_REQUIRE(preds && hs && ws && bstrides && ground_mask && loss_mask && losses && workspace, "fp_seg_loss_fwd_bwd: null pointer");
  FP_REQUIRE(B > 0 && H > 0 && W > 0, "fp_seg_loss_fwd_bwd: empty problem");
  FP_REQUIRE(workspace_bytes >= fp_seg_loss_workspace(B, H, W), "fp_seg_loss_fwd_bwd: workspace too small");
  SegMaps m;
  for (int s = 0; s < 4; ++s) {
    FP_REQUIRE(preds[s] && hs[s] > 0 && ws[s] > 0 && hs[s] <= H && ws[s] <= W && bstrides[s] >= (int64_t)hs[s] * ws[s],
               "fp_seg_loss_fwd_bwd: bad map %d", s);
    // the gradient's gather window [S (l - 1), S (l + 2)) with S = H / h covers every contributing full-resolution pixel only for integer
    // scales (for H = 100, h = 40 rows would be missed silently); the segmentation decoder's maps are H / 8 ... H / 1
    FP_REQUIRE(!dpreds || (H % hs[s] == 0 && W % ws[s] == 0), "fp_seg_loss_fwd_bwd: map %d (%d x %d) is not an integer fraction of %d x %d",
               s, hs[s], ws[s], H, W);
    m.p[s] = preds[s]; m.g[s] = dpreds ? dpreds[s] : nullptr; m.h[s] = hs[s]; m.w[s] = ws[s]; m.bstride[s] = bstrides[s];
    FP_REQUIRE(!dpreds || dpreds[s], "fp_seg_loss_fwd_bwd: gradient map %d missing", s);
  }
  float* gfull = (float*)workspace;
  float* vpart = gfull + (size_t)4 * B * H * W;
  float* lpart = vpart + (size_t)B * SEG_BLK;
  fp_launch(seg_valid_kernel, dim3(SEG_BLK, B), dim3(256), 0, stream, loss_mask, H * W, vpart);
  fp_launch(seg_loss_kernel, dim3(SEG_BLK, B), dim3(256), 0, stream, m, ground_mask, loss_mask, (int)B, (int)H, (int)W, (const float*)vpart, lpart, gfull);
  fp_launch(seg_final_kernel, dim3(1), dim3(64), 0, stream, (const float*)vpart, (const float*)lpart, (int)B, losses);
  if (dpreds)
    for (int s = 0; s < 4; ++s) {
      const int total = B * hs[s] * ws[s];
      fp_launch(seg_down_kernel, dim3((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256), dim3(256), 0, stream, m, s, (int)B, (int)H, (int)W,
                (const float*)gfull);
    }
  return fp_check_launch("fp_seg_loss_fwd_bwd");
}
