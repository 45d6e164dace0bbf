// Fused multi-head loss (forward + backward in one pass), fused Adam, weight packing, column sums, the backward
// of the virtual nearest-x2 + concat, and layout transforms.  All HBM-bound streaming kernels: one pass over the
// data, float4 where the layout allows, fixed-order reductions (no atomics) so reruns are bit-stable.
#include <hip/hip_fp16.h>

#include "fp_common.h"

namespace {

int ew_grid(size_t total, int cap = 8192) {
  size_t g = (total + 255) / 256;
  return (int)(g > (size_t)cap ? cap : (g < 1 ? 1 : g));
}

// ------------------------------------------------------------------------------------------------------------
// Loss: reference footprints/training/losses.py:31-152 (+ utils.py:36-42), closed form per pixel.
// ------------------------------------------------------------------------------------------------------------
struct LossArgs {
  const float* pred[4];
  float* dpred[4];
  const float *vg, *ag, *depth, *gdepth, *mov, *dm;
  float min_disp, disp_range, prior, gscale;
  int B, HW;
  float* part;
};

__device__ __forceinline__ float bce_logits(float x, float t) {  // BCEWithLogitsLoss(reduction='none')
  return fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
}
__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }

__global__ void __launch_bounds__(256) loss_kernel(const LossArgs a) {
  __shared__ float sm[4][16];
  const size_t npix = (size_t)a.B * a.HW;
  float sums[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) sums[i] = 0.f;
  for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < npix; p += (size_t)gridDim.x * 256) {
    const int b = (int)(p / a.HW);
    const size_t yx = p - (size_t)b * a.HW;
    const float vg = a.vg[p], ag = a.ag[p], depth = a.depth[p], gdepth = a.gdepth[p];
    const float keep = 1.f - a.mov[p];                         // losses.py:44 (mask is 1 if moving)
    const float m = (ag + a.dm[p]) > 0.f ? 1.f : 0.f;           // losses.py:138
    const float vd = depth > 0.f ? 1.f : 0.f, vgd = gdepth > 0.f ? 1.f : 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const size_t base = (size_t)b * 4 * a.HW + yx;
      const float o0 = a.pred[s][base], o1 = a.pred[s][base + a.HW], o2 = a.pred[s][base + 2 * (size_t)a.HW],
                  o3 = a.pred[s][base + 3 * (size_t)a.HW];
      // ch0 visible ground
      sums[s * 4 + 0] += bce_logits(o0, vg);
      // ch1 three-class loss
      const float w1 = m * keep, w0 = a.prior * (1.f - m);
      sums[s * 4 + 1] += bce_logits(o1, ag) * w1 + bce_logits(o1, 0.f) * w0;
      // ch2/ch3 log-L1 on depth = 1/(min_disp + range*sigmoid_disp)
      const float sc2 = __fadd_rn(a.min_disp, __fmul_rn(a.disp_range, o2));
      const float d2 = 1.f / sc2;
      const float e2 = d2 - depth;
      sums[s * 4 + 2] += logf(fabsf(e2) + 1.f) * vd;
      const float sc3 = __fadd_rn(a.min_disp, __fmul_rn(a.disp_range, o3));
      const float d3 = 1.f / sc3;
      const float e3 = d3 - gdepth;
      sums[s * 4 + 3] += logf(fabsf(e3) + 1.f) * vgd;
      if (a.dpred[0]) {
        const float s0 = sigm(o0), s1 = sigm(o1);
        const float sg2 = e2 > 0.f ? 1.f : (e2 < 0.f ? -1.f : 0.f), sg3 = e3 > 0.f ? 1.f : (e3 < 0.f ? -1.f : 0.f);
        a.dpred[s][base] = (s0 - vg) * a.gscale;
        a.dpred[s][base + a.HW] = ((s1 - ag) * w1 + s1 * w0) * a.gscale;
        a.dpred[s][base + 2 * (size_t)a.HW] = vd * sg2 / (fabsf(e2) + 1.f) * (-a.disp_range * d2 * d2) * a.gscale;
        a.dpred[s][base + 3 * (size_t)a.HW] = vgd * sg3 / (fabsf(e3) + 1.f) * (-a.disp_range * d3 * d3) * a.gscale;
      }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const float v = fp_wave_sum(sums[i]);
    if (lane == 0) sm[wave][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < 16) a.part[(size_t)blockIdx.x * 16 + threadIdx.x] = sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x];
}

__global__ void __launch_bounds__(1024) loss_final_kernel(const float* __restrict__ part, int nblk, double inv_n, float* __restrict__ out) {
  __shared__ float mean[16];
  {  // wave w reduces loss term w (16 waves), double accumulation, fixed order
    const int lane = threadIdx.x & 63, term = threadIdx.x >> 6;
    double s = 0.0;
    for (int b = lane; b < nblk; b += 64) s += (double)part[(size_t)b * 16 + term];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if (lane == 0) mean[term] = (float)(s * inv_n);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float total = 0.f;
    for (int s = 0; s < 4; ++s) {
      const float vg = mean[s * 4 + 0], ag = mean[s * 4 + 1], d = mean[s * 4 + 2], gd = mean[s * 4 + 3];
      out[s * 5 + 0] = vg; out[s * 5 + 1] = ag; out[s * 5 + 2] = d; out[s * 5 + 3] = gd;
      const float ls = ((d + vg) + ag) + gd;   // losses.py:80-83 order
      out[s * 5 + 4] = ls;
      total += ls;
    }
    out[20] = total / 4.f;                      // losses.py:87
  }
}

int loss_blocks(int64_t npix) {
  int64_t b = fp_ceil_div(npix, 256 * 4);
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (int)b;
}

// ------------------------------------------------------------------------------------------------------------
// Adam (torch.optim.Adam single-tensor math: lerp, addcmul, sqrt/bc2_sqrt + eps, addcdiv)
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, size_t n, float one_minus_b1, float b2,
                                                   float one_minus_b2, float step_size, float inv_bc2_sqrt, float eps,
                                                   float gscale) {
  const size_t n4 = n >> 2;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += (size_t)gridDim.x * 256) {
    float4 P = reinterpret_cast<float4*>(p)[e], G = reinterpret_cast<const float4*>(g)[e];
    float4 Mv = reinterpret_cast<float4*>(m)[e], V = reinterpret_cast<float4*>(v)[e];
    float* pp = &P.x; float* gg = &G.x; float* mm = &Mv.x; float* vv = &V.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gr = gg[j] * gscale;
      mm[j] = mm[j] + (gr - mm[j]) * one_minus_b1;
      vv[j] = vv[j] * b2 + one_minus_b2 * gr * gr;
      const float denom = sqrtf(vv[j]) * inv_bc2_sqrt + eps;
      pp[j] = pp[j] - step_size * (mm[j] / denom);
    }
    reinterpret_cast<float4*>(p)[e] = P;
    reinterpret_cast<float4*>(m)[e] = Mv;
    reinterpret_cast<float4*>(v)[e] = V;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const size_t e = (n4 << 2) + threadIdx.x;
    const float gr = g[e] * gscale;
    const float mn = m[e] + (gr - m[e]) * one_minus_b1;
    const float vn = v[e] * b2 + one_minus_b2 * gr * gr;
    m[e] = mn; v[e] = vn;
    p[e] = p[e] - step_size * (mn / (sqrtf(vn) * inv_bc2_sqrt + eps));
  }
}

// ------------------------------------------------------------------------------------------------------------
// Column sums (bias gradients)
// ------------------------------------------------------------------------------------------------------------
int colsum_blocks(int64_t M, int C) {
  const int rows = 256 / (C / 4);
  int64_t b = fp_ceil_div(M, (int64_t)rows * 16);
  if (b > 512) b = 512;
  if (b < 1) b = 1;
  return (int)b;
}
__global__ void __launch_bounds__(256) colsum_kernel(const float* __restrict__ x, int M, int C, float* __restrict__ part) {
  __shared__ float sm[256 * 4];
  const int C4 = C >> 2, R = 256 / C4;   // threads >= R*C4 idle when C4 does not divide 256
  const int cq = threadIdx.x % C4, rr = threadIdx.x / C4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int m = blockIdx.x * R + rr; rr < R && m < M; m += gridDim.x * R) {
    const float4 v = *reinterpret_cast<const float4*>(x + (size_t)m * C + cq * 4);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  reinterpret_cast<float4*>(sm)[threadIdx.x] = s;
  __syncthreads();
  if (rr == 0) {
    for (int r = 1; r < R; ++r) {
      const float4 o = reinterpret_cast<float4*>(sm)[r * C4 + cq];
      s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
    }
    *reinterpret_cast<float4*>(part + (size_t)blockIdx.x * C + cq * 4) = s;
  }
}
__global__ void __launch_bounds__(256) colsum_final_kernel(const float* __restrict__ part, int nblk, int C, float* out, int accumulate) {
  const int lane = threadIdx.x & 63;                      // one wave per channel, fixed reduction tree
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= C) return;
  float s = 0.f;
  for (int b = lane; b < nblk; b += 64) s += part[(size_t)b * C + c];
  s = fp_wave_sum(s);
  if (lane == 0) out[c] = accumulate ? out[c] + s : s;
}

// ------------------------------------------------------------------------------------------------------------
// backward of cat[nearest_x2(low), skip]  (network.py:154-155, :98)
// ------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) up2cat_bwd_kernel(const float* __restrict__ dxv, int N, int h, int w, int C0, int C1,
                                                         const float* __restrict__ addend, const float* __restrict__ ylow,
                                                         float* __restrict__ dlow, float* __restrict__ dskip, int acc_skip) {
  const int C = C0 + C1, Q0 = C0 >> 2, Q1 = C1 >> 2, W2 = 2 * w;
  const size_t nlow = (size_t)N * h * w * Q0;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < nlow; e += (size_t)gridDim.x * 256) {
    const int q = (int)(e % Q0);
    size_t r = e / Q0;
    const int x = (int)(r % w); r /= w;
    const int y = (int)(r % h);
    const int n = (int)(r / h);
    const float* p = dxv + (((size_t)(n * 2 * h + 2 * y)) * W2 + 2 * x) * C + q * 4;
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + C);
    const float4 c = *reinterpret_cast<const float4*>(p + (size_t)W2 * C), d = *reinterpret_cast<const float4*>(p + (size_t)W2 * C + C);
    float4 g = make_float4((a.x + b.x) + (c.x + d.x), (a.y + b.y) + (c.y + d.y), (a.z + b.z) + (c.z + d.z), (a.w + b.w) + (c.w + d.w));
    const size_t o = (e / Q0) * C0 + q * 4;
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
  }
  if (C1 > 0) {
    const size_t nskip = (size_t)N * 4 * h * w * Q1;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < nskip; e += (size_t)gridDim.x * 256) {
      const int q = (int)(e % Q1);
      const size_t pixel = e / Q1;
      float4 g = *reinterpret_cast<const float4*>(dxv + pixel * C + C0 + q * 4);
      float4* o = reinterpret_cast<float4*>(dskip + pixel * C1 + q * 4);
      if (acc_skip) { const float4 v = *o; g.x += v.x; g.y += v.y; g.z += v.z; g.w += v.w; }
      *o = g;
    }
  }
}

__global__ void __launch_bounds__(256) nchw_to_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int C, int H, int W) {
  const size_t total = (size_t)N * C * H * W, HW = (size_t)H * W;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int c = (int)(e % C);
    const size_t r = e / C;
    const size_t hw = r % HW, n = r / HW;
    y[e] = x[(n * C + c) * HW + hw];
  }
}
__global__ void __launch_bounds__(256) nhwc_to_nchw_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int C, int H, int W) {
  const size_t total = (size_t)N * C * H * W, HW = (size_t)H * W;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const size_t hw = e % HW;
    const size_t r = e / HW;
    const int c = (int)(r % C);
    const size_t n = r / C;
    y[e] = x[(n * HW + hw) * C + c];
  }
}
__global__ void __launch_bounds__(256) fill_kernel(float* x, size_t n, float v) {
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) x[e] = v;
}

}  // namespace

extern "C" int64_t fp_loss_workspace(int32_t B, int32_t H, int32_t W) {
  return (int64_t)loss_blocks((int64_t)B * H * W) * 16 * (int64_t)sizeof(float);
}

extern "C" int fp_loss_fwd_bwd(const float* const preds[4], const float* visible_ground, const float* all_ground, const float* depth,
                               const float* ground_depth, const float* moving_object_mask, const float* depth_mask, float min_depth,
                               float max_depth, float prior_weight, float* const dpreds[4], float* losses_out, int32_t B, int32_t H,
                               int32_t W, void* workspace, int64_t workspace_bytes, fp_stream_t stream) {
  FP_REQUIRE(preds && visible_ground && all_ground && depth && ground_depth && moving_object_mask && depth_mask && losses_out && workspace,
             "fp_loss_fwd_bwd: null pointer");
  FP_REQUIRE(workspace_bytes >= fp_loss_workspace(B, H, W), "fp_loss_fwd_bwd: workspace too small");
  LossArgs a;
  for (int s = 0; s < 4; ++s) {
    FP_REQUIRE(preds[s], "fp_loss_fwd_bwd: null prediction");
    a.pred[s] = preds[s];
    a.dpred[s] = dpreds ? dpreds[s] : nullptr;
  }
  if (dpreds) FP_REQUIRE(dpreds[0] && dpreds[1] && dpreds[2] && dpreds[3], "fp_loss_fwd_bwd: null gradient buffer");
  a.vg = visible_ground; a.ag = all_ground; a.depth = depth; a.gdepth = ground_depth; a.mov = moving_object_mask; a.dm = depth_mask;
  // utils.py:38-41: python doubles, cast to float32 when they meet the float32 tensor
  const double min_disp = 1.0 / (double)max_depth, max_disp = 1.0 / (double)min_depth;
  a.min_disp = (float)min_disp;
  a.disp_range = (float)(max_disp - min_disp);
  a.prior = prior_weight;
  const int64_t npix = (int64_t)B * H * W;
  a.gscale = (float)(1.0 / (4.0 * (double)npix));
  a.B = B; a.HW = H * W;
  a.part = (float*)workspace;
  const int nblk = loss_blocks(npix);
  fp_launch(loss_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, a);
  fp_launch(loss_final_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const float*)workspace, nblk, 1.0 / (double)npix,
                     losses_out);
  return fp_check_launch("fp_loss_fwd_bwd");
}

extern "C" int fp_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, double lr, double beta1,
                            double beta2, double eps, int32_t step, double grad_scale, fp_stream_t stream) {
  FP_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1, "fp_adam_step: bad arguments");
  FP_REQUIRE(((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) % 16 == 0, "fp_adam_step: buffers must be 16-byte aligned");
  // hyper-parameters arrive as the python doubles torch.optim.Adam uses; every derived scalar is formed in double and
  // rounded to float once, exactly like torch's scalar arguments (1 - beta2 = 0.001, not 1.f - 0.999f)
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  const float step_size = (float)(lr / bc1);
  const float inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
  fp_launch(adam_kernel, dim3(ew_grid((size_t)n / 4 + 1, 4096)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                     exp_avg_sq, (size_t)n, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), step_size, inv_bc2_sqrt, (float)eps,
                     (float)grad_scale);
  return fp_check_launch("fp_adam_step");
}

// Same update with the seven per-step scalars read from device memory (hipGraph replay: the launch arguments are frozen at
// capture time, the scalars are refreshed by a small host-to-device copy before every replay).
__global__ void __launch_bounds__(256) adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, size_t n, const float* __restrict__ hyper) {
  const float one_minus_b1 = hyper[0], b2 = hyper[1], one_minus_b2 = hyper[2], step_size = hyper[3], inv_bc2_sqrt = hyper[4],
              eps = hyper[5], gscale = hyper[6];
  const size_t n4 = n >> 2;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += (size_t)gridDim.x * 256) {
    float4 P = reinterpret_cast<float4*>(p)[e], G = reinterpret_cast<const float4*>(g)[e];
    float4 Mv = reinterpret_cast<float4*>(m)[e], V = reinterpret_cast<float4*>(v)[e];
    float* pp = &P.x; float* gg = &G.x; float* mm = &Mv.x; float* vv = &V.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gr = gg[j] * gscale;
      mm[j] = mm[j] + (gr - mm[j]) * one_minus_b1;
      vv[j] = vv[j] * b2 + one_minus_b2 * gr * gr;
      const float denom = sqrtf(vv[j]) * inv_bc2_sqrt + eps;
      pp[j] = pp[j] - step_size * (mm[j] / denom);
    }
    reinterpret_cast<float4*>(p)[e] = P;
    reinterpret_cast<float4*>(m)[e] = Mv;
    reinterpret_cast<float4*>(v)[e] = V;
  }
}

// host side of the above: the seven floats fp_adam_step would pass to its kernel for this step
extern "C" int fp_adam_hyper(double lr, double beta1, double beta2, double eps, int32_t step, double grad_scale, float* hyper7_host) {
  FP_REQUIRE(hyper7_host && step >= 1, "fp_adam_hyper: bad arguments");
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  hyper7_host[0] = (float)(1.0 - beta1); hyper7_host[1] = (float)beta2; hyper7_host[2] = (float)(1.0 - beta2);
  hyper7_host[3] = (float)(lr / bc1); hyper7_host[4] = (float)(1.0 / sqrt(bc2)); hyper7_host[5] = (float)eps;
  hyper7_host[6] = (float)grad_scale;
  return FP_OK;
}

extern "C" int fp_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, const float* hyper7_dev,
                                fp_stream_t stream) {
  FP_REQUIRE(param && grad && exp_avg && exp_avg_sq && hyper7_dev && n > 0 && n % 4 == 0, "fp_adam_step_dev: bad arguments");
  FP_REQUIRE(((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) % 16 == 0, "fp_adam_step_dev: buffers must be 16-byte aligned");
  fp_launch(adam_dev_kernel, dim3(ew_grid((size_t)n / 4 + 1, 4096)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                     exp_avg_sq, (size_t)n, hyper7_dev);
  return fp_check_launch("fp_adam_step_dev");
}

// out[r][k] = w[r][k] * scale[r]  (folding eval-mode BatchNorm into the preceding conv's OIHW weights)
__global__ void __launch_bounds__(256) scale_rows_kernel(const float* __restrict__ w, const float* __restrict__ scale,
                                                         float* __restrict__ out, size_t total, int inner) {
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) out[e] = w[e] * scale[e / inner];
}

extern "C" int fp_scale_rows(const float* w, const float* scale, float* out, int64_t rows, int64_t inner, fp_stream_t stream) {
  FP_REQUIRE(w && scale && out && rows > 0 && inner > 0 && inner < ((int64_t)1 << 31), "fp_scale_rows: bad arguments");
  fp_launch(scale_rows_kernel, dim3(ew_grid((size_t)rows * inner)), dim3(256), 0, (hipStream_t)stream, w, scale, out,
                     (size_t)rows * inner, (int)inner);
  return fp_check_launch("fp_scale_rows");
}

// Test-set inference output (reference evaluation/inference.py:105-108 + datasets/inference_dataset.py:35-38): sigmoid on the two
// mask channels (0, 1), depth channels (2, 3) unchanged, everything rounded to float16 (round-to-nearest-even == numpy astype).
__global__ void __launch_bounds__(256) pack_pred_fp16_kernel(const float* __restrict__ pred, __half* __restrict__ out, size_t plane,
                                                             size_t total) {
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int c = (int)((e / plane) & 3);
    float v = pred[e];
    if (c < 2) v = 1.f / (1.f + expf(-v));
    out[e] = __float2half_rn(v);
  }
}

extern "C" int fp_pack_pred_fp16(const float* pred_nchw, void* out_half, int32_t B, int32_t H, int32_t W, fp_stream_t stream) {
  FP_REQUIRE(pred_nchw && out_half && B > 0 && H > 0 && W > 0, "fp_pack_pred_fp16: bad arguments");
  const size_t plane = (size_t)H * W, total = plane * 4 * B;
  fp_launch(pack_pred_fp16_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, pred_nchw, (__half*)out_half, plane,
                     total);
  return fp_check_launch("fp_pack_pred_fp16");
}

extern "C" int64_t fp_colsum_workspace(int64_t M, int32_t C) {
  if (C < 4 || C % 4 || C > 1024) return 0;
  return (int64_t)colsum_blocks(M, C) * C * (int64_t)sizeof(float);
}

extern "C" int fp_colsum(const float* x, int64_t M, int32_t C, float* out, int accumulate, void* workspace, int64_t workspace_bytes,
                         fp_stream_t stream) {
  FP_REQUIRE(x && out && workspace, "fp_colsum: null pointer");
  FP_REQUIRE(C >= 4 && C % 4 == 0 && C <= 1024 && M > 0 && M < ((int64_t)1 << 31), "fp_colsum: unsupported C=%d", C);
  FP_REQUIRE(workspace_bytes >= fp_colsum_workspace(M, C), "fp_colsum: workspace too small");
  const int nblk = colsum_blocks(M, C);
  fp_launch(colsum_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, x, (int)M, C, (float*)workspace);
  fp_launch(colsum_final_kernel, dim3((C + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, nblk, C,
                     out, accumulate);
  return fp_check_launch("fp_colsum");
}

extern "C" int fp_up2cat_bwd(const float* dxv, int32_t N, int32_t h, int32_t w, int32_t C0, int32_t C1, const float* addend,
                             const float* ylow_elu, float* dlow, float* dskip, int accumulate_skip, fp_stream_t stream) {
  FP_REQUIRE(dxv && dlow && C0 > 0 && C0 % 4 == 0 && C1 % 4 == 0 && (C1 == 0 || dskip), "fp_up2cat_bwd: bad arguments");
  const size_t nlow = (size_t)N * h * w * (C0 / 4), nskip = (size_t)N * 4 * h * w * (C1 / 4);
  const size_t total = nlow > nskip ? nlow : nskip;
  fp_launch(up2cat_bwd_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, dxv, N, h, w, C0, C1, addend,
                     ylow_elu, dlow, dskip, accumulate_skip);
  return fp_check_launch("fp_up2cat_bwd");
}

extern "C" int fp_nchw_to_nhwc(const float* x, float* y, int32_t N, int32_t C, int32_t H, int32_t W, fp_stream_t stream) {
  FP_REQUIRE(x && y, "fp_nchw_to_nhwc: null pointer");
  fp_launch(nchw_to_nhwc_kernel, dim3(ew_grid((size_t)N * C * H * W)), dim3(256), 0, (hipStream_t)stream, x, y, N, C, H, W);
  return fp_check_launch("fp_nchw_to_nhwc");
}
extern "C" int fp_nhwc_to_nchw(const float* x, float* y, int32_t N, int32_t C, int32_t H, int32_t W, fp_stream_t stream) {
  FP_REQUIRE(x && y, "fp_nhwc_to_nchw: null pointer");
  fp_launch(nhwc_to_nchw_kernel, dim3(ew_grid((size_t)N * C * H * W)), dim3(256), 0, (hipStream_t)stream, x, y, N, C, H, W);
  return fp_check_launch("fp_nhwc_to_nchw");
}
extern "C" int fp_fill(float* x, int64_t n, float value, fp_stream_t stream) {
  FP_REQUIRE(x && n >= 0, "fp_fill: bad arguments");
  if (n == 0) return FP_OK;
  fp_launch(fill_kernel, dim3(ew_grid((size_t)n)), dim3(256), 0, (hipStream_t)stream, x, (size_t)n, value);
  return fp_check_launch("fp_fill");
}

// ---- amax slots of the fp16-pair operand format (fp_common.h) -----------------------------------------------------------
namespace {
__global__ void __launch_bounds__(256) amax_kernel(const float* __restrict__ x, size_t n, unsigned* __restrict__ slot) {
  const size_t n4 = n >> 2;
  float m = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = fmaxf(m, fabsf(x[(n4 << 2) + threadIdx.x]));
  m = fp_wave_max(m);
  __shared__ float wm[4];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
    fp_amax_publish(slot, blockIdx.x, m);
  }
}
__global__ void __launch_bounds__(256) zero_u32_kernel(unsigned* __restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0u;
}
}  // namespace

extern "C" int32_t fp_amax_slot_elems(void) { return FP_AMAX_ELEMS; }

// ---- clock probe (bench.py --sustain): the shader clock the chip REALLY sustains, from its own counters --------------------------------------
// One workgroup per XCD slot writes (s_memtime = shader cycles, s_memrealtime = constant-rate ticks) into out[xcc][2]; between two probes on
// one stream,  d cycles / d ticks x (wall-clock rate) = the average shader clock of that XCD over the interval -- no sysfs, no SMU sampling
// period (on the test boxes hwmon's freq1_input reads the idle clock while a process keeps the GPU busy; rounds 2-4 derived 1.65-1.74 GHz
// under the tile kernels from per-workgroup stamps of the same counter, profiles/round2_notes.md).
__global__ void __launch_bounds__(64) clock_probe_kernel(unsigned long long* __restrict__ out) {
  if (threadIdx.x == 0) {
    const unsigned x = fp_xcc_id() & 7u;
    const unsigned long long c = __builtin_readcyclecounter(), r = __builtin_amdgcn_s_memrealtime();
    out[x * 2] = c;                      // any workgroup of the XCD: they run within microseconds of each other
    out[x * 2 + 1] = r;
  }
}

extern "C" int fp_clock_probe(uint64_t* out16, fp_stream_t stream) {
  FP_REQUIRE(out16, "fp_clock_probe: null pointer");
  fp_launch(clock_probe_kernel, dim3(64), dim3(64), 0, (hipStream_t)stream, (unsigned long long*)out16);
  return fp_check_launch("fp_clock_probe");
}

// ---- matrix-pipe probe (bench.py's roofline leg, round 6): what v_mfma_f32_32x32x16_bf16 sustains on THIS chip with operands that toggle like
// real activations.  768 workgroups (three waves per SIMD), two accumulator chains per wave, nothing but MFMAs in the loop; mode 0 = constant
// operands, mode 1 = eight pseudo-random bf16 operand pairs per lane cycled without a vector instruction.  Measured (profiles/
// round6_mfma_sustained_clock.txt): constant operands 2.46 PFLOP/s at 2.39 GHz for seconds on end; pseudo-random operands 1.85 PFLOP/s at
// 1.81 GHz -- the clock the chip holds depends on how much the operand buses toggle, and the nominal dense peak (2.5 PFLOP/s = 2.4 GHz) is a
// constant-data figure.  out: 768 * 256 floats (keeps the accumulators alive); clk2 (optional): shader cycles and constant-rate ticks of
// workgroup 0 across the launch.
namespace {
typedef __bf16 fp_probe_bf16x8 __attribute__((ext_vector_type(8)));
__global__ void __launch_bounds__(256) mfma_probe_kernel(float* __restrict__ out, int iters, unsigned long long* __restrict__ clk, int mode) {
  f32x16 acc[2];
  for (int j = 0; j < 2; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  fp_probe_bf16x8 av[8], bv[8];
  {
    unsigned x = 0x9E3779B9u * (threadIdx.x + 1) + 0x85EBCA6Bu * (blockIdx.x + 1);
    for (int q = 0; q < 8; ++q) {
      unsigned w[8];
      for (int e = 0; e < 8; ++e) {
        x ^= x << 13; x ^= x >> 17; x ^= x << 5;
        w[e] = mode ? ((x & 0x807f807fu) | 0x3f003f00u) : (e < 4 ? 0x3f803f80u : 0x3c003c00u);      // sign + mantissa random, exponent of [0.5, 1)
      }
      av[q] = __builtin_bit_cast(fp_probe_bf16x8, make_uint4(w[0], w[1], w[2], w[3]));
      bv[q] = __builtin_bit_cast(fp_probe_bf16x8, make_uint4(w[4], w[5], w[6], w[7]));
    }
  }
  unsigned long long c0 = 0, r0 = 0;
  if (clk && blockIdx.x == 0 && threadIdx.x == 0) { c0 = __builtin_readcyclecounter(); r0 = __builtin_amdgcn_s_memrealtime(); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 12; ++rep)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[(rep * 2 + j) & 7], bv[(rep * 5 + j * 3) & 7], acc[j], 0, 0, 0);
  }
  float sum = 0.f;
  for (int j = 0; j < 2; ++j)
    for (int r = 0; r < 16; ++r) sum += acc[j][r];
  out[blockIdx.x * 256 + threadIdx.x] = sum;
  if (clk && blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_readcyclecounter() - c0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
}
}  // namespace

// one launch of 768 x 4 waves x iters x 24 MFMAs = iters * 2.416e12 FLOP (fp_mfma_probe_flop); asynchronous on the stream
extern "C" int fp_mfma_probe(float* out, uint64_t* clk2, int32_t iters, int32_t mode, fp_stream_t stream) {
  FP_REQUIRE(out && iters > 0, "fp_mfma_probe: bad arguments");
  fp_launch(mfma_probe_kernel, dim3(768), dim3(256), 0, (hipStream_t)stream, out, (int)iters, (unsigned long long*)clk2, (int)mode);
  return fp_check_launch("fp_mfma_probe");
}
extern "C" double fp_mfma_probe_flop(int32_t iters) { return 768.0 * 4 * (double)iters * 24 * 2.0 * 32 * 32 * 16; }

extern "C" int fp_wall_clock_khz(void) {
  int dev = 0, khz = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess) return -1;
  return khz;
}

extern "C" int fp_zero_u32(uint32_t* p, int64_t n, fp_stream_t stream) {
  FP_REQUIRE(p && n > 0, "fp_zero_u32: bad arguments");
  fp_launch(zero_u32_kernel, dim3(ew_grid((size_t)n, 64)), dim3(256), 0, (hipStream_t)stream, p, (size_t)n);
  return fp_check_launch("fp_zero_u32");
}

extern "C" int fp_amax_f32(const float* x, int64_t n, uint32_t* slot, fp_stream_t stream) {
  FP_REQUIRE(x && slot && n > 0, "fp_amax_f32: bad arguments");
  FP_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0, "fp_amax_f32: x must be 16-byte aligned");
  fp_launch(amax_kernel, dim3(ew_grid((size_t)n / 16 + 1, 1024)), dim3(256), 0, (hipStream_t)stream, x, (size_t)n, slot);
  return fp_check_launch("fp_amax_f32");
}
