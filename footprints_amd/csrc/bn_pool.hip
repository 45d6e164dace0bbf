// Train-mode BatchNorm (batch statistics), its backward, eval-mode coefficients, and the 3x3/2 max-pool of the
// ResNet-34 encoder (torchvision layers behind footprints/network.py:38-44).  All NHWC [M][C]: a channel is a
// column, so every reduction is a column reduction -- float4 (4 channels) per thread, rows strided across the
// block and the grid, Welford / Chan merges in a fixed tree => HBM-bound, deterministic, cancellation-safe.
#include <stdlib.h>

#include "fp_common.h"

namespace {

#ifndef FP_BN_MAX_BLOCKS
#define FP_BN_MAX_BLOCKS 512
#endif
constexpr int BN_MAX_BLOCKS = FP_BN_MAX_BLOCKS;

typedef FpWf Wf;   // Welford triple (fp_common.h)
__device__ __forceinline__ void wf_add(Wf& a, float x, float n, float rn) { fp_wf_add(a, x, n, rn); }
__device__ __forceinline__ void wf_merge(Wf& a, const Wf& b) { fp_wf_merge(a, b); }

int bn_blocks(int64_t M, int C) {
  const int rows = 256 / (C / 4);
  static const int iters = getenv("FP_BN_ROWS_PER_THREAD") ? atoi(getenv("FP_BN_ROWS_PER_THREAD")) : 4;      // 16 left the 6x20 ... 24x80 layers with 45-180 workgroups on 256 CUs, and these
                                                                                                          // reductions run alone on the GPU (encoder spine): step 14.39 -> 14.21 ms
  int64_t b = fp_ceil_div(M, (int64_t)rows * iters);
  if (b > BN_MAX_BLOCKS) b = BN_MAX_BLOCKS;
  if (b < 1) b = 1;
  return (int)b;
}

// partial[block][c] = Welford(n, mean, M2) over the block's rows
__global__ void __launch_bounds__(256) bn_stats_kernel(const float* __restrict__ z, int M, int C, float* __restrict__ part) {
  __shared__ float sm[3 * 256 * 4];
  const int C4 = C >> 2, R = 256 / C4;
  const int cq = threadIdx.x % C4, rr = threadIdx.x / C4;
  Wf w[4] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  float cnt = 0.f;
  for (int m = blockIdx.x * R + rr; m < M; m += gridDim.x * R) {
    const float4 v = *reinterpret_cast<const float4*>(z + (size_t)m * C + cq * 4);
    cnt += 1.f;
    const float rn = 1.f / cnt;
    wf_add(w[0], v.x, cnt, rn); wf_add(w[1], v.y, cnt, rn); wf_add(w[2], v.z, cnt, rn); wf_add(w[3], v.w, cnt, rn);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    sm[(0 * 256 + threadIdx.x) * 4 + j] = w[j].n;
    sm[(1 * 256 + threadIdx.x) * 4 + j] = w[j].mean;
    sm[(2 * 256 + threadIdx.x) * 4 + j] = w[j].m2;
  }
  __syncthreads();
  if (rr == 0) {
    for (int r = 1; r < R; ++r) {
      const int tt = r * C4 + cq;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        Wf o{sm[(0 * 256 + tt) * 4 + j], sm[(1 * 256 + tt) * 4 + j], sm[(2 * 256 + tt) * 4 + j]};
        wf_merge(w[j], o);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float* p = part + ((size_t)blockIdx.x * C + cq * 4 + j) * 3;
      p[0] = w[j].n; p[1] = w[j].mean; p[2] = w[j].m2;
    }
  }
}

// one wave per channel: lane l merges partials l, l+64, ... in order, then a fixed shuffle tree merges the lanes
// WPC = waves per channel: 1 (four channels per workgroup) or 4 (one channel per workgroup: the 720 .. 2 880 tile partials of the layer-1 / stem
// convolutions took one wave 10-14 us, on the critical path of the encoder's forward chain; the four waves' results merge through LDS in order)
template <int WPC>
__global__ void __launch_bounds__(256) bn_stats_final_kernel(const float* __restrict__ part, int nblk, int C,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float eps, float momentum, float* running_mean, float* running_var,
                                                             long long* nbt, float* save_mean, float* save_invstd, float* scale,
                                                             float* shift) {
  __shared__ double wsm[WPC == 4 ? 12 : 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = WPC == 4 ? blockIdx.x : blockIdx.x * 4 + wave;
  if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) *nbt += 1;
  if (c >= C) return;
  // the merge of the per-block triples runs in double (round 4; ATen's CPU BatchNorm accumulates in double too, at::acc_type<float, false>): a
  // few hundred double operations per channel.  It was introduced as a suspect for encoder.layer4.2.bn2.weight sitting at 9-33x the fp32 CPU
  // path's error and turned out NOT to be the cause (one ReLU decision is: profiles/round4_notes.md section 8); kept because it is free.
  struct Wd { double n, mean, m2; };
  auto merge = [](Wd& a, const Wd& b) {
    const double n = a.n + b.n;
    if (n == 0.0) return;
    const double d = b.mean - a.mean, f = b.n / n;
    a.mean += d * f;
    a.m2 += b.m2 + d * d * a.n * f;
    a.n = n;
  };
  Wd wd{0.0, 0.0, 0.0};
  for (int b = WPC == 4 ? threadIdx.x : lane; b < nblk; b += 64 * WPC) {
    const float* p = part + ((size_t)b * C + c) * 3;
    merge(wd, Wd{(double)p[0], (double)p[1], (double)p[2]});
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const Wd t{__shfl_down(wd.n, o, 64), __shfl_down(wd.mean, o, 64), __shfl_down(wd.m2, o, 64)};
    merge(wd, t);
  }
  if (WPC == 4) {                                     // (the whole workgroup serves one channel: no thread has left)
    if (lane == 0 && wave > 0) { wsm[wave * 3 + 0] = wd.n; wsm[wave * 3 + 1] = wd.mean; wsm[wave * 3 + 2] = wd.m2; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    for (int k = 1; k < 4; ++k) merge(wd, Wd{wsm[k * 3 + 0], wsm[k * 3 + 1], wsm[k * 3 + 2]});
  }
  const Wf w{(float)wd.n, (float)wd.mean, (float)wd.m2};
  if (lane != 0) return;
  const float var = w.m2 / w.n;                       // biased: used for normalisation
  const float invstd = 1.f / sqrtf(var + eps);
  save_mean[c] = w.mean;
  save_invstd[c] = invstd;
  const float sc = gamma[c] * invstd;
  scale[c] = sc;
  shift[c] = beta[c] - w.mean * sc;
  if (running_mean) {
    const float unbiased = w.n > 1.f ? w.m2 / (w.n - 1.f) : var;   // torch: running_var uses the unbiased estimate
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * w.mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
  }
}

__global__ void __launch_bounds__(256) bn_eval_coeffs_kernel(const float* gamma, const float* beta, const float* rm,
                                                             const float* rv, float eps, int C, float* scale, float* shift) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float sc = gamma[c] / sqrtf(rv[c] + eps);
  scale[c] = sc;
  shift[c] = beta[c] - rm[c] * sc;
}

__global__ void __launch_bounds__(256) bn_apply_kernel(const float* __restrict__ z, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const float* __restrict__ res,
                                                       float* __restrict__ y, size_t total4, int C4, int relu, unsigned* amax_out) {
  float ymax = 0.f;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total4; e += (size_t)gridDim.x * 256) {
    const int cq = (int)(e % C4);
    const float4 v = reinterpret_cast<const float4*>(z)[e];
    const float4 sc = reinterpret_cast<const float4*>(scale)[cq];
    const float4 sh = reinterpret_cast<const float4*>(shift)[cq];
    float4 o = make_float4(v.x * sc.x + sh.x, v.y * sc.y + sh.y, v.z * sc.z + sh.z, v.w * sc.w + sh.w);
    if (res) {
      const float4 r = reinterpret_cast<const float4*>(res)[e];
      o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
    }
    if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
    reinterpret_cast<float4*>(y)[e] = o;
    ymax = fp_amax4(ymax, o);
  }
  if (amax_out) fp_amax_publish_block(amax_out, ymax);
}

// partial[block][c][2] = (sum g, sum g*xhat), g = dy * (relu_out > 0)
__global__ void __launch_bounds__(256) bn_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ ro,
                                                            const float* __restrict__ z, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, int M, int C,
                                                            float* __restrict__ part) {
  __shared__ float sm[2 * 256 * 4];
  const int C4 = C >> 2, R = 256 / C4;
  const int cq = threadIdx.x % C4, rr = threadIdx.x / C4;
  const float4 mu = reinterpret_cast<const float4*>(mean)[cq];
  const float4 is = reinterpret_cast<const float4*>(invstd)[cq];
  float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  for (int m = blockIdx.x * R + rr; m < M; m += gridDim.x * R) {
    const size_t o = (size_t)m * C + cq * 4;
    float4 g = *reinterpret_cast<const float4*>(dy + o);
    if (ro) {
      const float4 r = *reinterpret_cast<const float4*>(ro + o);
      g.x = r.x > 0.f ? g.x : 0.f; g.y = r.y > 0.f ? g.y : 0.f; g.z = r.z > 0.f ? g.z : 0.f; g.w = r.w > 0.f ? g.w : 0.f;
    }
    const float4 v = *reinterpret_cast<const float4*>(z + o);
    s1[0] += g.x; s2[0] += g.x * ((v.x - mu.x) * is.x);
    s1[1] += g.y; s2[1] += g.y * ((v.y - mu.y) * is.y);
    s1[2] += g.z; s2[2] += g.z * ((v.z - mu.z) * is.z);
    s1[3] += g.w; s2[3] += g.w * ((v.w - mu.w) * is.w);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    sm[(0 * 256 + threadIdx.x) * 4 + j] = s1[j];
    sm[(1 * 256 + threadIdx.x) * 4 + j] = s2[j];
  }
  __syncthreads();
  if (rr == 0) {
    for (int r = 1; r < R; ++r) {
      const int tt = r * C4 + cq;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s1[j] += sm[(0 * 256 + tt) * 4 + j];
        s2[j] += sm[(1 * 256 + tt) * 4 + j];
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float* p = part + ((size_t)blockIdx.x * C + cq * 4 + j) * 2;
      p[0] = s1[j]; p[1] = s2[j];
    }
  }
}

// coef[c] = (sum g / M, sum g*xhat / M); dgamma/dbeta written.  One wave per channel, fixed reduction tree.
__global__ void __launch_bounds__(256) bn_bwd_final_kernel(const float* __restrict__ part, int nblk, int C, float invM,
                                                           float* coef, float* dgamma, float* dbeta, int accumulate) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= C) return;
  double d1 = 0.0, d2 = 0.0;                          // the per-block sums are combined in double (cheap: nblk values per channel)
  for (int b = lane; b < nblk; b += 64) {
    const float* p = part + ((size_t)b * C + c) * 2;
    d1 += (double)p[0]; d2 += (double)p[1];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    d1 += __shfl_down(d1, o, 64);
    d2 += __shfl_down(d2, o, 64);
  }
  if (lane != 0) return;
  const float s1 = (float)d1, s2 = (float)d2;
  coef[c * 2 + 0] = s1 * invM;
  coef[c * 2 + 1] = s2 * invM;
  if (dgamma) dgamma[c] = accumulate ? dgamma[c] + s2 : s2;
  if (dbeta) dbeta[c] = accumulate ? dbeta[c] + s1 : s1;
}

__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ ro,
                                                           const float* __restrict__ z, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ coef, float* __restrict__ dz,
                                                           float* __restrict__ gout, size_t total4, int C4, unsigned* amax_out) {
  float ymax = 0.f;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total4; e += (size_t)gridDim.x * 256) {
    const int cq = (int)(e % C4);
    float4 g = reinterpret_cast<const float4*>(dy)[e];
    if (ro) {
      const float4 r = reinterpret_cast<const float4*>(ro)[e];
      g.x = r.x > 0.f ? g.x : 0.f; g.y = r.y > 0.f ? g.y : 0.f; g.z = r.z > 0.f ? g.z : 0.f; g.w = r.w > 0.f ? g.w : 0.f;
    }
    if (gout) reinterpret_cast<float4*>(gout)[e] = g;
    const float4 v = reinterpret_cast<const float4*>(z)[e];
    const float4 mu = reinterpret_cast<const float4*>(mean)[cq];
    const float4 is = reinterpret_cast<const float4*>(invstd)[cq];
    const float4 ga = reinterpret_cast<const float4*>(gamma)[cq];
    const float4 c01 = reinterpret_cast<const float4*>(coef)[cq * 2];      // (c1,c2) of channels 0,1
    const float4 c23 = reinterpret_cast<const float4*>(coef)[cq * 2 + 1];  // channels 2,3
    float4 o;
    o.x = ga.x * is.x * (g.x - c01.x - (v.x - mu.x) * is.x * c01.y);
    o.y = ga.y * is.y * (g.y - c01.z - (v.y - mu.y) * is.y * c01.w);
    o.z = ga.z * is.z * (g.z - c23.x - (v.z - mu.z) * is.z * c23.y);
    o.w = ga.w * is.w * (g.w - c23.z - (v.w - mu.w) * is.w * c23.w);
    reinterpret_cast<float4*>(dz)[e] = o;
    ymax = fp_amax4(ymax, o);
  }
  if (amax_out) fp_amax_publish_block(amax_out, ymax);
}

// ---- max-pool 3x3 stride 2 pad 1; argmax = first maximum in (ky,kx) scan order (ATen max_pool2d tie rule) ----
__global__ void __launch_bounds__(256) maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          uint8_t* __restrict__ am, int N, int H, int W, int C, unsigned* amax_out) {
  const int OH = (H + 1) / 2, OW = (W + 1) / 2, C4 = C >> 2;
  const size_t total = (size_t)N * OH * OW * C4;
  float ymax = 0.f;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int cq = (int)(e % C4);
    size_t r = e / C4;
    const int ox = (int)(r % OW); r /= OW;
    const int oy = (int)(r % OH);
    const int n = (int)(r / OH);
    float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int bi[4] = {0, 0, 0, 0};
    bool first = true;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * 2 + ky - 1;
      if (iy < 0 || iy >= H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox * 2 + kx - 1;
        if (ix < 0 || ix >= W) continue;
        const float4 v = *reinterpret_cast<const float4*>(x + ((size_t)(n * H + iy) * W + ix) * C + cq * 4);
        const float vv[4] = {v.x, v.y, v.z, v.w};
        const int t = ky * 3 + kx;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (first || vv[j] > best[j] || vv[j] != vv[j]) { best[j] = vv[j]; bi[j] = t; }
        first = false;
      }
    }
    reinterpret_cast<float4*>(y)[e] = make_float4(best[0], best[1], best[2], best[3]);
    reinterpret_cast<uchar4*>(am)[e] = make_uchar4((unsigned char)bi[0], (unsigned char)bi[1], (unsigned char)bi[2], (unsigned char)bi[3]);
    ymax = fp_amax4(ymax, make_float4(best[0], best[1], best[2], best[3]));
  }
  if (amax_out) fp_amax_publish_block(amax_out, ymax);
}

__global__ void __launch_bounds__(256) maxpool_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ am,
                                                          float* __restrict__ dx, int N, int H, int W, int C, int accumulate) {
  const int OH = (H + 1) / 2, OW = (W + 1) / 2, C4 = C >> 2;
  const size_t total = (size_t)N * H * W * C4;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int cq = (int)(e % C4);
    size_t r = e / C4;
    const int ix = (int)(r % W); r /= W;
    const int iy = (int)(r % H);
    const int n = (int)(r / H);
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    // output windows containing (iy, ix): oy*2 + ky - 1 == iy
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int ty = iy + 1 - ky;
      if (ty < 0 || (ty & 1)) continue;
      const int oy = ty >> 1;
      if (oy >= OH) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int tx = ix + 1 - kx;
        if (tx < 0 || (tx & 1)) continue;
        const int ox = tx >> 1;
        if (ox >= OW) continue;
        const size_t o = ((size_t)(n * OH + oy) * OW + ox) * C4 + cq;
        const uchar4 a = reinterpret_cast<const uchar4*>(am)[o];
        const float4 d = reinterpret_cast<const float4*>(dy)[o];
        const unsigned char t = (unsigned char)(ky * 3 + kx);
        if (a.x == t) g.x += d.x;
        if (a.y == t) g.y += d.y;
        if (a.z == t) g.z += d.z;
        if (a.w == t) g.w += d.w;
      }
    }
    if (accumulate) {
      const float4 o = reinterpret_cast<float4*>(dx)[e];
      g.x += o.x; g.y += o.y; g.z += o.z; g.w += o.w;
    }
    reinterpret_cast<float4*>(dx)[e] = g;
  }
}

bool bn_c_ok(int C) { return C >= 4 && C % 4 == 0 && 256 % (C / 4) == 0 && C / 4 <= 256; }
int ew_grid(size_t total, size_t cap = 8192) {
  size_t g = (total + 255) / 256;
  return (int)(g > cap ? cap : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int64_t fp_bn_workspace(int64_t M, int32_t C) {
  const int nb = bn_blocks(M, C);
  return (int64_t)nb * C * 3 * (int64_t)sizeof(float) + (int64_t)C * 2 * sizeof(float);
}

extern "C" int fp_bn_train_stats(const float* z, int64_t M, int32_t C, const float* gamma, const float* beta, float eps,
                                 float momentum, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                 float* save_mean, float* save_invstd, float* scale, float* shift, void* workspace,
                                 int64_t workspace_bytes, fp_stream_t stream) {
  FP_REQUIRE(z && gamma && beta && save_mean && save_invstd && scale && shift && workspace, "fp_bn_train_stats: null pointer");
  FP_REQUIRE(bn_c_ok(C) && M > 0 && M < ((int64_t)1 << 31), "fp_bn_train_stats: unsupported C=%d", C);
  FP_REQUIRE(workspace_bytes >= fp_bn_workspace(M, C), "fp_bn_train_stats: workspace too small");
  const int nblk = bn_blocks(M, C);
  fp_launch(bn_stats_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, z, (int)M, C, (float*)workspace);
  fp_launch(bn_stats_final_kernel<1>, dim3((C + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, nblk,
                     C, gamma, beta, eps, momentum, running_mean, running_var, (long long*)num_batches_tracked, save_mean,
                     save_invstd, scale, shift);
  return fp_check_launch("fp_bn_train_stats");
}

// the second half of fp_bn_train_stats on Welford partials somebody else wrote: part[blk][c] = (n, mean, M2), e.g. the tile convolution's
// epilogue (fp_aux.bn_part) -- one launch instead of two, and the activation is not read again
extern "C" int fp_bn_train_stats_partials(const float* part, int32_t nblk, int32_t C, const float* gamma, const float* beta, float eps,
                                          float momentum, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                          float* save_mean, float* save_invstd, float* scale, float* shift, fp_stream_t stream) {
  FP_REQUIRE(part && gamma && beta && save_mean && save_invstd && scale && shift, "fp_bn_train_stats_partials: null pointer");
  FP_REQUIRE(nblk > 0 && C > 0, "fp_bn_train_stats_partials: empty problem");
  if (nblk > 256)
    fp_launch(bn_stats_final_kernel<4>, dim3(C), dim3(256), 0, (hipStream_t)stream, part, (int)nblk, C, gamma, beta, eps, momentum,
            running_mean, running_var, (long long*)num_batches_tracked, save_mean, save_invstd, scale, shift);
  else
    fp_launch(bn_stats_final_kernel<1>, dim3((C + 3) / 4), dim3(256), 0, (hipStream_t)stream, part, (int)nblk, C, gamma, beta, eps, momentum,
            running_mean, running_var, (long long*)num_batches_tracked, save_mean, save_invstd, scale, shift);
  return fp_check_launch("fp_bn_train_stats_partials");
}

extern "C" int fp_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                                 float eps, int32_t C, float* scale, float* shift, fp_stream_t stream) {
  FP_REQUIRE(gamma && beta && running_mean && running_var && scale && shift, "fp_bn_eval_coeffs: null pointer");
  fp_launch(bn_eval_coeffs_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta, running_mean,
                     running_var, eps, C, scale, shift);
  return fp_check_launch("fp_bn_eval_coeffs");
}

extern "C" int fp_bn_apply(const float* z, const float* scale, const float* shift, const float* residual, float* y, int64_t M,
                           int32_t C, int32_t relu, const fp_aux* aux, fp_stream_t stream) {
  unsigned* amax_out = fp_amax_out_of(aux);
  FP_REQUIRE(z && scale && shift && y && C % 4 == 0, "fp_bn_apply: bad arguments");
  const size_t total4 = (size_t)M * (C / 4);
  fp_launch(bn_apply_kernel, dim3(ew_grid(total4, 8192)), dim3(256), 0, (hipStream_t)stream, z, scale, shift,
            residual, y, total4, C / 4, relu, amax_out);
  return fp_check_launch("fp_bn_apply");
}

extern "C" int fp_bn_bwd(const float* dy, const float* relu_out, const float* z, const float* save_mean, const float* save_invstd,
                         const float* gamma, float* dz, float* g_out, float* dgamma, float* dbeta, int accumulate, int64_t M,
                         int32_t C, void* workspace, int64_t workspace_bytes, const fp_aux* aux, fp_stream_t stream) {
  unsigned* amax_out = fp_amax_out_of(aux);
  FP_REQUIRE(dy && z && save_mean && save_invstd && gamma && dz && workspace, "fp_bn_bwd: null pointer");
  FP_REQUIRE(bn_c_ok(C) && M > 0 && M < ((int64_t)1 << 31), "fp_bn_bwd: unsupported C=%d", C);
  FP_REQUIRE(workspace_bytes >= fp_bn_workspace(M, C), "fp_bn_bwd: workspace too small");
  const int nblk = bn_blocks(M, C);
  float* part = (float*)workspace;
  float* coef = part + (size_t)nblk * C * 3;   // 16-byte aligned: nblk*C*3 floats with C%4==0
  fp_launch(bn_bwd_reduce_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, dy, relu_out, z, save_mean, save_invstd,
                     (int)M, C, part);
  fp_launch(bn_bwd_final_kernel, dim3((C + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const float*)part, nblk, C,
                     1.f / (float)M, coef, dgamma, dbeta, accumulate);
  const size_t total4 = (size_t)M * (C / 4);
  fp_launch(bn_bwd_apply_kernel, dim3(ew_grid(total4, 8192)), dim3(256), 0, (hipStream_t)stream, dy, relu_out, z,
            save_mean, save_invstd, gamma, (const float*)coef, dz, g_out, total4, C / 4, amax_out);
  return fp_check_launch("fp_bn_bwd");
}

// fp_bn_bwd's second and third launch on partial sums (sum g, sum g * xhat) that a data-gradient epilogue already wrote (fp_aux.bnb_*):
// `g` is the masked gradient as stored by that launch (no relu_out here), `part` = [nblk][C][2], `coef` = 2 C floats of scratch
extern "C" int fp_bn_bwd_partials(const float* g, const float* z, const float* save_mean, const float* save_invstd, const float* gamma,
                                  float* dz, float* dgamma, float* dbeta, int accumulate, int64_t M, int32_t C, const float* part,
                                  int32_t nblk, float* coef, const fp_aux* aux, fp_stream_t stream) {
  unsigned* amax_out = fp_amax_out_of(aux);
  FP_REQUIRE(g && z && save_mean && save_invstd && gamma && dz && part && coef, "fp_bn_bwd_partials: null pointer");
  FP_REQUIRE(bn_c_ok(C) && M > 0 && M < ((int64_t)1 << 31) && nblk > 0, "fp_bn_bwd_partials: unsupported C=%d / nblk=%d", C, nblk);
  fp_launch(bn_bwd_final_kernel, dim3((C + 3) / 4), dim3(256), 0, (hipStream_t)stream, part, (int)nblk, C, 1.f / (float)M, coef, dgamma,
            dbeta, accumulate);
  const size_t total4 = (size_t)M * (C / 4);
  fp_launch(bn_bwd_apply_kernel, dim3(ew_grid(total4, 8192)), dim3(256), 0, (hipStream_t)stream, g, (const float*)nullptr, z, save_mean,
            save_invstd, gamma, (const float*)coef, dz, (float*)nullptr, total4, C / 4, amax_out);
  return fp_check_launch("fp_bn_bwd_partials");
}

extern "C" int fp_maxpool_fwd(const float* x, float* y, uint8_t* argmax, int32_t N, int32_t H, int32_t W, int32_t C,
                              const fp_aux* aux, fp_stream_t stream) {
  unsigned* amax_out = fp_amax_out_of(aux);
  FP_REQUIRE(x && y && argmax && C % 4 == 0, "fp_maxpool_fwd: bad arguments");
  const size_t total = (size_t)N * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4);
  fp_launch(maxpool_fwd_kernel, dim3(ew_grid(total, 8192)), dim3(256), 0, (hipStream_t)stream, x, y, argmax, N, H, W,
            C, amax_out);
  return fp_check_launch("fp_maxpool_fwd");
}

extern "C" int fp_maxpool_bwd(const float* dy, const uint8_t* argmax, float* dx, int32_t N, int32_t H, int32_t W, int32_t C,
                              int accumulate, fp_stream_t stream) {
  FP_REQUIRE(dy && dx && argmax && C % 4 == 0, "fp_maxpool_bwd: bad arguments");
  const size_t total = (size_t)N * H * W * (C / 4);
  fp_launch(maxpool_bwd_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, dy, argmax, dx, N, H, W, C,
                     accumulate);
  return fp_check_launch("fp_maxpool_bwd");
}
