// Pyramid-pooling pieces of the ground-segmentation network (reference footprints/preprocessing/segmentation/network.py:174-207):
// nn.AdaptiveAvgPool2d(P) on the 1/32 feature map, bilinear interpolation with align_corners=True back to its resolution, and the
// channel concatenation torch.cat([x, x6, x4, x2, x1], 1) -- plus their gradients.  The maps are tiny (at most 16 x 20 pixels,
// P <= 6), so every kernel is one thread per output element (4 channels each), gather form, no atomics: HBM / launch-latency bound.
// The 1x1 reduce convolution between the two is fp_conv_igemm / fp_conv_wgrad.
#include "fp_common.h"

namespace {

// AdaptiveAvgPool2d window of output index i (ATen adaptive pooling: start = floor(i * in / out), end = ceil((i + 1) * in / out))
__device__ __forceinline__ int ap_start(int i, int in, int out) { return (i * in) / out; }
__device__ __forceinline__ int ap_end(int i, int in, int out) { return ((i + 1) * in + out - 1) / out; }

__global__ void __launch_bounds__(256) avgpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C, int P) {
  const int C4 = C >> 2;
  const int64_t total = (int64_t)N * P * P * C4;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(e % C4);
    int64_t r = e / C4;
    const int px = (int)(r % P); r /= P;
    const int py = (int)(r % P);
    const int n = (int)(r / P);
    const int y0 = ap_start(py, H, P), y1 = ap_end(py, H, P), x0 = ap_start(px, W, P), x1 = ap_end(px, W, P);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int yy = y0; yy < y1; ++yy)
      for (int xx = x0; xx < x1; ++xx) {
        const float4 v = *reinterpret_cast<const float4*>(x + ((size_t)(n * H + yy) * W + xx) * C + c4 * 4);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
    const float inv = 1.f / (float)((y1 - y0) * (x1 - x0));
    *reinterpret_cast<float4*>(y + ((size_t)(n * P + py) * P + px) * C + c4 * 4) = make_float4(s.x * inv, s.y * inv, s.z * inv, s.w * inv);
  }
}

// dx[n][y][x][c] (+)= sum over the windows that contain (y, x) of dy / window area  (windows overlap when H % P != 0)
__global__ void __launch_bounds__(256) avgpool_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int N, int H, int W, int C, int P,
                                                          int accumulate) {
  const int C4 = C >> 2;
  const int64_t total = (int64_t)N * H * W * C4;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(e % C4);
    int64_t r = e / C4;
    const int xx = (int)(r % W); r /= W;
    const int yy = (int)(r % H);
    const int n = (int)(r / H);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int py = 0; py < P; ++py) {
      const int y0 = ap_start(py, H, P), y1 = ap_end(py, H, P);
      if (yy < y0 || yy >= y1) continue;
      for (int px = 0; px < P; ++px) {
        const int x0 = ap_start(px, W, P), x1 = ap_end(px, W, P);
        if (xx < x0 || xx >= x1) continue;
        const float inv = 1.f / (float)((y1 - y0) * (x1 - x0));
        const float4 v = *reinterpret_cast<const float4*>(dy + ((size_t)(n * P + py) * P + px) * C + c4 * 4);
        s.x += v.x * inv; s.y += v.y * inv; s.z += v.z * inv; s.w += v.w * inv;
      }
    }
    float4* d = reinterpret_cast<float4*>(dx + ((size_t)(n * H + yy) * W + xx) * C + c4 * 4);
    if (accumulate) { const float4 o = *d; s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w; }
    *d = s;
  }
}

// align_corners=True source coordinate (ATen area_pixel_compute_scale / compute_source_index, float accumulation type)
__device__ __forceinline__ void ac_coord(int o, int in, int out, int& i0, int& i1, float& l1) {
  const float scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
  const float src = scale * (float)o;
  i0 = min((int)src, in - 1);                       // guard_index_and_lambda
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = fminf(fmaxf(src - (float)i0, 0.f), 1.f);
}

// dst[n][y][x][c_off + c] = bilinear(src [N][P][P][C], align_corners=True) at (y, x); dst has dstC channels per pixel
__global__ void __launch_bounds__(256) bilinear_ac_fwd_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int P, int C, int H, int W,
                                                              int dstC, int c_off) {
  const int C4 = C >> 2;
  const int64_t total = (int64_t)N * H * W * C4;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(e % C4);
    int64_t r = e / C4;
    const int xx = (int)(r % W); r /= W;
    const int yy = (int)(r % H);
    const int n = (int)(r / H);
    int y0, y1, x0, x1;
    float ly, lx;
    ac_coord(yy, P, H, y0, y1, ly);
    ac_coord(xx, P, W, x0, x1, lx);
    const float hy = 1.f - ly, hx = 1.f - lx;
    const float* b = src + (size_t)n * P * P * C + c4 * 4;
    const float4 v00 = *reinterpret_cast<const float4*>(b + (size_t)(y0 * P + x0) * C), v01 = *reinterpret_cast<const float4*>(b + (size_t)(y0 * P + x1) * C);
    const float4 v10 = *reinterpret_cast<const float4*>(b + (size_t)(y1 * P + x0) * C), v11 = *reinterpret_cast<const float4*>(b + (size_t)(y1 * P + x1) * C);
    float4 o;                                     // ATen: h0 * (w0 * a + w1 * b) + h1 * (w0 * c + w1 * d)
    o.x = hy * (hx * v00.x + lx * v01.x) + ly * (hx * v10.x + lx * v11.x);
    o.y = hy * (hx * v00.y + lx * v01.y) + ly * (hx * v10.y + lx * v11.y);
    o.z = hy * (hx * v00.z + lx * v01.z) + ly * (hx * v10.z + lx * v11.z);
    o.w = hy * (hx * v00.w + lx * v01.w) + ly * (hx * v10.w + lx * v11.w);
    *reinterpret_cast<float4*>(dst + ((size_t)(n * H + yy) * W + xx) * dstC + c_off + c4 * 4) = o;
  }
}

// dsrc[n][py][px][c] = sum over the (y, x) whose interpolation touches (py, px) of weight * ddst[n][y][x][c_off + c]  (gather: every
// thread walks the H x W destination pixels of its image -- at most 16 x 20)
__global__ void __launch_bounds__(256) bilinear_ac_bwd_kernel(const float* __restrict__ ddst, float* __restrict__ dsrc, int N, int P, int C, int H, int W,
                                                              int dstC, int c_off) {
  const int C4 = C >> 2;
  const int64_t total = (int64_t)N * P * P * C4;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(e % C4);
    int64_t r = e / C4;
    const int px = (int)(r % P); r /= P;
    const int py = (int)(r % P);
    const int n = (int)(r / P);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int yy = 0; yy < H; ++yy) {
      int y0, y1;
      float ly;
      ac_coord(yy, P, H, y0, y1, ly);
      const float wy = (y0 == py ? 1.f - ly : 0.f) + (y1 == py ? ly : 0.f);
      if (wy == 0.f) continue;
      for (int xx = 0; xx < W; ++xx) {
        int x0, x1;
        float lx;
        ac_coord(xx, P, W, x0, x1, lx);
        const float wx = (x0 == px ? 1.f - lx : 0.f) + (x1 == px ? lx : 0.f);
        if (wx == 0.f) continue;
        const float4 g = *reinterpret_cast<const float4*>(ddst + ((size_t)(n * H + yy) * W + xx) * dstC + c_off + c4 * 4);
        const float wgt = wy * wx;
        s.x += wgt * g.x; s.y += wgt * g.y; s.z += wgt * g.z; s.w += wgt * g.w;
      }
    }
    *reinterpret_cast<float4*>(dsrc + ((size_t)(n * P + py) * P + px) * C + c4 * 4) = s;
  }
}

// dst[m][dst_off + c] (+)= src[m][src_off + c], c < C  (channel-slice copy / add between [M][srcC] and [M][dstC] tensors)
__global__ void __launch_bounds__(256) copy_channels_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t M, int C, int srcC, int src_off,
                                                            int dstC, int dst_off, int accumulate) {
  const int C4 = C >> 2;
  const int64_t total = M * C4;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(e % C4);
    const int64_t m = e / C4;
    float4 v = *reinterpret_cast<const float4*>(src + m * srcC + src_off + c4 * 4);
    float4* d = reinterpret_cast<float4*>(dst + m * dstC + dst_off + c4 * 4);
    if (accumulate) { const float4 o = *d; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
    *d = v;
  }
}

int grid_for(int64_t total) {
  int64_t g = fp_ceil_div(total, 256);
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

extern "C" int fp_adaptive_avgpool_fwd(const float* x, float* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t P, fp_stream_t stream_) {
  FP_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && P > 0 && C > 0 && C % 4 == 0, "fp_adaptive_avgpool_fwd: bad arguments (C must be a multiple of 4)");
  fp_launch(avgpool_fwd_kernel, dim3(grid_for((int64_t)N * P * P * (C / 4))), dim3(256), 0, (hipStream_t)stream_, x, y, N, H, W, C, P);
  return fp_check_launch("fp_adaptive_avgpool_fwd");
}

extern "C" int fp_adaptive_avgpool_bwd(const float* dy, float* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t P, int accumulate,
                                       fp_stream_t stream_) {
  FP_REQUIRE(dy && dx && N > 0 && H > 0 && W > 0 && P > 0 && C > 0 && C % 4 == 0, "fp_adaptive_avgpool_bwd: bad arguments (C must be a multiple of 4)");
  fp_launch(avgpool_bwd_kernel, dim3(grid_for((int64_t)N * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream_, dy, dx, N, H, W, C, P,
                     accumulate);
  return fp_check_launch("fp_adaptive_avgpool_bwd");
}

extern "C" int fp_bilinear_ac_fwd(const float* src, float* dst, int32_t N, int32_t P, int32_t C, int32_t H, int32_t W, int32_t dstC, int32_t c_off,
                                  fp_stream_t stream_) {
  FP_REQUIRE(src && dst && N > 0 && P > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && dstC % 4 == 0 && c_off % 4 == 0 && c_off + C <= dstC,
             "fp_bilinear_ac_fwd: bad arguments");
  fp_launch(bilinear_ac_fwd_kernel, dim3(grid_for((int64_t)N * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream_, src, dst, N, P, C, H, W,
                     dstC, c_off);
  return fp_check_launch("fp_bilinear_ac_fwd");
}

extern "C" int fp_bilinear_ac_bwd(const float* ddst, float* dsrc, int32_t N, int32_t P, int32_t C, int32_t H, int32_t W, int32_t dstC, int32_t c_off,
                                  fp_stream_t stream_) {
  FP_REQUIRE(ddst && dsrc && N > 0 && P > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && dstC % 4 == 0 && c_off % 4 == 0 && c_off + C <= dstC,
             "fp_bilinear_ac_bwd: bad arguments");
  fp_launch(bilinear_ac_bwd_kernel, dim3(grid_for((int64_t)N * P * P * (C / 4))), dim3(256), 0, (hipStream_t)stream_, ddst, dsrc, N, P, C, H, W,
                     dstC, c_off);
  return fp_check_launch("fp_bilinear_ac_bwd");
}

extern "C" int fp_copy_channels(const float* src, float* dst, int64_t M, int32_t C, int32_t srcC, int32_t src_off, int32_t dstC, int32_t dst_off,
                                int accumulate, fp_stream_t stream_) {
  FP_REQUIRE(src && dst && M > 0 && C > 0 && C % 4 == 0 && srcC % 4 == 0 && dstC % 4 == 0 && src_off % 4 == 0 && dst_off % 4 == 0 &&
                 src_off + C <= srcC && dst_off + C <= dstC,
             "fp_copy_channels: bad arguments");
  fp_launch(copy_channels_kernel, dim3(grid_for(M * (C / 4))), dim3(256), 0, (hipStream_t)stream_, src, dst, M, C, srcC, src_off, dstC, dst_off,
                     accumulate);
  return fp_check_launch("fp_copy_channels");
}
