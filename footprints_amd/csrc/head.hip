// 2-channel output heads (OutConvBlock, reference footprints/network.py:161-183): reflection-pad 3x3 conv
// Cin->2 (+sigmoid), bilinear upsample (align_corners=False) into the NCHW network output, and their
// backward passes.  ~8.7 flop/byte => HBM-bound and MFMA-hostile: direct VALU kernels, one pixel per Cin/4
// adjacent lanes (float4 NHWC loads, 128-B contiguous per pixel), wave-shuffle reductions, no atomics
// (every reduction has a fixed order => bit-reproducible).
#include "fp_common.h"

namespace {

constexpr int MAXC = 128;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// Ws0/Ws1[tap][Cin] <- w_oihw[2][Cin][3][3]
__device__ __forceinline__ void load_head_weights(const float* __restrict__ w, int Cin, float* Ws0, float* Ws1) {
  for (int e = threadIdx.x; e < 9 * Cin; e += blockDim.x) {
    const int tap = e / Cin, c = e - tap * Cin;
    Ws0[e] = w[(size_t)c * 9 + tap];
    Ws1[e] = w[(size_t)(Cin + c) * 9 + tap];
  }
}

// Workgroup mapping shared by the three 3x3 head kernels: 256 threads = 256/Q adjacent pixels of ONE image row (Q = Cin/4 lanes
// per pixel, four channels each), walking `rows` consecutive rows of one image with the 3x3 window's three rows kept in
// registers (each row is loaded once per workgroup, not three times) and the 2 x 9 x 4 weights of the lane's channels in
// registers.  blockIdx.x = column segment, blockIdx.y = image * groups_per_image + row group.  No per-pixel integer division.
struct HeadGrid {
  int colblocks, rows, gpi;
};

struct HeadLane {
  int q, ox, n, row0, rend;
  bool live;
};

__device__ __forceinline__ HeadLane head_lane(int H, int W, int Cin, int rows, int gpi) {
  HeadLane l;
  const int Q = Cin >> 2, lq = 31 - __clz(Q);
  l.q = threadIdx.x & (Q - 1);
  const int oxr = (blockIdx.x * 256 + threadIdx.x) >> lq;
  l.live = oxr < W;
  l.ox = l.live ? oxr : W - 1;
  l.n = blockIdx.y / gpi;
  l.row0 = (blockIdx.y - l.n * gpi) * rows;
  l.rend = min(l.row0 + rows, H);
  return l;
}

__device__ __forceinline__ float dot4(const float4& a, const float4& b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

__global__ void __launch_bounds__(256) head_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ low, int H, int W,
                                                       int Cin, int rows, int gpi, int sig) {
  __shared__ __attribute__((aligned(16))) float Ws0[9 * MAXC];
  __shared__ __attribute__((aligned(16))) float Ws1[9 * MAXC];
  load_head_weights(w, Cin, Ws0, Ws1);
  __syncthreads();
  const HeadLane l = head_lane(H, W, Cin, rows, gpi);
  const int Q = Cin >> 2;
  float4 w0[9], w1[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    w0[t] = *reinterpret_cast<const float4*>(Ws0 + t * Cin + l.q * 4);
    w1[t] = *reinterpret_cast<const float4*>(Ws1 + t * Cin + l.q * 4);
  }
  const float b0 = bias[0], b1 = bias[1];
  const int cm = fp_reflect(l.ox - 1, W) * Cin, c0 = l.ox * Cin, cp = fp_reflect(l.ox + 1, W) * Cin;
  const float* img = x + (size_t)l.n * H * W * Cin + l.q * 4;
  auto ldrow = [&](int iy, float4(&r)[3]) {
    const float* p = img + (size_t)iy * W * Cin;
    r[0] = *reinterpret_cast<const float4*>(p + cm);
    r[1] = *reinterpret_cast<const float4*>(p + c0);
    r[2] = *reinterpret_cast<const float4*>(p + cp);
  };
  // rows oy-1, oy, oy+1 in registers and row oy+2 in flight while row oy is computed: a ring of four row buffers, rotated by
  // unrolling (a register copy of a prefetched row would wait for its load)
  auto body = [&](int oy, float4(&top)[3], float4(&mid)[3], float4(&bot)[3], float4(&nxt)[3]) {
    ldrow(fp_reflect(min(oy + 2, H), H), nxt);
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      a0 += dot4(top[k], w0[k]) + dot4(mid[k], w0[3 + k]) + dot4(bot[k], w0[6 + k]);
      a1 += dot4(top[k], w1[k]) + dot4(mid[k], w1[3 + k]) + dot4(bot[k], w1[6 + k]);
    }
    for (int o = Q >> 1; o > 0; o >>= 1) {   // Q is a power of two <= 32: the pixel's lanes are adjacent
      a0 += __shfl_xor(a0, o, 64);
      a1 += __shfl_xor(a1, o, 64);
    }
    if (l.q == 0 && l.live) {
      float y0 = a0 + b0, y1 = a1 + b1;
      if (sig) { y0 = sigmoidf_(y0); y1 = sigmoidf_(y1); }
      *reinterpret_cast<float2*>(low + ((size_t)(l.n * H + oy) * W + l.ox) * 2) = make_float2(y0, y1);
    }
  };
  float4 r0[3], r1[3], r2[3], r3[3];
  ldrow(fp_reflect(l.row0 - 1, H), r0);
  ldrow(l.row0, r1);
  ldrow(fp_reflect(l.row0 + 1, H), r2);
  for (int oy = l.row0; oy < l.rend; oy += 4) {
    body(oy, r0, r1, r2, r3);
    if (oy + 1 >= l.rend) break;
    body(oy + 1, r1, r2, r3, r0);
    if (oy + 2 >= l.rend) break;
    body(oy + 2, r2, r3, r0, r1);
    if (oy + 3 >= l.rend) break;
    body(oy + 3, r3, r0, r1, r2);
  }
}

// bilinear source coordinate, PyTorch align_corners=False: src = (dst+0.5)/scale - 0.5, clamped at 0
__device__ __forceinline__ void bilin(int dst, float rscale, int n, int& i0, int& i1, float& l0, float& l1) {
  float src = ((float)dst + 0.5f) * rscale - 0.5f;
  src = src < 0.f ? 0.f : src;
  i0 = (int)src;
  i1 = i0 + (i0 < n - 1 ? 1 : 0);
  l1 = src - (float)i0;
  l0 = 1.f - l1;
}

__global__ void __launch_bounds__(256) head_upsample_kernel(const float* __restrict__ low, float* __restrict__ out, int N, int h,
                                                            int w, int scale, int OC, int c0) {
  const int H = h * scale, W = w * scale;
  const size_t total = (size_t)N * H * W;
  const float rs = 1.f / (float)scale;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int X = (int)(e % W);
    const size_t r = e / W;
    const int Y = (int)(r % H), n = (int)(r / H);
    float2 v;
    if (scale == 1) {
      v = *reinterpret_cast<const float2*>(low + e * 2);
    } else {
      int y0, y1, x0, x1;
      float ly0, ly1, lx0, lx1;
      bilin(Y, rs, h, y0, y1, ly0, ly1);
      bilin(X, rs, w, x0, x1, lx0, lx1);
      const float2 v00 = *reinterpret_cast<const float2*>(low + ((size_t)(n * h + y0) * w + x0) * 2);
      const float2 v01 = *reinterpret_cast<const float2*>(low + ((size_t)(n * h + y0) * w + x1) * 2);
      const float2 v10 = *reinterpret_cast<const float2*>(low + ((size_t)(n * h + y1) * w + x0) * 2);
      const float2 v11 = *reinterpret_cast<const float2*>(low + ((size_t)(n * h + y1) * w + x1) * 2);
      v.x = ly0 * (lx0 * v00.x + lx1 * v01.x) + ly1 * (lx0 * v10.x + lx1 * v11.x);
      v.y = ly0 * (lx0 * v00.y + lx1 * v01.y) + ly1 * (lx0 * v10.y + lx1 * v11.y);
    }
    const size_t plane = (size_t)H * W;
    float* o = out + ((size_t)n * OC + c0) * plane + (size_t)Y * W + X;
    o[0] = v.x;
    o[plane] = v.y;
  }
}

// gather-form transpose of the bilinear upsample: every low-res pixel sums the hi-res pixels that read it.
// Generic version (any scale): scans the conservative (3*scale+1)^2 window.
__global__ void __launch_bounds__(256) head_upsample_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ low,
                                                                float* __restrict__ dz, int N, int h, int w, int scale, int OC,
                                                                int c0, int sig) {
  const int H = h * scale, W = w * scale;
  const size_t total = (size_t)N * h * w;
  const size_t plane = (size_t)H * W;
  const float rs = 1.f / (float)scale;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int x = (int)(e % w);
    const size_t r = e / w;
    const int y = (int)(r % h), n = (int)(r / h);
    const float* d0 = dout + ((size_t)n * OC + c0) * plane;
    float g0 = 0.f, g1 = 0.f;
    if (scale == 1) {
      g0 = d0[(size_t)y * W + x];
      g1 = d0[plane + (size_t)y * W + x];
    } else {
      const int Y0 = max(0, scale * (y - 1)), Y1 = min(H - 1, scale * (y + 2));
      const int X0 = max(0, scale * (x - 1)), X1 = min(W - 1, scale * (x + 2));
      for (int Y = Y0; Y <= Y1; ++Y) {
        int i0, i1; float l0, l1;
        bilin(Y, rs, h, i0, i1, l0, l1);
        const float wy = (i0 == y ? l0 : 0.f) + (i1 == y ? l1 : 0.f);
        if (wy == 0.f) continue;
        float r0 = 0.f, r1 = 0.f;
        for (int X = X0; X <= X1; ++X) {
          int j0, j1; float m0, m1;
          bilin(X, rs, w, j0, j1, m0, m1);
          const float wx = (j0 == x ? m0 : 0.f) + (j1 == x ? m1 : 0.f);
          if (wx == 0.f) continue;
          r0 += wx * d0[(size_t)Y * W + X];
          r1 += wx * d0[plane + (size_t)Y * W + X];
        }
        g0 += wy * r0;
        g1 += wy * r1;
      }
    }
    if (sig) {
      const float2 s = *reinterpret_cast<const float2*>(low + e * 2);
      g0 *= s.x * (1.f - s.x);
      g1 *= s.y * (1.f - s.y);
    }
    *reinterpret_cast<float2*>(dz + e * 2) = make_float2(g0, g1);
  }
}

// weight with which hi-res index `dst` reads low-res index `i` (0 when it does not)
__device__ __forceinline__ float bilin_weight(int dst, float rscale, int n, int i) {
  int i0, i1;
  float l0, l1;
  bilin(dst, rscale, n, i0, i1, l0, l1);
  return (i0 == i ? l0 : 0.f) + (i1 == i ? l1 : 0.f);
}

// Even scales S = 2, 4, 8: low-res index i is read exactly by the 2S hi-res indices [S*i - S/2, S*i + 3S/2) (src in (i-1, i+1);
// the clamps at both borders stay inside that range), so the window is 2S x 2S, its column weights are computed once per thread,
// and for S >= 4 the rows of the window are shared by four adjacent lanes (fixed xor-tree sum => still bit-reproducible).
template <int S>
__global__ void __launch_bounds__(256) head_upsample_bwd_even_kernel(const float* __restrict__ dout, const float* __restrict__ low,
                                                                     float* __restrict__ dz, int N, int h, int w, int OC, int c0,
                                                                     int sig) {
  constexpr int L = S >= 4 ? 4 : 1;
  const int H = h * S, W = w * S;
  const size_t total = (size_t)N * h * w, plane = (size_t)H * W;
  const float rs = 1.f / (float)S;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int sub = (int)(idx % L);
  const size_t er = idx / L;
  const bool live = er < total;
  const size_t e = live ? er : total - 1;
  const int x = (int)(e % w);
  const size_t r = e / w;
  const int y = (int)(r % h), n = (int)(r / h);
  const float* d0 = dout + ((size_t)n * OC + c0) * plane;
  const int X0 = S * x - S / 2, Y0 = S * y - S / 2;
  float wx[2 * S];
  int xo[2 * S];
#pragma unroll
  for (int j = 0; j < 2 * S; ++j) {
    const int X = X0 + j;
    const bool in = X >= 0 && X < W;
    xo[j] = in ? X : (X < 0 ? 0 : W - 1);
    wx[j] = in ? bilin_weight(X, rs, w, x) : 0.f;
  }
  float g0 = 0.f, g1 = 0.f;
#pragma unroll
  for (int jy = 0; jy < 2 * S / L; ++jy) {
    const int Y = Y0 + jy * L + sub;
    if (Y < 0 || Y >= H) continue;
    const float wy = bilin_weight(Y, rs, h, y);
    const float* d = d0 + (size_t)Y * W;
    float r0 = 0.f, r1 = 0.f;
#pragma unroll
    for (int j = 0; j < 2 * S; ++j) {
      r0 += wx[j] * d[xo[j]];
      r1 += wx[j] * d[plane + xo[j]];
    }
    g0 += wy * r0;
    g1 += wy * r1;
  }
#pragma unroll
  for (int o = 1; o < L; o <<= 1) {
    g0 += __shfl_xor(g0, o, 64);
    g1 += __shfl_xor(g1, o, 64);
  }
  if (sub != 0 || !live) return;
  if (sig) {
    const float2 s = *reinterpret_cast<const float2*>(low + e * 2);
    g0 *= s.x * (1.f - s.x);
    g1 *= s.y * (1.f - s.y);
  }
  *reinterpret_cast<float2*>(dz + e * 2) = make_float2(g0, g1);
}

// The sets of head-output pixels whose 3x3 reflection-padded window reads input pixel p through tap k, per dimension:
// the base pixel p - k + 1 (when inside the image) plus, on rows/columns 1 and n-2, the border pixel whose out-of-image tap
// reflects onto p (k = 0 at p == 1 reads pixel 0; k = 2 at p == n-2 reads pixel n-1).  The data gradient needs
// Z[k] = sum of dz over that set; with the three neighbouring dz rows / columns (clamped loads) in
// registers it is two multiply-adds with 0/1 factors, so borders cost no branches.
struct HeadFold {
  float p, m1, m, p2;   // next * p + prev * m1 -> k = 0 ;  prev * m + next * p2 -> k = 2
};
__device__ __forceinline__ HeadFold head_fold(int i, int n) {
  return HeadFold{i + 1 < n ? 1.f : 0.f, i == 1 ? 1.f : 0.f, i >= 1 ? 1.f : 0.f, i == n - 2 ? 1.f : 0.f};
}

struct HeadDzWindow {
  const float2* img;
  int W, H, xm, x0, xp;
  HeadFold fx;
  __device__ __forceinline__ void init(const float* dz, int n, int H_, int W_, int ox) {
    img = reinterpret_cast<const float2*>(dz) + (size_t)n * H_ * W_;
    W = W_; H = H_;
    xm = max(ox - 1, 0); x0 = ox; xp = min(ox + 1, W_ - 1);
    fx = head_fold(ox, W_);
  }
  // column-folded dz row r (clamped; rows outside the image are masked by the caller's row factors)
  __device__ __forceinline__ void row(int r, float2 (&z)[3]) {
    const float2* p = img + (size_t)min(max(r, 0), H - 1) * W;
    const float2 a = p[xm], b = p[x0], c = p[xp];
    z[0] = make_float2(fx.p * c.x + fx.m1 * a.x, fx.p * c.y + fx.m1 * a.y);
    z[1] = b;
    z[2] = make_float2(fx.m * a.x + fx.p2 * c.x, fx.m * a.y + fx.p2 * c.y);
  }
};

__global__ void __launch_bounds__(256) head_dgrad_kernel(const float* __restrict__ dz, const float* __restrict__ w,
                                                         const float* __restrict__ elu_src, float* __restrict__ dx, int H, int W,
                                                         int Cin, int rows, int gpi, unsigned* amax_out) {
  float ymax = 0.f;
  __shared__ __attribute__((aligned(16))) float Ws0[9 * MAXC];
  __shared__ __attribute__((aligned(16))) float Ws1[9 * MAXC];
  load_head_weights(w, Cin, Ws0, Ws1);
  __syncthreads();
  const HeadLane l = head_lane(H, W, Cin, rows, gpi);
  float4 w0[9], w1[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    w0[t] = *reinterpret_cast<const float4*>(Ws0 + t * Cin + l.q * 4);
    w1[t] = *reinterpret_cast<const float4*>(Ws1 + t * Cin + l.q * 4);
  }
  HeadDzWindow win;
  win.init(dz, l.n, H, W, l.ox);
  const size_t pix0 = ((size_t)l.n * H * W + l.ox) * Cin + l.q * 4;
  const size_t rowstride = (size_t)W * Cin;
  // dz rows py-1, py, py+1 in registers, row py+2 and the next row's ELU operand in flight (ring rotated by unrolling)
  auto body = [&](int py, float2(&up)[3], float2(&md)[3], float2(&dn)[3], float2(&nxt)[3], float4& s_use, float4& s_load) {
    win.row(py + 2, nxt);
    if (elu_src) s_load = *reinterpret_cast<const float4*>(elu_src + pix0 + (size_t)min(py + 1, H - 1) * rowstride);
    const HeadFold fy = head_fold(py, H);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float2 z[3] = {make_float2(fy.p * dn[k].x + fy.m1 * up[k].x, fy.p * dn[k].y + fy.m1 * up[k].y), md[k],
                           make_float2(fy.m * up[k].x + fy.p2 * dn[k].x, fy.m * up[k].y + fy.p2 * dn[k].y)};
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const float4 a = w0[ky * 3 + k], b = w1[ky * 3 + k];
        acc.x += z[ky].x * a.x + z[ky].y * b.x;
        acc.y += z[ky].x * a.y + z[ky].y * b.y;
        acc.z += z[ky].x * a.z + z[ky].y * b.z;
        acc.w += z[ky].x * a.w + z[ky].y * b.w;
      }
    }
    if (elu_src) {
      const float4 s = s_use;
      acc.x *= (s.x > 0.f ? 1.f : s.x + 1.f); acc.y *= (s.y > 0.f ? 1.f : s.y + 1.f);
      acc.z *= (s.z > 0.f ? 1.f : s.z + 1.f); acc.w *= (s.w > 0.f ? 1.f : s.w + 1.f);
    }
    if (l.live) {
      *reinterpret_cast<float4*>(dx + pix0 + (size_t)py * rowstride) = acc;
      ymax = fp_amax4(ymax, acc);
    }
  };
  float2 r0[3], r1[3], r2[3], r3[3];
  float4 e0 = make_float4(0.f, 0.f, 0.f, 0.f), e1 = e0;
  win.row(l.row0 - 1, r0);
  win.row(l.row0, r1);
  win.row(l.row0 + 1, r2);
  if (elu_src) e0 = *reinterpret_cast<const float4*>(elu_src + pix0 + (size_t)l.row0 * rowstride);
  for (int py = l.row0; py < l.rend; py += 4) {
    body(py, r0, r1, r2, r3, e0, e1);
    if (py + 1 >= l.rend) break;
    body(py + 1, r1, r2, r3, r0, e1, e0);
    if (py + 2 >= l.rend) break;
    body(py + 2, r2, r3, r0, r1, e0, e1);
    if (py + 3 >= l.rend) break;
    body(py + 3, r3, r0, r1, r2, e1, e0);
  }
  if (amax_out) fp_amax_publish_block(amax_out, ymax);
}

// Weight gradient.  Round 4: SCATTER form over the input pixels -- a thread reads its four channels of x ONCE and multiplies them with the nine
// neighbouring dZ values (two floats per pixel: 1/16 of x at 32 channels), instead of gathering x nine times per output pixel (nine float4
// loads per thread: 17 TB/s through the L1s at 192 x 640, the kernel ran at 26 % of the HBM roof, profiles/round4_hbm_kernels.txt).  In terms
// of the reflection-padded image xp (H+2 x W+2): dW[t] = sum over padded positions p of xp(p) * dZ(p - t); an input pixel (iy, ix) sits at the
// padded position (iy+1, ix+1) and, when it is a mirror source, also at row 0 (iy == 1) / row H+1 (iy == H-2) and column 0 / W+1 likewise.
// The pixel coordinates advance by carries (no division in the loop).
// partial[block][(tap*Cin + c)*2 + o] and partial_b[block][2]
// the two output channels of one (tap, input channel): a plain pair of floats in the library build -- a vector type invites the backend to
// emit v_pk_add_f32 / v_pk_fma_f32 wherever both elements see the same operation -- and a two-element vector in the A/B builds below
#if !defined(FP_HEAD_WGRAD_FMA) || FP_HEAD_WGRAD_FMA == 0
#if !defined(FP_HEAD_WGRAD_PKRED)
#define FP_HEAD_PLAIN_PAIRS 1
#endif
#endif
#ifdef FP_HEAD_PLAIN_PAIRS
struct fp_v2f {
  float e[2];
  __device__ __forceinline__ float& operator[](int i) { return e[i]; }
  __device__ __forceinline__ const float& operator[](int i) const { return e[i]; }
};
#else
typedef float fp_v2f __attribute__((ext_vector_type(2)));
#endif
// acc (two output channels of one tap and input channel) += x * (z.x, z.y).  FP_HEAD_WGRAD_FMA: 0 = two v_fma_f32 (default: the library keeps
// packed fp32 VALU out of its code objects until round 1's run-to-run differences of THIS kernel's SLP-vectorised build next to the bf16 tile
// convolution are understood, tests/test_host_cpu.py); 1 = one v_pk_fma_f32 with the x operand broadcast by op_sel (the form the SLP build
// had); 2 = v_pk_fma_f32 on a materialised (x, x) register pair, no op_sel -- the two A/B builds of scripts/debug_head_wgrad_det.py.  The
// loop is latency-bound, the forms time the same.
#ifndef FP_HEAD_WGRAD_FMA
#define FP_HEAD_WGRAD_FMA 0
#endif
__device__ __forceinline__ void head_wfma(fp_v2f& acc, float xv, const fp_v2f& zz) {
#if FP_HEAD_WGRAD_FMA == 0 || FP_HEAD_WGRAD_FMA == 5
  acc[0] = fmaf(xv, zz[0], acc[0]);
  acc[1] = fmaf(xv, zz[1], acc[1]);
#elif FP_HEAD_WGRAD_FMA == 1
  acc = __builtin_elementwise_fma(fp_v2f{xv, xv}, zz, acc);
#elif FP_HEAD_WGRAD_FMA == 2
  fp_v2f xx = {xv, xv};
  asm volatile("" : "+v"(xx));
  acc = __builtin_elementwise_fma(xx, zz, acc);
#else       // 3: the broadcast on the SECOND source operand (op_sel_hi:[1,0,1] / op_sel:[0,1,0], the forms of round 1's SLP build), by inline asm
#if FP_HEAD_WGRAD_FMA == 3
  const fp_v2f xx = {xv, 0.f};
  asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(zz), "v"(xx));
#else      // 4: ... and the other one, op_sel:[0,1,0] (both results read the HIGH half of the second source): the accumulators that differed
  const fp_v2f xx = {FP_HEAD_WGRAD_FMA == 4 ? 0.f : 3.f * xv, xv};      // 6: a low half that shows when it is read instead (scripts/pk_opsel_probe.py)
  asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc) : "v"(zz), "v"(xx));
#endif
#endif
}
// all four input channels of one tap.  FP_HEAD_WGRAD_FMA == 5 (A/B build): pairs over CHANNELS instead -- a[0] = (c0, c1) x z.x, a[1] = (c0, c1) x z.y,
// a[2] / a[3] likewise for (c2, c3) -- so that the broadcast operand is the dZ pair as the load wrote it, selected by op_sel_hi:[1,0,1] (low
// half) and op_sel:[0,1,0] (high half): instruction for instruction the inner loop of round 1's SLP build
__device__ __forceinline__ void head_wfma4(fp_v2f (&a)[4], const float4& v, const fp_v2f& zz) {
#if FP_HEAD_WGRAD_FMA == 5
  const fp_v2f x01 = {v.x, v.y}, x23 = {v.z, v.w};
  asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a[0]) : "v"(x01), "v"(zz));
  asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(a[1]) : "v"(x01), "v"(zz));
  asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a[2]) : "v"(x23), "v"(zz));
  asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(a[3]) : "v"(x23), "v"(zz));
#else
  head_wfma(a[0], v.x, zz);
  head_wfma(a[1], v.y, zz);
  head_wfma(a[2], v.z, zz);
  head_wfma(a[3], v.w, zz);
#endif
}
__global__ void __launch_bounds__(256) head_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                         float* __restrict__ part, int N, int H, int W, int Cin) {
  __shared__ float red[4 * 32 * 74];
  const int Q = Cin >> 2, PPB = 256 / Q;
  const int q = threadIdx.x % Q, slot = threadIdx.x / Q;
  const int M = N * H * W;
  fp_v2f acc[9][4];
  fp_v2f bsum = {0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[t][c] = fp_v2f{0.f, 0.f};
  // all nine taps of the padded position (py, px) that input pixel m occupies; dZ(oy, ox) = dz[m + (oy - iy) * W + (ox - ix)]
  auto accum = [&](const float4& v, int m, int iy, int ix, int py, int px) __attribute__((always_inline)) {
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int oy = py - ky;
      const bool vy = (unsigned)oy < (unsigned)H;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ox = px - kx;
        const bool ok = vy && (unsigned)ox < (unsigned)W;
        const int mo = ok ? m + (oy - iy) * W + (ox - ix) : m;
        float2 z = *reinterpret_cast<const float2*>(dz + (size_t)mo * 2);
        z.x = ok ? z.x : 0.f;
        z.y = ok ? z.y : 0.f;
        const fp_v2f zz = {z.x, z.y};
        const int t = ky * 3 + kx;
        head_wfma4(acc[t], v, zz);
      }
    }
  };
  const int stride = gridDim.x * PPB;
  const int sx = stride % W, sr = stride / W, sy = sr % H;       // per-iteration advance of (ix, iy) with carries
  int m = blockIdx.x * PPB + slot;
  int ix = m % W, iy = (m / W) % H;
  // the x operand of the NEXT pixel is in flight while this one is multiplied (the loop is latency-bound: one float4 per lane and iteration)
  float4 vnext = m < M ? *reinterpret_cast<const float4*>(x + (size_t)m * Cin + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  for (; m < M; m += stride) {
    const float4 v = vnext;
    if (m + stride < M) vnext = *reinterpret_cast<const float4*>(x + (size_t)(m + stride) * Cin + q * 4);
    const float2 zc = *reinterpret_cast<const float2*>(dz + (size_t)m * 2);
    bsum[0] += zc.x;
    bsum[1] += zc.y;
    const bool inner = iy >= 2 && iy <= H - 3 && ix >= 2 && ix <= W - 3;       // all nine taps inside the image, not a mirror source
    const bool fast = __builtin_amdgcn_ballot_w64(!inner) == 0;             // the whole wave: nine loads at fixed offsets, no selects
    float2 z9[9];
    if (fast) {
      const float* zr = dz + (size_t)m * 2;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) z9[ky * 3 + kx] = *reinterpret_cast<const float2*>(zr + ((1 - ky) * W + (1 - kx)) * 2);
    } else {
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int oy = iy + 1 - ky;
        const bool vy = (unsigned)oy < (unsigned)H;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int ox = ix + 1 - kx;
          const bool ok = vy && (unsigned)ox < (unsigned)W;
          const int mo = ok ? m + (1 - ky) * W + (1 - kx) : m;
          float2 z = *reinterpret_cast<const float2*>(dz + (size_t)mo * 2);
          z.x = ok ? z.x : 0.f;
          z.y = ok ? z.y : 0.f;
          z9[ky * 3 + kx] = z;
        }
      }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const fp_v2f zz = {z9[t].x, z9[t].y};
      head_wfma4(acc[t], v, zz);
    }
    const bool my0 = iy == 1, my1 = iy == H - 2, mx0 = ix == 1, mx1 = ix == W - 2;
    if (my0 | my1 | mx0 | mx1) {                                  // mirror sources: the extra padded rows / columns this pixel also fills
      for (int a = 0; a < 3; ++a) {
        if ((a == 1 && !my0) || (a == 2 && !my1)) continue;
        const int py = a == 0 ? iy + 1 : (a == 1 ? 0 : H + 1);
        for (int b = 0; b < 3; ++b) {
          if ((a == 0 && b == 0) || (b == 1 && !mx0) || (b == 2 && !mx1)) continue;
          accum(v, m, iy, ix, py, b == 0 ? ix + 1 : (b == 1 ? 0 : W + 1));
        }
      }
    }
    ix += sx;
    iy += sy;
    if (ix >= W) { ix -= W; iy += 1; }
    if (iy >= H) iy -= H;
  }
  // reduce over the pixel slots of this wave (lanes q, q+Q, q+2Q, ...), fixed xor tree
  for (int o = Q; o < 64; o <<= 1) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
#if defined(FP_HEAD_WGRAD_PKRED)      // A/B build: the shuffle tree's additions as v_pk_add_f32 on the two ds_bpermute results (round 1's SLP form)
        const fp_v2f o2 = {__shfl_xor(acc[t][c][0], o, 64), __shfl_xor(acc[t][c][1], o, 64)};
        acc[t][c] += o2;
#else
        acc[t][c][0] += __shfl_xor(acc[t][c][0], o, 64);
        acc[t][c][1] += __shfl_xor(acc[t][c][1], o, 64);
#endif
      }
    bsum[0] += __shfl_xor(bsum[0], o, 64);
    bsum[1] += __shfl_xor(bsum[1], o, 64);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane < Q) {
    float* dst = red + (wave * 32 + lane) * 74;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
#if FP_HEAD_WGRAD_FMA == 5
        dst[(t * 4 + c) * 2 + 0] = acc[t][(c >> 1) * 2 + 0][c & 1];
        dst[(t * 4 + c) * 2 + 1] = acc[t][(c >> 1) * 2 + 1][c & 1];
#else
        dst[(t * 4 + c) * 2 + 0] = acc[t][c][0];
        dst[(t * 4 + c) * 2 + 1] = acc[t][c][1];
#endif
      }
    dst[72] = bsum[0];
    dst[73] = bsum[1];
  }
  __syncthreads();
  float* out = part + (size_t)blockIdx.x * (9 * Cin * 2 + 2);
  for (int e = threadIdx.x; e < Q * 72; e += 256) {
    const int qq = e / 72, j = e - qq * 72;
    const float s = red[(0 * 32 + qq) * 74 + j] + red[(1 * 32 + qq) * 74 + j] + red[(2 * 32 + qq) * 74 + j] +
                    red[(3 * 32 + qq) * 74 + j];
    const int t = j / 8, c = (j >> 1) & 3, o = j & 1;
    out[(t * Cin + qq * 4 + c) * 2 + o] = s;
  }
  if (threadIdx.x < 2) {
    // each wave's Q lanes hold the same wave-total bias sum (all slots reduced): take lane 0 of each wave
    const int o = threadIdx.x;
    out[9 * Cin * 2 + o] = red[(0 * 32) * 74 + 72 + o] + red[(1 * 32) * 74 + 72 + o] + red[(2 * 32) * 74 + 72 + o] +
                           red[(3 * 32) * 74 + 72 + o];
  }
}

__global__ void __launch_bounds__(256) head_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                                float* __restrict__ db, int nblk, int Cin, int accumulate) {
  const int per = 9 * Cin * 2 + 2;
  const int lane = threadIdx.x & 63;                      // one wave per output element, fixed reduction tree
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= per) return;
  float s = 0.f;
  for (int b = lane; b < nblk; b += 64) s += part[(size_t)b * per + e];
  s = fp_wave_sum(s);
  if (lane != 0) return;
  if (e < 9 * Cin * 2) {
    const int o = e & 1, tc = e >> 1, c = tc % Cin, t = tc / Cin;
    float* p = dw + ((size_t)o * Cin + c) * 9 + t;
    *p = accumulate ? *p + s : s;
  } else {
    float* p = db + (e - 9 * Cin * 2);
    *p = accumulate ? *p + s : s;
  }
}

// rows walked per workgroup: as many as max_rows while the launch still has >= 2048 workgroups (8 per CU)
HeadGrid head_grid(int N, int H, int W, int Cin, int max_rows) {
  HeadGrid g;
  g.colblocks = (int)fp_ceil_div((int64_t)W * (Cin / 4), 256);
  g.rows = max_rows;
  while (g.rows > 4 && (int64_t)g.colblocks * N * fp_ceil_div(H, g.rows) < 2048) g.rows >>= 1;
  g.gpi = (int)fp_ceil_div(H, g.rows);
  return g;
}

int head_wgrad_blocks(int64_t M, int Cin) {
  const int64_t ppb = 256 / (Cin / 4);
  int64_t b = fp_ceil_div(M, ppb * 8);
  if (b > 1024) b = 1024;
  if (b < 1) b = 1;
  return (int)b;
}

bool head_cin_ok(int Cin) { return Cin >= 4 && Cin <= MAXC && (Cin & (Cin - 1)) == 0; }

}  // namespace

extern "C" int fp_head_fwd(const float* x, const float* w_oihw, const float* bias, float* low, int32_t N, int32_t h, int32_t w,
                           int32_t Cin, int32_t apply_sigmoid, fp_stream_t stream) {
  FP_REQUIRE(x && w_oihw && bias && low, "fp_head_fwd: null pointer");
  FP_REQUIRE(head_cin_ok(Cin) && h >= 2 && w >= 2, "fp_head_fwd: Cin=%d must be a power of two in [4,128], dims >= 2", Cin);
  const HeadGrid g = head_grid(N, h, w, Cin, 16);
  fp_launch(head_fwd_kernel, dim3(g.colblocks, N * g.gpi), dim3(256), 0, (hipStream_t)stream, x, w_oihw, bias, low, h, w, Cin,
                     g.rows, g.gpi, apply_sigmoid);
  return fp_check_launch("fp_head_fwd");
}

extern "C" int fp_head_upsample(const float* low, float* out_nchw, int32_t N, int32_t h, int32_t w, int32_t scale,
                                int32_t out_channels, int32_t c0, fp_stream_t stream) {
  FP_REQUIRE(low && out_nchw && scale >= 1 && c0 + 2 <= out_channels, "fp_head_upsample: bad arguments");
  const int64_t total = (int64_t)N * h * scale * w * scale;
  int grid = (int)fp_ceil_div(total, 256);
  if (grid > 16384) grid = 16384;
  fp_launch(head_upsample_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, low, out_nchw, N, h, w, scale,
                     out_channels, c0);
  return fp_check_launch("fp_head_upsample");
}

extern "C" int fp_head_upsample_bwd(const float* dout_nchw, const float* low, float* dzlow, int32_t N, int32_t h, int32_t w,
                                    int32_t scale, int32_t out_channels, int32_t c0, int32_t apply_sigmoid, fp_stream_t stream) {
  FP_REQUIRE(dout_nchw && dzlow && scale >= 1 && c0 + 2 <= out_channels, "fp_head_upsample_bwd: bad arguments");
  FP_REQUIRE(!apply_sigmoid || low, "fp_head_upsample_bwd: sigmoid needs the saved head output");
  const int64_t total = (int64_t)N * h * w;
  const hipStream_t st = (hipStream_t)stream;
  if (scale == 2) {
    fp_launch(head_upsample_bwd_even_kernel<2>, dim3((int)fp_ceil_div(total, 256)), dim3(256), 0, st, dout_nchw, low, dzlow, N, h,
                       w, out_channels, c0, apply_sigmoid);
  } else if (scale == 4) {
    fp_launch(head_upsample_bwd_even_kernel<4>, dim3((int)fp_ceil_div(total * 4, 256)), dim3(256), 0, st, dout_nchw, low, dzlow,
                       N, h, w, out_channels, c0, apply_sigmoid);
  } else if (scale == 8) {
    fp_launch(head_upsample_bwd_even_kernel<8>, dim3((int)fp_ceil_div(total * 4, 256)), dim3(256), 0, st, dout_nchw, low, dzlow,
                       N, h, w, out_channels, c0, apply_sigmoid);
  } else {
    int grid = (int)fp_ceil_div(total, 256);
    if (grid > 16384) grid = 16384;
    fp_launch(head_upsample_bwd_kernel, dim3(grid), dim3(256), 0, st, dout_nchw, low, dzlow, N, h, w, scale, out_channels, c0,
                       apply_sigmoid);
  }
  return fp_check_launch("fp_head_upsample_bwd");
}

extern "C" int fp_head_dgrad(const float* dzlow, const float* w_oihw, const float* elu_src, float* dx, int32_t N, int32_t h,
                             int32_t w, int32_t Cin, const fp_aux* aux, fp_stream_t stream) {
  unsigned* amax_out = fp_amax_out_of(aux);
  FP_REQUIRE(dzlow && w_oihw && dx, "fp_head_dgrad: null pointer");
  FP_REQUIRE(head_cin_ok(Cin) && h >= 2 && w >= 2, "fp_head_dgrad: unsupported Cin=%d", Cin);
  const HeadGrid g = head_grid(N, h, w, Cin, 16);
  fp_launch(head_dgrad_kernel, dim3(g.colblocks, N * g.gpi), dim3(256), 0, (hipStream_t)stream, dzlow, w_oihw, elu_src, dx, h, w,
                     Cin, g.rows, g.gpi, amax_out);
  return fp_check_launch("fp_head_dgrad");
}

extern "C" int64_t fp_head_wgrad_workspace(int32_t N, int32_t h, int32_t w, int32_t Cin) {
  return (int64_t)head_wgrad_blocks((int64_t)N * h * w, Cin) * (9 * Cin * 2 + 2) * (int64_t)sizeof(float);
}

extern "C" int fp_head_wgrad(const float* x, const float* dzlow, float* dw_oihw, float* db, int32_t N, int32_t h, int32_t w,
                             int32_t Cin, int accumulate, void* workspace, int64_t workspace_bytes, fp_stream_t stream) {
  FP_REQUIRE(x && dzlow && dw_oihw && db && workspace, "fp_head_wgrad: null pointer");
  FP_REQUIRE(head_cin_ok(Cin) && h >= 2 && w >= 2, "fp_head_wgrad: unsupported Cin=%d", Cin);
  FP_REQUIRE(workspace_bytes >= fp_head_wgrad_workspace(N, h, w, Cin), "fp_head_wgrad: workspace too small");
  const int nblk = head_wgrad_blocks((int64_t)N * h * w, Cin);
  fp_launch(head_wgrad_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, x, dzlow, (float*)workspace, N, h, w, Cin);
  int rc = fp_check_launch("fp_head_wgrad");
  if (rc) return rc;
  const int per = 9 * Cin * 2 + 2;
  fp_launch(head_wgrad_reduce_kernel, dim3((int)fp_ceil_div(per, 4)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)workspace, dw_oihw, db, nblk, Cin, accumulate);
  return fp_check_launch("fp_head_wgrad(reduce)");
}
