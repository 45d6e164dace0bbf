// Microbenchmark: what does v_mfma_f32_32x32x2_f32 sustain on this part?
//   hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak && ./mfma_peak
// mode 0: pure MFMA (4 independent accumulators / wave), mode 1: + one ds_read_b128 per 4 MFMAs (the conv kernels' operand feed),
// mode 2: mode 1 + a global float4 load per 8 MFMAs.  Sweeps waves per SIMD (1..4) via the block count per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, const float* g, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = (float)i * 1e-6f;
  __syncthreads();
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float4 a = make_float4(1.f, 2.f, 3.f, 4.f), b = make_float4(.1f, .2f, .3f, .4f);
  const float* lp = lds + (threadIdx.x & 63) * 20;
  const float4* gp = reinterpret_cast<const float4*>(g) + threadIdx.x + blockIdx.x * 256;
  float4 gv = make_float4(0, 0, 0, 0);
  for (int it = 0; it < iters; ++it) {
    if (MODE >= 1) {
      a = *reinterpret_cast<const float4*>(lp + (it & 63) * 16);
      b = *reinterpret_cast<const float4*>(lp + 1280 + (it & 31) * 16);
    }
    if (MODE >= 2) { gv = gp[(it & 15) * 4096]; }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float av = c == 0 ? a.x : c == 1 ? a.y : c == 2 ? a.z : a.w;
      const float bv = c == 0 ? b.x : c == 1 ? b.y : c == 2 ? b.z : b.w;
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv + gv.x, acc[j], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(int blocks, int iters, float* out, float* g) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, g, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, g, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * 4 * iters * 16 * 4096.0;
  printf("mode %d blocks %5d (%.1f waves/SIMD) iters %d: %8.3f ms  %7.1f TF/s\n", MODE, blocks, blocks / 256.0, iters, ms, flops / ms / 1e9);
}

int main() {
  float *out, *g;
  hipMalloc(&out, 256 * 4096 * 4 * 4);
  hipMalloc(&g, 64 << 20);
  hipMemset(g, 0, 64 << 20);
  for (int w = 1; w <= 4; ++w) {
    run<0>(256 * w, 4000, out, g);
    run<1>(256 * w, 4000, out, g);
    run<2>(256 * w, 4000, out, g);
  }
  run<0>(256 * 2, 40000, out, g);   // ~1 s sustained: does the clock hold?
  run<0>(256 * 2, 40000, out, g);
  return 0;
}
