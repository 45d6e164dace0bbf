// What lowers the clock under the tile kernel: the MFMAs, or what runs beside them?  v_mfma_f32_32x32x16_bf16 on pseudo-random operands (three waves
// per SIMD, two accumulator chains) with, per MFMA, NV independent v_fma_f32 and NL conflict-free ds_read_b128 interleaved -- the tile kernel's
// own ratios are ~4.5 VALU and ~0.5 LDS reads per MFMA.  Reports the MFMA rate (TFLOP/s of the MFMAs alone) and the shader clock after ~2 s.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NV, int NL2>      // NV VALU per MFMA; NL2 ds_read_b128 per TWO MFMAs
__global__ void __launch_bounds__(256) k(float* out, int iters, unsigned long long* clk) {
  __shared__ uint4 lds[2048];
  for (int e = threadIdx.x; e < 2048; e += 256) lds[e] = make_uint4(e * 2654435761u, e * 40503u, ~e, e ^ 0x5bd1e995u);
  __syncthreads();
  f32x16 acc[2];
  for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  bf16x8 av[8], bv[8];
  unsigned x = 0x9E3779B9u * (threadIdx.x + 1) + 0x85EBCA6Bu * (blockIdx.x + 1);
  for (int q = 0; q < 8; ++q) {
    unsigned w[8];
    for (int e = 0; e < 8; ++e) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; w[e] = (x & 0x807f807fu) | 0x3f003f00u; }
    av[q] = __builtin_bit_cast(bf16x8, make_uint4(w[0], w[1], w[2], w[3]));
    bv[q] = __builtin_bit_cast(bf16x8, make_uint4(w[4], w[5], w[6], w[7]));
  }
  float f[8];
  for (int e = 0; e < 8; ++e) f[e] = 1.0f + 1e-3f * (threadIdx.x + e);
  const float mul = 0.99991f, add = 1e-4f * threadIdx.x;
  uint4 junk = make_uint4(0, 0, 0, 0);
  const int lane = threadIdx.x & 63;
  unsigned long long c0 = 0, r0 = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) { c0 = __builtin_readcyclecounter(); r0 = __builtin_amdgcn_s_memrealtime(); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 12; ++rep) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[(rep * 2 + j) & 7], bv[(rep * 5 + j * 3) & 7], acc[j], 0, 0, 0);
#pragma unroll
        for (int v = 0; v < NV; ++v) f[(v + j * 4) & 7] = __builtin_fmaf(f[(v + j * 4) & 7], mul, add);
      }
#pragma unroll
      for (int l = 0; l < NL2; ++l) {
        const uint4 t = lds[(lane + 64 * ((rep * NL2 + l + it) & 31)) & 2047];     // 64 consecutive 16-byte slots: conflict-free
        junk.x ^= t.x; junk.y ^= t.y; junk.z ^= t.z; junk.w ^= t.w;
      }
    }
  }
  float s = 0.f;
  for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  for (int e = 0; e < 8; ++e) s += f[e];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)(junk.x ^ junk.y ^ junk.z ^ junk.w);
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_readcyclecounter() - c0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
}

template <int NV, int NL2>
void run(double seconds) {
  const int blocks = 768, iters = 8000;
  float* out; unsigned long long* clk;
  hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, 16);
  int khz = 0; hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double flop = (double)blocks * 4 * iters * 24 * 2.0 * 32 * 32 * 16;
  double total = 0, tf = 0, mhz = 0; int n = 0; unsigned long long h[2];
  while (total < seconds * 1e3) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, NL2>), dim3(blocks), dim3(256), 0, 0, out, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    tf = flop / ms / 1e9; mhz = (double)h[0] / (double)h[1] * khz / 1e3; total += ms; ++n;
  }
  // MFMA issue slots: one MFMA per 32 cycles per SIMD at full rate -> busy fraction = tf / (1024 SIMDs x 1024 FLOP per cycle x clock)
  printf("per MFMA: %d VALU + %.1f ds_read_b128   MFMA rate %7.1f TFLOP/s  clock %5.0f MHz  matrix pipe busy %5.1f %%\n", NV, NL2 / 2.0, tf, mhz,
         100.0 * tf * 1e12 / (1024.0 * 1024.0 * mhz * 1e6));
  hipFree(out); hipFree(clk);
}

int main() {
  run<0, 0>(2.0);
  run<2, 0>(2.0);
  run<4, 0>(2.0);
  run<6, 0>(2.0);
  run<0, 1>(2.0);
  run<0, 2>(2.0);
  run<4, 1>(2.0);
  run<0, 0>(1.0);
  return 0;
}
