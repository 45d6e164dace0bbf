// v_mfma_f32_32x32x16_bf16 back to back for a given time: does the chip hold its burst rate?  (round 6: the tile kernel runs at 1.65 GHz)
// three waves per SIMD, two independent accumulator chains per wave; one launch of `iters` x 24 MFMAs per wave, repeated for ~T seconds;
// reports TFLOP/s of the first launch (cold), of the last one, and the clock from s_memtime / s_memrealtime of wave 0 around the whole run.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ void __launch_bounds__(256) k(float* out, int iters, unsigned long long* clk, int mode) {
  f32x16 acc[2];
  for (int j = 0; j < 2; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  // MODE 0: constant operands (every lane, every MFMA the same bits); MODE 1: eight operand pairs of pseudo-random bf16 bit patterns per lane,
  // cycled without any vector instruction in the loop -- the operand buses toggle like they do on real activations
  bf16x8 av[8], bv[8];
  {
    unsigned x = 0x9E3779B9u * (threadIdx.x + 1) + 0x85EBCA6Bu * (blockIdx.x + 1);
    for (int q = 0; q < 8; ++q) {
      unsigned w[8];
      for (int e = 0; e < 8; ++e) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; w[e] = mode ? ((x & 0x807f807fu) | 0x3f003f00u) : (e < 4 ? 0x3f803f80u : 0x3c003c00u); }
      av[q] = __builtin_bit_cast(bf16x8, make_uint4(w[0], w[1], w[2], w[3]));
      bv[q] = __builtin_bit_cast(bf16x8, make_uint4(w[4], w[5], w[6], w[7]));
    }
  }
  unsigned long long c0 = 0, r0 = 0;
  if (clk && blockIdx.x == 0 && threadIdx.x == 0) { c0 = __builtin_readcyclecounter(); r0 = __builtin_amdgcn_s_memrealtime(); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 12; ++rep)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        // mode 2: A fixed for the whole unrolled block (only B toggles); mode 3: both operands change only every fourth MFMA
        const int ia = mode == 2 ? 0 : (mode == 3 ? ((rep * 2 + j) >> 2) & 7 : (rep * 2 + j) & 7);
        const int ib = mode == 3 ? ((rep * 2 + j) >> 2) * 3 & 7 : (rep * 5 + j * 3) & 7;
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[ia], bv[ib], acc[j], 0, 0, 0);
      }
  }
  float s = 0.f;
  for (int j = 0; j < 2; ++j)
    for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (clk && blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_readcyclecounter() - c0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 2.0;
  const int mode = argc > 2 ? atoi(argv[2]) : 0;
  const int blocks = 256 * 3, iters = 20000;          // ~10 ms per launch at the burst rate
  float* out; unsigned long long* clk;
  hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, 16);
  int khz = 0; hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double flop = (double)blocks * 4 * iters * 24 * 2.0 * 32 * 32 * 16;
  double first = 0, last = 0, total = 0; int n = 0;
  unsigned long long h[2];
  while (total < seconds * 1e3) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, clk, mode);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double tf = flop / ms / 1e9, mhz = (double)h[0] / (double)h[1] * khz / 1e3;
    if (n == 0) first = tf;
    last = tf; total += ms; ++n;
    if (n <= 3 || n % 25 == 0) printf("launch %3d: %7.2f ms  %7.1f TF/s bf16  shader clock %5.0f MHz\n", n, ms, tf, mhz);
  }
  printf("mode %d (%s): ", mode, mode == 0 ? "constant operands" : mode == 1 ? "pseudo-random operands, both change every MFMA" : mode == 2 ? "pseudo-random, A fixed, B changes every MFMA" : "pseudo-random, both change every fourth MFMA"); printf("first %.1f TF/s, last %.1f TF/s after %.1f s of back-to-back MFMAs (%d launches)\n", first, last, total / 1e3, n);
  return 0;
}
