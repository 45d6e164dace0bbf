// which MFMA form sustains the most on pseudo-random operands?  same harness as mfma_sustained.hip (768 workgroups, two chains per wave, ~2 s each)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int FORM>
__global__ void __launch_bounds__(256) k(float* out, int iters, unsigned long long* clk) {
  uint4 av[8], bv[8];
  unsigned x = 0x9E3779B9u * (threadIdx.x + 1) + 0x85EBCA6Bu * (blockIdx.x + 1);
  for (int q = 0; q < 8; ++q) {
    unsigned w[8];
    for (int e = 0; e < 8; ++e) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; w[e] = FORM == 2 ? ((x & 0x83ff83ffu) | 0x38003800u) : ((x & 0x807f807fu) | 0x3f003f00u); }
    av[q] = make_uint4(w[0], w[1], w[2], w[3]); bv[q] = make_uint4(w[4], w[5], w[6], w[7]);
  }
  unsigned long long c0 = 0, r0 = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) { c0 = __builtin_readcyclecounter(); r0 = __builtin_amdgcn_s_memrealtime(); }
  float s = 0.f;
  if (FORM == 1) {                       // 16x16x32 bf16: 16384 MACs, 4 accumulator registers; eight chains
    f32x4 acc[8];
    for (int j = 0; j < 8; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int rep = 0; rep < 6; ++rep)
#pragma unroll
        for (int j = 0; j < 8; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av[(rep + j) & 7]), __builtin_bit_cast(bf16x8, bv[(rep * 3 + j) & 7]), acc[j], 0, 0, 0);
    for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][3];
  } else {
    f32x16 acc[2];
    for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int rep = 0; rep < 12; ++rep)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (FORM == 0) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[(rep * 2 + j) & 7]), __builtin_bit_cast(bf16x8, bv[(rep * 5 + j * 3) & 7]), acc[j], 0, 0, 0);
          else acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av[(rep * 2 + j) & 7]), __builtin_bit_cast(f16x8, bv[(rep * 5 + j * 3) & 7]), acc[j], 0, 0, 0);
        }
    for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  }
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_readcyclecounter() - c0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
}

template <int FORM>
void run(const char* name, double flop_per_iter_per_wave, double seconds) {
  const int blocks = 768, iters = 20000;
  float* out; unsigned long long* clk;
  hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, 16);
  int khz = 0; hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double flop = (double)blocks * 4 * iters * flop_per_iter_per_wave;
  double total = 0, tf = 0, mhz = 0; int n = 0; unsigned long long h[2];
  while (total < seconds * 1e3) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<FORM>, dim3(blocks), dim3(256), 0, 0, out, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    tf = flop / ms / 1e9; mhz = (double)h[0] / (double)h[1] * khz / 1e3; total += ms; ++n;
  }
  printf("%-28s pseudo-random operands, last of %3d launches: %7.1f TFLOP/s at %5.0f MHz\n", name, n, tf, mhz);
  hipFree(out); hipFree(clk);
}

int main() {
  run<0>("v_mfma_f32_32x32x16_bf16", 24 * 2.0 * 32 * 32 * 16, 2.0);
  run<1>("v_mfma_f32_16x16x32_bf16", 48 * 2.0 * 16 * 16 * 32, 2.0);
  run<2>("v_mfma_f32_32x32x16_f16", 24 * 2.0 * 32 * 32 * 16, 2.0);
  run<0>("v_mfma_f32_32x32x16_bf16", 24 * 2.0 * 32 * 32 * 16, 1.0);
  return 0;
}
