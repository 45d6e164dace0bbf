// v_mfma_f32_32x32x16_bf16: throughput vs number of independent accumulator chains per wave and waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NCH>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
  f32x16 acc[NCH];
  for (int j = 0; j < NCH; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  uint4 au = make_uint4(0x3f803f80u + threadIdx.x, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u), bu = make_uint4(0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u + blockIdx.x);
  const bf16x8 a = __builtin_bit_cast(bf16x8, au), b = __builtin_bit_cast(bf16x8, bu);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 24 / NCH; ++rep)
#pragma unroll
      for (int j = 0; j < NCH; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
  }
  float s = 0.f;
  for (int j = 0; j < NCH; ++j)
    for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NCH>
void run(int blocks, int iters, float* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NCH>, dim3(blocks), dim3(256), 0, 0, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NCH>, dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double n_mfma = (double)blocks * 4 * iters * 24;
  const double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * 24) ;   // per MFMA per wave (wall cycles / MFMAs issued by ONE wave)
  printf("chains %d, %.0f waves/SIMD: %7.3f ms  %7.1f TF/s bf16  (%5.1f cycles per MFMA per wave)\n", NCH, blocks / 256.0, ms,
         n_mfma * 2.0 * 32 * 32 * 16 / ms / 1e9, cyc);
}

int main() {
  float* out;
  hipMalloc(&out, 4096 * 256 * 4);
  for (int w = 1; w <= 3; ++w) {
    run<1>(256 * w, 2000, out);
    run<2>(256 * w, 2000, out);
    run<3>(256 * w, 2000, out);
    run<4>(256 * w, 2000, out);
    run<6>(256 * w, 2000, out);
  }
  return 0;
}
