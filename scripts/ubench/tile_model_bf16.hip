// Microbenchmark: steady state of a 3-way bf16 split ("bf16x6": hh, hm, mh, hl, lh, mm) version of the halo-tile conv kernel.
// Per tap and wave: TM row blocks x 3 planes ds_read_b128 (A), 3 global float4 loads (B planes), TM x 6 v_mfma_f32_32x32x16_bf16.
// Reports fp32-equivalent TF/s (2*M*N*K per tap, as the fp32 kernel would count).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int TM, int NG, int NH, bool BAR>
__global__ void __launch_bounds__(256) k(float* out, const float* wts, const float* act, int chunks) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 6000];
  for (int i = threadIdx.x; i < 12000; i += 256) lds[i] = (float)i * 1e-6f;
  __syncthreads();
  f32x16 acc[TM];
  for (int j = 0; j < TM; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const int lane = threadIdx.x & 63;
  const float* lp = lds + (lane & 31) * 12 + (lane >> 5) * 4;      // 48-byte pixel rows: conflict-free b128
  const float4* wp = reinterpret_cast<const float4*>(wts) + (lane & 31) * 2 + (lane >> 5);
  const float4* ap = reinterpret_cast<const float4*>(act) + (size_t)blockIdx.x * 4096 + threadIdx.x;
  float4 b[3][3], hreg[3];
  for (int i = 0; i < 3; ++i) { b[0][i] = wp[i * 64]; b[1][i] = wp[256 + i * 64]; }
  for (int i = 0; i < NH; ++i) hreg[i] = ap[i * 256];
  for (int cc = 0; cc < chunks; ++cc) {
    const float* Hb = lp + (cc & 1) * 6000;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      float4 a[TM][3];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) a[i][p] = *reinterpret_cast<const float4*>(Hb + tap * 12 + i * 384 + p * 1600);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if (i == 1 || TM == 1) {
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int g = 0; g < NG; ++g) b[(tap + 2) % 3][g] = wp[((cc * 9 + tap + 2) & 63) * 256 + g * 64];
          __builtin_amdgcn_sched_barrier(0);
        }
        const bf16x8 ah = __builtin_bit_cast(bf16x8, a[i][0]), am = __builtin_bit_cast(bf16x8, a[i][1]), al = __builtin_bit_cast(bf16x8, a[i][2]);
        const bf16x8 bh = __builtin_bit_cast(bf16x8, b[tap % 3][0]), bm = __builtin_bit_cast(bf16x8, b[tap % 3][1]), bl = __builtin_bit_cast(bf16x8, b[tap % 3][2]);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[i], 0, 0, 0);
      }
    }
    if (NH > 0) {
#pragma unroll
      for (int i = 0; i < NH; ++i) *reinterpret_cast<float4*>(lds + ((cc + 1) & 1) * 6000 + threadIdx.x * 4 + i * 1024) = hreg[i];
#pragma unroll
      for (int i = 0; i < NH; ++i) hreg[i] = ap[(size_t)((cc + 1) & 15) * 768 + i * 256];
    }
    if (BAR) __syncthreads();
  }
  float s = 0.f;
  for (int j = 0; j < TM; ++j)
    for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int TM, int NG, int NH, bool BAR>
void run(const char* what, int blocks, int chunks, float* out, float* w, float* act) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<TM, NG, NH, BAR>), dim3(blocks), dim3(256), 0, 0, out, w, act, chunks);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<TM, NG, NH, BAR>), dim3(blocks), dim3(256), 0, 0, out, w, act, chunks);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * 4 * chunks * 9 * TM * 2.0 * 32 * 32 * 16;     // fp32-equivalent
  printf("%-58s blocks %5d chunks %4d: %8.3f ms  %7.1f TF/s (fp32-equivalent)\n", what, blocks, chunks, ms, flops / ms / 1e9);
}

int main() {
  float *out, *w, *act;
  hipMalloc(&out, 10000 * 256 * 4);
  hipMalloc(&w, 4 << 20);
  hipMalloc(&act, (size_t)10000 * 4096 * 16 + (1 << 24));
  hipMemset(w, 0, 4 << 20);
  hipMemset(act, 0, (size_t)10000 * 4096 * 16 + (1 << 24));
  const int B = 768;
  run<2, 0, 0, false>("TM=2: 6 ds_read + 12 MFMA / tap", B, 64, out, w, act);
  run<2, 0, 0, false>("TM=2: 6 ds_read + 12 MFMA / tap", B, 64, out, w, act);
  run<2, 3, 0, false>("TM=2: + 3 weight loads / tap", B, 64, out, w, act);
  run<2, 3, 3, true>("TM=2: + halo stream, ds_write, barrier", B, 64, out, w, act);
  run<2, 3, 3, true>("TM=2: same, 4 chunks x 12 rounds", B * 12, 4, out, w, act);
  run<4, 0, 0, false>("TM=4: 12 ds_read + 24 MFMA / tap", B, 64, out, w, act);
  run<4, 3, 0, false>("TM=4: + 3 weight loads / tap", B, 64, out, w, act);
  run<4, 3, 3, true>("TM=4: + halo stream, ds_write, barrier", B, 64, out, w, act);
  run<4, 3, 3, true>("TM=4: same, 4 chunks x 6 rounds", B * 6, 4, out, w, act);
  run<4, 3, 3, true>("TM=4: 2 WG/CU, 64 chunks", 512, 64, out, w, act);
  return 0;
}
