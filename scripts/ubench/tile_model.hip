// Microbenchmark: a synthetic model of the halo-tile conv kernel's steady state (conv3x3_tile.hip), feature by feature.
//   per "tap" (16 MFMAs / wave):  NR ds_read_b128  +  NG global float4 loads (weights, L2 resident)
//   per "chunk" (9 taps):         NH global float4 loads (halo, streamed from a big buffer) + NH ds_write_b128 + optional barrier
// Prints TF/s for several feature sets at 3 workgroups / CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NR, int NG, int NH, bool BAR, bool STREAM, int PE = 0>
__global__ void __launch_bounds__(256) k(float* out, const float* wts, const float* act, int chunks, float* out2) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 3600];
  if (PE & 1) {   // realistic prologue: dependent global load -> LDS store -> barrier before the first MFMA
    const float4* pp = reinterpret_cast<const float4*>(act) + (size_t)blockIdx.x * 4096 + 2048 + threadIdx.x;
    float4 v0 = pp[0], v1 = pp[256], v2 = pp[512];
    *reinterpret_cast<float4*>(lds + threadIdx.x * 4) = v0;
    *reinterpret_cast<float4*>(lds + 1024 + threadIdx.x * 4) = v1;
    *reinterpret_cast<float4*>(lds + 2048 + threadIdx.x * 4) = v2;
    for (int i = 3072 + threadIdx.x; i < 7200; i += 256) lds[i] = 0.f;
  } else {
    for (int i = threadIdx.x; i < 7200; i += 256) lds[i] = (float)i * 1e-6f;
  }
  __syncthreads();
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const int lane = threadIdx.x & 63;
  const float* lp = lds + (lane & 31) * 20 + (lane >> 5) * 4;
#ifndef WPAT
#define WPAT 0
#endif
  const float4* wp = reinterpret_cast<const float4*>(wts) + (WPAT == 0 ? (lane & 31) * 4 + (lane >> 5) : (WPAT == 1 ? lane : (lane & 31) * 4 + (lane >> 5) + (threadIdx.x >> 6) * 8192));
#ifndef WSTEP
#define WSTEP 2
#endif
  const float4* ap = reinterpret_cast<const float4*>(act) + (size_t)blockIdx.x * 4096 + threadIdx.x;
  float4 b[3][2], hreg[3];
  for (int i = 0; i < 2; ++i) { b[0][i] = wp[i * 2]; b[1][i] = wp[128 + i * 2]; }
  for (int i = 0; i < NH; ++i) hreg[i] = ap[i * 256];
  for (int cc = 0; cc < chunks; ++cc) {
    const float* Hb = lp + (cc & 1) * 3600;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      float4 a[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = i < NR ? *reinterpret_cast<const float4*>(Hb + tap * 20 + i * 640 + (i & 1) * 8) : make_float4(1, 2, 3, 4);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
#ifndef SPLIT
#define SPLIT 0
#endif
        if (SPLIT == 0 && c == 1) {
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < NG; ++i) b[(tap + 2) % 3][i] = wp[((cc * 9 + tap + 2) & 63) * 128 + i * (WPAT == 1 ? 64 : WSTEP)];
          __builtin_amdgcn_sched_barrier(0);
        }
        if (SPLIT == 1 && c >= 1 && c - 1 < NG) {      // one load per MFMA group
          __builtin_amdgcn_sched_barrier(0);
          b[(tap + 2) % 3][c - 1] = wp[((cc * 9 + tap + 2) & 63) * 128 + (c - 1) * (WPAT == 1 ? 64 : WSTEP)];
          __builtin_amdgcn_sched_barrier(0);
        }
        if (SPLIT == 2 && c == 1) {                    // no pinning: let the compiler place the loads
#pragma unroll
          for (int i = 0; i < NG; ++i) b[(tap + 2) % 3][i] = wp[((cc * 9 + tap + 2) & 63) * 128 + i * (WPAT == 1 ? 64 : WSTEP)];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 av = a[j], bv = b[tap % 3][j >> 1];
          const float x = c == 0 ? av.x : c == 1 ? av.y : c == 2 ? av.z : av.w;
          const float y = c == 0 ? bv.x : c == 1 ? bv.y : c == 2 ? bv.z : bv.w;
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[j], 0, 0, 0);
        }
      }
    }
    if (NH > 0) {
#pragma unroll
      for (int i = 0; i < NH; ++i) *reinterpret_cast<float4*>(lds + ((cc + 1) & 1) * 3600 + threadIdx.x * 4 + i * 1024) = hreg[i];
#pragma unroll
      for (int i = 0; i < NH; ++i) hreg[i] = ap[(STREAM ? (size_t)((cc + 1) & 15) * 768 : 0) + i * 256];
    }
    if (BAR) __syncthreads();
  }
  if (PE & 2) {   // realistic epilogue: 128 x 64 outputs, ELU, 128-byte rows
    float* o = out2 + (size_t)blockIdx.x * 8192 + (threadIdx.x >> 6) * 2048 + (lane >> 5) * 128 + (lane & 31);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[j][r] + acc[j + 2][r] + 0.5f;
#ifndef ELU
#define ELU 1
#endif
        if (ELU == 1) v = v > 0.f ? v : expm1f(v);
        if (ELU == 2) v = v > 0.f ? v : __expf(v) - 1.f;
        o[j * 1024 + (r >> 2) * 256 + (r & 3) * 32] = v;
      }
  } else {
    float s = 0.f;
    for (int j = 0; j < 4; ++j)
      for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  }
}

float* g_out2;
template <int NR, int NG, int NH, bool BAR, bool STREAM, int PE = 0>
void run(const char* what, int blocks, int chunks, float* out, float* w, float* act) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NR, NG, NH, BAR, STREAM, PE>), dim3(blocks), dim3(256), 0, 0, out, w, act, chunks, g_out2);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NR, NG, NH, BAR, STREAM, PE>), dim3(blocks), dim3(256), 0, 0, out, w, act, chunks, g_out2);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * 4 * chunks * 9 * 16 * 4096.0;
  printf("%-58s blocks %5d chunks %4d: %8.3f ms  %7.1f TF/s\n", what, blocks, chunks, ms, flops / ms / 1e9);
}

int main() {
  float *out, *w, *act;
  hipMalloc(&out, 10000 * 256 * 4);
  hipMalloc(&w, 1 << 20);
  hipMalloc(&act, (size_t)10000 * 4096 * 16 + (1 << 24));
  hipMemset(w, 0, 1 << 20);
  hipMemset(act, 0, (size_t)10000 * 4096 * 16 + (1 << 24));
  hipMalloc(&g_out2, (size_t)10000 * 8192 * 4);
  const int B = 768;
  run<4, 2, 3, true, true, 0>("4 chunks, 12 rounds, no prologue/epilogue", 768 * 12, 4, out, w, act);
  run<4, 2, 3, true, true, 0>("4 chunks, 12 rounds, no prologue/epilogue", 768 * 12, 4, out, w, act);
  run<4, 2, 3, true, true, 2>("4 chunks, 12 rounds + epilogue", 768 * 12, 4, out, w, act);
  run<4, 2, 3, true, true, 2>("4 chunks, 12 rounds + epilogue", 768 * 12, 4, out, w, act);
  return 0;
}
