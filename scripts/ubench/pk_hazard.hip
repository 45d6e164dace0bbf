// Standalone check of the "packed-fp32 VALU next to another wave's bf16 MFMA returns wrong sums" observation
// (profiles/round1_notes.md).  No library code, no LDS, no cross-lane operations in the victims:
//   victim A  pk_fma : every thread accumulates 8 float2 sums with v_pk_fma_f32 (inline asm) over a stream of loads
//   victim B  pk_add : out[i] = a[i] + b[i] with v_pk_add_f32 -- what a collective's reduction kernel does
//   victim C  fma    : control, the same as A with scalar v_fma_f32
// aggressors: a v_mfma_f32_32x32x16_bf16 spinner, a v_mfma_f32_32x32x2_f32 spinner (control), none.
// Each victim runs REPS times next to the aggressor (two streams, aggressor sized to stay resident on every CU with room for the
// victim's waves) and its output is compared bitwise with the output of the first solo run.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(256) agg_bf16(float* out, int iters) {
  f32x16 acc[2];
  for (int j = 0; j < 2; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  uint4 au = make_uint4(0x3f803f80u + threadIdx.x, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u), bu = make_uint4(0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u + blockIdx.x);
  const bf16x8 a = __builtin_bit_cast(bf16x8, au), b = __builtin_bit_cast(bf16x8, bu);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 8; ++rep) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[1], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int j = 0; j < 2; ++j)
    for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) agg_f32(float* out, int iters) {
  f32x16 acc[2];
  for (int j = 0; j < 2; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f + blockIdx.x * 1e-4f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[1], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int j = 0; j < 2; ++j)
    for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <bool PK>
__global__ void __launch_bounds__(256) vic_fma(const float4* __restrict__ x, const float2* __restrict__ z, float* __restrict__ out, int n_per_thread) {
  f32x2 acc[8];
  for (int j = 0; j < 8; ++j) acc[j] = f32x2{0.f, 0.f};
  const size_t base = (size_t)(blockIdx.x * 256 + threadIdx.x);
  const size_t stride = (size_t)gridDim.x * 256;
  for (int i = 0; i < n_per_thread; ++i) {
    const float4 v = x[base + i * stride];
    const float2 zz = z[(base + i * stride) & 0xfffff];
    const f32x2 zv = f32x2{zz.x, zz.y};
    const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const f32x2 vb = f32x2{vv[c], vv[c]};
      if (PK) {
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(vb), "v"(zv));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[4 + c]) : "v"(zv), "v"(vb));
      } else {
        acc[c][0] = __builtin_fmaf(vb[0], zv[0], acc[c][0]);
        acc[c][1] = __builtin_fmaf(vb[1], zv[1], acc[c][1]);
        acc[4 + c][0] = __builtin_fmaf(zv[0], vb[0], acc[4 + c][0]);
        acc[4 + c][1] = __builtin_fmaf(zv[1], vb[1], acc[4 + c][1]);
      }
    }
  }
  for (int j = 0; j < 8; ++j) {
    out[(base * 8 + j) * 2 + 0] = acc[j][0];
    out[(base * 8 + j) * 2 + 1] = acc[j][1];
  }
}

// victim D: the source pattern of head_wgrad_kernel's inner loop, left to the compiler (this file is built WITH the SLP vectoriser,
// so adjacent scalar fp32 multiply-adds become v_pk_fma_f32 with op_sel broadcast forms -- exactly what the library's first build had)
__global__ void __launch_bounds__(256) vic_slp(const float4* __restrict__ x, const float2* __restrict__ z, float* __restrict__ out, int n_per_thread) {
  float acc[9][4][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[t][c][0] = acc[t][c][1] = 0.f;
  const size_t base = (size_t)(blockIdx.x * 256 + threadIdx.x);
  const size_t stride = (size_t)gridDim.x * 256;
  for (int i = 0; i + 9 <= n_per_thread; i += 9) {
    const float2 zz = z[(base + i * stride) & 0xfffff];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const float4 v = x[base + (i + t) * stride];
      acc[t][0][0] += v.x * zz.x; acc[t][0][1] += v.x * zz.y;
      acc[t][1][0] += v.y * zz.x; acc[t][1][1] += v.y * zz.y;
      acc[t][2][0] += v.z * zz.x; acc[t][2][1] += v.z * zz.y;
      acc[t][3][0] += v.w * zz.x; acc[t][3][1] += v.w * zz.y;
    }
  }
  // same wave reduction as head_wgrad (lanes q, q + 8, ...: fixed xor tree)
  for (int o = 8; o < 64; o <<= 1) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        acc[t][c][0] += __shfl_xor(acc[t][c][0], o, 64);
        acc[t][c][1] += __shfl_xor(acc[t][c][1], o, 64);
      }
  }
  float* dst = out + base * 72;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      dst[(t * 4 + c) * 2 + 0] = acc[t][c][0];
      dst[(t * 4 + c) * 2 + 1] = acc[t][c][1];
    }
}

__global__ void __launch_bounds__(256) vic_pk_add(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ out, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 p = a[i], q = b[i];
    f32x2 r0, r1;
    const f32x2 p0 = f32x2{p.x, p.y}, p1 = f32x2{p.z, p.w}, q0 = f32x2{q.x, q.y}, q1 = f32x2{q.z, q.w};
    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r0) : "v"(p0), "v"(q0));
    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r1) : "v"(p1), "v"(q1));
    out[i] = make_float4(r0[0], r0[1], r1[0], r1[1]);
  }
}

static std::vector<float> fetch(const float* d, size_t n) {
  std::vector<float> h(n);
  CK(hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost));
  return h;
}

int main(int argc, char** argv) {
  const int REPS = argc > 1 ? atoi(argv[1]) : 6;
  const size_t NX = (size_t)48 << 20;            // float4 elements of x (768 MB)
  float4 *x, *b4, *o4; float2* z; float *out, *aggout;
  CK(hipMalloc(&x, NX * 16)); CK(hipMalloc(&b4, NX * 16)); CK(hipMalloc(&o4, NX * 16));
  CK(hipMalloc(&z, (1 << 20) * 8)); CK(hipMalloc(&out, (size_t)2048 * 256 * 72 * 4)); CK(hipMalloc(&aggout, 4096 * 256 * 4));
  {
    std::vector<float> h(NX * 4);
    unsigned s = 12345u;
    for (size_t i = 0; i < h.size(); ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }
    CK(hipMemcpy(x, h.data(), NX * 16, hipMemcpyHostToDevice));
    for (size_t i = 0; i < h.size(); ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) & 0xffff) / 65536.0f * 1e-3f; }
    CK(hipMemcpy(b4, h.data(), NX * 16, hipMemcpyHostToDevice));
    CK(hipMemcpy(z, h.data(), (1 << 20) * 8, hipMemcpyHostToDevice));
  }
  hipStream_t sv, sa;
  CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  const int VB = 2048, NPT = (int)(NX / ((size_t)VB * 256));
  const size_t n_fma_out = (size_t)VB * 256 * 16;
  struct Vic { const char* name; int kind; } vics[4] = {{"pk_fma (v_pk_fma_f32 accumulate)", 0}, {"pk_add (v_pk_add_f32 elementwise, reduction-kernel style)", 1}, {"fma (scalar v_fma_f32 control)", 2},
                                                       {"slp (head_wgrad's loop, compiler-packed + shfl tree)", 3}};
  const char* aggn[3] = {"none", "bf16 MFMA (v_mfma_f32_32x32x16_bf16)", "fp32 MFMA (v_mfma_f32_32x32x2_f32)"};
  auto launch_vic = [&](int kind) {
    if (kind == 0) hipLaunchKernelGGL(vic_fma<true>, dim3(VB), dim3(256), 0, sv, x, z, out, NPT);
    else if (kind == 2) hipLaunchKernelGGL(vic_fma<false>, dim3(VB), dim3(256), 0, sv, x, z, out, NPT);
    else if (kind == 3) hipLaunchKernelGGL(vic_slp, dim3(VB), dim3(256), 0, sv, x, z, out, NPT);
    else hipLaunchKernelGGL(vic_pk_add, dim3(VB), dim3(256), 0, sv, x, b4, o4, NX);
  };
  int total_bad = 0;
  for (auto& v : vics) {
    launch_vic(v.kind);
    CK(hipStreamSynchronize(sv));
    const size_t n = v.kind == 1 ? NX * 4 : (v.kind == 3 ? (size_t)VB * 256 * 72 : n_fma_out);
    const float* dptr = v.kind == 1 ? (const float*)o4 : out;
    std::vector<float> ref = fetch(dptr, n);
    for (int ag = 0; ag < 3; ++ag) {
      int differ = 0; size_t words = 0;
      for (int r = 0; r < REPS; ++r) {
        CK(hipMemsetAsync((void*)dptr, 0xff, n * 4, sv));
        CK(hipStreamSynchronize(sv));
        // one workgroup of the aggressor per CU (x2): it stays resident for the whole victim run and leaves room for victim waves
        if (ag == 1) hipLaunchKernelGGL(agg_bf16, dim3(512), dim3(256), 0, sa, aggout, 6000);
        if (ag == 2) hipLaunchKernelGGL(agg_f32, dim3(512), dim3(256), 0, sa, aggout, 6000);
        launch_vic(v.kind);
        CK(hipStreamSynchronize(sv));
        CK(hipStreamSynchronize(sa));
        std::vector<float> got = fetch(dptr, n);
        size_t w = 0;
        for (size_t i = 0; i < n; ++i) w += memcmp(&got[i], &ref[i], 4) != 0;
        differ += w != 0; words += w;
      }
      printf("victim %-62s aggressor %-40s: %d of %d runs differ (%zu words)\n", v.name, aggn[ag], differ, REPS, words);
      total_bad += differ;
    }
  }
  // timing sanity: did the kernels really overlap?  victim alone vs with the bf16 aggressor
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms0, ms1;
  CK(hipEventRecord(e0, sv)); launch_vic(0); CK(hipEventRecord(e1, sv)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms0, e0, e1));
  hipLaunchKernelGGL(agg_bf16, dim3(512), dim3(256), 0, sa, aggout, 6000);
  CK(hipEventRecord(e0, sv)); launch_vic(0); CK(hipEventRecord(e1, sv)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms1, e0, e1));
  CK(hipEventRecord(e0, sa)); CK(hipDeviceSynchronize());
  hipLaunchKernelGGL(agg_bf16, dim3(512), dim3(256), 0, sa, aggout, 6000);
  CK(hipEventRecord(e1, sa)); CK(hipEventSynchronize(e1));
  float msa; CK(hipEventElapsedTime(&msa, e0, e1));
  printf("victim pk_fma alone %.3f ms, next to the bf16 aggressor %.3f ms; aggressor alone ~%.3f ms\n", ms0, ms1, msa);
  printf("RESULT: %s\n", total_bad ? "DIFFERENCES OBSERVED" : "no differences in any combination");
  return 0;
}
