// ARCHIVE (not built, not linked): round 3's train-mode BatchNorm with in-kernel cross-workgroup dependencies -- one launch (or two) per
// layer and direction instead of three.  Correct and bit-reproducible (its tests passed on MI355X), and SLOWER in the training step
// (16.2-16.7 ms fused, 14.7 ms ticket form, 13.67 ms three launches): the XCDs' L2s are not coherent, every exchange goes through memory
// at ~0.7 us per dependent hop, and a grid small enough for a short combine cannot pull HBM bandwidth (profiles/round3_notes.md, "Fusing
// launches with in-kernel cross-workgroup dependencies").  Retired from the library in round 4 (the spinning forms are launched
// with a plain hipLaunchKernel, nothing guarantees that all their workgroups are resident -- two of them on two streams can deadlock).
// Kept here as the record of the negative result: the fp_gs_* primitives (fp_common.h), the kernels and their entry points (bn_pool.hip).

// ======== fp_common.h part ========
// ---- in-kernel grid synchronisation (fused BatchNorm, fused split reductions) ---------------------------------------------------
// A kernel whose workgroups are all co-resident (grid <= a few workgroups per CU) can contain a grid-wide dependency: every
// workgroup publishes a partial result, takes a ticket, and the LAST one to arrive combines the partials in a fixed order (so the
// result does not depend on who was last: run-to-run bit-identical) and raises a flag the others spin on -- one launch instead of
// three on the encoder's serial spine, where a dependent launch costs as much as these small kernels themselves.
//  * Visibility without cache flushes: partials are written with agent-scope relaxed atomic stores (sc1: written through to
//    memory; the per-XCD L2s are not coherent with each other), the writer waits for vmcnt(0) before its ticket, readers use
//    agent-scope atomic loads (sc1: never served from a stale L1 / L2 line).  -DFP_GSYNC_FORMAL builds the same protocol with
//    agent-scope release / acquire fences (buffer_wbl2 / buffer_inv) instead, for A/B runs.
//  * Tickets: same-address agent-scope atomics execute at the memory side and serialise at ~90 ns each, so arrivals go up a tree
//    of fan-in 8 (distinct addresses proceed in parallel): ~0.7 us per level.  A node's last arriver re-arms it, the last
//    workgroup to leave re-arms the flag: the sync block is zeroed once by its owner and reusable by every later launch on the
//    same stream.
constexpr int FP_GSYNC_TREE = 640;                      // counters of one arrival tree: up to 4096 workgroups (512 + 64 + 8 + 1)
constexpr int FP_GSYNC_WORDS = 2 * FP_GSYNC_TREE + 64;  // arrival tree, exit tree, flag (its own 128-byte line)
__device__ __forceinline__ void fp_gs_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float fp_gs_load(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// every thread, after its fp_gs_store calls and before the workgroup's ticket
__device__ __forceinline__ void fp_gs_publish() {
#ifdef FP_GSYNC_FORMAL
  __atomic_thread_fence(__ATOMIC_RELEASE);              // agent scope: buffer_wbl2 sc1 + s_waitcnt
#else
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  __syncthreads();
}
// every thread of the workgroup that is about to read what others published
__device__ __forceinline__ void fp_gs_acquire() {
#ifdef FP_GSYNC_FORMAL
  __atomic_thread_fence(__ATOMIC_ACQUIRE);              // agent scope: buffer_inv sc1
#endif
}
// one thread per workgroup: true for exactly one workgroup of the grid, the last to call
__device__ __forceinline__ bool fp_gs_ticket(unsigned* tree, int wg, int nwg) {
  int idx = wg, n = nwg, off = 0;
  while (n > 1) {
    const int g = idx >> 3, gsize = min(8, n - (g << 3));
    unsigned* c = tree + off + g;
    const unsigned old = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old != (unsigned)(gsize - 1)) return false;
    __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int ng = (n + 7) >> 3;
    off += ng;
    idx = g;
    n = ng;
  }
  return true;
}
// whole workgroup: did THIS workgroup arrive last?  (call after fp_gs_publish)
__device__ __forceinline__ bool fp_gs_arrive_last(unsigned* sync, int wg, int nwg) {
  __shared__ int fp_gs_last;
  if (threadIdx.x == 0) fp_gs_last = fp_gs_ticket(sync, wg, nwg) ? 1 : 0;
  __syncthreads();
  const bool last = fp_gs_last != 0;
  if (last) fp_gs_acquire();
  return last;
}
__device__ __forceinline__ unsigned* fp_gs_flag(unsigned* sync) { return sync + 2 * FP_GSYNC_TREE; }
// last arriver, after its own fp_gs_publish of the combined result
__device__ __forceinline__ void fp_gs_release_all(unsigned* sync) {
  if (threadIdx.x == 0) __hip_atomic_store(fp_gs_flag(sync), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// every workgroup: wait until the combined result is published
__device__ __forceinline__ void fp_gs_wait(unsigned* sync) {
  if (threadIdx.x == 0)
    while (__hip_atomic_load(fp_gs_flag(sync), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(4);
  __syncthreads();
  fp_gs_acquire();
}
// every workgroup, once it no longer needs the flag: the last one out re-arms it
__device__ __forceinline__ void fp_gs_leave(unsigned* sync, int wg, int nwg) {
  if (threadIdx.x == 0 && fp_gs_ticket(sync + FP_GSYNC_TREE, wg, nwg))
    __hip_atomic_store(fp_gs_flag(sync), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ======== bn_pool.hip kernels ========
// ---- fused train-mode BatchNorm: ONE launch per layer and direction (statistics -> combine -> apply inside the kernel) ----------
// The three-launch form above costs the encoder's serial spine a dependent launch per stage (~10 us each, as much as these
// kernels themselves: 216 launches / 2.3 ms per KITTI step).  Here a workgroup owns a CONTIGUOUS block of rows: it reduces
// them, publishes its partial, and -- once the last workgroup to arrive has combined all partials in a fixed order and published
// the per-channel coefficients (fp_gs_* in fp_common.h) -- normalises the same rows, which are still in its XCD's L2.  The grid
// is sized so that every workgroup is resident (<= 256, >= ~64 KB of rows each); results do not depend on arrival order.
struct BnGeom {
  int M, C, rows_per_wg;
};
constexpr int BN_UNROLL = 8;

int bn_fused_grid(int64_t M, int C, int* rows_per_wg) {
  const int R = 256 / (C / 4);
  static const int kb = getenv("FP_BN_FUSED_KB") ? atoi(getenv("FP_BN_FUSED_KB")) : 32;       // bytes of z per workgroup (KB), lower bound
  static const int maxwg = getenv("FP_BN_FUSED_MAXWG") ? atoi(getenv("FP_BN_FUSED_MAXWG")) : 512;   // all of them must be resident at once
  int64_t rows = ((int64_t)kb * 1024) / ((int64_t)C * 4);
  if (rows * maxwg < M) rows = fp_ceil_div(M, maxwg);
  rows = fp_ceil_div(rows, R) * R;
  if (rows < R) rows = R;
  *rows_per_wg = (int)rows;
  return (int)fp_ceil_div(M, rows);
}

// ticket form (statistics + combination in one launch, apply in the next): nobody waits, so the grid only has to keep the last
// arriver's combine short -- G x C partials <= 8192 (two to four batches of loads) -- and the row loop deeply unrolled
int bn_ticket_grid(int64_t M, int C, int* rows_per_wg) {
  const int R = 256 / (C / 4);
  static const int cap = getenv("FP_BN_TICKET_PARTIALS") ? atoi(getenv("FP_BN_TICKET_PARTIALS")) : 8192;
  int g = cap / C;
  if (g < 8) g = 8;
  if (g > 256) g = 256;
  int64_t rows = fp_ceil_div(M, g);
  const int64_t min_rows = (int64_t)16 * 1024 / ((int64_t)C * 4);      // at least 16 KB per workgroup
  if (rows < min_rows) rows = min_rows;
  rows = fp_ceil_div(rows, R) * R;
  if (rows < R) rows = R;
  *rows_per_wg = (int)rows;
  return (int)fp_ceil_div(M, rows);
}

// channel-major combine helpers of the last-arriving workgroup: work item i = (channel, sub) with P = max(1, 256 / C) threads per
// channel; thread `sub` combines partials sub, sub + P, ... in order, then the P threads of a channel (adjacent lanes) merge in a
// fixed shuffle tree
template <bool APPLY>
__global__ void __launch_bounds__(256) bn_fused_fwd_kernel(const float* __restrict__ z, const float* __restrict__ res, float* __restrict__ y,
                                                           const BnGeom g, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float eps, float momentum, float* running_mean, float* running_var, long long* nbt,
                                                           float* save_mean, float* save_invstd, float* scale, float* shift, float* part,
                                                           unsigned* sync, int relu, unsigned* amax_out) {
  __shared__ float sm[3 * 256 * 4];
  const int C = g.C, M = g.M, C4 = C >> 2, R = 256 / C4, G = gridDim.x, b = blockIdx.x;
  const int cq = threadIdx.x % C4, rr = threadIdx.x / C4;
  const int m0 = b * g.rows_per_wg, m1 = min(M, m0 + g.rows_per_wg);
  // ---- phase 1: Welford over this workgroup's rows ------------------------------------------------------------------------
  Wf w[4] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  float cnt = 0.f;
  for (int m = m0 + rr; m < m1; m += BN_UNROLL * R) {       // BN_UNROLL independent loads in flight per thread
    float4 v[BN_UNROLL];
#pragma unroll
    for (int u = 0; u < BN_UNROLL; ++u) {
      const int mm = min(m + u * R, m1 - 1);
      v[u] = *reinterpret_cast<const float4*>(z + (size_t)mm * C + cq * 4);
    }
#pragma unroll
    for (int u = 0; u < BN_UNROLL; ++u) {
      if (m + u * R >= m1) break;
      cnt += 1.f;
      const float rn = 1.f / cnt;
      wf_add(w[0], v[u].x, cnt, rn); wf_add(w[1], v[u].y, cnt, rn); wf_add(w[2], v[u].z, cnt, rn); wf_add(w[3], v[u].w, cnt, rn);
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    sm[(0 * 256 + threadIdx.x) * 4 + j] = w[j].n;
    sm[(1 * 256 + threadIdx.x) * 4 + j] = w[j].mean;
    sm[(2 * 256 + threadIdx.x) * 4 + j] = w[j].m2;
  }
  __syncthreads();
  if (rr == 0) {
    for (int r = 1; r < R; ++r) {
      const int tt = r * C4 + cq;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        Wf o{sm[(0 * 256 + tt) * 4 + j], sm[(1 * 256 + tt) * 4 + j], sm[(2 * 256 + tt) * 4 + j]};
        wf_merge(w[j], o);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float* p = part + ((size_t)b * C + cq * 4 + j) * 3;
      fp_gs_store(p + 0, w[j].n); fp_gs_store(p + 1, w[j].mean); fp_gs_store(p + 2, w[j].m2);
    }
  }
  fp_gs_publish();
  if (fp_gs_arrive_last(sync, b, G)) {
    // ---- combine: every channel's G partials in a fixed order -> mean / invstd / scale / shift / running statistics ----------
    const int P = C >= 256 ? 1 : 256 / C;
    for (int i = threadIdx.x; i < C * P; i += 256) {
      const int c = i / P, sub = i % P;
      Wf a{0, 0, 0};
      for (int k0 = sub; k0 < G; k0 += 8 * P) {         // eight partial triples in flight per thread, merged in index order
        Wf o[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int k = min(k0 + u * P, G - 1);
          const float* p = part + ((size_t)k * C + c) * 3;
          o[u] = Wf{fp_gs_load(p), fp_gs_load(p + 1), fp_gs_load(p + 2)};
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (k0 + u * P < G) wf_merge(a, o[u]);
      }
      for (int o = 1; o < P; o <<= 1) {               // P in {1, 2, 4}: lanes sub .. sub + P - 1 are adjacent
        Wf t{__shfl_down(a.n, o, 64), __shfl_down(a.mean, o, 64), __shfl_down(a.m2, o, 64)};
        if ((sub & (2 * o - 1)) == 0) wf_merge(a, t);
      }
      if (sub != 0) continue;
      const float var = a.m2 / a.n;                   // biased: used for normalisation
      const float invstd = 1.f / sqrtf(var + eps);
      const float sc = gamma[c] * invstd;
      save_mean[c] = a.mean;
      save_invstd[c] = invstd;
      fp_gs_store(scale + c, sc);
      fp_gs_store(shift + c, beta[c] - a.mean * sc);
      if (running_mean) {
        const float unbiased = a.n > 1.f ? a.m2 / (a.n - 1.f) : var;   // torch: running_var uses the unbiased estimate
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * a.mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
      }
    }
    if (threadIdx.x == 0 && nbt) *nbt += 1;
    if (APPLY) {
      fp_gs_publish();
      fp_gs_release_all(sync);
    }
  }
  if (!APPLY) return;                                // ticket form: statistics + their combination only (the apply launch follows)
  // ---- phase 2: y = act(z * scale + shift (+ residual)) over the same rows ---------------------------------------------------
  fp_gs_wait(sync);
  float4 sc, sh;
  sc.x = fp_gs_load(scale + cq * 4 + 0); sc.y = fp_gs_load(scale + cq * 4 + 1); sc.z = fp_gs_load(scale + cq * 4 + 2); sc.w = fp_gs_load(scale + cq * 4 + 3);
  sh.x = fp_gs_load(shift + cq * 4 + 0); sh.y = fp_gs_load(shift + cq * 4 + 1); sh.z = fp_gs_load(shift + cq * 4 + 2); sh.w = fp_gs_load(shift + cq * 4 + 3);
  fp_gs_leave(sync, b, G);
  float ymax = 0.f;
  for (int m = m0 + rr; m < m1; m += BN_UNROLL * R) {
    float4 v[BN_UNROLL], rq[BN_UNROLL];
#pragma unroll
    for (int u = 0; u < BN_UNROLL; ++u) {
      const size_t e = (size_t)min(m + u * R, m1 - 1) * C + cq * 4;
      v[u] = *reinterpret_cast<const float4*>(z + e);
      rq[u] = res ? *reinterpret_cast<const float4*>(res + e) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < BN_UNROLL; ++u) {
      if (m + u * R >= m1) break;
      const size_t e = (size_t)(m + u * R) * C + cq * 4;
      float4 o = make_float4(v[u].x * sc.x + sh.x, v[u].y * sc.y + sh.y, v[u].z * sc.z + sh.z, v[u].w * sc.w + sh.w);
      o.x += rq[u].x; o.y += rq[u].y; o.z += rq[u].z; o.w += rq[u].w;
      if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
      *reinterpret_cast<float4*>(y + e) = o;
      ymax = fp_amax4(ymax, o);
    }
  }
  if (amax_out) fp_amax_publish_block(amax_out, ymax);
}

// backward: (sum g, sum g * xhat) per workgroup -> combine -> coefficients, dgamma, dbeta -> dz over the same rows
template <bool APPLY>
__global__ void __launch_bounds__(256) bn_fused_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ ro, const float* __restrict__ z,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma, const BnGeom g, float* __restrict__ dz,
                                                           float* __restrict__ gout, float* dgamma, float* dbeta, int accumulate, float* part,
                                                           float* coef, unsigned* sync, unsigned* amax_out) {
  __shared__ float sm[2 * 256 * 4];
  const int C = g.C, M = g.M, C4 = C >> 2, R = 256 / C4, G = gridDim.x, b = blockIdx.x;
  const int cq = threadIdx.x % C4, rr = threadIdx.x / C4;
  const int m0 = b * g.rows_per_wg, m1 = min(M, m0 + g.rows_per_wg);
  const float4 mu = reinterpret_cast<const float4*>(mean)[cq];
  const float4 is = reinterpret_cast<const float4*>(invstd)[cq];
  float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  for (int m = m0 + rr; m < m1; m += BN_UNROLL * R) {
    float4 gv[BN_UNROLL], rv[BN_UNROLL], zv[BN_UNROLL];
#pragma unroll
    for (int u = 0; u < BN_UNROLL; ++u) {
      const size_t o = (size_t)min(m + u * R, m1 - 1) * C + cq * 4;
      gv[u] = *reinterpret_cast<const float4*>(dy + o);
      rv[u] = ro ? *reinterpret_cast<const float4*>(ro + o) : make_float4(1.f, 1.f, 1.f, 1.f);
      zv[u] = *reinterpret_cast<const float4*>(z + o);
    }
#pragma unroll
    for (int u = 0; u < BN_UNROLL; ++u) {
      if (m + u * R >= m1) break;
      float4 gq = gv[u];
      const float4 r = rv[u], v = zv[u];
      gq.x = r.x > 0.f ? gq.x : 0.f; gq.y = r.y > 0.f ? gq.y : 0.f; gq.z = r.z > 0.f ? gq.z : 0.f; gq.w = r.w > 0.f ? gq.w : 0.f;
      s1[0] += gq.x; s2[0] += gq.x * ((v.x - mu.x) * is.x);
      s1[1] += gq.y; s2[1] += gq.y * ((v.y - mu.y) * is.y);
      s1[2] += gq.z; s2[2] += gq.z * ((v.z - mu.z) * is.z);
      s1[3] += gq.w; s2[3] += gq.w * ((v.w - mu.w) * is.w);
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    sm[(0 * 256 + threadIdx.x) * 4 + j] = s1[j];
    sm[(1 * 256 + threadIdx.x) * 4 + j] = s2[j];
  }
  __syncthreads();
  if (rr == 0) {
    for (int r = 1; r < R; ++r) {
      const int tt = r * C4 + cq;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s1[j] += sm[(0 * 256 + tt) * 4 + j];
        s2[j] += sm[(1 * 256 + tt) * 4 + j];
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float* p = part + ((size_t)b * C + cq * 4 + j) * 2;
      fp_gs_store(p, s1[j]); fp_gs_store(p + 1, s2[j]);
    }
  }
  fp_gs_publish();
  if (fp_gs_arrive_last(sync, b, G)) {
    const int P = C >= 256 ? 1 : 256 / C;
    const float invM = 1.f / (float)M;
    for (int i = threadIdx.x; i < C * P; i += 256) {
      const int c = i / P, sub = i % P;
      float a1 = 0.f, a2 = 0.f;
      for (int k0 = sub; k0 < G; k0 += 8 * P) {
        float o1[8], o2[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int k = min(k0 + u * P, G - 1);
          const float* p = part + ((size_t)k * C + c) * 2;
          o1[u] = fp_gs_load(p); o2[u] = fp_gs_load(p + 1);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (k0 + u * P < G) { a1 += o1[u]; a2 += o2[u]; }
      }
      for (int o = 1; o < P; o <<= 1) {
        const float t1 = __shfl_down(a1, o, 64), t2 = __shfl_down(a2, o, 64);
        if ((sub & (2 * o - 1)) == 0) { a1 += t1; a2 += t2; }
      }
      if (sub != 0) continue;
      fp_gs_store(coef + c * 2 + 0, a1 * invM);
      fp_gs_store(coef + c * 2 + 1, a2 * invM);
      if (dgamma) dgamma[c] = accumulate ? dgamma[c] + a2 : a2;
      if (dbeta) dbeta[c] = accumulate ? dbeta[c] + a1 : a1;
    }
    if (APPLY) {
      fp_gs_publish();
      fp_gs_release_all(sync);
    }
  }
  if (!APPLY) return;
  fp_gs_wait(sync);
  float c1[4], c2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    c1[j] = fp_gs_load(coef + (cq * 4 + j) * 2 + 0);
    c2[j] = fp_gs_load(coef + (cq * 4 + j) * 2 + 1);
  }
  fp_gs_leave(sync, b, G);
  const float4 ga = reinterpret_cast<const float4*>(gamma)[cq];
  float ymax = 0.f;
  for (int m = m0 + rr; m < m1; m += BN_UNROLL * R) {
    float4 gv[BN_UNROLL], rv[BN_UNROLL], zv[BN_UNROLL];
#pragma unroll
    for (int u = 0; u < BN_UNROLL; ++u) {
      const size_t e = (size_t)min(m + u * R, m1 - 1) * C + cq * 4;
      gv[u] = *reinterpret_cast<const float4*>(dy + e);
      rv[u] = ro ? *reinterpret_cast<const float4*>(ro + e) : make_float4(1.f, 1.f, 1.f, 1.f);
      zv[u] = *reinterpret_cast<const float4*>(z + e);
    }
#pragma unroll
    for (int u = 0; u < BN_UNROLL; ++u) {
      if (m + u * R >= m1) break;
      const size_t e = (size_t)(m + u * R) * C + cq * 4;
      float4 gq = gv[u];
      const float4 r = rv[u], v = zv[u];
      gq.x = r.x > 0.f ? gq.x : 0.f; gq.y = r.y > 0.f ? gq.y : 0.f; gq.z = r.z > 0.f ? gq.z : 0.f; gq.w = r.w > 0.f ? gq.w : 0.f;
      if (gout) *reinterpret_cast<float4*>(gout + e) = gq;
      float4 o;
      o.x = ga.x * is.x * (gq.x - c1[0] - (v.x - mu.x) * is.x * c2[0]);
      o.y = ga.y * is.y * (gq.y - c1[1] - (v.y - mu.y) * is.y * c2[1]);
      o.z = ga.z * is.z * (gq.z - c1[2] - (v.z - mu.z) * is.z * c2[2]);
      o.w = ga.w * is.w * (gq.w - c1[3] - (v.w - mu.w) * is.w * c2[3]);
      *reinterpret_cast<float4*>(dz + e) = o;
      ymax = fp_amax4(ymax, o);
    }
  }
  if (amax_out) fp_amax_publish_block(amax_out, ymax);
}


// ======== bn_pool.hip entry points ========
extern "C" int32_t fp_grid_sync_words(void) { return FP_GSYNC_WORDS; }

// fused forms: one launch; `sync` = fp_grid_sync_words() uint32 owned by ONE stream, zeroed once by the caller (self re-arming)
extern "C" int fp_bn_train_fused(const float* z, const float* residual, float* y, int64_t M, int32_t C, const float* gamma, const float* beta,
                                 float eps, float momentum, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                 float* save_mean, float* save_invstd, float* scale, float* shift, int32_t relu, void* workspace,
                                 int64_t workspace_bytes, uint32_t* sync, fp_stream_t stream) {
  unsigned* amax_out = fp_take_amax_out();     // consumed first: an argument error below must not leave the sink armed
  FP_REQUIRE(z && y && gamma && beta && save_mean && save_invstd && scale && shift && workspace && sync, "fp_bn_train_fused: null pointer");
  FP_REQUIRE(bn_c_ok(C) && M > 0 && M < ((int64_t)1 << 31), "fp_bn_train_fused: unsupported C=%d", C);
  FP_REQUIRE(workspace_bytes >= fp_bn_workspace(M, C), "fp_bn_train_fused: workspace too small");
  BnGeom g;
  g.M = (int)M; g.C = C;
  const int grid = bn_fused_grid(M, C, &g.rows_per_wg);
  fp_launch(bn_fused_fwd_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, z, residual, y, g, gamma, beta, eps, momentum, running_mean,
            running_var, (long long*)num_batches_tracked, save_mean, save_invstd, scale, shift, (float*)workspace, (unsigned*)sync, (int)relu,
            amax_out);
  return fp_check_launch("fp_bn_train_fused");
}

extern "C" int fp_bn_bwd_fused(const float* dy, const float* relu_out, const float* z, const float* save_mean, const float* save_invstd,
                               const float* gamma, float* dz, float* g_out, float* dgamma, float* dbeta, int accumulate, int64_t M, int32_t C,
                               void* workspace, int64_t workspace_bytes, uint32_t* sync, fp_stream_t stream) {
  unsigned* amax_out = fp_take_amax_out();
  FP_REQUIRE(dy && z && save_mean && save_invstd && gamma && dz && workspace && sync, "fp_bn_bwd_fused: null pointer");
  FP_REQUIRE(bn_c_ok(C) && M > 0 && M < ((int64_t)1 << 31), "fp_bn_bwd_fused: unsupported C=%d", C);
  FP_REQUIRE(workspace_bytes >= fp_bn_workspace(M, C), "fp_bn_bwd_fused: workspace too small");
  BnGeom g;
  g.M = (int)M; g.C = C;
  const int grid = bn_fused_grid(M, C, &g.rows_per_wg);
  float* part = (float*)workspace;
  float* coef = (float*)((char*)workspace + fp_bn_workspace(M, C)) - (size_t)C * 2;      // the last 2 C floats of the workspace
  fp_launch(bn_fused_bwd_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, dy, relu_out, z, save_mean, save_invstd, gamma, g, dz, g_out,
            dgamma, dbeta, accumulate, part, coef, (unsigned*)sync, amax_out);
  return fp_check_launch("fp_bn_bwd_fused");
}

// ticket forms: the statistics / reduction kernel and its per-channel combination in ONE launch (the last workgroup to arrive combines;
// nobody waits), the element-wise apply launch as before: two launches per layer and direction instead of three
extern "C" int fp_bn_train_stats_ticket(const float* z, int64_t M, int32_t C, const float* gamma, const float* beta, float eps, float momentum,
                                        float* running_mean, float* running_var, int64_t* num_batches_tracked, float* save_mean,
                                        float* save_invstd, float* scale, float* shift, void* workspace, int64_t workspace_bytes, uint32_t* sync,
                                        fp_stream_t stream) {
  FP_REQUIRE(z && gamma && beta && save_mean && save_invstd && scale && shift && workspace && sync, "fp_bn_train_stats_ticket: null pointer");
  FP_REQUIRE(bn_c_ok(C) && M > 0 && M < ((int64_t)1 << 31), "fp_bn_train_stats_ticket: unsupported C=%d", C);
  FP_REQUIRE(workspace_bytes >= fp_bn_workspace(M, C), "fp_bn_train_stats_ticket: workspace too small");
  BnGeom g;
  g.M = (int)M; g.C = C;
  const int grid = bn_ticket_grid(M, C, &g.rows_per_wg);
  fp_launch(bn_fused_fwd_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, z, (const float*)nullptr, (float*)nullptr, g, gamma, beta, eps,
            momentum, running_mean, running_var, (long long*)num_batches_tracked, save_mean, save_invstd, scale, shift, (float*)workspace,
            (unsigned*)sync, 0, (unsigned*)nullptr);
  return fp_check_launch("fp_bn_train_stats_ticket");
}

extern "C" int fp_bn_bwd_ticket(const float* dy, const float* relu_out, const float* z, const float* save_mean, const float* save_invstd,
                                const float* gamma, float* dz, float* g_out, float* dgamma, float* dbeta, int accumulate, int64_t M, int32_t C,
                                void* workspace, int64_t workspace_bytes, uint32_t* sync, fp_stream_t stream) {
  unsigned* amax_out = fp_take_amax_out();
  FP_REQUIRE(dy && z && save_mean && save_invstd && gamma && dz && workspace && sync, "fp_bn_bwd_ticket: null pointer");
  FP_REQUIRE(bn_c_ok(C) && M > 0 && M < ((int64_t)1 << 31), "fp_bn_bwd_ticket: unsupported C=%d", C);
  FP_REQUIRE(workspace_bytes >= fp_bn_workspace(M, C), "fp_bn_bwd_ticket: workspace too small");
  BnGeom g;
  g.M = (int)M; g.C = C;
  const int grid = bn_ticket_grid(M, C, &g.rows_per_wg);
  float* part = (float*)workspace;
  float* coef = (float*)((char*)workspace + fp_bn_workspace(M, C)) - (size_t)C * 2;
  fp_launch(bn_fused_bwd_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, dy, relu_out, z, save_mean, save_invstd, gamma, g,
            (float*)nullptr, (float*)nullptr, dgamma, dbeta, accumulate, part, coef, (unsigned*)sync, (unsigned*)nullptr);
  const size_t total4 = (size_t)M * (C / 4);
  fp_launch(bn_bwd_apply_kernel, dim3(ew_grid(total4, 8192)), dim3(256), 0, (hipStream_t)stream, dy, relu_out, z, save_mean, save_invstd, gamma,
            (const float*)coef, dz, g_out, total4, C / 4, amax_out);
  return fp_check_launch("fp_bn_bwd_ticket");
}

