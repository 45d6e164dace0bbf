// Data-parallel gradient exchange over RCCL / xGMI behind the C ABI (include/footprints_hip.h, fp_comm_*).
//
// One process per GPU; the live gradients are ONE flat fp32 buffer, cut into contiguous buckets in the order the backward pass
// completes them; every bucket is summed across ranks in place by one ncclAllReduce on a stream the CALLER names (the engine
// orders that stream behind the gradient-writing streams with fp_event_record / fp_event_wait).  While a launch plan records
// (plan.cpp) a collective is executed AND appended to the plan, so a whole data-parallel training step -- kernels, ordering edges
// and bucket all-reduces -- replays from C in one call.
//
// RCCL is resolved with dlopen / dlsym at fp_comm_unique_id / fp_comm_init time (the copy the process already holds -- PyTorch
// ships its own librccl.so -- else librccl.so.1 from the loader path, else /opt/rocm/lib): libfootprints_hip.so itself has no
// link-time dependency on it and single-GPU users never load it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "footprints_hip.h"

// The handful of NCCL / RCCL ABI items this file uses, declared here instead of through <rccl/rccl.h>: the library is resolved at run time,
// so a box without the RCCL development headers can still BUILD libfootprints_hip.so.  Values are the public NCCL ABI.
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
typedef int ncclDataType_t;
typedef int ncclRedOp_t;
}
static constexpr ncclResult_t ncclSuccess = 0;
static constexpr ncclDataType_t ncclFloat = 7;      // ncclFloat32
static constexpr ncclRedOp_t ncclSum = 0;

int fp_set_error(int code, const char* fmt, ...);
bool fp_plan_recording();
void fp_plan_push_allreduce(void* comm, float* buf, int64_t count, hipStream_t stream);
void fp_plan_mark_failed();
bool fp_ktime_active();                                             // plan.cpp: bench.py's per-kernel event timing is collecting
hipEvent_t fp_ktime_open(hipStream_t stream);
void fp_ktime_close(const char* name, hipStream_t stream, hipEvent_t opened);

namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
};
Rccl g_rccl;

struct Comm {
  ncclComm_t comm;
  int rank, world, device;
};

int load_rccl() {
  if (g_rccl.handle) return FP_OK;
  void* h = nullptr;
  const char* resident[] = {"librccl.so", "librccl.so.1"};
  for (const char* n : resident)
    if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);          // the copy another library of this process (PyTorch) already loaded
  const char* fresh[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : fresh)
    if (!h) h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
  if (!h) return fp_set_error(FP_EINVAL, "fp_comm: librccl.so not found (%s)", dlerror());
  Rccl r;
  r.handle = h;
  *(void**)&r.GetUniqueId = dlsym(h, "ncclGetUniqueId");
  *(void**)&r.CommInitRank = dlsym(h, "ncclCommInitRank");
  *(void**)&r.CommDestroy = dlsym(h, "ncclCommDestroy");
  *(void**)&r.AllReduce = dlsym(h, "ncclAllReduce");
  *(void**)&r.Broadcast = dlsym(h, "ncclBroadcast");
  *(void**)&r.GetErrorString = dlsym(h, "ncclGetErrorString");
  *(void**)&r.GetVersion = dlsym(h, "ncclGetVersion");
  *(void**)&r.CommCount = dlsym(h, "ncclCommCount");
  if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce || !r.Broadcast || !r.GetErrorString)
    return fp_set_error(FP_EINVAL, "fp_comm: librccl.so lacks an expected symbol");
  g_rccl = r;
  return FP_OK;
}

int nccl_fail(const char* what, ncclResult_t e) {
  return fp_set_error(-100 - (int)e, "%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "rccl error");
}

}  // namespace

// used by fp_plan_replay (plan.cpp): the collective of a recorded node
int fp_comm_allreduce_raw(void* comm_, float* buf, int64_t count, hipStream_t stream) {
  Comm* c = (Comm*)comm_;
  const bool timed = fp_ktime_active();                // bench.py: the collective's duration on its stream, one row per bucket size
  hipEvent_t opened = timed ? fp_ktime_open(stream) : nullptr;
  const ncclResult_t e = g_rccl.AllReduce(buf, buf, (size_t)count, ncclFloat, ncclSum, c->comm, stream);
  if (timed) {
    char name[96];
    snprintf(name, sizeof(name), "rccl_allreduce(%lld floats, %d ranks)", (long long)count, c->world);
    fp_ktime_close(name, stream, opened);
  }
  return e == ncclSuccess ? FP_OK : nccl_fail("ncclAllReduce", e);
}

// ranks of the communicator as RCCL itself counts them (ncclCommCount), or the world it was created with when the symbol is missing
extern "C" int32_t fp_comm_count(void* comm) {
  Comm* c = (Comm*)comm;
  if (!c) return 0;
  int n = 0;
  if (g_rccl.CommCount && g_rccl.CommCount(c->comm, &n) == ncclSuccess) return (int32_t)n;
  return (int32_t)c->world;
}

extern "C" int32_t fp_comm_unique_id_bytes(void) { return (int32_t)sizeof(ncclUniqueId); }

extern "C" int fp_comm_unique_id(void* id_out, int32_t cap) {
  if (!id_out || cap < (int32_t)sizeof(ncclUniqueId)) return fp_set_error(FP_EINVAL, "fp_comm_unique_id: buffer of %d bytes needed", (int)sizeof(ncclUniqueId));
  if (int r = load_rccl()) return r;
  ncclUniqueId id;
  const ncclResult_t e = g_rccl.GetUniqueId(&id);
  if (e != ncclSuccess) return nccl_fail("ncclGetUniqueId", e);
  memcpy(id_out, &id, sizeof(id));
  return FP_OK;
}

extern "C" int fp_comm_init(const void* id_bytes, int32_t rank, int32_t world, void** comm_out) {
  if (!id_bytes || !comm_out || world < 1 || rank < 0 || rank >= world) return fp_set_error(FP_EINVAL, "fp_comm_init: bad rank %d / world %d", rank, world);
  if (int r = load_rccl()) return r;
  ncclUniqueId id;
  memcpy(&id, id_bytes, sizeof(id));
  Comm* c = new Comm();
  c->rank = rank;
  c->world = world;
  (void)hipGetDevice(&c->device);
  const ncclResult_t e = g_rccl.CommInitRank(&c->comm, world, id, rank);     // blocks until every rank of the world has called it
  if (e != ncclSuccess) {
    delete c;
    return nccl_fail("ncclCommInitRank", e);
  }
  *comm_out = c;
  return FP_OK;
}

extern "C" int32_t fp_comm_version(void) {
  if (load_rccl() != FP_OK || !g_rccl.GetVersion) return -1;
  int v = 0;
  return g_rccl.GetVersion(&v) == ncclSuccess ? (int32_t)v : -1;
}

extern "C" int fp_comm_allreduce_async(void* comm, float* buf, int64_t count, fp_stream_t stream_) {
  if (!comm || !buf || count <= 0) return fp_set_error(FP_EINVAL, "fp_comm_allreduce_async: null communicator / buffer or empty range");
  hipStream_t stream = (hipStream_t)stream_;
  const int r = fp_comm_allreduce_raw(comm, buf, count, stream);
  if (fp_plan_recording()) {
    if (r != FP_OK) fp_plan_mark_failed();
    else fp_plan_push_allreduce(comm, buf, count, stream);
  }
  return r;
}

extern "C" int fp_comm_broadcast(void* comm, float* buf, int64_t count, int32_t root, fp_stream_t stream) {
  Comm* c = (Comm*)comm;
  if (!c || !buf || count <= 0 || root < 0 || root >= c->world) return fp_set_error(FP_EINVAL, "fp_comm_broadcast: bad arguments");
  if (fp_plan_recording()) return fp_set_error(FP_EINVAL, "fp_comm_broadcast: not recordable into a launch plan");
  const ncclResult_t e = g_rccl.Broadcast(buf, buf, (size_t)count, ncclFloat, root, c->comm, (hipStream_t)stream);
  return e == ncclSuccess ? FP_OK : nccl_fail("ncclBroadcast", e);
}

// `consumer` waits for everything queued on `comm_stream` so far (the all-reduces issued there): one event edge, recorded into a
// plan like any other ordering edge of the library
extern "C" int fp_comm_wait(void* comm, fp_stream_t comm_stream, fp_stream_t consumer) {
  if (!comm) return fp_set_error(FP_EINVAL, "fp_comm_wait: null communicator");
  const int64_t ev = fp_event_record(comm_stream);
  if (ev < 0) return (int)ev;
  return fp_event_wait(consumer, ev);
}

extern "C" int fp_comm_destroy(void* comm) {
  Comm* c = (Comm*)comm;
  if (!c) return FP_OK;
  const ncclResult_t e = g_rccl.CommDestroy(c->comm);
  delete c;
  return e == ncclSuccess ? FP_OK : nccl_fail("ncclCommDestroy", e);
}
