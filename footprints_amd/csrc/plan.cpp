// Recorded launch plans: the C side of "replay a training step without Python" (include/footprints_hip.h, fp_plan_*).
//
// A training step is ~740 kernel launches on five streams plus ~150 event record / wait pairs, every one of them with the same
// arguments as in the previous step (static activation arena, static workspaces, static packed-weight tables).  Issued from Python
// through ctypes that costs ~15 ms of host time per 18 ms step.  While a plan is recording (fp_plan_begin .. fp_plan_end, thread
// local) every kernel launch of the library -- they all go through fp_launch (fp_common.h) -- and every fp_event_record /
// fp_event_wait is executed AND appended to the plan with a private copy of its argument bytes; fp_plan_replay then re-issues the
// nodes [begin, end) with plain hipLaunchKernel / hipEventRecord / hipStreamWaitEvent calls.  No hipGraph: the streams, the
// concurrency and the results are exactly those of the recorded eager step (a graph replay measured slower on ROCm 7.2).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <cxxabi.h>

#include <atomic>
#include <map>
#include <string>
#include <vector>

#include "footprints_hip.h"

int fp_set_error(int code, const char* fmt, ...);
hipError_t fp_launch_timed(const void* func, dim3 grid, dim3 block, void** args, unsigned shmem, hipStream_t stream);
int fp_comm_allreduce_raw(void* comm, float* buf, int64_t count, hipStream_t stream);     // comm.cpp

namespace {

enum NodeKind { NODE_KERNEL = 0, NODE_RECORD = 1, NODE_WAIT = 2, NODE_ALLREDUCE = 3 };

struct Node {
  int kind;
  const void* func;
  dim3 grid, block;
  unsigned shmem;
  hipStream_t stream;
  uint32_t arg_first, nargs;   // into Plan::arg_off
  int event;                   // plan-local event index (record / wait)
  void* comm;                  // NODE_ALLREDUCE: communicator, buffer (in place), element count
  float* buf;
  int64_t count;
};

constexpr uint32_t MAX_KERNEL_ARGS = 64;     // fp_plan_replay's argv; the library's widest kernel takes 31 parameters

struct Plan {
  bool failed = false;                  // a launch failed / was rejected while recording: fp_plan_end reports it
  std::vector<Node> nodes;
  std::vector<unsigned char> bytes;     // argument storage (16-byte aligned slots)
  std::vector<uint32_t> arg_off;        // byte offset of every argument
  std::vector<hipEvent_t> events;
};

thread_local Plan* g_rec = nullptr;

// events outside a recording: a ring of reusable, timing-disabled events (a step uses ~150; the ring holds 4096)
constexpr int RING = 4096;
hipEvent_t g_ring[RING];
bool g_ring_made[RING];
std::atomic<int> g_ring_next{0};      // fp_event_record may be called from several host threads (loader thread + trainer)
std::atomic<bool> g_ring_lock{false}; // guards the lazy creation of a ring slot's event

}  // namespace

bool fp_plan_recording() { return g_rec != nullptr; }

// ---- per-kernel timing (fp_ktime_*): HIP events on the launch stream around every kernel launch of the library --------------------
// bench.py's roofline leg: the duration of every launch, measured live with events recorded on the stream the kernel is launched
// on, aggregated per kernel symbol -- the same quantity `rocprofv3 --kernel-trace --stats` reports per kernel.
namespace {
struct TimedLaunch { const void* func; hipEvent_t a, b; };
bool g_kt_on = false;
std::vector<TimedLaunch> g_kt;
std::vector<hipEvent_t> g_kt_pool;
struct KRow { std::string name; int64_t launches; double ms; };
std::vector<KRow> g_kt_rows;
hipEvent_t kt_event() {
  if (!g_kt_pool.empty()) { hipEvent_t e = g_kt_pool.back(); g_kt_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}
}  // namespace

// work that is not one of the library's kernels (the RCCL collectives of comm.cpp): the same event pair, a row named by the caller
namespace {
std::map<std::string, int> g_kt_names;                 // the key's address identifies the row (std::map nodes do not move)
}
bool fp_ktime_active() { return g_kt_on; }
hipEvent_t fp_ktime_open(hipStream_t stream) {
  hipEvent_t a = kt_event();
  (void)hipEventRecord(a, stream);
  return a;
}
void fp_ktime_close(const char* name, hipStream_t stream, hipEvent_t opened) {
  hipEvent_t b = kt_event();
  (void)hipEventRecord(b, stream);
  auto it = g_kt_names.emplace(std::string(name), 0).first;
  g_kt.push_back(TimedLaunch{(const void*)&it->first, opened, b});
}

hipError_t fp_launch_timed(const void* func, dim3 grid, dim3 block, void** args, unsigned shmem, hipStream_t stream) {
  if (!g_kt_on) return hipLaunchKernel(func, grid, block, args, shmem, stream);
  TimedLaunch t{func, kt_event(), kt_event()};
  (void)hipEventRecord(t.a, stream);
  const hipError_t e = hipLaunchKernel(func, grid, block, args, shmem, stream);
  (void)hipEventRecord(t.b, stream);
  g_kt.push_back(t);
  return e;
}

extern "C" int fp_ktime_begin(void) {
  for (auto& t : g_kt) { g_kt_pool.push_back(t.a); g_kt_pool.push_back(t.b); }
  g_kt.clear();
  g_kt_rows.clear();
  g_kt_on = true;
  return FP_OK;
}

// stops collecting, waits for the device, and aggregates per kernel symbol; returns the number of distinct kernels (rows of fp_ktime_row)
extern "C" int32_t fp_ktime_end(void) {
  g_kt_on = false;
  const hipError_t se = hipDeviceSynchronize();
  if (se != hipSuccess) return fp_set_error((int)se, "fp_ktime_end: %s", hipGetErrorString(se));
  std::map<const void*, size_t> index;
  g_kt_rows.clear();
  for (auto& t : g_kt) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, t.a, t.b) != hipSuccess) ms = 0.f;
    auto it = index.find(t.func);
    if (it == index.end()) {
      std::string name;
      bool named = false;
      for (auto& kv : g_kt_names)
        if ((const void*)&kv.first == t.func) { name = kv.first; named = true; break; }
      if (!named) {
        const char* mangled = hipKernelNameRefByPtr(t.func, nullptr);
        name = mangled ? mangled : "?";
        int st = 0;
        char* dm = mangled ? abi::__cxa_demangle(mangled, nullptr, nullptr, &st) : nullptr;
        if (dm && st == 0) name = dm;
        free(dm);
      }
      it = index.emplace(t.func, g_kt_rows.size()).first;
      g_kt_rows.push_back(KRow{name, 0, 0.0});
    }
    g_kt_rows[it->second].launches += 1;
    g_kt_rows[it->second].ms += ms;
    g_kt_pool.push_back(t.a);
    g_kt_pool.push_back(t.b);
  }
  g_kt.clear();
  return (int32_t)g_kt_rows.size();
}

extern "C" int fp_ktime_row(int32_t i, char* name, int32_t name_cap, int64_t* launches, double* total_ms) {
  if (i < 0 || i >= (int32_t)g_kt_rows.size() || !name || name_cap < 2 || !launches || !total_ms) return fp_set_error(FP_EINVAL, "fp_ktime_row: bad row / buffer");
  const KRow& r = g_kt_rows[i];
  strncpy(name, r.name.c_str(), (size_t)name_cap - 1);
  name[name_cap - 1] = 0;
  *launches = r.launches;
  *total_ms = r.ms;
  return FP_OK;
}

// called by fp_launch for every SUCCESSFUL kernel launch of the library while a plan records (a failed launch marks the plan instead)
void fp_plan_mark_failed() {
  if (g_rec) g_rec->failed = true;
}

void fp_plan_push_kernel(const void* func, dim3 grid, dim3 block, unsigned shmem, hipStream_t stream, void** args, const size_t* sizes, int nargs) {
  Plan* p = g_rec;
  if (nargs < 0 || (uint32_t)nargs > MAX_KERNEL_ARGS) {
    p->failed = true;
    fp_set_error(FP_EINVAL, "fp_plan: a kernel with %d parameters cannot be recorded (limit %u)", nargs, MAX_KERNEL_ARGS);
    return;
  }
  Node n;
  n.kind = NODE_KERNEL; n.func = func; n.grid = grid; n.block = block; n.shmem = shmem; n.stream = stream;
  n.arg_first = (uint32_t)p->arg_off.size(); n.nargs = (uint32_t)nargs; n.event = -1;
  for (int i = 0; i < nargs; ++i) {
    const size_t off = (p->bytes.size() + 15) & ~(size_t)15;
    p->bytes.resize(off + sizes[i]);
    memcpy(p->bytes.data() + off, args[i], sizes[i]);
    p->arg_off.push_back((uint32_t)off);
  }
  p->nodes.push_back(n);
}

// called by fp_comm_allreduce_async (comm.cpp) after it issued the collective while a plan records
void fp_plan_push_allreduce(void* comm, float* buf, int64_t count, hipStream_t stream) {
  Node n;
  n.kind = NODE_ALLREDUCE; n.func = nullptr; n.shmem = 0; n.stream = stream; n.arg_first = n.nargs = 0; n.event = -1;
  n.comm = comm; n.buf = buf; n.count = count;
  g_rec->nodes.push_back(n);
}

extern "C" void* fp_plan_begin(void) {
  if (g_rec) return nullptr;
  g_rec = new Plan();
  return g_rec;
}

extern "C" int32_t fp_plan_mark(void* plan) { return plan ? (int32_t)((Plan*)plan)->nodes.size() : -1; }

extern "C" int32_t fp_plan_end(void* plan) {
  if (!plan || g_rec != (Plan*)plan) return fp_set_error(FP_EINVAL, "fp_plan_end: this plan is not recording on this thread");
  g_rec = nullptr;
  if (((Plan*)plan)->failed) return fp_set_error(FP_EINVAL, "fp_plan_end: a launch failed or was rejected while this plan recorded");
  return (int32_t)((Plan*)plan)->nodes.size();
}

extern "C" void fp_plan_destroy(void* plan) {
  Plan* p = (Plan*)plan;
  if (!p) return;
  if (g_rec == p) g_rec = nullptr;
  for (hipEvent_t e : p->events) (void)hipEventDestroy(e);
  delete p;
}

extern "C" int fp_plan_replay(void* plan, int32_t begin, int32_t end) {
  Plan* p = (Plan*)plan;
  if (!p || g_rec == p) return fp_set_error(FP_EINVAL, "fp_plan_replay: null plan or plan still recording");
  if (end < 0 || end > (int32_t)p->nodes.size()) end = (int32_t)p->nodes.size();
  if (begin < 0 || begin > end) return fp_set_error(FP_EINVAL, "fp_plan_replay: bad node range");
  void* argv[MAX_KERNEL_ARGS];
  for (int32_t i = begin; i < end; ++i) {
    const Node& n = p->nodes[i];
    hipError_t e;
    if (n.kind == NODE_KERNEL) {
      for (uint32_t k = 0; k < n.nargs; ++k) argv[k] = p->bytes.data() + p->arg_off[n.arg_first + k];
      e = fp_launch_timed(n.func, n.grid, n.block, argv, n.shmem, n.stream);
    } else if (n.kind == NODE_ALLREDUCE) {
      const int r = fp_comm_allreduce_raw(n.comm, n.buf, n.count, n.stream);
      if (r != FP_OK) return r;
      e = hipSuccess;
    } else if (n.kind == NODE_RECORD) {
      e = hipEventRecord(p->events[n.event], n.stream);
    } else {
      e = hipStreamWaitEvent(n.stream, p->events[n.event], 0);
    }
    if (e != hipSuccess) return fp_set_error((int)e, "fp_plan_replay: node %d: %s", i, hipGetErrorString(e));
  }
  return FP_OK;
}

// ---- events (used by the engine instead of framework events so that a recording sees them) --------------------------------------
// returns an event id >= 0 (plan-local while recording: only meaningful to fp_event_wait during the same recording), < 0 on error
extern "C" int64_t fp_event_record(fp_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  hipEvent_t ev;
  int64_t id;
  if (g_rec) {
    hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e != hipSuccess) return fp_set_error(-(int)e - 1000, "fp_event_record: %s", hipGetErrorString(e));
    g_rec->events.push_back(ev);
    id = (int64_t)g_rec->events.size() - 1;
    Node n;
    n.kind = NODE_RECORD; n.func = nullptr; n.shmem = 0; n.stream = stream; n.arg_first = n.nargs = 0; n.event = (int)id;
    g_rec->nodes.push_back(n);
    id |= (int64_t)1 << 40;                           // tag: plan-local id
  } else {
    const int slot = (int)((unsigned)g_ring_next.fetch_add(1, std::memory_order_relaxed) % RING);
    while (g_ring_lock.exchange(true, std::memory_order_acquire)) {}
    hipError_t ce = hipSuccess;
    if (!g_ring_made[slot]) {
      ce = hipEventCreateWithFlags(&g_ring[slot], hipEventDisableTiming);
      if (ce == hipSuccess) g_ring_made[slot] = true;
    }
    g_ring_lock.store(false, std::memory_order_release);
    if (ce != hipSuccess) return fp_set_error(-(int)ce - 1000, "fp_event_record: %s", hipGetErrorString(ce));
    ev = g_ring[slot];
    id = slot;
  }
  hipError_t e = hipEventRecord(ev, stream);
  if (e != hipSuccess) return fp_set_error(-(int)e - 1000, "fp_event_record: %s", hipGetErrorString(e));
  return id;
}

extern "C" int fp_event_wait(fp_stream_t stream_, int64_t id) {
  hipStream_t stream = (hipStream_t)stream_;
  hipEvent_t ev;
  if (id < 0) return fp_set_error(FP_EINVAL, "fp_event_wait: bad event id");
  if (id >> 40) {
    const int64_t k = id & (((int64_t)1 << 40) - 1);
    if (!g_rec || k >= (int64_t)g_rec->events.size()) return fp_set_error(FP_EINVAL, "fp_event_wait: plan-local event outside its recording");
    ev = g_rec->events[k];
    Node n;
    n.kind = NODE_WAIT; n.func = nullptr; n.shmem = 0; n.stream = stream; n.arg_first = n.nargs = 0; n.event = (int)k;
    g_rec->nodes.push_back(n);
  } else {
    if (id >= RING || !g_ring_made[id]) return fp_set_error(FP_EINVAL, "fp_event_wait: unknown event id");
    if (g_rec) return fp_set_error(FP_EINVAL, "fp_event_wait: an event recorded before the plan began cannot be waited for inside it");
    ev = g_ring[id];
  }
  hipError_t e = hipStreamWaitEvent(stream, ev, 0);
  if (e != hipSuccess) return fp_set_error((int)e, "fp_event_wait: %s", hipGetErrorString(e));
  return FP_OK;
}
