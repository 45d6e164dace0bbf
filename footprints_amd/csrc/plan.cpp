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
#include <string.h>

#include <vector>

#include "footprints_hip.h"

int fp_set_error(int code, const char* fmt, ...);

namespace {

enum NodeKind { NODE_KERNEL = 0, NODE_RECORD = 1, NODE_WAIT = 2 };

struct Node {
  int kind;
  const void* func;
  dim3 grid, block;
  unsigned shmem;
  hipStream_t stream;
  uint32_t arg_first, nargs;   // into Plan::arg_off
  int event;                   // plan-local event index (record / wait)
};

struct Plan {
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
int g_ring_next = 0;

}  // namespace

bool fp_plan_recording() { return g_rec != nullptr; }

// called by fp_launch for every kernel launch of the library while a plan records
void fp_plan_push_kernel(const void* func, dim3 grid, dim3 block, unsigned shmem, hipStream_t stream, void** args, const size_t* sizes, int nargs) {
  Plan* p = g_rec;
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

extern "C" void* fp_plan_begin(void) {
  if (g_rec) return nullptr;
  g_rec = new Plan();
  return g_rec;
}

extern "C" int32_t fp_plan_mark(void* plan) { return plan ? (int32_t)((Plan*)plan)->nodes.size() : -1; }

extern "C" int32_t fp_plan_end(void* plan) {
  if (!plan || g_rec != (Plan*)plan) return fp_set_error(FP_EINVAL, "fp_plan_end: this plan is not recording on this thread");
  g_rec = nullptr;
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
  void* argv[64];
  for (int32_t i = begin; i < end; ++i) {
    const Node& n = p->nodes[i];
    hipError_t e;
    if (n.kind == NODE_KERNEL) {
      for (uint32_t k = 0; k < n.nargs; ++k) argv[k] = p->bytes.data() + p->arg_off[n.arg_first + k];
      e = hipLaunchKernel(n.func, n.grid, n.block, argv, n.shmem, n.stream);
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
    const int slot = g_ring_next;
    g_ring_next = (g_ring_next + 1) % RING;
    if (!g_ring_made[slot]) {
      hipError_t e = hipEventCreateWithFlags(&g_ring[slot], hipEventDisableTiming);
      if (e != hipSuccess) return fp_set_error(-(int)e - 1000, "fp_event_record: %s", hipGetErrorString(e));
      g_ring_made[slot] = true;
    }
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
