"""GPU box: second round of experiments on what the RCCL communicator's mere existence costs the training step (profiles/round3_notes.md).
    python scripts/dp_tax_probe2.py                     every experiment in a fresh process, one RESULT line each
    python scripts/dp_tax_probe2.py --only a,b,c        the named ones
    python scripts/dp_tax_probe2.py <name>              one experiment in this process
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# name -> (environment, keyword switches)
EXPERIMENTS = {
    "single": ({}, {}),
    "comm_after_engine": ({}, {"comm": "after"}),                 # = round 3a's rccl_nocoll: communicator created, never used
    "comm_before_engine": ({}, {"comm": "before"}),
    "comm_after_touch": ({}, {"comm": "after", "touch": True}),   # every engine stream runs one kernel BEFORE the communicator exists
    "comm_then_destroy": ({}, {"comm": "after", "destroy": True}),
    "comm_reset_stack": ({}, {"comm": "after", "reset_stack": True}),
    "single_dynq0": ({"DEBUG_HIP_DYNAMIC_QUEUES": "0"}, {}),
    "comm_dynq0": ({"DEBUG_HIP_DYNAMIC_QUEUES": "0"}, {"comm": "after"}),
    "single_dynq1": ({"DEBUG_HIP_DYNAMIC_QUEUES": "1"}, {}),
    "comm_dynq1": ({"DEBUG_HIP_DYNAMIC_QUEUES": "1"}, {"comm": "after"}),
    "single_hwq6": ({"GPU_MAX_HW_QUEUES": "6"}, {}),
    "comm_hwq6": ({"GPU_MAX_HW_QUEUES": "6"}, {"comm": "after"}),
    "single_hwq5": ({"GPU_MAX_HW_QUEUES": "5"}, {}),
    "comm_hwq5": ({"GPU_MAX_HW_QUEUES": "5"}, {"comm": "after"}),
    "single_serial": ({"FP_SERIAL": "1"}, {}),
    "comm_serial": ({"FP_SERIAL": "1"}, {"comm": "after"}),
}
HIP_LIMIT_STACK = 0      # hipLimitStackSize


def stack_limit(hip):
    v = ctypes.c_size_t(0)
    rc = hip.hipDeviceGetLimit(ctypes.byref(v), HIP_LIMIT_STACK)
    return v.value if rc == 0 else -rc


def run(name):
    import torch
    env, kw = EXPERIMENTS[name]
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")
    hip = ctypes.CDLL("libamdhip64.so")
    from footprints_amd import ops, parallel
    from footprints_amd.model_manager import ModelManager
    from footprints_amd.training.train import SEED, TrainStep, synthetic_batch
    note = ["stack0=%d" % stack_limit(hip)]
    comm = None
    if kw.get("comm") == "before":
        comm = parallel.Communicator()
    torch.manual_seed(SEED)
    mm = ModelManager(use_cuda=True)
    ts = TrainStep(mm.model, mm.optimiser)
    eng = ts.eng
    if os.environ.get("PROBE_PRINT_STREAMS"):
        print("STREAMS main=%#x aux=%#x wg=%#x dwg0=%#x dwg1=%#x" % (torch.cuda.current_stream().cuda_stream, eng.aux.cuda_stream, eng.wg.cuda_stream,
                                                                  eng.dwg[0].cuda_stream, eng.dwg[1].cuda_stream), flush=True)
    if kw.get("touch"):
        tiny = torch.zeros(64, device="cuda")
        for s in (eng.wg, eng.aux, eng.dwg[0], eng.dwg[1]):
            with ops.on_stream(s):
                ops.fill(tiny, 1.0)
        torch.cuda.synchronize()
    if kw.get("comm") == "after":
        comm = parallel.Communicator()
    note.append("stack1=%d" % stack_limit(hip))
    if kw.get("reset_stack"):
        hip.hipDeviceSetLimit(HIP_LIMIT_STACK, ctypes.c_size_t(int(note[0].split("=")[1])))
        note.append("stack2=%d" % stack_limit(hip))
    if kw.get("destroy") and comm is not None:
        comm.destroy()
    batch = synthetic_batch(12, 192, 640, "cuda")
    nsteps = int(os.environ.get("PROBE_STEPS", "30"))
    for _ in range(min(8, nsteps)):
        ts(batch)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(nsteps):
        ts(batch)
    e1.record()
    torch.cuda.synchronize()
    print("RESULT %-20s %.3f ms/step  %s" % (name, e0.elapsed_time(e1) / nsteps, " ".join(note)), flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] != "--only":
        return run(sys.argv[1])
    names = sys.argv[2].split(",") if len(sys.argv) > 2 else list(EXPERIMENTS)
    for n in names:
        e = dict(os.environ)
        e.update(EXPERIMENTS[n][0])
        p = subprocess.run([sys.executable, os.path.abspath(__file__), n], env=e, capture_output=True, text=True, timeout=300)
        line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
        print(line[0] if line else "RESULT %s FAILED rc=%d %s" % (n, p.returncode, (p.stderr or "")[-300:].replace("\n", " | ")), flush=True)


if __name__ == "__main__":
    main()
