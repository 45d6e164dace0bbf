"""GPU box: stream -> software queue -> hardware queue map of a training step, from the runtime's own log (AMD_LOG_LEVEL=4).
    python scripts/queue_map_probe.py [experiment of dp_tax_probe2.py ...]
Runs two eager steps per experiment in a child with the log redirected to /tmp, then pairs every `hipLaunchKernel(... stream:X)` API
line with the dispatch line that follows it (SWq / HWq) and prints the map plus every queue creation / acquisition line."""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    for exp in (sys.argv[1:] or ["single", "comm_after_engine", "comm_after_touch"]):
        env = dict(os.environ, AMD_LOG_LEVEL="4", FP_PLAN="0", PROBE_STEPS="2", PROBE_PRINT_STREAMS="1")
        log = "/tmp/qmap_%s.log" % exp
        with open(log, "w") as fh:
            p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "dp_tax_probe2.py"), exp], env=env, stdout=subprocess.PIPE, stderr=fh, text=True)
        print("=== %s: %s" % (exp, " | ".join(l for l in p.stdout.splitlines() if l.startswith(("RESULT", "STREAMS")))))
        pair = collections.Counter()
        cur = None
        created = []
        for line in open(log, errors="ignore"):
            if "hipLaunchKernel" in line or "hipExtModuleLaunchKernel" in line or "hipModuleLaunchKernel" in line:
                m = re.search(r"stream:<?(0x[0-9a-f]+|null|<null>)", line)
                cur = m.group(1) if m else "?"
            elif "Dispatch Header" in line and cur is not None:
                m = re.search(r"SWq=(0x[0-9a-f]+), HWq=(0x[0-9a-f]+), id=(\d+)", line)
                if m:
                    pair[(cur, m.group(3), m.group(2))] += 1
                cur = None
            elif "Created SWq" in line or "acquireQueue" in line or "releaseQueue" in line or "allocated hardware queues" in line or "Selected queue" in line:
                created.append(re.sub(r"^.*?\] ", "", line.strip()))
        for k, v in sorted(pair.items(), key=lambda kv: (kv[0][1], kv[0][0])):
            print("  stream %-16s -> SWq id %s  HWq %s : %d dispatches" % (k[0], k[1], k[2], v))
        print("  queue events (%d):" % len(created))
        for c in created[:60]:
            print("   ", c)
        os.remove(log)


if __name__ == "__main__":
    main()
