"""Timeline analysis of a rocprofv3 --kernel-trace rocpd database: one steady-state training step (adam to adam).

    python scripts/timeline.py gpurun_out/prof/x_results.db [step_index] [trace] [delim=<kernel name fragment>]
delim: the once-per-step kernel that bounds a step (default: the Adam launch; `delim=loss_kernel` for FP_ADAM_STAGED=1, whose update is six launches).
Prints wall time of the step, GPU-busy union, time at concurrency 0/1/2/3+, the largest idle gaps and per-kernel sums.
"""
import sqlite3
import sys
from collections import defaultdict

db = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -2
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute("select name, start, end, stream_id from kernels order by start"))
delim = next((a.split("=", 1)[1] for a in sys.argv[2:] if a.startswith("delim=")), None)
adam = [i for i, r in enumerate(rows) if ((delim in r[0]) if delim else ("adam_kernel" in r[0] or "adam_dev_kernel" in r[0]))]
if len(adam) < 3:
    sys.exit("need >= 3 launches of the step delimiter")
a0, a1 = adam[which - 1], adam[which]
step = rows[a0 + 1:a1 + 1]
t0, t1 = rows[a0][2], rows[a1][2]
wall = (t1 - t0) / 1e3
print("step: %d kernels, wall %.1f us (%s end -> %s end)" % (len(step), wall, delim or "adam", delim or "adam"))
ev = []
for n, s, e, st in step:
    ev.append((max(s, t0), 1))
    ev.append((min(e, t1), -1))
ev.sort()
conc = defaultdict(float)
level, last = 0, t0
for t, d in ev:
    conc[min(level, 3)] += (t - last) / 1e3
    level += d
    last = t
conc[min(level, 3)] += (t1 - last) / 1e3
print("concurrency: " + "  ".join("%s: %.0f us (%.1f%%)" % (("3+" if k == 3 else k), v, 100 * v / wall) for k, v in sorted(conc.items())))
# idle gaps
gaps = []
level, last = 0, t0
for t, d in ev:
    if level == 0 and t > last:
        gaps.append((t - last) / 1e3)
    level += d
    last = t
gaps.sort(reverse=True)
print("idle gaps: n=%d total %.0f us; largest: %s" % (len(gaps), sum(gaps), ", ".join("%.1f" % g for g in gaps[:12])))
print("gaps > 1us: %d (%.0f us);  gaps <= 1us: %d (%.0f us)" % (sum(g > 1 for g in gaps), sum(g for g in gaps if g > 1),
                                                                 sum(g <= 1 for g in gaps), sum(g for g in gaps if g <= 1)))
agg = defaultdict(lambda: [0, 0.0])
for n, s, e, st in step:
    n = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    agg[n][0] += 1
    agg[n][1] += (e - s) / 1e3
tot = sum(v[1] for v in agg.values())
print("sum of kernel durations %.0f us" % tot)
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print("  %-70s %5d %9.0f us %6.1f avg  %5.1f%%" % (n[:70], c, d, d / c, 100 * d / tot))
streams = defaultdict(float)
for n, s, e, st in step:
    streams[st] += (e - s) / 1e3
print("per-stream busy: " + ", ".join("%s: %.0f us" % kv for kv in sorted(streams.items())))

# which kernels run ALONE (concurrency 1): the serial part of the schedule
ev2 = []
for i, (n, s, e, st) in enumerate(step):
    ev2 += [(max(s, t0), 1, i), (min(e, t1), -1, i)]
ev2.sort()
alone = defaultdict(float)
active, last = set(), t0
for t, d, i in ev2:
    if len(active) == 1 and t > last:
        (j,) = tuple(active)
        alone[step[j][0].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]] += (t - last) / 1e3
    if d > 0:
        active.add(i)
    else:
        active.discard(i)
    last = t
print("time with exactly one kernel running, by kernel: total %.0f us" % sum(alone.values()))
for n, d in sorted(alone.items(), key=lambda kv: -kv[1])[:22]:
    print("  %-62s %8.0f us" % (n, d))

if len(sys.argv) > 3 and sys.argv[3] == "buckets":
    nb = int(wall // 1000) + 1
    print("\nper-ms buckets: [busy1 busy2+] top kernels")
    for b in range(nb):
        lo, hi = t0 + b * 1e6, t0 + (b + 1) * 1e6
        names = defaultdict(float)
        evs = []
        for n, s, e, st in step:
            ss, ee = max(s, lo), min(e, hi)
            if ee > ss:
                names[n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:34] + "@%d" % st] += (ee - ss) / 1e3
                evs += [(ss, 1), (ee, -1)]
        evs.sort()
        lvl, last, c1, c2 = 0, lo, 0.0, 0.0
        for t, d in evs:
            if lvl == 1:
                c1 += t - last
            elif lvl >= 2:
                c2 += t - last
            lvl += d
            last = t
        top = sorted(names.items(), key=lambda kv: -kv[1])[:3]
        print("%3d ms: [%3.0f%% %3.0f%%] %s" % (b, c1 / 1e4, c2 / 1e4, "  ".join("%s %.0f" % kv for kv in top)))

if len(sys.argv) > 3 and sys.argv[3] == "trace":
    # the whole step in launch order: start / end in us from the step's start, stream, kernels running when it started, name
    print("\nordered trace of the step (us from the previous Adam's end):")
    import bisect
    ends = sorted(e for _, _, e, _ in step)
    for k, (n, s, e, st) in enumerate(step):
        running = sum(1 for (n2, s2, e2, st2) in step[max(0, k - 12):k] if e2 > s)
        nm = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        nm = nm.replace("conv3x3_tile_bf3_kernel", "tile").replace("wgrad3x3_hp_pf_kernel", "wgrad_pf")
        print("%9.1f %9.1f %7.1f s%d +%d %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, st, running, nm[:90]))
