"""Turn a rocprofv3 (--kernel-trace --stats, rocpd sqlite output) database into a committed text summary.

    python scripts/rocprof_summary.py gpurun_out/prof/x_results.db profiles/round1_kernel_stats.txt "command line"
"""
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
cmd = sys.argv[3] if len(sys.argv) > 3 else ""
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
total = sum(r[2] for r in rows)
with open(out, "w") as fh:
    fh.write("# rocprofv3 --kernel-trace --stats summary (durations in microseconds, from the rocpd `top_kernels` view)\n")
    fh.write("# command: %s\n# total kernel time: %.3f ms over %d kernel symbols (all profiled steps together)\n" % (cmd, total / 1e3, len(rows)))
    fh.write("%-100s %8s %14s %12s %7s\n" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, tot, avg, pct in rows:
        name = name.replace("(anonymous namespace)::", "").replace("void ", "")
        fh.write("%-100s %8d %14.0f %12.1f %6.2f%%\n" % (name[:100], calls, tot, avg, pct))
print("wrote", out)
