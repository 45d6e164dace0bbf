"""Effective host core count (cgroup quota and affinity aware).  Test infrastructure (see oracle/__init__.py).

`os.cpu_count()` reports the machine's CPUs even when the container is limited to a few by a cgroup quota;
running the CPU oracle with hundreds of OpenMP threads on a handful of cores is pathologically slow, so the
tests and bench.py's cpu_baseline leg size torch's intra-op pool with this instead."""
import os


def effective_cores():
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != "max":
                n = min(n, max(1, int(int(quota) / int(period))))
        except (OSError, ValueError):
            pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and p > 0:
            n = min(n, max(1, q // p))
    except (OSError, ValueError):
        pass
    return max(1, n)
