"""How many host cores this process may really use: the smaller of the affinity mask and the cgroup CPU quota
(a GPU box exposes all 256 logical CPUs but grants a quota of 16 per GPU; oversubscribing it only adds
throttling and cache thrash)."""
import math
import os


def effective_cores():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != "max":
                n = min(n, max(1, math.ceil(int(quota) / int(period))))
        except (OSError, ValueError):
            pass
    try:                                                    # cgroup v1
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and p > 0:
            n = min(n, max(1, math.ceil(q / p)))
    except (OSError, ValueError):
        pass
    return n
