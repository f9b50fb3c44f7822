"""Host environment facts."""
import os


def usable_cpus(cap=None):
    """CPU threads this process may really use: affinity mask and cgroup quota, not the host's core
    count (a 1-GPU box shows 256 CPUs but grants a 16-CPU quota; 256 OpenMP threads there crawl)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    cap = int(os.environ.get("KPGNN_MAX_THREADS", "16")) if cap is None else cap
    return max(1, min(n, cap))
