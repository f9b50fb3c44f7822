"""Per-step kernel summary from a rocprofv3 --kernel-trace --stats output directory:
    python scripts/kstats.py <dir> [top_n]
With the kernel trace present (preferred): the launches between the last optimiser launches (adam_kernel, or the
framework's fused Adam) - i.e. STEADY-STATE steps, graph replays only - averaged over up to the last 10 steps.
Without it: kernel_stats.csv totals over the whole run (warm-up and capture passes included) divided by the step count."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(n):
    n = n.replace("(anonymous namespace)::", "")
    m = re.search(r"(\w+_kernel)(<[^(]*>)?", n)
    if m:
        return m.group(1) + (m.group(2) or "")
    return n.split("(")[0][:70]


def is_opt(n):
    return "adam_kernel" in n or "FusedOptimizerTensorListMetadata" in n or "FusedAdam" in n


def newest(d, pat):
    f = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return max(f, key=os.path.getmtime) if f else None


def from_trace(f, top):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ends = [i for i, r in enumerate(rows) if is_opt(r["Kernel_Name"]) and (i + 1 == len(rows) or not is_opt(rows[i + 1]["Kernel_Name"]))]
    if len(ends) < 3:
        return False
    use = ends[-min(11, len(ends)):]
    steps = len(use) - 1
    agg = defaultdict(lambda: [0, 0])
    for i in range(use[0] + 1, use[-1] + 1):
        a = agg[short(rows[i]["Kernel_Name"])]
        a[0] += 1
        a[1] += int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])
    wall = (int(rows[use[-1]]["End_Timestamp"]) - int(rows[use[0]]["End_Timestamp"])) / steps / 1e3
    report(agg, steps, top, f"steady state, last {steps} steps of the trace: {wall:.1f} us wall / step; ")
    return True


def report(agg, steps, top, head):
    tot_calls = sum(a[0] for a in agg.values())
    tot_ns = sum(a[1] for a in agg.values())
    print(f"{head}{tot_calls / steps:.1f} launches / step, {tot_ns / steps / 1e3:.1f} us kernel time / step")
    for name, (c, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"{name[:86]:86s} {c / steps:7.1f} x {ns / c / 1e3:8.1f} us = {ns / steps / 1e3:8.1f} us/step")


def main():
    d, top = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 25
    t = newest(d, "*kernel_trace.csv")
    if t and from_trace(t, top):
        return
    f = newest(d, "*kernel_stats.csv")
    rows = list(csv.DictReader(open(f)))
    steps = max([int(r["Calls"]) for r in rows if is_opt(r["Name"])] or [1])
    agg = defaultdict(lambda: [0, 0])
    for r in rows:
        a = agg[short(r["Name"])]
        a[0] += int(r["Calls"])
        a[1] += int(r["TotalDurationNs"])
    report(agg, steps, top, f"whole run (warm-up and capture included), {steps} steps: ")


if __name__ == "__main__":
    main()
