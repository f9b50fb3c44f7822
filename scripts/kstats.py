"""Per-step kernel summary from a rocprofv3 --kernel-trace --stats output directory:
    python scripts/kstats.py <dir> [top_n]
Groups the kernel_stats.csv rows by short kernel name and prints launches and microseconds per step (the step count
is taken from the fused-Adam launches), so that a bench step can be read as a budget."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(n):
    m = re.search(r"(\w+_kernel)(<[^(]*>)?", n)
    if m:
        return m.group(1) + (m.group(2) or "")
    return n.split("(")[0][:70]


def main():
    d, top = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 25
    f = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    steps = max([int(r["Calls"]) for r in rows if "FusedOptimizerTensorListMetadata" in r["Name"] or "FusedAdam" in r["Name"]] or [1])
    agg = defaultdict(lambda: [0, 0])
    for r in rows:
        a = agg[short(r["Name"])]
        a[0] += int(r["Calls"])
        a[1] += int(r["TotalDurationNs"])
    tot_calls = sum(a[0] for a in agg.values())
    tot_ns = sum(a[1] for a in agg.values())
    print(f"steps {steps}: {tot_calls / steps:.1f} launches / step, {tot_ns / steps / 1e3:.1f} us kernel time / step")
    for name, (c, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"{name[:86]:86s} {c / steps:7.1f} x {ns / c / 1e3:8.1f} us = {ns / steps / 1e3:8.1f} us/step")


if __name__ == "__main__":
    main()
