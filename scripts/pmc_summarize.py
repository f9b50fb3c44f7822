"""Turns rocprofv3 --pmc counter_collection.csv files into the per-kernel JSON summaries kept under profiles/.

  python scripts/pmc_summarize.py traffic  <fetch_dir> <write_dir> <out.json> <workload_key> [csrc_digest]
  python scripts/pmc_summarize.py mfma     <pmc_dir> <out.json>
  python scripts/pmc_summarize.py sq       <pmc_dir> <out.json>

traffic: HBM bytes per launch = 2 x FETCH_SIZE(KB) x 1024 + WRITE_SIZE(KB) x 1024 (gfx950 correction of
/opt/skills/guides/MI355X_MICROARCH.md, HBM section: 16-B-per-lane streaming reads are tallied at half), averaged
over the launches of each kernel.
sq: per kernel, averages of the SQ wave counters of one --pmc pass (SQ_WAVE_CYCLES, SQ_WAIT_ANY, SQ_WAIT_INST_ANY,
SQ_ACTIVE_INST_ANY, SQ_INSTS_VALU, SQ_BUSY_CYCLES ...) and the derived wait fraction SQ_WAIT_ANY / SQ_WAVE_CYCLES.
mfma: per kernel, sums of SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES / SQ_INSTS_VALU_MFMA_MOPS_F32 and
GRBM_GUI_ACTIVE and the derived matrix-core busy fraction.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(\w+_kernel)", name)
    if not m:
        return name.split("(")[0][:60]
    k = m.group(1)
    if k == "agg_fwd_kernel":
        # agg_fwd_kernel<VEC, G, GCN, TAB, FAST, BF>: the table-less non-FAST instantiation is, in the KP-GIN+ workloads, the PULL
        # form of the backward gather (ops.khop_pull_gather) - kept apart from the forward launches it would be averaged with
        a = re.search(r"agg_fwd_kernel<([^>]*)>", name)
        if a:
            t = [x.strip() for x in a.group(1).split(",")]
            if len(t) >= 5 and t[3] == "0" and t[4] == "false":
                return "agg_pull_kernel"
    return k


def load(d):
    rows = defaultdict(lambda: defaultdict(list))      # kernel -> counter -> values (one per dispatch)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return rows


def main():
    mode = sys.argv[1]
    if mode == "traffic":
        fetch, write, out, key = load(sys.argv[2]), load(sys.argv[3]), sys.argv[4], sys.argv[5]
        res = {}
        for k in sorted(set(fetch) & set(write)):
            if not k.endswith("_kernel") or "FETCH_SIZE" not in fetch[k] or "WRITE_SIZE" not in write[k]:
                continue
            fa = sum(fetch[k]["FETCH_SIZE"]) / len(fetch[k]["FETCH_SIZE"])
            wa = sum(write[k]["WRITE_SIZE"]) / len(write[k]["WRITE_SIZE"])
            res[k] = {"FETCH_SIZE_KB_avg": round(fa, 1), "WRITE_SIZE_KB_avg": round(wa, 1),
                      "launches_sampled": len(fetch[k]["FETCH_SIZE"]),
                      "traffic_bytes_per_launch": int(2 * fa * 1024 + wa * 1024)}
        doc = {"note": "HBM traffic per launch from the PMC counters, averaged over the launches of the bench's own step "
                       "mix, collected in two separate --pmc passes with --kernel-trace only and corrected as "
                       "/opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes for gfx950: bytes = 2 x "
                       "FETCH_SIZE(KB) x 1024 (16-B-per-lane streaming reads are tallied at half) + WRITE_SIZE(KB) x 1024.",
               "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 bench.py --steps 4 --warmup 2 "
                          "--no-cpu-baseline --no-roofline --no-graph   (and the same with --pmc WRITE_SIZE)",
               "workload_key": key, "csrc_digest": sys.argv[6] if len(sys.argv) > 6 else None, "kernels": res}
    elif mode == "sq":
        rows, out = load(sys.argv[2]), sys.argv[3]
        res = {}
        for k, c in sorted(rows.items()):
            if not k.endswith("_kernel") or "SQ_WAVE_CYCLES" not in c:
                continue
            e = {"launches_sampled": len(c["SQ_WAVE_CYCLES"])}
            for name, vals in sorted(c.items()):
                e[name + "_avg"] = round(sum(vals) / len(vals), 1)
            if e.get("SQ_WAVE_CYCLES_avg", 0) > 0 and "SQ_WAIT_ANY_avg" in e:
                e["wait_frac"] = round(e["SQ_WAIT_ANY_avg"] / e["SQ_WAVE_CYCLES_avg"], 4)
            if e.get("SQ_WAVE_CYCLES_avg", 0) > 0 and "SQ_ACTIVE_INST_ANY_avg" in e:
                e["issue_frac"] = round(e["SQ_ACTIVE_INST_ANY_avg"] / e["SQ_WAVE_CYCLES_avg"], 4)
            res[k] = e
        doc = {"note": "SQ wave counters of one rocprofv3 --pmc pass (kernel-trace only, eager launches of the default bench "
                       "step): wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES (share of wave-cycles spent in s_waitcnt), issue_frac = "
                       "SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES.", "kernels": res}
    else:
        rows, out = load(sys.argv[2]), sys.argv[3]
        res = {}
        for k, c in sorted(rows.items()):
            if not k.endswith("_kernel") or "SQ_VALU_MFMA_BUSY_CYCLES" not in c:
                continue
            mf, n = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]), len(c["SQ_VALU_MFMA_BUSY_CYCLES"])
            if mf == 0:
                continue
            e = {"launches_sampled": n, "SQ_VALU_MFMA_BUSY_CYCLES_avg": round(mf / n, 1)}
            for name in ("SQ_BUSY_CU_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F32", "GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES"):
                if name in c:
                    e[name + "_avg"] = round(sum(c[name]) / len(c[name]), 1)
            if e.get("GRBM_GUI_ACTIVE_avg", 0) > 0:
                # MFMA_BUSY is summed over the chip's 1024 SIMDs (checked: linear_fwd issues 308k 64-cycle MFMAs per
                # launch = 19.7 M); GRBM_GUI_ACTIVE is summed over the 8 XCDs (microarch guide, DVFS note)
                e["mfma_util"] = round(e["SQ_VALU_MFMA_BUSY_CYCLES_avg"] / (1024.0 * e["GRBM_GUI_ACTIVE_avg"] / 8.0), 4)
            res[k] = e
        doc = {"note": "matrix-core utilisation of the dense kernels from one --pmc pass (kernel-trace only, eager "
                       "launches): mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), i.e. the "
                       "fraction of the fp32 matrix-core peak (157 TFLOP/s dense) while the kernel is resident; dispatch "
                       "durations under --pmc run ~25 % long, so the unprofiled utilisation is higher by that factor.",
               "kernels": res}
    with open(out, "w") as fh:
        json.dump(doc, fh, indent=1)
    print(json.dumps(doc["kernels"], indent=1))


if __name__ == "__main__":
    main()
