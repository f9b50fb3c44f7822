# Attention line only: bench + kernel trace  ->  gpurun_out/attn/
cd $GRAFT_REPO_ROOT && R=$GRAFT_REPO_ROOT/gpurun_out/attn && rm -rf $R && mkdir -p $R
timeout -k 10 300 python bench.py --combine attention --no-cpu-baseline > $R/bench_attention.json 2> $R/bench_attention.log; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_attention -- python3 $GRAFT_REPO_ROOT/bench.py --combine attention --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > $R/prof_attention.log 2>&1; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT
python scripts/kstats.py $R/prof_attention 60 > $R/kstats_attention.txt 2>&1
python - $R/prof_attention > $R/each_attention.txt 2>&1 <<'PY'
import csv, glob, os, sys
f = max(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
for r in rows[ends[-2] + 1:ends[-1] + 1]:
    n = r["Kernel_Name"]
    if "attn" in n or "agg_" in n or "wgrad3" in n:
        print(f'{(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:8.1f} us  grid {r.get("Grid_Size", "?"):>8}  {n[:90]}')
PY
rm -rf $R/prof_attention
