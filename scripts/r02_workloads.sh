# usage: bash scripts/r02_workloads.sh TAG   (GPU box) - the non-default bench workloads, one JSON each
cd $GRAFT_REPO_ROOT && R=$GRAFT_REPO_ROOT/gpurun_out/r02 && mkdir -p $R && T=$1
run() { name=$1; shift; timeout -k 10 400 python bench.py "$@" > $R/${name}_$T.json 2> $R/${name}_$T.log; echo "$name rc=$?"; tail -2 $R/${name}_$T.log | cut -c1-300; }
run wl_qm9 --workload qm9 --cpu-graphs 64
run wl_regular_b1 --workload regular --batch 1 --steps 50
run wl_gd16 --workload zinc_gd16 --batch 512 --cpu-graphs 16
run wl_attention --combine attention --cpu-graphs 64
run wl_kpgin --model KPGIN --cpu-graphs 64
run wl_dense --dense-peripheral --no-cpu-baseline
run wl_b64 --batch 64 --steps 100 --no-cpu-baseline
python - <<PY
import json, glob
for f in sorted(glob.glob("$R/wl_*_$T.json")):
    try:
        d = json.load(open(f))
        print(f.split("/")[-1], d["value"], d["ms_per_step"], (d.get("roofline") or {}).get("frac"), (d.get("kernels") or {}).get("agg_fwd_kmax", {}).get("frac"), (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(f, "no result", e)
PY
