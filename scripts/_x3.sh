cd $GRAFT_REPO_ROOT && R=$GRAFT_REPO_ROOT/gpurun_out/q && mkdir -p $R
run() { name=$1; shift; timeout -k 10 400 python bench.py "$@" > $R/$name.json 2> $R/$name.log; python - <<PY
import json; d=json.load(open("$R/$name.json")); print("$name", d["value"], d["ms_per_step"], (d.get("roofline") or {}).get("frac"), {k:(v["avg_launch_ms"],v["algorithmic_GBps"]) for k,v in (d.get("kernels") or {}).items()})
PY
}
run bench_zinc_gd16 --workload zinc_gd16 --batch 512 --no-cpu-baseline
run bench_kpgin512 --model KPGIN --batch 512 --no-cpu-baseline
run bench_kpgin256 --model KPGIN --batch 256 --no-cpu-baseline
