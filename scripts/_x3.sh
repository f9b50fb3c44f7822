cd $GRAFT_REPO_ROOT && R=$GRAFT_REPO_ROOT/gpurun_out/q && mkdir -p $R
timeout -k 10 900 python -m pytest tests -q -m gpu -x > $R/pt.log 2>&1; tail -2 $R/pt.log
run() { name=$1; shift; timeout -k 10 400 python bench.py "$@" > $R/$name.json 2> $R/$name.log; python - <<PY
import json; d=json.load(open("$R/$name.json")); print("$name", d["value"], d["ms_per_step"], (d.get("roofline") or {}).get("frac"), {k:(v["avg_launch_ms"],v["algorithmic_GBps"]) for k,v in (d.get("kernels") or {}).items()})
PY
}
run bench_default --no-cpu-baseline
run bench_kpgin --model KPGIN --no-cpu-baseline
run bench_b64 --batch 64 --steps 100 --no-cpu-baseline
