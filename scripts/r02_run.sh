# usage: bash scripts/r02_run.sh TAG [prof]  (GPU box) - GPU tests, the default bench line, the batch-64 line, optional rocprofv3 stats
cd $GRAFT_REPO_ROOT && R=$GRAFT_REPO_ROOT/gpurun_out/r02 && mkdir -p $R && T=$1 && \
timeout -k 10 900 python -m pytest tests -q -m gpu -x > $R/pytest_$T.log 2>&1; echo "pytest rc=$?"; tail -4 $R/pytest_$T.log; \
timeout -k 10 300 python bench.py --no-cpu-baseline > $R/b2048_$T.json 2> $R/b2048_$T.log; echo "bench rc=$?"; \
timeout -k 10 300 python bench.py --no-cpu-baseline --batch 64 --steps 50 --no-roofline > $R/b64_$T.json 2> $R/b64_$T.log; echo "bench64 rc=$?"; \
python - <<PY
import json
for f in ("b2048_$T", "b64_$T"):
    try:
        d = json.load(open("$R/%s.json" % f))
        print(f, d["value"], d["ms_per_step"], d.get("roofline", {}).get("frac"), {k: v["avg_launch_ms"] for k, v in d.get("kernels", {}).items()})
    except Exception as e:
        print(f, "no result", e)
PY
if [ "$2" = "prof" ]; then
  cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_$T -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > $R/prof_$T.log 2>&1; echo "prof rc=$?"
  cd $GRAFT_REPO_ROOT && python scripts/kstats.py $R/prof_$T 30
fi
