# Quick round-2 check (one gpurun call): full GPU tests, the default / batch-64 / dense-peripheral bench lines and a
# kernel-trace of the default line -> gpurun_out/q/   usage: bash scripts/r02_quick.sh [notest]
cd $GRAFT_REPO_ROOT && R=$GRAFT_REPO_ROOT/gpurun_out/q && mkdir -p $R
if [ "$1" != "notest" ]; then timeout -k 10 900 python -m pytest tests -q -m gpu -x > $R/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $R/pytest_gpu.log; fi
run() { name=$1; shift; timeout -k 10 400 python bench.py "$@" > $R/$name.json 2> $R/$name.log; echo "$name rc=$? $(cut -c90-220 $R/$name.json)"; }
run bench_default --no-cpu-baseline
run bench_b64 --batch 64 --steps 100 --no-cpu-baseline
run bench_dense --dense-peripheral --no-cpu-baseline
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_default -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > $R/prof_default.log 2>&1; echo "prof rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_dense -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --dense-peripheral --no-cpu-baseline --no-roofline > $R/prof_dense.log 2>&1; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT && python scripts/kstats.py $R/prof_default 40 > $R/kstats_default.txt 2>&1; python scripts/kstats.py $R/prof_dense 16 > $R/kstats_dense.txt 2>&1
find $R -name "*agent_info.csv" -delete; find $R -name "*kernel_trace.csv" -size +20M -delete
