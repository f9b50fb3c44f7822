# Headline line only: bench + kernel trace  ->  gpurun_out/dq/
cd $GRAFT_REPO_ROOT && R=$GRAFT_REPO_ROOT/gpurun_out/dq && rm -rf $R && mkdir -p $R
timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > $R/bench.json 2> $R/bench.log; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline "$@" > $R/prof.log 2>&1; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT
python scripts/kstats.py $R/prof 60 > $R/kstats.txt 2>&1
rm -rf $R/prof
