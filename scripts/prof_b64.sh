# kernel trace of the batch-64 line -> gpurun_out/q/prof_b64
cd $GRAFT_REPO_ROOT && R=$GRAFT_REPO_ROOT/gpurun_out/q && mkdir -p $R && rm -rf $R/prof_b64
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_b64 -- python3 $GRAFT_REPO_ROOT/bench.py --batch 64 --steps 50 --warmup 3 --no-cpu-baseline --no-roofline > $R/prof_b64.log 2>&1; echo "prof rc=$?"
find $R -name "*agent_info.csv" -delete
