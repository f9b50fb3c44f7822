# Attention line only: bench + kernel trace  ->  gpurun_out/attn/
cd $GRAFT_REPO_ROOT && R=$GRAFT_REPO_ROOT/gpurun_out/attn && rm -rf $R && mkdir -p $R
timeout -k 10 300 python bench.py --combine attention --no-cpu-baseline > $R/bench_attention.json 2> $R/bench_attention.log; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_attention -- python3 $GRAFT_REPO_ROOT/bench.py --combine attention --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > $R/prof_attention.log 2>&1; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT
python scripts/kstats.py $R/prof_attention 60 > $R/kstats_attention.txt 2>&1
rm -rf $R/prof_attention
