# Round-3 measurements, one gpurun call:  bash scripts/measure_round.sh   -> gpurun_out/r03fin/ (copy what is judged to profiles/r03/)
cd $GRAFT_REPO_ROOT && R=$GRAFT_REPO_ROOT/gpurun_out/r03fin && rm -rf $R && mkdir -p $R
B="python bench.py"
PB="python3 $GRAFT_REPO_ROOT/bench.py"
run() { name=$1; shift; timeout -k 10 400 $B "$@" > $R/$name.json 2> $R/$name.log; echo "$name rc=$?"; }
timeout -k 10 900 python -m pytest tests -q -m gpu > $R/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $R/pytest_gpu.log
run bench_default
run bench_fresh --fresh-batches
run bench_fresh_eager --fresh-batches --no-graph --no-cpu-baseline --no-roofline
run bench_b64 --batch 64 --steps 100
run bench_kpgin --model KPGIN --cpu-graphs 64
run bench_attention --combine attention --cpu-graphs 64
run bench_bf16 --dtype bf16 --no-cpu-baseline
run bench_dense_peripheral --dense-peripheral --no-cpu-baseline
run bench_qm9 --workload qm9 --cpu-graphs 64
run bench_regular_b1 --workload regular --batch 1 --steps 50
run bench_regular_b100 --workload regular --batch 100 --steps 10 --warmup 2 --num-batches 1 --no-cpu-baseline
run bench_zinc_gd16 --workload zinc_gd16 --cpu-graphs 16 --num-batches 2
run bench_strong1 --scaling strong --no-cpu-baseline --no-roofline
timeout -k 10 200 python scripts/ub_dense.py > $R/ub_dense.txt 2>&1; echo "ub_dense rc=$?"
UB_MATH=f32 timeout -k 10 200 python scripts/ub_dense.py > $R/ub_dense_f32.txt 2>&1; echo "ub_dense f32 rc=$?"
timeout -k 10 200 python scripts/det_check.py > $R/det_check.txt 2>&1; echo "det_check rc=$?"
cd /tmp && export TMPDIR=/tmp
prof() { name=$1; shift; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_$name -- $PB "$@" --no-cpu-baseline --no-roofline > $R/prof_$name.log 2>&1; echo "prof $name rc=$?"; }
prof default --steps 20 --warmup 3
prof fresh --fresh-batches --steps 20 --warmup 3
prof b64 --batch 64 --steps 50 --warmup 3
prof kpgin --model KPGIN --steps 20 --warmup 3
prof attention --combine attention --steps 20 --warmup 3
prof bf16 --dtype bf16 --steps 20 --warmup 3
prof zinc_gd16 --workload zinc_gd16 --num-batches 2 --steps 10 --warmup 3
prof qm9 --workload qm9 --steps 50 --warmup 3
prof regular_b1 --workload regular --batch 1 --steps 50 --warmup 3
prof regular_b100 --workload regular --batch 100 --steps 10 --warmup 2 --num-batches 1
prof dense_peripheral --dense-peripheral --steps 10 --warmup 3
pmc() { name=$1; shift; timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/pmc_$name -- $PB --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-graph > $R/pmc_$name.log 2>&1; echo "pmc $name rc=$?"; }
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVES
pmc sq2 SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_ANY
pmc mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE
pmca() { name=$1; shift; timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/pmc_$name -- $PB --combine attention --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-graph > $R/pmc_$name.log 2>&1; echo "pmc $name rc=$?"; }
pmca fetch_att FETCH_SIZE
pmca write_att WRITE_SIZE
pmca mfma_att SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE
cd $GRAFT_REPO_ROOT
for n in default fresh b64 kpgin attention bf16 zinc_gd16 qm9 regular_b1 regular_b100 dense_peripheral; do python scripts/kstats.py $R/prof_$n 48 > $R/kstats_$n.txt 2>&1; done
DIG=$(python -c "import bench; print(bench.csrc_digest())")
python scripts/pmc_summarize.py traffic $R/pmc_fetch $R/pmc_write $R/pmc_traffic.json "zinc|KPGINPlus|B2048|K8|L8|h104|geometric" $DIG > /dev/null 2>&1; echo "traffic rc=$?"
# the headline line again, now that the traffic figure of THESE sources exists (bench.py reads profiles/r03/pmc_traffic.json)
mkdir -p profiles/r03 && cp $R/pmc_traffic.json profiles/r03/pmc_traffic.json && run bench_default
python scripts/pmc_summarize.py sq $R/pmc_sq1 $R/pmc_sq1.json > /dev/null 2>&1; python scripts/pmc_summarize.py sq $R/pmc_sq2 $R/pmc_sq2.json > /dev/null 2>&1
python scripts/pmc_summarize.py mfma $R/pmc_mfma $R/pmc_mfma.json > /dev/null 2>&1
python scripts/pmc_summarize.py traffic $R/pmc_fetch_att $R/pmc_write_att $R/pmc_traffic_attention.json "zinc|KPGINPlus|B2048|K8|L8|h104|attention" $DIG > /dev/null 2>&1; echo "traffic attention rc=$?"
python scripts/pmc_summarize.py mfma $R/pmc_mfma_att $R/pmc_mfma_attention.json > /dev/null 2>&1
# (what is kept must fit the 64 MiB that travel back: the summaries above are made, the raw traces and counter dumps go)
find $R -name "*agent_info.csv" -delete; find $R -name "*kernel_trace.csv" -delete; find $R -name "*counter_collection.csv" -delete
du -sh $R | tail -1
echo done
