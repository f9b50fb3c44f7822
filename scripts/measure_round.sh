cd $GRAFT_REPO_ROOT && R=$GRAFT_REPO_ROOT/gpurun_out/fin1 && mkdir -p $R && \
timeout -k 10 500 python -m pytest tests -q -m gpu -x > $R/pytest_gpu.log 2>&1 && \
timeout -k 10 500 python bench.py > $R/bench_default.json 2> $R/bench_default.log && \
timeout -k 10 300 python bench.py --model KPGIN --cpu-graphs 64 > $R/bench_kpgin.json 2> $R/bench_kpgin.log && \
cd /tmp && export TMPDIR=/tmp && \
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_default -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $R/prof_default.log 2>&1 && \
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_kpgin -- python3 $GRAFT_REPO_ROOT/bench.py --model KPGIN --steps 20 --warmup 3 --no-cpu-baseline > $R/prof_kpgin.log 2>&1 && \
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-graph > $R/pmc_fetch.log 2>&1 && \
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-graph > $R/pmc_write.log 2>&1 && \
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $R/pmc_mfma_kpgin -- python3 $GRAFT_REPO_ROOT/bench.py --model KPGIN --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-graph > $R/pmc_mfma_kpgin.log 2>&1 && \
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $R/pmc_mfma_default -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-graph > $R/pmc_mfma_default.log 2>&1; echo rc=$?
