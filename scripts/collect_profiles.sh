# Copies what is judged from the scratch directory of scripts/measure_round.sh into the tracked profiles/r03/:
#   bash scripts/collect_profiles.sh        (run in the build container after the gpurun call returned)
S=gpurun_out/r03fin; D=profiles/r03; mkdir -p $D
cp $S/pytest_gpu.log $S/bench_*.json $S/kstats_*.txt $S/pmc_traffic.json $S/pmc_sq1.json $S/pmc_sq2.json $S/pmc_mfma.json $D/
cp $S/pmc_traffic_attention.json $S/pmc_mfma_attention.json $D/ 2>/dev/null
cp $S/ub_dense.txt $S/ub_dense_f32.txt $S/det_check.txt $D/ 2>/dev/null
for d in $S/prof_*; do [ -d "$d" ] || continue; n=$(basename $d); f=$(find $d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $D/${n}_kernel_stats.csv; done
ls $D | wc -l
