cd $GRAFT_REPO_ROOT && R=$GRAFT_REPO_ROOT/gpurun_out/r02 && mkdir -p $R && \
timeout -k 10 120 scripts/ubench/ub_launch_atomics > $R/ub_default.log 2>&1 && \
HIP_FORCE_DEV_KERNARG=1 timeout -k 10 120 scripts/ubench/ub_launch_atomics > $R/ub_devkernarg.log 2>&1 && \
timeout -k 10 300 python bench.py --no-cpu-baseline > $R/base_b2048.json 2> $R/base_b2048.log && \
timeout -k 10 300 python bench.py --no-cpu-baseline --batch 64 --steps 50 > $R/base_b64.json 2> $R/base_b64.log && \
HIP_FORCE_DEV_KERNARG=1 timeout -k 10 300 python bench.py --no-cpu-baseline --batch 64 --steps 50 --no-roofline > $R/base_b64_devkernarg.json 2> $R/base_b64_devkernarg.log && \
timeout -k 10 600 python -m pytest tests -q -m gpu -x > $R/pytest_gpu_base.log 2>&1; echo rc=$?
