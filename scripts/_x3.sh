cd $GRAFT_REPO_ROOT && R=$GRAFT_REPO_ROOT/gpurun_out/q && mkdir -p $R
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 3 --share-device --backend gloo --no-roofline > $R/dp2.json 2> $R/dp2.log; echo rc=$?; cut -c1-400 $R/dp2.json; tail -3 $R/dp2.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
