cd $GRAFT_REPO_ROOT
python -m pytest tests/test_wavtok_gpu_parity.py -x -q 2>&1 | tail -3
python bench.py --codec wavtokenizer --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_r2l_wt.json 2>/dev/null; echo "wt rc $?"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_r2l_torchrun1.json 2> gpurun_out/bench_r2l_torchrun1.err; echo "torchrun rc $?"; tail -3 gpurun_out/bench_r2l_torchrun1.err
