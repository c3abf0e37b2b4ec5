cd $GRAFT_REPO_ROOT
export BENCH_ONE_GPU=1 BENCH_COMM_INIT_TIMEOUT_S=60 MASTER_ADDR=127.0.0.1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29561 bench.py --gpus 2 --log-n 13 --steps 3 --warmup 1 --msm26-log 15 --kzg22-log 11 --cpu-sample-log 12 > gpurun_out/two_rank.out 2> gpurun_out/two_rank.err
echo rc=$?; tail -c 1500 gpurun_out/two_rank.out; grep -v "^W\|^\[W" gpurun_out/two_rank.err | tail -40
