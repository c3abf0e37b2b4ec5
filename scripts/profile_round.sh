cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/prof
cd $R
python bench.py --pipelined --no-cpu-baseline --no-adversarial > gpurun_out/prof/bench_msm20_pipelined.json 2>> gpurun_out/prof/bench.err
python bench.py > gpurun_out/prof/bench_msm20.json 2> gpurun_out/prof/bench_msm20.err
python bench.py --workload kzg22 > gpurun_out/prof/bench_kzg22.json 2> gpurun_out/prof/bench_kzg22.err
python bench.py --workload kzg22 --log-n 12 > gpurun_out/prof/bench_kzg12.json 2>> gpurun_out/prof/bench.err
python bench.py --workload kzg22 --log-n 16 > gpurun_out/prof/bench_kzg16.json 2>> gpurun_out/prof/bench.err
rocprofv3 --kernel-trace --stats -d gpurun_out/prof/msm20 -o msm20 --output-format csv -- python3 bench.py --no-adversarial --no-cpu-baseline > gpurun_out/prof/bench_msm20_under_rocprof.json 2> gpurun_out/prof/rocprof_msm20.err
rocprofv3 --kernel-trace --stats -d gpurun_out/prof/kzg22 -o kzg22 --output-format csv -- python3 bench.py --workload kzg22 --steps 10 > gpurun_out/prof/bench_kzg22_under_rocprof.json 2> gpurun_out/prof/rocprof_kzg22.err
find gpurun_out/prof -name "*kernel_stats.csv" | head; 
tail -c 1500 gpurun_out/prof/bench_kzg22.json
