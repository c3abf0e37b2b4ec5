#!/bin/bash
# Round evidence on the GPU box: default bench line (what the driver runs), msm26 / kzg22 workloads, rocprofv3 kernel
# stats of the headline and of the 2^22 commit+open.  Outputs under gpurun_out/prof; copy what is judged to profiles/.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/prof
cd $R
python bench.py > gpurun_out/prof/bench_default.json 2> gpurun_out/prof/bench_default.err
python bench.py --workload kzg22 --steps 10 --warmup 3 > gpurun_out/prof/bench_kzg22.json 2> gpurun_out/prof/bench_kzg22.err
python bench.py --workload msm26 --steps 10 --warmup 2 > gpurun_out/prof/bench_msm26.json 2> gpurun_out/prof/bench_msm26.err
rocprofv3 --kernel-trace --stats -d gpurun_out/prof/msm20 -o msm20 --output-format csv -- python3 bench.py --headline-only > gpurun_out/prof/bench_msm20_under_rocprof.json 2> gpurun_out/prof/rocprof_msm20.err
rocprofv3 --kernel-trace --stats -d gpurun_out/prof/kzg22 -o kzg22 --output-format csv -- python3 bench.py --workload kzg22 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/prof/bench_kzg22_under_rocprof.json 2> gpurun_out/prof/rocprof_kzg22.err
find gpurun_out/prof -name "*kernel_stats.csv" | head
