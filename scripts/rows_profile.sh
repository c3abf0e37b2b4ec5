#!/bin/bash
# rocprofv3 kernel stats of the commit+open of short rows (2^16 mainnet row, 2^12 testnet row)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/prof
for lg in 16 12; do
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof/kzg$lg -o kzg$lg --output-format csv -- python3 bench.py --workload kzg22 --log-n $lg --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/prof/bench_kzg${lg}_under_rocprof.json 2> gpurun_out/prof/rocprof_kzg$lg.err
done
