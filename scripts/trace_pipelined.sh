cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
rocprofv3 --kernel-trace -d gpurun_out/trace2 -o t --output-format csv -- python3 bench.py --headline-only --in-flight 2 --steps 8 --warmup 3 > gpurun_out/trace2.json 2> gpurun_out/trace2.err
ls gpurun_out/trace2
