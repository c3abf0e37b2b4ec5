#!/bin/bash
# rocprofv3 kernel trace of the two-requests-in-flight region (bench.py --in-flight 2), summarised by
# scripts/trace_pipelined_summary.py: which hardware queue each lane's stream landed on, and how much of a request's tail
# ran under the other request's sort / accumulate.  With scripts/ab_trees/wb present (the tree at 1b467d4: round 4's
# regression, see scripts/ab_pipelined_bisect.sh) the same trace is taken there for comparison.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
rocprofv3 --kernel-trace -d gpurun_out/trace_head -o t --output-format csv -- python3 bench.py --headline-only --in-flight 2 --steps 12 --warmup 3 > gpurun_out/trace_head.json 2> gpurun_out/trace_head.err
echo "== HEAD"; python3 scripts/trace_pipelined_summary.py gpurun_out/trace_head/t_kernel_trace.csv
if [ -d scripts/ab_trees/wb ]; then
  (cd scripts/ab_trees/wb && rocprofv3 --kernel-trace -d $R/gpurun_out/trace_wb -o t --output-format csv -- python3 bench.py --headline-only --in-flight 2 --steps 12 --warmup 3 > $R/gpurun_out/trace_wb.json 2> $R/gpurun_out/trace_wb.err)
  echo "== tree at 1b467d4 (copy stream created between aux and the lanes' streams)"; python3 scripts/trace_pipelined_summary.py gpurun_out/trace_wb/t_kernel_trace.csv
fi
