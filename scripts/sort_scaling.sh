#!/bin/bash
# Stage split of ONE MSM at growing sizes (is the sort still a few % of the step above 2^20?).  Output: gpurun_out/sort_scaling.log
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
: > gpurun_out/sort_scaling.log
for lg in ${SIZES:-20 21 22 23 24 26}; do
  python bench.py --headline-only --no-cpu-baseline --log-n $lg --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); s=d['stages_ms']
print('lg=$lg c=%d nwin=%d ms=%.3f Mpts/s=%.1f sort=%.3f acc=%.3f fold=%.3f tree=%.3f final=%.3f' % (d['config']['window_bits'], d['config']['windows'], d['ms_per_step'], d['value']/1e6, s['digits'], s['accumulate'], s['fixup'], s['tree'], s['final']))" >> gpurun_out/sort_scaling.log
done
cat gpurun_out/sort_scaling.log
