#!/bin/bash
# Same-box A/B of two library builds on the commit+open rows and the headline MSM (see scripts/ab_bench.sh).
cd ${GRAFT_REPO_ROOT:-/root/repo}
ROUNDS=${1:-2}
for r in $(seq $ROUNDS); do
  for v in ${VARIANTS:-A B}; do
    KZG_MI355X_LIB=$PWD/zkp_subnet_amd/ab/$v.so python bench.py --no-adversarial --no-cpu-baseline --no-pipelined --steps 20 --kzg-rows ${ROWS:-22,20,16,12} 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
s=d['stages_ms']
print('$v msm20 %.3f ms  acc %.3f tree %.3f final %.3f digits %.3f |' % (d['ms_per_step'], s['accumulate'], s['tree'], s['final'], s['digits']), ' '.join('%s %.3f (ntt %.3f poly %.3f dig %.3f fix %.3f tree %.3f fin %.3f)' % (k, v['ms'], v['stages_ms_profiled_serial']['ntt'], v['stages_ms_profiled_serial']['poly'], v['stages_ms_profiled_serial']['digits'], v['stages_ms_profiled_serial']['fixup'], v['stages_ms_profiled_serial']['tree'], v['stages_ms_profiled_serial']['final']) for k, v in d['kzg_commit_open'].items()))
"
  done
done
