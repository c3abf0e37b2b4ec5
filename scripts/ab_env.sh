#!/bin/bash
# Same-box A/B of ONE build under several environment settings on the commit+open rows (boxes of the pool differ by up to
# 10 %, so only numbers from one call compare).  SETTINGS: '|'-separated, each a comma-separated list of VAR=value.
#   gpurun -- 'SETTINGS="KZG_NTT_RADIX2=1|KZG_NTT_RADIX2=0,KZG_NTT_TILE_LOG=10" ROWS=22,20 bash scripts/ab_env.sh 3'
cd ${GRAFT_REPO_ROOT:-/root/repo}
ROUNDS=${1:-2}
IFS='|' read -ra SETS <<< "${SETTINGS:-KZG_NTT_RADIX2=0|KZG_NTT_RADIX2=1}"
for r in $(seq $ROUNDS); do
  for v in "${SETS[@]}"; do
    env ${v//,/ } python bench.py --no-adversarial --no-cpu-baseline --no-pipelined --steps 10 --kzg-rows ${ROWS:-22,20} 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v msm20 %.3f ms |' % d['ms_per_step'], ' '.join('%s %.3f [p10 %.3f p90 %.3f ntt %.4f poly %.4f]' % (k, v['ms'], v['p10'], v['p90'], v['stages_ms_profiled_serial']['ntt'], v['stages_ms_profiled_serial']['poly']) for k, v in d['kzg_commit_open'].items()))
"
  done
done
