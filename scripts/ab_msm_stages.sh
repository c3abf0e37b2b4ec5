#!/bin/bash
# Same-box A/B of ONE build under several environment settings on the 2^20-point MSM, with the stage profile
# (every stage bracketed by HIP events).  SETTINGS as in ab_env.sh.
#   gpurun -- 'SETTINGS="KZG_TREE_PF_MIN=0|KZG_TREE_PF_MIN=131072" bash scripts/ab_msm_stages.sh 3'
cd ${GRAFT_REPO_ROOT:-/root/repo}
ROUNDS=${1:-2}
IFS='|' read -ra SETS <<< "${SETTINGS}"
for r in $(seq $ROUNDS); do
  for v in "${SETS[@]}"; do
    env ${v//,/ } python bench.py --no-adversarial --no-cpu-baseline --no-pipelined --steps 30 --no-kzg-rows 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
s=d['stages_ms']
print('$v msm20 %.3f ms | accumulate %.3f fixup %.4f tree %.4f final %.4f digits %.4f' % (d['ms_per_step'], s['accumulate'], s['fixup'], s['tree'], s['final'], s['digits']))
"
  done
done
