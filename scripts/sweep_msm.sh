#!/bin/bash
# Stage profile of the single-GPU MSM over sizes / windows:  SIZES="20 22" WINDOWS="0 21" bash scripts/sweep_msm.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
for lg in ${SIZES:-20 21 22 23 24}; do
  for w in ${WINDOWS:-0}; do
    python bench.py --headline-only --steps 10 --warmup 3 --log-n $lg --window $w 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
s=d['stages_ms']; c=d['config']
adds=(1<<$lg)*-(-256//c['window_bits'])
print('2^$lg c=%d  %.3f ms/step | accumulate %.3f (live %.3f: %.4f ns/add) tree %.3f fixup %.3f final %.3f digits %.3f' % (c['window_bits'], d['ms_per_step'], s['accumulate'], d['roofline']['kernel_ms'], d['roofline']['kernel_ms']*1e6/adds, s['tree'], s['fixup'], s['final'], s['digits']))
"
  done
done
