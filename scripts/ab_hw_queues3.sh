cd $GRAFT_REPO_ROOT
for r in 1 2; do
for q in 4 8 12 16; do
  export GPU_MAX_HW_QUEUES=$q
  echo "== GPU_MAX_HW_QUEUES=$q: concurrency (requests/s with 1/2/4/8 host threads), then 2^20 MSM one at a time / two in flight"
  python scripts/concurrency_throughput.py 12 16 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    r=json.loads(l); print('  2^%d'%r['log2_T'], r['requests_per_s_1_threads'], r['requests_per_s_2_threads'], r['requests_per_s_4_threads'], r['requests_per_s_8_threads'])"
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kzg-rows --no-adversarial --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.readlines() if l.startswith('{')][-1]); print('  ', round(d['ms_per_step'],4), round(d['pipelined']['ms_per_step'],4))"
done
done
