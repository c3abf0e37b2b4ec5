cd $GRAFT_REPO_ROOT
for r in 1 2; do
for q in default 8; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  echo "== GPU_MAX_HW_QUEUES=$q: concurrency (requests/s with 1/2/4/8 host threads)"
  python scripts/concurrency_throughput.py 12 16 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    r=json.loads(l); print('  2^%d'%r['log2_T'], r['requests_per_s_1_threads'], r['requests_per_s_2_threads'], r['requests_per_s_4_threads'], r['requests_per_s_8_threads'])"
  echo "== GPU_MAX_HW_QUEUES=$q: 2^22 commit+open (two-lane form), ms"
  python bench.py --workload kzg22 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.readlines() if l.startswith('{')][-1]); print('  ', round(d['ms_per_step'],3))"
done
done
