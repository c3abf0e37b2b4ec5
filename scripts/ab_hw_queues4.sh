cd ${GRAFT_REPO_ROOT:-/root/repo}
python scripts/ab_hw_queues_two_processes.py 4 5 6 8
for q in 5 6; do
  export GPU_MAX_HW_QUEUES=$q
  echo "== GPU_MAX_HW_QUEUES=$q: concurrency (requests/s with 1/2/4/8 host threads)"
  python scripts/concurrency_throughput.py 12 16 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    r=json.loads(l); print('  2^%d'%r['log2_T'], r['requests_per_s_1_threads'], r['requests_per_s_2_threads'], r['requests_per_s_4_threads'], r['requests_per_s_8_threads'])"
done
unset GPU_MAX_HW_QUEUES
bash scripts/ab_lanes.sh
