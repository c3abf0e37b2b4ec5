#!/bin/bash
# Lanes per context and stream creation order, with the library's eight hardware queues (same box):
#   shipped  4 lanes, streams created lane by lane (stream, verification stream, ...)
#   MSF      4 lanes, all lanes' MAIN streams created first, then the verification side streams (-DKZG_MAIN_STREAMS_FIRST)
#   L6 / L8  6 / 8 lanes, main streams first
# requests/s through Client.worker_commit_and_open with 1 / 2 / 4 / 8 host threads on one context at 2^12 and 2^16, then the
# 2^20 MSM one at a time / two in flight.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2; do
for v in shipped MSF L6 L8; do
  if [ $v = shipped ]; then unset KZG_MI355X_LIB; else export KZG_MI355X_LIB=$PWD/zkp_subnet_amd/ab/$v.so; fi
  echo "== $v"
  python scripts/concurrency_throughput.py 12 16 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    r=json.loads(l); print('  2^%d'%r['log2_T'], r['requests_per_s_1_threads'], r['requests_per_s_2_threads'], r['requests_per_s_4_threads'], r['requests_per_s_8_threads'])"
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kzg-rows --no-adversarial --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.readlines() if l.startswith('{')][-1]); print('  ', round(d['ms_per_step'],4), round(d['pipelined']['ms_per_step'],4))"
done
done
