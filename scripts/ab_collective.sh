#!/bin/bash
# Same-box A/B of the collective step of the SRS-sharded MSM on a ONE-rank group (this pool's boxes have one GPU): what the
# exchange costs a 2^20 step beyond the plain single-GPU MSM.
#   plain    python bench.py (no process group)
#   library  BENCH_FORCE_DIST=1: the library's own communicator, ncclAllGather on the lane's stream (kzg_msm_sharded)
#   torch    BENCH_FORCE_DIST=1 BENCH_BACKEND=nccl BENCH_COLLECTIVE=torch: torch's RCCL group, lane -> torch's stream -> lane
#            chained by events (kzg_msm_sharded_begin / _finish): the round-4 form
# Prints: variant, ms_per_step, single-request latency, live accumulate ms, collective description.
cd ${GRAFT_REPO_ROOT:-/root/repo}
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-kzg-rows --no-adversarial --no-e2e --no-dist-extra --no-pipelined"
show='import json,sys
ls=[l for l in sys.stdin.readlines() if l.startswith("{")]
d=json.loads(ls[-1]); print(sys.argv[1], round(d["ms_per_step"],4), d["single_request_latency_ms"], round(d["roofline"]["kernel_ms"],4), (d["config"].get("collective") or "-")[:60])'
for r in 1 2 3; do
  python bench.py $ARGS 2>/dev/null | python -c "$show" plain
  BENCH_FORCE_DIST=1 MASTER_PORT=$((29600+r)) python bench.py $ARGS 2>/dev/null | python -c "$show" library
  BENCH_FORCE_DIST=1 BENCH_BACKEND=nccl BENCH_COLLECTIVE=torch MASTER_PORT=$((29610+r)) python bench.py $ARGS 2>/dev/null | python -c "$show" torch
done
