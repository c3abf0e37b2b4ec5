#!/bin/bash
# Runtime environment knobs against the latency-bound parts (same box): the 2^20 MSM one at a time / two in flight and the
# device-resident commit+open of 2^16 / 2^12 / 2^10 rows (~40 dependent launches each).
cd ${GRAFT_REPO_ROOT:-/root/repo}
show='import json,sys
d=json.loads([l for l in sys.stdin.readlines() if l.startswith("{")][-1]); r=d["kzg_commit_open"]
print(sys.argv[1], round(d["ms_per_step"],4), round(d["pipelined"]["ms_per_step"],4), "rows", r["2^16"]["ms"], r["2^12"]["ms"], r["2^10"]["ms"])'
for r in 1 2 3; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-adversarial --no-e2e --kzg-rows 16,12,10 2>/dev/null | python -c "$show" baseline
  HIP_FORCE_DEV_KERNARG=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-adversarial --no-e2e --kzg-rows 16,12,10 2>/dev/null | python -c "$show" dev_kernarg
  HSA_NO_SCRATCH_RECLAIM=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-adversarial --no-e2e --kzg-rows 16,12,10 2>/dev/null | python -c "$show" no_scratch_reclaim
  HIP_FORCE_DEV_KERNARG=1 HSA_NO_SCRATCH_RECLAIM=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-adversarial --no-e2e --kzg-rows 16,12,10 2>/dev/null | python -c "$show" both
done
