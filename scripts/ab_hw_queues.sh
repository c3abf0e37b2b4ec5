#!/bin/bash
# GPU_MAX_HW_QUEUES against the overlap of two MSMs in flight, in this round's stream order (head) and in round 4's (wb: needs
# scripts/ab_trees/wb = a `git worktree` checkout of 1b467d4 with its own built library, see scripts/ab_pipelined_bisect.sh).
# Prints: variant, ms per step one at a time, ms per step with two in flight.  Block 1 of profiles/r05_ab_hw_queues.log.
cd $GRAFT_REPO_ROOT
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-kzg-rows --no-adversarial"
show='import json,sys
d=json.loads([l for l in sys.stdin.readlines() if l.startswith("{")][-1]); print(sys.argv[1], round(d["ms_per_step"],4), round(d["pipelined"]["ms_per_step"],4))'
for r in 1 2; do
python bench.py $ARGS --no-e2e 2>/dev/null | python -c "$show" head_default
GPU_MAX_HW_QUEUES=8 python bench.py $ARGS --no-e2e 2>/dev/null | python -c "$show" head_q8
(cd scripts/ab_trees/wb && python bench.py $ARGS 2>/dev/null | python -c "$show" wb_default)
(cd scripts/ab_trees/wb && GPU_MAX_HW_QUEUES=8 python bench.py $ARGS 2>/dev/null | python -c "$show" wb_q8)
(cd scripts/ab_trees/wb && GPU_MAX_HW_QUEUES=2 python bench.py $ARGS 2>/dev/null | python -c "$show" wb_q2)
GPU_MAX_HW_QUEUES=2 python bench.py $ARGS --no-e2e 2>/dev/null | python -c "$show" head_q2
done
