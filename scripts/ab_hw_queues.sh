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
