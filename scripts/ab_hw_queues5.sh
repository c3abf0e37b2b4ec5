cd ${GRAFT_REPO_ROOT:-/root/repo}
show='import json,sys
ls=[l for l in sys.stdin.readlines() if l.startswith("{")]
d=json.loads(ls[-1]); print(sys.argv[1], "headline", round(d["ms_per_step"],3), "pipelined", round(d["pipelined"]["ms_per_step"],3), "pianist", round(d["pianist_kzg22"]["ms_per_step"],3), "msm26", round(d["msm26"]["ms_per_step"],3))'
for r in 1 2; do
for q in 4 8; do
  GPU_MAX_HW_QUEUES=$q BENCH_ONE_GPU=1 python bench.py --gpus 2 --steps 10 --warmup 3 --msm26-log 22 --kzg22-log 18 --no-cpu-baseline 2>/dev/null | python -c "$show" "two ranks on one GPU, GPU_MAX_HW_QUEUES=$q:"
done
done
