#!/bin/bash
# Same-box bisect of the `pipelined` regression VERDICT r4 reported (two requests in flight bought nothing on HEAD):
# Needs (git-ignored, built in the authoring container before the gpurun call): scripts/ab_trees/wa and wb = `git worktree`
# checkouts of a4f126c / 1b467d4 with their own built libkzg_mi355x.so, zkp_subnet_amd/ab/C.so and D.so.
#   wa = the tree at a4f126c (before the tile-streamed upload), wb = 1b467d4 (the commit that added the h2d copy stream),
#   C  = HEAD's library as built at round start, D = the candidate fix (h2d stream created after the lanes' streams),
#   Cq8 = C with GPU_MAX_HW_QUEUES=8.  Prints: variant, ms_per_step (one at a time), pipelined ms_per_step, live accumulate ms.
cd ${GRAFT_REPO_ROOT:-/root/repo}
ROOT=$PWD
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-kzg-rows --no-adversarial"
show='import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],4), round(d["pipelined"]["ms_per_step"],4), round(d["roofline"]["kernel_ms"],4))'
for r in 1 2 3; do
  for v in wa wb; do (cd scripts/ab_trees/$v && python bench.py $ARGS 2>/dev/null | python -c "$show" $v); done
  for v in C D; do KZG_MI355X_LIB=$ROOT/zkp_subnet_amd/ab/$v.so python bench.py $ARGS 2>/dev/null | python -c "$show" $v; done
  GPU_MAX_HW_QUEUES=8 KZG_MI355X_LIB=$ROOT/zkp_subnet_amd/ab/C.so python bench.py $ARGS 2>/dev/null | python -c "$show" Cq8
done
