#!/bin/bash
# Same-box A/B of two builds of libkzg_mi355x.so (boxes of the pool differ by up to 10 % on this VALU-bound path, so
# numbers from different gpurun calls are not comparable).  Put the candidates at zkp_subnet_amd/ab/A.so and B.so
# (git-ignored), then:  gpurun -- 'bash scripts/ab_bench.sh "<bench.py args>" [rounds]'
cd ${GRAFT_REPO_ROOT:-/root/repo}
ARGS=${1:---headline-only --steps 15}
ROUNDS=${2:-3}
for r in $(seq $ROUNDS); do
  for v in ${VARIANTS:-A B}; do
    echo -n "$v "; KZG_MI355X_LIB=$PWD/zkp_subnet_amd/ab/$v.so python bench.py $ARGS 2>&1 | tail -1 | grep -o "ms_per_step[^,]*\|\"accumulate[^,]*\|\"digits[^,]*\|\"tree[^,]*\|single_request_latency_ms[^,]*\|pipelined\": {[^}]*}" | tr "\n" " "; echo
  done
done
