#!/bin/bash
# Same-box A/B of library builds on the concurrent-request throughput (scripts/concurrency_throughput.py).
#   VARIANTS="A B" SIZES="12 16" bash scripts/ab_concurrency.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in ${VARIANTS:-A B}; do
  echo "== $v"
  KZG_MI355X_LIB=$PWD/zkp_subnet_amd/ab/$v.so python scripts/concurrency_throughput.py ${SIZES:-12 16} 2>&1 | tail -4
done
