#!/bin/bash
# Same-box A/B of library builds on the concurrent-request throughput (scripts/concurrency_throughput.py).
#   VARIANTS="A B" SIZES="12 16" bash scripts/ab_concurrency.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
cp zkp_subnet_amd/libkzg_mi355x.so /tmp/_orig.so
for v in ${VARIANTS:-A B}; do
  cp zkp_subnet_amd/ab/$v.so zkp_subnet_amd/libkzg_mi355x.so
  echo "== $v"
  python scripts/concurrency_throughput.py ${SIZES:-12 16} 2>&1 | tail -4
done
cp /tmp/_orig.so zkp_subnet_amd/libkzg_mi355x.so
