#!/bin/bash
# After `gpurun -- 'bash scripts/evidence_round.sh'`: the round's evidence from gpurun_out/ into profiles/<round>_*.
# Bench lines go through scripts/evidence_keep.py (refused unless their identity.source_sha16 is this tree's, and unless no
# other evidence file has the same bytes); kernel stats, PMC tables and logs are copied.   ROUND=r06 bash scripts/evidence_collect.sh
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
ROUND=${ROUND:-r06}; P=gpurun_out/prof; rc=0
keep() { python scripts/evidence_keep.py "$@" || rc=1; }
keep $P/bench_driver_style.json profiles/${ROUND}_bench_driver_style.json \
     $P/bench_default.json profiles/${ROUND}_bench_default.json \
     $P/bench_kzg22.json profiles/${ROUND}_bench_kzg22.json \
     $P/bench_msm26.json profiles/${ROUND}_bench_msm26.json \
     $P/bench_msm20_under_rocprof.json profiles/${ROUND}_bench_msm20_under_rocprof.json \
     $P/bench_kzg22_under_rocprof.json profiles/${ROUND}_bench_kzg22_under_rocprof.json \
     $P/bench_two_ranks_one_gpu.json profiles/${ROUND}_bench_two_ranks_one_gpu_selftest.json \
     $P/bench_one_rank_library_collective.json profiles/${ROUND}_bench_one_rank_library_collective.json
for k in msm20 kzg22 kzg16 kzg12; do
  f=$(find $P/$k -name "*kernel_stats.csv" -newer scripts/evidence_collect.sh 2>/dev/null | head -1)
  [ -z "$f" ] && f=$(find $P/$k -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" profiles/${ROUND}_${k}_kernel_stats.csv || { echo "MISSING kernel stats of $k"; rc=1; }
done
cp $P/${ROUND}_msm20_pmc.csv profiles/${ROUND}_msm20_pmc.csv 2>/dev/null || { echo "MISSING pmc csv"; rc=1; }
cp $P/pmc_drift.log profiles/${ROUND}_pmc_drift_check.log
for n in e2e_latency concurrency start_time validator_step; do
  dst=$n; [ $n = concurrency ] && dst=concurrency_throughput
  cp $P/$n.json profiles/${ROUND}_$dst.json || rc=1
done
secs=$(grep -o "fuzz ok" $P/fuzz.log >/dev/null && echo ok || echo FAILED)
cp $P/fuzz.log profiles/${ROUND}_fuzz_${FUZZ_S:-420}s_seed${FUZZ_SEED:-20261401}_head.log; echo "fuzz: $secs"
exit $rc
