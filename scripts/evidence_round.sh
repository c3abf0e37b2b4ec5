#!/bin/bash
# Everything the round's profiles/ are refreshed from, in one GPU-box call: bench lines, rocprofv3 kernel stats, PMC
# passes, short-row profiles, end-to-end latency from text, concurrency, start time, validator step, the full-size two-rank
# self-test of the multi-rank bench on one GPU, fuzz.  Outputs under gpurun_out/.
# Back in the authoring container, bench lines go into profiles/ ONLY through scripts/evidence_keep.py, which refuses a
# line whose identity.source_sha16 is not this tree's (and two evidence files with equal bytes).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export ROUND=${ROUND:-r06}
bash scripts/profile_round.sh
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/prof/bench_driver_style.json 2> gpurun_out/prof/bench_driver_style.err
bash scripts/pmc_round.sh
# the traffic figure bench.py quotes (profiles/pmc_traffic.json) against THIS pass: exit 3 = stale (drift > 2 %)
cp profiles/pmc_traffic.json gpurun_out/prof/pmc_traffic.json
python scripts/pmc_summarise.py gpurun_out/pmc gpurun_out/prof/${ROUND}_msm20_pmc.csv > gpurun_out/prof/pmc_drift.log 2>&1; echo "pmc_summarise rc=$?" >> gpurun_out/prof/pmc_drift.log
cat gpurun_out/prof/pmc_drift.log
bash scripts/rows_profile.sh
python scripts/e2e_latency.py > gpurun_out/prof/e2e_latency.json 2> gpurun_out/prof/e2e_latency.err
python scripts/concurrency_throughput.py > gpurun_out/prof/concurrency.json 2> gpurun_out/prof/concurrency.err
python scripts/start_time.py --out gpurun_out/prof/start_time.json > gpurun_out/prof/start_time.log 2>&1
python scripts/validator_step.py --out gpurun_out/prof/validator_step.json > gpurun_out/prof/validator_step.log 2>&1
# NOT a scaling number: the launcher-less N = 2 flow at FULL sizes with both ranks on this one GPU (RCCL refuses a shared
# device: the library's preflight fails on both ranks and the run falls back to the gloo exchange -- the fallback path itself)
BENCH_ONE_GPU=1 python bench.py --gpus 2 --steps 10 --warmup 3 > gpurun_out/prof/bench_two_ranks_one_gpu.json 2> gpurun_out/prof/bench_two_ranks_one_gpu.err
# the library's own collective on a one-rank communicator, all three workloads
BENCH_FORCE_DIST=1 MASTER_PORT=29633 python bench.py --steps 20 --warmup 5 > gpurun_out/prof/bench_one_rank_library_collective.json 2> gpurun_out/prof/bench_one_rank_library_collective.err
python tests/fuzz_gpu.py ${FUZZ_S:-300} ${FUZZ_SEED:-20261401} > gpurun_out/prof/fuzz.log 2>&1
tail -2 gpurun_out/prof/fuzz.log
