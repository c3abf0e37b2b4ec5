#!/bin/bash
# Everything the round's profiles/ are refreshed from, in one GPU-box call: bench lines, rocprofv3 kernel stats, PMC
# passes, short-row profiles, end-to-end latency from text, concurrency, fuzz.  Outputs under gpurun_out/.
# Back in the authoring container, bench lines go into profiles/ ONLY through scripts/evidence_keep.py, which refuses a
# line whose identity.source_sha16 is not this tree's (and two evidence files with equal bytes).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash scripts/profile_round.sh
bash scripts/pmc_round.sh
# the traffic figure bench.py quotes (profiles/pmc_traffic.json) against THIS pass: exit 3 = stale (drift > 2 %)
cp profiles/pmc_traffic.json gpurun_out/prof/pmc_traffic.json
python scripts/pmc_summarise.py gpurun_out/pmc gpurun_out/prof/${ROUND:-r05}_msm20_pmc.csv > gpurun_out/prof/pmc_drift.log 2>&1; echo "pmc_summarise rc=$?" >> gpurun_out/prof/pmc_drift.log
cat gpurun_out/prof/pmc_drift.log
bash scripts/rows_profile.sh
python scripts/e2e_latency.py > gpurun_out/prof/e2e_latency.json 2> gpurun_out/prof/e2e_latency.err
python scripts/concurrency_throughput.py > gpurun_out/prof/concurrency.json 2> gpurun_out/prof/concurrency.err
python tests/fuzz_gpu.py ${FUZZ_S:-300} ${FUZZ_SEED:-20261007} > gpurun_out/prof/fuzz.log 2>&1
tail -2 gpurun_out/prof/fuzz.log
