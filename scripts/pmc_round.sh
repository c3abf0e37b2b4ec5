#!/bin/bash
# PMC passes for the default bench (2^20 MSM): one rocprofv3 --pmc pass per counter group, kernel-trace only
# (MI355X_MICROARCH.md "rocprofv3 PMC slots": FETCH_SIZE and WRITE_SIZE do not fit one pass).  Run on the GPU box:
#   bash scripts/pmc_round.sh ; python scripts/pmc_summarise.py gpurun_out/pmc profiles/<round>_msm20_pmc.csv
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/pmc
ARGS="bench.py --steps 3 --warmup 1 --headline-only"
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace -d gpurun_out/pmc/$tag -o p --output-format csv -- python3 $ARGS > gpurun_out/pmc/$tag.json 2> gpurun_out/pmc/$tag.err
done
ls gpurun_out/pmc/*/
