#!/bin/bash
# PMC passes for the 2^22 commit+open (NTT / opening / sort / accumulate traffic and instruction counts), one rocprofv3
# --pmc pass per counter group, kernel-trace only.  Summarise with scripts/pmc_summarise.py gpurun_out/pmc_kzg22 <csv>.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/pmc_kzg22
ARGS="bench.py --workload kzg22 --steps 3 --warmup 1 --no-cpu-baseline"
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace -d gpurun_out/pmc_kzg22/$tag -o p --output-format csv -- python3 $ARGS > gpurun_out/pmc_kzg22/$tag.json 2> gpurun_out/pmc_kzg22/$tag.err
done
ls gpurun_out/pmc_kzg22/*/ | head
