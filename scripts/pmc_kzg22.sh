cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/pmc_kzg
ARGS="bench.py --workload kzg22 --steps 3 --warmup 1 --no-cpu-baseline"
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace -d gpurun_out/pmc_kzg/$tag -o p --output-format csv -- python3 $ARGS > gpurun_out/pmc_kzg/$tag.json 2> gpurun_out/pmc_kzg/$tag.err
done
ls gpurun_out/pmc_kzg
